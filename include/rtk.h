/*
 * rtk.h -- public C interface of the rtk ray-tracing kernel, as served by the
 * MI355X-native implementation in this repository (librtk_amd.so).
 *
 * This header is a clean-room restatement of the reference interface
 * (reference rtk.h:11-130): every type has the same name, size, field order and
 * meaning, and the nine entry points have the same signatures, so a C host
 * program written against the reference header compiles and links unchanged.
 * Sizes (x86-64 SysV): rtk_vec3 12, rtk_vertex 16, rtk_ray 32, rtk_hit 68,
 * rtk_buffer 24, rtk_mesh 96, rtk_scene 56, rtk_scene_desc 32, rtk_task 40 --
 * checked by static asserts in rtk_amd/csrc/rtk_layout_check.h.
 *
 * The batch / device entry points that a GPU needs in addition (the reference
 * only has a per-ray synchronous call) live in rtk_amd.h and are purely additive.
 */
#ifndef RTK_H_MI355X_RESTATED
#define RTK_H_MI355X_RESTATED

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* "Infinity" used for open ray intervals; it is a large finite float, not IEEE inf
 * (reference rtk.h:11). The literal must stay exactly this one: it rounds to
 * 0x7f7ffffd, two ulps below FLT_MAX. */
#define RTK_INF (3.402823e+38f)

/* Scalar type of positions. Only the 32-bit build exists (reference rtk.h:13). */
typedef float rtk_real;

/* 3-vector addressable by name or by axis number (reference rtk.h:15-22). */
typedef struct rtk_vec3 {
	union {
		struct { rtk_real x, y, z; };
		rtk_real v[3];
	};
} rtk_vec3;

/* A vertex as stored in a scene and returned in a hit: position plus the index the
 * vertex had in the caller's mesh (reference rtk.h:24-27). 16 bytes. */
typedef struct rtk_vertex {
	rtk_vec3 position;
	uint32_t index;
} rtk_vertex;

/* Ray: origin + t*direction for t in the OPEN interval (min_t, max_t). The direction
 * need not be normalised; t is measured in units of it (reference rtk.h:29-34,
 * interval test rtk.c:354). 32 bytes. */
typedef struct rtk_ray {
	rtk_vec3 origin;
	rtk_vec3 direction;
	rtk_real min_t;
	rtk_real max_t;
} rtk_ray;

/* Closest hit (reference rtk.h:36-43). u is the barycentric weight of vertex[0] and
 * v the weight of vertex[1] (rtk.c:362-375) -- NOT the Moller-Trumbore convention.
 * mesh_index counts meshes of the scene description, triangle_index counts
 * triangles inside that mesh (rtk.c:1168-1169). 68 bytes. */
typedef struct rtk_hit {
	rtk_real t;
	rtk_real u;
	rtk_real v;
	rtk_vertex vertex[3];
	uint32_t mesh_index;
	uint32_t triangle_index;
} rtk_hit;

/* Element type of a caller buffer (reference rtk.h:45-52). */
typedef enum rtk_type {
	RTK_TYPE_DEFAULT,
	RTK_TYPE_F32,
	RTK_TYPE_F64,
	RTK_TYPE_REAL,
	RTK_TYPE_U16,
	RTK_TYPE_U32,
} rtk_type;

/* Strided view of caller memory; stride 0 means tightly packed triples
 * (reference rtk.h:54-58, defaults rtk.c:1037, 1048, 1086). */
typedef struct rtk_buffer {
	const void *data;
	size_t stride;
	rtk_type type;
} rtk_buffer;

typedef struct rtk_mesh rtk_mesh;

/* Optional data sources called by the builder instead of reading buffers
 * (reference rtk.h:61-62; call sites rtk.c:1031, 1075): at most 128 triangles per call. */
typedef void rtk_position_callback_fn(void *user, const rtk_mesh *mesh, rtk_vec3 *dst, const uint32_t *indices, size_t count);
typedef void rtk_index_callback_fn(void *user, const rtk_mesh *mesh, uint32_t *dst, size_t offset, size_t count);

/* One triangle mesh of a scene (reference rtk.h:64-76). position defaults to
 * RTK_TYPE_REAL, index to U32; index.data == NULL means triangle i uses vertices
 * 3i, 3i+1, 3i+2 (rtk.c:1061-1068). All memory is borrowed until the build ends. */
struct rtk_mesh {
	void *user;
	size_t num_triangles;

	rtk_buffer position;
	rtk_buffer index;

	rtk_position_callback_fn *position_cb;
	void *position_cb_user;

	rtk_index_callback_fn *index_cb;
	void *index_cb_user;
};

/* Header of a built scene. A scene is ONE position-independent byte blob and this
 * struct is its first 56 bytes (reference rtk.h:78-89, written at rtk.c:1737-1755);
 * all offsets are bytes from the start of the blob. Layout of the rest:
 * DESIGN.md "Scene blob". */
typedef struct rtk_scene {
	char magic[8];
	uint16_t endian;
	uint8_t sizeof_real;
	uint8_t pad_0;
	uint32_t version;
	uint32_t pad_1;
	uint64_t size_in_bytes;
	uint64_t node_offset;
	uint64_t leaf_offset;
	uint64_t vertex_offset;
} rtk_scene;

typedef struct rtk_build rtk_build;
typedef struct rtk_task rtk_task;
typedef struct rtk_task_ctx rtk_task_ctx;

/* Text log sink, called from build tasks (reference rtk.h:95, rtk.c:686-696). */
typedef void rtk_log_fn(void *user, rtk_build *build, const char *str);

/* What to build (reference rtk.h:97-105). Copied by value at rtk_start_build; the
 * mesh array and every buffer it points at stay borrowed (rtk.c:1661). */
typedef struct rtk_scene_desc {

	const rtk_mesh *meshes;
	size_t num_meshes;

	rtk_log_fn *log_fn;
	void *log_user;

} rtk_scene_desc;

/* Unit of build work handed to the caller's scheduler (reference rtk.h:108-115).
 * fn/index/arg are opaque to the caller; cost is a scheduling hint. */
typedef void rtk_task_fn(const rtk_task *task, rtk_task_ctx *ctx);
struct rtk_task {
	rtk_build *build;
	rtk_task_fn *fn;
	double cost;
	size_t index;
	uintptr_t arg;
};

/* Candidate-hit filter (reference rtk.h:117): return true to accept the hit. */
typedef bool rtk_filter_fn(void *user, const rtk_ray *ray, const rtk_hit *hit);

/* -- Build (reference rtk.h:119-127) -- */

/* Begin a build. With first_task != NULL the caller drives the task graph through
 * rtk_run_task; with NULL the whole build runs before returning. NULL on failure. */
rtk_build *rtk_start_build(const rtk_scene_desc *desc, rtk_task *first_task);

/* Run one task; tasks it spawns are written to queue[0..ret). */
size_t rtk_run_task(const rtk_task *task, rtk_task *queue, size_t queue_size);

/* Bytes the finished scene blob needs. */
size_t rtk_get_build_size(const rtk_build *build);

/* Emit the scene into caller memory (NULL and the build stays alive if size is too
 * small), or into memory owned by the library. Both free the build on success. */
rtk_scene *rtk_finish_build_to(rtk_build *build, void *buffer, size_t size);
rtk_scene *rtk_finish_build(rtk_build *build);

/* start + finish in one call; release with rtk_free_scene. */
rtk_scene *rtk_build_scene(const rtk_scene_desc *desc);
void rtk_free_scene(rtk_scene *scene);

/* -- Trace (reference rtk.h:129-130) -- */

/* Closest hit of one ray. Returns false and leaves *hit untouched on a miss. */
bool rtk_trace_ray(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit);

/* As rtk_trace_ray, but every candidate is offered to filter first. */
bool rtk_trace_ray_filter(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit, rtk_filter_fn *filter, void *filter_user);

#ifdef __cplusplus
}
#endif

#endif /* RTK_H_MI355X_RESTATED */
