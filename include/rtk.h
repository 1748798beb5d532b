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

#include <stdbool.h>   /* bool in the trace entry points        */
#include <stddef.h>    /* size_t                                */
#include <stdint.h>    /* fixed-width fields of the scene blob  */

#ifdef __cplusplus
extern "C" {           /* plain C ABI (reference rtk.h:7-9)     */
#endif

/* "Infinity" used for open ray intervals; it is a large finite float, not IEEE inf
 * (reference rtk.h:11). The literal must stay exactly this one: it rounds to
 * 0x7f7ffffd, two ulps below FLT_MAX. */
#define RTK_INF (3.402823e+38f)

/* Scalar type of positions. Only the 32-bit build exists (reference rtk.h:13). */
typedef float rtk_real;

/* 3-vector addressable by name or by axis number (reference rtk.h:15-22). 12 bytes. */
typedef struct rtk_vec3 {
	union {
		struct { rtk_real x, y, z; };   /* by name              */
		rtk_real v[3];                  /* by axis: 0=x 1=y 2=z */
	};
} rtk_vec3;

/* A vertex as stored in a scene and returned in a hit (reference rtk.h:24-27). 16 bytes. */
typedef struct rtk_vertex {
	rtk_vec3 position;   /* offset 0                                        */
	uint32_t index;      /* offset 12: the vertex's index in the caller's mesh */
} rtk_vertex;

/* Ray: origin + t*direction for t in the OPEN interval (min_t, max_t). The direction
 * need not be normalised; t is measured in units of it (reference rtk.h:29-34,
 * interval test rtk.c:354). 32 bytes. */
typedef struct rtk_ray {
	rtk_vec3 origin;      /* offset 0  */
	rtk_vec3 direction;   /* offset 12 */
	rtk_real min_t;       /* offset 24: hits at t <= min_t are ignored */
	rtk_real max_t;       /* offset 28: hits at t >= max_t are ignored */
} rtk_ray;

/* Closest hit (reference rtk.h:36-43). 68 bytes. */
typedef struct rtk_hit {
	rtk_real t;                /* distance along the ray, in units of |direction|          */
	rtk_real u;                /* barycentric weight of vertex[0] (rtk.c:362-375) -- NOT   */
	rtk_real v;                /* ... of vertex[1]; this is not the Moller-Trumbore pair   */
	rtk_vertex vertex[3];      /* the triangle's vertices with the caller's vertex indices */
	uint32_t mesh_index;       /* which mesh of the scene description                      */
	uint32_t triangle_index;   /* which triangle INSIDE that mesh (rtk.c:1168-1169)        */
} rtk_hit;

/* Element type of a caller buffer (reference rtk.h:45-52). */
typedef enum rtk_type {
	RTK_TYPE_DEFAULT,   /* 0: positions -> REAL, indices -> U32 */
	RTK_TYPE_F32,       /* 1 */
	RTK_TYPE_F64,       /* 2 */
	RTK_TYPE_REAL,      /* 3: rtk_real, i.e. F32 here */
	RTK_TYPE_U16,       /* 4 */
	RTK_TYPE_U32,       /* 5 */
} rtk_type;

/* Strided view of caller memory (reference rtk.h:54-58). 24 bytes. */
typedef struct rtk_buffer {
	const void *data;   /* first element                                              */
	size_t stride;      /* bytes between triples; 0 = tightly packed (rtk.c:1037, 1086) */
	rtk_type type;      /* element type                                               */
} rtk_buffer;

typedef struct rtk_mesh rtk_mesh;

/* Optional data sources called by the builder instead of reading buffers
 * (reference rtk.h:61-62; call sites rtk.c:1031, 1075): at most 128 triangles per call. */
typedef void rtk_position_callback_fn(void *user, const rtk_mesh *mesh, rtk_vec3 *dst, const uint32_t *indices, size_t count);
typedef void rtk_index_callback_fn(void *user, const rtk_mesh *mesh, uint32_t *dst, size_t offset, size_t count);

/* One triangle mesh of a scene (reference rtk.h:64-76). 96 bytes. All memory is borrowed
 * until the build has run. */
struct rtk_mesh {
	void *user;                              /* offset 0: free for the caller                     */
	size_t num_triangles;                    /* offset 8                                          */

	rtk_buffer position;                     /* offset 16: default type REAL                      */
	rtk_buffer index;                        /* offset 40: default type U32; data == NULL means    */
	                                         /* triangle i uses vertices 3i..3i+2 (rtk.c:1061-1068) */

	rtk_position_callback_fn *position_cb;   /* offset 64: if set, replaces `position`            */
	void *position_cb_user;                  /* offset 72                                         */

	rtk_index_callback_fn *index_cb;         /* offset 80: if set, replaces `index`               */
	void *index_cb_user;                     /* offset 88                                         */
};

/* Header of a built scene. A scene is ONE position-independent byte blob and this
 * struct is its first 56 bytes (reference rtk.h:78-89, written at rtk.c:1737-1755);
 * all offsets are bytes from the start of the blob. Rest of the layout: DESIGN.md. */
typedef struct rtk_scene {
	char magic[8];            /* "\0RTK\r\n\x1a\n"                      */
	uint16_t endian;          /* 0xaabb as written by the producer       */
	uint8_t sizeof_real;      /* 4                                       */
	uint8_t pad_0;            /* 0                                       */
	uint32_t version;         /* 1                                       */
	uint32_t pad_1;           /* 0                                       */
	uint64_t size_in_bytes;   /* whole blob                              */
	uint64_t node_offset;     /* 128: also the hard-wired root (rtk.c:569) */
	uint64_t leaf_offset;     /* leaf section, starts with the null leaf */
	uint64_t vertex_offset;   /* vertex groups                           */
} rtk_scene;

typedef struct rtk_build rtk_build;         /* opaque: a build in progress */
typedef struct rtk_task rtk_task;           /* a unit of build work        */
typedef struct rtk_task_ctx rtk_task_ctx;   /* opaque: per-call context    */

/* Text log sink, called from build tasks (reference rtk.h:95, rtk.c:686-696). */
typedef void rtk_log_fn(void *user, rtk_build *build, const char *str);

/* What to build (reference rtk.h:97-105). 32 bytes. Copied by value at rtk_start_build;
 * the mesh array and every buffer it points at stay borrowed (rtk.c:1661). */
typedef struct rtk_scene_desc {

	const rtk_mesh *meshes;   /* offset 0           */
	size_t num_meshes;        /* offset 8           */

	rtk_log_fn *log_fn;       /* offset 16: may be NULL */
	void *log_user;           /* offset 24          */

} rtk_scene_desc;

/* Unit of build work handed to the caller's scheduler (reference rtk.h:108-115). 40 bytes.
 * fn/index/arg are opaque to the caller. */
typedef void rtk_task_fn(const rtk_task *task, rtk_task_ctx *ctx);
struct rtk_task {
	rtk_build *build;   /* the build this task belongs to */
	rtk_task_fn *fn;    /* library-internal entry         */
	double cost;        /* scheduling hint                */
	size_t index;       /* library-internal               */
	uintptr_t arg;      /* library-internal               */
};

/* Candidate-hit filter (reference rtk.h:117): return true to accept the hit. */
typedef bool rtk_filter_fn(void *user, const rtk_ray *ray, const rtk_hit *hit);

/* ---- build (reference rtk.h:119-127) ---- */

/* Begin a build. With first_task != NULL the caller drives the task graph through
 * rtk_run_task; with NULL the whole build runs before returning. NULL on failure. */
rtk_build *rtk_start_build(const rtk_scene_desc *desc, rtk_task *first_task);

/* Run one task; tasks it spawns are written to queue[0..ret). */
size_t rtk_run_task(const rtk_task *task, rtk_task *queue, size_t queue_size);

/* Bytes the finished scene blob needs. */
size_t rtk_get_build_size(const rtk_build *build);

/* Emit the scene into caller memory: NULL, and the build stays alive, if `size` is too small. */
rtk_scene *rtk_finish_build_to(rtk_build *build, void *buffer, size_t size);

/* Emit the scene into memory owned by the library. Both finish calls free the build on success. */
rtk_scene *rtk_finish_build(rtk_build *build);

/* start + finish in one call. */
rtk_scene *rtk_build_scene(const rtk_scene_desc *desc);

/* Release a scene returned by rtk_finish_build or rtk_build_scene. */
void rtk_free_scene(rtk_scene *scene);

/* ---- trace (reference rtk.h:129-130) ---- */

/* Closest hit of one ray. Returns false and leaves *hit untouched on a miss. */
bool rtk_trace_ray(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit);

/* As rtk_trace_ray, but every candidate is offered to filter first. */
bool rtk_trace_ray_filter(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit, rtk_filter_fn *filter, void *filter_user);

#ifdef __cplusplus
}
#endif

#endif /* RTK_H_MI355X_RESTATED */
