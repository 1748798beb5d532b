/*
 * rtk_amd.h -- additive batch / device entry points of librtk_amd.so (MI355X, gfx950).
 *
 * The reference's trace call is per ray and synchronous (reference rtk.h:129-130,
 * rtk.c:543-577); a GPU needs whole batches and device residency. Everything here is
 * NEW surface next to the nine unchanged rtk.h symbols. All functions are plain C:
 * pointers + sizes, no C++ or torch types. `stream` arguments are a hipStream_t passed
 * as void* (NULL = the default stream); "d_" pointers are DEVICE memory.
 *
 * Which reference interface each entry point stands in for:
 *   rtk_dev_scene_upload   : the hand-over of a built scene blob to the tracer. In the
 *                            reference that is just the rtk_scene* itself (rtk.c:546);
 *                            here the blob (format: SURVEY.md appendix A, reader
 *                            rtk.c:181-193, 457-465) is validated and re-laid-out in HBM.
 *   rtk_dev_scene_build    : rtk_build_scene / rtk_start_build..rtk_finish_build
 *                            (rtk.h:119-126; rtk.c:1625-1792) with the result left
 *                            device-resident (LBVH on the GPU instead of binned SAH tasks).
 *   rtk_dev_scene_export   : rtk_finish_build_to (rtk.h:123; rtk.c:1732-1774): emits the
 *                            device BVH as a reference-format blob.
 *   rtk_dev_trace_rays     : a loop of rtk_trace_ray over a ray array (rtk.h:129).
 *   rtk_dev_trace_rays_any : a loop of rtk_trace_ray_filter with a filter that accepts
 *                            the first candidate (rtk.h:117, 130; stub at rtk.c:579-582).
 *   rtk_dev_expand_hits    : the *hit = rt.hit copy-out of rtk.c:571-573 (full rtk_hit).
 *   rtk_trace_rays         : host-pointer convenience over the three above.
 */
#ifndef RTK_AMD_H
#define RTK_AMD_H

#include "rtk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes (0 = success). rtk_amd_last_error() describes the last failure on the
 * calling thread. */
enum {
	RTK_AMD_OK = 0,
	RTK_AMD_ERR_NO_DEVICE = -1,   /* no HIP device / driver */
	RTK_AMD_ERR_BAD_ARG = -2,
	RTK_AMD_ERR_OOM = -3,
	RTK_AMD_ERR_HIP = -4,         /* a HIP runtime call failed */
	RTK_AMD_ERR_BAD_SCENE = -5,   /* blob failed validation */
	RTK_AMD_ERR_UNSUPPORTED = -6,
};

const char *rtk_amd_last_error(void);
int rtk_amd_device_count(void);
int rtk_amd_set_device(int device);

/* Device-resident scene. */
typedef struct rtk_dev_scene rtk_dev_scene;

/* Compact hit record, 16 bytes. prim is the GLOBAL primitive id: the triangle's position
 * in concatenated mesh order (mesh_base[mesh_index] + triangle_index, cf. rtk.c:1131-1178);
 * RTK_PRIM_NONE on a miss (then t = ray.max_t, u = v = 0). u, v as in rtk_hit. */
typedef struct rtk_hit_record {
	float t, u, v;
	uint32_t prim;
} rtk_hit_record;
#define RTK_PRIM_NONE 0xffffffffu

typedef struct rtk_dev_scene_info {
	uint64_t num_triangles;
	uint64_t num_meshes;
	uint64_t num_nodes;        /* 4-wide nodes, 128 B each */
	uint64_t node_bytes;
	uint64_t triangle_bytes;   /* 48 B per triangle */
	uint64_t total_device_bytes;
	uint32_t max_depth;        /* deepest 4-wide node level (root = 1) */
	uint32_t stack_entries;    /* traversal stack entries a ray can need */
	double build_ms;           /* wall time inside rtk_dev_scene_build (0 for uploads) */
} rtk_dev_scene_info;

/* Trace options; pass NULL for defaults. */
typedef struct rtk_trace_opts {
	uint32_t struct_size;      /* sizeof(rtk_trace_opts) */
	uint32_t flags;            /* RTK_TRACE_* */
	uint32_t image_width;      /* if both non-zero and width*height == n: rays are a   */
	uint32_t image_height;     /* row-major image; lanes are mapped to 8x8 pixel tiles. Without it (or with NULL options) a closest-hit batch of whole
	                              64x64-pixel blocks is LOOKED AT: a regular step from ray to ray that jumps at the same distance every time is an image */
	uint32_t refill_min;       /* 0 = default; idle lanes needed before a wave refills  */
	uint32_t blocks_per_cu;    /* 0 = default; persistent grid size                     */
	uint32_t node_exit;        /* 0 = default; see DESIGN.md 3.1 (divergence control)   */
} rtk_trace_opts;
#define RTK_TRACE_STATIC    1u   /* one fixed ray per lane, no persistent refill (A/B only) */
#define RTK_TRACE_NO_PACKET 2u   /* image-shaped batch, but use the per-lane kernel (A/B only) */
#define RTK_TRACE_EXACT_NODES 8u  /* per-lane kernels: read the 128 B exact nodes instead of the 64 B compressed ones (A/B only) */
#define RTK_TRACE_NO_ASM 16u      /* C++ kernels only, not the hand-written ones (rtk_packet_hot.S, rtk_lane_hot.S) (A/B only) */
#define RTK_TRACE_NO_ENTRIES 32u  /* image-shaped batch: every tile starts at the root, no per-block entry lists (A/B only) */
#define RTK_TRACE_NO_BEAM 64u     /* image-shaped batch: rtk_packet_hot (per-lane slab tests) instead of the beam kernels (A/B only) */
#define RTK_TRACE_ONE_TILE_BEAM 128u /* image-shaped batch: rtk_packet_beam (one tile per wave) instead of rtk_packet_beam2 (A/B only) */
#define RTK_TRACE_NO_DETECT 256u   /* no image hint given: do not look whether the batch is a row-major image anyway (A/B; saves the two small launches
                                     and the wait of the look when the caller knows its rays are not an image) */
#define RTK_TRACE_SORT_RAYS 4u   /* reorder the batch by (origin cell, direction octant) before tracing; hits
                                    still land in input order. Pays off for large incoherent batches. */

/* Visit counters of the counting build (algorithmic-bytes model). */
typedef struct rtk_trace_counters {
	uint64_t rays;
	uint64_t nodes;            /* 128 B node records fetched by rays  */
	uint64_t leaves;
	uint64_t triangles;        /* 48 B triangle records fetched       */
	uint64_t hits;
	uint64_t stack_spills;     /* pushes that went past the LDS stack */
	uint64_t wave_node_steps;      /* wave-level node-loop trips (divergence diagnostics) */
	uint64_t wave_triangle_steps;  /* wave-level triangle-loop trips */
	uint64_t wave_rays;            /* reserved */
} rtk_trace_counters;

/* -- scenes -- */
/* Validates the blob (every offset range-checked without wrap-around; it must be a tree: a node or leaf
 * reached twice is refused) and lays it out in HBM. rtk_dev_scene_upload trusts scene->size_in_bytes to be
 * readable, as the reference's bare rtk_scene* forces it to; a loader of untrusted files uses
 * rtk_dev_scene_upload_buffer, which also checks the header against the buffer size. */
rtk_dev_scene *rtk_dev_scene_upload(const rtk_scene *scene);
rtk_dev_scene *rtk_dev_scene_upload_buffer(const void *blob, size_t blob_bytes);
rtk_dev_scene *rtk_dev_scene_build(const rtk_scene_desc *desc);
void rtk_dev_scene_free(rtk_dev_scene *ds);
int rtk_dev_scene_get_info(const rtk_dev_scene *ds, rtk_dev_scene_info *info);
/* mesh_base[0..num_meshes] (prefix sums of per-mesh triangle counts) */
int rtk_dev_scene_mesh_base(const rtk_dev_scene *ds, uint64_t *out, size_t capacity);
/* out[slot] = global primitive id of the triangle stored at that slot (device order: leaf by leaf;
 * Morton order for device-built scenes). capacity in elements; returns the triangle count or < 0. */
long long rtk_dev_scene_primitive_order(const rtk_dev_scene *ds, uint32_t *out, size_t capacity);
size_t rtk_dev_scene_export_size(const rtk_dev_scene *ds);
rtk_scene *rtk_dev_scene_export(const rtk_dev_scene *ds, void *buffer, size_t size);

/* Structural check of a device scene, run on the device (the loader/validator the reference lacks,
 * SURVEY.md section 5; blob-level checks happen in rtk_dev_scene_upload). Every child box must contain
 * what is below it, every triangle slot must sit in exactly one leaf, every node but the root must be
 * referenced exactly once by an earlier node, leaf headers must be well formed (1..63 triangles,
 * rtk.c:188). Returns RTK_AMD_OK when all error counts are zero, RTK_AMD_ERR_BAD_SCENE otherwise;
 * `loose_boxes` (a box that contains its contents without being their exact union) is legal and only
 * reported. content_hash covers nodes and triangle records: equal for two builds of the same input. */
typedef struct rtk_dev_scene_check {
	uint64_t nodes_checked, leaves_checked, triangles_checked;
	uint64_t box_violations;        /* child box does not contain its subtree / empty slot can be hit */
	uint64_t loose_boxes;
	uint64_t bad_references;        /* child index out of range or not after its parent */
	uint64_t leaf_format_errors;
	uint64_t triangles_missing, triangles_duplicated;
	uint64_t nodes_unreachable, nodes_shared;
	uint64_t primitive_id_errors;   /* id out of range, repeated, or prim -> slot table inconsistent */
	uint64_t compressed_node_errors; /* a 64 B compressed node whose decoded child box does not contain the exact one */
	uint64_t first_bad_index;       /* smallest node / slot index that raised an error, ~0 if none */
	uint64_t content_hash;
} rtk_dev_scene_check;
int rtk_dev_scene_validate(const rtk_dev_scene *ds, rtk_dev_scene_check *out);

/* Which builder rtk_start_build / rtk_build_scene use (rtk.h:119-126). Default: the device LBVH build, a graph of
 * ONE task. RTK_AMD_BUILDER_CPU_TASKS selects the reference's caller-scheduled task graph on the CPU (rtk.c:1362-1507:
 * <= 128 triangle-setup tasks, one binned-SAH task per node, vertex-group finalisation), so a host's own scheduler
 * keeps driving a real graph from any number of threads; it needs no GPU to BUILD (tracing the blob still does).
 * Never selected implicitly. Environment RTK_AMD_BUILDER=cpu is read once if this was never called. */
enum { RTK_AMD_BUILDER_DEVICE = 0, RTK_AMD_BUILDER_CPU_TASKS = 1 };
int rtk_amd_set_builder(int builder);
/* Where the two PER-RAY symbols of rtk.h (rtk_trace_ray, rtk_trace_ray_filter; reference rtk.h:129-130) run. HOST (default): on
 * the calling thread, from the caller's blob -- a synchronous call for one ray cannot drive a GPU (a launch and its completion
 * are ~7 us before any work; the reference's call is under 1 us), SURVEY.md 8b serves this one symbol on the CPU. GPU: a batch
 * of one through the one-ray kernel (~25 us). Both return the same bytes. Batch entry points are not affected by this and have
 * no host form. Environment RTK_AMD_PER_RAY=gpu is read once if this was never called. */
enum { RTK_AMD_PER_RAY_HOST = 0, RTK_AMD_PER_RAY_GPU = 1 };
int rtk_amd_set_per_ray(int where);
int rtk_amd_get_builder(void);

/* Device builds draw their temporaries from one workspace per device that is kept between builds
 * (about 330 bytes per triangle); this releases it. */
void rtk_amd_release_workspace(void);

/* -- batches; asynchronous on `stream` --
 * Thread-safe: a scene is never modified by a launch. What a launch writes besides its outputs (work-queue
 * heads, the global part of the traversal stacks) lives in a scratch set per (scene, stream), so launches on
 * different streams or from different host threads never share state; launches on one stream are ordered by
 * the stream (reference: rtk_trace_ray is a pure function of a const scene, rtk.c:543-577). */
int rtk_dev_trace_rays(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	rtk_hit_record *d_hits, const rtk_trace_opts *opts, void *stream);
/* (any-hit with an image hint of whole 64x64-pixel blocks, image_width >= 128: traced by the packet kernels in their any-hit form
 * -- a ray is retired at its first hit -- the same flags at about three times the per-lane rate on rays that run side by side;
 * other shapes ignore the hint) */
int rtk_dev_trace_rays_any(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	uint8_t *d_occluded, const rtk_trace_opts *opts, void *stream);
int rtk_dev_expand_hits(const rtk_dev_scene *ds, const rtk_hit_record *d_records, size_t n,
	rtk_hit *d_hits, uint8_t *d_mask, void *stream);
/* Built-in candidate filters evaluated on the device (reference rtk.h:117,130: rtk_filter_fn /
 * rtk_trace_ray_filter, a stub at rtk.c:579-582; arbitrary C callbacks cannot run on the GPU, these cover
 * the common ones). A candidate hit is considered only if EVERY filter that is set accepts it; the result is
 * the closest accepted candidate (closest-hit call) or "is there an accepted candidate" (any-hit call).
 *   d_mesh_mask    bit m (word m/32, bit m%32) set = triangles of mesh m are visible; meshes >= mesh_mask_bits are not
 *   d_ignore_prim  one per ray: global primitive id that is never a candidate (self-intersection); RTK_PRIM_NONE = none
 *   d_after        one per ray: only candidates that come AFTER (t, prim) in lexicographic (t, prim) order are
 *                  considered; prim = RTK_PRIM_NONE switches it off for that ray. Feeding a ray's previous result
 *                  back enumerates ALL candidates of the ray in order, equal-t ones included -- the device half of
 *                  rtk_trace_rays_filter's host-callback loop.
 * All pointers are device memory and optional (NULL). */
typedef struct rtk_dev_filter {
	uint32_t struct_size;               /* sizeof(rtk_dev_filter) */
	uint32_t mesh_mask_bits;
	const uint32_t *d_mesh_mask;
	const uint32_t *d_ignore_prim;
	const rtk_hit_record *d_after;
} rtk_dev_filter;
int rtk_dev_trace_rays_filtered(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	rtk_hit_record *d_hits, const rtk_dev_filter *filter, const rtk_trace_opts *opts, void *stream);
int rtk_dev_trace_rays_any_filtered(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	uint8_t *d_occluded, const rtk_dev_filter *filter, const rtk_trace_opts *opts, void *stream);
/* Waits for `stream` and reports whether a launch of this scene on it overflowed a traversal stack
 * (RTK_AMD_ERR_BAD_SCENE; impossible for a validated tree -- the push is dropped, never written out of bounds). */
int rtk_dev_trace_status(const rtk_dev_scene *ds, void *stream);

/* Is the device-resident batch a row-major image (a regular step from ray to ray -- origin and direction -- that jumps at the same
 * distance every time)? *width, *height = its shape, or 0, 0. What rtk_dev_trace_rays does by itself for a closest-hit batch that
 * comes without the image hint (two small launches and a wait for `stream`, ~20 us): a host that traces many frames of one
 * shape asks once and passes the shape in rtk_trace_opts afterwards. A wrong answer can only cost speed, never a hit. */
int rtk_dev_detect_image(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n, uint32_t *width, uint32_t *height, void *stream);

/* Same result as rtk_dev_trace_rays, plus visit counts. Synchronous; not for timing. */
int rtk_dev_trace_rays_counted(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	rtk_hit_record *d_hits, const rtk_trace_opts *opts, rtk_trace_counters *out);

int rtk_dev_trace_rays_any_counted(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	uint8_t *d_occluded, const rtk_trace_opts *opts, rtk_trace_counters *out);

/* Step counts of the hand-written packet kernel ITSELF: the launch runs rtk_packet_count2, which is rtk_packet_beam2.S assembled
 * with three scalar counters per pair of tiles ("visit counts come from a counting build of the same kernel"), and
 * the counting build of the C++ packet kernel on the tiles it hands back. Same records as rtk_dev_trace_rays. Synchronous; not
 * for timing. RTK_AMD_ERR_UNSUPPORTED if the batch would not run on rtk_packet_beam2 (no image hint, big leaves, ...). */
typedef struct rtk_packet_counters {
	uint64_t tiles;                      /* 8x8-pixel tiles of the batch (n / 64) */
	uint64_t pairs;                      /* pairs of tiles walked by a wave (handed-back pairs included, with the steps they took until then) */
	uint64_t node_steps;                 /* 128 B nodes fetched: one per node step of a PAIR */
	uint64_t triangles_fetched;          /* 48 B triangle records fetched: one per triangle of a leaf a pair enters */
	uint64_t triangle_group_tests;       /* triangle tests: one per (triangle, group of 64 rays whose beam reaches the leaf) */
	uint64_t tiles_handed_back;          /* tiles traced again from the start by the C++ packet kernel */
	uint64_t handed_back_node_steps;     /* ... its wave-level node steps (one tile per wave) */
	uint64_t handed_back_triangle_steps; /* ... and triangle steps */
} rtk_packet_counters;
int rtk_dev_trace_rays_packet_counted(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	rtk_hit_record *d_hits, const rtk_trace_opts *opts, rtk_packet_counters *out);

/* -- several GPUs of one node, one process --
 * Rays shard, nothing else: the scene is replicated (the deterministic build or the upload runs on every GPU),
 * shard r of R owns the contiguous range rtk_amd_shard_range(n, r, R) of the batch, and the only exchange is the
 * gather of the 16-byte records, each GPU copying straight to the final place on its own stream (its own xGMI
 * link into the root), piece by piece while the rest of its shard is still being traced. Synchronous calls. */
typedef struct rtk_mgpu rtk_mgpu;
void rtk_amd_shard_range(size_t n, int rank, int num_shards, size_t *first, size_t *count);
rtk_mgpu *rtk_mgpu_create(const int *devices, int num_devices);   /* NULL / 0: all devices; an id may repeat (virtual shards) */
void rtk_mgpu_destroy(rtk_mgpu *m);
int rtk_mgpu_num_devices(const rtk_mgpu *m);
const rtk_dev_scene *rtk_mgpu_scene(const rtk_mgpu *m, int index);
int rtk_mgpu_build(rtk_mgpu *m, const rtk_scene_desc *desc);      /* rtk_dev_scene_build on every GPU */
int rtk_mgpu_upload(rtk_mgpu *m, const rtk_scene *scene);         /* rtk_dev_scene_upload on every GPU */
/* host rays in, host records out (records[i] belongs to rays[i]) */
int rtk_mgpu_trace_rays(rtk_mgpu *m, const rtk_ray *rays, size_t n, rtk_hit_record *records, const rtk_trace_opts *opts);
/* device-resident shards: d_rays[r] / d_records[r] (counts[r] elements) live on GPU r of the context; if d_gathered
 * is not NULL it is memory of GPU `root_index` that receives all records, shard after shard. opts applies per shard
 * (an image-shaped shard is traced in bands of whole tile rows by the packet kernel). */
int rtk_mgpu_trace_rays_device(rtk_mgpu *m, const rtk_ray *const *d_rays, const size_t *counts, rtk_hit_record *const *d_records,
	rtk_hit_record *d_gathered, int root_index, const rtk_trace_opts *opts);
/* The same with a STRIPED exchange instead of a gather onto one root (which is bound by that root's links: eight GPUs
 * gathered onto one deliver ~2x one GPU): GPU j ends up with stripe j of EVERY shard in d_striped[j] (memory of GPU j),
 * segments in shard order; stripe j of a shard of c records is rtk_amd_shard_range(c, j, R) of it, and
 * rtk_mgpu_striped_segment(counts, R, shard, stripe, &first, &count) says where it sits in d_striped[stripe]. Every GPU
 * sends 1/R of each piece over each of its links while the rest of its shard is still being traced. */
int rtk_mgpu_trace_rays_device_striped(rtk_mgpu *m, const rtk_ray *const *d_rays, const size_t *counts, rtk_hit_record *const *d_records,
	rtk_hit_record *const *d_striped, const rtk_trace_opts *opts);
void rtk_mgpu_striped_segment(const size_t *counts, int num_shards, int shard, int stripe, size_t *first, size_t *count);

/* -- host-pointer convenience (PCIe-inclusive, synchronous) --
 * Closest hits of n rays against a scene blob. hits[i] is written where the ray hit
 * (left untouched on a miss, like rtk_trace_ray); hit_mask[i] (optional) gets 0/1.
 * Returns the number of hits, or (size_t)-1 on error. The device copy of the blob is cached per (scene pointer,
 * current device) until rtk_free_scene / rtk_amd_forget_scene. Every way this library itself writes a blob to an
 * address (rtk_finish_build, rtk_finish_build_to with either builder) drops what was cached for that address. A
 * caller that overwrites a blob in place by its own means MUST call rtk_amd_forget_scene(address) before tracing it
 * again. As a safety net every lookup re-checks the header, the root node and one 4 KB stripe of the blob (a different
 * one each time, against hashes of all stripes taken at upload): another scene at the same address is noticed at
 * once, a small in-place edit within size / 4 KB lookups -- not immediately. A blob in caller-owned memory keeps its
 * device copy until rtk_amd_forget_scene is called for it. Each calling thread uses its own stream and staging
 * buffers. rtk_trace_ray / rtk_trace_ray_filter (no error channel, and `false` means "miss" to a host that knows only
 * rtk.h): a failure is never silent -- it prints one line on stderr (rate-limited after the first eight) and sets
 * rtk_amd_last_error(). A lone failure that may pass (out of memory, an interrupted launch) then returns false; one
 * that every later call would repeat (no usable GPU, a scene that does not validate), or a SECOND failure in a row on
 * the calling thread (a stream error that sticks), abort()s unless RTK_AMD_SOFT_ERRORS is set in the environment. */
size_t rtk_trace_rays(const rtk_scene *scene, const rtk_ray *rays, size_t n, rtk_hit *hits, uint8_t *hit_mask);
/* Batch form of rtk_trace_ray_filter (rtk.h:130) with a host callback: per ray the closest candidate that
 * `filter` accepts. Every candidate of a ray is offered, in increasing (t, primitive id) order (equal-t ones
 * included), until one is accepted. The batch runs in rounds: one launch collects the k closest candidates of
 * every undecided ray (k = 4 for 16 384 rays ... 64 for a single ray); rays whose k candidates were all rejected
 * continue after the last one in the next round. */
size_t rtk_trace_rays_filter(const rtk_scene *scene, const rtk_ray *rays, size_t n, rtk_hit *hits, uint8_t *hit_mask,
	rtk_filter_fn *filter, void *filter_user);
void rtk_amd_forget_scene(const rtk_scene *scene);

#ifdef __cplusplus
}
#endif
#endif /* RTK_AMD_H */
