/*
 * rtk_oracle.h -- CPU ORACLE for the rtk hot path. TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (bqqbarbhg/rtk @ v0,
 * rtk.c) used as the checker for the HIP implementation. It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; librtk_amd.so never links or calls it.
 *
 * Pinning: the reference ships no tests or golden vectors, so
 * the oracle is pinned against outputs of the REAL reference source compiled in the
 * build container (oracle/_ref, see oracle/Makefile + oracle/ref_shim.c):
 *   - bit-exact (t,u,v,ids) on identical single-leaf blobs (tests/test_oracle_ref.py,
 *     runs only where oracle/_ref exists), and
 *   - through the committed fixtures tests/golden/ made by oracle/gen_golden.py.
 */
#ifndef RTK_ORACLE_H
#define RTK_ORACLE_H

#include "rtk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tie handling when two triangles report bit-equal t.
 * ORA_TIES_REFERENCE: first candidate encountered wins (strict '<', rtk.c:371) and
 *   nodes are skipped when entry >= hit.t (rtk.c:432) -- depends on BVH visit order.
 * ORA_TIES_CANONICAL: lowest (mesh_index, triangle_index) wins; nodes are skipped
 *   only when entry > hit.t. This is what the reference returns when leaves are
 *   visited in triangle order (the leaf-chain oracle), and what the HIP path does. */
enum { ORA_TIES_REFERENCE = 0, ORA_TIES_CANONICAL = 1 };

/* Visit counters for the algorithmic-bytes model. */
typedef struct ora_counters {
	uint64_t rays;
	uint64_t nodes;       /* 4-wide nodes slab-tested            (rtk.c:457-472) */
	uint64_t leaves;      /* leaves entered                      (rtk.c:441-447) */
	uint64_t tri_groups;  /* groups of 4 triangle slots tested   (rtk.c:212)     */
	uint64_t hits;
} ora_counters;

/* Build a scene blob on the CPU (binned SAH -> 4-wide collapse -> blob), following
 * the INTENT of rtk.c:1362-1622, 1719-1774 where the v0 code is defective
 * (SURVEY.md appendix B). Returns a 128-byte aligned allocation (release with
 * ora_free) and its size through *size_out; NULL on allocation failure. */
void *ora_build_scene(const rtk_scene_desc *desc, size_t *size_out);
void ora_free(void *p);

/* Smallest legal blob: header + root node whose slot 0 is one leaf holding the n
 * (<= 63) given triangles, slots 1-3 empty (the leaf-chain
 * oracle of section 8c). verts holds 3*n rtk_vertex. Returns bytes written or 0 if
 * cap is too small / n > 63. dst should be 64-byte aligned. */
size_t ora_make_leaf_blob(const rtk_vertex *verts, const uint32_t *mesh_index,
	const uint32_t *triangle_index, size_t n, void *dst, size_t cap);

/* Restatement of rtk_trace_ray (rtk.c:543-577) with the stack fix B5. */
bool ora_trace_ray(const void *blob, const rtk_ray *ray, rtk_hit *hit, int ties, ora_counters *ctr);

/* Many rays, optionally on several threads (OpenMP); hits[i] valid where mask[i]. */
void ora_trace_rays(const void *blob, const rtk_ray *rays, size_t n, rtk_hit *hits,
	uint8_t *mask, int ties, int threads, ora_counters *total);

/* Candidate filters, the oracle's statement of what include/rtk_amd.h's rtk_dev_filter / rtk_trace_rays_filter
 * mean (reference: rtk_filter_fn rtk.h:117, stub rtk.c:579-582): a candidate is considered only if every filter
 * that is set lets it through; per-ray arrays are indexed by the ray's position in the batch. */
typedef bool ora_filter_fn(void *user, size_t ray_index, const rtk_hit *candidate);
typedef struct ora_filter {
	const uint32_t *mesh_mask;      /* bit m set = mesh m visible */
	uint32_t mesh_mask_bits;
	const uint32_t *ignore_mesh;    /* per ray: (mesh, triangle) never a candidate; NULL = none */
	const uint32_t *ignore_tri;
	const float *after_t;           /* per ray: only candidates after (t, mesh, triangle); mesh 0xffffffff = off */
	const uint32_t *after_mesh;
	const uint32_t *after_tri;
	ora_filter_fn *callback;
	void *user;
} ora_filter;
void ora_trace_rays_filter(const void *blob, const rtk_ray *rays, size_t n, rtk_hit *hits, uint8_t *mask,
	int threads, const ora_filter *filter);

/* Leaf chain: trace every ray against `num_blobs` blobs in order, feeding
 * ray.max_t = best.t after each hit. */
void ora_trace_chain(const void *const *blobs, size_t num_blobs, const rtk_ray *rays, size_t n,
	rtk_hit *hits, uint8_t *mask, int threads);

/* Ray setup only (rtk.c:550-566): kz and shear constants, for unit tests. */
void ora_ray_setup(const rtk_ray *ray, uint32_t k[3], float shear[3], uint32_t *sign_mask);

/* Sanity walk of a blob: header fields, every node/leaf/vertex offset in range.
 * Returns 0 if fine, else a negative code; counts are optional outputs. */
int ora_validate_blob(const void *blob, size_t size, uint64_t *num_nodes, uint64_t *num_leaves, uint64_t *num_tris);

int ora_max_threads(void);

/* The CPU baseline's timing driver (bench.py): closest hits as 16-byte records -- t, u, v, triangle_index (0xffffffff and
 * t = max_t for a miss; single-mesh scenes) -- into a buffer the caller allocated and touched, chunks of 1024 rays; and an
 * allocation whose pages the worker threads first-touch round robin (optionally filled from src). */
typedef struct ora_record { float t, u, v; uint32_t triangle_index; } ora_record;
void ora_trace_rays_records(const void *blob, const rtk_ray *rays, size_t n, ora_record *out, int ties, int threads);
void *ora_alloc_spread(size_t size, const void *src, int threads);

#ifdef __cplusplus
}
#endif
#endif
