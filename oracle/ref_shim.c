/*
 * ref_shim.c -- builds the REAL reference (bqqbarbhg/rtk @ v0) into oracle/_ref/ so
 * that the CPU restatement (rtk_oracle.c) can be validated against it and golden
 * fixtures can be generated. TEST INFRASTRUCTURE ONLY; exists only in the build
 * container (the reference never travels to the GPU box; oracle/_ref/ is git-ignored).
 *
 * The reference source is compiled VERBATIM from where it lies (/root/reference/rtk.c,
 * path supplied by the Makefile as RTK_REF_C); nothing of it is copied into this
 * repository. The v0 snapshot defines five compiler-portability macros only under
 * _MSC_VER (rtk.c:47-58, 170-175) and therefore does not compile with gcc as-is
 * (SURVEY.md section 8c); they are supplied here, before the include, in terms of
 * GCC builtins. rtk_alloc is the reference's own documented override hook
 * (rtk.c:32-45).
 *
 * Only the reference's TRACE path is used (rtk_trace_ray, rtk.c:543-577 and what it
 * calls). Its build path cannot produce a scene at v0, and
 * its traversal stack push is off by one (B5), so it is only ever given blobs whose
 * nodes have at most one non-empty child: the "leaf chain" of SURVEY.md section 8c.
 */
#include <immintrin.h>   /* SSE4.1 _mm_blendv_ps (rtk.c:165) needs more than the three headers rtk.c includes */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef size_t _rtk_atomic_size;
#define _rtk_atomic_size_add(a, v) __atomic_fetch_add((a), (size_t)(v), __ATOMIC_SEQ_CST) /* returns the OLD value, cf. rtk.c:828 */
#define RTK_FIRSTBIT4(index, mask) ((index) = (uint32_t)__builtin_ctz((unsigned)(mask)))
#define RTK_POPCOUNT4(mask) ((uint32_t)__builtin_popcount((unsigned)(mask)))
#define RTK_ALIGN16 __attribute__((aligned(16)))

static void *ref_alloc64(size_t size)
{
	void *p = NULL;
	size = (size + 63u) & ~(size_t)63u;
	if (posix_memalign(&p, 64, size ? size : 64) != 0) return NULL;
	memset(p, 0, size ? size : 64);
	return p;
}
#define rtk_alloc 1
#define rtk_mem_alloc(size) ref_alloc64(size)
#define rtk_mem_free(ptr, size) free(ptr)

#ifndef RTK_REF_C
#error "RTK_REF_C must name the reference rtk.c (see oracle/Makefile)"
#endif
#include RTK_REF_C

/* -- drivers (own code) -- */

/* One call of the reference's rtk_trace_ray. */
int ref_trace_ray(const void *blob, const rtk_ray *ray, rtk_hit *hit)
{
	return rtk_trace_ray((const rtk_scene *)blob, ray, hit) ? 1 : 0;
}

/* Leaf chain: every ray against every single-leaf blob in order, feeding
 * ray.max_t = best.t after each hit. */
void ref_trace_chain(const void *const *blobs, size_t num_blobs, const rtk_ray *rays, size_t n,
	rtk_hit *hits, uint8_t *mask, int threads)
{
	if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
#endif
	for (long long i = 0; i < (long long)n; i++) {
		rtk_ray r = rays[i];
		rtk_hit best;
		int any = 0;
		for (size_t b = 0; b < num_blobs; b++) {
			rtk_hit h;
			if (rtk_trace_ray((const rtk_scene *)blobs[b], &r, &h)) {
				best = h;
				any = 1;
				r.max_t = h.t;
			}
		}
		if (any && hits) hits[i] = best;
		if (mask) mask[i] = (uint8_t)any;
	}
}

size_t ref_sizeof(int what)
{
	switch (what) {
	case 0: return sizeof(_rtk_bvh_node);
	case 1: return sizeof(_rtk_bvh_leaf);
	case 2: return sizeof(_rtk_leaf_triangle);
	case 3: return sizeof(rtk_hit);
	case 4: return sizeof(rtk_ray);
	case 5: return sizeof(rtk_scene);
	case 6: return sizeof(rtk_mesh);
	case 7: return sizeof(rtk_task);
	default: return 0;
	}
}
