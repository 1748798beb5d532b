// rtk_node_finish.h -- what turns the boxes and child words of a 4-wide node into its final form: the front-to-back child
// order words (DevNode::order) and the 64-byte compressed copy (DevNodeQ). Shared by k_quantize (rtk_quant.hip: uploads, and
// the top of a device-built tree) and by the tile-local collapse of the device build (rtk_build.hip), which finishes its
// nodes in the pass that makes them instead of re-reading them.
#pragma once

#include "rtk_dev.h"

#include <math.h>


// largest power of two s.t. 254 steps still cover `extent` is too coarse by up to 2x; this picks the smallest
// power of two with 254 * s >= extent (one level is kept in reserve for the round-up of the high planes)
__device__ __forceinline__ float grid_step(float extent)
{
	if (!(extent > 0.0f)) return 1.17549435e-38f;            // flat on this axis: every plane sits at q = 0
	int e;
	(void)frexpf(extent, &e);                                // extent = m * 2^e, m in [0.5, 1)
	float s = ldexpf(1.0f, e - 8);                           // 256 * s = 2^e > extent
	if (254.0f * s < extent) s *= 2.0f;
	return s;
}

// Front-to-back order of the children per direction octant (DevNode::order): by the centre of the child box along the
// octant's diagonal, empty slots last, ties by slot number.
__device__ __forceinline__ void child_order(const DevNode &nd, uint32_t order[4])
{
	order[0] = order[1] = order[2] = order[3] = 0u;
	float cx[4], cy[4], cz[4];
#pragma unroll
	for (int k = 0; k < 4; k++) { cx[k] = nd.bx[0][k] + nd.bx[1][k]; cy[k] = nd.by[0][k] + nd.by[1][k]; cz[k] = nd.bz[0][k] + nd.bz[1][k]; }
#pragma unroll
	for (uint32_t o = 0; o < 8u; o++) {
		float key[4];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			float s = ((o & 1u) ? -cx[k] : cx[k]) + ((o & 2u) ? -cy[k] : cy[k]) + ((o & 4u) ? -cz[k] : cz[k]);
			if (!(s == s)) s = INFINITY;                           // NaN boxes sort behind everything real
			key[k] = nd.child[k] == RTK_REF_NONE ? INFINITY : s;
		}
		// For the six pairs i < j: does i come before j? (ties go to the lower slot; no key is a NaN.) The rank of a slot = how
		// many slots come before it; a pair bit says that the SECOND of the pair comes first.
		const uint32_t b01 = key[0] <= key[1], b02 = key[0] <= key[2], b03 = key[0] <= key[3], b12 = key[1] <= key[2], b13 = key[1] <= key[3], b23 = key[2] <= key[3];
		const uint32_t r0 = (1u - b01) + (1u - b02) + (1u - b03), r1 = b01 + (1u - b12) + (1u - b13), r2 = b02 + b12 + (1u - b23), r3 = b03 + b13 + b23;
		uint32_t word = (0u << (2u * r0)) | (1u << (2u * r1)) | (2u << (2u * r2)) | (3u << (2u * r3));
		const uint32_t pair = (1u - b01) | ((1u - b02) << 1) | ((1u - b03) << 2) | ((1u - b12) << 3) | ((1u - b13) << 4) | ((1u - b23) << 5);
		word |= pair << RTK_ORDER_PAIR_SHIFT;
		order[o >> 1] |= word << (16u * (o & 1u));
	}
}


// The compressed copy of `nd` (child words included). Returns false if some child box does not fit the 8-bit grid (extents
// that are not finite in float): the scene then keeps to its exact nodes.
__device__ __forceinline__ bool quantize_node(const DevNode &nd, DevNodeQ &q)
{
	bool misfit = false;
	const float *lo[3] = { nd.bx[0], nd.by[0], nd.bz[0] }, *hi[3] = { nd.bx[1], nd.by[1], nd.bz[1] };
#pragma unroll
	for (int a = 0; a < 3; a++) {
		float mn = INFINITY, mx = -INFINITY;
		for (int k = 0; k < 4; k++) if (nd.child[k] != RTK_REF_NONE) { mn = fminf(mn, lo[a][k]); mx = fmaxf(mx, hi[a][k]); }
		if (!(mn <= mx)) { mn = 0.0f; mx = 0.0f; }             // a node without children (empty scene)
		float s = grid_step(mx - mn);
		uint32_t wl = 0, wh = 0;
		for (int attempt = 0; attempt < 4; attempt++) {
			bool fits = true;
			wl = wh = 0;
			// (the step is a power of two: multiplying by its reciprocal IS the division, bit for bit, without the divide's
			// dozen instructions -- eight of them per axis were a quarter of what k_collapse_tile executes per node)
			// (a step below 2^-126 -- a box whose extent is a denormal -- has no finite reciprocal: those divide)
			const bool recip = s >= 0x1p-126f && s <= 0x1p126f;     // (and its reciprocal is a normal number too)
			const float inv_s = recip ? 1.0f / s : 0.0f;
			for (int k = 0; k < 4; k++) {
				uint32_t ql = 255u, qh = 0u;                       // empty slot: inverted, can never be entered
				if (nd.child[k] != RTK_REF_NONE) {
					// floor / ceil in float, then made safe in double: org + q * s is exact there
					float fl, fh;
					if (recip) { fl = floorf((lo[a][k] - mn) * inv_s); fh = ceilf((hi[a][k] - mn) * inv_s); }
					else { fl = floorf((lo[a][k] - mn) / s); fh = ceilf((hi[a][k] - mn) / s); }
					fl = fminf(fmaxf(fl, 0.0f), 255.0f);
					fh = fminf(fmaxf(fh, 0.0f), 300.0f);
					ql = (uint32_t)fl; qh = (uint32_t)fh;
					// (the corrections below run zero times for almost every plane; written as plain `while` loops hipcc tests four
					// candidates per trip -- four double-precision checks before the first one is looked at, a third of what the
					// tile collapse executed -- so the first test stands alone and the loop behind it is one that is seldom entered)
					const double mn_d = (double)mn, s_d = (double)s, lo_d = (double)lo[a][k], hi_d = (double)hi[a][k];
					if (ql > 0u && mn_d + (double)ql * s_d > lo_d) {
						do ql--; while (ql > 0u && mn_d + (double)ql * s_d > lo_d);
					}
					if (qh < 300u && mn_d + (double)qh * s_d < hi_d) {
						do qh++; while (qh < 300u && mn_d + (double)qh * s_d < hi_d);
					}
					if (qh > 255u) fits = false;
				}
				wl |= (ql & 255u) << (8 * k);
				wh |= (qh & 255u) << (8 * k);
			}
			if (fits) break;
			// (an extent that is not finite in float -- planes beyond +-1.7e38 or inf -- never fits: frexpf(inf) gives a tiny step)
			if (attempt == 3) misfit = true;
			s *= 2.0f;
		}
		q.org[a] = mn;
		q.scale[a] = s;
		q.q[a][0] = wl;
		q.q[a][1] = wh;
	}
	for (int k = 0; k < 4; k++) q.child[k] = nd.child[k];
	return !misfit;
}

// The scene bound from the root node (every box of a tree the device builds lies inside the root's child boxes): the largest
// absolute plane, INFINITY if one is not finite. bound_hint: a bound the caller already knows (uploads), 0 = none.
__device__ __forceinline__ float root_bound(const DevNode &nd, float bound_hint)
{
	float b = bound_hint;
	for (int k = 0; k < 4; k++) {
		if (nd.child[k] == RTK_REF_NONE) continue;
		const float v[6] = { nd.bx[0][k], nd.bx[1][k], nd.by[0][k], nd.by[1][k], nd.bz[0][k], nd.bz[1][k] };
		for (int c = 0; c < 6; c++) b = (fabsf(v[c]) <= 3.0e38f) ? fmaxf(b, fabsf(v[c])) : INFINITY;   // NaN / inf planes: no bound
	}
	return b;
}
