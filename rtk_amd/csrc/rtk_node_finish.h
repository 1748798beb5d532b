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
	for (uint32_t o = 0; o < 8u; o++) {
		float key[4];
		for (int k = 0; k < 4; k++) {
			const float cx = nd.bx[0][k] + nd.bx[1][k], cy = nd.by[0][k] + nd.by[1][k], cz = nd.bz[0][k] + nd.bz[1][k];
			float s = ((o & 1u) ? -cx : cx) + ((o & 2u) ? -cy : cy) + ((o & 4u) ? -cz : cz);
			if (!(s == s)) s = INFINITY;                           // NaN boxes sort behind everything real
			key[k] = nd.child[k] == RTK_REF_NONE ? INFINITY : s;
		}
		// rank of slot k = how many slots come before it
		uint32_t word = 0u, pair = 0u, bit = 0u;
		for (int i = 0; i < 4; i++) {
			uint32_t rank = 0;
			for (int j = 0; j < 4; j++) if (j != i && (key[j] < key[i] || (key[j] == key[i] && j < i))) rank++;
			word |= (uint32_t)i << (2u * rank);
			for (int j = i + 1; j < 4; j++, bit++) if (key[j] < key[i]) pair |= 1u << bit;   // the second of the pair comes first
		}
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
			for (int k = 0; k < 4; k++) {
				uint32_t ql = 255u, qh = 0u;                       // empty slot: inverted, can never be entered
				if (nd.child[k] != RTK_REF_NONE) {
					// floor / ceil in float, then made safe in double: org + q * s is exact there
					float fl = floorf((lo[a][k] - mn) / s), fh = ceilf((hi[a][k] - mn) / s);
					fl = fminf(fmaxf(fl, 0.0f), 255.0f);
					fh = fminf(fmaxf(fh, 0.0f), 300.0f);
					ql = (uint32_t)fl; qh = (uint32_t)fh;
					while (ql > 0u && (double)mn + (double)ql * (double)s > (double)lo[a][k]) ql--;
					while (qh < 300u && (double)mn + (double)qh * (double)s < (double)hi[a][k]) qh++;
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
