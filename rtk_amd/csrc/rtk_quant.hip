// rtk_quant.hip -- DevNode (128 B, exact child boxes) -> DevNodeQ (64 B, 8-bit planes), one thread per node.
// Used after a device build and after a blob upload; see rtk_dev.h for the format and why it exists.
#include "rtk_dev.h"

#include <math.h>

namespace {

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

// copy_to: also store the exact node there (the device build hands its workspace copy over in the same pass)
__global__ void k_quantize(const DevNode *nodes, uint32_t n, DevNodeQ *out, DevNode *copy_to)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const DevNode nd = nodes[i];
	if (copy_to) copy_to[i] = nd;
	DevNodeQ q;
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
			s *= 2.0f;
		}
		q.org[a] = mn;
		q.scale[a] = s;
		q.q[a][0] = wl;
		q.q[a][1] = wh;
	}
	for (int k = 0; k < 4; k++) q.child[k] = nd.child[k];
	out[i] = q;
}

} // namespace

int rtk_quantize_nodes(rtk_dev_scene *ds, hipStream_t stream, const DevNode *src, DevNodeQ *dst)
{
	const uint32_t n = ds->view.num_nodes;
	void *p = dst;
	if (!p) {
		RTK_HIP_CHECK(hipMalloc(&p, (size_t)(n ? n : 1) * sizeof(DevNodeQ)), RTK_AMD_ERR_OOM);
		ds->allocs.push_back(p);
		ds->total_bytes += (size_t)n * sizeof(DevNodeQ);
	}
	// src: the nodes still sit in a workspace; ds->view.nodes (allocated, not yet filled) receives them in the same pass
	if (n) hipLaunchKernelGGL(k_quantize, dim3((n + 255u) / 256u), dim3(256), 0, stream, src ? src : ds->view.nodes, n, (DevNodeQ *)p,
		src ? const_cast<DevNode *>(ds->view.nodes) : (DevNode *)nullptr);
	RTK_HIP_CHECK(hipGetLastError(), RTK_AMD_ERR_HIP);
	ds->view.qnodes = (const DevNodeQ *)p;
	return RTK_AMD_OK;
}
