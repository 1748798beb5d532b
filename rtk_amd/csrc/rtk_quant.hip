// rtk_quant.hip -- DevNode (128 B, exact child boxes) -> DevNodeQ (64 B, 8-bit planes), one thread per node.
// Used after a device build and after a blob upload; see rtk_dev.h for the format and why it exists.
#include "rtk_dev.h"
#include "rtk_node_finish.h"

#include <math.h>

namespace {

// copy_to: also store the exact node there (the device build hands its workspace copy over in the same pass)
__global__ void k_quantize(DevNode *nodes, uint32_t n, DevNodeQ *out, DevNode *copy_to, DevSceneConsts *consts, float bound_hint)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	DevNode nd = nodes[i];
	child_order(nd, nd.order);
	if (copy_to) copy_to[i] = nd;
	else *reinterpret_cast<uint4 *>(nodes[i].order) = make_uint4(nd.order[0], nd.order[1], nd.order[2], nd.order[3]);
	if (i == 0u) {
		// every box of a tree the device builds lies inside the root's (exact unions); an upload passes the bound over all nodes
		const float b = root_bound(nd, bound_hint);
		consts->bound_raw = b;
		consts->bound_abs = fmaxf(b, 1.0f);
	}
	DevNodeQ q;
	const bool misfit = !quantize_node(nd, q);
	out[i] = q;
	// a compressed box that does not contain its exact box would cull real hits: the host then keeps the scene on its exact nodes
	if (misfit) atomicAdd(&consts->qnode_misfits, 1u);
}

} // namespace

// The scene's constants word block, allocated on first use and cleared on `stream`.
int rtk_scene_consts(rtk_dev_scene *ds, hipStream_t stream)
{
	if (!ds->view.consts) {
		void *c = nullptr;
		RTK_HIP_CHECK(hipMalloc(&c, sizeof(DevSceneConsts)), RTK_AMD_ERR_OOM);
		ds->allocs.push_back(c);
		ds->view.consts = (const DevSceneConsts *)c;
	}
	RTK_HIP_CHECK(hipMemsetAsync(const_cast<DevSceneConsts *>(ds->view.consts), 0, sizeof(DevSceneConsts), stream), RTK_AMD_ERR_HIP);   // (bound_abs 0 = no nodes; kernels read it as max(bound, 1))
	return RTK_AMD_OK;
}

int rtk_quantize_nodes(rtk_dev_scene *ds, hipStream_t stream, const DevNode *src, DevNodeQ *dst, float bound_hint, uint32_t only_first, bool keep_consts, bool readback)
{
	const uint32_t n = ds->view.num_nodes < only_first ? ds->view.num_nodes : only_first;
	void *p = dst;
	if (!keep_consts || !ds->view.consts) { const int rc = rtk_scene_consts(ds, stream); if (rc != RTK_AMD_OK) return rc; }
	DevSceneConsts *consts = const_cast<DevSceneConsts *>(ds->view.consts);
	if (!p) {
		RTK_HIP_CHECK(hipMalloc(&p, (size_t)(ds->view.num_nodes ? ds->view.num_nodes : 1) * sizeof(DevNodeQ)), RTK_AMD_ERR_OOM);
		ds->allocs.push_back(p);
		ds->total_bytes += (size_t)ds->view.num_nodes * sizeof(DevNodeQ);
	}
	// src: the nodes still sit in a workspace; ds->view.nodes (allocated, not yet filled) receives them in the same pass
	if (n) hipLaunchKernelGGL(k_quantize, dim3((n + 255u) / 256u), dim3(256), 0, stream, const_cast<DevNode *>(src ? src : ds->view.nodes), n, (DevNodeQ *)p,
		src ? const_cast<DevNode *>(ds->view.nodes) : (DevNode *)nullptr, consts, bound_hint);
	RTK_HIP_CHECK(hipGetLastError(), RTK_AMD_ERR_HIP);
	ds->view.qnodes = (const DevNodeQ *)p;
	// the misfit count comes back with the caller's own synchronisation of `stream` (rtk_quantize_finish)
	if (readback) RTK_HIP_CHECK(hipMemcpyAsync(&ds->consts_readback, consts, sizeof(DevSceneConsts), hipMemcpyDeviceToHost, stream), RTK_AMD_ERR_HIP);
	return RTK_AMD_OK;
}

// After the stream of rtk_quantize_nodes has been synchronised: a scene with a node that does not fit the 8-bit grid
// (non-finite child extents) is traced with its exact nodes only.
void rtk_quantize_finish(rtk_dev_scene *ds)
{
	ds->bound_abs = ds->consts_readback.bound_abs;
	ds->bound_raw = ds->consts_readback.bound_raw;
	if (ds->consts_readback.qnode_misfits != 0u) ds->view.qnodes = nullptr;
}
