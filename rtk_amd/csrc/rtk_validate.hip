// rtk_validate.hip -- structural check of a device-resident BVH, on the device.
//
// The reference has no validator ("No loader/validator exists"); the blob reader
// of rtk_upload.hip checks offsets, and this checks what the builder promises, independently of any
// traversal: results that only compare a traversal of the SAME tree (oracle on the exported blob) cannot
// see a node bound that is wrong for both.
//   - every child box contains what is below it: a leaf child's box is compared with its triangles'
//     vertices, an inner child's box with the boxes stored in that child node; "loose" counts boxes
//     that contain but are not the exact union (legal, the device builder never produces them);
//   - every triangle slot belongs to exactly one leaf, every primitive id occurs exactly once and
//     prim_slot is its inverse;
//   - every node except the root is referenced exactly once, children come after their parent (hence no cycles). A tree
//     from the device build's tile collapse has two runs of numbers: the nodes of the refit tiles [1, first_top), and the
//     nodes above the tiles, 0 and [first_top, n); inside each run children come after their parent, a node above the tiles
//     may also point down into the tiles' run, a tile's node never out of it (still no cycles);
//   - leaf headers: 1..63 triangles (rtk.c:188), count in the first record, end flag on the last.
// A content hash (order-sensitive per element, combined commutatively) tells two builds apart.
#include "rtk_dev.h"

#include <math.h>
#include <string.h>

namespace {

enum { C_NODES, C_LEAVES, C_TRIS, C_BOX_VIOLATION, C_LOOSE, C_BAD_REF, C_LEAF_FORMAT, C_TRI_MISSING, C_TRI_DUP,
	C_NODE_UNREACHED, C_NODE_SHARED, C_PRIM_BAD, C_QUANT, C_FIRST_BAD, C_HASH, C_WORDS };

__device__ __forceinline__ void report(unsigned long long *c, int kind, unsigned long long where)
{
	atomicAdd(c + kind, 1ull);
	atomicMin(c + C_FIRST_BAD, where);
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
	return x;
}

__global__ void k_check_nodes(DevSceneView sc, uint32_t first_top, uint32_t *slot_seen, uint32_t *node_seen, unsigned long long *c)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= sc.num_nodes) return;
	const DevNode nd = sc.nodes[i];
	atomicAdd(c + C_NODES, 1ull);
	unsigned long long h = mix64(0x9e3779b97f4a7c15ull * (i + 1ull));
	const uint32_t *w = reinterpret_cast<const uint32_t *>(&nd);
	for (int k = 0; k < 28; k++) h = mix64(h ^ w[k]);          // boxes and child references, not the padding
	atomicAdd(c + C_HASH, h);
	if (sc.qnodes) {
		// the compressed copy of this node: same children, and every decoded box contains the exact one
		const DevNodeQ qn = sc.qnodes[i];
		for (int k = 0; k < 4; k++) {
			bool ok = qn.child[k] == nd.child[k];
			if (nd.child[k] != RTK_REF_NONE) {
				const float *lo[3] = { nd.bx[0], nd.by[0], nd.bz[0] }, *hi[3] = { nd.bx[1], nd.by[1], nd.bz[1] };
				for (int a = 0; a < 3; a++) {
					const double qlo = (double)qn.org[a] + (double)((qn.q[a][0] >> (8 * k)) & 255u) * (double)qn.scale[a];
					const double qhi = (double)qn.org[a] + (double)((qn.q[a][1] >> (8 * k)) & 255u) * (double)qn.scale[a];
					if (!(qlo <= (double)lo[a][k] && qhi >= (double)hi[a][k])) ok = false;
				}
			}
			if (!ok) report(c, C_QUANT, i);
		}
	}
	for (int k = 0; k < 4; k++) {
		const uint32_t ref = nd.child[k];
		const float mn[3] = { nd.bx[0][k], nd.by[0][k], nd.bz[0][k] }, mx[3] = { nd.bx[1][k], nd.by[1][k], nd.bz[1][k] };
		if (ref == RTK_REF_NONE) {
			// an empty slot must never be hit, and the packet kernel's fast slab test relies on the arithmetic alone for
			// that (it does not read the child word): the inverted box every producer writes, +1 / -1 on every axis
			// (rtk.c:1612-1620), nothing weaker
			if (!(mn[0] == 1.0f && mn[1] == 1.0f && mn[2] == 1.0f && mx[0] == -1.0f && mx[1] == -1.0f && mx[2] == -1.0f)) report(c, C_BOX_VIOLATION, i);
			continue;
		}
		float cmn[3] = { INFINITY, INFINITY, INFINITY }, cmx[3] = { -INFINITY, -INFINITY, -INFINITY };
		if (ref & RTK_REF_LEAF) {
			const uint32_t first = ref & 0x7fffffffu;
			if (first >= sc.num_tris) { report(c, C_BAD_REF, i); continue; }
			const uint32_t cnt = sc.tris[first].spare;
			if (cnt < 1u || cnt > 63u || (unsigned long long)first + cnt > sc.num_tris) { report(c, C_LEAF_FORMAT, i); continue; }
			atomicAdd(c + C_LEAVES, 1ull);
			for (uint32_t t = 0; t < cnt; t++) {
				const DevTri tr = sc.tris[first + t];
				const bool last = (tr.flags & RTK_TRI_LAST) != 0u;
				if (last != (t + 1u == cnt) || (t > 0u && tr.spare != 0u)) report(c, C_LEAF_FORMAT, i);
				atomicAdd(&slot_seen[first + t], 1u);
				for (int a = 0; a < 3; a++) {
					cmn[a] = fminf(cmn[a], fminf(fminf(tr.v0[a], tr.v1[a]), tr.v2[a]));
					cmx[a] = fmaxf(cmx[a], fmaxf(fmaxf(tr.v0[a], tr.v1[a]), tr.v2[a]));
				}
			}
		} else {
			// (first_top 0: one run of numbers)
			const bool tile_node = i != 0u && i < first_top, tile_child = ref != 0u && ref < first_top;
			const bool order_ok = tile_node ? (ref > i && tile_child) : (ref > i || tile_child);
			if (ref >= sc.num_nodes || !order_ok) { report(c, C_BAD_REF, i); continue; }
			atomicAdd(&node_seen[ref], 1u);
			const DevNode ch = sc.nodes[ref];
			for (int q = 0; q < 4; q++) {
				if (ch.child[q] == RTK_REF_NONE) continue;
				cmn[0] = fminf(cmn[0], ch.bx[0][q]); cmx[0] = fmaxf(cmx[0], ch.bx[1][q]);
				cmn[1] = fminf(cmn[1], ch.by[0][q]); cmx[1] = fmaxf(cmx[1], ch.by[1][q]);
				cmn[2] = fminf(cmn[2], ch.bz[0][q]); cmx[2] = fmaxf(cmx[2], ch.bz[1][q]);
			}
		}
		bool contains = true, exact = true;
		for (int a = 0; a < 3; a++) {
			if (!(mn[a] <= cmn[a] && mx[a] >= cmx[a])) contains = false;   // NaN bounds fail too
			if (mn[a] != cmn[a] || mx[a] != cmx[a]) exact = false;
		}
		if (!contains) report(c, C_BOX_VIOLATION, i);
		else if (!exact) atomicAdd(c + C_LOOSE, 1ull);
	}
}

__global__ void k_check_slots(DevSceneView sc, const uint32_t *slot_seen, uint32_t *prim_seen, unsigned long long *c)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= sc.num_tris) return;
	atomicAdd(c + C_TRIS, 1ull);
	const uint32_t seen = slot_seen[s];
	if (seen == 0u) report(c, C_TRI_MISSING, s);
	else if (seen > 1u) report(c, C_TRI_DUP, s);
	const DevTri tr = sc.tris[s];
	unsigned long long h = mix64(0xd6e8feb86659fd93ull * (s + 1ull));
	const uint32_t *w = reinterpret_cast<const uint32_t *>(&tr);
	for (int k = 0; k < 12; k++) h = mix64(h ^ w[k]);
	atomicAdd(c + C_HASH, h);
	if (tr.prim >= sc.num_prims) { report(c, C_PRIM_BAD, s); return; }
	atomicAdd(&prim_seen[tr.prim], 1u);
	if (sc.prim_slot[tr.prim] != s) report(c, C_PRIM_BAD, s);
}

__global__ void k_check_counts(DevSceneView sc, const uint32_t *node_seen, const uint32_t *prim_seen, uint32_t expect_all_prims, unsigned long long *c)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < sc.num_nodes) {
		const uint32_t seen = node_seen[i];
		if (i == 0u) { if (seen != 0u) report(c, C_NODE_SHARED, i); }
		else if (seen == 0u) report(c, C_NODE_UNREACHED, i);
		else if (seen > 1u) report(c, C_NODE_SHARED, i);
	}
	if (i < sc.num_prims) {
		const uint32_t seen = prim_seen[i];
		if (seen > 1u || (expect_all_prims && seen == 0u)) report(c, C_PRIM_BAD, i);
	}
}

} // namespace

extern "C" int rtk_dev_scene_validate(const rtk_dev_scene *ds, rtk_dev_scene_check *out)
{
	if (!ds || !out) { rtk_set_error("rtk_dev_scene_validate: NULL argument"); return RTK_AMD_ERR_BAD_ARG; }
	if (rtk_scene_side_arrays(ds, nullptr) != RTK_AMD_OK) return RTK_AMD_ERR_OOM;
	const DevSceneView &v = ds->view;
	const size_t words = (size_t)v.num_tris + v.num_nodes + v.num_prims + 16;
	uint32_t *d_seen = nullptr;
	unsigned long long *d_c = nullptr;
	unsigned long long h[C_WORDS];
	int rc = RTK_AMD_OK;
	do {
		if (hipMalloc(&d_seen, words * 4) != hipSuccess || hipMalloc(&d_c, C_WORDS * 8) != hipSuccess) { rtk_set_error("rtk_dev_scene_validate: out of device memory"); rc = RTK_AMD_ERR_OOM; break; }
		memset(h, 0, sizeof(h));
		h[C_FIRST_BAD] = ~0ull;
		if (hipMemset(d_seen, 0, words * 4) != hipSuccess || hipMemcpy(d_c, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) { rc = RTK_AMD_ERR_HIP; break; }
		uint32_t *slot_seen = d_seen, *node_seen = d_seen + v.num_tris, *prim_seen = node_seen + v.num_nodes;
		hipLaunchKernelGGL(k_check_nodes, dim3((v.num_nodes + 127u) / 128u), dim3(128), 0, 0, v, ds->first_top, slot_seen, node_seen, d_c);
		if (v.num_tris) hipLaunchKernelGGL(k_check_slots, dim3((v.num_tris + 255u) / 256u), dim3(256), 0, 0, v, slot_seen, prim_seen, d_c);
		const uint32_t m = v.num_nodes > v.num_prims ? v.num_nodes : v.num_prims;
		// a scene built here holds every primitive of its meshes; an uploaded blob may leave ids unused
		hipLaunchKernelGGL(k_check_counts, dim3((m + 255u) / 256u), dim3(256), 0, 0, v, node_seen, prim_seen, v.num_prims == v.num_tris ? 1u : 0u, d_c);
		if (hipGetLastError() != hipSuccess || hipMemcpy(h, d_c, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
			rtk_set_error("rtk_dev_scene_validate: %s", hipGetErrorString(hipGetLastError()));
			rc = RTK_AMD_ERR_HIP; break;
		}
	} while (0);
	if (d_seen) (void)hipFree(d_seen);
	if (d_c) (void)hipFree(d_c);
	if (rc != RTK_AMD_OK) return rc;
	out->nodes_checked = h[C_NODES]; out->leaves_checked = h[C_LEAVES]; out->triangles_checked = h[C_TRIS];
	out->box_violations = h[C_BOX_VIOLATION]; out->loose_boxes = h[C_LOOSE]; out->bad_references = h[C_BAD_REF];
	out->leaf_format_errors = h[C_LEAF_FORMAT]; out->triangles_missing = h[C_TRI_MISSING]; out->triangles_duplicated = h[C_TRI_DUP];
	out->nodes_unreachable = h[C_NODE_UNREACHED]; out->nodes_shared = h[C_NODE_SHARED]; out->primitive_id_errors = h[C_PRIM_BAD];
	out->compressed_node_errors = h[C_QUANT];
	out->first_bad_index = h[C_FIRST_BAD]; out->content_hash = h[C_HASH];
	const unsigned long long bad = h[C_BOX_VIOLATION] + h[C_BAD_REF] + h[C_LEAF_FORMAT] + h[C_TRI_MISSING] + h[C_TRI_DUP] +
		h[C_NODE_UNREACHED] + h[C_NODE_SHARED] + h[C_PRIM_BAD] + h[C_QUANT];
	if (bad) {
		rtk_set_error("rtk_dev_scene_validate: %llu structural errors (first at index %llu)", bad, h[C_FIRST_BAD]);
		return RTK_AMD_ERR_BAD_SCENE;
	}
	return RTK_AMD_OK;
}
