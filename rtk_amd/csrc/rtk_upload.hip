// rtk_upload.hip -- scene blob -> device BVH.
//
// Reads a scene blob in the reference's format (reader side
// rtk.c:64-86, 181-193, 457-465, root at byte 128 per rtk.c:569), validates every offset
// (the reference has no loader/validator), and re-lays it out for the GPU (rtk_dev.h):
// nodes breadth-first with 32-bit child references, triangles pre-gathered into 48 B
// records in leaf order. This is a format conversion done once per scene on the host;
// no intersection work happens here.
#include "rtk_dev.h"

#include <math.h>

#include <string.h>

#include <unordered_map>

namespace {

struct BlobNode {          // rtk.c:69-74
	float bx[2][4], by[2][4], bz[2][4];
	uint64_t child[4];
};
struct BlobLeafTri {       // rtk.c:82-86
	uint8_t v[3];
	uint8_t local_mesh;
	uint32_t triangle_index;
};
static_assert(sizeof(BlobNode) == 128 && sizeof(BlobLeafTri) == 8, "blob records");

struct LeafRef {
	uint64_t offset;       // byte offset of the leaf header
	uint32_t count;
	uint32_t first_slot;
};

bool slot_is_empty(const BlobNode &n, int i)
{
	// empty slots carry inverted bounds that never pass the slab test (rtk.c:1612-1620)
	return !(n.bx[0][i] <= n.bx[1][i] && n.by[0][i] <= n.by[1][i] && n.bz[0][i] <= n.bz[1][i]);
}

} // namespace

// `avail`: bytes that are known to be readable at `scene` (the loader of an untrusted file passes the
// file size; callers that only have the reference's bare rtk_scene* pass size_in_bytes itself).
int rtk_blob_to_host_bvh(const rtk_scene *scene, size_t avail, HostBvh *out)
{
	static const char magic[8] = { 0, 'R', 'T', 'K', '\r', '\n', 0x1a, '\n' };
	if (!scene) { rtk_set_error("scene is NULL"); return RTK_AMD_ERR_BAD_ARG; }
	if (memcmp(scene->magic, magic, 8) != 0) { rtk_set_error("scene blob: bad magic"); return RTK_AMD_ERR_BAD_SCENE; }
	if (scene->endian != 0xaabb) { rtk_set_error("scene blob: foreign endianness"); return RTK_AMD_ERR_BAD_SCENE; }
	if (scene->sizeof_real != 4 || scene->version != 1) { rtk_set_error("scene blob: unsupported sizeof_real/version"); return RTK_AMD_ERR_BAD_SCENE; }
	const uint64_t size = scene->size_in_bytes;
	if (size < 256 || size > ((uint64_t)1 << 48) || scene->node_offset != 128) { rtk_set_error("scene blob: bad size or node offset"); return RTK_AMD_ERR_BAD_SCENE; }
	if (size > avail) { rtk_set_error("scene blob: header claims %llu bytes, only %zu are there", (unsigned long long)size, avail); return RTK_AMD_ERR_BAD_SCENE; }
	const char *blob = reinterpret_cast<const char *>(scene);

	// pass 1: breadth-first walk; number nodes, collect leaves. All range checks are written so that they
	// cannot wrap (offsets come from the file and may be anything up to 2^64-1). A blob must be a TREE: a
	// node or a non-empty leaf that is reached twice (shared subtree, or a child pointing back at an
	// ancestor) is refused -- the traversal stacks are sized from the depth of a tree.
	std::unordered_map<uint64_t, uint32_t> node_index;   // blob offset -> device node index
	std::unordered_map<uint64_t, uint32_t> leaf_index;   // blob offset -> index into leaves
	std::vector<uint64_t> node_offsets;
	std::vector<uint32_t> node_depth;
	std::vector<LeafRef> leaves;
	node_index[128] = 0;
	node_offsets.push_back(128);
	node_depth.push_back(1);
	uint32_t max_depth = 1;
	uint64_t total_tris = 0;
	for (size_t qi = 0; qi < node_offsets.size(); qi++) {
		const uint64_t off = node_offsets[qi];
		BlobNode n;
		memcpy(&n, blob + off, sizeof(n));                // off was range-checked when it was queued (root: size >= 256)
		for (int i = 0; i < 4; i++) {
			if (slot_is_empty(n, i)) continue;
			const uint64_t p = n.child[i];
			if (p & 1u) {
				const uint64_t lo = p ^ 1u;
				if (lo < 128 || lo > size - 8) { rtk_set_error("scene blob: leaf at %llu out of range", (unsigned long long)lo); return RTK_AMD_ERR_BAD_SCENE; }
				uint64_t info;
				memcpy(&info, blob + lo, 8);
				const uint32_t cnt = (uint32_t)(info & 0x3f);
				const uint64_t n4 = (cnt + 3u) & ~3ull;
				if (8 * n4 > size - lo - 8) { rtk_set_error("scene blob: leaf triangles out of range"); return RTK_AMD_ERR_BAD_SCENE; }
				if (leaf_index.count(lo)) {
					if (cnt) { rtk_set_error("scene blob: leaf at %llu is referenced twice (not a tree)", (unsigned long long)lo); return RTK_AMD_ERR_BAD_SCENE; }
					continue;
				}
				leaf_index[lo] = (uint32_t)leaves.size();
				leaves.push_back(LeafRef{ lo, cnt, 0 });
				total_tris += cnt;
			} else {
				if (p < 128 || (p & 127u) || p > size - sizeof(BlobNode)) { rtk_set_error("scene blob: node at %llu out of range or misaligned", (unsigned long long)p); return RTK_AMD_ERR_BAD_SCENE; }
				if (node_index.count(p)) { rtk_set_error("scene blob: node at %llu is referenced twice (shared subtree or cycle)", (unsigned long long)p); return RTK_AMD_ERR_BAD_SCENE; }
				node_index[p] = (uint32_t)node_offsets.size();
				node_offsets.push_back(p);
				node_depth.push_back(node_depth[qi] + 1);
				if (node_depth[qi] + 1 > max_depth) max_depth = node_depth[qi] + 1;
			}
		}
	}
	if (total_tris >= 0x7ffffff0ull || node_offsets.size() >= 0x7ffffff0ull) { rtk_set_error("scene too large for 31-bit references"); return RTK_AMD_ERR_UNSUPPORTED; }

	// pass 2: per-mesh triangle counts -> global primitive ids (rtk.c:1131-1178 order)
	std::vector<uint64_t> mesh_count;
	for (const LeafRef &lf : leaves) {
		const uint64_t n4 = (lf.count + 3u) & ~3ull;
		const BlobLeafTri *lt = reinterpret_cast<const BlobLeafTri *>(blob + lf.offset + 8);
		const char *table = blob + lf.offset + 8 + 8 * n4;
		for (uint32_t i = 0; i < lf.count; i++) {
			BlobLeafTri t;
			memcpy(&t, lt + i, 8);
			const uint64_t at = (uint64_t)(table - blob) + 4ull * t.local_mesh;     // <= size + 1020: cannot wrap
			if (at > size - 4) { rtk_set_error("scene blob: mesh table out of range"); return RTK_AMD_ERR_BAD_SCENE; }
			uint32_t mesh;
			memcpy(&mesh, blob + at, 4);
			if (mesh >= (1u << 24)) { rtk_set_error("scene blob: implausible mesh index %u", mesh); return RTK_AMD_ERR_BAD_SCENE; }
			if (mesh >= mesh_count.size()) mesh_count.resize(mesh + 1, 0);
			if ((uint64_t)t.triangle_index + 1 > mesh_count[mesh]) mesh_count[mesh] = (uint64_t)t.triangle_index + 1;
		}
	}
	out->mesh_base.assign(mesh_count.size() + 1, 0);
	for (size_t m = 0; m < mesh_count.size(); m++) out->mesh_base[m + 1] = out->mesh_base[m] + mesh_count[m];
	if (out->mesh_base.back() >= 0xfffffff0ull) { rtk_set_error("primitive ids exceed 32 bits"); return RTK_AMD_ERR_UNSUPPORTED; }

	// pass 3: emit triangles in leaf order
	out->tris.resize(total_tris);
	out->vertex_index.resize(3 * total_tris);
	out->slot_mesh.resize(total_tris);
	out->slot_tri.resize(total_tris);
	uint32_t slot = 0;
	for (LeafRef &lf : leaves) {
		lf.first_slot = slot;
		uint64_t info;
		memcpy(&info, blob + lf.offset, 8);
		const uint64_t vg = info & ~0x3full;
		const uint64_t n4 = (lf.count + 3u) & ~3ull;
		const BlobLeafTri *lt = reinterpret_cast<const BlobLeafTri *>(blob + lf.offset + 8);
		const char *table = blob + lf.offset + 8 + 8 * n4;
		for (uint32_t i = 0; i < lf.count; i++, slot++) {
			BlobLeafTri t;
			memcpy(&t, lt + i, 8);
			uint32_t mesh;
			memcpy(&mesh, table + 4ull * t.local_mesh, 4);
			DevTri &d = out->tris[slot];
			float *dst[3] = { d.v0, d.v1, d.v2 };
			for (int c = 0; c < 3; c++) {
				if (vg > size - 16 || 16ull * t.v[c] > size - 16 - vg) { rtk_set_error("scene blob: vertex out of range"); return RTK_AMD_ERR_BAD_SCENE; }
				const uint64_t at = vg + 16ull * t.v[c];
				rtk_vertex v;
				memcpy(&v, blob + at, 16);
				dst[c][0] = v.position.x; dst[c][1] = v.position.y; dst[c][2] = v.position.z;
				out->vertex_index[3 * (size_t)slot + c] = v.index;
			}
			d.prim = (uint32_t)(out->mesh_base[mesh] + t.triangle_index);
			d.flags = ((i + 1 == lf.count) ? RTK_TRI_LAST : 0u) | (mesh << 8);   // RTK_TRI_MESH_SHIFT
			d.spare = (i == 0) ? lf.count : 0u;   // leaf size rides in the first record
			out->slot_mesh[slot] = mesh;
			out->slot_tri[slot] = t.triangle_index;
		}
	}

	// pass 4: emit nodes
	out->nodes.resize(node_offsets.size());
	for (size_t qi = 0; qi < node_offsets.size(); qi++) {
		BlobNode n;
		memcpy(&n, blob + node_offsets[qi], sizeof(n));
		DevNode &d = out->nodes[qi];
		memcpy(d.bx, n.bx, sizeof(d.bx));
		memcpy(d.by, n.by, sizeof(d.by));
		memcpy(d.bz, n.bz, sizeof(d.bz));
		for (int i = 0; i < 4; i++) {
			d.order[i] = 0;
			uint32_t ref = RTK_REF_NONE;
			if (!slot_is_empty(n, i)) {
				const uint64_t p = n.child[i];
				if (p & 1u) {
					const LeafRef &lf = leaves[leaf_index[p ^ 1u]];
					if (lf.count) ref = RTK_REF_LEAF | lf.first_slot;
				} else {
					ref = node_index[p];
				}
			}
			d.child[i] = ref;
			if (ref == RTK_REF_NONE) {
				d.bx[0][i] = d.by[0][i] = d.bz[0][i] = +1.0f;
				d.bx[1][i] = d.by[1][i] = d.bz[1][i] = -1.0f;
			}
		}
	}
	out->max_depth = max_depth;
	return RTK_AMD_OK;
}

template <typename T>
static bool upload_vec(rtk_dev_scene *ds, const std::vector<T> &v, const T **dst, size_t min_elems = 1)
{
	const size_t n = v.size() > min_elems ? v.size() : min_elems;
	void *p = nullptr;
	if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return false;
	ds->allocs.push_back(p);
	ds->total_bytes += n * sizeof(T);
	if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return false;
	*dst = static_cast<const T *>(p);
	return true;
}

rtk_dev_scene *rtk_dev_scene_from_host_bvh(const HostBvh &h)
{
	rtk_dev_scene *ds = new rtk_dev_scene();
	hipDeviceProp_t prop;
	if (hipGetDevice(&ds->device) != hipSuccess || hipGetDeviceProperties(&prop, ds->device) != hipSuccess) {
		rtk_set_error("no usable HIP device: %s", hipGetErrorString(hipGetLastError()));
		delete ds;
		return nullptr;
	}
	ds->num_cus = prop.multiProcessorCount;
	ds->mesh_base = h.mesh_base;
	ds->max_depth = h.max_depth;
	ds->stack_entries = 3u * h.max_depth + 1u;   // at most three pushes per level of descent

	std::vector<uint32_t> prim_slot(h.mesh_base.empty() ? 0 : (size_t)h.mesh_base.back(), 0xffffffffu);
	for (size_t s = 0; s < h.tris.size(); s++) if (h.tris[s].prim < prim_slot.size()) prim_slot[h.tris[s].prim] = (uint32_t)s;

	bool ok = upload_vec(ds, h.nodes, &ds->view.nodes) && upload_vec(ds, h.tris, &ds->view.tris) &&
		upload_vec(ds, h.vertex_index, &ds->view.vertex_index) && upload_vec(ds, prim_slot, &ds->view.prim_slot) &&
		upload_vec(ds, h.slot_mesh, &ds->view.slot_mesh) && upload_vec(ds, h.slot_tri, &ds->view.slot_tri);
	if (!ok) {
		rtk_set_error("device allocation/copy failed: %s", hipGetErrorString(hipGetLastError()));
		rtk_dev_scene_free(ds);
		return nullptr;
	}
	ds->view.num_nodes = (uint32_t)h.nodes.size();
	ds->view.num_tris = (uint32_t)h.tris.size();
	ds->view.num_prims = (uint32_t)prim_slot.size();
	// boxes of a blob need not nest, so the bound of |plane| the packet kernel's slab margins rest on is taken over every node;
	// a scene with planes that are not finite (or beyond 1.7e38: their extent would not be) keeps to its exact nodes
	float bound = 0.0f;                    // (k_quantize applies the floor of 1 the packet kernels' empty-slot boxes need)
	bool finite = true;
	for (const DevNode &nd : h.nodes) {
		for (int k = 0; k < 4; k++) {
			if (nd.child[k] == RTK_REF_NONE) continue;
			const float v[6] = { nd.bx[0][k], nd.bx[1][k], nd.by[0][k], nd.by[1][k], nd.bz[0][k], nd.bz[1][k] };
			for (float x : v) { if (!(fabsf(x) <= 1.7e38f)) finite = false; else if (fabsf(x) > bound) bound = fabsf(x); }
		}
	}
	if (!finite) bound = INFINITY;
	// leaves of more than three triangles (the assembly packet kernel hands tiles that meet one to the C++ kernel)
	size_t leaves = 0, big = 0;
	for (const DevTri &t : h.tris) if (t.spare != 0u) { leaves++; if (t.spare > 3u) big++; }
	ds->big_leaf_fraction = leaves ? (double)big / (double)leaves : 0.0;
	if (rtk_quantize_nodes(ds, 0, nullptr, nullptr, bound) != RTK_AMD_OK || hipStreamSynchronize(0) != hipSuccess) {
		rtk_dev_scene_free(ds);
		return nullptr;
	}
	rtk_quantize_finish(ds);
	return ds;
}
