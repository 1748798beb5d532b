// rtk_dev.h -- internal: device-side BVH layout and the host-side scene object.
//
// Data layout in HBM (DESIGN.md "Device layout"):
//   DevNode  128 B = one L2 cache line: the reference's 4-wide SoA box block verbatim
//            (rtk.c:69-74: bounds_x/y/z[min|max][slot]) followed by four 32-bit child
//            references instead of four 64-bit byte offsets.
//   DevTri   48 B = three float4: the three vertex positions pre-gathered (the reference
//            chases leaf -> u8 index -> 16 B vertex, rtk.c:215-228), with the global
//            primitive id and the end-of-leaf flag riding in the w lanes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <vector>

#include "rtk.h"
#include "rtk_amd.h"

#define RTK_MAX_DEVICES 64       // per-device tables (workspaces, cached properties)
#define RTK_REF_NONE 0xffffffffu  // empty child slot / "no node"
#define RTK_REF_LEAF 0x80000000u  // leaf: low 31 bits = first triangle slot
#define RTK_TRI_LAST 1u           // DevTri.flags: last triangle of its leaf
// Per-launch scratch words: [0] ray pool head, [1..9] visit counters, then RTK_QUEUES work-queue
// heads, each on its own 128-byte line (one word serves only ~88 atomics/us on MI355X).
#define RTK_QUEUES 8
#define RTK_QUEUE_WORD(q) (16 + 16 * (q))
#define RTK_COUNTER_WORDS (16 + 16 * RTK_QUEUES)
// non-zero: a traversal stack overflowed (cannot happen for a validated tree). One word past the ones a launch clears: it
// stays set until rtk_trace_status has reported it.
#define RTK_ERROR_WORD RTK_COUNTER_WORDS

struct DevNode {
	float bx[2][4];
	float by[2][4];
	float bz[2][4];
	uint32_t child[4];
	// Front-to-back order of the four children for each of the eight direction-sign octants (octant o: bit 0 = x negative,
	// bit 1 = y negative, bit 2 = z negative), written by k_quantize from the child boxes alone (centre of the box along
	// the octant's diagonal; empty slots last). order[o >> 1], half (o & 1): bits 0-7 = the permutation (position q, nearest
	// first -> child slot, two bits each), bits 8-13 = for each pair of slots (0,1) (0,2) (0,3) (1,2) (1,3) (2,3) whether the
	// SECOND comes before the first. The packet kernel orders the children a tile enters by these instead of sorting entry
	// distances (any order gives the same hits: DESIGN.md 3.2); not part of the content hash, not exported.
	uint32_t order[4];
};
#define RTK_ORDER_PAIR_SHIFT 8
static_assert(sizeof(DevNode) == 128, "node must be one 128 B line");

// Compressed node for the per-lane kernels, 64 B = half a cache line (two children of one parent share a line):
// child boxes quantised to 8 bits per plane on a per-node, per-axis power-of-two grid anchored at the node's own
// min corner. Decoded plane = org + q * scale; low planes round down, high planes round up, so a decoded box
// always CONTAINS the exact one (checked in double precision when it is made, and again by the validator).
// Incoherent and shadow rays are bound by bytes through the fabric; this halves the bytes of a
// node visit. Hits do not change: culling only ever gets more conservative, the triangles decide the result.
struct DevNodeQ {
	float org[3];
	float scale[3];           // powers of two
	uint32_t q[3][2];         // [axis][0 = low planes, 1 = high planes], byte k = child k
	uint32_t child[4];
};
static_assert(sizeof(DevNodeQ) == 64, "quantised node must be half a 128 B line");

// 48 B of payload at a stride of RTK_TRI_STRIDE bytes. At 48 a record straddles two 128-B lines three times in
// eight (and two 64-B scalar-cache lines every other time); at 64 it never does, for 16 B more per triangle.
#ifndef RTK_TRI_STRIDE
#define RTK_TRI_STRIDE 48
#endif
struct DevTri {
	float v0[3]; uint32_t prim;   // global primitive id
	float v1[3]; uint32_t flags;  // RTK_TRI_LAST | mesh index << 8
	float v2[3]; uint32_t spare;  // first record of a leaf: number of triangles in the leaf
#if RTK_TRI_STRIDE == 64
	uint32_t pad[4];
#endif
};
static_assert(sizeof(DevTri) == RTK_TRI_STRIDE, "triangle record stride");

// A few words per scene that kernels read and k_quantize writes (device memory).
struct DevSceneConsts {
	float bound_abs;           // no plane of any node lies farther than this from the origin on its axis (>= 1: empty slots carry +1 / -1)
	uint32_t qnode_misfits;    // nodes whose child boxes do not fit the 8-bit grid (non-finite extents): the scene then keeps to its exact nodes
	float bound_raw;           // the same bound without the floor of 1 (the per-lane assembly kernels' slab margin is relative to it: a scene 1e-6 wide keeps a 1e-12 margin)
	uint32_t reserved;
};

// Everything a kernel needs to know about a scene (passed by value).
struct DevSceneView {
	const DevSceneConsts *consts;
	const DevNode *nodes;
	const DevNodeQ *qnodes;        // same tree, compressed boxes (may be NULL)
	const DevTri *tris;
	const uint32_t *vertex_index;  // [3*slot+k] original vertex index (rtk_vertex.index)
	const uint32_t *prim_slot;     // [prim] -> triangle slot
	const uint32_t *slot_mesh;     // [slot] -> mesh_index
	const uint32_t *slot_tri;      // [slot] -> triangle_index (per mesh)
	uint32_t num_nodes;
	uint32_t num_tris;
	uint32_t num_prims;
};

// Device memory a launch writes besides its outputs: work-queue heads and visit counters, the global
// part of the traversal stacks, and the ray-reordering buffers. One set per (scene, stream): launches on
// one stream are ordered by the stream, launches on different streams (or from different host threads)
// never share a set, so tracing one scene from many threads is safe (the reference's rtk_trace_ray is a
// pure function of a const scene, rtk.c:543-577).
struct LaunchScratch {
	hipStream_t stream = nullptr;
	unsigned long long *d_counter = nullptr;   // RTK_COUNTER_WORDS
	uint2 *d_spill = nullptr;
	size_t spill_entries_per_lane = 0;
	size_t spill_lanes = 0;
	void *d_sort = nullptr;                     // ray reordering scratch (RTK_TRACE_SORT_RAYS), grown on demand
	size_t sort_capacity = 0;                   // rays
	void *d_entries = nullptr;                  // packet kernels: entry lists of the image's 64x64-pixel blocks (PkBlockEntries), grown on demand
	size_t entries_capacity = 0;                // blocks
	volatile uint32_t *h_verdict = nullptr;     // pinned: where k_detect_check leaves (width, height) of an image nobody announced
	uint32_t *d_leftover = nullptr;             // tiles the assembly packet kernel hands to the C++ one, grown on demand
	size_t leftover_capacity = 0;               // tiles
};

struct rtk_dev_scene {
	int device = 0;
	DevSceneView view = {};
	// host copies kept for export / info
	std::vector<uint64_t> mesh_base;  // num_meshes + 1
	uint32_t max_depth = 0;
	uint32_t stack_entries = 0;
	uint32_t first_top = 0;                // device build, tile collapse: nodes [1, first_top) are the tiles', 0 and [first_top, n) the ones above them (0: one run)
	uint64_t total_bytes = 0;
	double build_ms = 0.0;
	double big_leaf_fraction = 0.0;        // leaves of more than three triangles (uploads; device builds make ~none): the assembly packet kernel hands those tiles back
	DevSceneConsts consts_readback = {};   // filled by the stream that ran k_quantize; read by rtk_quantize_finish after its synchronisation
	float bound_abs = 0.0f;
	float bound_raw = 0.0f;
	// owned device allocations
	std::vector<void *> allocs;
	// per-stream launch scratch, created on first use; the mutex covers the list and the enqueue of a launch
	std::mutex scratch_mutex;
	std::vector<LaunchScratch *> scratch;
	int num_cus = 0;
	// Device-built scenes make the four side arrays of the view (vertex_index, prim_slot, slot_mesh, slot_tri: what the expansion of
	// hit records, the validator and the exporter read -- never a traversal) on first use, not in every build: 52 of the 100
	// bytes per triangle the build's emit kernel wrote, one of them scattered (rtk_scene_side_arrays, rtk_build.hip).
	std::mutex side_mutex;
	bool side_ready = true;                    // (uploads arrive with the arrays)
	const uint32_t *d_vidx_in = nullptr;       // [3 * prim + k] original vertex indices in input order; NULL: every mesh has implicit indices
	const unsigned long long *d_mesh_base = nullptr;   // num_meshes + 1, on the device
};
// makes the side arrays if they are not there yet (synchronises `stream` the one time it has to work)
int rtk_scene_side_arrays(const rtk_dev_scene *ds, hipStream_t stream);

// -- error plumbing (rtk_capi.hip) --
void rtk_set_error(const char *fmt, ...);
#define RTK_HIP_CHECK(expr, ret)                                                         \
	do {                                                                                 \
		hipError_t e_ = (expr);                                                          \
		if (e_ != hipSuccess) {                                                          \
			rtk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return ret;                                                                  \
		}                                                                                \
	} while (0)

// -- upload (rtk_upload.hip) --
struct HostBvh {
	std::vector<DevNode> nodes;
	std::vector<DevTri> tris;
	std::vector<uint32_t> vertex_index;
	std::vector<uint32_t> slot_mesh, slot_tri;
	std::vector<uint64_t> mesh_base;
	uint32_t max_depth = 0;
};
int rtk_blob_to_host_bvh(const rtk_scene *scene, size_t avail, HostBvh *out);
rtk_dev_scene *rtk_dev_scene_from_host_bvh(const HostBvh &h);

// -- compressed node array (rtk_quant.hip): fills ds->view.qnodes from ds->view.nodes on `stream` --
// src: read the exact nodes from there and store them to ds->view.nodes as well; dst: compressed array the caller allocated
// Also writes every node's child order words and the scene constants (DevSceneConsts, allocated here). bound_hint: a bound of
// |plane| over all nodes the caller already knows (uploads: computed on the host, boxes of a blob need not nest); the root's
// own planes are always taken in.
// only_first: finish just the first so many nodes (the device build's tile collapse has finished the others itself);
// keep_consts: the constants block is already set up (rtk_scene_consts) and holds counts that must survive.
int rtk_quantize_nodes(rtk_dev_scene *ds, hipStream_t stream, const DevNode *src = nullptr, DevNodeQ *dst = nullptr, float bound_hint = 0.0f,
	uint32_t only_first = 0xffffffffu, bool keep_consts = false, bool readback = true);   // (bound_hint: 0 = none; the floor of 1 is applied inside; readback: enqueue the copy of the constants to the host -- the device build brings them home with its other results)
int rtk_scene_consts(rtk_dev_scene *ds, hipStream_t stream);
void rtk_quantize_finish(rtk_dev_scene *ds);   // after that stream has been synchronised

// -- radix sort shared with the builder (rtk_build.hip) --
size_t rtk_sort_scratch_words(uint32_t n);
bool rtk_sort_pairs_async(unsigned long long *keys_a, unsigned long long *keys_b, uint32_t *vals_a, uint32_t *vals_b,
	uint32_t n, uint32_t key_bits, uint32_t *scratch, hipStream_t stream);
bool rtk_sort_words_async(unsigned long long *keys_a, unsigned long long *keys_b, uint32_t n, uint32_t first_bit, uint32_t last_bit,
	uint32_t *scratch, hipStream_t stream);

// -- trace launches (rtk_trace.hip) --
int rtk_launch_trace(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n, rtk_hit_record *d_hits,
	uint8_t *d_occluded, const rtk_trace_opts *opts, hipStream_t stream, bool any_hit, rtk_trace_counters *counted,
	const rtk_dev_filter *filter = nullptr, rtk_hit_record *d_cand = nullptr, uint32_t *d_cand_count = nullptr, uint32_t cand_k = 0,
	rtk_packet_counters *pk_counted = nullptr);
int rtk_detect_image(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n, hipStream_t stream, uint32_t *w_out, uint32_t *h_out);
int rtk_trace_status(const rtk_dev_scene *ds, hipStream_t stream);
void rtk_scratch_free(LaunchScratch *s);
void rtk_scene_drop_stream(rtk_dev_scene *ds, hipStream_t stream);   // the stream is about to be destroyed (and has been synchronised)
// h_status (host-visible): also receives the stream's launch-error word (see rtk_trace_status), or is left alone if the stream has none.
// ticket != 0 and n <= 256: the word becomes (ticket << 32 | error) once every result of the launch is visible to the host.
int rtk_launch_trace_one(const rtk_dev_scene *ds, const rtk_ray *d_ray, rtk_hit *d_hit, uint8_t *d_mask, hipStream_t stream,
	unsigned long long *h_status, uint32_t ticket);
int rtk_launch_expand(const rtk_dev_scene *ds, const rtk_hit_record *d_records, size_t n, rtk_hit *d_hits,
	uint8_t *d_mask, hipStream_t stream, unsigned long long *h_status = nullptr, uint32_t ticket = 0);
