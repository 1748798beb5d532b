// rtk_trace_shared.h -- pieces shared by the two traversal kernels (rtk_trace.hip: one ray
// per lane; rtk_trace_packet.hip: one 8x8 tile per wave).
#pragma once

#include "rtk_dev.h"

#include <stddef.h>

#ifndef LDS_STACK
#define LDS_STACK 15           // entries per lane held in LDS: 30 KB per workgroup, so that FIVE workgroups share a CU's 160 KB (with 16
                               // entries = 32 KB only four are placed: -5 % on incoherent rays, -4 % on shadow rays; 14 and 13 spill more)
#endif
#define TRACE_WAVES_PER_BLOCK 4
#define TRACE_BLOCK_THREADS (64 * TRACE_WAVES_PER_BLOCK)
#define WAVES_PER_BLOCK TRACE_WAVES_PER_BLOCK
#define BLOCK_THREADS TRACE_BLOCK_THREADS

struct TraceParams {
	DevSceneView sc;
	const rtk_ray *rays;
	rtk_hit_record *hits;
	uint8_t *occluded;
	unsigned long long *counter;   // [0] pool head, [1..6] visit counters
	uint2 *spill;
	unsigned long long n;
	uint32_t spill_stride;         // lanes in the launch
	uint32_t spill_cap;            // entries per lane in spill
	uint32_t image_w, image_h;     // 0 = no tiling
	uint32_t refill_min;
	uint32_t dynamic;
	uint32_t node_exit;            // leave the node loop when fewer lanes than this still need node steps and a leaf is waiting
	const unsigned long long *perm; // optional: trace rays in this order (sorted words of the ray reordering, ray number in the low 32 bits); results go to the ray's own slot
	// built-in candidate filters (rtk_dev_filter, FILT kernels only); all optional
	const rtk_hit_record *after;   // per ray: only candidates that come after (t, prim) in (t, prim) order
	const uint32_t *ignore_prim;   // per ray: global primitive id that is never a candidate
	const uint32_t *mesh_mask;     // bit m set = triangles of mesh m are candidates
	uint32_t mesh_mask_bits;       // meshes covered by mesh_mask; meshes beyond it are not candidates
	rtk_hit_record *cand;          // MODE 2: cand_k records per ray, the closest candidates in (t, prim) order
	uint32_t *cand_count;          // MODE 2: how many of them are valid
	uint32_t cand_k;
	uint32_t tile_blocks;          // image batches: tiles are numbered block by block (8x8 tiles = 64x64 pixels), not row by row
	const unsigned long long *n_indirect;   // rtk_trace_kernel: if set, the number of rays is read from here when the kernel starts (the list the assembly
	                                        // per-lane kernel left over: `perm` then points at it, its length is counter[RTK_LANE_LEFTOVER_WORD])
	const struct PkBlockEntries *entries;   // packet kernels only: per 64x64-pixel block the entry points of its rays (or NULL: every tile starts at the root)
	const uint32_t *tile_list;     // packet kernel only: trace the tiles listed here (counter[RTK_LEFTOVER_COUNT_WORD] of them, dealt through
	                               // counter[RTK_LEFTOVER_HEAD_WORD]) instead of all tiles: what the assembly kernel handed back
};

// Scratch counter words of the hand-over from the assembly packet kernel (rtk_packet_hot.S) to the C++ one; cleared with the rest
#define RTK_LEFTOVER_COUNT_WORD 10
#define RTK_LEFTOVER_HEAD_WORD 11
// ... and from the assembly per-lane kernels (rtk_lane_hot.S) to rtk_trace_kernel: how many rays they left over
#define RTK_LANE_LEFTOVER_WORD 12

// Entry points shared by the 64 tiles of a 64x64-pixel block (rtk_packet_entries_kernel, rtk_trace_packet.hip): the block's
// rays are bounded by a box of origins and a box of reciprocal directions (one sign per axis), that BEAM is walked down the
// top of the tree by the interval form of the slab test, and the nodes it reaches at the cut are listed front to back by a
// lower bound of their entry distance. A tile whose rays lie inside the beam starts at these instead of at the root
// (config 2: 26.7 wave node steps per tile instead of 34.3) and stops at the first entry that lies behind every lane's hit.
// Every entry is an INNER node: a node with a leaf child is listed itself and not opened -- a listed leaf would cost every
// tile of the block a triangle test whether or not any of its rays enters the leaf's box (measured: +25 % triangle tests).
// (Listed in the order a traversal from the root reaches them, with the smallest bound of the rest for the early exit, tiles
// visited more nodes and were slower than from the root: profiles/r04_packet_entries.log.)
#define PK_MAX_ENTRIES 56
struct PkBlockEntries {
	float olo[3], ohi[3];          //  0  the beam: origins ...
	float rlo[3], rhi[3];          // 24  ... and reciprocal directions of the block's rays
	uint32_t count;                // 48  entries; 0 = none (rays of mixed signs, not tame, too many entries): tiles start at the root
	uint32_t pad[3];
	struct { uint32_t ref; float tlo; } e[PK_MAX_ENTRIES];   // 64  node reference, lower bound of the entry distance (ascending)
};
static_assert(sizeof(PkBlockEntries) == 512 && offsetof(PkBlockEntries, count) == 48 && offsetof(PkBlockEntries, e) == 64, "rtk_packet_hot.S reads this layout");

// Kernel argument of rtk_packet_hot (rtk_packet_hot.S reads these offsets)
struct PkHotParams {
	const void *nodes;             //  0
	const void *tris;              //  8
	const rtk_ray *rays;           // 16
	rtk_hit_record *hits;          // 24
	unsigned long long *counter;   // 32
	uint32_t *leftover;            // 40  tile numbers handed to the C++ kernel
	uint32_t num_blocks;           // 48  64x64-pixel blocks of the image
	uint32_t image_w;              // 52
	uint32_t blocks_per_row;       // 56
	uint32_t bpr_magic;            // 60  ceil(2^32 / blocks_per_row): block / blocks_per_row = mul_hi(block, magic)
	float bound_abs;               // 64  max(largest |plane| of the scene, 1)
	uint32_t pad;
	const PkBlockEntries *entries; // 72  per block (numbered like the tiles' blocks), or NULL
};
static_assert(sizeof(PkHotParams) == 80 && offsetof(PkHotParams, num_blocks) == 48 && offsetof(PkHotParams, bound_abs) == 64 && offsetof(PkHotParams, entries) == 72,
	"rtk_packet_hot.S reads this layout");

// Kernel argument of rtk_lane_hot_closest / rtk_lane_hot_any (rtk_lane_hot.S reads these offsets)
struct LnHotParams {
	const void *qnodes;            //  0  64-byte compressed nodes
	const void *tris;              //  8
	const rtk_ray *rays;           // 16
	void *out;                     // 24  closest: rtk_hit_record per ray; any-hit: one byte per ray
	unsigned long long *counter;   // 32
	unsigned long long *leftover;  // 40  ray numbers handed to rtk_trace_kernel (8-byte words, the number in the low half)
	const unsigned long long *perm;// 48  optional ray order (sort words, ray number in the low half)
	uint32_t n;                    // 56
	uint32_t refill_min;           // 60
	uint32_t node_exit;            // 64
	float bound_abs;               // 68  largest |plane| of the scene (DevSceneConsts::bound_raw)
	uint2 *spill;                  // 72  stack entries beyond a lane's LDS column: [entry][lane of the launch]
	uint32_t spill_stride;         // 80  lanes of the launch (>= lanes of this kernel's grid)
	uint32_t spill_cap;            // 84  entries per lane
};
static_assert(sizeof(LnHotParams) == 88 && offsetof(LnHotParams, n) == 56 && offsetof(LnHotParams, bound_abs) == 68 && offsetof(LnHotParams, spill) == 72 &&
	offsetof(LnHotParams, spill_cap) == 84, "rtk_lane_hot.S reads this layout");

// DevTri.flags: bit 0 = last triangle of its leaf, bits 8.. = mesh index (for the mesh-mask filter)
#define RTK_TRI_MESH_SHIFT 8

// _mm_min_ps/_mm_max_ps semantics (second operand when the compare is false, NaN included)
__device__ __forceinline__ float sse_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sse_max(float a, float b) { return a > b ? a : b; }


// Row-major image -> 8x8 pixel tiles, so that the 64 lanes of a wave share BVH nodes.
// With `blocks` the tiles themselves are numbered block by block (64 consecutive tiles = one 64x64 pixel block), so that
// the tiles a work queue hands out one after the other -- to waves that run at the same time on one XCD -- are
// neighbours in the image and walk the same part of the tree (L2 / scalar cache hits instead of fabric traffic).
__device__ __forceinline__ unsigned long long map_index(unsigned long long i, uint32_t w, uint32_t h, uint32_t blocks = 0)
{
	if (w == 0) return i;
	const unsigned long long tile = i >> 6;
	const uint32_t in = (uint32_t)i & 63u;
	const uint32_t tiles_per_row = w >> 3;
	unsigned long long ty;
	uint32_t tx;
	if (blocks) {
		const unsigned long long blk = tile >> 6;
		const uint32_t b = (uint32_t)tile & 63u, blocks_per_row = w >> 6;
		const unsigned long long by = blk / blocks_per_row;
		tx = (uint32_t)(blk - by * blocks_per_row) * 8u + (b & 7u);
		ty = by * 8u + (b >> 3);
	} else {
		ty = tile / tiles_per_row;
		tx = (uint32_t)(tile - ty * tiles_per_row);
	}
	const unsigned long long x = (unsigned long long)tx * 8u + (in & 7u);
	const unsigned long long y = ty * 8u + (in >> 3);
	return y * w + x;
}


// the hand-written per-lane kernels (rtk_lane_hot.S; loader in rtk_trace.hip)
bool rtk_lane_hot_available(int device, int *blocks_per_cu);
int rtk_lane_hot_launch(int device, const LnHotParams &hp, unsigned blocks, hipStream_t stream, bool any_hit);
// rtk_trace_packet.hip
int rtk_packet_occupancy(bool counted);
void rtk_packet_launch(const TraceParams &p, unsigned blocks, hipStream_t stream, bool counted);
// the blocks' entry lists (p.image_w / image_h, 64x64-pixel blocks numbered row by row) into `out`, one wave per block
void rtk_packet_entries_launch(const TraceParams &p, PkBlockEntries *out, float bound_abs, unsigned target, unsigned max_levels, hipStream_t stream);
// the hand-written kernel (rtk_packet_hot.S): can this device run it (module loads), and its launch. blocks_per_cu: resident workgroups.
// beam = 1: rtk_packet_beam, the variant whose node test is the interval test of the tile's own beam (one child plane per lane);
// 2: rtk_packet_beam2, two adjacent tiles per wave (the two halves of the wave test a node for the two tiles' beams);
// 3: rtk_packet_count2, the counting form of rtk_packet_beam2 (the same source assembled with -DRTK_COUNT)
bool rtk_packet_hot_available(int device, int *blocks_per_cu, int beam = 0);
int rtk_packet_hot_launch(int device, const PkHotParams &hp, unsigned blocks, hipStream_t stream, int beam = 0);
