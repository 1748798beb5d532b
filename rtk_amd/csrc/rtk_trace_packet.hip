// rtk_trace_packet.hip -- wave-packet BVH4 traversal for image-shaped (coherent) batches.
//
// Same results as rtk_trace_kernel (rtk_trace.hip) -- the per-lane arithmetic of the slab
// test (rtk.c:457-472), the triangle test (rtk.c:284-364) and the group-of-four double
// precision rule (rtk.c:302-336) is identical -- but the 64 rays of an 8x8 pixel tile walk
// the tree TOGETHER:
//   * the node / triangle being processed is wave-uniform, so its 128 B / 48 B come through
//     the scalar cache (s_load_dwordx16) into SGPRs: one request per wave instead of one
//     per lane, no per-lane address arithmetic, and the data feeds VALU ops as SGPR operands;
//   * the traversal stack is wave-uniform and lives in ONE VGPR (entry i in lane i,
//     v_writelane/v_readlane); only the per-lane entry distance of a pushed child goes to
//     LDS ([entry][lane] floats) so that each lane culls exactly like its own traversal would
//     (rtk.c:432: skip an entry that starts behind the lane's current hit);
//   * every lane keeps its own "live" bit: it takes part in a node or leaf only if its own
//     slab test admitted that child -- per-lane visit semantics, packet-wide control flow;
//   * child order is taken from the first lane that hits (scalar sort on SALU).
// Per-lane traversal wastes ~55 % (nodes) and ~80 % (triangles) of the lanes on config 2
// because neighbouring rays reach leaves at different times (profiles/r01b_coherent_*);
// a tile of the 4096^2 frame is ~3e-3 wide against ~2e-2 triangles, so a packet visits
// little more than a single ray does.
//
// FLOATING POINT: compiled with -ffp-contract=off like rtk_trace.hip.
#include "rtk_dev.h"
#include "rtk_trace_shared.h"

#include <math.h>

#define PK_LDS_STACK 16            // per-lane entry distances held in LDS
#define PK_WAVE_STACK 64           // wave-uniform references held in one VGPR
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 7             // waves per SIMD the register allocator must leave room for: 69 VGPRs without the SLP vectoriser (Makefile), so 7 fit (72): +4.5 % over 6; 8 (64 VGPRs, spills) is back at 6's rate
#endif

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float asf(int v) { return __int_as_float(v); }

// 128 B node through the scalar cache. The six plane rows are fetched at byte offsets chosen once
// per packet from the direction sign bits (near row first), exactly like the reference indexes
// bounds_x[sign_x] (rtk.c:458-463): no selects afterwards.
// `ord`: the word of DevNode::order that holds the front-to-back child order for this packet's direction octant.
struct PkNode { i32x4 nx, fx, ny, fy, nz, fz, ch; int ord; };
__device__ __forceinline__ void s_load_node(const char *addr, uint32_t onx, uint32_t ofx, uint32_t ony, uint32_t ofy,
	uint32_t onz, uint32_t ofz, uint32_t oord, PkNode &n)
{
	asm volatile(
		"s_load_dwordx4 %0, %8, %9\n\t"
		"s_load_dwordx4 %1, %8, %10\n\t"
		"s_load_dwordx4 %2, %8, %11\n\t"
		"s_load_dwordx4 %3, %8, %12\n\t"
		"s_load_dwordx4 %4, %8, %13\n\t"
		"s_load_dwordx4 %5, %8, %14\n\t"
		"s_load_dwordx4 %6, %8, 0x60\n\t"
		"s_load_dword %7, %8, %15\n\t"
		"s_waitcnt lgkmcnt(0)"
		: "=&s"(n.nx), "=&s"(n.fx), "=&s"(n.ny), "=&s"(n.fy), "=&s"(n.nz), "=&s"(n.fz), "=&s"(n.ch), "=&s"(n.ord)
		: "s"(addr), "s"(onx), "s"(ofx), "s"(ony), "s"(ofy), "s"(onz), "s"(ofz), "s"(oord)
		: "memory");
}

// 48 B triangle: a = v0.xyz, prim, v1.xyz, flags; b = v2.xyz, spare
__device__ __forceinline__ void s_load_tri(const char *addr, i32x8 &a, i32x4 &b)
{
	asm volatile(
		"s_load_dwordx8 %0, %2, 0x0\n\t"
		"s_load_dwordx4 %1, %2, 0x20\n\t"
		"s_waitcnt lgkmcnt(0)"
		: "=&s"(a), "=&s"(b) : "s"(addr) : "memory");
}

// One float of this wave's LDS stack by its LDS byte address (the low half of its flat address): a real ds_read_b32.
__device__ __forceinline__ float lds_read_f32(uint32_t addr)
{
	float v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
	return v;
}

__device__ __forceinline__ uint32_t stack_write(uint32_t stack, uint32_t value, uint32_t entry, uint32_t lane)
{
	// (value and entry are wave-uniform: one v_writelane_b32 with the lane select in M0, not a compare and a select per lane)
	// (no builtin for it in this compiler; M0 because a VALU instruction reads one scalar register besides M0. hipcc sets M0
	// itself right before every use, so nothing of its own is live in it here)
	asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(stack) : "s"(value), "s"(entry) : "m0");
	return stack;
}

__device__ __forceinline__ int sort_key(float f)
{
	// float order -> signed int order
	const int b = __float_as_int(f);
	return b ^ ((b >> 31) & 0x7fffffff);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct PkLane {
	// ray. The slab test of the fast path is t = plane * (1/d) - c with c = o * (1/d) -+ margin folded in (see pk_slab2): per axis
	// the pair (1/d, c of the plane row fetched first), then the constants of the rows fetched second, and min_t. Pairs, so that
	// v_pk_fma_f32 can take both of its per-lane operands from 64-bit registers by half (op_sel) without duplicating any value.
	f32x2 px, py, pz;      // (1/d, c0) per axis
	f32x2 q1, q2;          // (c1x, c1y), (c1z, min_t)
	float tmax;
	float sox, soy, soz, shx, shy, shz;
	bool kz0, kz1, sx, sy, sz;
	// best hit
	float t, u, v;
	uint32_t prim;
};

// One triangle against every live lane. DBL: use the double-precision edge functions.
// KZ = 0,1,2: every lane of the packet has this dominant axis, the permutation to (kx,ky,kz)
// (rtk.c:232-243) is a compile-time pick of scalar registers; KZ = 3: per lane.
// Returns (per lane) whether a float edge function was exactly zero.
template <bool DBL, int KZ>
__device__ __forceinline__ bool pk_triangle(PkLane &L, bool lanes, const i32x8 &a, const i32x4 &b)
{
	float ax, ay, az, bx, by, bz, cx, cy, cz;
	if (KZ == 0) {
		ax = asf(a[1]); ay = asf(a[2]); az = asf(a[0]);
		bx = asf(a[5]); by = asf(a[6]); bz = asf(a[4]);
		cx = asf(b[1]); cy = asf(b[2]); cz = asf(b[0]);
	} else if (KZ == 1) {
		ax = asf(a[2]); ay = asf(a[0]); az = asf(a[1]);
		bx = asf(a[6]); by = asf(a[4]); bz = asf(a[5]);
		cx = asf(b[2]); cy = asf(b[0]); cz = asf(b[1]);
	} else if (KZ == 2) {
		ax = asf(a[0]); ay = asf(a[1]); az = asf(a[2]);
		bx = asf(a[4]); by = asf(a[5]); bz = asf(a[6]);
		cx = asf(b[0]); cy = asf(b[1]); cz = asf(b[2]);
	} else {
		const float A0 = asf(a[0]), A1 = asf(a[1]), A2 = asf(a[2]);
		const float B0 = asf(a[4]), B1 = asf(a[5]), B2 = asf(a[6]);
		const float C0 = asf(b[0]), C1 = asf(b[1]), C2 = asf(b[2]);
		ax = L.kz0 ? A1 : (L.kz1 ? A2 : A0); ay = L.kz0 ? A2 : (L.kz1 ? A0 : A1); az = L.kz0 ? A0 : (L.kz1 ? A1 : A2);
		bx = L.kz0 ? B1 : (L.kz1 ? B2 : B0); by = L.kz0 ? B2 : (L.kz1 ? B0 : B1); bz = L.kz0 ? B0 : (L.kz1 ? B1 : B2);
		cx = L.kz0 ? C1 : (L.kz1 ? C2 : C0); cy = L.kz0 ? C2 : (L.kz1 ? C0 : C1); cz = L.kz0 ? C0 : (L.kz1 ? C1 : C2);
	}
	// move the origin, shear (rtk.c:256-292)
	const float v0x = ax - L.sox, v0y = ay - L.soy, v0z = az - L.soz;
	const float v1x = bx - L.sox, v1y = by - L.soy, v1z = bz - L.soz;
	const float v2x = cx - L.sox, v2y = cy - L.soy, v2z = cz - L.soz;
	const float x0 = v0x + L.shx * v0z, y0 = v0y + L.shy * v0z, z0 = L.shz * v0z;
	const float x1 = v1x + L.shx * v1z, y1 = v1y + L.shy * v1z, z1 = L.shz * v1z;
	const float x2 = v2x + L.shx * v2z, y2 = v2y + L.shy * v2z, z2 = L.shz * v2z;
	float u, v, w;
	bool zero = false;
	if (DBL) {
		const double xd0 = x0, yd0 = y0, xd1 = x1, yd1 = y1, xd2 = x2, yd2 = y2;   // rtk.c:307-335
		u = (float)(xd1 * yd2 - yd1 * xd2);
		v = (float)(xd2 * yd0 - yd2 * xd0);
		w = (float)(xd0 * yd1 - yd0 * xd1);
	} else {
		u = x1 * y2 - y1 * x2;                                                    // rtk.c:298-300
		v = x2 * y0 - y2 * x0;
		w = x0 * y1 - y0 * x1;
		zero = u == 0.0f || v == 0.0f || w == 0.0f;
	}
	const bool neg = sse_min(sse_min(u, v), w) < 0.0f;                           // rtk.c:340-342
	const bool pos = sse_max(sse_max(u, v), w) > 0.0f;
	// nobody in the packet passes the sign test: skip the divide and the rest (rtk.c:344 does the same per group)
	if (__builtin_amdgcn_ballot_w64(lanes && !(neg && pos)) == 0ull) return zero;
	const float det = (u + v) + w;                                               // rtk.c:346-353
	const float rcp = 1.0f / det;
	float zz = u * z0;
	zz = zz + v * z1;
	zz = zz + w * z2;
	const float t = zz * rcp;
	const uint32_t prim = (uint32_t)a[3];
	const bool ok = lanes && !(neg && pos) && t > L.q2.y && t < L.tmax;          // rtk.c:354
	const bool take = ok && (t < L.t || (t == L.t && prim < L.prim));            // rtk.c:371 + canonical ties
	// selects, not branches: a branch costs three exec-mask regions per triangle on the scalar unit
	const float nu = u * rcp, nv = v * rcp;
	L.t = take ? t : L.t; L.u = take ? nu : L.u; L.v = take ? nv : L.v; L.prim = take ? prim : L.prim;
	return zero;
}

__device__ __forceinline__ f32x2 mk2(int a, int b) { f32x2 r; r.x = asf(a); r.y = asf(b); return r; }

// rows (two children's planes, scalar registers) * A[sa] - B[sb] for both children in one instruction; A, B: 64-bit per-lane
// pairs of which op_sel picks the same half for both results. One rounding (it is an fma), which the margin in c accounts for.
#define PK_FMA(dst_, rows_, A_, sa_, B_, sb_) \
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #sa_ "," #sb_ "] op_sel_hi:[1," #sa_ "," #sb_ "] neg_lo:[0,0,1] neg_hi:[0,0,1]" \
		: "=v"(dst_) : "s"(rows_), "v"(A_), "v"(B_))

__device__ __forceinline__ float pk_max2(float a, float b) { float r; asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float pk_min2(float a, float b) { float r; asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float pk_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float pk_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Slab test of children C0 and C0+1 for every lane; node rows in SGPRs (near rows first when the
// packet's direction signs are uniform). Two children at a time. Writes the per-lane entry distance, or
// NaN where the lane does not enter the child.
//
// FAST (every ray of the packet is "tame", see the kernel): plane parameter t = plane * (1/d) - c in ONE v_pk_fma_f32 per
// row and pair of children (the reference's (plane - o) * (1/d), rtk.c:458-463, is two), where c = o * (1/d) + m for the row
// that bounds the interval from below and c = o * (1/d) - m for the other, m = 2^-21 * |1/d| * (|o| + B), B = the largest
// |plane| of the scene. The two forms differ by rounding only: |fl(fl(p - o) * r) - (p - o) r| <= 2.01 u |p - o||r| and
// |fma(p, r, -fl(o r)) - (p - o) r| <= u |o r| + 1.01 u |p - o||r| (u = 2^-24), together below 5 u (|o| + B)|r| < m with the
// rounding of c itself. So every near parameter here is <= the reference's and every far parameter >=: each child the exact
// test admits is admitted (as with the compressed nodes of the per-lane kernels, culling only gets more conservative and the
// triangles decide the result), at 12 instead of 24 packed instructions per node step.
// !FAST: the reference's arithmetic verbatim, SSE operand order of min / max, child words looked at.
template <bool UNIFORM_SIGN, bool FAST, int C0>
__device__ __forceinline__ void pk_slab2(const PkLane &L, const PkNode &n, bool live, float &pay0, float &pay1)
{
	const f32x2 rx0 = mk2(n.nx[C0], n.nx[C0 + 1]), rx1 = mk2(n.fx[C0], n.fx[C0 + 1]);
	const f32x2 ry0 = mk2(n.ny[C0], n.ny[C0 + 1]), ry1 = mk2(n.fy[C0], n.fy[C0 + 1]);
	const f32x2 rz0 = mk2(n.nz[C0], n.nz[C0 + 1]), rz1 = mk2(n.fz[C0], n.fz[C0 + 1]);
	f32x2 nx, fx, ny, fy, nz, fz;
	const float tmin = L.q2.y;
	if (FAST) {
		PK_FMA(nx, rx0, L.px, 0, L.px, 1); PK_FMA(fx, rx1, L.px, 0, L.q1, 0);
		PK_FMA(ny, ry0, L.py, 0, L.py, 1); PK_FMA(fy, ry1, L.py, 0, L.q1, 1);
		PK_FMA(nz, rz0, L.pz, 0, L.pz, 1); PK_FMA(fz, rz1, L.pz, 0, L.q2, 0);
	} else {
		// the origin back out of its sheared permutation (rtk.c:232-243 the other way round), then rtk.c:458-463
		const float a_ = L.sox, b_ = L.soy, c_ = L.soz;        // (values first: a conditional between members is a conditional between ADDRESSES, and the whole lane state then lives in scratch)
		const float ox = L.kz0 ? c_ : (L.kz1 ? b_ : a_), oy = L.kz0 ? a_ : (L.kz1 ? c_ : b_), oz = L.kz0 ? b_ : (L.kz1 ? a_ : c_);
		nx = (rx0 - ox) * L.px.x; fx = (rx1 - ox) * L.px.x;
		ny = (ry0 - oy) * L.py.x; fy = (ry1 - oy) * L.py.x;
		nz = (rz0 - oz) * L.pz.x; fz = (rz1 - oz) * L.pz.x;
	}
	if (!UNIFORM_SIGN) {
		// rows were fetched as (min, max); pick near/far per lane
		const f32x2 ax = nx, bx = fx, ay = ny, by = fy, az = nz, bz = fz;
		nx = L.sx ? bx : ax; fx = L.sx ? ax : bx;
		ny = L.sy ? by : ay; fy = L.sy ? ay : by;
		nz = L.sz ? bz : az; fz = L.sz ? az : bz;
	}
	float tn0, tf0, tn1, tf1;
	if (FAST) {
		// (written out: behind fmaxf / fminf hipcc first quiets every operand that comes out of an asm statement -- a v_max_f32 x, x
		// each, twelve per pair of children; no NaN can arise for tame rays, and v_max / v_min quiet their operands anyway)
		tn0 = pk_max3(pk_max2(nx.x, ny.x), nz.x, tmin); tf0 = pk_min3(pk_min2(fx.x, fy.x), fz.x, L.t);
		tn1 = pk_max3(pk_max2(nx.y, ny.y), nz.y, tmin); tf1 = pk_min3(pk_min2(fx.y, fy.y), fz.y, L.t);
	} else {
		tn0 = sse_max(sse_max(nx.x, ny.x), sse_max(nz.x, tmin)); tf0 = sse_min(sse_min(fx.x, fy.x), sse_min(fz.x, L.t));   // rtk.c:464-465
		tn1 = sse_max(sse_max(nx.y, ny.y), sse_max(nz.y, tmin)); tf1 = sse_min(sse_min(fx.y, fy.y), sse_min(fz.y, L.t));
	}
	const float miss = __builtin_nanf("");
	if (FAST) {
		// empty slots carry inverted boxes (every producer of DevNode writes +1 / -1 there) and FAST rays are tame: see `special`
		pay0 = (live && tn0 <= tf0) ? tn0 : miss;
		pay1 = (live && tn1 <= tf1) ? tn1 : miss;
	} else {
		pay0 = (live && tn0 <= tf0 && (uint32_t)n.ch[C0] != RTK_REF_NONE) ? tn0 : miss;
		pay1 = (live && tn1 <= tf1 && (uint32_t)n.ch[C0 + 1] != RTK_REF_NONE) ? tn1 : miss;
	}
}

// All triangles of one leaf for the packet, in the reference's groups of four (rtk.c:212, 302-336).
template <int KZ, bool COUNT>
__device__ __forceinline__ void pk_leaf(PkLane &L, bool live, const char *tris, uint32_t slot0, uint32_t lane,
	unsigned long long *counter, uint32_t &c_tris)
{
	i32x8 ta;
	i32x4 tb;
	s_load_tri(tris + (size_t)slot0 * RTK_TRI_STRIDE, ta, tb);
	const uint32_t n = (uint32_t)tb[3];                                // leaf size rides in the first record
	if (n == 1u) {
		// the usual case (the device build's leaves hold 1.008 triangles on average): a lone triangle is a partial
		// group, so it takes the double-precision edge functions, and there is nothing to snapshot or redo
		if (COUNT && live) c_tris++;
		if (COUNT && lane == 0) atomicAdd(counter + 8, 1ull);
		pk_triangle<true, KZ>(L, live, ta, tb);
		return;
	}
	for (uint32_t g = 0; g < n; g += 4u) {
		const uint32_t m = n - g < 4u ? n - g : 4u;
		const bool force = m < 4u;                                     // padding slots make the whole group double (rtk.c:306)
		const float sn_t = L.t, sn_u = L.u, sn_v = L.v;
		const uint32_t sn_prim = L.prim;
		bool zero_seen = false;
		for (uint32_t j = 0; j < m; j++) {
			if (g + j != 0u) s_load_tri(tris + (size_t)(slot0 + g + j) * RTK_TRI_STRIDE, ta, tb);
			if (COUNT && live) c_tris++;
			if (COUNT && lane == 0) atomicAdd(counter + 8, 1ull);
			if (force) pk_triangle<true, KZ>(L, live, ta, tb);
			else zero_seen |= pk_triangle<false, KZ>(L, live, ta, tb);
		}
		const bool redo = live && zero_seen;
		if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
			// an exact zero in a full group: those lanes redo the group in double (rtk.c:302-336)
			if (redo) { L.t = sn_t; L.u = sn_u; L.v = sn_v; L.prim = sn_prim; }
			for (uint32_t j = 0; j < m; j++) {
				s_load_tri(tris + (size_t)(slot0 + g + j) * RTK_TRI_STRIDE, ta, tb);
				pk_triangle<true, KZ>(L, redo, ta, tb);
			}
		}
	}
}

// Push one child: its reference into the wave-uniform stack register, every lane's own entry distance into
// LDS (or the spill area). A push that does not fit (impossible for a tree: the launcher refuses scenes
// needing more than PK_WAVE_STACK entries and sizes the spill area from the depth) is dropped without
// advancing sp and remembered in a wave-uniform flag that is reported once per tile.
// PK_PUSH_LDS: the caller has checked that three more entries fit the LDS part (one scalar compare per node step
// instead of two compares and two branches per push).
#define PK_PUSH_LDS(dist_, ref_)                                                                                  \
	do {                                                                                                          \
		lds_t[sp][lane] = (dist_);                                                                                \
		stack = stack_write(stack, (ref_), sp, lane);                                                             \
		sp++;                                                                                                     \
	} while (0)
#define PK_PUSH(dist_, ref_)                                                                                      \
	do {                                                                                                          \
		if (sp < PK_LDS_STACK + p.spill_cap) {                                                                    \
			if (sp < PK_LDS_STACK) lds_t[sp][lane] = (dist_);                                                     \
			else { spill_t[(size_t)(sp - PK_LDS_STACK) * p.spill_stride + glane] = (dist_); if (COUNT) c_spills++; } \
			stack = stack_write(stack, (ref_), sp, lane);                                                         \
			sp++;                                                                                                 \
		} else overflow = true;                                                                                   \
	} while (0)

template <bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK_THREADS, PK_MIN_WAVES) rtk_trace_packet_kernel(TraceParams p)
{
	__shared__ float s_t[TRACE_WAVES_PER_BLOCK][PK_LDS_STACK][64];

	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	float (*lds_t)[64] = s_t[wave];
	const uint32_t lds_base = (uint32_t)(size_t)&lds_t[0][lane];      // LDS byte address of this lane's column
	const uint32_t glane = blockIdx.x * TRACE_BLOCK_THREADS + threadIdx.x;
	float *const spill_t = reinterpret_cast<float *>(p.spill);
	const char *const nodes = reinterpret_cast<const char *>(p.sc.nodes);
	const char *const tris = reinterpret_cast<const char *>(p.sc.tris);
	// largest |plane| of the scene (>= 1: empty slots), for the slab margins; a scene without such a bound (non-finite planes)
	// or too far out for the margins (see `tame`) sends every ray down the exact path
	const float bound_abs = fmaxf(p.sc.consts->bound_abs, 1.0f);
	const bool bound_ok = bound_abs < 0x1p19f;

	// Tiles are dealt through RTK_QUEUES work queues (tile t belongs to queue t % RTK_QUEUES); a wave
	// starts on the queue of its workgroup (blockIdx % 8 = one XCD under the usual round-robin
	// placement, speed only) and moves on to the next queue when its own is empty. One atomic per
	// tile keeps the load balanced (taking 8/16/32 tiles per atomic cost 6 %/21 %/46 %), and eight
	// words lift the ~88 atomics/us limit of a single word, which capped the kernel at 3.0 ms.
	const unsigned long long num_tiles = (p.n + 63ull) >> 6;
	uint32_t queue = blockIdx.x % RTK_QUEUES;
	uint32_t queues_left = RTK_QUEUES;
	const unsigned long long list_count = p.tile_list ? p.counter[RTK_LEFTOVER_COUNT_WORD] : 0ull;
	unsigned long long list_next = (unsigned long long)blockIdx.x * TRACE_WAVES_PER_BLOCK + wave;
	for (;;) {
		// ------------------------------------------------------------ next tile of 64 rays
		unsigned long long tile = 0;
		bool have = false;
		if (p.tile_list) {
			// the tiles the assembly kernel handed back: usually none or a handful, dealt by wave number (one atomic per wave
			// on one word, only to learn that the list is empty, made this launch take 0.14 ms: a word serves ~88 atomics/us)
			if (list_next < list_count) { tile = p.tile_list[list_next]; have = true; list_next += (unsigned long long)gridDim.x * TRACE_WAVES_PER_BLOCK; }
		} else
		while (queues_left) {
			unsigned long long got = 0;
			if (lane == 0) got = atomicAdd(p.counter + RTK_QUEUE_WORD(queue), 1ull);
			got = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32)) << 32) |
				(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got);
			// tile_blocks: a queue hands out the 64 tiles of one 64x64-pixel block one after the other (blocks are dealt
			// round robin over the queues); otherwise single tiles are dealt round robin
			tile = p.tile_blocks ? ((((got >> 6) * RTK_QUEUES + queue) << 6) | (got & 63ull)) : got * RTK_QUEUES + queue;
			if (tile < num_tiles) { have = true; break; }
			queue = (queue + 1u) % RTK_QUEUES;       // this queue is drained for good
			queues_left--;
		}
		if (!have) break;
		const unsigned long long base = tile << 6;
		const unsigned long long idx = base + lane;
		const bool alive = idx < p.n;
		const unsigned long long ray_index = map_index(alive ? idx : base, p.image_w, p.image_h, p.tile_blocks);

		PkLane L;
		bool special;
		float cnx, cfx, cny, cfy, cnz, cfz, rdx_, rdy_, rdz_, tmin_;
		{
			// (rays and hit records are streamed past the caches, "nt": read / written once, and the L2 is wanted for the BVH)
			const f32x4 r0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p.rays + ray_index)));
			const f32x4 r1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p.rays + ray_index) + 16));
			const float ox = r0.x, oy = r0.y, oz = r0.z;
			const float dx = r0.w, dy = r1.x, dz = r1.y;
			const float tmin = r1.z;
			L.tmax = r1.w;
			// rtk.c:550-566
			const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
			const float m = sse_max(sse_max(ax, ay), az);
			L.kz0 = ax == m;
			L.kz1 = !L.kz0 && ay == m;
			const float dkx = L.kz0 ? dy : (L.kz1 ? dz : dx);
			const float dky = L.kz0 ? dz : (L.kz1 ? dx : dy);
			const float dkz = L.kz0 ? dx : (L.kz1 ? dy : dz);
			L.shx = -dkx / dkz; L.shy = -dky / dkz; L.shz = 1.0f / dkz;
			L.sox = L.kz0 ? oy : (L.kz1 ? oz : ox);
			L.soy = L.kz0 ? oz : (L.kz1 ? ox : oy);
			L.soz = L.kz0 ? ox : (L.kz1 ? oy : oz);
			const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;           // rtk.c:410
			L.sx = __float_as_uint(dx) >> 31; L.sy = __float_as_uint(dy) >> 31; L.sz = __float_as_uint(dz) >> 31;
			L.t = L.tmax; L.u = 0.0f; L.v = 0.0f; L.prim = RTK_PRIM_NONE;
			// "special": the slab products can be NaN, or the ray is so far out or so extreme in direction that the margins of the fast
			// slab test (pk_slab2) stop being small against the inverted box of an empty child slot (+1 / -1 on every axis), which that
			// test must miss by arithmetic alone (it does not look at the child word: three scalar instructions per child and node
			// step). Everybody else (|origin| and the scene's planes < 2^19, 2^-100 < |1/d| < 2^100): an empty slot's near and far
			// parameter differ by 2 |1/d| against margins of 2^-21 |1/d| (|o| + B) < |1/d| / 2 each, so near > far on every axis.
			const bool tame = fabsf(ox) < 0x1p19f && fabsf(oy) < 0x1p19f && fabsf(oz) < 0x1p19f &&
				fabsf(rdx) > 0x1p-100f && fabsf(rdx) < 0x1p100f && fabsf(rdy) > 0x1p-100f && fabsf(rdy) < 0x1p100f &&
				fabsf(rdz) > 0x1p-100f && fabsf(rdz) < 0x1p100f;
			special = !(tame && bound_ok && tmin == tmin && L.tmax == L.tmax);
			// the fast slab test's constants: c = o * (1/d), the margin added for the row that bounds from below, subtracted for the other
			const float cx = ox * rdx, cy = oy * rdy, cz = oz * rdz;
			const float mx = 0x1p-21f * (fabsf(rdx) * (fabsf(ox) + bound_abs)), my = 0x1p-21f * (fabsf(rdy) * (fabsf(oy) + bound_abs)),
				mz = 0x1p-21f * (fabsf(rdz) * (fabsf(oz) + bound_abs));
			cnx = cx + mx; cfx = cx - mx; cny = cy + my; cfy = cy - my; cnz = cz + mz; cfz = cz - mz;
			rdx_ = rdx; rdy_ = rdy; rdz_ = rdz; tmin_ = tmin;
		}
		// wave-uniform facts about the packet
		const unsigned long long m_alive = __builtin_amdgcn_ballot_w64(alive);
		const bool wave_fast = __builtin_amdgcn_ballot_w64(alive && special) == 0ull;
		const unsigned long long bsx = __builtin_amdgcn_ballot_w64(alive && L.sx), bsy = __builtin_amdgcn_ballot_w64(alive && L.sy), bsz = __builtin_amdgcn_ballot_w64(alive && L.sz);
		const bool sign_uniform = (bsx == 0ull || bsx == m_alive) && (bsy == 0ull || bsy == m_alive) && (bsz == 0ull || bsz == m_alive);
		const bool usx = sign_uniform && bsx != 0ull, usy = sign_uniform && bsy != 0ull, usz = sign_uniform && bsz != 0ull;
		// byte offsets of the near / far plane rows inside a node (uniform signs), else (min, max)
		const uint32_t onx = usx ? 16u : 0u, ofx = 16u - onx;
		const uint32_t ony = 32u + (usy ? 16u : 0u), ofy = 80u - ony;
		const uint32_t onz = 64u + (usz ? 16u : 0u), ofz = 144u - onz;
		{
			// the row fetched first is this lane's near row, unless the signs differ inside the packet (rows come as (min, max)
			// then) and this lane's direction is negative on the axis
			const bool fx_ = !sign_uniform && L.sx, fy_ = !sign_uniform && L.sy, fz_ = !sign_uniform && L.sz;
			L.px = (f32x2){ rdx_, fx_ ? cfx : cnx };
			L.py = (f32x2){ rdy_, fy_ ? cfy : cny };
			L.pz = (f32x2){ rdz_, fz_ ? cfz : cnz };
			L.q1 = (f32x2){ fx_ ? cnx : cfx, fy_ ? cny : cfy };
			L.q2 = (f32x2){ fz_ ? cnz : cfz, tmin_ };
		}
		// child order: the node's front-to-back order for the packet's direction octant (the first ray's, if they differ: any
		// order gives the same hits). Word octant >> 1 of DevNode::order, half octant & 1.
		const uint32_t lead0 = (uint32_t)__builtin_ctzll(m_alive | (1ull << 63));
		const uint32_t octant = (uint32_t)((bsx >> lead0) & 1ull) | ((uint32_t)((bsy >> lead0) & 1ull) << 1) | ((uint32_t)((bsz >> lead0) & 1ull) << 2);
		const uint32_t oord = 0x70u + 4u * (octant >> 1), oshift = 16u * (octant & 1u);
		const unsigned long long bk0 = __builtin_amdgcn_ballot_w64(alive && L.kz0), bk1 = __builtin_amdgcn_ballot_w64(alive && L.kz1);
		const bool kz_uniform = (bk0 == 0ull || bk0 == m_alive) && (bk1 == 0ull || bk1 == m_alive);
		const uint32_t kzmode = !kz_uniform ? 3u : (bk0 != 0ull ? 0u : (bk1 != 0ull ? 1u : 2u));

		uint32_t c_nodes = 0, c_leaves = 0, c_tris = 0, c_spills = 0;
		uint32_t stack = 0;          // wave-uniform references, entry i in lane i
		uint32_t sp = 0;             // wave-uniform
		uint32_t top = 0;            // wave-uniform: root
		// the block's shared entry points (PkBlockEntries), if every ray of the tile lies inside the block's beam: the tile then
		// starts at the listed nodes, front to back, instead of at the root
		const PkBlockEntries *ent = p.entries;
		uint32_t ent_next = 0, ent_count = 0;      // ent_count != 0: the tile uses the list
		if (p.entries && p.tile_blocks && wave_fast && sign_uniform) {
			// (the tile number is wave-uniform -- one atomic or one list slot per wave --; say so, or the entry's node reference is
			// taken for a per-lane value and cannot address a scalar load)
			const uint32_t blk_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tile >> 6));
			const PkBlockEntries *eb = p.entries + blk_u;
			const float ox = L.sox, oy = L.soy, oz = L.soz;      // (permuted copies: undo by dominant axis)
			const float o3[3] = { L.kz0 ? oz : (L.kz1 ? oy : ox), L.kz0 ? ox : (L.kz1 ? oz : oy), L.kz0 ? oy : (L.kz1 ? ox : oz) };
			const float r3[3] = { rdx_, rdy_, rdz_ };
			bool in = tmin_ >= __uint_as_float(eb->pad[0]);
			for (int a = 0; a < 3; a++) in = in && o3[a] >= eb->olo[a] && o3[a] <= eb->ohi[a] && r3[a] >= eb->rlo[a] && r3[a] <= eb->rhi[a];
			const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)eb->count);      // (wave-uniform by construction; say so)
			if (__builtin_amdgcn_ballot_w64(alive && !in) == 0ull && m_alive == ~0ull && cnt != 0u) { ent = eb; ent_count = cnt; }
		}
		// which lanes take part in the node / leaf on top: kept as the wave's 64-bit mask (scalar registers) and turned into
		// the per-lane condition where the vector side needs it -- as a per-lane bool carried around the loop hipcc
		// rebuilt the mask from a 0/1 vector register at every vote (two vector instructions per ballot)
		unsigned long long live_m = m_alive;
		bool overflow = false;       // wave-uniform: a push did not fit (corrupted scene)
		bool done = false;
		if (ent_count != 0u) {
			// the first entry
			// (at the start nothing is culled: every lane's t is its max_t)
			top = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent->e[0].ref);
			ent_next = 1;
		}

		while (!done) {
			bool pop = false;
			if ((int32_t)top >= 0) {
				// ---------------------------------------------------- node (wave-uniform)
				const bool live = __builtin_amdgcn_inverse_ballot_w64(live_m);
				PkNode nd;
				s_load_node(nodes + (size_t)top * 128u, onx, ofx, ony, ofy, onz, ofz, oord, nd);
				if (COUNT && live) c_nodes++;
				if (COUNT && lane == 0) atomicAdd(p.counter + 7, 1ull);
				// per lane: entry distance of each child, NaN = this lane does not enter it
				float pay[4];
				if (sign_uniform) {
					if (wave_fast) { pk_slab2<true, true, 0>(L, nd, live, pay[0], pay[1]); pk_slab2<true, true, 2>(L, nd, live, pay[2], pay[3]); }
					else { pk_slab2<true, false, 0>(L, nd, live, pay[0], pay[1]); pk_slab2<true, false, 2>(L, nd, live, pay[2], pay[3]); }
				} else {
					if (wave_fast) { pk_slab2<false, true, 0>(L, nd, live, pay[0], pay[1]); pk_slab2<false, true, 2>(L, nd, live, pay[2], pay[3]); }
					else { pk_slab2<false, false, 0>(L, nd, live, pay[0], pay[1]); pk_slab2<false, false, 2>(L, nd, live, pay[2], pay[3]); }
				}
				// wave-level: which children does anybody enter
				const unsigned long long m0 = __builtin_amdgcn_ballot_w64(pay[0] == pay[0]), m1 = __builtin_amdgcn_ballot_w64(pay[1] == pay[1]),
					m2 = __builtin_amdgcn_ballot_w64(pay[2] == pay[2]), m3 = __builtin_amdgcn_ballot_w64(pay[3] == pay[3]);
				// (a 4-bit set + s_bcnt1 on the scalar unit; summing four booleans went through VGPRs and readfirstlane: +2.3 %)
				// bit c = somebody enters child c. Written out: compare, then shift the bit in with the carry (eight scalar
				// instructions; from the C expression hipcc made ~20, one of them a trip through a vector register)
				uint32_t any_mask;
				asm("s_cmp_lg_u64 %4, 0\n\ts_cselect_b32 %0, 1, 0\n\ts_cmp_lg_u64 %3, 0\n\ts_addc_u32 %0, %0, %0\n\t"
					"s_cmp_lg_u64 %2, 0\n\ts_addc_u32 %0, %0, %0\n\ts_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, %0"
					: "=&s"(any_mask) : "s"(m0), "s"(m1), "s"(m2), "s"(m3) : "scc");
				const bool a0 = (any_mask & 1u) != 0u, a1 = (any_mask & 2u) != 0u, a2 = (any_mask & 4u) != 0u;
				const uint32_t n_any = (uint32_t)__builtin_popcount(any_mask);
				uint32_t ref[4] = { (uint32_t)nd.ch[0], (uint32_t)nd.ch[1], (uint32_t)nd.ch[2], (uint32_t)nd.ch[3] };
				if (n_any == 0u) {
					pop = true;
				} else if (n_any == 1u) {
					// one child for the whole packet: a lane enters it iff it enters anything
					live_m = m0 | m1 | m2 | m3;
					top = a0 ? ref[0] : (a1 ? ref[1] : (a2 ? ref[2] : ref[3]));
				} else if (n_any == 2u) {
					// two children: pick them out (wave-uniform slot numbers) and take the node's own front-to-back order for this
					// packet's octant -- one bit test -- instead of comparing entry distances (two v_readlane, their keys, a compare).
					// Six possible pairs, one scalar branch each: plain register moves instead of selects of vector registers by
					// scalar slot numbers (each such select is a compare, a 64-bit mask and a v_cndmask)
					const uint32_t ow = (uint32_t)nd.ord >> oshift;
					float p0, p1;
					uint32_t r0, r1;
					bool swap;
#define PK_PAIR(i_, j_, bit_) p0 = pay[i_]; p1 = pay[j_]; r0 = ref[i_]; r1 = ref[j_]; swap = (ow & (1u << (RTK_ORDER_PAIR_SHIFT + bit_))) != 0u
					switch (any_mask) {
					case 3u: PK_PAIR(0, 1, 0); break;
					case 5u: PK_PAIR(0, 2, 1); break;
					case 9u: PK_PAIR(0, 3, 2); break;
					case 6u: PK_PAIR(1, 2, 3); break;
					case 10u: PK_PAIR(1, 3, 4); break;
					default: PK_PAIR(2, 3, 5); break;
					}
#undef PK_PAIR
					const float pfar = swap ? p0 : p1, pnear = swap ? p1 : p0;
					if (sp + 3u <= PK_LDS_STACK) PK_PUSH_LDS(pfar, swap ? r0 : r1);
					else PK_PUSH(pfar, swap ? r0 : r1);
					live_m = __builtin_amdgcn_ballot_w64(pnear == pnear);
					top = swap ? r1 : r0;
				} else {
					// three or four: walk the node's order for this octant from the far end; every child somebody enters is pushed,
					// the last one found (the nearest) is entered. Children by scalar slot number: one branch per slot, so that
					// payload and reference are fixed registers in each arm (no sort, no keys, no v_readlane).
					const uint32_t ow = (uint32_t)nd.ord >> oshift;
					const bool lds_ok = sp + 3u <= PK_LDS_STACK;
					uint32_t left = n_any;
#define PK_PLACE(c_) { left--; \
		if (left == 0u) { live_m = __builtin_amdgcn_ballot_w64(pay[c_] == pay[c_]); top = ref[c_]; } \
		else if (lds_ok) PK_PUSH_LDS(pay[c_], ref[c_]); else PK_PUSH(pay[c_], ref[c_]); }
#pragma unroll
					for (int q = 3; q >= 0; q--) {
						const uint32_t c = (ow >> (2 * q)) & 3u;
						if (!((any_mask >> c) & 1u)) continue;
						if (c == 0u) PK_PLACE(0) else if (c == 1u) PK_PLACE(1) else if (c == 2u) PK_PLACE(2) else PK_PLACE(3)
					}
#undef PK_PLACE
				}
			} else {
				// ---------------------------------------------------- leaf (wave-uniform)
				const uint32_t slot0 = top & 0x7fffffffu;
				const bool live = __builtin_amdgcn_inverse_ballot_w64(live_m);
				if (COUNT && live) c_leaves++;
				if (kzmode == 2u) pk_leaf<2, COUNT>(L, live, tris, slot0, lane, p.counter, c_tris);
				else if (kzmode == 0u) pk_leaf<0, COUNT>(L, live, tris, slot0, lane, p.counter, c_tris);
				else if (kzmode == 1u) pk_leaf<1, COUNT>(L, live, tris, slot0, lane, p.counter, c_tris);
				else pk_leaf<3, COUNT>(L, live, tris, slot0, lane, p.counter, c_tris);
				pop = true;
			}
			if (pop) {
				// pop until some lane still needs the entry (rtk.c:432, canonical: skip only if it starts BEHIND the hit)
				bool found = false;
				while (sp > PK_LDS_STACK) {                         // entries beyond the LDS part (deep trees only)
					sp--;
					const float te = spill_t[(size_t)(sp - PK_LDS_STACK) * p.spill_stride + glane];
					live_m = __builtin_amdgcn_ballot_w64(te <= L.t) & m_alive;
					if (live_m != 0ull) { found = true; break; }
				}
				if (!found) {
					// The LDS part, written out: ten instructions per entry (hipcc's version of the same loop took ~25, with the
					// stack pointer decremented on the vector unit and three flag masks per trip). Leaves with sp at the entry
					// that was taken and `hit` = 1, or with sp = 0 and `hit` = 0.
					uint32_t hit, off, addr;
					float te;
					asm volatile(
						"s_mov_b32 %[hit], 0\n"
						"1:\n\t"
						"s_cmp_eq_u32 %[sp], 0\n\t"
						"s_cbranch_scc1 2f\n\t"
						"s_sub_u32 %[sp], %[sp], 1\n\t"
						"s_lshl_b32 %[off], %[sp], 8\n\t"
						"v_add_u32_e32 %[addr], %[off], %[base]\n\t"
						"ds_read_b32 %[te], %[addr]\n\t"
						"s_waitcnt lgkmcnt(0)\n\t"
						"v_cmp_le_f32_e32 vcc, %[te], %[t]\n\t"
						"s_and_b64 %[live], vcc, %[alive]\n\t"
						"s_cbranch_scc0 1b\n\t"
						"s_mov_b32 %[hit], 1\n"
						"2:"
						: [sp] "+s"(sp), [hit] "=&s"(hit), [live] "=&s"(live_m), [off] "=&s"(off), [addr] "=&v"(addr), [te] "=&v"(te)
						: [base] "v"(lds_base), [t] "v"(L.t), [alive] "s"(m_alive)
						: "vcc", "scc", "memory");
					found = hit != 0u;
				}
				if (!found) {
					// the next entry of the block's list; they are sorted by the lower bound of the entry distance, so the first one
					// behind every lane's hit ends the tile
					bool took = false;
					if (ent_next < ent_count) {
						const float tlo = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(ent->e[ent_next].tlo)));
						live_m = __builtin_amdgcn_ballot_w64(tlo <= L.t) & m_alive;
						top = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent->e[ent_next].ref);
						ent_next++;
						took = live_m != 0ull;
					}
					if (!took) break;
					continue;
				}
				top = (uint32_t)__builtin_amdgcn_readlane((int)stack, (int)sp);
			}
		}

		if (overflow && lane == 0) p.counter[RTK_ERROR_WORD] = 1ull;
		if (alive) {
			if (p.occluded) p.occluded[ray_index] = L.prim != RTK_PRIM_NONE ? 1 : 0;       // an any-hit batch on an image: "the closest hit exists"
			else {
				f32x4 rec;
				rec.x = L.t; rec.y = L.u; rec.z = L.v; rec.w = __uint_as_float(L.prim);
				__builtin_nontemporal_store(rec, reinterpret_cast<f32x4 *>(p.hits + ray_index));
			}
			if (COUNT) {
				atomicAdd(p.counter + 1, 1ull);
				atomicAdd(p.counter + 2, (unsigned long long)c_nodes);
				atomicAdd(p.counter + 3, (unsigned long long)c_leaves);
				atomicAdd(p.counter + 4, (unsigned long long)c_tris);
				atomicAdd(p.counter + 5, L.prim != RTK_PRIM_NONE ? 1ull : 0ull);
				atomicAdd(p.counter + 6, (unsigned long long)c_spills);
			}
		}
	}
}


// ---- entry points shared by the 64 tiles of a 64x64-pixel block (PkBlockEntries, rtk_trace_shared.h) ---------------------
// One wave per block. The beam: nine of the block's rays (corners, edge midpoints, centre) give a box of origins, a box of reciprocal directions (the same
// IEEE quotient the tiles compute) and the smallest min_t; every tile checks its own rays against these before it uses the list
// (for a pinhole camera the boundary bounds the interior exactly; any other camera just fails the check and starts at the root).
// The interval slab test: per axis a lower bound of the entry parameter and an upper bound of the exit parameter over the
// corners of (origin box) x (reciprocal box) -- float subtraction and multiplication are monotone, so the bounds hold for
// what the reference computes for any ray of the beam, rtk.c:458-470 -- widened by the packet kernels' own margin, so that
// every child a tile's slab test admits is admitted here. The walk goes level by level until `target` nodes are on the list.
namespace {
struct PkBeam { float olo[3], ohi[3], rlo[3], rhi[3], m[3], tmin; uint32_t neg; };

__device__ __forceinline__ bool pk_beam_child(const DevNode &nd, int k, const PkBeam &b, float &tlo)
{
	const float lo[3] = { nd.bx[0][k], nd.by[0][k], nd.bz[0][k] }, hi[3] = { nd.bx[1][k], nd.by[1][k], nd.bz[1][k] };
	float n = b.tmin, f = INFINITY;
#pragma unroll
	for (int a = 0; a < 3; a++) {
		const bool neg = (b.neg >> a) & 1u;
		const float pn = neg ? hi[a] : lo[a], pf = neg ? lo[a] : hi[a];
		const float n0 = (pn - b.olo[a]) * b.rlo[a], n1 = (pn - b.olo[a]) * b.rhi[a], n2 = (pn - b.ohi[a]) * b.rlo[a], n3 = (pn - b.ohi[a]) * b.rhi[a];
		const float f0 = (pf - b.olo[a]) * b.rlo[a], f1 = (pf - b.olo[a]) * b.rhi[a], f2 = (pf - b.ohi[a]) * b.rlo[a], f3 = (pf - b.ohi[a]) * b.rhi[a];
		n = fmaxf(n, fminf(fminf(n0, n1), fminf(n2, n3)) - b.m[a]);
		f = fminf(f, fmaxf(fmaxf(f0, f1), fmaxf(f2, f3)) + b.m[a]);
	}
	tlo = n;
	return n <= f;
}

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

#define PK_FRONTIER 128
__global__ void __launch_bounds__(64) rtk_packet_entries_kernel(const DevNode *nodes, const rtk_ray *rays, uint32_t image_w, uint32_t blocks_per_row,
	float bound_abs, uint32_t target, uint32_t max_levels, PkBlockEntries *out, unsigned long long *counter)
{
	__shared__ uint32_t s_ref[2][PK_FRONTIER];
	// (the launch's queue heads and counters start from zero: cleared here, ahead of the traversal kernels, instead of by a fill kernel)
	if (blockIdx.x == 0) for (uint32_t w = threadIdx.x; w < (uint32_t)RTK_COUNTER_WORDS; w += 64u) counter[w] = 0ull;
	__shared__ float s_t[2][PK_FRONTIER];
	__shared__ uint32_t s_out_ref[PK_MAX_ENTRIES];
	__shared__ float s_out_t[PK_MAX_ENTRIES];
	const uint32_t lane = threadIdx.x, blk = blockIdx.x;
	const uint32_t bx = blk % blocks_per_row, by = blk / blocks_per_row;
	PkBeam b;
	bool ok = true;
	uint32_t s_and = 7u, s_or = 0u;
	float tmin = INFINITY;
	for (int a = 0; a < 3; a++) { b.olo[a] = INFINITY; b.ohi[a] = -INFINITY; b.rlo[a] = INFINITY; b.rhi[a] = -INFINITY; }
	if (lane < 9u) {
		// Nine rays of the block: corners, edge midpoints, centre -- three rows of the image (with one ray per lane along the whole
		// boundary, 64 rows and as many pages of a 512 MB ray buffer, the loads alone took ~10 us of this kernel's 30). For a pinhole
		// camera -- reciprocal directions monotone in the pixel -- the corners bound the block exactly; what any other camera puts
		// outside these bounds fails the tiles' own check against the beam and costs speed, never a hit.
		const uint32_t x = (lane % 3u) == 0u ? 0u : ((lane % 3u) == 1u ? 31u : 63u);
		const uint32_t y = (lane / 3u) == 0u ? 0u : ((lane / 3u) == 1u ? 31u : 63u);
		const rtk_ray r = rays[(size_t)(by * 64u + y) * image_w + bx * 64u + x];
		const float o[3] = { r.origin.x, r.origin.y, r.origin.z }, d[3] = { r.direction.x, r.direction.y, r.direction.z };
		uint32_t sg = 0;
		for (int a = 0; a < 3; a++) {
			const float rd = 1.0f / d[a];
			ok = ok && fabsf(o[a]) < 0x1p19f && fabsf(rd) > 0x1p-100f && fabsf(rd) < 0x1p100f;      // "tame", as the tiles test it
			b.olo[a] = o[a]; b.ohi[a] = o[a];
			b.rlo[a] = rd; b.rhi[a] = rd;
			sg |= (__float_as_uint(d[a]) >> 31) << a;
		}
		ok = ok && r.min_t == r.min_t && r.max_t == r.max_t;
		tmin = r.min_t;
		s_and = sg; s_or = sg;
	}
	for (int o = 32; o > 0; o >>= 1) { s_and &= __shfl_xor(s_and, o); s_or |= __shfl_xor(s_or, o); }
	ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull && s_and == s_or && bound_abs < 0x1p19f;
	for (int a = 0; a < 3; a++) {
		b.olo[a] = wave_min(b.olo[a]); b.ohi[a] = wave_max(b.ohi[a]); b.rlo[a] = wave_min(b.rlo[a]); b.rhi[a] = wave_max(b.rhi[a]);
		// two ulps outward: rtk_packet_beam2 checks its rays' v_rcp_f32 reciprocals (one ulp) against this box; the quotients here are
		// the correctly rounded ones, and a corner ray of the block must not fall out of its own block's beam by that ulp
		b.rlo[a] -= 0x1p-22f * fabsf(b.rlo[a]); b.rhi[a] += 0x1p-22f * fabsf(b.rhi[a]);
		b.m[a] = 0x1p-21f * (fmaxf(fabsf(b.rlo[a]), fabsf(b.rhi[a])) * (fmaxf(fabsf(b.olo[a]), fabsf(b.ohi[a])) + bound_abs));
	}
	b.tmin = wave_min(tmin);
	b.neg = s_or;
	PkBlockEntries *e = out + blk;
	if (lane == 0) {
		for (int a = 0; a < 3; a++) { e->olo[a] = b.olo[a]; e->ohi[a] = b.ohi[a]; e->rlo[a] = b.rlo[a]; e->rhi[a] = b.rhi[a]; }
		e->pad[0] = __float_as_uint(b.tmin); e->pad[1] = 0u; e->pad[2] = 0u;
		if (!ok) e->count = 0u;
	}
	if (!ok) return;
	// level by level from the root; wave-uniform counts, appends by ballot rank
	uint32_t n_cur = 1u, n_out = 0u, cur = 0u;
	bool over = false;
	if (lane == 0) { s_ref[0][0] = 0u; s_t[0][0] = b.tmin; }
	__syncthreads();
	// (a block that looks past the scene's edge finds few nodes per level and would walk to the leaves: the deepest walk is the
	// kernel's duration -- 30 us at 14 levels, 26 at 8 --, so the walk is capped)
	for (int level = 0; level < (int)max_levels && n_cur != 0u; level++) {
		if (level > 0 && n_out + n_cur >= target) break;
		uint32_t n_next = 0u;
		for (uint32_t base = 0; base < n_cur; base += 64u) {
			const bool have = base + lane < n_cur;
			const uint32_t ref = have ? s_ref[cur][base + lane] : 0u;
			const float t_self = have ? s_t[cur][base + lane] : 0.0f;
			const DevNode nd = nodes[ref];
			bool pass[4], leaf_below = false;
			float tlo[4];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				tlo[k] = 0.0f;
				pass[k] = have && nd.child[k] != RTK_REF_NONE && pk_beam_child(nd, k, b, tlo[k]);
				leaf_below = leaf_below || (pass[k] && (nd.child[k] & RTK_REF_LEAF) != 0u);
			}
			// a node with a leaf among the children the beam reaches is listed itself (a listed leaf would be tested by every
			// tile of the block); a node the beam reaches no child of is dropped
			const bool list_self = have && leaf_below;
			const unsigned long long below = (1ull << lane) - 1ull;
			const unsigned long long m_out = __builtin_amdgcn_ballot_w64(list_self);
			if (list_self) {
				const uint32_t at = n_out + (uint32_t)__popcll(m_out & below);
				if (at < PK_MAX_ENTRIES) { s_out_ref[at] = ref; s_out_t[at] = t_self; }
			}
			n_out += (uint32_t)__popcll(m_out);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const bool push = pass[k] && !list_self;
				const unsigned long long m_next = __builtin_amdgcn_ballot_w64(push);
				if (push) {
					const uint32_t at = n_next + (uint32_t)__popcll(m_next & below);
					if (at < PK_FRONTIER) { s_ref[cur ^ 1u][at] = nd.child[k]; s_t[cur ^ 1u][at] = tlo[k]; }
				}
				n_next += (uint32_t)__popcll(m_next);
			}
		}
		over = over || n_out > PK_MAX_ENTRIES || n_next > PK_FRONTIER;
		if (over) break;
		__syncthreads();
		cur ^= 1u;
		n_cur = n_next;
	}
	// the nodes that were not opened are entries too
	if (!over) {
		for (uint32_t i = lane; i < n_cur; i += 64u) if (n_out + i < PK_MAX_ENTRIES) { s_out_ref[n_out + i] = s_ref[cur][i]; s_out_t[n_out + i] = s_t[cur][i]; }
		n_out += n_cur;
		over = n_out > PK_MAX_ENTRIES;
	}
	__syncthreads();
	if (over) { if (lane == 0) e->count = 0u; return; }
	// front to back by the lower bound of the entry distance (rank = entries that come before; ties by position)
	if (lane < n_out) {
		const float t = s_out_t[lane];
		uint32_t rank = 0;
		for (uint32_t j = 0; j < n_out; j++) { const float tj = s_out_t[j]; rank += (tj < t || (tj == t && j < lane)) ? 1u : 0u; }
		e->e[rank].ref = s_out_ref[lane];
		e->e[rank].tlo = t;
	}
	if (lane == 0) e->count = n_out;
}
} // namespace

void rtk_packet_entries_launch(const TraceParams &p, PkBlockEntries *out, float bound_abs, unsigned target, unsigned max_levels, hipStream_t stream)
{
	const uint32_t bpr = p.image_w >> 6, rows = p.image_h >> 6;
	hipLaunchKernelGGL(rtk_packet_entries_kernel, dim3(bpr * rows), dim3(64), 0, stream, p.sc.nodes, p.rays, p.image_w, bpr, bound_abs, target, max_levels, out, p.counter);
}

// ---- the hand-written kernel: a code object of its own (rtk_packet_hot.S, assembled by the Makefile), carried in this
// library as a byte array and loaded once per device
#include "rtk_packet_hot_image.h"
#include <mutex>

namespace {
struct HotModule { hipModule_t mod = nullptr; hipFunction_t fn = nullptr, fn_beam = nullptr, fn_beam2 = nullptr, fn_count2 = nullptr, fn_any2 = nullptr; int blocks_per_cu = 0, beam_blocks_per_cu = 0, beam2_blocks_per_cu = 0; bool tried = false; };
std::mutex g_hot_mutex;
HotModule g_hot[RTK_MAX_DEVICES];

HotModule *hot_module(int device)
{
	if (device < 0 || device >= RTK_MAX_DEVICES) return nullptr;
	std::lock_guard<std::mutex> lock(g_hot_mutex);
	HotModule &h = g_hot[device];
	if (!h.tried) {
		int cur = -1;
		if (hipGetDevice(&cur) != hipSuccess || cur != device) return nullptr;      // loaded by a thread that has this device current (asked again later)
		h.tried = true;
		if (hipModuleLoadData(&h.mod, rtk_packet_hot_image) != hipSuccess || hipModuleGetFunction(&h.fn, h.mod, "rtk_packet_hot") != hipSuccess) {
			(void)hipGetLastError();
			h.fn = nullptr;
		} else {
			// 64 VGPRs, 96 SGPRs, 20 KB of LDS per workgroup (20 stack entries per lane): seven waves per SIMD (the occupancy query reports one more for
			// kernels at this SGPR count on ROCm 7.2; 800 / (96 + 16) = 7)
			int nb = 0;
			if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, h.fn, TRACE_BLOCK_THREADS, 0) != hipSuccess || nb < 1) nb = 1;
			h.blocks_per_cu = nb > 7 ? 7 : nb;
			// rtk_packet_beam (the same file assembled with -DRTK_BEAM): 64 VGPRs, 94 SGPRs, no LDS: eight waves per SIMD
			if (hipModuleGetFunction(&h.fn_beam, h.mod, "rtk_packet_beam") != hipSuccess) { (void)hipGetLastError(); h.fn_beam = nullptr; }
			else {
				nb = 0;
				if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, h.fn_beam, TRACE_BLOCK_THREADS, 0) != hipSuccess || nb < 1) nb = 1;
				h.beam_blocks_per_cu = nb > 8 ? 8 : nb;
			}
			// rtk_packet_beam2 (rtk_packet_beam2.S: two adjacent tiles per wave): 72 VGPRs: seven waves per SIMD
			if (hipModuleGetFunction(&h.fn_beam2, h.mod, "rtk_packet_beam2") != hipSuccess) { (void)hipGetLastError(); h.fn_beam2 = nullptr; }
			else {
				nb = 0;
				if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, h.fn_beam2, TRACE_BLOCK_THREADS, 0) != hipSuccess || nb < 1) nb = 1;
				h.beam2_blocks_per_cu = nb > 7 ? 7 : nb;
			}
			// rtk_packet_count2: rtk_packet_beam2.S assembled with -DRTK_COUNT (the counting form of the kernel that is timed)
			if (hipModuleGetFunction(&h.fn_count2, h.mod, "rtk_packet_count2") != hipSuccess) { (void)hipGetLastError(); h.fn_count2 = nullptr; }
			// rtk_packet_any2: ... with -DRTK_ANY (one flag per ray, a ray retired at its first hit)
			if (hipModuleGetFunction(&h.fn_any2, h.mod, "rtk_packet_any2") != hipSuccess) { (void)hipGetLastError(); h.fn_any2 = nullptr; }
		}
	}
	return h.fn ? &h : nullptr;
}
} // namespace

bool rtk_packet_hot_available(int device, int *blocks_per_cu, int beam)
{
	HotModule *h = hot_module(device);
	if (!h || (beam == 1 && !h->fn_beam) || (beam == 2 && !h->fn_beam2) || (beam == 3 && !h->fn_count2) || (beam == 4 && !h->fn_any2)) return false;
	if (blocks_per_cu) *blocks_per_cu = beam >= 2 ? h->beam2_blocks_per_cu : beam == 1 ? h->beam_blocks_per_cu : h->blocks_per_cu;
	return true;
}

int rtk_packet_hot_launch(int device, const PkHotParams &hp_in, unsigned blocks, hipStream_t stream, int beam)
{
	HotModule *h = hot_module(device);
	if (!h || (beam == 1 && !h->fn_beam) || (beam == 2 && !h->fn_beam2) || (beam == 3 && !h->fn_count2) || (beam == 4 && !h->fn_any2)) { rtk_set_error("rtk_dev_trace: the assembly packet kernel is not loaded"); return RTK_AMD_ERR_HIP; }
	PkHotParams hp = hp_in;
	size_t size = sizeof(hp);
	void *config[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &hp, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
	RTK_HIP_CHECK(hipModuleLaunchKernel(beam == 4 ? h->fn_any2 : beam == 3 ? h->fn_count2 : beam == 2 ? h->fn_beam2 : beam == 1 ? h->fn_beam : h->fn, blocks, 1, 1, TRACE_BLOCK_THREADS, 1, 1, 0, stream, nullptr, config), RTK_AMD_ERR_HIP);
	return RTK_AMD_OK;
}

int rtk_packet_occupancy(bool counted)
{
	int nb = 0;
	hipError_t e = counted ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk_trace_packet_kernel<true>, TRACE_BLOCK_THREADS, 0)
	                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk_trace_packet_kernel<false>, TRACE_BLOCK_THREADS, 0);
	return (e == hipSuccess && nb >= 1) ? nb : 1;
}

void rtk_packet_launch(const TraceParams &p, unsigned blocks, hipStream_t stream, bool counted)
{
	if (counted) hipLaunchKernelGGL((rtk_trace_packet_kernel<true>), dim3(blocks), dim3(TRACE_BLOCK_THREADS), 0, stream, p);
	else hipLaunchKernelGGL((rtk_trace_packet_kernel<false>), dim3(blocks), dim3(TRACE_BLOCK_THREADS), 0, stream, p);
}
