// rtk_trace_pool.hip -- the per-lane traversal with its rays in an LDS pool (closest hit and any hit, rtk.c:543-577 per ray).
//
// rtk_trace.hip binds a ray to a lane until the ray is finished. Its waves spend their time half empty: a lane that
// reached a leaf waits while the others descend, the triangle step runs for the lanes that happen to be at a leaf (30 % of
// them on incoherent rays), new rays are set up a few at a time. Counters and the wave-level simulation in
// scripts/bvh_lab.cpp (-ws 0 / -ws 3) agree on 47 % lane use, with the vector unit 95 % busy: the kernel is bound by
// instructions that half of the lanes do not need.
//
// Here a ray belongs to no lane. One 1024-thread workgroup owns a CU: the state of POOL_SLOTS rays (72 B) and the top
// POOL_STACK entries of their traversal stacks live in its 160 KB of LDS, and three queues in LDS name the rays that want
// a node step, a leaf, or a new ray. Every trip a wave claims up to 64 rays from ONE queue, loads their state, does that
// one kind of step for all of them, stores what changed and hands every ray to the queue of its next state. Node steps
// and triangle steps run with (nearly) all lanes; the price is the state traffic through LDS and the queue bookkeeping
// (~45 instructions per trip). No barrier after start-up: the queues are multi-producer / multi-consumer rings
// under LDS atomics, every wait is bounded (a bound that is hit sets the launch's error word and ends the kernel).
//
// Results are those of rtk_trace_kernel bit for bit: the node step is its compressed-node fast path, the leaf loop is its
// leaf loop (group-of-four double precision rule, canonical ties), and the order in which a ray's nodes are visited does
// not enter the result. Rays that kernel treats specially (non-finite or zero components) and batches it counts, filters
// or collects for are not taken here: special rays are appended to a list that rtk_trace_kernel traces afterwards.
#include "rtk_dev.h"

#include "rtk_trace_lane.h"

#include <math.h>
#include <stdlib.h>

#include <mutex>

#define POOL_WAVES 16
#define POOL_THREADS (64 * POOL_WAVES)
#define POOL_SLOTS 1088u          // 17 waves' worth: 16 in flight + 64 rays of slack, what 160 KB holds at 136 B per ray
#define POOL_STACK 8u             // stack entries per ray in LDS; deeper ones in the launch's spill area
#define POOL_RING 2048u           // entries per queue ring (a power of two above POOL_SLOTS: a ring never fills)
#define POOL_EMPTY 0xffffu
#define POOL_SPIN_LIMIT (1u << 22)

enum { Q_NODE = 0, Q_LEAF = 1, Q_FREE = 2, KIND_DONE = 4 };
enum { C_LIVE = 6, C_EXHAUSTED = 7, C_ABORT = 8 };         // ctl words 0..5: head, tail of the three queues; rays that are set up and not finished

struct PoolLds {
	float4 f0[POOL_SLOTS];                    // origin, min_t
	float4 f1[POOL_SLOTS];                    // 1 / direction, best t
	uint4 f2[POOL_SLOTS];                     // top (node / leaf reference, RTK_REF_RETRY, RTK_REF_NONE), stack size | dominant axis << 16, best primitive, ray number
	float4 f3[POOL_SLOTS];                    // shear constants (rtk.c:550-566), best u
	float2 f4[POOL_SLOTS];                    // best v, max_t
	uint2 stack[POOL_STACK][POOL_SLOTS];      // (entry distance, reference)
	uint16_t ring[3][POOL_RING];
	uint32_t ctl[16];
};
static_assert(sizeof(PoolLds) <= 163840, "the pool must fit one CU's LDS");

#define RTK_REF_RETRY 0xfffffffeu

__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the six queue words, the live count and the exhausted flag in one go (eight ds_read_b32 behind one wait; a volatile vector read
// through the struct reference came out as flat loads, which wait on the vector-memory counter too)
struct PoolSnap { uint32_t nn, nl, nf, live, ex; uint32_t h0, h1, h2; };
__device__ __forceinline__ uint32_t first_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ PoolSnap pool_snapshot(const uint32_t *ctl)
{
	const uint32_t h0 = lds_load(ctl + 0), t0 = lds_load(ctl + 1), h1 = lds_load(ctl + 2), t1 = lds_load(ctl + 3), h2 = lds_load(ctl + 4), t2 = lds_load(ctl + 5);
	const uint32_t live = lds_load(ctl + 6), ex = lds_load(ctl + 7);
	PoolSnap s;
	s.h0 = h0; s.h1 = h1; s.h2 = h2; s.nn = t0 - h0; s.nl = t1 - h1; s.nf = ex ? 0u : t2 - h2; s.live = live; s.ex = ex;
	return s;
}
__device__ __forceinline__ uint32_t lane_rank(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// Ring entries carry the lap of their position (entry = slot | lap << 11, lap = position / POOL_RING mod 32): a reader knows
// when the entry it was promised has been written without anybody ever clearing entries, and a writer never has to look before
// it writes (the ring has almost twice as many positions as there are rays, so a position's last reader is long gone; should
// timing ever break that, the reader finds a lap it does not expect, runs into its spin limit and the launch is flagged).
__device__ __forceinline__ uint32_t ring_entry(uint32_t slot, uint32_t pos) { return slot | (((pos / POOL_RING) & 31u) << 11); }

// Hand every ray to the queue of its next state: ONE LDS atomic instruction reserves the positions in all three rings (lane q
// adds for queue q), then every lane writes its entry. q_of: 0..2, or 3 for lanes that hand nothing on.
__device__ __forceinline__ void pool_push_all(PoolLds &L, uint32_t q_of, uint32_t slot, uint32_t lane)
{
	const unsigned long long m0 = __builtin_amdgcn_ballot_w64(q_of == 0u), m1 = __builtin_amdgcn_ballot_w64(q_of == 1u), m2 = __builtin_amdgcn_ballot_w64(q_of == 2u);
	const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2);
	uint32_t base = 0;
	const uint32_t mine = lane == 0u ? n0 : (lane == 1u ? n1 : n2);
	if (lane < 3u && mine != 0u) base = __hip_atomic_fetch_add(&L.ctl[2u * lane + 1u], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)base, 0), b1 = (uint32_t)__builtin_amdgcn_readlane((int)base, 1), b2 = (uint32_t)__builtin_amdgcn_readlane((int)base, 2);
	if (q_of < 3u) {
		const unsigned long long m = q_of == 0u ? m0 : (q_of == 1u ? m1 : m2);
		const uint32_t pos = (q_of == 0u ? b0 : (q_of == 1u ? b1 : b2)) + lane_rank(m);
		__hip_atomic_store(&L.ring[q_of][pos & (POOL_RING - 1u)], (uint16_t)ring_entry(slot, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

template <int MODE /*0 closest hit, 1 any hit*/>
__global__ void __launch_bounds__(POOL_THREADS, 1) rtk_trace_pool_kernel(TraceParams p)
{
	extern __shared__ __align__(16) unsigned char pool_lds_raw[];
	PoolLds &L = *reinterpret_cast<PoolLds *>(pool_lds_raw);
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const char *const qnodes = reinterpret_cast<const char *>(p.sc.qnodes);
	const char *const tris = reinterpret_cast<const char *>(p.sc.tris);
	uint2 *const spill = p.spill + (size_t)blockIdx.x * POOL_SLOTS * p.spill_cap;     // this workgroup's part: [entry - POOL_STACK][slot]

	// ---- start: every slot is free
	for (uint32_t i = threadIdx.x; i < 3u * POOL_RING; i += POOL_THREADS) (&L.ring[0][0])[i] = (uint16_t)POOL_EMPTY;
	if (threadIdx.x < 16u) L.ctl[threadIdx.x] = 0u;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < POOL_SLOTS; i += POOL_THREADS) L.ring[Q_FREE][i] = (uint16_t)i;
	if (threadIdx.x == 0u) L.ctl[2 * Q_FREE + 1] = POOL_SLOTS;
	__syncthreads();

	// chunks of 64 rays are dealt through RTK_QUEUES queue heads, as in rtk_trace_kernel (second word of each head's line)
	uint32_t queue = (blockIdx.x * POOL_WAVES + wave) % RTK_QUEUES, queues_left = RTK_QUEUES;
	const unsigned long long num_chunks = (p.n + 63ull) >> 6;

#ifdef POOL_STATS
	unsigned long long st_trips[3] = { 0, 0, 0 }, st_rays[3] = { 0, 0, 0 }, st_spins = 0, st_steps = 0, st_clk[4] = { 0, 0, 0, 0 }, st_cas = 0, st_seg[4] = { 0, 0, 0, 0 };
#define POOL_STAT(x) x
#else
#define POOL_STAT(x)
#endif
	for (;;) {
		// ------------------------------------------------------------ claim up to 64 rays that want the same kind of step
		uint32_t kind = KIND_DONE, k = 0, h = 0;
		POOL_STAT(const unsigned long long c_t0 = __builtin_readcyclecounter();)
		if (lane == 0u) {
			for (uint32_t spin = 0;; spin++) {
				// one look at all the queue words (a snapshot that is not atomic as a whole: the compare-and-swap below decides)
				const PoolSnap sn = pool_snapshot(L.ctl);
				const uint32_t h0 = sn.h0, h1 = sn.h1, h2 = sn.h2, live = sn.live, ex = sn.ex, nn = sn.nn, nl = sn.nl, nf = sn.nf;
				int q = -1;
				if (nl >= 64u) q = Q_LEAF;                    // leaves first: they are what shortens the rays
				else if (nn >= 64u) q = Q_NODE;
				else if (nf >= 64u) q = Q_FREE;
				else if (nn >= 32u && spin != 0u) q = Q_NODE;                         // half a wave of node steps beats waiting on
				else if ((nn | nl) == 0u) { if (ex && live == 0u) break; }           // nothing queued, no ray unfinished, no rays left: done
				else if (nn + nl >= live || spin >= 48u) q = nn >= nl ? Q_NODE : Q_LEAF;   // nobody holds a ray that could top a queue up (or has for a while)
				if (q >= 0) {
					const uint32_t hq = q == Q_NODE ? h0 : q == Q_LEAF ? h1 : h2, avail = q == Q_NODE ? nn : q == Q_LEAF ? nl : nf;
					const uint32_t kk = avail < 64u ? avail : 64u;
					uint32_t expect = hq;
					if (__hip_atomic_compare_exchange_strong(&L.ctl[2 * q], &expect, hq + kk, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
						kind = (uint32_t)q; k = kk; h = hq;
						break;
					}
					POOL_STAT(st_cas++;)
					__builtin_amdgcn_s_sleep(1);                // another wave was faster
					continue;
				}
				if ((spin & 15u) == 15u && lds_load(&L.ctl[C_ABORT]) != 0u) break;
				if (spin >= POOL_SPIN_LIMIT) {
					__hip_atomic_store(&L.ctl[C_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					p.counter[RTK_ERROR_WORD] = 2ull;
					break;
				}
				__builtin_amdgcn_s_sleep(2);
				POOL_STAT(st_spins++;)
			}
		}
		kind = first_lane(kind); k = first_lane(k); h = first_lane(h);
		if (kind == KIND_DONE) break;
		POOL_STAT(const unsigned long long c_t1 = __builtin_readcyclecounter(); st_trips[kind]++; st_rays[kind] += k; st_clk[3] += c_t1 - c_t0;)
		const bool act = lane < k;
		uint32_t slot = 0;
		bool ok = true;
		if (act) {
			const uint32_t pos = h + lane, lap = (pos / POOL_RING) & 31u;
			const uint16_t *e = &L.ring[kind][pos & (POOL_RING - 1u)];
			uint32_t spin = 0, v;
			while (((v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 11) != lap && spin < POOL_SPIN_LIMIT) { spin++; __builtin_amdgcn_s_sleep(1); }
			ok = spin < POOL_SPIN_LIMIT;
			slot = ok ? (v & 2047u) : 0u;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // the ray's state was written before its slot number
		POOL_STAT(const unsigned long long c_t2 = __builtin_readcyclecounter(); if (kind == Q_NODE) st_seg[0] += c_t2 - c_t1;)

		bool to_node = false, to_leaf = false, to_free = false;
		if (kind == Q_NODE) {
			// -------------------------------------------------------- node step (rtk.c:457-517 on the compressed node)
			uint4 s2 = make_uint4(RTK_REF_NONE, 0u, 0u, 0u);
			float4 s0 = make_float4(0, 0, 0, 0), s1 = make_float4(0, 0, 0, 0);
			if (act) { s2 = L.f2[slot]; s1 = L.f1[slot]; s0 = L.f0[slot]; }
			uint32_t top = s2.x, sp = s2.y & 0xffffu;
			const float best_t = s1.w, tmin_ray = s0.w;
			// a ray that comes from a leaf, from a node it missed entirely or from a culled entry pops here: ONE site for the
			// stack read (LDS or spill), and the node it yields is fetched in this same trip
			if (__builtin_amdgcn_ballot_w64(act && top == RTK_REF_RETRY) != 0ull) {
				if (act && top == RTK_REF_RETRY) {
					if (sp == 0u) top = RTK_REF_NONE;
					else {
						--sp;
						uint2 e = L.stack[sp < POOL_STACK ? sp : POOL_STACK - 1u][slot];
						if (sp >= POOL_STACK) {
							const unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(spill + (size_t)(sp - POOL_STACK) * POOL_SLOTS + slot));
							e = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
						}
						top = __uint_as_float(e.x) > best_t ? RTK_REF_RETRY : e.y;
					}
				}
			}
			POOL_STAT(st_steps += __popcll(__builtin_amdgcn_ballot_w64(act && (int32_t)top >= 0)); const unsigned long long c_t3 = __builtin_readcyclecounter(); st_seg[1] += c_t3 - c_t2;)
			if (act && (int32_t)top >= 0) {
				const float ox = s0.x, oy = s0.y, oz = s0.z, rdx = s1.x, rdy = s1.y, rdz = s1.z;
				f32x4 l0;
				u32x4 l1, l2, l3;
				load_qnode(qnodes, top << 6, l0, l1, l2, l3);
				// (the arithmetic of rtk_trace_kernel's compressed-node step, see the comments there)
				const float Ax = (l0.x - ox) * rdx, Ay = (l0.y - oy) * rdy, Az = (l0.z - oz) * rdz;
				const float Sx = l0.w * rdx, Sy = __uint_as_float(l1.x) * rdy, Sz = __uint_as_float(l1.y) * rdz;
				const float ex = 0x1p-21f * __builtin_fmaf(fabsf(Sx), 255.0f, fabsf(Ax));
				const float ey = 0x1p-21f * __builtin_fmaf(fabsf(Sy), 255.0f, fabsf(Ay));
				const float ez = 0x1p-21f * __builtin_fmaf(fabsf(Sz), 255.0f, fabsf(Az));
				const float Anx = Ax - ex, Afx = Ax + ex, Any = Ay - ey, Afy = Ay + ey, Anz = Az - ez, Afz = Az + ez;
				const bool ngx = (__float_as_uint(rdx) >> 31) != 0u, ngy = (__float_as_uint(rdy) >> 31) != 0u, ngz = (__float_as_uint(rdz) >> 31) != 0u;
				const uint32_t wnx = ngx ? l1.w : l1.z, wfx = ngx ? l1.z : l1.w;
				const uint32_t wny = ngy ? l2.y : l2.x, wfy = ngy ? l2.x : l2.y;
				const uint32_t wnz = ngz ? l2.w : l2.z, wfz = ngz ? l2.z : l2.w;
				uint32_t ref[4] = { l3.x, l3.y, l3.z, l3.w };
				float key[4];
				uint32_t nhit = 0;
#pragma unroll
				for (int i = 0; i < 4; i++) {
					const f32x2 px = __builtin_elementwise_fma((f32x2){ ubyte_f32(wnx, i), ubyte_f32(wfx, i) }, (f32x2){ Sx, Sx }, (f32x2){ Anx, Afx });
					const f32x2 py = __builtin_elementwise_fma((f32x2){ ubyte_f32(wny, i), ubyte_f32(wfy, i) }, (f32x2){ Sy, Sy }, (f32x2){ Any, Afy });
					const f32x2 pz = __builtin_elementwise_fma((f32x2){ ubyte_f32(wnz, i), ubyte_f32(wfz, i) }, (f32x2){ Sz, Sz }, (f32x2){ Anz, Afz });
					const float tn = fmaxf(fmaxf(fmaxf(px.x, py.x), pz.x), tmin_ray);
					const float tf = fminf(fminf(fminf(px.y, py.y), pz.y), best_t);
					const bool hit = tn <= tf && ref[i] != RTK_REF_NONE;
					key[i] = hit ? tn : __builtin_inff();
					nhit += hit ? 1u : 0u;
				}
				// nearest first (rtk.c:496-517 orders by entry distance)
				cswap(key[0], ref[0], key[1], ref[1]);
				cswap(key[2], ref[2], key[3], ref[3]);
				cswap(key[0], ref[0], key[2], ref[2]);
				cswap(key[1], ref[1], key[3], ref[3]);
				cswap(key[1], ref[1], key[2], ref[2]);
				if (nhit == 0u) top = RTK_REF_RETRY;
				else {
					top = ref[0];
					const uint32_t np = nhit - 1u;              // sorted slots np..1 go on the stack, far to near
#pragma unroll
					for (int i = 1; i <= 3; i++) {
						if ((uint32_t)i <= np) {
							const uint32_t row = sp + np - (uint32_t)i;
							const uint2 e = make_uint2(__float_as_uint(key[i]), ref[i]);
							if (row < POOL_STACK) L.stack[row][slot] = e;
							else if (row - POOL_STACK < p.spill_cap) spill[(size_t)(row - POOL_STACK) * POOL_SLOTS + slot] = e;
							else p.counter[RTK_ERROR_WORD] = 1ull;         // cannot happen for a tree (rtk_upload.hip rejects anything else)
						}
					}
					sp += np;
				}
			}
			POOL_STAT(const unsigned long long c_t4 = __builtin_readcyclecounter(); st_seg[2] += c_t4 - c_t3;)
			if (act) {
				if (top == RTK_REF_NONE) {
					// the stack ran empty: the ray is finished
					const uint32_t ray_index = s2.w;
					if (MODE == 1) p.occluded[ray_index] = 0;
					else st_f4_stream(p.hits + ray_index, best_t, L.f3[slot].w, L.f4[slot].x, __uint_as_float(s2.z));
					to_free = true;
				} else {
					*reinterpret_cast<uint2 *>(&L.f2[slot]) = make_uint2(top, (s2.y & 0xffff0000u) | sp);
					to_leaf = (int32_t)top < 0 && top != RTK_REF_RETRY;
					to_node = !to_leaf;
				}
			}
		} else if (kind == Q_LEAF) {
			// -------------------------------------------------------- the triangles of one leaf (rtk.c:212-386), as rtk_trace_kernel takes them
			uint4 s2 = make_uint4(0u, 0u, RTK_PRIM_NONE, 0u);
			float4 s0 = make_float4(0, 0, 0, 0), s3 = make_float4(0, 0, 0, 0);
			float2 s4 = make_float2(0, 0);
			float best_t = 0;
			if (act) { s2 = L.f2[slot]; s0 = L.f0[slot]; s3 = L.f3[slot]; s4 = L.f4[slot]; best_t = L.f1[slot].w; }
			const bool kz0 = (s2.y & 0x10000u) != 0u, kz1 = (s2.y & 0x20000u) != 0u;
			const float sox = kz0 ? s0.y : (kz1 ? s0.z : s0.x), soy = kz0 ? s0.z : (kz1 ? s0.x : s0.y), soz = kz0 ? s0.x : (kz1 ? s0.y : s0.z);
			const float shx = s3.x, shy = s3.y, shz = s3.z, tmin_ray = s0.w, tmax_ray = s4.y;
			float best_u = s3.w, best_v = s4.x;
			uint32_t best_prim = s2.z;
			if (act) {
				const uint32_t slot0 = s2.x & 0x7fffffffu;
				uint32_t i = 0, n = 1;
				bool force = false, redo = false;
				float sn_t = best_t, sn_u = best_u, sn_v = best_v;
				uint32_t sn_prim = best_prim;
				while (i < n) {
					f32x4 A, B, C;
					load_tri(tris, (slot0 + i) * (uint32_t)RTK_TRI_STRIDE, A, B, C);
					if (i == 0u) n = __float_as_uint(C.w);          // leaf size rides in the first record
					if ((i & 3u) == 0u) {
						if (redo) { force = true; redo = false; }
						else {
							if (MODE == 1 && best_prim != RTK_PRIM_NONE) break;   // any-hit: a whole group accepted something
							force = (n - i) < 4u;
							sn_t = best_t; sn_u = best_u; sn_v = best_v; sn_prim = best_prim;
						}
					}
					// permute to (kx,ky,kz) and move the origin (rtk.c:232-280)
					const float v0x = (kz0 ? A.y : (kz1 ? A.z : A.x)) - sox;
					const float v0y = (kz0 ? A.z : (kz1 ? A.x : A.y)) - soy;
					const float v0z = (kz0 ? A.x : (kz1 ? A.y : A.z)) - soz;
					const float v1x = (kz0 ? B.y : (kz1 ? B.z : B.x)) - sox;
					const float v1y = (kz0 ? B.z : (kz1 ? B.x : B.y)) - soy;
					const float v1z = (kz0 ? B.x : (kz1 ? B.y : B.z)) - soz;
					const float v2x = (kz0 ? C.y : (kz1 ? C.z : C.x)) - sox;
					const float v2y = (kz0 ? C.z : (kz1 ? C.x : C.y)) - soy;
					const float v2z = (kz0 ? C.x : (kz1 ? C.y : C.z)) - soz;
					// shear (rtk.c:284-292)
					const float x0 = v0x + shx * v0z, y0 = v0y + shy * v0z, z0 = shz * v0z;
					const float x1 = v1x + shx * v1z, y1 = v1y + shy * v1z, z1 = shz * v1z;
					const float x2 = v2x + shx * v2z, y2 = v2y + shy * v2z, z2 = shz * v2z;
					// edge functions (rtk.c:298-300)
					float u, v, w;
					if (!force) {
						u = x1 * y2 - y1 * x2;
						v = x2 * y0 - y2 * x0;
						w = x0 * y1 - y0 * x1;
						if (u == 0.0f || v == 0.0f || w == 0.0f) {
							// rtk.c:306: the whole group switches to double precision
							best_t = sn_t; best_u = sn_u; best_v = sn_v; best_prim = sn_prim;
							redo = true;
							i &= ~3u;
							continue;
						}
					} else {
						const double xd0 = x0, yd0 = y0, xd1 = x1, yd1 = y1, xd2 = x2, yd2 = y2;
						u = (float)(xd1 * yd2 - yd1 * xd2);
						v = (float)(xd2 * yd0 - yd2 * xd0);
						w = (float)(xd0 * yd1 - yd0 * xd1);
					}
					// rtk.c:340-342
					const bool neg = sse_min(sse_min(u, v), w) < 0.0f;
					const bool pos = sse_max(sse_max(u, v), w) > 0.0f;
					// rtk.c:346-353
					const float det = (u + v) + w;
					const float rcp = 1.0f / det;
					float zz = u * z0;
					zz = zz + v * z1;
					zz = zz + w * z2;
					const float t = zz * rcp;
					const uint32_t prim = __float_as_uint(A.w);
					const bool in_range = !(neg && pos) && t > tmin_ray && t < tmax_ray;   // rtk.c:354
					if (MODE == 1) {
						if (in_range && best_prim == RTK_PRIM_NONE) { best_prim = prim; best_t = t; }
					} else {
						// rtk.c:371 with the canonical tie rule: lowest primitive id among bit-equal t
						if (in_range && (t < best_t || (t == best_t && prim < best_prim))) {
							best_t = t; best_u = u * rcp; best_v = v * rcp; best_prim = prim;
						}
					}
					i++;
				}
				if (MODE == 1 && best_prim != RTK_PRIM_NONE) {
					p.occluded[s2.w] = 1;
					to_free = true;
				} else {
					L.f1[slot].w = best_t;
					L.f3[slot].w = best_u;
					L.f4[slot].x = best_v;
					L.f2[slot] = make_uint4(RTK_REF_RETRY, s2.y, best_prim, s2.w);      // the next node trip pops
					to_node = true;
				}
			}
		} else {
			// -------------------------------------------------------- new rays into free slots (rtk.c:550-566)
			unsigned long long chunk = ~0ull;
			if (lane == 0u) {
				while (queues_left) {
					const unsigned long long got = atomicAdd(p.counter + RTK_QUEUE_WORD(queue) + 8, 1ull);
					const unsigned long long c = got * RTK_QUEUES + queue;
					if (c < num_chunks) { chunk = c; break; }
					queue = (queue + 1u) % RTK_QUEUES;
					queues_left--;
				}
			}
			queue = first_lane(queue); queues_left = first_lane(queues_left);
			chunk = ((unsigned long long)first_lane((uint32_t)(chunk >> 32)) << 32) | first_lane((uint32_t)chunk);
			if (chunk == ~0ull) {
				// no rays left: the slots just claimed are dropped, and nobody asks for free slots any more
				if (lane == 0u) __hip_atomic_store(&L.ctl[C_EXHAUSTED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			} else {
				const unsigned long long i = (chunk << 6) + lane;
				const bool valid = act && i < p.n;
				if (valid) {
					const uint32_t ray_index = p.perm ? (uint32_t)p.perm[i] : (uint32_t)i;      // perm: sort words, ray number in the low half
					const float4 r0 = ld_f4_stream(reinterpret_cast<const char *>(p.rays + ray_index));
					const float4 r1 = ld_f4_stream(reinterpret_cast<const char *>(p.rays + ray_index) + 16);
					const float ox = r0.x, oy = r0.y, oz = r0.z, dx = r0.w, dy = r1.x, dz = r1.y, tmin_ray = r1.z, tmax_ray = r1.w;
					const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
					const float m = sse_max(sse_max(ax, ay), az);
					const bool kz0 = ax == m, kz1 = !kz0 && ay == m;
					const float dkx = kz0 ? dy : (kz1 ? dz : dx), dky = kz0 ? dz : (kz1 ? dx : dy), dkz = kz0 ? dx : (kz1 ? dy : dz);
					const float shx = -dkx / dkz, shy = -dky / dkz;
					const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;          // rtk.c:410: true divides
					const float shz = kz0 ? rdx : (kz1 ? rdy : rdz);
					const bool special = !(isfinite(rdx) && isfinite(rdy) && isfinite(rdz) && rdx != 0.0f && rdy != 0.0f && rdz != 0.0f &&
						isfinite(ox) && isfinite(oy) && isfinite(oz) && tmin_ray == tmin_ray && tmax_ray == tmax_ray);
					if (special) {
						// rtk_trace_kernel's exact-node path with the reference's min/max operand order takes these afterwards
						const unsigned long long at = atomicAdd(p.counter + RTK_POOL_LEFTOVER_WORD, 1ull);
						p.pool_leftover[at] = ray_index;
						to_free = true;
					} else {
						L.f0[slot] = make_float4(ox, oy, oz, tmin_ray);
						L.f1[slot] = make_float4(rdx, rdy, rdz, tmax_ray);
						L.f2[slot] = make_uint4(0u, (kz0 ? 0x10000u : 0u) | (kz1 ? 0x20000u : 0u), RTK_PRIM_NONE, ray_index);
						L.f3[slot] = make_float4(shx, shy, shz, 0.0f);
						L.f4[slot] = make_float2(0.0f, tmax_ray);
						to_node = true;
					}
				}
			}
		}

		// ------------------------------------------------------------ every ray to the queue of its next state
		POOL_STAT(const unsigned long long c_t5 = __builtin_readcyclecounter();)
		{
			// the count of unfinished rays: up by the rays just set up, down by the rays just finished (no answer needed: no wait)
			const uint32_t up = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(kind == Q_FREE && to_node));
			const uint32_t down = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(kind != Q_FREE && to_free));
			if (lane == 0u && up != down) __hip_atomic_fetch_add(&L.ctl[C_LIVE], up - down, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		pool_push_all(L, to_node ? 0u : to_leaf ? 1u : to_free ? 2u : 3u, slot, lane);
		POOL_STAT(st_clk[kind] += __builtin_readcyclecounter() - c_t1; if (kind == Q_NODE) st_seg[3] += __builtin_readcyclecounter() - c_t5;)
		if (__builtin_amdgcn_ballot_w64(!ok) != 0ull && lane == 0u) {
			__hip_atomic_store(&L.ctl[C_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			p.counter[RTK_ERROR_WORD] = 2ull;
		}
	}
#ifdef POOL_STATS
	if (lane == 0u) {
		for (int q = 0; q < 3; q++) { atomicAdd(p.counter + 1 + q, st_trips[q]); atomicAdd(p.counter + 4 + q, st_rays[q]); }
		atomicAdd(p.counter + 7, st_spins); atomicAdd(p.counter + 8, st_steps);
		atomicAdd(p.counter + 0, st_cas); atomicAdd(p.counter + 10, st_seg[0]); atomicAdd(p.counter + 11, st_seg[1]); atomicAdd(p.counter + 24, st_seg[2]); atomicAdd(p.counter + 25, st_seg[3]);
		atomicAdd(p.counter + 9, st_clk[0]); atomicAdd(p.counter + 13, st_clk[1]); atomicAdd(p.counter + 14, st_clk[2]); atomicAdd(p.counter + 15, st_clk[3]);
	}
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The sticky form of the pool (rtk_trace_pool_sticky_kernel below) shares these pieces with nothing else: the node arithmetic, the
// leaf trip and the set-up trip of the kernel above as functions (same text, same results).

// node arithmetic, child sort and pushes for one ray whose compressed node has arrived (l0..l3); top / sp updated
template <int MODE>
__device__ __forceinline__ void pool_node_math(const TraceParams &p, PoolLds &L, uint2 *spill, uint32_t slot, const float4 s0, const float4 s1,
	const f32x4 l0, const u32x4 l1, const u32x4 l2, const u32x4 l3, uint32_t &top, uint32_t &sp)
{
	const float best_t = s1.w, tmin_ray = s0.w;
	const float ox = s0.x, oy = s0.y, oz = s0.z, rdx = s1.x, rdy = s1.y, rdz = s1.z;
	// (the arithmetic of rtk_trace_kernel's compressed-node step, see the comments there)
	const float Ax = (l0.x - ox) * rdx, Ay = (l0.y - oy) * rdy, Az = (l0.z - oz) * rdz;
	const float Sx = l0.w * rdx, Sy = __uint_as_float(l1.x) * rdy, Sz = __uint_as_float(l1.y) * rdz;
	const float ex = 0x1p-21f * __builtin_fmaf(fabsf(Sx), 255.0f, fabsf(Ax));
	const float ey = 0x1p-21f * __builtin_fmaf(fabsf(Sy), 255.0f, fabsf(Ay));
	const float ez = 0x1p-21f * __builtin_fmaf(fabsf(Sz), 255.0f, fabsf(Az));
	const float Anx = Ax - ex, Afx = Ax + ex, Any = Ay - ey, Afy = Ay + ey, Anz = Az - ez, Afz = Az + ez;
	const bool ngx = (__float_as_uint(rdx) >> 31) != 0u, ngy = (__float_as_uint(rdy) >> 31) != 0u, ngz = (__float_as_uint(rdz) >> 31) != 0u;
	const uint32_t wnx = ngx ? l1.w : l1.z, wfx = ngx ? l1.z : l1.w;
	const uint32_t wny = ngy ? l2.y : l2.x, wfy = ngy ? l2.x : l2.y;
	const uint32_t wnz = ngz ? l2.w : l2.z, wfz = ngz ? l2.z : l2.w;
	uint32_t ref[4] = { l3.x, l3.y, l3.z, l3.w };
	float key[4];
	uint32_t nhit = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const f32x2 px = __builtin_elementwise_fma((f32x2){ ubyte_f32(wnx, i), ubyte_f32(wfx, i) }, (f32x2){ Sx, Sx }, (f32x2){ Anx, Afx });
		const f32x2 py = __builtin_elementwise_fma((f32x2){ ubyte_f32(wny, i), ubyte_f32(wfy, i) }, (f32x2){ Sy, Sy }, (f32x2){ Any, Afy });
		const f32x2 pz = __builtin_elementwise_fma((f32x2){ ubyte_f32(wnz, i), ubyte_f32(wfz, i) }, (f32x2){ Sz, Sz }, (f32x2){ Anz, Afz });
		const float tn = fmaxf(fmaxf(fmaxf(px.x, py.x), pz.x), tmin_ray);
		const float tf = fminf(fminf(fminf(px.y, py.y), pz.y), best_t);
		const bool hit = tn <= tf && ref[i] != RTK_REF_NONE;
		key[i] = hit ? tn : __builtin_inff();
		nhit += hit ? 1u : 0u;
	}
	// nearest first (rtk.c:496-517 orders by entry distance)
	cswap(key[0], ref[0], key[1], ref[1]);
	cswap(key[2], ref[2], key[3], ref[3]);
	cswap(key[0], ref[0], key[2], ref[2]);
	cswap(key[1], ref[1], key[3], ref[3]);
	cswap(key[1], ref[1], key[2], ref[2]);
	if (nhit == 0u) top = RTK_REF_RETRY;
	else {
		top = ref[0];
		const uint32_t np = nhit - 1u;              // sorted slots np..1 go on the stack, far to near
#pragma unroll
		for (int i = 1; i <= 3; i++) {
			if ((uint32_t)i <= np) {
				const uint32_t row = sp + np - (uint32_t)i;
				const uint2 e = make_uint2(__float_as_uint(key[i]), ref[i]);
				if (row < POOL_STACK) L.stack[row][slot] = e;
				else if (row - POOL_STACK < p.spill_cap) spill[(size_t)(row - POOL_STACK) * POOL_SLOTS + slot] = e;
				else p.counter[RTK_ERROR_WORD] = 1ull;         // cannot happen for a tree (rtk_upload.hip rejects anything else)
			}
		}
		sp += np;
	}
}

// one stack entry off a ray's stack (LDS or spill area): the node or leaf it names, RTK_REF_RETRY if the entry lies behind the hit,
// RTK_REF_NONE if the stack is empty
__device__ __forceinline__ void pool_pop(PoolLds &L, const uint2 *spill, uint32_t slot, float best_t, uint32_t &top, uint32_t &sp)
{
	if (sp == 0u) { top = RTK_REF_NONE; return; }
	--sp;
	uint2 e = L.stack[sp < POOL_STACK ? sp : POOL_STACK - 1u][slot];
	if (sp >= POOL_STACK) {
		const unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(spill + (size_t)(sp - POOL_STACK) * POOL_SLOTS + slot));
		e = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
	}
	top = __uint_as_float(e.x) > best_t ? RTK_REF_RETRY : e.y;
}

// the triangles of the leaves of up to 64 rays (`act` lanes, `slot` each); afterwards a ray goes back to the node queue (to_node) or is finished (to_free)
template <int MODE>
__device__ __forceinline__ void pool_leaf_trip(const TraceParams &p, PoolLds &L, const char *tris, uint32_t slot, bool act, bool &to_node, bool &to_free)
{
	// -------------------------------------------------------- the triangles of one leaf (rtk.c:212-386), as rtk_trace_kernel takes them
	uint4 s2 = make_uint4(0u, 0u, RTK_PRIM_NONE, 0u);
	float4 s0 = make_float4(0, 0, 0, 0), s3 = make_float4(0, 0, 0, 0);
	float2 s4 = make_float2(0, 0);
	float best_t = 0;
	if (act) { s2 = L.f2[slot]; s0 = L.f0[slot]; s3 = L.f3[slot]; s4 = L.f4[slot]; best_t = L.f1[slot].w; }
	const bool kz0 = (s2.y & 0x10000u) != 0u, kz1 = (s2.y & 0x20000u) != 0u;
	const float sox = kz0 ? s0.y : (kz1 ? s0.z : s0.x), soy = kz0 ? s0.z : (kz1 ? s0.x : s0.y), soz = kz0 ? s0.x : (kz1 ? s0.y : s0.z);
	const float shx = s3.x, shy = s3.y, shz = s3.z, tmin_ray = s0.w, tmax_ray = s4.y;
	float best_u = s3.w, best_v = s4.x;
	uint32_t best_prim = s2.z;
	if (act) {
		const uint32_t slot0 = s2.x & 0x7fffffffu;
		uint32_t i = 0, n = 1;
		bool force = false, redo = false;
		float sn_t = best_t, sn_u = best_u, sn_v = best_v;
		uint32_t sn_prim = best_prim;
		while (i < n) {
			f32x4 A, B, C;
			load_tri(tris, (slot0 + i) * (uint32_t)RTK_TRI_STRIDE, A, B, C);
			if (i == 0u) n = __float_as_uint(C.w);          // leaf size rides in the first record
			if ((i & 3u) == 0u) {
				if (redo) { force = true; redo = false; }
				else {
					if (MODE == 1 && best_prim != RTK_PRIM_NONE) break;   // any-hit: a whole group accepted something
					force = (n - i) < 4u;
					sn_t = best_t; sn_u = best_u; sn_v = best_v; sn_prim = best_prim;
				}
			}
			// permute to (kx,ky,kz) and move the origin (rtk.c:232-280)
			const float v0x = (kz0 ? A.y : (kz1 ? A.z : A.x)) - sox;
			const float v0y = (kz0 ? A.z : (kz1 ? A.x : A.y)) - soy;
			const float v0z = (kz0 ? A.x : (kz1 ? A.y : A.z)) - soz;
			const float v1x = (kz0 ? B.y : (kz1 ? B.z : B.x)) - sox;
			const float v1y = (kz0 ? B.z : (kz1 ? B.x : B.y)) - soy;
			const float v1z = (kz0 ? B.x : (kz1 ? B.y : B.z)) - soz;
			const float v2x = (kz0 ? C.y : (kz1 ? C.z : C.x)) - sox;
			const float v2y = (kz0 ? C.z : (kz1 ? C.x : C.y)) - soy;
			const float v2z = (kz0 ? C.x : (kz1 ? C.y : C.z)) - soz;
			// shear (rtk.c:284-292)
			const float x0 = v0x + shx * v0z, y0 = v0y + shy * v0z, z0 = shz * v0z;
			const float x1 = v1x + shx * v1z, y1 = v1y + shy * v1z, z1 = shz * v1z;
			const float x2 = v2x + shx * v2z, y2 = v2y + shy * v2z, z2 = shz * v2z;
			// edge functions (rtk.c:298-300)
			float u, v, w;
			if (!force) {
				u = x1 * y2 - y1 * x2;
				v = x2 * y0 - y2 * x0;
				w = x0 * y1 - y0 * x1;
				if (u == 0.0f || v == 0.0f || w == 0.0f) {
					// rtk.c:306: the whole group switches to double precision
					best_t = sn_t; best_u = sn_u; best_v = sn_v; best_prim = sn_prim;
					redo = true;
					i &= ~3u;
					continue;
				}
			} else {
				const double xd0 = x0, yd0 = y0, xd1 = x1, yd1 = y1, xd2 = x2, yd2 = y2;
				u = (float)(xd1 * yd2 - yd1 * xd2);
				v = (float)(xd2 * yd0 - yd2 * xd0);
				w = (float)(xd0 * yd1 - yd0 * xd1);
			}
			// rtk.c:340-342
			const bool neg = sse_min(sse_min(u, v), w) < 0.0f;
			const bool pos = sse_max(sse_max(u, v), w) > 0.0f;
			// rtk.c:346-353
			const float det = (u + v) + w;
			const float rcp = 1.0f / det;
			float zz = u * z0;
			zz = zz + v * z1;
			zz = zz + w * z2;
			const float t = zz * rcp;
			const uint32_t prim = __float_as_uint(A.w);
			const bool in_range = !(neg && pos) && t > tmin_ray && t < tmax_ray;   // rtk.c:354
			if (MODE == 1) {
				if (in_range && best_prim == RTK_PRIM_NONE) { best_prim = prim; best_t = t; }
			} else {
				// rtk.c:371 with the canonical tie rule: lowest primitive id among bit-equal t
				if (in_range && (t < best_t || (t == best_t && prim < best_prim))) {
					best_t = t; best_u = u * rcp; best_v = v * rcp; best_prim = prim;
				}
			}
			i++;
		}
		if (MODE == 1 && best_prim != RTK_PRIM_NONE) {
			p.occluded[s2.w] = 1;
			to_free = true;
		} else {
			L.f1[slot].w = best_t;
			L.f3[slot].w = best_u;
			L.f4[slot].x = best_v;
			L.f2[slot] = make_uint4(RTK_REF_RETRY, s2.y, best_prim, s2.w);      // the next node trip pops
			to_node = true;
		}
	}
}

// new rays into up to 64 free slots (`act` lanes): to_node for a ray that is set up, to_free for a slot whose ray was left to rtk_trace_kernel
__device__ __forceinline__ void pool_setup_trip(const TraceParams &p, PoolLds &L, uint32_t slot, bool act, uint32_t lane, uint32_t &queue, uint32_t &queues_left,
	unsigned long long num_chunks, bool &to_node, bool &to_free)
{
	// -------------------------------------------------------- new rays into free slots (rtk.c:550-566)
	unsigned long long chunk = ~0ull;
	if (lane == 0u) {
		while (queues_left) {
			const unsigned long long got = atomicAdd(p.counter + RTK_QUEUE_WORD(queue) + 8, 1ull);
			const unsigned long long c = got * RTK_QUEUES + queue;
			if (c < num_chunks) { chunk = c; break; }
			queue = (queue + 1u) % RTK_QUEUES;
			queues_left--;
		}
	}
	queue = first_lane(queue); queues_left = first_lane(queues_left);
	chunk = ((unsigned long long)first_lane((uint32_t)(chunk >> 32)) << 32) | first_lane((uint32_t)chunk);
	if (chunk == ~0ull) {
		// no rays left: the slots just claimed are dropped, and nobody asks for free slots any more
		if (lane == 0u) __hip_atomic_store(&L.ctl[C_EXHAUSTED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	} else {
		const unsigned long long i = (chunk << 6) + lane;
		const bool valid = act && i < p.n;
		if (valid) {
			const uint32_t ray_index = p.perm ? (uint32_t)p.perm[i] : (uint32_t)i;      // perm: sort words, ray number in the low half
			const float4 r0 = ld_f4_stream(reinterpret_cast<const char *>(p.rays + ray_index));
			const float4 r1 = ld_f4_stream(reinterpret_cast<const char *>(p.rays + ray_index) + 16);
			const float ox = r0.x, oy = r0.y, oz = r0.z, dx = r0.w, dy = r1.x, dz = r1.y, tmin_ray = r1.z, tmax_ray = r1.w;
			const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
			const float m = sse_max(sse_max(ax, ay), az);
			const bool kz0 = ax == m, kz1 = !kz0 && ay == m;
			const float dkx = kz0 ? dy : (kz1 ? dz : dx), dky = kz0 ? dz : (kz1 ? dx : dy), dkz = kz0 ? dx : (kz1 ? dy : dz);
			const float shx = -dkx / dkz, shy = -dky / dkz;
			const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;          // rtk.c:410: true divides
			const float shz = kz0 ? rdx : (kz1 ? rdy : rdz);
			const bool special = !(isfinite(rdx) && isfinite(rdy) && isfinite(rdz) && rdx != 0.0f && rdy != 0.0f && rdz != 0.0f &&
				isfinite(ox) && isfinite(oy) && isfinite(oz) && tmin_ray == tmin_ray && tmax_ray == tmax_ray);
			if (special) {
				// rtk_trace_kernel's exact-node path with the reference's min/max operand order takes these afterwards
				const unsigned long long at = atomicAdd(p.counter + RTK_POOL_LEFTOVER_WORD, 1ull);
				p.pool_leftover[at] = ray_index;
				to_free = true;
			} else {
				L.f0[slot] = make_float4(ox, oy, oz, tmin_ray);
				L.f1[slot] = make_float4(rdx, rdy, rdz, tmax_ray);
				L.f2[slot] = make_uint4(0u, (kz0 ? 0x10000u : 0u) | (kz1 ? 0x20000u : 0u), RTK_PRIM_NONE, ray_index);
				L.f3[slot] = make_float4(shx, shy, shz, 0.0f);
				L.f4[slot] = make_float2(0.0f, tmax_ray);
				to_node = true;
			}
		}
	}
}

// the four 16-byte pieces of a compressed node requested WITHOUT waiting for them, and the wait. Between the two a wave does
// LDS work only (no other vector memory instruction may be issued in between: the wait is for all of them)
__device__ __forceinline__ void qnode_issue(const char *base, uint32_t a_node, f32x4 &l0, u32x4 &l1, u32x4 &l2, u32x4 &l3)
{
	asm volatile(
		"global_load_dwordx4 %0, %4, %5\n\t"
		"global_load_dwordx4 %1, %4, %5 offset:16\n\t"
		"global_load_dwordx4 %2, %4, %5 offset:32\n\t"
		"global_load_dwordx4 %3, %4, %5 offset:48"
		: "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
		: "v"(a_node), "s"(base)
		: "memory");
}
__device__ __forceinline__ void qnode_wait(f32x4 &l0, u32x4 &l1, u32x4 &l2, u32x4 &l3)
{
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3) :: "memory");
}

// claim up to 64 entries of ring q (lane 0 decides; a full batch only if `full`): how many, and the first position
__device__ __forceinline__ uint32_t pool_claim(PoolLds &L, uint32_t q, bool full, uint32_t want, uint32_t lane, uint32_t &first)
{
	uint32_t k = 0, h = 0;
	if (lane == 0u) {
		for (int attempt = 0; attempt < 4; attempt++) {
			const uint32_t hq = lds_load(&L.ctl[2u * q]), tq = lds_load(&L.ctl[2u * q + 1u]);
			uint32_t kk = tq - hq;
			if (kk > want) kk = want;
			if (kk == 0u || (full && kk < want)) break;
			uint32_t expect = hq;
			if (__hip_atomic_compare_exchange_strong(&L.ctl[2u * q], &expect, hq + kk, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { k = kk; h = hq; break; }
		}
	}
	first = first_lane(h);
	return first_lane(k);
}

// the slot number at ring position pos of queue q, once its writer has written it
__device__ __forceinline__ uint32_t pool_ring_read(PoolLds &L, uint32_t q, uint32_t pos, bool &ok)
{
	const uint32_t lap = (pos / POOL_RING) & 31u;
	const uint16_t *e = &L.ring[q][pos & (POOL_RING - 1u)];
	uint32_t spin = 0, v;
	while (((v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 11) != lap && spin < POOL_SPIN_LIMIT) { spin++; __builtin_amdgcn_s_sleep(1); }
	ok = ok && spin < POOL_SPIN_LIMIT;
	return spin < POOL_SPIN_LIMIT ? (v & 2047u) : 0u;
}

#define POOL_NO_SLOT 0xffffu
#define POOL_TOPUP_MIN 8u          // holes in a wave before it asks the node queue for rays

// Sticky form: a ray that stays in node state stays in its LANE, with its state in registers (origin, reciprocal direction, interval,
// top, stack size; its stack is in LDS by slot as before). Only rays that change state go through LDS: a ray that reached a leaf is
// written back (two words) and handed to the leaf queue, a finished one is retired; the holes are filled from the node queue -- rays
// that come back from leaf trips, and new ones -- WHILE the node fetch of the rays that stayed is in flight (the queue bookkeeping of
// the first form was a chain of LDS round trips in front of every fetch: 9 500 clk per trip; here the chain in front of the fetch is the
// pop). Leaf and set-up trips are as above, on full batches, by whichever wave finds one waiting.
template <int MODE /*0 closest hit, 1 any hit*/>
__global__ void __launch_bounds__(POOL_THREADS, 1) rtk_trace_pool_sticky_kernel(TraceParams p)
{
	extern __shared__ __align__(16) unsigned char pool_lds_raw[];
	PoolLds &L = *reinterpret_cast<PoolLds *>(pool_lds_raw);
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const char *const qnodes = reinterpret_cast<const char *>(p.sc.qnodes);
	const char *const tris = reinterpret_cast<const char *>(p.sc.tris);
	uint2 *const spill = p.spill + (size_t)blockIdx.x * POOL_SLOTS * p.spill_cap;

	for (uint32_t i = threadIdx.x; i < 3u * POOL_RING; i += POOL_THREADS) (&L.ring[0][0])[i] = (uint16_t)POOL_EMPTY;
	if (threadIdx.x < 16u) L.ctl[threadIdx.x] = 0u;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < POOL_SLOTS; i += POOL_THREADS) L.ring[Q_FREE][i] = (uint16_t)i;
	if (threadIdx.x == 0u) L.ctl[2 * Q_FREE + 1] = POOL_SLOTS;
	__syncthreads();

	uint32_t queue = (blockIdx.x * POOL_WAVES + wave) % RTK_QUEUES, queues_left = RTK_QUEUES;
	const unsigned long long num_chunks = (p.n + 63ull) >> 6;

	// the ray this lane keeps (node state): slot (POOL_NO_SLOT: none), state, top, stack size | axis bits
	uint32_t n_slot = POOL_NO_SLOT, n_top = RTK_REF_NONE, n_spw = 0u;
	float4 n_s0 = make_float4(0, 0, 0, 0), n_s1 = make_float4(0, 0, 0, 0);
	// what the queues looked like the last time this wave looked (during its last node fetch)
	uint32_t sn_nn = 0u, sn_nl = 0u, sn_nf = POOL_SLOTS, sn_live = 0u, sn_ex = 0u;
	uint32_t idle = 0u;
	bool ok = true;
#ifdef POOL_STATS
	unsigned long long ss_trips = 0, ss_step = 0, ss_have = 0, ss_top_try = 0, ss_top_got = 0, ss_top_rays = 0, ss_batch[2] = { 0, 0 }, ss_batch_rays[2] = { 0, 0 }, ss_idle = 0, ss_pop = 0;
#endif

	for (;;) {
		const unsigned long long have_m = __builtin_amdgcn_ballot_w64(n_slot != POOL_NO_SLOT);
		const uint32_t have = (uint32_t)__popcll(have_m);
		// ------------------------------------------------------------ a full batch of leaves or of free slots waits: take it
		int batch = -1;
		if (sn_nl >= 64u) batch = Q_LEAF;
		else if (sn_nf >= 64u && !sn_ex) batch = Q_FREE;
		else if (have == 0u && sn_nn == 0u && sn_nl != 0u && (sn_nl >= sn_live || idle >= 8u)) batch = Q_LEAF;     // the stragglers: a partial batch
		if (batch >= 0) {
			uint32_t h = 0;
			const uint32_t k = pool_claim(L, (uint32_t)batch, sn_nl >= 64u || batch == Q_FREE, 64u, lane, h);
			bool to_node = false, to_free = false;
			uint32_t slot = 0;
			const bool act = lane < k;
			POOL_STAT(if (k) { ss_batch[batch == Q_FREE]++; ss_batch_rays[batch == Q_FREE] += k; })
			if (k != 0u) {
				if (act) slot = pool_ring_read(L, (uint32_t)batch, h + lane, ok);
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				if (batch == Q_LEAF) pool_leaf_trip<MODE>(p, L, tris, slot, act, to_node, to_free);
				else pool_setup_trip(p, L, slot, act, lane, queue, queues_left, num_chunks, to_node, to_free);
				const uint32_t up = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(batch == Q_FREE && to_node));
				const uint32_t down = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(batch != Q_FREE && to_free));
				if (lane == 0u && up != down) __hip_atomic_fetch_add(&L.ctl[C_LIVE], up - down, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				pool_push_all(L, to_node ? 0u : to_free ? 2u : 3u, slot, lane);
			}
		}
		if (batch >= 0 || (have == 0u && sn_nn == 0u)) {
			// look again (a wave without rays of its own has no node fetch to look behind)
			PoolSnap sn = { 0, 0, 0, 0, 0, 0, 0, 0 };
			uint32_t aborted = 0u;
			if (lane == 0u) { sn = pool_snapshot(L.ctl); aborted = lds_load(&L.ctl[C_ABORT]); }
			sn_nn = first_lane(sn.nn); sn_nl = first_lane(sn.nl); sn_ex = first_lane(sn.ex); sn_nf = first_lane(sn.nf); sn_live = first_lane(sn.live);
			if (first_lane(aborted) != 0u) break;
			if (batch < 0) {
				if (sn_nn == 0u && sn_nl == 0u && sn_ex && sn_live == 0u) break;          // nothing queued, no ray unfinished, no rays left: done
				if (sn_nn == 0u && !(sn_nl >= 64u) && !(sn_nf >= 64u && !sn_ex)) {
					if (++idle >= POOL_SPIN_LIMIT) { ok = false; break; }
					POOL_STAT(ss_idle++;)
					__builtin_amdgcn_s_sleep(4);
				}
			} else idle = 0u;
			if (have == 0u && sn_nn == 0u) continue;
			if (batch >= 0) continue;
		}
		idle = 0u;

		// ------------------------------------------------------------ node trip for the rays in the lanes
		// a ray that comes from a leaf, from a node it missed entirely or from a culled entry pops first
		if (__builtin_amdgcn_ballot_w64(n_slot != POOL_NO_SLOT && n_top == RTK_REF_RETRY) != 0ull) {
			if (n_slot != POOL_NO_SLOT && n_top == RTK_REF_RETRY) { uint32_t sp = n_spw & 0xffffu; pool_pop(L, spill, n_slot, n_s1.w, n_top, sp); n_spw = (n_spw & 0xffff0000u) | sp; }
		}
		const bool stepping = n_slot != POOL_NO_SLOT && (int32_t)n_top >= 0;
		POOL_STAT(ss_trips++; ss_step += __popcll(__builtin_amdgcn_ballot_w64(stepping)); ss_have += have;)
		f32x4 l0 = { 0, 0, 0, 0 };
		u32x4 l1 = { 0, 0, 0, 0 }, l2 = { 0, 0, 0, 0 }, l3 = { 0, 0, 0, 0 };
		if (stepping) qnode_issue(qnodes, n_top << 6, l0, l1, l2, l3);
		// ... and while the nodes are on their way: how do the queues look, and rays for the lanes that have none (they step next trip)
		{
			PoolSnap sn = { 0, 0, 0, 0, 0, 0, 0, 0 };
			if (lane == 0u) sn = pool_snapshot(L.ctl);
			sn_nn = first_lane(sn.nn); sn_nl = first_lane(sn.nl); sn_ex = first_lane(sn.ex); sn_nf = first_lane(sn.nf); sn_live = first_lane(sn.live);
			const bool hole = n_slot == POOL_NO_SLOT;
			const unsigned long long hole_m = __builtin_amdgcn_ballot_w64(hole);
			const uint32_t nh = (uint32_t)__popcll(hole_m);
			if (sn_nn != 0u && (nh >= POOL_TOPUP_MIN || nh == 64u)) {
				uint32_t h = 0;
				const uint32_t k = pool_claim(L, Q_NODE, false, nh, lane, h);
				POOL_STAT(ss_top_try++; if (k) { ss_top_got++; ss_top_rays += k; })
				const uint32_t rank = lane_rank(hole_m);
				if (hole && rank < k) {
					n_slot = pool_ring_read(L, Q_NODE, h + rank, ok);
				}
				if (k != 0u) {
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					if (hole && rank < k) {
						const uint4 s2 = L.f2[n_slot];
						n_s1 = L.f1[n_slot]; n_s0 = L.f0[n_slot];
						n_top = s2.x; n_spw = s2.y;
					}
				}
			}
		}
		qnode_wait(l0, l1, l2, l3);
		if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) break;             // a ring entry that never came: give up (flagged below)
		if (stepping) { uint32_t sp = n_spw & 0xffffu; pool_node_math<MODE>(p, L, spill, n_slot, n_s0, n_s1, l0, l1, l2, l3, n_top, sp); n_spw = (n_spw & 0xffff0000u) | sp; }
		// rays that leave their lane: to the leaf queue, or finished. (Lanes filled during this trip keep their ray whatever its state:
		// a RETRY pops next trip; a ray can only have come off the node queue in node or retry state.)
		const bool was_mine = (have_m >> lane) & 1ull;
		const bool leaf = was_mine && (int32_t)n_top < 0 && n_top != RTK_REF_RETRY && n_top != RTK_REF_NONE;
		const bool done = was_mine && n_top == RTK_REF_NONE;
		if (done) {
			const uint4 s2 = L.f2[n_slot];
			if (MODE == 1) p.occluded[s2.w] = 0;
			else st_f4_stream(p.hits + s2.w, n_s1.w, L.f3[n_slot].w, L.f4[n_slot].x, __uint_as_float(s2.z));
		}
		if (leaf) *reinterpret_cast<uint2 *>(&L.f2[n_slot]) = make_uint2(n_top, n_spw);
		if (__builtin_amdgcn_ballot_w64(leaf || done) != 0ull) {
			const uint32_t down = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(done));
			if (lane == 0u && down != 0u) __hip_atomic_fetch_add(&L.ctl[C_LIVE], 0u - down, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			pool_push_all(L, leaf ? 1u : done ? 2u : 3u, n_slot, lane);
			if (leaf || done) { n_slot = POOL_NO_SLOT; n_top = RTK_REF_NONE; }
		}
	}
	if (__builtin_amdgcn_ballot_w64(!ok) != 0ull && lane == 0u) {
		__hip_atomic_store(&L.ctl[C_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		p.counter[RTK_ERROR_WORD] = 2ull;
	}
#ifdef POOL_STATS
	if (lane == 0u) {
		atomicAdd(p.counter + 1, ss_trips); atomicAdd(p.counter + 2, ss_batch[0]); atomicAdd(p.counter + 3, ss_batch[1]); atomicAdd(p.counter + 4, ss_step); atomicAdd(p.counter + 5, ss_batch_rays[0]);
		atomicAdd(p.counter + 6, ss_batch_rays[1]); atomicAdd(p.counter + 7, ss_idle); atomicAdd(p.counter + 8, ss_have); atomicAdd(p.counter + 9, ss_top_try); atomicAdd(p.counter + 13, ss_top_got); atomicAdd(p.counter + 14, ss_top_rays);
	}
#endif
}

#ifdef POOL_STATS
// (debug build only: trips, rays and clocks per kind of trip, in the launch's visit-counter words)
#endif
int rtk_pool_slots() { return (int)POOL_SLOTS; }
int rtk_pool_lds_stack() { return (int)POOL_STACK; }

int rtk_pool_launch(const TraceParams &p, unsigned blocks, hipStream_t stream, bool any_hit)
{
	// (the attribute belongs to the function ON THE CURRENT DEVICE: set once per device)
	static std::mutex mutex;
	static bool set[RTK_MAX_DEVICES];
	hipError_t attr = hipSuccess;
	{
		std::lock_guard<std::mutex> lock(mutex);
		int dev = 0;
		(void)hipGetDevice(&dev);
		if (dev < 0 || dev >= RTK_MAX_DEVICES) dev = 0;
		if (!set[dev]) {
			attr = hipFuncSetAttribute(reinterpret_cast<const void *>(rtk_trace_pool_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PoolLds));
			if (attr == hipSuccess) attr = hipFuncSetAttribute(reinterpret_cast<const void *>(rtk_trace_pool_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PoolLds));
			if (attr == hipSuccess) attr = hipFuncSetAttribute(reinterpret_cast<const void *>(rtk_trace_pool_sticky_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PoolLds));
			if (attr == hipSuccess) attr = hipFuncSetAttribute(reinterpret_cast<const void *>(rtk_trace_pool_sticky_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PoolLds));
			set[dev] = attr == hipSuccess;
		}
	}
	if (attr != hipSuccess) { rtk_set_error("rtk_dev_trace: a workgroup cannot have %zu bytes of LDS on this device (%s)", sizeof(PoolLds), hipGetErrorString(attr)); return RTK_AMD_ERR_HIP; }
	const char *sticky_env = getenv("RTK_AMD_POOL_STICKY");
	const bool sticky = sticky_env ? atoi(sticky_env) != 0 : true;
	if (sticky) {
		if (any_hit) hipLaunchKernelGGL(rtk_trace_pool_sticky_kernel<1>, dim3(blocks), dim3(POOL_THREADS), sizeof(PoolLds), stream, p);
		else hipLaunchKernelGGL(rtk_trace_pool_sticky_kernel<0>, dim3(blocks), dim3(POOL_THREADS), sizeof(PoolLds), stream, p);
	} else if (any_hit) hipLaunchKernelGGL(rtk_trace_pool_kernel<1>, dim3(blocks), dim3(POOL_THREADS), sizeof(PoolLds), stream, p);
	else hipLaunchKernelGGL(rtk_trace_pool_kernel<0>, dim3(blocks), dim3(POOL_THREADS), sizeof(PoolLds), stream, p);
	return RTK_AMD_OK;
}
