// rtk_trace.hip -- BVH4 traversal + watertight triangle test for gfx950 (MI355X).
//
// What it computes is the reference's rtk_trace_ray (rtk.c:543-577) for a whole batch:
//   ray setup        rtk.c:550-566   -> ray_setup()
//   4-wide slab test rtk.c:457-472   -> node_step()
//   triangle test    rtk.c:284-364   -> tri_test()   (per triangle, not per group of 4)
//   closest update   rtk.c:366-386   -> canonical tie rule, DESIGN.md "Ties"
// How it runs is CDNA4-specific: one ray per lane of a 64-wide wave, persistent waves
// that pull ray chunks from a global counter and re-fill idle lanes by ballot rank, the
// per-lane traversal stack in LDS ([entry][lane] so that a wave's push/pop is one
// conflict-free ds_write_b64/ds_read_b64), 128 B nodes = one cache line per visit.
//
// FLOATING POINT: this file must be compiled with -ffp-contract=off. The float operation
// order in tri_test() is normative (sign of u,v,w decides hit/miss); see SURVEY.md
// section 0. Divisions are IEEE (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
#include "rtk_dev.h"

#include <algorithm>
#include "rtk_trace_shared.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "rtk_trace_lane.h"

// A push that does not fit the stack (LDS entries + the global spill area sized from the tree depth) is
// dropped WITHOUT advancing sp and flagged; it cannot happen for a tree (at most three pushes per level,
// rtk_upload.hip rejects blobs that are not trees), so a set flag means a corrupted scene, never an
// out-of-bounds access.
#define RTK_PUSH(e_)                                                                                        \
	do {                                                                                                    \
		if (sp < LDS_STACK) { stk[sp][lane] = (e_); sp++; }                                                 \
		else if (sp - LDS_STACK < p.spill_cap) {                                                            \
			p.spill[(size_t)(sp - LDS_STACK) * p.spill_stride + glane] = (e_); sp++; if (COUNT) c_spills++; \
		} else p.counter[RTK_ERROR_WORD] = 1ull;                                                            \
	} while (0)

// Next node from the stack: ONE entry per call. An entry that starts behind the current hit (rtk.c:432; canonical ties:
// an entry AT the hit distance may still hold an equal-t candidate with a lower id) leaves top = RTK_REF_RETRY and the
// lane pops again at the head of the next node-loop trip, beside the other lanes' node steps. A `while` here cost 5.4
// wave-level trips per step on incoherent rays (7 of a ray's ~27 pops are culled, and the wave waits for its unluckiest
// lane): as many instructions as the box tests. top = NONE when the stack is empty.
#define RTK_REF_RETRY 0xfffffffeu
#define RTK_POP()                                                                                           \
	do {                                                                                                    \
		if (sp == 0u) top = RTK_REF_NONE;                                                                   \
		else {                                                                                              \
			--sp;                                                                                           \
			uint2 e_ = stk[sp < LDS_STACK ? sp : LDS_STACK - 1u][lane];          /* always an LDS read (ds_read_b64), never a flat one */ \
			if (sp >= LDS_STACK) {                       /* nontemporal: read once, and keeps hipcc from merging both into a flat load */ \
				const unsigned long long w_ = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(p.spill + (size_t)(sp - LDS_STACK) * p.spill_stride + glane)); \
				e_ = make_uint2((uint32_t)w_, (uint32_t)(w_ >> 32));                                                \
			}                                                                                                   \
			/* (reference, entry distance). Any-hit: an entry never lies behind best_t, which stays max_t until the ray is over */ \
			top = (MODE != 1 && __uint_as_float(e_.y) > best_t) ? RTK_REF_RETRY : e_.x;                                    \
		}                                                                                                   \
	} while (0)
// One child of a node step: the lane enters it if its slab test passed and the slot is not empty; its key is the entry
// distance (+inf otherwise) and nhit counts the children entered. The condition is kept as the wave's mask so that the
// select reads it directly and the count is ONE add-with-carry (as a bool: a 0/1 select and an add).
#define RTK_CHILD_HIT(slab_ok_, i_)                                                                                   \
	do {                                                                                                              \
		const unsigned long long hm_ = __builtin_amdgcn_ballot_w64(slab_ok_) & __builtin_amdgcn_ballot_w64(ref[i_] != RTK_REF_NONE); \
		key[i_] = __builtin_amdgcn_inverse_ballot_w64(hm_) ? tn : __builtin_inff();                                   \
		asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(nhit) : "s"(hm_) : "vcc");                                 \
	} while (0)
#define RTK_IS_LEAF(top_) ((top_) < RTK_REF_RETRY && (int32_t)(top_) < 0)

#ifndef PL_MIN_WAVES
#define PL_MIN_WAVES 5             // waves per SIMD the register allocator must leave room for (78 VGPRs used; LDS allows five workgroups per CU)
#endif

template <int MODE /*0 closest, 1 any, 2 collect the k closest candidates*/, bool COUNT, bool FILT /*built-in candidate filters*/, bool QN /*64 B compressed nodes*/>
__global__ void __launch_bounds__(BLOCK_THREADS, PL_MIN_WAVES) rtk_trace_kernel(TraceParams p)
{
	__shared__ uint2 s_stack[WAVES_PER_BLOCK][LDS_STACK][64];

	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	uint2 (*stk)[64] = s_stack[wave];
	const uint32_t glane = blockIdx.x * BLOCK_THREADS + threadIdx.x;
	const char *const nodes = reinterpret_cast<const char *>(p.sc.nodes);
	const char *const qnodes = reinterpret_cast<const char *>(p.sc.qnodes);
	const char *const tris = reinterpret_cast<const char *>(p.sc.tris);

	// (a list whose length an earlier kernel of the launch wrote: the rays the assembly kernel, rtk_lane_hot.S, left over)
	if (p.n_indirect) p.n = *p.n_indirect;
	// wave-uniform ray range owned by this wave
	unsigned long long w_next, w_end;
	bool pool_empty;
	uint32_t queue = blockIdx.x % RTK_QUEUES, queues_left = RTK_QUEUES;
	if (p.dynamic) {
		w_next = w_end = 0;
		pool_empty = false;
	} else {
		w_next = ((unsigned long long)blockIdx.x * WAVES_PER_BLOCK + wave) * 64ull;
		w_end = w_next + 64ull < p.n ? w_next + 64ull : p.n;
		if (w_next > p.n) w_next = p.n;
		pool_empty = true;
	}

	// per-lane ray state
	bool active = false;
	uint32_t ray_index = 0;               // the launcher keeps batches below 2^32 rays
	float ox = 0, oy = 0, oz = 0, rdx = 0, rdy = 0, rdz = 0, tmin_ray = 0, tmax_ray = 0;
	float sox = 0, soy = 0, soz = 0, shx = 0, shy = 0, shz = 0;
	bool kz0 = false, kz1 = false;
	uint32_t onx = 0, ony = 0, onz = 0;     // byte offset of the NEAR plane row of each axis inside a node; far = the other row
	float best_t = 0, best_u = 0, best_v = 0;
	uint32_t best_prim = RTK_PRIM_NONE;
	uint32_t cand_n = 0;                                  // MODE 2: candidates collected for this ray (<= p.cand_k), best_t = what the k-th one beats
	float after_t = 0;                                    // FILT: candidates must come after (after_t, after_prim)
	uint32_t after_prim = 0, skip_prim = RTK_PRIM_NONE;   // FILT: ... and must not be skip_prim
	bool has_after = false;
	uint32_t top = RTK_REF_NONE;
	uint32_t sp = 0;
	uint32_t c_nodes = 0, c_leaves = 0, c_tris = 0, c_spills = 0;
	unsigned long long w_node_steps = 0, w_tri_steps = 0;   // COUNT only: wave-level loop trips (divergence diagnostics)
	// A ray is "special" if its slab products can be NaN (0*inf) or its inputs are not finite;
	// only then does the SSE operand order of min/max matter (see node step).
	bool special = false;
	bool wave_fast = true;   // wave-uniform: no active lane is special

	for (;;) {
		// ---------------------------------------------------------------- refill
		const unsigned long long idle = __builtin_amdgcn_ballot_w64(!active);
		const uint32_t n_idle = (uint32_t)__popcll(idle);
		if (n_idle == 64u || (n_idle >= p.refill_min && !(pool_empty && w_next >= w_end))) {
			if (w_next >= w_end && !pool_empty) {
				// chunks of 64 rays are dealt through RTK_QUEUES queue heads (chunk c belongs to queue c % 8,
				// a wave starts on blockIdx % 8 and moves on when a queue is drained): one word only
				// serves ~88 atomics/us, and bigger chunks per atomic unbalance the tail
				const unsigned long long num_chunks = (p.n + 63ull) >> 6;
				pool_empty = true;
				while (queues_left) {
					unsigned long long got = 0;
					if (lane == 0) got = atomicAdd(p.counter + RTK_QUEUE_WORD(queue), 1ull);
					// all 64 lanes are converged here; lane 0's value becomes wave-uniform (SGPRs)
					got = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32)) << 32) |
						(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got);
					const unsigned long long chunk = got * RTK_QUEUES + queue;
					if (chunk < num_chunks) {
						w_next = chunk << 6;
						w_end = w_next + 64ull < p.n ? w_next + 64ull : p.n;
						pool_empty = false;
						break;
					}
					queue = (queue + 1u) % RTK_QUEUES;
					queues_left--;
				}
			}
			const unsigned long long avail = w_end - w_next;
			if (avail) {
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
					__builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
				const uint32_t take = n_idle < avail ? n_idle : (uint32_t)avail;
				if (!active && rank < take) {
					ray_index = p.perm ? (uint32_t)p.perm[w_next + rank] : (uint32_t)map_index(w_next + rank, p.image_w, p.image_h, p.tile_blocks);   // perm: sort words, ray number in the low half
					const float4 r0 = ld_f4_stream(reinterpret_cast<const char *>(p.rays + ray_index));
					const float4 r1 = ld_f4_stream(reinterpret_cast<const char *>(p.rays + ray_index) + 16);
					ox = r0.x; oy = r0.y; oz = r0.z;
					const float dx = r0.w, dy = r1.x, dz = r1.y;
					tmin_ray = r1.z; tmax_ray = r1.w;
					// rtk.c:550-566
					const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
					const float m = sse_max(sse_max(ax, ay), az);
					kz0 = ax == m;
					kz1 = !kz0 && ay == m;
					// (kx,ky,kz): kz==2 -> (x,y,z); kz==0 -> (y,z,x); kz==1 -> (z,x,y)
					const float dkx = kz0 ? dy : (kz1 ? dz : dx);
					const float dky = kz0 ? dz : (kz1 ? dx : dy);
					const float dkz = kz0 ? dx : (kz1 ? dy : dz);
					shx = -dkx / dkz;
					shy = -dky / dkz;
					sox = kz0 ? oy : (kz1 ? oz : ox);
					soy = kz0 ? oz : (kz1 ? ox : oy);
					soz = kz0 ? ox : (kz1 ? oy : oz);
					// rtk.c:410: true divides
					rdx = 1.0f / dx; rdy = 1.0f / dy; rdz = 1.0f / dz;
					shz = kz0 ? rdx : (kz1 ? rdy : rdz);          // 1 / d[kz] is one of the three reciprocals above, bit for bit
					// near/far plane offsets inside the node by direction sign BIT (rtk.c:152-154, 458-463)
					const uint32_t sx = __float_as_uint(dx) >> 31, sy = __float_as_uint(dy) >> 31, sz = __float_as_uint(dz) >> 31;
					onx = sx * 16u;
					ony = 32u + sy * 16u;
					onz = 64u + sz * 16u;
					special = !(isfinite(rdx) && isfinite(rdy) && isfinite(rdz) && rdx != 0.0f && rdy != 0.0f && rdz != 0.0f &&
						isfinite(ox) && isfinite(oy) && isfinite(oz) && tmin_ray == tmin_ray && tmax_ray == tmax_ray);
					best_t = tmax_ray; best_u = 0.0f; best_v = 0.0f; best_prim = RTK_PRIM_NONE;
					if (FILT) {
						has_after = p.after != nullptr;
						if (has_after) { const rtk_hit_record a = p.after[ray_index]; after_t = a.t; after_prim = a.prim; has_after = a.prim != RTK_PRIM_NONE; }
						skip_prim = p.ignore_prim ? p.ignore_prim[ray_index] : RTK_PRIM_NONE;
					}
					top = 0u;  // root node
					sp = 0u;
					cand_n = 0u;
					active = true;
				}
				w_next += take;
			}
			wave_fast = __builtin_amdgcn_ballot_w64(active && special) == 0ull;
			if (__builtin_amdgcn_ballot_w64(active) == 0ull) {
				if (pool_empty && w_next >= w_end) break;
				continue;
			}
		}

		// the active lanes as a wave mask in SCALAR registers (first-lane reads: left to itself hipcc keeps this loop-carried
		// value in a vector register pair): the votes below are votes on plain comparisons ANDed with it on the scalar unit
		// -- a vote on `active && ...` rebuilds the mask from a 0/1 vector register, two vector instructions each
		const unsigned long long m_act_ = __builtin_amdgcn_ballot_w64(active);
		const unsigned long long m_active = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m_act_ >> 32)) << 32) |
			(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m_act_);
		// ---------------------------------------------------------------- inner nodes
		for (;;) {
			// Lanes that reached a leaf wait here for the others. When only a few lanes are
			// still descending and leaves are waiting, go and do the leaves first.
			const bool retry = MODE != 1 && active && top == RTK_REF_RETRY;
			if (retry) RTK_POP();
			const unsigned long long m_node = __builtin_amdgcn_ballot_w64((int32_t)top >= 0) & m_active;
			const bool want_node = __builtin_amdgcn_inverse_ballot_w64(m_node);
			if (m_node == 0ull) {
				if (MODE != 1 && (__builtin_amdgcn_ballot_w64(top == RTK_REF_RETRY) & m_active) != 0ull) continue;   // somebody is still popping
				break;
			}
			// (lanes still popping are not counted as descending: counting them was 1 % slower)
			// (RTK_IS_LEAF as one signed comparison: sign bit set and below RTK_REF_RETRY = -2)
			if ((uint32_t)__popcll(m_node) < p.node_exit && (__builtin_amdgcn_ballot_w64((int32_t)top < (int32_t)RTK_REF_RETRY) & m_active) != 0ull) break;
			if (COUNT) w_node_steps++;
			if (!want_node) continue;
			uint32_t ref[4];
			float key[4];
			uint32_t nhit = 0;
			if (COUNT) c_nodes++;
			if (QN && wave_fast) {
				// Compressed node: plane = org + q * scale, so its ray parameter is A + q * S with A = (org - o) * rcp,
				// S = scale * rcp. Low planes were rounded down and high planes up when the node was made, so the
				// decoded slab interval contains the exact one; a small margin covers the rounding of this arithmetic,
				// so that every child the exact test admits is admitted here too.
				f32x4 l0;
				u32x4 l1, l2, l3;
				load_qnode(qnodes, top << 6, l0, l1, l2, l3);
				const float Ax = (l0.x - ox) * rdx, Ay = (l0.y - oy) * rdy, Az = (l0.z - oz) * rdz;
				const float Sx = l0.w * rdx, Sy = __uint_as_float(l1.x) * rdy, Sz = __uint_as_float(l1.y) * rdz;
				// per axis: the rounding of A + q * S (and of the exact test it stands in for) is below 2^-22 of
				// |A| + 255 |S|; near planes start from A - e, far planes from A + e. Per AXIS on purpose: one tiny
				// direction component makes that axis' A and S huge, and a common margin would open every box.
				const float ex = 0x1p-21f * __builtin_fmaf(fabsf(Sx), 255.0f, fabsf(Ax));
				const float ey = 0x1p-21f * __builtin_fmaf(fabsf(Sy), 255.0f, fabsf(Ay));
				const float ez = 0x1p-21f * __builtin_fmaf(fabsf(Sz), 255.0f, fabsf(Az));
				const float Anx = Ax - ex, Afx = Ax + ex, Any = Ay - ey, Afy = Ay + ey, Anz = Az - ez, Afz = Az + ez;
				const bool ngx = onx != 0u, ngy = ony != 32u, ngz = onz != 64u;      // direction sign bits
				const uint32_t wnx = ngx ? l1.w : l1.z, wfx = ngx ? l1.z : l1.w;
				const uint32_t wny = ngy ? l2.y : l2.x, wfy = ngy ? l2.x : l2.y;
				const uint32_t wnz = ngz ? l2.w : l2.z, wfz = ngz ? l2.z : l2.w;
				ref[0] = l3.x; ref[1] = l3.y; ref[2] = l3.z; ref[3] = l3.w;
#pragma unroll
				for (int i = 0; i < 4; i++) {
					// near and far plane of one axis in one v_pk_fma_f32 (each half is the same single-rounded fma)
					const f32x2 px = __builtin_elementwise_fma((f32x2){ ubyte_f32(wnx, i), ubyte_f32(wfx, i) }, (f32x2){ Sx, Sx }, (f32x2){ Anx, Afx });
					const f32x2 py = __builtin_elementwise_fma((f32x2){ ubyte_f32(wny, i), ubyte_f32(wfy, i) }, (f32x2){ Sy, Sy }, (f32x2){ Any, Afy });
					const f32x2 pz = __builtin_elementwise_fma((f32x2){ ubyte_f32(wnz, i), ubyte_f32(wfz, i) }, (f32x2){ Sz, Sz }, (f32x2){ Anz, Afz });
					const float tn = fmaxf(fmaxf(fmaxf(px.x, py.x), pz.x), tmin_ray);
					const float tf = fminf(fminf(fminf(px.y, py.y), pz.y), best_t);
					RTK_CHILD_HIT(tn <= tf, i);
				}
			} else {
			const uint32_t a_node = top << 7;
			f32x4 nx, fx, ny, fy, nz, fz;
			u32x4 ch;
			load_node(nodes, a_node + onx, (a_node + 16u) - onx, a_node + ony, (a_node + 80u) - ony, a_node + onz, (a_node + 144u) - onz, a_node,
				nx, fx, ny, fy, nz, fz, ch);
			ref[0] = ch.x; ref[1] = ch.y; ref[2] = ch.z; ref[3] = ch.w;
			if (wave_fast) {
				// No NaN can arise for these rays, so min/max are order-free: v_max3/v_min3.
#pragma unroll
				for (int i = 0; i < 4; i++) {
					const float ax = (nx[i] - ox) * rdx, bx = (fx[i] - ox) * rdx;
					const float ay = (ny[i] - oy) * rdy, by = (fy[i] - oy) * rdy;
					const float az = (nz[i] - oz) * rdz, bz = (fz[i] - oz) * rdz;
					const float tn = fmaxf(fmaxf(fmaxf(ax, ay), az), tmin_ray);
					const float tf = fminf(fminf(fminf(bx, by), bz), best_t);
					RTK_CHILD_HIT(tn <= tf, i);
				}
			} else {
#pragma unroll
				for (int i = 0; i < 4; i++) {
					// rtk.c:458-465: (bound - origin) * rcp_dir, then the folded interval test with
					// _mm_max_ps/_mm_min_ps operand order (decides what a NaN from 0*inf does)
					const float ax = (nx[i] - ox) * rdx, bx = (fx[i] - ox) * rdx;
					const float ay = (ny[i] - oy) * rdy, by = (fy[i] - oy) * rdy;
					const float az = (nz[i] - oz) * rdz, bz = (fz[i] - oz) * rdz;
					const float tn = sse_max(sse_max(ax, ay), sse_max(az, tmin_ray));
					const float tf = sse_min(sse_min(bx, by), sse_min(bz, best_t));
					const bool h = (tn <= tf) && (ref[i] != RTK_REF_NONE);
					key[i] = h ? tn : __builtin_inff();
					nhit += h ? 1u : 0u;
				}
			}
			}
			// nearest first (rtk.c:496-517 orders by entry distance)
			// ... as a 5-comparator network on (distance, reference) PAIRS held as one 64-bit value each, distance in the high
			// half: read as a double such a pair orders like its distance (the bit patterns of non-NaN floats order like their
			// values under a sign-magnitude compare, which is what a double compare of the pair is; +inf, the key of a child that
			// is not entered, becomes a large finite double; equal distances fall back on the reference, any order of which is
			// right), so a comparator is one v_min_f64 and one v_max_f64 instead of a compare and four selects. No key is a NaN.
			double pr[4];
#pragma unroll
			for (int i = 0; i < 4; i++) pr[i] = __hiloint2double((int)__float_as_uint(key[i]), (int)ref[i]);
			cswap_pair(pr[0], pr[1]);
			cswap_pair(pr[2], pr[3]);
			cswap_pair(pr[0], pr[2]);
			cswap_pair(pr[1], pr[3]);
			cswap_pair(pr[1], pr[2]);
			if (nhit == 0u) {
				RTK_POP();
			} else {
				top = (uint32_t)__double2loint(pr[0]);
				const uint32_t np = nhit - 1u;              // sorted slots np..1 go on the stack, far to near
				if (sp + 3u <= LDS_STACK) {
					// Branch-free: rows sp..sp+2 all exist. Slot i <= np goes to row sp+np-i; the others (misses) are
					// written too, to the distinct rows sp+np..sp+2 above the new top, where garbage is harmless.
#pragma unroll
					for (int i = 1; i <= 3; i++) {
						const uint32_t row = sp + np - (uint32_t)i + ((uint32_t)i > np ? 3u : 0u);
						stk[row][lane] = make_uint2((uint32_t)__double2loint(pr[i]), (uint32_t)__double2hiint(pr[i]));
					}
					sp += np;
				} else {
#pragma unroll
					for (int i = 3; i >= 1; i--) {
						if (nhit > (uint32_t)i) {
							const uint2 e = make_uint2((uint32_t)__double2loint(pr[i]), (uint32_t)__double2hiint(pr[i]));
							RTK_PUSH(e);
						}
					}
				}
			}
		}

		// ---------------------------------------------------------------- leaf
		// Triangles are taken one at a time but in the reference's groups of four
		// (rtk.c:212): if any slot of a group -- padding slots of a partial last group
		// included -- has an edge function that is exactly zero, ALL slots of the group
		// use the double-precision edge functions (rtk.c:302-336). A partial group is
		// known up front; a zero inside a full group is rare, so the group is simply
		// redone from a snapshot of the best hit. This keeps t/u/v bit-identical to
		// rtk.c traversing the same leaves.
		if (active && RTK_IS_LEAF(top)) {
			const uint32_t slot0 = top & 0x7fffffffu;
			if (COUNT) c_leaves++;
			uint32_t i = 0, n = 1;
			bool force = false, redo = false;
			// MODE 2 cannot undo list insertions, so a full group is first scanned for exact zeros, then evaluated
			bool scan = false, scanned = false, zero_in_group = false;
			float sn_t = best_t, sn_u = best_u, sn_v = best_v;
			uint32_t sn_prim = best_prim;
			while (i < n) {
				f32x4 A, B, C;
				load_tri(tris, (slot0 + i) * (uint32_t)RTK_TRI_STRIDE, A, B, C);
				if (COUNT && lane == (uint32_t)__ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1u) w_tri_steps++;
				if (i == 0u) n = __float_as_uint(C.w);          // leaf size rides in the first record
				if (MODE == 2 && (i & 3u) == 0u) {
					if (scanned) scanned = false;                      // second pass over the group: `force` is decided
					else { force = (n - i) < 4u; scan = !force; zero_in_group = false; }
				} else if ((i & 3u) == 0u) {
					if (redo) { force = true; redo = false; }
					else {
						if (MODE == 1 && best_prim != RTK_PRIM_NONE) break;   // any-hit: a whole group accepted something
						force = (n - i) < 4u;
						sn_t = best_t; sn_u = best_u; sn_v = best_v; sn_prim = best_prim;
					}
				}
				if (COUNT) c_tris++;
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
				if (MODE == 2 && scan) {
					u = x1 * y2 - y1 * x2;
					v = x2 * y0 - y2 * x0;
					w = x0 * y1 - y0 * x1;
					zero_in_group = zero_in_group || u == 0.0f || v == 0.0f || w == 0.0f;
					if ((i & 3u) == 3u) { scan = false; scanned = true; force = zero_in_group; i &= ~3u; }   // rtk.c:306
					else i++;
					continue;
				}
				if (!force) {
					u = x1 * y2 - y1 * x2;
					v = x2 * y0 - y2 * x0;
					w = x0 * y1 - y0 * x1;
					if (MODE != 2 && (u == 0.0f || v == 0.0f || w == 0.0f)) {
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
				bool in_range = !(neg && pos) && t > tmin_ray && t < tmax_ray;   // rtk.c:354
				if (FILT) {
					// built-in filters: the candidate is offered only if every filter accepts it (rtk.h:117 semantics
					// with the filter evaluated on the device)
					if (has_after) in_range = in_range && (t > after_t || (t == after_t && prim > after_prim));
					in_range = in_range && prim != skip_prim;
					if (p.mesh_mask) {
						const uint32_t mesh = __float_as_uint(B.w) >> RTK_TRI_MESH_SHIFT;
						in_range = in_range && mesh < p.mesh_mask_bits && ((p.mesh_mask[mesh >> 5] >> (mesh & 31u)) & 1u);
					}
				}
				if (MODE == 2) {
					// keep the k closest candidates of this ray, sorted by (t, prim), in the ray's slice of p.cand; once the
					// list is full, best_t (the culling distance) is the last entry's t
					rtk_hit_record *list = p.cand + (size_t)ray_index * p.cand_k;
					if (in_range && cand_n == p.cand_k) {
						const rtk_hit_record last = list[p.cand_k - 1u];
						in_range = t < last.t || (t == last.t && prim < last.prim);
					}
					if (in_range) {
						uint32_t j = cand_n < p.cand_k ? cand_n : p.cand_k - 1u;
						while (j > 0u) {
							const rtk_hit_record e = list[j - 1u];
							if (e.t < t || (e.t == t && e.prim < prim)) break;
							list[j] = e;
							j--;
						}
						rtk_hit_record r;
						r.t = t; r.u = u * rcp; r.v = v * rcp; r.prim = prim;
						list[j] = r;
						if (cand_n < p.cand_k) cand_n++;
						if (cand_n == p.cand_k) best_t = list[p.cand_k - 1u].t;
					}
				} else if (MODE == 1) {
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
				top = RTK_REF_NONE;
				sp = 0u;
			} else {
				RTK_POP();
			}
		}

		// ---------------------------------------------------------------- retire
		if (active && top == RTK_REF_NONE) {
			if (MODE == 2) {
				p.cand_count[ray_index] = cand_n;
			} else if (MODE == 1) {
				p.occluded[ray_index] = best_prim != RTK_PRIM_NONE ? 1 : 0;
			} else {
				rtk_hit_record r;
				r.t = best_t; r.u = best_u; r.v = best_v; r.prim = best_prim;
				st_f4_stream(p.hits + ray_index, r.t, r.u, r.v, __uint_as_float(r.prim));
			}
			if (COUNT) {
				atomicAdd(p.counter + 1, 1ull);
				atomicAdd(p.counter + 2, (unsigned long long)c_nodes);
				atomicAdd(p.counter + 3, (unsigned long long)c_leaves);
				atomicAdd(p.counter + 4, (unsigned long long)c_tris);
				atomicAdd(p.counter + 5, best_prim != RTK_PRIM_NONE ? 1ull : 0ull);
				atomicAdd(p.counter + 6, (unsigned long long)c_spills);
				c_nodes = c_leaves = c_tris = c_spills = 0;
			}
			active = false;
		}
	}
	if (COUNT) {
		// wave-level trip counts: node steps are counted on a wave-uniform variable, triangle steps
		// on whichever lane was first in the leaf loop; rays/64 gives steps per 64 rays
		if (lane == 0) atomicAdd(p.counter + 7, w_node_steps);
		atomicAdd(p.counter + 8, w_tri_steps);
	}
}

// ---- ray reordering (RTK_TRACE_SORT_RAYS): 16-bit key = 4 bits of origin cell per axis inside the
// batch's origin bounds + direction octant; rays of one key start close together and head the same way,
// so the 64 rays a wave pulls from the sorted order share nodes (L1/L2 hits instead of fabric traffic).
__device__ __forceinline__ uint32_t f2ord_(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ord2f_(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// Origin bounds from every `stride`-th ray: the cells only have to spread the batch over the key range, outliers are
// clamped into the border cells by the key kernel.
__global__ void rtk_ray_bounds_kernel(const rtk_ray *rays, unsigned long long n, unsigned long long stride, uint32_t *bounds)
{
	__shared__ float s_mn[3][4], s_mx[3][4];
	float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k * stride < n; k += (unsigned long long)gridDim.x * blockDim.x) {
		const unsigned long long i = k * stride;
		const float4 r0 = *reinterpret_cast<const float4 *>(rays + i);
		const float o[3] = { r0.x, r0.y, r0.z };
		for (int a = 0; a < 3; a++) if (isfinite(o[a])) { mn[a] = fminf(mn[a], o[a]); mx[a] = fmaxf(mx[a], o[a]); }
	}
	for (int a = 0; a < 3; a++) {
		for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o)); }
		if ((threadIdx.x & 63u) == 0) { s_mn[a][threadIdx.x >> 6] = mn[a]; s_mx[a][threadIdx.x >> 6] = mx[a]; }
	}
	__syncthreads();
	if (threadIdx.x < 3) {
		const int a = threadIdx.x;
		float lo = s_mn[a][0], hi = s_mx[a][0];
		for (int w = 1; w < 4; w++) { lo = fminf(lo, s_mn[a][w]); hi = fmaxf(hi, s_mx[a][w]); }
		atomicMin(&bounds[a], f2ord_(lo));
		atomicMax(&bounds[3 + a], f2ord_(hi));
	}
}

__device__ __forceinline__ uint32_t morton_cells_(const uint32_t q[3], uint32_t cell_bits)
{
	// Morton-interleave the cell coordinates (x lowest) so that consecutive keys are neighbours in space
	uint32_t key = 0;
	for (uint32_t b = 0; b < cell_bits; b++)
		key |= (((q[0] >> b) & 1u) << (3u * b)) | (((q[1] >> b) & 1u) << (3u * b + 1u)) | (((q[2] >> b) & 1u) << (3u * b + 2u));
	return key;
}

__device__ __forceinline__ uint32_t cell_of_(float x, float lo, float hi, uint32_t cells)
{
	const float ext = hi - lo;
	float t = ext > 0.0f ? (x - lo) / ext : 0.0f;
	t = t >= 0.0f ? (t <= 1.0f ? t : 1.0f) : 0.0f;           // NaN -> 0
	const uint32_t c = (uint32_t)(t * (float)cells);
	return c > cells - 1u ? cells - 1u : c;
}

// Key = cell of the ray's origin inside the batch's (sampled) origin bounds.
__global__ void rtk_ray_keys_kernel(const rtk_ray *rays, uint32_t n, const uint32_t *bounds, unsigned long long *keys,
	uint32_t cell_bits, uint32_t with_octant)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float4 r0 = *reinterpret_cast<const float4 *>(rays + i);
	const float o[3] = { r0.x, r0.y, r0.z };
	uint32_t q[3];
	for (int a = 0; a < 3; a++) q[a] = cell_of_(o[a], ord2f_(bounds[a]), ord2f_(bounds[3 + a]), 1u << cell_bits);
	uint32_t key = morton_cells_(q, cell_bits);
	if (with_octant) {
		const float4 r1 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(rays + i) + 16);
		key = (key << 3) | ((__float_as_uint(r0.w) >> 31) | ((__float_as_uint(r1.x) >> 31) << 1) | ((__float_as_uint(r1.y) >> 31) << 2));
	}
	keys[i] = ((unsigned long long)key << 32) | i;      // sorted by the key, the ray's number rides below it
}

// Key = cell, inside the SCENE's bounds (union of the root's child boxes), of the point where the ray's [min_t, max_t]
// interval enters those bounds: the origin itself for rays that start inside (shadow / bounce rays), the entry point for
// rays that start outside (camera rays, config 3). That is where traversal starts doing work, so rays of one key share
// the nodes and leaves they touch. Rays that miss the bounds get the largest key: they end at the root, together.
__global__ void rtk_ray_entry_keys_kernel(const rtk_ray *rays, uint32_t n, const DevNode *root, unsigned long long *keys,
	uint32_t cell_bits, uint32_t with_octant)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (int k = 0; k < 4; k++) {
		if (root->child[k] == RTK_REF_NONE) continue;
		lo[0] = fminf(lo[0], root->bx[0][k]); hi[0] = fmaxf(hi[0], root->bx[1][k]);
		lo[1] = fminf(lo[1], root->by[0][k]); hi[1] = fmaxf(hi[1], root->by[1][k]);
		lo[2] = fminf(lo[2], root->bz[0][k]); hi[2] = fmaxf(hi[2], root->bz[1][k]);
	}
	const float4 r0 = *reinterpret_cast<const float4 *>(rays + i);
	const float4 r1 = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(rays + i) + 16);
	const float o[3] = { r0.x, r0.y, r0.z }, d[3] = { r0.w, r1.x, r1.y };
	float tn = r1.z, tf = r1.w;
	bool miss = false;
	for (int a = 0; a < 3; a++) {
		if (d[a] != 0.0f) {
			const float t0 = (lo[a] - o[a]) / d[a], t1 = (hi[a] - o[a]) / d[a];
			tn = fmaxf(tn, fminf(t0, t1));
			tf = fminf(tf, fmaxf(t0, t1));
		} else if (!(o[a] >= lo[a] && o[a] <= hi[a])) miss = true;
	}
	miss = miss || !(tn <= tf);                                   // NaN anywhere -> miss
	const uint32_t key_bits = 3u * cell_bits + (with_octant ? 3u : 0u);
	uint32_t key = (1u << key_bits) - 1u;
	if (!miss) {
		uint32_t q[3];
		for (int a = 0; a < 3; a++) q[a] = cell_of_(o[a] + d[a] * tn, lo[a], hi[a], 1u << cell_bits);
		key = morton_cells_(q, cell_bits);
		if (with_octant) key = (key << 3) | ((__float_as_uint(d[0]) >> 31) | ((__float_as_uint(d[1]) >> 31) << 1) | ((__float_as_uint(d[2]) >> 31) << 2));
	}
	keys[i] = ((unsigned long long)key << 32) | i;      // sorted by the key, the ray's number rides below it
}

// Full rtk_hit from a compact record (rtk.c:372-380 copy-out).
__device__ __forceinline__ void rtk_expand_one(const DevSceneView &sc, const rtk_hit_record *rec, unsigned long long i, rtk_hit *hits, uint8_t *mask)
{
	const rtk_hit_record r = rec[i];
	const bool hit = r.prim != RTK_PRIM_NONE && r.prim < sc.num_prims;
	if (mask) mask[i] = hit ? 1 : 0;
	if (!hit || !hits) return;
	const uint32_t slot = sc.prim_slot[r.prim];
	const DevTri tr = sc.tris[slot];
	rtk_hit h;
	h.t = r.t; h.u = r.u; h.v = r.v;
	h.vertex[0].position.x = tr.v0[0]; h.vertex[0].position.y = tr.v0[1]; h.vertex[0].position.z = tr.v0[2];
	h.vertex[1].position.x = tr.v1[0]; h.vertex[1].position.y = tr.v1[1]; h.vertex[1].position.z = tr.v1[2];
	h.vertex[2].position.x = tr.v2[0]; h.vertex[2].position.y = tr.v2[1]; h.vertex[2].position.z = tr.v2[2];
	h.vertex[0].index = sc.vertex_index[3u * slot + 0u];
	h.vertex[1].index = sc.vertex_index[3u * slot + 1u];
	h.vertex[2].index = sc.vertex_index[3u * slot + 2u];
	h.mesh_index = sc.slot_mesh[slot];
	h.triangle_index = sc.slot_tri[slot];
	hits[i] = h;
}

// status_out (host-visible memory): the launch-error word of this stream's trace launches is copied there as well, so that a
// host call needs no transfer of its own to read it (rtk_trace_ray: one copy and ~10 us less per call). A one-workgroup
// launch with ticket != 0 writes (ticket << 32 | error) there AFTER all its results: the host may poll for the ticket
// instead of waiting for the stream.
__global__ void rtk_expand_kernel(DevSceneView sc, const rtk_hit_record *rec, unsigned long long n, rtk_hit *hits, uint8_t *mask,
	const unsigned long long *status_word, unsigned long long *status_out, uint32_t ticket)
{
	const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (status_out && ticket == 0u && i == 0) *status_out = *status_word;
	if (i < n) rtk_expand_one(sc, rec, i, hits, mask);
	if (status_out && ticket != 0u) {
		__threadfence_system();
		__syncthreads();
		if (threadIdx.x == 0) {
			__hip_atomic_store(status_out, ((unsigned long long)ticket << 32) | (*status_word ? 1ull : 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}


// ---- rtk_trace_ray: ONE ray, one wave (reference rtk.h:129, rtk.c:543-577) --------------------------------------------------
// A per-ray call is a chain of dependent fetches: rtk_trace_kernel walks ~20 nodes and leaves one after the other for a single
// ray, ~1 us each from a cold start. Here the wave's 64 lanes walk the ray's FRONTIER breadth first: every node of a level the
// ray enters is fetched and tested at once (one lane each), the leaves found on the way are tested (one lane each) before the
// next level so that the hit culls what is behind it, and the best candidate is agreed on by the wave with the canonical tie
// rule. The chain is then as long as the tree is deep along the ray, not as long as the list of nodes visited. Exact 128-byte
// nodes and the reference's slab arithmetic with its SSE operand order (rtk.c:458-470) for every ray, the triangle groups of
// rtk_trace_kernel (rtk.c:212-386): the result is what the exact path of rtk_trace_kernel returns. The same wave expands the
// hit into the caller-visible rtk_hit and signs off with the ticket (one launch per call instead of two). A frontier that does
// not fit LDS (a degenerate scene: thousands of boxes on one ray) is reported as "not done" and the host takes the batch path.
#define ONE_FRONTIER 512
#define RTK_ONE_NOT_DONE 2ull

__device__ __forceinline__ void one_leaf(const char *tris, uint32_t slot0, bool kz0, bool kz1, float sox, float soy, float soz, float shx, float shy,
	float shz, float tmin_ray, float tmax_ray, float &best_t, float &best_u, float &best_v, uint32_t &best_prim)
{
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
			else { force = (n - i) < 4u; sn_t = best_t; sn_u = best_u; sn_v = best_v; sn_prim = best_prim; }
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
		const bool neg = sse_min(sse_min(u, v), w) < 0.0f;          // rtk.c:340-342
		const bool pos = sse_max(sse_max(u, v), w) > 0.0f;
		const float det = (u + v) + w;                              // rtk.c:346-353
		const float rcp = 1.0f / det;
		float zz = u * z0;
		zz = zz + v * z1;
		zz = zz + w * z2;
		const float t = zz * rcp;
		const uint32_t prim = __float_as_uint(A.w);
		const bool in_range = !(neg && pos) && t > tmin_ray && t < tmax_ray;   // rtk.c:354
		// rtk.c:371 with the canonical tie rule: lowest primitive id among bit-equal t
		if (in_range && (t < best_t || (t == best_t && prim < best_prim))) { best_t = t; best_u = u * rcp; best_v = v * rcp; best_prim = prim; }
		i++;
	}
}

__global__ void __launch_bounds__(64) rtk_trace_one_kernel(DevSceneView sc, rtk_ray ray_in, rtk_hit *hit_out, uint8_t *mask_out,
	unsigned long long *status_out, uint32_t ticket)
{
	__shared__ uint32_t s_front[2][ONE_FRONTIER];
	__shared__ uint32_t s_leaf[2][ONE_FRONTIER];
	const uint32_t lane = threadIdx.x;
	const char *const nodes = reinterpret_cast<const char *>(sc.nodes);
	const char *const tris = reinterpret_cast<const char *>(sc.tris);
	// (the ray travels in the kernel argument: a load from the host's pinned memory would be a PCIe round trip of ~2 us)
	const float ox = ray_in.origin.x, oy = ray_in.origin.y, oz = ray_in.origin.z, dx = ray_in.direction.x, dy = ray_in.direction.y, dz = ray_in.direction.z,
		tmin_ray = ray_in.min_t, tmax_ray = ray_in.max_t;
	// rtk.c:550-566 (as rtk_trace_kernel)
	const float ax_ = fabsf(dx), ay_ = fabsf(dy), az_ = fabsf(dz);
	const float m = sse_max(sse_max(ax_, ay_), az_);
	const bool kz0 = ax_ == m, kz1 = !kz0 && ay_ == m;
	const float dkx = kz0 ? dy : (kz1 ? dz : dx), dky = kz0 ? dz : (kz1 ? dx : dy), dkz = kz0 ? dx : (kz1 ? dy : dz);
	const float shx = -dkx / dkz, shy = -dky / dkz;
	const float sox = kz0 ? oy : (kz1 ? oz : ox), soy = kz0 ? oz : (kz1 ? ox : oy), soz = kz0 ? ox : (kz1 ? oy : oz);
	const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;      // rtk.c:410: true divides
	const float shz = kz0 ? rdx : (kz1 ? rdy : rdz);
	const uint32_t onx = (__float_as_uint(dx) >> 31) * 16u, ony = 32u + (__float_as_uint(dy) >> 31) * 16u, onz = 64u + (__float_as_uint(dz) >> 31) * 16u;
	float best_t = tmax_ray, best_u = 0.0f, best_v = 0.0f;
	uint32_t best_prim = RTK_PRIM_NONE;
	uint32_t n_cur = sc.num_nodes ? 1u : 0u, cur = 0u, n_leaf = 0u;
	bool not_done = false;
	if (lane == 0) s_front[0][0] = 0u;
	__syncthreads();
	// One round = one memory round trip: the nodes of the current level AND the leaves the previous level found are fetched and
	// tested together (leaves one lane each, then nodes one lane each); what a leaf's hit culls it culls one level later.
	while ((n_cur != 0u || n_leaf != 0u) && !not_done) {
		uint32_t n_next = 0u, n_leaf_next = 0u;
		const bool had_leaves = n_leaf != 0u;
		for (uint32_t base = 0; base < n_leaf; base += 64u)
			if (base + lane < n_leaf) one_leaf(tris, s_leaf[cur][base + lane], kz0, kz1, sox, soy, soz, shx, shy, shz, tmin_ray, tmax_ray, best_t, best_u, best_v, best_prim);
		for (uint32_t base = 0; base < n_cur; base += 64u) {
			const bool have = base + lane < n_cur;
			const uint32_t a_node = (have ? s_front[cur][base + lane] : 0u) << 7;
			f32x4 nx, fx, ny, fy, nz, fz;
			u32x4 ch;
			load_node(nodes, a_node + onx, (a_node + 16u) - onx, a_node + ony, (a_node + 80u) - ony, a_node + onz, (a_node + 144u) - onz, a_node,
				nx, fx, ny, fy, nz, fz, ch);
			const uint32_t ref[4] = { ch.x, ch.y, ch.z, ch.w };
			const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				// rtk.c:458-465: (bound - origin) * rcp_dir, the folded interval test with _mm_max_ps / _mm_min_ps operand order
				// (best_t here is the lane's own: at least as far as the wave's -- the test only gets more conservative)
				const float ax = (nx[i] - ox) * rdx, bx = (fx[i] - ox) * rdx;
				const float ay = (ny[i] - oy) * rdy, by = (fy[i] - oy) * rdy;
				const float az = (nz[i] - oz) * rdz, bz = (fz[i] - oz) * rdz;
				const float tn = sse_max(sse_max(ax, ay), sse_max(az, tmin_ray));
				const float tf = sse_min(sse_min(bx, by), sse_min(bz, best_t));
				const bool h = have && (tn <= tf) && ref[i] != RTK_REF_NONE;
				const bool leaf = (ref[i] & RTK_REF_LEAF) != 0u;
				const unsigned long long m_leaf = __builtin_amdgcn_ballot_w64(h && leaf), m_node = __builtin_amdgcn_ballot_w64(h && !leaf);
				if (h && leaf) { const uint32_t at = n_leaf_next + (uint32_t)__popcll(m_leaf & below); if (at < ONE_FRONTIER) s_leaf[cur ^ 1u][at] = ref[i] & 0x7fffffffu; }
				if (h && !leaf) { const uint32_t at = n_next + (uint32_t)__popcll(m_node & below); if (at < ONE_FRONTIER) s_front[cur ^ 1u][at] = ref[i]; }
				n_leaf_next += (uint32_t)__popcll(m_leaf);
				n_next += (uint32_t)__popcll(m_node);
			}
		}
		if (n_leaf_next > ONE_FRONTIER || n_next > ONE_FRONTIER) { not_done = true; break; }
		// the wave agrees on the best candidate (lowest t, lowest primitive id among equals)
		if (had_leaves) {
			float t_min = best_t;
			for (int o = 32; o > 0; o >>= 1) t_min = fminf(t_min, __shfl_xor(t_min, o));
			uint32_t p_min = best_t == t_min ? best_prim : RTK_PRIM_NONE;
			for (int o = 32; o > 0; o >>= 1) { const uint32_t q = (uint32_t)__shfl_xor((int)p_min, o); p_min = q < p_min ? q : p_min; }
			const unsigned long long owner = __builtin_amdgcn_ballot_w64(best_t == t_min && best_prim == p_min);
			const int src = owner ? (int)__builtin_ctzll(owner) : 0;
			best_u = __shfl(best_u, src);
			best_v = __shfl(best_v, src);
			best_t = t_min;
			best_prim = p_min;
		}
		cur ^= 1u;
		n_cur = n_next;
		n_leaf = n_leaf_next;
		__syncthreads();
	}
	if (lane == 0) {
		if (!not_done) {
			const bool hit = best_prim != RTK_PRIM_NONE && best_prim < sc.num_prims;
			*mask_out = hit ? 1 : 0;
			if (hit) {
				const uint32_t slot = sc.prim_slot[best_prim];
				const DevTri tr = sc.tris[slot];
				rtk_hit h;
				h.t = best_t; h.u = best_u; h.v = best_v;
				h.vertex[0].position.x = tr.v0[0]; h.vertex[0].position.y = tr.v0[1]; h.vertex[0].position.z = tr.v0[2];
				h.vertex[1].position.x = tr.v1[0]; h.vertex[1].position.y = tr.v1[1]; h.vertex[1].position.z = tr.v1[2];
				h.vertex[2].position.x = tr.v2[0]; h.vertex[2].position.y = tr.v2[1]; h.vertex[2].position.z = tr.v2[2];
				h.vertex[0].index = sc.vertex_index[3u * slot + 0u];
				h.vertex[1].index = sc.vertex_index[3u * slot + 1u];
				h.vertex[2].index = sc.vertex_index[3u * slot + 2u];
				h.mesh_index = sc.slot_mesh[slot];
				h.triangle_index = sc.slot_tri[slot];
				*hit_out = h;
			}
		}
		__threadfence_system();
		__hip_atomic_store(status_out, ((unsigned long long)ticket << 32) | (not_done ? RTK_ONE_NOT_DONE : 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------

namespace {

typedef void (*trace_kernel_fn)(TraceParams);

// variant index: any_hit | counted << 1 | filtered << 2 | compressed nodes << 3
trace_kernel_fn trace_variant(int v)
{
	switch (v) {
	case 0: return rtk_trace_kernel<0, false, false, false>;
	case 1: return rtk_trace_kernel<1, false, false, false>;
	case 2: return rtk_trace_kernel<0, true, false, false>;
	case 3: return rtk_trace_kernel<1, true, false, false>;
	case 4: return rtk_trace_kernel<0, false, true, false>;
	case 5: return rtk_trace_kernel<1, false, true, false>;
	case 6: return rtk_trace_kernel<0, true, true, false>;
	case 7: return rtk_trace_kernel<1, true, true, false>;
	case 8: return rtk_trace_kernel<0, false, false, true>;
	case 9: return rtk_trace_kernel<1, false, false, true>;
	case 10: return rtk_trace_kernel<0, true, false, true>;
	case 11: return rtk_trace_kernel<1, true, false, true>;
	case 12: return rtk_trace_kernel<0, false, true, true>;
	case 13: return rtk_trace_kernel<1, false, true, true>;
	case 14: return rtk_trace_kernel<0, true, true, true>;
	case 15: return rtk_trace_kernel<1, true, true, true>;
	case 16: return rtk_trace_kernel<2, false, true, false>;      // collect the k closest candidates (host-callback filters)
	default: return rtk_trace_kernel<2, false, true, true>;
	}
}
enum { VARIANT_COLLECT = 16, VARIANT_PACKET = 18, VARIANT_PACKET_COUNTED = 19, NUM_VARIANTS = 20 };

// resident workgroups per CU of each kernel variant, per device; filled on first use
std::mutex g_occ_mutex;
int g_occ[RTK_MAX_DEVICES][NUM_VARIANTS];

int blocks_per_cu_of(int device, int variant)
{
	std::lock_guard<std::mutex> lock(g_occ_mutex);
	if (device < 0 || device >= RTK_MAX_DEVICES) device = 0;
	int &o = g_occ[device][variant];
	if (o == 0) {
		int nb = 0;
		if (variant >= VARIANT_PACKET) nb = rtk_packet_occupancy(variant == VARIANT_PACKET_COUNTED);
		else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_variant(variant), BLOCK_THREADS, 0) != hipSuccess) nb = 0;
		o = nb >= 1 ? nb : 1;
	}
	return o;
}

// The scratch set of (scene, stream). Called with ds->scratch_mutex held.
// ---- is the batch a row-major image nobody told us about? (the reference's interface has no notion of an image, rtk.h:129; a host
// that only hands over rays should still get the packet kernels: VERDICT round 4, item 5)
// Along a row of an image the step from one ray to the next (origin and direction, six numbers) changes slowly; from the last ray
// of a row to the first of the next it jumps by about a row's width. k_detect_row finds the first such jump among the first
// 2^17 rays: the candidate width. k_detect_check then looks at up to 256 row ends (there must be a jump at every one) and at
// places inside rows (there must be none). A wrong guess can only cost speed -- tiles whose rays do not form a beam are handed
// back by the packet kernels, records are the same on every path -- so this is a heuristic with a cheap test, not a proof.
#define RTK_DETECT_WORDS 4
#define RTK_DETECT_WORD (RTK_ERROR_WORD + 1)
__device__ __forceinline__ bool ray_step_jumps(const rtk_ray *rays, size_t i)
{
	// rays i-1, i, i+1: does the step i -> i+1 differ from the step i-1 -> i by more than eight times the latter?
	const float *a = reinterpret_cast<const float *>(rays + i - 1), *b = reinterpret_cast<const float *>(rays + i), *c = reinterpret_cast<const float *>(rays + i + 1);
	float m = 0.0f, dmax = 0.0f;
#pragma unroll
	for (int k = 0; k < 6; k++) {
		const float s0 = b[k] - a[k], s1 = c[k] - b[k];
		m = fmaxf(m, fabsf(s0));
		dmax = fmaxf(dmax, fabsf(s1 - s0));
	}
	return !(dmax <= 8.0f * m);          // (NaN anywhere: a jump)
}

__global__ void k_detect_row(const rtk_ray *rays, uint32_t limit, uint32_t *first_jump)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1u;
	if (i + 1u >= limit) return;
	if (ray_step_jumps(rays, i)) atomicMin(first_jump, i);
}

// word[0] = first jump (set by k_detect_row). One workgroup: up to 256 rows' ends must jump, places inside those rows must not;
// the verdict goes straight into host-visible memory (verdict[0] = width or 0, verdict[1] = height): no copy behind the kernel.
__global__ void __launch_bounds__(256) k_detect_check(const rtk_ray *rays, unsigned long long n, const uint32_t *word, uint32_t *verdict)
{
	__shared__ uint32_t s_bad;
	if (threadIdx.x == 0) s_bad = 0u;
	__syncthreads();
	const uint32_t w = word[0] + 1u;                                      // candidate width
	const bool candidate = word[0] != 0xffffffffu && w >= 64u && (n % w) == 0ull && n / w >= 2ull && n / w <= 0xffffffffull;
	if (candidate) {
		const unsigned long long rows = n / w;
		const unsigned long long stride = rows > 256ull ? rows / 256ull : 1ull;
		const unsigned long long r = (unsigned long long)threadIdx.x * stride;
		if (r + 1ull < rows) {
			const size_t end = (size_t)((r + 1ull) * w - 1ull);
			bool bad = !ray_step_jumps(rays, end);
			for (uint32_t q = 1; q < 4u; q++) bad = bad || ray_step_jumps(rays, (size_t)(r * w) + (size_t)q * (w / 4u));
			if (bad) atomicAdd(&s_bad, 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const bool ok = candidate && s_bad == 0u;
		verdict[0] = ok ? w : 0u;
		verdict[1] = ok ? (uint32_t)(n / w) : 0u;
		__threadfence_system();
	}
}

LaunchScratch *scratch_for(rtk_dev_scene *ds, hipStream_t stream)
{
	for (LaunchScratch *s : ds->scratch) if (s->stream == stream) return s;
	LaunchScratch *s = new LaunchScratch();
	s->stream = stream;
	// (cleared ON THE LAUNCH STREAM: a hipMemset goes to the NULL stream, which non-blocking streams do not wait for -- behind
	// another thread's device build there it ran milliseconds late, and a first launch read a stale error word: an intermittent
	// "traversal stack overflow" in test_builds_and_traces_from_several_threads_at_once)
	if (hipMalloc(&s->d_counter, (RTK_COUNTER_WORDS + 1 + RTK_DETECT_WORDS) * sizeof(unsigned long long)) != hipSuccess ||
		hipMemsetAsync(s->d_counter, 0, (RTK_COUNTER_WORDS + 1 + RTK_DETECT_WORDS) * sizeof(unsigned long long), stream) != hipSuccess) {
		rtk_set_error("rtk_dev_trace: out of device memory (launch scratch)");
		delete s;
		return nullptr;
	}
	ds->scratch.push_back(s);
	return s;
}

} // namespace

// ---- the hand-written per-lane kernels: a code object of its own (rtk_lane_hot.S, assembled by the Makefile), carried in
// this library as a byte array and loaded once per device
#include "rtk_lane_hot_image.h"

namespace {
struct LaneModule { hipModule_t mod = nullptr; hipFunction_t fn[2] = { nullptr, nullptr }; int blocks_per_cu = 0; bool tried = false; };
std::mutex g_lane_mutex;
LaneModule g_lane[RTK_MAX_DEVICES];

LaneModule *lane_module(int device)
{
	if (device < 0 || device >= RTK_MAX_DEVICES) return nullptr;
	std::lock_guard<std::mutex> lock(g_lane_mutex);
	LaneModule &h = g_lane[device];
	if (!h.tried) {
		int cur = -1;
		if (hipGetDevice(&cur) != hipSuccess || cur != device) return nullptr;      // loaded by a thread that has this device current (asked again later)
		h.tried = true;
		if (hipModuleLoadData(&h.mod, rtk_lane_hot_image) != hipSuccess ||
			hipModuleGetFunction(&h.fn[0], h.mod, "rtk_lane_hot_closest") != hipSuccess ||
			hipModuleGetFunction(&h.fn[1], h.mod, "rtk_lane_hot_any") != hipSuccess) {
			(void)hipGetLastError();
			h.fn[0] = h.fn[1] = nullptr;
		} else {
			// 80 VGPRs, 30 KB of LDS per workgroup: five workgroups per CU
			int nb = 0;
			if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, h.fn[0], BLOCK_THREADS, 0) != hipSuccess || nb < 1) nb = 1;
			static const int cap = getenv("RTK_AMD_LANE_BLOCKS") ? atoi(getenv("RTK_AMD_LANE_BLOCKS")) : 5;
			h.blocks_per_cu = nb > cap ? cap : nb;
		}
	}
	return h.fn[0] ? &h : nullptr;
}
} // namespace

bool rtk_lane_hot_available(int device, int *blocks_per_cu)
{
	LaneModule *h = lane_module(device);
	if (!h) return false;
	if (blocks_per_cu) *blocks_per_cu = h->blocks_per_cu;
	return true;
}

int rtk_lane_hot_launch(int device, const LnHotParams &hp_in, unsigned blocks, hipStream_t stream, bool any_hit)
{
	LaneModule *h = lane_module(device);
	if (!h) { rtk_set_error("rtk_dev_trace: the assembly per-lane kernels are not loaded"); return RTK_AMD_ERR_HIP; }
	LnHotParams hp = hp_in;
	size_t size = sizeof(hp);
	void *config[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &hp, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
	RTK_HIP_CHECK(hipModuleLaunchKernel(h->fn[any_hit ? 1 : 0], blocks, 1, 1, BLOCK_THREADS, 1, 1, 0, stream, nullptr, config), RTK_AMD_ERR_HIP);
	return RTK_AMD_OK;
}

void rtk_scratch_free(LaunchScratch *s)
{
	if (!s) return;
	if (s->d_counter) (void)hipFree(s->d_counter);
	if (s->h_verdict) (void)hipHostFree((void *)s->h_verdict);
	if (s->d_spill) (void)hipFree(s->d_spill);
	if (s->d_sort) (void)hipFree(s->d_sort);
	if (s->d_leftover) (void)hipFree(s->d_leftover);
	if (s->d_entries) (void)hipFree(s->d_entries);
	delete s;
}

// *w, *h = the image the batch is (row-major, w * h = n), or 0, 0. Two small launches and a wait for `stream`.
int rtk_detect_image(const rtk_dev_scene *ds_c, const rtk_ray *d_rays, size_t n, hipStream_t stream, uint32_t *w_out, uint32_t *h_out)
{
	rtk_dev_scene *ds = const_cast<rtk_dev_scene *>(ds_c);
	*w_out = *h_out = 0u;
	if (!ds || !d_rays || n < 4u || n > 0x40000000ull) return RTK_AMD_OK;
	// (the look uses two words of the (scene, stream) scratch set and its pinned verdict: the scene's scratch mutex is held until the
	// verdict has been read, so that two host threads feeding one stream cannot interleave their looks; ~30 us)
	std::lock_guard<std::mutex> lock(ds->scratch_mutex);
	LaunchScratch *sc0 = scratch_for(ds, stream);
	if (!sc0) return RTK_AMD_ERR_OOM;
	uint32_t *d_word = reinterpret_cast<uint32_t *>(sc0->d_counter + RTK_DETECT_WORD);
	if (!sc0->h_verdict) RTK_HIP_CHECK(hipHostMalloc((void **)&sc0->h_verdict, 64, hipHostMallocDefault), RTK_AMD_ERR_OOM);     // (pinned: the kernel writes the verdict there)
	volatile uint32_t *h_verdict = sc0->h_verdict;
	const uint32_t limit = (uint32_t)std::min<size_t>(n, (size_t)1 << 17);
	RTK_HIP_CHECK(hipMemsetAsync(d_word, 0xff, 4, stream), RTK_AMD_ERR_HIP);
	hipLaunchKernelGGL(k_detect_row, dim3((limit + 255u) / 256u), dim3(256), 0, stream, d_rays, limit, d_word);
	hipLaunchKernelGGL(k_detect_check, dim3(1), dim3(256), 0, stream, d_rays, (unsigned long long)n, d_word, const_cast<uint32_t *>(h_verdict));
	RTK_HIP_CHECK(hipGetLastError(), RTK_AMD_ERR_HIP);
	RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP);
	*w_out = h_verdict[0];
	*h_out = h_verdict[1];
	return RTK_AMD_OK;
}

int rtk_launch_trace(const rtk_dev_scene *ds_c, const rtk_ray *d_rays, size_t n, rtk_hit_record *d_hits,
	uint8_t *d_occluded, const rtk_trace_opts *opts, hipStream_t stream, bool any_hit, rtk_trace_counters *counted,
	const rtk_dev_filter *filter, rtk_hit_record *d_cand, uint32_t *d_cand_count, uint32_t cand_k, rtk_packet_counters *pk_counted)
{
	rtk_dev_scene *ds = const_cast<rtk_dev_scene *>(ds_c);
	const bool collect = d_cand != nullptr;
	if (pk_counted && (counted || any_hit || filter || collect)) { rtk_set_error("rtk_dev_trace_rays_packet_counted: closest-hit batches only"); return RTK_AMD_ERR_BAD_ARG; }
	if (pk_counted) *pk_counted = rtk_packet_counters();
	if (collect && (!d_cand_count || cand_k == 0 || any_hit || counted)) { rtk_set_error("rtk_dev_trace: bad collect arguments"); return RTK_AMD_ERR_BAD_ARG; }
	if (!ds || (!d_rays && n) || (!collect && (any_hit ? !d_occluded : !d_hits) && n)) { rtk_set_error("rtk_dev_trace: bad argument"); return RTK_AMD_ERR_BAD_ARG; }
	if (n == 0) { if (counted) *counted = rtk_trace_counters(); return RTK_AMD_OK; }
	{
		// the scene's memory, its scratch and `stream` must all belong to the device this thread has current: a launch from a
		// thread on another GPU would read the scene across devices (a fault without peer access)
		int cur = -1;
		if (hipGetDevice(&cur) != hipSuccess || cur != ds->device) {
			rtk_set_error("rtk_dev_trace: the scene lives on device %d, the calling thread's current device is %d", ds->device, cur);
			return RTK_AMD_ERR_BAD_ARG;
		}
	}
	// the kernel addresses nodes and triangles as SGPR base + 32-bit byte offset
	if ((uint64_t)ds->view.num_nodes * 128u > 0xffffff00ull || (uint64_t)ds->view.num_tris * RTK_TRI_STRIDE > 0xffffff00ull) {
		rtk_set_error("rtk_dev_trace: scene exceeds 4 GiB of nodes or triangles (%u nodes, %u triangles)", ds->view.num_nodes, ds->view.num_tris);
		return RTK_AMD_ERR_UNSUPPORTED;
	}

	TraceParams p = {};
	p.sc = ds->view;
	p.rays = d_rays;
	p.hits = d_hits;
	p.occluded = d_occluded;
	p.cand = d_cand;
	p.cand_count = d_cand_count;
	p.cand_k = cand_k;
	p.n = n;
	p.dynamic = n > BLOCK_THREADS ? 1u : 0u;      // a batch that fits one workgroup needs no work queue (and no counter reset)
	// Defaults from sweeps on MI355X (profiles/r01_sweep_opts*.log, r02_ab_r2o/p.log): leave the node
	// loop once fewer than 32 lanes still descend (24 for image-shaped batches); image-shaped (tiled, coherent)
	// batches refill a wave only when it is empty, everything else as soon as 8 lanes are idle.
	p.refill_min = 8;
	p.node_exit = 32;
	uint32_t blocks_per_cu = 0;
	if (opts && opts->struct_size >= 16) {
		if (opts->flags & RTK_TRACE_STATIC) p.dynamic = 0;
		if (opts->image_width && opts->image_height && (size_t)opts->image_width * opts->image_height == n &&
			(opts->image_width % 8u) == 0 && (opts->image_height % 8u) == 0) {
			p.image_w = opts->image_width;
			p.image_h = opts->image_height;
			p.refill_min = 64;
			p.node_exit = 24;
		}
		if (opts->struct_size >= 24) {
			if (opts->refill_min) p.refill_min = opts->refill_min > 64 ? 64 : opts->refill_min;
			blocks_per_cu = opts->blocks_per_cu;
		}
		if (opts->struct_size >= 28 && opts->node_exit) p.node_exit = opts->node_exit > 64 ? 64 : opts->node_exit;
	}
	// No image hint: is the batch an image anyway? Only worth asking where the packet kernels would take it (a closest-hit or any-hit
	// batch without filters, whole 64x64-pixel blocks); costs two small launches and one wait for `stream` (~20 us; the wait also
	// stands between this batch and the host's next enqueue: a caller that knows its image says so in the options).
	static const int detect_default = getenv("RTK_AMD_DETECT_IMAGE") ? atoi(getenv("RTK_AMD_DETECT_IMAGE")) : 1;
	if (detect_default != 0 && p.image_w == 0 && !filter && !collect && !counted && !pk_counted && n >= 16384u && (n % 4096u) == 0u && n <= 0x40000000ull &&
		p.dynamic && ds->stack_entries <= 64 && !(opts && opts->struct_size >= 16 && (opts->flags & (RTK_TRACE_NO_DETECT | RTK_TRACE_NO_PACKET | RTK_TRACE_SORT_RAYS | RTK_TRACE_STATIC)))) {
		uint32_t w = 0, h = 0;
		const int rc = rtk_detect_image(ds, d_rays, n, stream, &w, &h);
		if (rc != RTK_AMD_OK) return rc;
		if (w >= 128u && (w % 64u) == 0u && (h % 64u) == 0u) {
			p.image_w = w;
			p.image_h = h;
			p.refill_min = 64;
			p.node_exit = 24;
		}
	}
	static const int tile_blocks_default = getenv("RTK_AMD_TILE_BLOCKS") ? atoi(getenv("RTK_AMD_TILE_BLOCKS")) : 1;
	p.tile_blocks = (tile_blocks_default && p.image_w && p.image_w % 64u == 0 && p.image_h % 64u == 0) ? 1u : 0u;
	bool filtered = false;
	if (filter) {
		if (filter->struct_size < sizeof(rtk_dev_filter)) { rtk_set_error("rtk_dev_trace: rtk_dev_filter.struct_size is too small"); return RTK_AMD_ERR_BAD_ARG; }
		if (filter->d_mesh_mask && filter->mesh_mask_bits == 0) { rtk_set_error("rtk_dev_trace: mesh mask without mesh_mask_bits"); return RTK_AMD_ERR_BAD_ARG; }
		p.mesh_mask = filter->d_mesh_mask;
		p.mesh_mask_bits = filter->mesh_mask_bits;
		p.ignore_prim = filter->d_ignore_prim;
		p.after = filter->d_after;
		filtered = p.mesh_mask || p.ignore_prim || p.after;
	}

	// image-shaped closest-hit batches go to the wave-packet kernels (rtk_trace_packet.hip). So do image-shaped ANY-HIT batches of
	// whole 64x64-pixel blocks: "is there a hit in (min_t, max_t)" is what a closest-hit traversal answers, at several times the rate of
	// a ray per lane where the rays run side by side (coherent shadow / visibility rays); rtk_packet_any2 retires a ray at its first
	// hit and writes the flags, the C++ kernel (the tiles handed back) writes "the closest hit exists". RTK_AMD_ANY_PACKETS=0: per lane.
	static const int any_packets_default = getenv("RTK_AMD_ANY_PACKETS") ? atoi(getenv("RTK_AMD_ANY_PACKETS")) : 1;
	const bool any_packet = any_hit && any_packets_default != 0 && !counted && p.image_w >= 128u && (p.image_w % 64u) == 0u && (p.image_h % 64u) == 0u &&
		!(opts && opts->struct_size >= 16 && (opts->flags & (RTK_TRACE_SORT_RAYS | RTK_TRACE_STATIC)));
	const bool packet = (!any_hit || any_packet) && !filtered && !collect && p.image_w != 0 && ds->stack_entries <= 64 && !(opts && (opts->flags & RTK_TRACE_NO_PACKET));
	// per-lane kernels read the 64 B compressed nodes unless told otherwise (A/B, and tests that compare the two)
	static const int qnodes_default = getenv("RTK_AMD_QNODES") ? atoi(getenv("RTK_AMD_QNODES")) : 1;
	const bool qn = ds->view.qnodes != nullptr && qnodes_default != 0 && !(opts && opts->struct_size >= 16 && (opts->flags & RTK_TRACE_EXACT_NODES));
	const int variant = packet ? (counted ? VARIANT_PACKET_COUNTED : VARIANT_PACKET) : collect ? VARIANT_COLLECT + (qn ? 1 : 0)
		: ((any_hit ? 1 : 0) | (counted ? 2 : 0) | (filtered ? 4 : 0) | (qn ? 8 : 0));
	// ... and of those, the hand-written kernel (rtk_packet_hot.S) takes every tile it can and hands the rest to the C++ kernel:
	// whole 64x64-pixel blocks, at least two per row, a scene whose planes bound the slab margins and whose leaves are small
	static const int asm_default = getenv("RTK_AMD_PACKET_ASM") ? atoi(getenv("RTK_AMD_PACKET_ASM")) : 1;
	// RTK_AMD_PACKET_BEAM (default 2): 2 = rtk_packet_beam2 (two tiles per wave), 1 = rtk_packet_beam, 0 = rtk_packet_hot (the
	// per-lane slab tests); A/B and tests
	static const int beam_default = getenv("RTK_AMD_PACKET_BEAM") ? atoi(getenv("RTK_AMD_PACKET_BEAM")) : 2;
	int beam = (opts && opts->struct_size >= 16 && (opts->flags & RTK_TRACE_NO_BEAM)) ? 0 : beam_default;
	if (opts && opts->struct_size >= 16 && (opts->flags & RTK_TRACE_ONE_TILE_BEAM) && beam == 2) beam = 1;
	while (beam > 0 && !rtk_packet_hot_available(ds->device, nullptr, beam)) beam--;
	// the counting form of the kernel that is timed (rtk_packet_count2 = rtk_packet_beam2.S with -DRTK_COUNT): only where that kernel runs
	if (any_hit && packet) beam = (beam == 2 && rtk_packet_hot_available(ds->device, nullptr, 4)) ? 4 : -1;     // (the any-hit form exists of rtk_packet_beam2 only; -1: the C++ kernel)
	if (pk_counted) {
		if (beam != 2 || !rtk_packet_hot_available(ds->device, nullptr, 3)) { rtk_set_error("rtk_dev_trace_rays_packet_counted: rtk_packet_beam2 is not the kernel of this launch"); return RTK_AMD_ERR_UNSUPPORTED; }
		beam = 3;
	}
	int hot_blocks_per_cu = 0;
	const bool hot = packet && beam >= 0 && !counted && asm_default != 0 && p.tile_blocks && p.image_w >= 128u && p.image_w <= 65536u && n <= 0x40000000ull &&
		ds->bound_abs < 0x1p19f && (beam >= 2 || ds->big_leaf_fraction <= 0.02) && !(opts && (opts->flags & RTK_TRACE_NO_ASM)) &&
		rtk_packet_hot_available(ds->device, &hot_blocks_per_cu, beam);       // (rtk_packet_beam2 has the group rule for leaves of four and more triangles; the one-tile kernels hand such tiles back)
	if (pk_counted && !hot) { rtk_set_error("rtk_dev_trace_rays_packet_counted: this batch does not run on the assembly packet kernel (image hint, whole 64x64-pixel blocks, small leaves)"); return RTK_AMD_ERR_UNSUPPORTED; }
	static const bool path_log = getenv("RTK_AMD_LOG_PATH") != nullptr;
	if (path_log) fprintf(stderr, "rtk_dev_trace: n %zu image %u x %u packet %d hot %d beam %d opts %p flags %x\n", n, p.image_w, p.image_h, (int)packet, (int)hot, beam, (const void *)opts, opts ? opts->flags : 0u);
	const int occ = blocks_per_cu_of(ds->device, variant);
	if (blocks_per_cu == 0 || blocks_per_cu > (uint32_t)occ) blocks_per_cu = (uint32_t)occ;
	// Plain closest-hit / any-hit batches on compressed nodes go to the hand-written per-lane kernels (rtk_lane_hot.S); the rays
	// they hand back (not tame, a leaf of four or more triangles, a stack deeper than the LDS column) follow in rtk_trace_kernel.
	// Byte offsets into nodes, triangles, rays and the ray order are 32-bit and kept below 2^31 there.
	static const int lane_asm_default = getenv("RTK_AMD_LANE_ASM") ? atoi(getenv("RTK_AMD_LANE_ASM")) : 1;
	int lane_blocks_per_cu = 0;
	const bool lane_hot = !packet && !collect && !counted && !filtered && qn && p.dynamic && p.image_w == 0 && lane_asm_default != 0 &&
		RTK_TRI_STRIDE == 48 && n <= ((size_t)1 << 26) && (uint64_t)ds->view.num_nodes * 64u < 0x80000000ull &&
		(uint64_t)ds->view.num_tris * RTK_TRI_STRIDE < 0x80000000ull && ds->bound_abs < 0x1p60f && (!any_hit || ds->big_leaf_fraction <= 0.02) &&
		!(opts && opts->struct_size >= 16 && (opts->flags & (RTK_TRACE_NO_ASM | RTK_TRACE_STATIC))) && ds->stack_entries < 512u &&
		rtk_lane_hot_available(ds->device, &lane_blocks_per_cu);

	const size_t blocks_needed = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
	size_t blocks = (p.dynamic || packet) ? (size_t)ds->num_cus * blocks_per_cu : blocks_needed;
	if (blocks > blocks_needed) blocks = blocks_needed;
	if (blocks > 0x7fffffffu) { rtk_set_error("rtk_dev_trace: batch too large for one launch"); return RTK_AMD_ERR_BAD_ARG; }

	// From here on the launch uses the scratch set of (scene, stream); the mutex is held until everything is
	// enqueued, so that two host threads feeding one stream cannot interleave "reset queue heads" and "launch".
	std::lock_guard<std::mutex> lock(ds->scratch_mutex);
	LaunchScratch *sc = scratch_for(ds, stream);
	if (!sc) return RTK_AMD_ERR_OOM;

	// spill area for rays whose stack outgrows LDS
	size_t lane_hot_blocks = (size_t)ds->num_cus * (size_t)lane_blocks_per_cu;
	if (lane_hot_blocks > blocks_needed) lane_hot_blocks = blocks_needed;
	const size_t lanes = (lane_hot && lane_hot_blocks > blocks ? lane_hot_blocks : blocks) * BLOCK_THREADS;     // (one spill area serves the assembly kernel and the C++ pass behind it)
	static const size_t lane_lds = getenv("RTK_AMD_LANE_LDS") ? (size_t)atoi(getenv("RTK_AMD_LANE_LDS")) : LDS_STACK;   // (A/B builds of rtk_lane_hot.S with fewer LDS entries)
	const size_t lds_entries = packet ? 16 : (lane_hot && lane_lds < LDS_STACK) ? lane_lds : LDS_STACK;   // PK_LDS_STACK in rtk_trace_packet.hip
	const size_t spill_cap = ds->stack_entries > lds_entries ? ds->stack_entries - lds_entries : 0;
	if (spill_cap && (sc->spill_lanes < lanes || sc->spill_entries_per_lane < spill_cap)) {
		// an earlier launch on this stream may still be using the old area
		if (sc->d_spill) { RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP); (void)hipFree(sc->d_spill); }
		sc->d_spill = nullptr;
		sc->spill_lanes = sc->spill_entries_per_lane = 0;
		RTK_HIP_CHECK(hipMalloc(&sc->d_spill, lanes * spill_cap * sizeof(uint2)), RTK_AMD_ERR_OOM);
		sc->spill_lanes = lanes;
		sc->spill_entries_per_lane = spill_cap;
	}
	// optional ray reordering pre-pass (per-lane kernels only)
	p.perm = nullptr;
	if (!packet && opts && (opts->flags & RTK_TRACE_SORT_RAYS) && n < 0x7fffffffu) {
		const uint32_t n32 = (uint32_t)n;
		const size_t words = rtk_sort_scratch_words(n32);
		if (sc->sort_capacity < n) {
			if (sc->d_sort) { RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP); (void)hipFree(sc->d_sort); }
			sc->d_sort = nullptr;
			sc->sort_capacity = 0;
			// [words_a | words_b] 8 B each, bounds 6 words + sort scratch
			RTK_HIP_CHECK(hipMalloc(&sc->d_sort, n * 16 + (words + 16) * 4), RTK_AMD_ERR_OOM);
			sc->sort_capacity = n;
		}
		unsigned long long *keys_a = (unsigned long long *)sc->d_sort, *keys_b = keys_a + sc->sort_capacity;
		uint32_t *bounds = (uint32_t *)(keys_b + sc->sort_capacity), *scratch = bounds + 16;
		static const uint32_t cell_bits = getenv("RTK_AMD_SORT_CELL_BITS") ? (uint32_t)atoi(getenv("RTK_AMD_SORT_CELL_BITS")) : 7u;   // 2^7 cells per axis: 3.44 against 3.32 Grays/s at 2^5 on the shadow batch (profiles/r03_ab_sort_cells.log)
		static const uint32_t with_octant = getenv("RTK_AMD_SORT_OCTANT") ? (uint32_t)atoi(getenv("RTK_AMD_SORT_OCTANT")) : 0u;
		static const int entry_key = getenv("RTK_AMD_SORT_KEY") ? atoi(getenv("RTK_AMD_SORT_KEY")) : 1;   // 0: origin cell in the batch's origin bounds
		if (entry_key && ds->view.num_nodes) {
			hipLaunchKernelGGL(rtk_ray_entry_keys_kernel, dim3((n32 + 255u) / 256u), dim3(256), 0, stream, d_rays, n32, ds->view.nodes, keys_a,
				cell_bits, with_octant);
		} else {
			static const uint32_t init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
			RTK_HIP_CHECK(hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, stream), RTK_AMD_ERR_HIP);
			hipLaunchKernelGGL(rtk_ray_bounds_kernel, dim3((unsigned)(ds->num_cus * 2)), dim3(256), 0, stream, d_rays, (unsigned long long)n,
				(unsigned long long)(n >= (1u << 16) ? 61 : 1), bounds);
			hipLaunchKernelGGL(rtk_ray_keys_kernel, dim3((n32 + 255u) / 256u), dim3(256), 0, stream, d_rays, n32, bounds, keys_a,
				cell_bits, with_octant);
		}
		// one 8-byte word per ray (key over the ray's number), no value array: two passes of 16 B per ray
		const bool in_b = rtk_sort_words_async(keys_a, keys_b, n32, 32u, 32u + 3u * cell_bits + (with_octant ? 3u : 0u), scratch, stream);
		p.perm = in_b ? keys_b : keys_a;
	}
	p.spill = sc->d_spill;
	p.spill_stride = (uint32_t)(spill_cap ? sc->spill_lanes : 0);
	p.spill_cap = (uint32_t)spill_cap;
	p.counter = sc->d_counter;

	// entry points shared by the tiles of a 64x64-pixel block (rtk_packet_entries_kernel, one small launch ahead of the traversal)
	static const int entries_default = getenv("RTK_AMD_PACKET_ENTRIES") ? atoi(getenv("RTK_AMD_PACKET_ENTRIES")) : 1;
	static const unsigned entries_target = getenv("RTK_AMD_ENTRY_TARGET") ? (unsigned)atoi(getenv("RTK_AMD_ENTRY_TARGET")) : 26u;   // (list size at which the walk stops: 20 / 24 / 28 / 32 / 36 -> 17.7 / 18.0 / 18.0 / 17.9 / 17.85 Grays/s on config 2, profiles/r04_packet_entries.log)
	static const unsigned entries_levels = getenv("RTK_AMD_ENTRY_LEVELS") ? (unsigned)atoi(getenv("RTK_AMD_ENTRY_LEVELS")) : 8u;
	const bool entries = packet && p.tile_blocks && entries_default != 0 && ds->bound_abs < 0x1p19f && ds->view.num_nodes != 0u &&
		!(opts && opts->struct_size >= 16 && (opts->flags & RTK_TRACE_NO_ENTRIES));
	// queue heads and visit counters start from zero; a one-block static launch uses neither. (With entry lists the pre-pass
	// kernel clears them itself: a 4.6 us fill kernel and its launch gap less per frame.)
	if ((p.dynamic || packet || counted) && !entries) RTK_HIP_CHECK(hipMemsetAsync(sc->d_counter, 0, RTK_COUNTER_WORDS * sizeof(unsigned long long), stream), RTK_AMD_ERR_HIP);
	if (entries) {
		const size_t nblk = (size_t)(p.image_w >> 6) * (p.image_h >> 6);
		if (sc->entries_capacity < nblk) {
			if (sc->d_entries) { RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP); (void)hipFree(sc->d_entries); }
			sc->d_entries = nullptr;
			sc->entries_capacity = 0;
			RTK_HIP_CHECK(hipMalloc(&sc->d_entries, nblk * sizeof(PkBlockEntries)), RTK_AMD_ERR_OOM);
			sc->entries_capacity = nblk;
		}
		rtk_packet_entries_launch(p, (PkBlockEntries *)sc->d_entries, ds->bound_abs > 1.0f ? ds->bound_abs : 1.0f, entries_target, entries_levels, stream);
		p.entries = (const PkBlockEntries *)sc->d_entries;
	}
	if (hot) {
		const size_t tiles = n >> 6;
		if (sc->leftover_capacity < tiles) {
			if (sc->d_leftover) { RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP); (void)hipFree(sc->d_leftover); }
			sc->d_leftover = nullptr;
			sc->leftover_capacity = 0;
			RTK_HIP_CHECK(hipMalloc(&sc->d_leftover, tiles * sizeof(uint32_t)), RTK_AMD_ERR_OOM);
			sc->leftover_capacity = tiles;
		}
		PkHotParams hp = {};
		hp.nodes = p.sc.nodes; hp.tris = p.sc.tris; hp.rays = p.rays; hp.hits = any_hit ? reinterpret_cast<rtk_hit_record *>(p.occluded) : p.hits; hp.counter = p.counter; hp.leftover = sc->d_leftover;
		hp.num_blocks = (uint32_t)(tiles >> 6);
		hp.image_w = p.image_w;
		hp.blocks_per_row = p.image_w >> 6;
		hp.bpr_magic = (uint32_t)((0x100000000ull + hp.blocks_per_row - 1u) / hp.blocks_per_row);
		hp.bound_abs = ds->bound_abs > 1.0f ? ds->bound_abs : 1.0f;
		hp.entries = p.entries;
		// (RTK_AMD_HOT_BLOCKS_PER_CU: fewer resident workgroups, to tell a latency-bound kernel from a throughput-bound one)
		static const int hot_bpc_env = getenv("RTK_AMD_HOT_BLOCKS_PER_CU") ? atoi(getenv("RTK_AMD_HOT_BLOCKS_PER_CU")) : 0;
		size_t hot_blocks = (size_t)ds->num_cus * (size_t)(hot_bpc_env > 0 && hot_bpc_env < hot_blocks_per_cu ? hot_bpc_env : hot_blocks_per_cu);
		if (hot_blocks > blocks_needed) hot_blocks = blocks_needed;
		const int rc = rtk_packet_hot_launch(ds->device, hp, (unsigned)hot_blocks, stream, beam);
		if (rc != RTK_AMD_OK) return rc;
		// the tiles it handed back (mixed signs or axes, untame rays, a big leaf, a deep stack), by the C++ kernel
		// (a small grid: the list is empty for most batches, and a launch that only finds that out should cost next to nothing)
		p.tile_list = sc->d_leftover;
		const size_t left_blocks = std::min<size_t>(blocks, (size_t)ds->num_cus * 2u);
		rtk_packet_launch(p, (unsigned)left_blocks, stream, pk_counted != nullptr);      // (counting: the handed-back tiles' steps are counted too)
	} else if (packet) rtk_packet_launch(p, (unsigned)blocks, stream, counted != nullptr);
	else if (lane_hot) {
		if (sc->leftover_capacity < n * 2u) {           // (counted in uint32: the list holds one 8-byte word per left-over ray)
			if (sc->d_leftover) { RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP); (void)hipFree(sc->d_leftover); }
			sc->d_leftover = nullptr;
			sc->leftover_capacity = 0;
			RTK_HIP_CHECK(hipMalloc(&sc->d_leftover, n * sizeof(unsigned long long)), RTK_AMD_ERR_OOM);
			sc->leftover_capacity = n * 2u;
		}
		LnHotParams hp = {};
		hp.qnodes = p.sc.qnodes; hp.tris = p.sc.tris; hp.rays = p.rays;
		hp.out = any_hit ? (void *)p.occluded : (void *)p.hits;
		hp.counter = p.counter;
		hp.leftover = reinterpret_cast<unsigned long long *>(sc->d_leftover);
		hp.perm = p.perm;
		hp.n = (uint32_t)n;
		// (re-swept for these kernels: refill at 16 idle lanes instead of 8 is +1 % / +2 %, profiles/r04_lane_sweep.log)
		hp.refill_min = (opts && opts->struct_size >= 24 && opts->refill_min) ? p.refill_min : 16u;
		hp.node_exit = p.node_exit;
		hp.bound_abs = ds->bound_raw;             // (no floor of 1: these kernels test child words, not inverted boxes)
		hp.spill = p.spill;
		hp.spill_stride = p.spill_stride;
		hp.spill_cap = p.spill_cap;
		const int rc = rtk_lane_hot_launch(ds->device, hp, (unsigned)lane_hot_blocks, stream, any_hit);
		if (rc != RTK_AMD_OK) return rc;
		static const int lane_stats = getenv("RTK_AMD_LANE_STATS") ? atoi(getenv("RTK_AMD_LANE_STATS")) : 0;
		if (lane_stats) {           // (diagnostics: how many rays the assembly kernel handed back; synchronises the stream)
			unsigned long long left = 0;
			(void)hipMemcpyAsync(&left, p.counter + RTK_LANE_LEFTOVER_WORD, sizeof(left), hipMemcpyDeviceToHost, stream);
			(void)hipStreamSynchronize(stream);
			fprintf(stderr, "rtk_lane_hot: %llu of %zu rays handed back (%.3f %%)\n", left, n, 100.0 * (double)left / (double)n);
		}
		// the rays it left over (none in most batches: a small grid that finds an empty list costs next to nothing)
		TraceParams lp = p;
		lp.perm = hp.leftover;
		lp.n_indirect = p.counter + RTK_LANE_LEFTOVER_WORD;
		lp.n = 0;
		const size_t left_blocks = std::min<size_t>(blocks, (size_t)ds->num_cus);
		hipLaunchKernelGGL(trace_variant(variant), dim3((unsigned)left_blocks), dim3(BLOCK_THREADS), 0, stream, lp);
	} else hipLaunchKernelGGL(trace_variant(variant), dim3((unsigned)blocks), dim3(BLOCK_THREADS), 0, stream, p);
	RTK_HIP_CHECK(hipGetLastError(), RTK_AMD_ERR_HIP);
	if (pk_counted) {
		unsigned long long c[16];
		RTK_HIP_CHECK(hipMemcpyAsync(c, sc->d_counter, sizeof(c), hipMemcpyDeviceToHost, stream), RTK_AMD_ERR_HIP);
		RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP);
		pk_counted->pairs = c[11]; pk_counted->node_steps = c[12]; pk_counted->triangles_fetched = c[13]; pk_counted->triangle_group_tests = c[14];
		pk_counted->tiles_handed_back = c[RTK_LEFTOVER_COUNT_WORD];
		pk_counted->handed_back_node_steps = c[7]; pk_counted->handed_back_triangle_steps = c[8];
		pk_counted->tiles = n >> 6;
	}
	if (counted) {
		unsigned long long c[16], err = 0;
		RTK_HIP_CHECK(hipMemcpyAsync(c, sc->d_counter, sizeof(c), hipMemcpyDeviceToHost, stream), RTK_AMD_ERR_HIP);
		RTK_HIP_CHECK(hipMemcpyAsync(&err, sc->d_counter + RTK_ERROR_WORD, sizeof(err), hipMemcpyDeviceToHost, stream), RTK_AMD_ERR_HIP);
		RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP);
		counted->rays = c[1]; counted->nodes = c[2]; counted->leaves = c[3];
		counted->triangles = c[4]; counted->hits = c[5]; counted->stack_spills = c[6];
		counted->wave_node_steps = c[7]; counted->wave_triangle_steps = c[8]; counted->wave_rays = c[9];
		if (err) {
			(void)hipMemsetAsync(sc->d_counter + RTK_ERROR_WORD, 0, sizeof(err), stream);
			rtk_set_error("rtk_dev_trace: traversal stack overflow (corrupted scene)");
			return RTK_AMD_ERR_BAD_SCENE;
		}
	}
	return RTK_AMD_OK;
}

// Did any launch of this scene on `stream` since the last call overflow a traversal stack? Synchronises the stream.
int rtk_trace_status(const rtk_dev_scene *ds_c, hipStream_t stream)
{
	rtk_dev_scene *ds = const_cast<rtk_dev_scene *>(ds_c);
	if (!ds) { rtk_set_error("rtk_dev_trace_status: NULL scene"); return RTK_AMD_ERR_BAD_ARG; }
	// the error word's address is looked up under the lock; the wait for the stream happens outside it, so that threads
	// tracing one scene on their own streams do not queue up behind each other's synchronisation
	unsigned long long *word = nullptr;
	{
		std::lock_guard<std::mutex> lock(ds->scratch_mutex);
		for (LaunchScratch *s : ds->scratch) if (s->stream == stream) { word = s->d_counter + RTK_ERROR_WORD; break; }
	}
	if (!word) {
		RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP);
		return RTK_AMD_OK;
	}
	unsigned long long e = 0;
	RTK_HIP_CHECK(hipMemcpyAsync(&e, word, sizeof(e), hipMemcpyDeviceToHost, stream), RTK_AMD_ERR_HIP);
	RTK_HIP_CHECK(hipStreamSynchronize(stream), RTK_AMD_ERR_HIP);
	if (e) {
		(void)hipMemsetAsync(word, 0, sizeof(e), stream);      // reported once
		rtk_set_error("rtk_dev_trace: traversal stack overflow (corrupted scene)");
		return RTK_AMD_ERR_BAD_SCENE;
	}
	return RTK_AMD_OK;
}

// A stream is going away (the host-pointer calls own one per thread): the scratch sets made for it are released, so that
// threads coming and going do not pile them up and a recycled stream handle never finds an old entry.
void rtk_scene_drop_stream(rtk_dev_scene *ds, hipStream_t stream)
{
	if (!ds) return;
	std::lock_guard<std::mutex> lock(ds->scratch_mutex);
	for (size_t i = 0; i < ds->scratch.size();) {
		if (ds->scratch[i]->stream == stream) { rtk_scratch_free(ds->scratch[i]); ds->scratch.erase(ds->scratch.begin() + (long)i); } else i++;
	}
}

int rtk_launch_expand(const rtk_dev_scene *ds_c, const rtk_hit_record *d_records, size_t n, rtk_hit *d_hits,
	uint8_t *d_mask, hipStream_t stream, unsigned long long *h_status, uint32_t ticket)
{
	rtk_dev_scene *ds = const_cast<rtk_dev_scene *>(ds_c);
	if (!ds || (!d_records && n)) { rtk_set_error("rtk_dev_expand_hits: bad argument"); return RTK_AMD_ERR_BAD_ARG; }
	if (n == 0) return RTK_AMD_OK;
	{
		int cur = -1;
		if (hipGetDevice(&cur) != hipSuccess || cur != ds->device) {
			rtk_set_error("rtk_dev_expand_hits: the scene lives on device %d, the calling thread's current device is %d", ds->device, cur);
			return RTK_AMD_ERR_BAD_ARG;
		}
	}
	if (d_hits && rtk_scene_side_arrays(ds, stream) != RTK_AMD_OK) return RTK_AMD_ERR_OOM;
	const unsigned long long *status_word = nullptr;
	if (h_status) {
		// the error word of the launches on this stream (rtk_launch_trace has made the scratch set)
		std::lock_guard<std::mutex> lock(ds->scratch_mutex);
		for (LaunchScratch *s : ds->scratch) if (s->stream == stream) status_word = s->d_counter + RTK_ERROR_WORD;
		if (!status_word) h_status = nullptr;
	}
	const size_t blocks = (n + 255) / 256;
	if (blocks != 1 || !h_status) ticket = 0u;             // the ticket is written by a lone workgroup after its results
	hipLaunchKernelGGL(rtk_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ds->view, d_records,
		(unsigned long long)n, d_hits, d_mask, status_word, h_status, ticket);
	RTK_HIP_CHECK(hipGetLastError(), RTK_AMD_ERR_HIP);
	return RTK_AMD_OK;
}

// rtk_trace_ray's own launch: ONE ray (read here, on the host: it travels in the kernel argument), one wave, the full rtk_hit and the mask written where the host
// reads them, then the ticket ((ticket << 32) | 0, or | 2 = "not done: take the batch path"). No scratch, no queue, no second launch.
int rtk_launch_trace_one(const rtk_dev_scene *ds, const rtk_ray *d_ray, rtk_hit *d_hit, uint8_t *d_mask, hipStream_t stream,
	unsigned long long *h_status, uint32_t ticket)
{
	if (!ds || !d_ray || !d_hit || !d_mask || !h_status || !ticket) { rtk_set_error("rtk_trace_ray: bad argument"); return RTK_AMD_ERR_BAD_ARG; }
	int cur = -1;
	if (hipGetDevice(&cur) != hipSuccess || cur != ds->device) {
		rtk_set_error("rtk_trace_ray: the scene lives on device %d, the calling thread's current device is %d", ds->device, cur);
		return RTK_AMD_ERR_BAD_ARG;
	}
	if ((uint64_t)ds->view.num_nodes * 128u > 0xffffff00ull || (uint64_t)ds->view.num_tris * RTK_TRI_STRIDE > 0xffffff00ull) {
		rtk_set_error("rtk_trace_ray: the one-ray kernel addresses nodes and triangles with 32-bit byte offsets (scene: %u nodes, %u triangles)", ds->view.num_nodes, ds->view.num_tris);
		return RTK_AMD_ERR_UNSUPPORTED;
	}
	if (rtk_scene_side_arrays(ds, stream) != RTK_AMD_OK) return RTK_AMD_ERR_OOM;
	hipLaunchKernelGGL(rtk_trace_one_kernel, dim3(1), dim3(64), 0, stream, ds->view, *d_ray, d_hit, d_mask, h_status, ticket);
	RTK_HIP_CHECK(hipGetLastError(), RTK_AMD_ERR_HIP);
	return RTK_AMD_OK;
}
