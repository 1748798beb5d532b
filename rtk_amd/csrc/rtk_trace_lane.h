// rtk_trace_lane.h -- device helpers shared by the kernels that keep one ray per lane (rtk_trace.hip: rays bound to
// lanes; rtk_trace_pool.hip: rays in an LDS pool): streamed ray / hit accesses, the node and triangle loads as one burst
// with one wait, the 5-comparator sort step.
#pragma once

#include "rtk_trace_shared.h"

__device__ __forceinline__ float4 ld_f4(const char *p) { return *reinterpret_cast<const float4 *>(p); }
typedef float f32x4_ __attribute__((ext_vector_type(4)));
// rays are read once and hit records written once: streamed past the caches ("nt") so that they do not push the BVH out of the L2
__device__ __forceinline__ float4 ld_f4_stream(const char *p)
{
	const f32x4_ v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ *>(p));
	return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_f4_stream(void *p, float x, float y, float z, float w)
{
	f32x4_ v;
	v.x = x; v.y = y; v.z = z; v.w = w;
	__builtin_nontemporal_store(v, reinterpret_cast<f32x4_ *>(p));
}
__device__ __forceinline__ uint4 ld_u4(const char *p) { return *reinterpret_cast<const uint4 *>(p); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// All seven 16-B pieces of a node are requested back to back and waited for once. Left to
// itself hipcc serialises them (load, wait, reuse the registers, load ...) to save VGPRs,
// which turns one memory round trip per node into four. SGPR base + 32-bit VGPR offsets.
__device__ __forceinline__ void load_node(const char *base, uint32_t a_nx, uint32_t a_fx, uint32_t a_ny, uint32_t a_fy,
	uint32_t a_nz, uint32_t a_fz, uint32_t a_node, f32x4 &nx, f32x4 &fx, f32x4 &ny, f32x4 &fy, f32x4 &nz, f32x4 &fz, u32x4 &ch)
{
	asm volatile(
		"global_load_dwordx4 %0, %7, %14\n\t"
		"global_load_dwordx4 %1, %8, %14\n\t"
		"global_load_dwordx4 %2, %9, %14\n\t"
		"global_load_dwordx4 %3, %10, %14\n\t"
		"global_load_dwordx4 %4, %11, %14\n\t"
		"global_load_dwordx4 %5, %12, %14\n\t"
		"global_load_dwordx4 %6, %13, %14 offset:96\n\t"
		"s_waitcnt vmcnt(0)"
		: "=&v"(nx), "=&v"(fx), "=&v"(ny), "=&v"(fy), "=&v"(nz), "=&v"(fz), "=&v"(ch)
		: "v"(a_nx), "v"(a_fx), "v"(a_ny), "v"(a_fy), "v"(a_nz), "v"(a_fz), "v"(a_node), "s"(base)
		: "memory");
}

__device__ __forceinline__ void load_tri(const char *base, uint32_t a_tri, f32x4 &A, f32x4 &B, f32x4 &C)
{
	asm volatile(
		"global_load_dwordx4 %0, %3, %4\n\t"
		"global_load_dwordx4 %1, %3, %4 offset:16\n\t"
		"global_load_dwordx4 %2, %3, %4 offset:32\n\t"
		"s_waitcnt vmcnt(0)"
		: "=&v"(A), "=&v"(B), "=&v"(C)
		: "v"(a_tri), "s"(base)
		: "memory");
}

// 64 B compressed node (DevNodeQ): four 16-B pieces, one wait.
__device__ __forceinline__ void load_qnode(const char *base, uint32_t a_node, f32x4 &l0, u32x4 &l1, u32x4 &l2, u32x4 &l3)
{
	asm volatile(
		"global_load_dwordx4 %0, %4, %5\n\t"
		"global_load_dwordx4 %1, %4, %5 offset:16\n\t"
		"global_load_dwordx4 %2, %4, %5 offset:32\n\t"
		"global_load_dwordx4 %3, %4, %5 offset:48\n\t"
		"s_waitcnt vmcnt(0)"
		: "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
		: "v"(a_node), "s"(base)
		: "memory");
}

__device__ __forceinline__ float ubyte_f32(uint32_t w, int k) { return (float)((w >> (8 * k)) & 255u); }   // v_cvt_f32_ubyteK

// one comparator of a sorting network on 64-bit (key in the high half) pairs: see the node step of rtk_trace_kernel
__device__ __forceinline__ void cswap_pair(double &a, double &b)
{
	double lo, hi;
	asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b));
	a = lo; b = hi;
}

__device__ __forceinline__ void cswap(float &ka, uint32_t &ra, float &kb, uint32_t &rb)
{
	const bool s = kb < ka;
	const float k0 = s ? kb : ka, k1 = s ? ka : kb;
	const uint32_t r0 = s ? rb : ra, r1 = s ? ra : rb;
	ka = k0; kb = k1; ra = r0; rb = r1;
}

