// does one 1024-thread workgroup get (nearly) the whole 160 KiB of a CU's LDS?  (throw-away probe)
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char dyn_lds[];
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned bytes) {
	unsigned *w = reinterpret_cast<unsigned *>(dyn_lds);
	for (unsigned i = threadIdx.x; i < bytes / 4; i += 1024) w[i] = i * 2654435761u;
	__syncthreads();
	unsigned acc = 0;
	for (unsigned i = threadIdx.x; i < bytes / 4; i += 1024) acc ^= w[(i * 7u + 13u) % (bytes / 4)];
	atomicXor(out + blockIdx.x, acc);
}
int main() {
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  sharedMemPerBlockOptin %zu  regsPerBlock %d  CUs %d\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin, p.regsPerBlock, p.multiProcessorCount);
	unsigned *d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0, 4096 * 4);
	for (unsigned bytes : { 65536u, 98304u, 131072u, 160320u, 163840u }) {
		hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
		hipLaunchKernelGGL(k, dim3(512), dim3(1024), bytes, 0, d, bytes);
		hipError_t e2 = hipDeviceSynchronize();
		printf("  %u bytes: set attribute %s, launch+sync %s / %s\n", bytes, hipGetErrorString(e), hipGetErrorString(hipGetLastError()), hipGetErrorString(e2));
	}
	return 0;
}
