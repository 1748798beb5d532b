// launch_latency_probe.hip -- what one GPU round trip costs a host thread on this machine, without any tracing:
//   (a) an empty one-workgroup kernel that writes a ticket into pinned host memory, the host polling for it;
//   (b) the same waited for with hipStreamSynchronize;
//   (c) two such kernels back to back (the per-ray call's trace + expand), polled;
//   (d) a kernel that chases K dependent 128-byte loads through a 256 MB table first (the shape of one ray's traversal).
// The floor under rtk_trace_ray.   hipcc --offload-arch=gfx950 -O2 -o probe launch_latency_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_ticket(unsigned long long *status, unsigned ticket)
{
	if (threadIdx.x == 0) __hip_atomic_store(status, (unsigned long long)ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_nothing(int *sink) { if (sink && threadIdx.x == 1000) *sink = 1; }
__global__ void k_chase(const unsigned *table, unsigned start, int hops, unsigned long long *status, unsigned ticket)
{
	unsigned at = start + threadIdx.x * 977u;
	for (int i = 0; i < hops; i++) at = table[(size_t)(at & 0x1fffffu) * 32u];
	if (threadIdx.x == 0) __hip_atomic_store(status, ((unsigned long long)(at & 1u) << 40) | ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
	hipStream_t s;
	hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	unsigned long long *status;
	hipHostMalloc((void **)&status, 64, hipHostMallocDefault);
	*status = 0;
	unsigned *table;
	const size_t entries = (size_t)1 << 21;                  // 2M lines of 128 bytes = 256 MB
	hipMalloc((void **)&table, entries * 128);
	std::vector<unsigned> h(entries * 32);
	unsigned x = 12345;
	for (size_t i = 0; i < entries; i++) { x = x * 1664525u + 1013904223u; h[i * 32] = x >> 8; }
	hipMemcpy(table, h.data(), entries * 128, hipMemcpyHostToDevice);
	const int reps = 2000;
	unsigned ticket = 0;
	auto wait = [&](unsigned t) { while ((unsigned)(__atomic_load_n((volatile unsigned long long *)status, __ATOMIC_ACQUIRE) & 0xffffffffu) != t) {} };
	for (int warm = 0; warm < 200; warm++) { k_ticket<<<1, 64, 0, s>>>(status, ++ticket); wait(ticket); }
	double t0 = now_us();
	for (int i = 0; i < reps; i++) { k_ticket<<<1, 64, 0, s>>>(status, ++ticket); wait(ticket); }
	printf("(a) one empty kernel + ticket in pinned memory, host polls:   %.2f us per round trip\n", (now_us() - t0) / reps);
	t0 = now_us();
	for (int i = 0; i < reps; i++) { k_ticket<<<1, 64, 0, s>>>(status, ++ticket); hipStreamSynchronize(s); }
	printf("(b) the same, hipStreamSynchronize instead of polling:         %.2f us\n", (now_us() - t0) / reps);
	t0 = now_us();
	for (int i = 0; i < reps; i++) { k_nothing<<<1, 64, 0, s>>>(nullptr); k_ticket<<<1, 64, 0, s>>>(status, ++ticket); wait(ticket); }
	printf("(c) two kernels back to back, polled:                          %.2f us\n", (now_us() - t0) / reps);
	for (int hops : { 8, 16, 32, 64 }) {
		t0 = now_us();
		for (int i = 0; i < reps; i++) { k_chase<<<1, 64, 0, s>>>(table, (unsigned)i * 7919u, hops, status, ++ticket); wait(ticket); }
		printf("(d) one kernel chasing %2d dependent 128-B loads (256 MB table):  %.2f us\n", hops, (now_us() - t0) / reps);
	}
	return 0;
}
