// Gather-rate probe (throw-away measurement tool, not part of the library): what the memory system of one MI355X delivers to a
// kernel shaped like the per-lane traversal -- every lane follows its own dependent chain of record fetches (the next index comes
// out of the record just read), 20 waves per CU, records of 64 B (a compressed node) read as 1..4 dwordx4 -- from a table that fits
// the Infinity Cache (169 MB, the 1M-triangle scene) or does not (1.7 GB, the 10M one), with a given fraction of the fetches going
// to a small hot set (the top of the tree, L2 hits).  Prints records/s and bytes/s; run under rocprofv3 --pmc FETCH_SIZE to see how
// the counter tallies this access shape.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe scripts/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int PIECES, int STRIDE /*record size in 16-B pieces*/>
__global__ void __launch_bounds__(256, 5) k_chase(const uint4 *table, uint32_t n_rec, uint32_t n_hot, uint32_t hot_per_256, int steps, uint32_t *out)
{
	const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
	uint32_t idx = mix(gid * 2654435761u + 12345u) % n_rec;
	uint32_t acc = 0;
	for (int s = 0; s < steps; s++) {
		const uint4 *rec = table + (size_t)idx * (uint32_t)STRIDE;
		uint4 a = rec[0];
		uint32_t v = a.x ^ a.w;
		if (PIECES > 1) { const uint4 b = rec[1]; v ^= b.y; }
		if (PIECES > 2) { const uint4 c = rec[2]; v ^= c.z; }
		if (PIECES > 3) { const uint4 d = rec[3]; v ^= d.x; }
		if (PIECES > 4) { const uint4 e = rec[4]; const uint4 f = rec[5]; const uint4 g = rec[6]; const uint4 i = rec[7]; v ^= e.x ^ f.y ^ g.z ^ i.w; }
		acc += v;
		const uint32_t h = mix(v + gid + (uint32_t)s * 0x9e3779b9u);
		idx = ((h & 255u) < hot_per_256) ? (h >> 8) % n_hot : (h >> 8) % n_rec;       // the chain depends on the data read
	}
	out[gid] = acc;
}

// two independent chains per lane: twice the requests in flight from the same number of lanes (is the rate above bound by latency x
// lanes, or by the memory system's request rate?)
__global__ void __launch_bounds__(256, 5) k_chase2(const uint4 *table, uint32_t n_rec, int steps, uint32_t *out)
{
	const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
	uint32_t i0 = mix(gid * 2654435761u + 12345u) % n_rec, i1 = mix(gid * 40503u + 977u) % n_rec;
	uint32_t acc = 0;
	for (int s = 0; s < steps; s++) {
		const uint4 a = table[(size_t)i0 * 4u], b = table[(size_t)i1 * 4u];
		acc += a.x ^ b.y;
		i0 = (mix(a.x + gid + (uint32_t)s) >> 8) % n_rec;
		i1 = (mix(b.w + gid * 3u + (uint32_t)s) >> 8) % n_rec;
	}
	out[gid] = acc;
}

int main(int argc, char **argv)
{
	const double table_mb = argc > 1 ? atof(argv[1]) : 169.0;
	const double hot_mb = argc > 2 ? atof(argv[2]) : 4.0;
	const int steps = argc > 3 ? atoi(argv[3]) : 256;
	const int blocks = argc > 4 ? atoi(argv[4]) : 256 * 5;
	const uint32_t n_rec = (uint32_t)(table_mb * 1e6 / 64.0), n_hot = (uint32_t)(hot_mb * 1e6 / 64.0);
	uint4 *d_table; uint32_t *d_out;
	CK(hipMalloc(&d_table, (size_t)n_rec * 64));
	CK(hipMalloc(&d_out, (size_t)blocks * 2 * 256 * 4));
	{
		std::vector<uint32_t> h((size_t)n_rec * 16);
		uint64_t s = 88172645463325252ull;
		for (auto &w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
		CK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	}
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	printf("table %.0f MB (%u records of 64 B), hot set %.1f MB, %d blocks x 256 lanes, %d dependent fetches per lane\n", table_mb, n_rec, hot_mb, blocks, steps);
	for (int hot = 0; hot <= 192; hot += 64) {
		for (int pieces = 1; pieces <= 8; pieces *= 2) {
			float best = 1e30f;
			for (int rep = 0; rep < 4; rep++) {
				CK(hipEventRecord(e0, 0));
				if (pieces == 1) hipLaunchKernelGGL((k_chase<1, 4>), dim3(blocks), dim3(256), 0, 0, d_table, n_rec, n_hot, (uint32_t)hot, steps, d_out);
				else if (pieces == 2) hipLaunchKernelGGL((k_chase<2, 4>), dim3(blocks), dim3(256), 0, 0, d_table, n_rec, n_hot, (uint32_t)hot, steps, d_out);
				else if (pieces == 4) hipLaunchKernelGGL((k_chase<4, 4>), dim3(blocks), dim3(256), 0, 0, d_table, n_rec, n_hot, (uint32_t)hot, steps, d_out);
				else hipLaunchKernelGGL((k_chase<8, 8>), dim3(blocks), dim3(256), 0, 0, d_table, n_rec / 2u, n_hot / 2u, (uint32_t)hot, steps, d_out);   // 128-B records, line aligned
				CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
			}
			const double fetches = (double)blocks * 256.0 * steps;
			printf("  hot %3d/256  %d x 16 B per record%s: %7.3f ms  %6.1f G records/s  %6.2f TB/s of bytes asked for, %6.2f TB/s if every cold fetch moves a 128-B line\n",
				hot, pieces, pieces == 8 ? " (128-B records)" : " (64-B records) ", best, fetches / best * 1e-6, fetches * pieces * 16.0 / best * 1e-9, fetches * (256 - hot) / 256.0 * 128.0 / best * 1e-9);
		}
	}
	// occupancy sweep, cold fetches only, one 16-B piece per record
	for (int nb = 256; nb <= blocks * 2; nb *= 2) {
		for (int chains = 1; chains <= 2; chains++) {
			float best = 1e30f;
			for (int rep = 0; rep < 4; rep++) {
				CK(hipEventRecord(e0, 0));
				if (chains == 1) hipLaunchKernelGGL((k_chase<1, 4>), dim3(nb), dim3(256), 0, 0, d_table, n_rec, n_hot, 0u, steps, d_out);
				else hipLaunchKernelGGL(k_chase2, dim3(nb), dim3(256), 0, 0, d_table, n_rec, steps, d_out);
				CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
			}
			const double fetches = (double)nb * 256.0 * steps * chains;
			printf("  %5d blocks (%4.1f waves per CU), %d chain(s) per lane: %7.3f ms  %6.1f G records/s  (%.2f us per dependent fetch)\n", nb, nb * 4.0 / 256.0, chains, best, fetches / best * 1e-6, best * 1e3 / steps);
		}
	}
	return 0;
}
