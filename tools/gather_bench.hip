// What random 64-byte gathers reach on this GPU: the ceiling the seed lookup (one table line per probed window) is priced against.
// Every lane draws addresses from a counter-based hash, loads one 4-byte word of a random 64-byte line of a T-byte table, U loads
// in flight per lane before the first is used.  Prints GB/s of lines fetched (64 B each) for a few table sizes and depths.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
template <int U>
__global__ void __launch_bounds__(256) k_gather(const uint32_t *tab, uint64_t lines_mask, uint32_t iters, uint32_t *sink)
{
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
	for (uint32_t it = 0; it < iters; it++) {
		uint32_t v[U];
#pragma unroll
		for (int u = 0; u < U; u++) { const uint64_t h = mix(tid * 0x9e3779b97f4a7c15ULL + (uint64_t)it * U + u); v[u] = tab[((h & lines_mask) << 4) + ((h >> 60) & 15u)]; }
#pragma unroll
		for (int u = 0; u < U; u++) acc += v[u];
	}
	if (acc == 0x12345678u) sink[0] = acc;
}
template <int U> static void run(const uint32_t *tab, uint64_t bytes, uint32_t *sink)
{
	const uint64_t lines = bytes / 64;
	const uint32_t blocks = 256 * 32, iters = 4096 / U;
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k_gather<U>, dim3(blocks), dim3(256), 0, 0, tab, lines - 1, iters / 4, sink);
	CHK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL(k_gather<U>, dim3(blocks), dim3(256), 0, 0, tab, lines - 1, iters, sink);
	CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
	float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
	const double n = (double)blocks * 256 * iters * U;
	printf("table %6.0f MiB  %2d loads in flight per lane: %7.1f G lines/s = %6.0f GB/s of 64-byte lines (%.1f ms)\n", bytes / 1048576.0, U, n / ms / 1e6, n * 64 / ms / 1e6, ms);
}
// `gather_bench MiB`: that table size only (a power of two), depths 4 / 8 / 16 (bench.py prints the best beside the seed lookup's rate)
int main(int argc, char **argv)
{
	uint32_t *tab = nullptr, *sink = nullptr;
	if (argc > 1) {
		const uint64_t b = (uint64_t)atoll(argv[1]) << 20;
		if (b < (1u << 20) || (b & (b - 1))) { fprintf(stderr, "usage: gather_bench [MiB, a power of two]\n"); return 2; }
		CHK(hipMalloc((void **)&tab, b)); CHK(hipMalloc((void **)&sink, 4)); CHK(hipMemset(tab, 1, b));
		run<4>(tab, b, sink); run<8>(tab, b, sink); run<16>(tab, b, sink);
		return 0;
	}
	const uint64_t maxb = 16ULL << 30;
	CHK(hipMalloc((void **)&tab, maxb)); CHK(hipMalloc((void **)&sink, 4));
	CHK(hipMemset(tab, 1, maxb));
	const uint64_t sizes[] = {64ULL << 20, 512ULL << 20, 2ULL << 30, 4ULL << 30, 16ULL << 30};
	for (uint64_t b : sizes) { run<1>(tab, b, sink); run<4>(tab, b, sink); run<8>(tab, b, sink); run<16>(tab, b, sink); }
	return 0;
}
