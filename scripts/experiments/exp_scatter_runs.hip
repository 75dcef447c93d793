// Experiment behind the scatter-pass notes in DESIGN.md: the memory pattern of one LSD pass without any ranking work --
// every tile of T keys is read contiguously and written as 256 runs of T/256 keys, run d going to the next free place of
// region d (regions are filled in tile order, like the real pass on uniformly distributed digits).
// hipcc --offload-arch=gfx950 -O3 exp_scatter_runs.hip -o exp_scatter_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(512) void runs(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out, uint32_t n,
                                            uint32_t T, uint32_t tiles_per_block) {
	const uint32_t run = T / 256, region = n / 256;
	for (uint32_t t = 0; t < tiles_per_block; ++t) {
		const uint32_t tile = blockIdx.x * tiles_per_block + t;
		const uint64_t base = uint64_t(tile) * T;
		if (base + T > n) return;
		for (uint32_t p = threadIdx.x; p < T; p += 512) {
			const uint32_t d = p / run;
			out[uint64_t(d) * region + uint64_t(tile) * run + (p - d * run)] = in[base + p];
		}
	}
}
int main() {
	const uint32_t n = 1u << 27;   // 134 M keys = 1 GiB
	unsigned long long *a, *b;
	if (hipMalloc(&a, size_t(n) * 8) != hipSuccess || hipMalloc(&b, size_t(n) * 8) != hipSuccess) return 1;
	(void)hipMemset(a, 1, size_t(n) * 8);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for (uint32_t T : {2048u, 4096u, 8192u, 16384u, 65536u, 262144u}) {
		const uint32_t n_tiles = n / T;
		for (uint32_t nblocks : {1024u, 4096u}) {
			if (nblocks > n_tiles) continue;
			const uint32_t tpb = n_tiles / nblocks;
			for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(runs, dim3(nblocks), dim3(512), 0, 0, a, b, n, T, tpb);
			(void)hipEventRecord(e0);
			for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(runs, dim3(nblocks), dim3(512), 0, 0, a, b, n, T, tpb);
			(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
			printf("tile %6u keys (runs of %5u B), %4u blocks: %.3f ms  %.0f GB/s (read + write)\n", T, T / 256 * 8, nblocks, ms, 16.0 * n / ms / 1e6);
		}
	}
	return 0;
}
