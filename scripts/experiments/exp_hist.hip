// Experiment behind the rs_hist notes in DESIGN.md: the histogram pass with block ranges vs interleaved tiles, with and
// without LDS atomics, over uniform random keys.  hipcc --offload-arch=gfx950 -O3 exp_hist.hip -o exp_hist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int T = 512, RADIX = 256, WAVES = T / 64;
// A: current mapping (block ranges), LDS atomics per wave
template <bool ATOMICS, bool INTERLEAVE, int NH>
__global__ __launch_bounds__(T) void hist(const unsigned long long *__restrict__ keys, uint32_t n, int shift, uint32_t tpb, uint32_t tile, uint32_t *__restrict__ out) {
	__shared__ uint32_t h[NH][RADIX];
	for (int j = threadIdx.x; j < NH * RADIX; j += T) (&h[0][0])[j] = 0;
	__syncthreads();
	const uint32_t w = (threadIdx.x / 64) % NH;
	unsigned long long acc = 0;
	const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(keys);
	for (uint32_t t = 0; t < tpb; ++t) {
		const uint64_t tl = INTERLEAVE ? (uint64_t(t) * gridDim.x + blockIdx.x) : (uint64_t(blockIdx.x) * tpb + t);
		const uint64_t base = tl * tile;
		if (base >= n) break;
		// tile = 4096 keys = 2048 pairs = 4 pairs per thread
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint64_t p = (base >> 1) + q * T + threadIdx.x;
			if (2 * p + 1 < n) {
				const ulonglong2 a = k2[p];
				if (ATOMICS) { atomicAdd(&h[w][uint32_t(a.x >> shift) & 0xFFu], 1u); atomicAdd(&h[w][uint32_t(a.y >> shift) & 0xFFu], 1u); }
				else acc ^= a.x ^ a.y;
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < RADIX) {
		uint32_t s = 0;
		for (int k = 0; k < NH; ++k) s += h[k][threadIdx.x];
		out[threadIdx.x * gridDim.x + blockIdx.x] = s + uint32_t(acc);
	} else if (!ATOMICS && acc == 0x1234567ull) out[0] = 1;
}
int main() {
	const uint32_t n = 100000000u, tile = 4096;
	unsigned long long *d; uint32_t *o;
	CK(hipMalloc(&d, size_t(n) * 8)); CK(hipMalloc(&o, 256 * 32768 * 4));
	std::vector<unsigned long long> h(n);
	unsigned long long x = 88172645463325252ull;
	for (uint32_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x; }
	CK(hipMemcpy(d, h.data(), size_t(n) * 8, hipMemcpyHostToDevice));
	const uint32_t n_tiles = (n + tile - 1) / tile;
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	auto run = [&](const char *name, auto kern, uint32_t nblocks) {
		const uint32_t tpb = (n_tiles + nblocks - 1) / nblocks;
		for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(T), 0, 0, d, n, 8, tpb, tile, o);
		hipEventRecord(a);
		for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(T), 0, 0, d, n, 8 + (i % 4) * 8, tpb, tile, o);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
		printf("%-44s blocks %5u  %.3f ms  %.0f GB/s\n", name, nblocks, ms, 8.0 * n / ms / 1e6);
	};
	for (uint32_t nb : {256u, 512u, 1024u, 2048u, 4096u}) {
		run("ranges, per-wave LDS atomics", hist<true, false, WAVES>, nb);
		run("ranges, one LDS histogram per block", hist<true, false, 1>, nb);
		run("ranges, no atomics (pure read)", hist<false, false, 1>, nb);
		run("interleaved tiles, per-wave atomics", hist<true, true, WAVES>, nb);
		run("interleaved tiles, no atomics", hist<false, true, 1>, nb);
	}
	return 0;
}
