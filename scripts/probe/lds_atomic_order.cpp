// probe: are LDS atomic adds of one wave instruction to the SAME address applied in lane order on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned *digit, unsigned *rank_out, int rounds) {
	__shared__ unsigned cnt[256];
	const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	for (int r = 0; r < rounds; ++r) {
		for (unsigned j = threadIdx.x; j < 256; j += blockDim.x) cnt[j] = 0;
		__syncthreads();
		if (w == 0) {
			const unsigned d = digit[r * 64 + lane];
			rank_out[r * 64 + lane] = atomicAdd(&cnt[d], 1u);
		}
		__syncthreads();
	}
}
int main() {
	const int rounds = 4096;
	std::vector<unsigned> h(rounds * 64), out(rounds * 64);
	unsigned x = 12345;
	for (int r = 0; r < rounds; ++r) {
		const unsigned nd = 1u + unsigned(r % 64);          // number of distinct digits in this round: 1 .. 64
		for (int l = 0; l < 64; ++l) { x = x * 1664525u + 1013904223u; h[r * 64 + l] = (x >> 16) % nd; }
	}
	unsigned *d_in, *d_out;
	hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, h.size() * 4);
	hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_in, d_out, rounds);
	hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
	long bad = 0;
	for (int r = 0; r < rounds; ++r) {
		unsigned seen[64] = {0};
		for (int l = 0; l < 64; ++l) { const unsigned d = h[r * 64 + l]; if (out[r * 64 + l] != seen[d]) ++bad; ++seen[d]; }
	}
	std::printf("lane-ordered: %s (%ld of %d ranks out of lane order)\n", bad ? "NO" : "yes", bad, rounds * 64);
	return 0;
}
