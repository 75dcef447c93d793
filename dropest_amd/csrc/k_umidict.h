// k_umidict.h -- a dictionary of the stream's UMIs for molecule keys that do not fit one 64-bit word.
//
// Reference: Estimation/StringIndexer.cpp:10-18 hands every UMI string a size_t index and Gene keeps std::map<umi index, UMI>
// (Gene.h:19): no width limit.  Here the molecule key is cell | gene | umi in ONE u64 with the UMI's own 2-bit code as its
// field -- a 24-base UMI beside 2^16 genes already needs 64 bits before any cell bit.  When that happens the UMI field
// becomes the UMI's RANK among the distinct clean UMIs of the gene-bearing reads (at most n_reads < 2^32 of them):
//   umi_dict_fill      the clean UMIs of gene-bearing reads (everything else -> ~0, which sorts last)
//   (radix sort, k_radix.h)
//   umi_dict_count / umi_dict_write   distinct values, ascending: the dictionary
//   umi_dict_rank      per read: rank of its UMI (binary search), escaped UMIs (strings with N) keep their escape id
// Ranks ascend with the codes, so every order that hangs on the key's UMI field (the molecule rows of a (cell, gene) pair,
// the rows ResultsPrinter walks) is the one the plain layout gives.  The key-building kernels read the ranked column in place
// of the UMI column and are otherwise unchanged (KeyLayout: strip mask ~0, escapes behind 2^bits).
// Integer work, HBM-bound, off the common path; no MFMA.
#pragma once

#include "k_cbhash.h"

namespace dropest {

constexpr int UD_T = 256, UD_PER = 16;   // a workgroup looks at 4 096 sorted values

__global__ __launch_bounds__(256) void umi_dict_fill_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene,
                                                            uint32_t n, unsigned long long *__restrict__ out) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
		const unsigned long long u = umi[r];
		out[r] = (gene[r] == NO_GENE || (u & ESCAPE_BIT)) ? ~0ull : u;
	}
}

// heads of the sorted values (a value that differs from its predecessor and is not the ~0 filler)
__device__ inline uint32_t ud_heads_of_thread(const unsigned long long *__restrict__ k, uint32_t n, uint32_t first, uint32_t &mask) {
	mask = 0;
	if (first >= n) return 0;
	unsigned long long prev = first ? k[first - 1] : ~0ull;
	uint32_t c = 0;
	for (int j = 0; j < UD_PER; ++j) {
		const uint32_t i = first + uint32_t(j);
		if (i >= n) break;
		const unsigned long long v = k[i];
		if (v != ~0ull && (i == 0 || v != prev)) { mask |= 1u << j; ++c; }
		prev = v;
	}
	return c;
}

__global__ __launch_bounds__(UD_T) void umi_dict_count_kernel(const unsigned long long *__restrict__ k, uint32_t n, uint32_t *__restrict__ block_count) {
	__shared__ uint32_t sum;
	if (threadIdx.x == 0) sum = 0;
	__syncthreads();
	uint32_t mask;
	const uint32_t c = ud_heads_of_thread(k, n, (blockIdx.x * UD_T + threadIdx.x) * UD_PER, mask);
	if (c) atomicAdd(&sum, c);
	__syncthreads();
	if (threadIdx.x == 0) block_count[blockIdx.x] = sum;
}

__global__ __launch_bounds__(UD_T) void umi_dict_write_kernel(const unsigned long long *__restrict__ k, uint32_t n, const uint32_t *__restrict__ block_base,
                                                              unsigned long long *__restrict__ dict) {
	__shared__ uint32_t pre[UD_T];
	uint32_t mask;
	const uint32_t first = (blockIdx.x * UD_T + threadIdx.x) * UD_PER;
	const uint32_t c = ud_heads_of_thread(k, n, first, mask);
	pre[threadIdx.x] = c;
	__syncthreads();
	for (int d = 1; d < UD_T; d <<= 1) {   // inclusive scan over the workgroup's 256 counts
		const uint32_t add = threadIdx.x >= uint32_t(d) ? pre[threadIdx.x - d] : 0u;
		__syncthreads();
		pre[threadIdx.x] += add;
		__syncthreads();
	}
	uint32_t at = block_base[blockIdx.x] + pre[threadIdx.x] - c;
	for (int j = 0; j < UD_PER; ++j)
		if (mask >> j & 1u) dict[at++] = k[first + uint32_t(j)];
}

__global__ __launch_bounds__(256) void umi_dict_rank_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene, uint32_t n,
                                                            const unsigned long long *__restrict__ dict, uint32_t n_dict,
                                                            unsigned long long *__restrict__ ranked) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
		const unsigned long long u = umi[r];
		unsigned long long out = u;                     // escaped: the escape id travels as it is
		if (gene[r] == NO_GENE) out = (u & ESCAPE_BIT) ? u : 0ull;   // (its UMI never enters a key: the field carries the chromosome or 0)
		else if (!(u & ESCAPE_BIT)) {
			uint32_t lo = 0, hi = n_dict;               // first entry >= u; u is in the dictionary
			while (lo < hi) {
				const uint32_t mid = lo + ((hi - lo) >> 1);
				if (dict[mid] < u) lo = mid + 1; else hi = mid;
			}
			out = lo;
		}
		ranked[r] = out;
	}
}

}  // namespace dropest
