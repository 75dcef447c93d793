// k_collisions.h -- Tools::CollisionsAdjuster on the device (Tools/CollisionsAdjuster.cpp:12-49) and the UMI
// distribution that feeds it (CellsDataContainer::umi_distribution, CellsDataContainer.cpp:182-197).
//
// adjusted_size[s] for s = 1..S:   total = s + floor(sum_collisions);  for every observed UMI i:
//   neg_prod[i] *= (1 - p[i]) ^ (total - last_total);   new_prob = SUM_i p[i] * (1 - neg_prod[i]);
//   sum_collisions += 1 / (1 - new_prob) - 1;   adjusted_size[s] = lround(s + sum_collisions)
// The recurrence over s is sequential (two small launches per s, scalars stay on the device); the work per s is a
// map + reduction over the UMIs.  Double precision, no FMA contraction (the library is built with
// -ffp-contract=off); the reduction order is fixed: per thread a strided left-to-right sum, then a wave shuffle
// tree, then waves and blocks in index order.
#pragma once

#include "util.h"

namespace dropest {

struct CollisionState {
	double sum_collisions;
	unsigned long long last_total;
	unsigned long long delta;      // exponent of the current step
};

__device__ inline double dev_fpow(double base, unsigned long long exp) {   // Tools::fpow, UtilFunctions.cpp:13-30
	if (exp == 1) return base;
	double result = 1;
	while (exp) {
		if (exp & 1) result *= base;
		exp >>= 1;
		base *= base;
	}
	return result;
}

constexpr int CA_BLOCKS = 256, CA_THREADS = 256;

__global__ __launch_bounds__(CA_THREADS) void collisions_step_kernel(const double *__restrict__ p, double *__restrict__ neg_prod,
                                                                     unsigned long long n, const CollisionState *st,
                                                                     double *__restrict__ partial) {
	__shared__ double wave_sum[CA_THREADS / 64];
	const unsigned long long delta = st->delta;
	double acc = 0;
	for (unsigned long long i = (unsigned long long)blockIdx.x * CA_THREADS + threadIdx.x; i < n;
	     i += (unsigned long long)CA_BLOCKS * CA_THREADS) {
		const double np = neg_prod[i] * dev_fpow(1 - p[i], delta);
		neg_prod[i] = np;
		acc += p[i] * (1 - np);
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
	if (lane_id() == 0) wave_sum[wave_id()] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		double s = 0;
		for (int w = 0; w < CA_THREADS / 64; ++w) s += wave_sum[w];
		partial[blockIdx.x] = s;
	}
}

// one thread: finishes step s (sum of the block partials in index order) and prepares the exponent of step s + 1
__global__ void collisions_finish_kernel(const double *__restrict__ partial, CollisionState *st, unsigned long long s,
                                         unsigned long long *__restrict__ adjusted) {
	if (threadIdx.x || blockIdx.x) return;
	if (s > 0) {
		double new_prob = 0;
		for (int b = 0; b < CA_BLOCKS; ++b) new_prob += partial[b];
		const double collision_num = 1.0 / (1.0 - new_prob) - 1.0;
		st->sum_collisions += collision_num;
		adjusted[s - 1] = (unsigned long long)lround(double(s) + st->sum_collisions);
	}
	const unsigned long long next_total = (s + 1) + (unsigned long long)st->sum_collisions;
	st->delta = next_total - st->last_total;
	st->last_total = next_total;
}

// molecules of filtered cells -> their UMI codes (for the distribution); others are dropped by the flag
__global__ __launch_bounds__(256) void emit_filtered_umis_kernel(const unsigned long long *__restrict__ mol_key, uint32_t n_mol,
                                                                 int umi_bits, int gene_bits, unsigned long long gene_none,
                                                                 const uint32_t *__restrict__ cell_flag,
                                                                 unsigned long long *__restrict__ out, uint32_t *__restrict__ count) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	bool keep = false;
	unsigned long long u = 0;
	if (i < n_mol) {
		const unsigned long long k = mol_key[i];
		const unsigned long long cg = k >> umi_bits;
		keep = (cg & gene_none) != gene_none && cell_flag[uint32_t(cg >> gene_bits)];
		u = k & ((1ull << umi_bits) - 1ull);
	}
	const unsigned long long m = __ballot(keep);
	uint32_t base = 0;
	if (lane_id() == 0 && m) base = atomicAdd(count, uint32_t(__popcll(m)));
	base = __shfl(base, 0, 64);
	if (keep) out[base + __popcll(m & ((1ull << lane_id()) - 1ull))] = u;
}

// sorted UMI codes -> (code, count) runs
struct UmiRuns {
	static constexpr int ITEMS = 8;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 1;
	static constexpr unsigned OR_MASK = 0;
	const unsigned long long *keys;
	unsigned long long *run_key;
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const { return keys[i]; }
	__device__ void load(uint32_t, uint32_t (&v)[NV]) const { v[0] = 1; }
	__device__ void write_head(uint32_t o, uint32_t, unsigned long long k) const { run_key[o] = k; }
};

}  // namespace dropest
