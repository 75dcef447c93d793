// k_mergepath.h -- re-sorting a molecule table after a FEW of its keys changed (CB merge: only the molecules of the
// merged cells get a new key, a few percent of the table at BASELINE sizes).  The unchanged rows are still sorted, so
// instead of eight radix passes over everything (256 B per row) the table is split into the unchanged rows (stable
// compaction) and the changed ones, the small changed part is radix-sorted, and the two sorted sequences are merged
// in ONE pass with the merge-path partition (one binary search along a cross diagonal per tile, one per thread inside
// the tile's LDS copy): 44 B per row.  Output = (key, old row) pairs ordered by key, what the radix sort would give up
// to the order of equal keys (which the fold that follows does not depend on).
#pragma once

#include "util.h"

namespace dropest {

constexpr int MP_THREADS = 256, MP_ITEMS = 8, MP_TILE = MP_THREADS * MP_ITEMS;

// changed rows per tile
// The new key of a row: read from new_key, or -- a barcode merge, where only the cell field changes -- made on the fly from the old key and
// the cell -> target table (`remap`): the re-keyed array is then never written or read (12 + 16 bytes per molecule row less).
struct MpRekey {
	const unsigned long long *new_key; const uint32_t *remap; int cell_shift;
	__device__ unsigned long long operator()(uint32_t i, unsigned long long old) const {
		if (!remap) return new_key[i];
		return ((unsigned long long)remap[uint32_t(old >> cell_shift)] << cell_shift) | (old & ((1ull << cell_shift) - 1ull));
	}
};
// key_or_and (remap mode only): OR / AND of all new keys, for the radix passes of the changed rows
__global__ __launch_bounds__(MP_THREADS) void mp_split_count_kernel(const unsigned long long *__restrict__ old_key, MpRekey rk, uint32_t n,
                                                                    uint32_t sorted_rows, uint32_t *__restrict__ tile_changed, unsigned long long *key_or_and) {
	__shared__ uint32_t scratch[MP_THREADS / 64 + 1];
	__shared__ unsigned long long w_or[MP_THREADS / 64], w_and[MP_THREADS / 64];
	const uint32_t t0 = blockIdx.x * MP_TILE;
	uint32_t c = 0;
	unsigned long long k_or = 0, k_and = ~0ull;
#pragma unroll
	for (int j = 0; j < MP_ITEMS; ++j) {
		const uint32_t i = t0 + j * MP_THREADS + threadIdx.x;
		if (i < n) {
			const unsigned long long o = old_key[i], nk = rk(i, o);
			c += (o != nk) || i >= sorted_rows;   // rows appended behind the sorted part count as changed
			k_or |= nk; k_and &= nk;
		}
	}
	uint32_t total;
	block_excl_scan_u32<MP_THREADS>(c, scratch, total);
	if (threadIdx.x == 0) tile_changed[blockIdx.x] = total;
	if (key_or_and) {
		k_or = wave_reduce_or_u64(k_or); k_and = wave_reduce_and_u64(k_and);
		if (lane_id() == 0) { w_or[threadIdx.x >> 6] = k_or; w_and[threadIdx.x >> 6] = k_and; }
		__syncthreads();
		if (threadIdx.x == 0) {
			for (int q = 1; q < MP_THREADS / 64; ++q) { w_or[0] |= w_or[q]; w_and[0] &= w_and[q]; }
			// (2e5 workgroups on two words: same-address atomics run one after the other -- 3.6 ms at C3 size -- so a workgroup first
			// looks whether its bits would change anything; a stale look only costs an atomic that changes nothing)
			const unsigned long long cur_or = __atomic_load_n(&key_or_and[0], __ATOMIC_RELAXED), cur_and = __atomic_load_n(&key_or_and[1], __ATOMIC_RELAXED);
			if (w_or[0] & ~cur_or) atomicOr(&key_or_and[0], w_or[0]);
			if (~w_and[0] & cur_and) atomicAnd(&key_or_and[1], w_and[0]);
		}
	}
}

// stable split: unchanged rows -> (a_key, a_row) in order, changed rows -> (b_key, b_row) in order
__global__ __launch_bounds__(MP_THREADS) void mp_split_write_kernel(const unsigned long long *__restrict__ old_key, MpRekey rk, uint32_t n,
                                                                    uint32_t sorted_rows, const uint32_t *__restrict__ tile_prefix,
                                                                    unsigned long long *__restrict__ a_key, uint32_t *__restrict__ a_row,
                                                                    unsigned long long *__restrict__ b_key, uint32_t *__restrict__ b_row) {
	__shared__ uint32_t scratch[MP_THREADS / 64 + 1];
	const uint32_t t0 = blockIdx.x * MP_TILE, r0 = t0 + threadIdx.x * MP_ITEMS;   // blocked: each thread owns consecutive rows
	unsigned long long k[MP_ITEMS];
	uint32_t flags = 0, c = 0;
#pragma unroll
	for (int j = 0; j < MP_ITEMS; ++j) {
		const uint32_t i = r0 + j;
		if (i < n) { const unsigned long long o = old_key[i]; k[j] = rk(i, o); if (o != k[j] || i >= sorted_rows) { flags |= 1u << j; ++c; } }
	}
	uint32_t total;
	const uint32_t ex = block_excl_scan_u32<MP_THREADS>(c, scratch, total);
	uint32_t b_at = tile_prefix[blockIdx.x] + ex;          // changed rows before this thread's first row
	uint32_t a_at = r0 - b_at;                             // unchanged rows before it
#pragma unroll
	for (int j = 0; j < MP_ITEMS; ++j) {
		const uint32_t i = r0 + j;
		if (i >= n) break;
		if (flags & (1u << j)) { b_key[b_at] = k[j]; b_row[b_at] = i; ++b_at; }
		else { a_key[a_at] = k[j]; a_row[a_at] = i; ++a_at; }
	}
}

// first index a in [lo, hi] such that taking a elements of A and (diag - a) of B is a valid merge prefix (A first on ties)
__device__ inline uint32_t mp_search(const unsigned long long *__restrict__ A, uint32_t na, const unsigned long long *__restrict__ B,
                                     uint32_t nb, uint32_t diag) {
	uint32_t lo = diag > nb ? diag - nb : 0u, hi = diag < na ? diag : na;
	while (lo < hi) {
		const uint32_t a = lo + ((hi - lo) >> 1);
		// a is too small if A[a] <= B[diag - a - 1]  (that A element must come before the B element already taken)
		if (A[a] <= B[diag - a - 1]) lo = a + 1; else hi = a;
	}
	return lo;
}

__global__ __launch_bounds__(256) void mp_partition_kernel(const unsigned long long *__restrict__ A, uint32_t na,
                                                           const unsigned long long *__restrict__ B, uint32_t nb, uint32_t n_tiles,
                                                           uint32_t *__restrict__ a_start) {
	const uint32_t t = blockIdx.x * 256 + threadIdx.x;
	if (t > n_tiles) return;
	const unsigned long long total = (unsigned long long)na + nb;
	const unsigned long long d = (unsigned long long)t * MP_TILE;
	a_start[t] = mp_search(A, na, B, nb, uint32_t(d < total ? d : total));
}

__global__ __launch_bounds__(MP_THREADS) void mp_merge_kernel(const unsigned long long *__restrict__ A, const uint32_t *__restrict__ Av, uint32_t na,
                                                              const unsigned long long *__restrict__ B, const uint32_t *__restrict__ Bv, uint32_t nb,
                                                              const uint32_t *__restrict__ a_start, unsigned long long *__restrict__ out_key,
                                                              uint32_t *__restrict__ out_val) {
	__shared__ unsigned long long sk[MP_TILE];
	__shared__ uint32_t sv[MP_TILE];
	const unsigned long long total = (unsigned long long)na + nb;
	const unsigned long long d0 = (unsigned long long)blockIdx.x * MP_TILE;
	const uint32_t d1 = uint32_t(d0 + MP_TILE < total ? d0 + MP_TILE : total);
	const uint32_t a0 = a_start[blockIdx.x], a1 = a_start[blockIdx.x + 1];
	const uint32_t b0 = uint32_t(d0) - a0, b1 = d1 - a1;
	const uint32_t ca = a1 - a0, cb = b1 - b0;           // ca + cb = outputs of this tile (<= MP_TILE)
	for (uint32_t i = threadIdx.x; i < ca; i += MP_THREADS) { sk[i] = A[a0 + i]; sv[i] = Av[a0 + i]; }
	for (uint32_t i = threadIdx.x; i < cb; i += MP_THREADS) { sk[ca + i] = B[b0 + i]; sv[ca + i] = Bv[b0 + i]; }
	__syncthreads();
	const uint32_t count = ca + cb;
	const uint32_t my0 = threadIdx.x * MP_ITEMS;
	if (my0 >= count) return;
	uint32_t ia = mp_search(sk, ca, sk + ca, cb, my0), ib = my0 - ia;
	const uint32_t out0 = uint32_t(d0) + my0;
#pragma unroll
	for (int j = 0; j < MP_ITEMS; ++j) {
		if (my0 + j >= count) break;
		const bool take_a = ib >= cb || (ia < ca && sk[ia] <= sk[ca + ib]);
		const uint32_t s = take_a ? ia : ca + ib;
		out_key[out0 + j] = sk[s]; out_val[out0 + j] = sv[s];
		if (take_a) ++ia; else ++ib;
	}
}

}  // namespace dropest
