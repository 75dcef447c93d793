// k_radix.h -- LSD radix sort of (u64 key, u32 value) records, 8-bit digits, hand-written for gfx950.
//
// This is the "radix partition" of the UMI de-duplication: records keyed (cell id | gene | UMI) are brought
// into key order so that molecules, (cell, gene) groups and cells become contiguous runs (k_segreduce.h
// then reduces the runs).  It replaces the per-read red-black-tree descents of the reference
// (Cell::genes().emplace / Gene::_umis.emplace, Estimation/CellsDataContainer.cpp:356-364, Gene.cpp:17-24).
//
// Per pass, three launches on one stream:
//   rs_hist     one histogram PER BLOCK (each block owns a contiguous range of tiles), 8 B/record read
//   rs_scan     row-wise exclusive scan of the [256][blocks] histogram + scan of the 256 digit totals
//   rs_scatter  each block walks its tiles carrying 256 running output cursors in LDS:
//               wave-level multisplit by ballot (8 ballots per key, 64-lane waves), ranks combined over
//               waves through LDS, keys re-ordered by digit in an LDS tile so that the global writes of
//               one digit are contiguous, then written out; (8 + VB) B read + (8 + VB) B written per record,
//               VB = bytes of the value that travels with the key (0, 1 or 4)
// HBM-bound integer work; no MFMA.  The sort is stable (required between LSD passes).
#pragma once

#include "util.h"

namespace dropest {

constexpr int RS_THREADS = 512;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;   // 4096 records per tile
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_RADIX = 256;

// RB = bits of the digit (8, or 9 for the passes radix_sort widens to save a whole pass): RADIX = 2^RB counters
template <int RB>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const unsigned long long *__restrict__ keys, uint32_t n,
                                                             int shift, uint32_t tiles_per_block, uint32_t tile,
                                                             uint32_t *__restrict__ hist /* [RADIX][gridDim.x] */) {
	constexpr uint32_t RADIX = 1u << RB, DMASK = RADIX - 1u;
	static_assert(RADIX <= RS_THREADS, "one thread per digit writes the block's histogram");
	__shared__ uint32_t h[RS_WAVES][RADIX];
	for (int j = threadIdx.x; j < RS_WAVES * int(RADIX); j += RS_THREADS) (&h[0][0])[j] = 0;
	__syncthreads();
	const uint64_t begin = uint64_t(blockIdx.x) * tiles_per_block * tile;
	uint64_t end = begin + uint64_t(tiles_per_block) * tile;
	if (end > n) end = n;
	const uint32_t w = wave_id();
	// Digits are often heavily skewed (the high bits of the cell id: most reads belong to a few thousand cells): 64 lanes
	// adding to ONE LDS address serialise.  The lanes that share the first lane's digit are counted with a ballot and
	// added once; the others add individually.  (Measured: 1.85 -> 1.46 ms per C2 step; peeling a second digit
	// changes nothing, matching all groups with 8 ballots per key as the scatter does is slower, 1.8 ms.)
	const unsigned long long lt_mask = (1ull << lane_id()) - 1ull;
	auto add = [&](unsigned long long k) {
		const uint32_t d = uint32_t(k >> shift) & DMASK;
		const uint32_t lead = __builtin_amdgcn_readfirstlane(d);
		const unsigned long long same = __ballot(d == lead);
		if (d != lead) atomicAdd(&h[w][d], 1u);
		else if ((same & lt_mask) == 0) atomicAdd(&h[w][lead], uint32_t(__popcll(same)));
	};
	// 16-byte loads (two keys each), four of them in flight per thread; `begin` is a multiple of the tile size, so the
	// pairs are aligned
	uint64_t i = begin + 2ull * threadIdx.x;
	const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(keys);
	for (; i + 6ull * RS_THREADS + 1 < end; i += 8ull * RS_THREADS) {
		const ulonglong2 a = k2[i >> 1], b = k2[(i >> 1) + RS_THREADS], c = k2[(i >> 1) + 2ull * RS_THREADS], d = k2[(i >> 1) + 3ull * RS_THREADS];
		add(a.x); add(a.y); add(b.x); add(b.y); add(c.x); add(c.y); add(d.x); add(d.y);
	}
	// (the tails are wave-divergent: plain atomics)
	for (; i + 1 < end; i += 2ull * RS_THREADS) {
		const ulonglong2 a = k2[i >> 1];
		atomicAdd(&h[w][uint32_t(a.x >> shift) & DMASK], 1u); atomicAdd(&h[w][uint32_t(a.y >> shift) & DMASK], 1u);
	}
	if (i < end) atomicAdd(&h[w][uint32_t(keys[i] >> shift) & DMASK], 1u);   // odd tail (end == n)
	__syncthreads();
	if (threadIdx.x < RADIX) {
		uint32_t s = 0;
#pragma unroll
		for (int k = 0; k < RS_WAVES; ++k) s += h[k][threadIdx.x];
		hist[threadIdx.x * gridDim.x + blockIdx.x] = s;
	}
}

// block d scans row d of hist in place (exclusive) and stores the row total
__global__ __launch_bounds__(256) void rs_scan_rows_kernel(uint32_t *__restrict__ hist, uint32_t nblocks,
                                                           uint32_t *__restrict__ row_total) {
	__shared__ uint32_t scratch[256 / 64 + 1];
	uint32_t *row = hist + size_t(blockIdx.x) * nblocks;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < nblocks; base += 256) {
		uint32_t i = base + threadIdx.x;
		uint32_t v = i < nblocks ? row[i] : 0;
		uint32_t total;
		uint32_t ex = block_excl_scan_u32<256>(v, scratch, total);
		if (i < nblocks) row[i] = carry + ex;
		carry += total;
	}
	if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

template <int RADIX>
__global__ __launch_bounds__(RADIX) void rs_scan_totals_kernel(const uint32_t *__restrict__ row_total,
                                                               uint32_t *__restrict__ digit_base) {
	__shared__ uint32_t scratch[RADIX / 64 + 1];
	uint32_t total;
	digit_base[threadIdx.x] = block_excl_scan_u32<RADIX>(row_total[threadIdx.x], scratch, total);
}

template <int VB> struct RsVal;
template <> struct RsVal<0> { typedef uint8_t T; };    // (unused)
template <> struct RsVal<1> { typedef uint8_t T; };
template <> struct RsVal<4> { typedef uint32_t T; };

// THREADS x ITEMS records per tile.  PREFETCH: the next tile's records are loaded into registers before the
// current tile is ranked, so the HBM latency of tile t+1 hides behind the LDS / ballot work of tile t.
// Full tiles (all but possibly the last one of the array) run without per-record bounds checks.
// VB: bytes of the value carried with each key (0 = keys only).
template <int THREADS, int ITEMS, bool PREFETCH, int VB, int RB = 8>
__global__ __launch_bounds__(THREADS) void rs_scatter_kernel_t(const unsigned long long *__restrict__ keys,
                                                               const void *__restrict__ vals_,
                                                               unsigned long long *__restrict__ okeys,
                                                               void *__restrict__ ovals_, uint32_t n, int shift,
                                                               uint32_t tiles_per_block,
                                                               const uint32_t *__restrict__ hist,
                                                               const uint32_t *__restrict__ digit_base) {
	typedef typename RsVal<VB>::T val_t;
	const val_t *__restrict__ vals = static_cast<const val_t *>(vals_);
	val_t *__restrict__ ovals = static_cast<val_t *>(ovals_);
	constexpr uint32_t TILE = THREADS * ITEMS, WAVES = THREADS / 64;
	constexpr uint32_t RS_RADIX = 1u << RB, DMASK = RS_RADIX - 1u;   // (shadows the 8-bit constant of the namespace)
	static_assert(THREADS >= RS_RADIX, "one thread per digit is needed for the digit scan");
	__shared__ uint32_t wcnt[WAVES][RS_RADIX];      // per-wave digit counts -> per-wave digit offsets
	__shared__ uint32_t tcnt[RS_RADIX];             // digit counts of the tile
	__shared__ uint32_t tstart[RS_RADIX];           // digit start inside the re-ordered tile
	__shared__ uint32_t goff[RS_RADIX];             // running global cursor per digit (this block)
	__shared__ uint32_t gdelta[RS_RADIX];           // goff - tstart: global position = gdelta[digit] + position in tile
	__shared__ uint32_t scratch[THREADS / 64 + 1];
	__shared__ unsigned long long sk[TILE];
	__shared__ val_t sv[VB ? TILE : 1];

	const uint32_t tid = threadIdx.x, w = wave_id(), lane = lane_id();
	if (tid < RS_RADIX) goff[tid] = digit_base[tid] + hist[tid * gridDim.x + blockIdx.x];

	const uint32_t n_tiles = (n + TILE - 1) / TILE;
	const uint32_t first_tile = blockIdx.x * tiles_per_block;
	const uint32_t lane_off = w * (64 * ITEMS) + lane;     // position of item 0 of this lane inside a tile
	unsigned long long key[ITEMS], nkey[ITEMS];
	val_t val[VB ? ITEMS : 1], nval[VB ? ITEMS : 1];
	uint32_t lrank[ITEMS];
	auto load_tile = [&](uint32_t tile, unsigned long long (&k)[ITEMS], val_t (&v)[VB ? ITEMS : 1]) {
		const uint32_t base = tile * TILE + lane_off;      // n < 2^32: 32-bit indices throughout
		if (tile + 1 < n_tiles || n % TILE == 0) {
#pragma unroll
			for (int i = 0; i < ITEMS; ++i) { k[i] = keys[base + i * 64]; if (VB) v[i] = vals[base + i * 64]; }
		} else {
#pragma unroll
			for (int i = 0; i < ITEMS; ++i) {
				const uint32_t idx = base + i * 64;
				const bool valid = idx < n;
				k[i] = valid ? keys[idx] : ~0ull;
				if (VB) v[i] = valid ? vals[idx] : val_t(0);
			}
		}
	};
	if (PREFETCH && first_tile < n_tiles) load_tile(first_tile, nkey, nval);

	for (uint32_t tt = 0; tt < tiles_per_block; ++tt) {
		const uint32_t tile = first_tile + tt;
		if (tile >= n_tiles) break;
		const bool full = tile + 1 < n_tiles || n % TILE == 0;
		const uint32_t in_tile = full ? TILE : n - tile * TILE;
		if (PREFETCH) {
#pragma unroll
			for (int i = 0; i < ITEMS; ++i) { key[i] = nkey[i]; if (VB) val[i] = nval[i]; }
			if (tt + 1 < tiles_per_block && tile + 1 < n_tiles) load_tile(tile + 1, nkey, nval);
		} else {
			load_tile(tile, key, val);
		}
		for (uint32_t j = tid; j < WAVES * RS_RADIX; j += THREADS) (&wcnt[0][0])[j] = 0;
		lds_barrier();

		// wave-level multisplit: rank of each key among the keys of its wave with the same digit
#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			const bool valid = full || (lane_off + i * 64) < in_tile;
			const uint32_t d = uint32_t(key[i] >> shift) & DMASK;
			// lanes of the wave with the same digit: a lane differs from me in bit b iff (ballot of bit b) ^ (my bit b,
			// broadcast to 0 / ~0) has its bit set; the eight "differs" masks are OR-ed and inverted.  Written on the two
			// 32-bit halves so that it compiles to bfe + cmp + 2 xor + or3 per bit (the 64-bit select form took 9).
			uint32_t diff_lo = 0, diff_hi = 0;
#pragma unroll
			for (int b = 0; b < RB; ++b) {
				const int32_t mine = int32_t(d << (31 - b)) >> 31;          // 0 or ~0
				const unsigned long long bal = __ballot(mine != 0);
				diff_lo |= uint32_t(bal) ^ uint32_t(mine);
				diff_hi |= uint32_t(bal >> 32) ^ uint32_t(mine);
			}
			unsigned long long m = ~(((unsigned long long)diff_hi << 32) | diff_lo);
			if (!full) m &= __ballot(valid);
			const uint32_t before = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));   // same-digit lanes below me
			const uint32_t old = wcnt[w][d];
			__builtin_amdgcn_wave_barrier();
			if (valid && before == 0) wcnt[w][d] = old + __popcll(m);
			__builtin_amdgcn_wave_barrier();
			lrank[i] = old + before;
		}
		lds_barrier();

		// combine waves: per digit exclusive offsets over waves, then exclusive scan over digits
		uint32_t run = 0;
		if (tid < RS_RADIX) {
#pragma unroll
			for (uint32_t k = 0; k < WAVES; ++k) { uint32_t c = wcnt[k][tid]; wcnt[k][tid] = run; run += c; }
			tcnt[tid] = run;
		}
		uint32_t total;
		uint32_t ex = block_excl_scan_u32<THREADS, true>(tid < RS_RADIX ? run : 0u, scratch, total);
		if (tid < RS_RADIX) { tstart[tid] = ex; gdelta[tid] = goff[tid] - ex; }
		lds_barrier();

		// re-order the tile by digit in LDS
#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			const bool valid = full || (lane_off + i * 64) < in_tile;
			if (valid) {
				const uint32_t d = uint32_t(key[i] >> shift) & DMASK;
				const uint32_t p = tstart[d] + wcnt[w][d] + lrank[i];
				sk[p] = key[i];
				if (VB) sv[p] = val[i];
			}
		}
		lds_barrier();

		// coalesced write-out: consecutive threads write consecutive addresses inside a digit run
		const uint32_t count = full ? TILE : total;
		for (uint32_t p = tid; p < count; p += THREADS) {
			const unsigned long long k = sk[p];
			const uint32_t g = gdelta[uint32_t(k >> shift) & DMASK] + p;
			okeys[g] = k;
			if (VB) ovals[g] = sv[p];
		}
		lds_barrier();
		if (tid < RS_RADIX) goff[tid] += tcnt[tid];
		// (the next iteration's barriers order this update before gdelta is recomputed)
	}
}

}  // namespace dropest
