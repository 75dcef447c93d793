// k_ssort.h -- splitter sort: two MSD partitions on sampled splitters + an LDS-resident finishing sort fused with the
// reads -> molecules reduce.  Hand-written for gfx950 (64-lane waves, LDS tiles); integer / HBM-bound work, no MFMA.
//
// Replaces, for the main sort of the pipeline, the 7-8 LSD passes of k_radix.h (each a histogram read + a scatter
// read/write of every record) and the seg_count / seg_reduce<ReadsToMolecules*> launches behind them by
//   ss_sample      F1*F2*OS keys (OS = 64) at a fixed stride -> sorted with the LSD sort (a few MB) -> F*F-1 fine splitters, every F-th
//                  of them a coarse splitter
//   L1  ss_hist / scan / ss_scatter   all records into F coarse buckets (bucket = number of coarse splitters <= key,
//                  branch-free binary search over the splitters in LDS)
//   L2  ss_hist / ss_scan_seg / ss_scatter   every coarse bucket into its F fine buckets (~1.5 k records each)
//   ss_local       one workgroup per fine bucket: the bucket is loaded ONCE into LDS, sorted there (LSD over the bits
//                  that vary inside the bucket: 3-4 passes of the wave-ballot multisplit), runs of equal molecule keys
//                  are folded (read count, mark, exon / intron reads) and written as molecule rows at the bucket's own
//                  offset -- the sorted reads never go back to HBM
//   ss_compact     scan of the per-bucket row counts, rows moved to the dense molecule table
// Records cross HBM 2.5 times (L1 r+w, L2 hist r, L2 r+w, local r) instead of 7 x 3: see DESIGN.md §2.
//
// Equal molecule keys always compare equal against every splitter, so a molecule never straddles two buckets; the
// partitions need not be stable (what is folded per molecule is commutative), so tile ranks come from LDS atomics.
// The reference code this stands in for: the std::map insert-position descents of Cell::genes() / Gene::_umis
// (Estimation/Cell.h:19, Gene.h:19) and Gene::add_umi / UMI::add_read (Gene.cpp:17-24, UMI.cpp:21-34).
#pragma once

#include "util.h"

namespace dropest {

constexpr int SS_T = 512, SS_I = 8, SS_TILE = SS_T * SS_I;   // partition tile: 4096 records
constexpr int SS_MAX_F = 512;                                // fan-out of one partition level (power of two, 16..512): kernels templated
                                                             // on MAXF = 512; MAXF = 1024 serves streams beyond ~4e8 reads (two LDS slots per thread)
constexpr uint32_t SS_LOCAL_MAX = 8192;                      // largest fine bucket the LDS sort takes (512 threads x 16)

// ---- sample + splitters -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ss_sample_kernel(const unsigned long long *__restrict__ keys, uint32_t n, int ms,
                                                        uint32_t n_sample, unsigned long long *__restrict__ out) {
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n_sample) return;
	const uint32_t pos = uint32_t((uint64_t(j) * n + n / 2) / n_sample);
	out[j] = keys[pos < n ? pos : n - 1] >> ms;
}

// fine[j] = sample[(j + 1) * os], j = 0 .. F2-2 (entry F2-1 = ~0, never compared); coarse[j] = fine[(j + 1) * F - 1], F = fine
// buckets per coarse bucket
__global__ __launch_bounds__(256) void ss_pick_splitters_kernel(const unsigned long long *__restrict__ sorted_sample, uint32_t os, uint32_t F, uint32_t F2,
                                                                unsigned long long *__restrict__ fine, unsigned long long *__restrict__ coarse) {
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j >= F2) return;
	const unsigned long long v = j + 1 < F2 ? sorted_sample[size_t(j + 1) * os] : ~0ull;
	fine[j] = v;
	if ((j + 1) % F == 0) coarse[(j + 1) / F - 1] = v;
}

// number of splitters <= mk among sp[0 .. F-2] (F = 1 << fb), branch-free upper bound
__device__ inline uint32_t ss_search(const unsigned long long *sp, int fb, unsigned long long mk) {
	uint32_t pos = 0;
	for (int b = fb - 1; b >= 0; --b) {
		const uint32_t step = 1u << b;
		if (sp[pos + step - 1] <= mk) pos += step;
	}
	return pos;
}

// ---- partition: histogram ------------------------------------------------------------------------------------------
// Counts of the F buckets over records [begin, end); the caller has the splitters in LDS.
__device__ inline void ss_hist_range(const unsigned long long *__restrict__ keys, uint32_t begin, uint32_t end, int ms, int fb,
                                     const unsigned long long *sp, uint32_t *cnt) {
	constexpr int U = 4;
	for (uint32_t i0 = begin + threadIdx.x; i0 < end; i0 += SS_T * U) {
		unsigned long long k[U];
		uint32_t pos[U];
#pragma unroll
		for (int q = 0; q < U; ++q) { const uint32_t i = i0 + q * SS_T; k[q] = i < end ? keys[i] >> ms : 0ull; pos[q] = 0; }
		for (int b = fb - 1; b >= 0; --b) {
			const uint32_t step = 1u << b;
#pragma unroll
			for (int q = 0; q < U; ++q) if (sp[pos[q] + step - 1] <= k[q]) pos[q] += step;
		}
#pragma unroll
		for (int q = 0; q < U; ++q) if (i0 + q * SS_T < end) atomicAdd(&cnt[pos[q]], 1u);
	}
}

// L1: block blk owns tiles [blk * tpb, (blk + 1) * tpb) of the whole array; hist[d * gridDim.x + blk]
template <int MAXF>
__global__ __launch_bounds__(SS_T) void ss_hist_l1_kernel(const unsigned long long *__restrict__ keys, uint32_t n, int ms, int fb,
                                                          const unsigned long long *__restrict__ coarse, uint32_t tiles_per_block,
                                                          uint32_t *__restrict__ hist) {
	__shared__ unsigned long long sp[MAXF];
	__shared__ uint32_t cnt[MAXF];
	const uint32_t F = 1u << fb;
	for (uint32_t j = threadIdx.x; j < F; j += SS_T) { sp[j] = j + 1 < F ? coarse[j] : ~0ull; cnt[j] = 0; }
	__syncthreads();
	const uint64_t b64 = uint64_t(blockIdx.x) * tiles_per_block * SS_TILE;
	uint64_t e64 = b64 + uint64_t(tiles_per_block) * SS_TILE;
	if (e64 > n) e64 = n;
	if (b64 < n) ss_hist_range(keys, uint32_t(b64), uint32_t(e64), ms, fb, sp, cnt);
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < F; j += SS_T) hist[j * gridDim.x + blockIdx.x] = cnt[j];
}

// row totals -> exclusive bases of the F coarse buckets; base[F] = n
template <int MAXF>
__global__ __launch_bounds__(MAXF) void ss_scan_totals_kernel(const uint32_t *__restrict__ row_total, uint32_t F, uint32_t n,
                                                               uint32_t *__restrict__ base) {
	__shared__ uint32_t scratch[MAXF / 64 + 1];
	uint32_t total;
	const uint32_t ex = block_excl_scan_u32<MAXF>(threadIdx.x < F ? row_total[threadIdx.x] : 0u, scratch, total);
	if (threadIdx.x < F) base[threadIdx.x] = ex;
	if (threadIdx.x == 0) base[F] = n;
}

// part p of coarse bucket s: whole tiles of the bucket, [begin, end) in record indices
__device__ inline void ss_l2_range(const uint32_t *__restrict__ base1, uint32_t s, uint32_t p, uint32_t parts, uint32_t &begin, uint32_t &end) {
	const uint32_t sb = base1[s], se = base1[s + 1];
	const uint32_t tiles = (se - sb + SS_TILE - 1) / SS_TILE, tpp = (tiles + parts - 1) / parts;
	const uint64_t b = uint64_t(sb) + uint64_t(p) * tpp * SS_TILE, e = b + uint64_t(tpp) * SS_TILE;
	begin = b < se ? uint32_t(b) : se;
	end = e < se ? uint32_t(e) : se;
}

// L2: block (s, p) = blockIdx.x / parts, % parts; cnt2[(s * F + d) * parts + p]
template <int MAXF>
__global__ __launch_bounds__(SS_T) void ss_hist_l2_kernel(const unsigned long long *__restrict__ keys, int ms, int fb,
                                                          const unsigned long long *__restrict__ fine, const uint32_t *__restrict__ base1,
                                                          uint32_t parts, uint32_t *__restrict__ cnt2) {
	__shared__ unsigned long long sp[MAXF];
	__shared__ uint32_t cnt[MAXF];
	const uint32_t F = 1u << fb, s = blockIdx.x / parts, p = blockIdx.x % parts;
	for (uint32_t j = threadIdx.x; j < F; j += SS_T) { sp[j] = j + 1 < F ? fine[size_t(s) * F + j] : ~0ull; cnt[j] = 0; }
	__syncthreads();
	uint32_t begin, end;
	ss_l2_range(base1, s, p, parts, begin, end);
	if (begin < end) ss_hist_range(keys, begin, end, ms, fb, sp, cnt);
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < F; j += SS_T) cnt2[(size_t(s) * F + j) * parts + p] = cnt[j];
}

// One block per coarse bucket s: turns cnt2[(s, f, p)] into absolute output cursors, writes base / size of every fine
// bucket and the largest size.
template <int MAXF>
__global__ __launch_bounds__(MAXF) void ss_scan_seg_kernel(uint32_t *__restrict__ cnt2, uint32_t F, uint32_t parts,
                                                            const uint32_t *__restrict__ base1, uint32_t *__restrict__ bucket_base,
                                                            uint32_t *__restrict__ bucket_cnt, uint32_t *__restrict__ max_cnt) {
	__shared__ uint32_t scratch[MAXF / 64 + 1];
	const uint32_t s = blockIdx.x, f = threadIdx.x;
	uint32_t *row = cnt2 + (size_t(s) * F + f) * parts;
	uint32_t sum = 0;
	if (f < F) for (uint32_t p = 0; p < parts; ++p) sum += row[p];
	uint32_t total;
	const uint32_t ex = block_excl_scan_u32<MAXF>(sum, scratch, total);
	if (f < F) {
		uint32_t run = base1[s] + ex;
		bucket_base[size_t(s) * F + f] = run;
		bucket_cnt[size_t(s) * F + f] = sum;
		for (uint32_t p = 0; p < parts; ++p) { const uint32_t c = row[p]; row[p] = run; run += c; }
	}
	const unsigned long long m = wave_reduce_max_u64(sum);
	if (lane_id() == 0 && m) atomicMax(max_cnt, uint32_t(m));
}

// ---- partition: scatter --------------------------------------------------------------------------------------------
// Records [begin, end) go to the F buckets whose running output cursors the caller has put into goff[] (LDS).
template <int VB, int MAXF>
__device__ inline void ss_scatter_range(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ vals,
                                        unsigned long long *__restrict__ okeys, uint8_t *__restrict__ ovals, uint32_t begin, uint32_t end,
                                        int ms, int fb, const unsigned long long *sp, uint32_t *goff, uint32_t *cnt, uint32_t *tstart,
                                        uint32_t *gdelta, uint32_t *scratch, unsigned long long *sk, uint16_t *sd, uint8_t *sv) {
	constexpr int PER = MAXF / SS_T;   // table entries per thread: 1, or 2 with 1024 buckets
	const uint32_t F = 1u << fb, tid = threadIdx.x;
	for (uint32_t t0 = begin; t0 < end; t0 += SS_TILE) {
		const uint32_t in_tile = end - t0 < uint32_t(SS_TILE) ? end - t0 : uint32_t(SS_TILE);
		for (uint32_t j = tid; j < F; j += SS_T) cnt[j] = 0;
		lds_barrier();
		unsigned long long key[SS_I];
		uint8_t val[VB ? SS_I : 1];
		uint32_t pos[SS_I], rk[SS_I];
#pragma unroll
		for (int i = 0; i < SS_I; ++i) {
			const uint32_t p = i * SS_T + tid;
			key[i] = p < in_tile ? keys[t0 + p] : 0ull;
			if (VB) val[i] = p < in_tile ? vals[t0 + p] : uint8_t(0);
			pos[i] = 0;
		}
		for (int b = fb - 1; b >= 0; --b) {
			const uint32_t step = 1u << b;
#pragma unroll
			for (int i = 0; i < SS_I; ++i) if (sp[pos[i] + step - 1] <= (key[i] >> ms)) pos[i] += step;
		}
#pragma unroll
		for (int i = 0; i < SS_I; ++i) rk[i] = (i * SS_T + tid) < in_tile ? atomicAdd(&cnt[pos[i]], 1u) : 0u;
		lds_barrier();
		uint32_t c[PER], mine = 0;
#pragma unroll
		for (int k = 0; k < PER; ++k) { const uint32_t e = tid * PER + k; c[k] = e < F ? cnt[e] : 0u; mine += c[k]; }
		uint32_t total;
		uint32_t ex = block_excl_scan_u32<SS_T, true>(mine, scratch, total);
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const uint32_t e = tid * PER + k;
			if (e < F) { tstart[e] = ex; gdelta[e] = goff[e] - ex; goff[e] += c[k]; }
			ex += c[k];
		}
		lds_barrier();
#pragma unroll
		for (int i = 0; i < SS_I; ++i)
			if ((i * SS_T + tid) < in_tile) {
				const uint32_t q = tstart[pos[i]] + rk[i];
				sk[q] = key[i]; sd[q] = uint16_t(pos[i]);
				if (VB) sv[q] = val[i];
			}
		lds_barrier();
		for (uint32_t q = tid; q < in_tile; q += SS_T) {
			const uint32_t g = gdelta[sd[q]] + q;
			okeys[g] = sk[q];
			if (VB) ovals[g] = sv[q];
		}
		lds_barrier();
	}
}

#define SS_SCATTER_LDS(VB, MAXF)                                                                            \
	__shared__ unsigned long long sp[MAXF];                                                                 \
	__shared__ uint32_t goff[MAXF], cnt[MAXF], tstart[MAXF], gdelta[MAXF], scratch[SS_T / 64 + 1];          \
	__shared__ unsigned long long sk[SS_TILE];                                                              \
	__shared__ uint16_t sd[SS_TILE];                                                                        \
	__shared__ uint8_t sv[VB ? SS_TILE : 1];

template <int VB, int MAXF>
__global__ __launch_bounds__(SS_T) void ss_scatter_l1_kernel(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ vals,
                                                             unsigned long long *__restrict__ okeys, uint8_t *__restrict__ ovals, uint32_t n,
                                                             int ms, int fb, const unsigned long long *__restrict__ coarse,
                                                             uint32_t tiles_per_block, const uint32_t *__restrict__ hist,
                                                             const uint32_t *__restrict__ base1) {
	SS_SCATTER_LDS(VB, MAXF)
	const uint32_t F = 1u << fb;
	for (uint32_t j = threadIdx.x; j < F; j += SS_T) {
		sp[j] = j + 1 < F ? coarse[j] : ~0ull;
		goff[j] = base1[j] + hist[j * gridDim.x + blockIdx.x];
	}
	__syncthreads();
	const uint64_t b64 = uint64_t(blockIdx.x) * tiles_per_block * SS_TILE;
	uint64_t e64 = b64 + uint64_t(tiles_per_block) * SS_TILE;
	if (e64 > n) e64 = n;
	if (b64 < n) ss_scatter_range<VB, MAXF>(keys, vals, okeys, ovals, uint32_t(b64), uint32_t(e64), ms, fb, sp, goff, cnt, tstart, gdelta, scratch, sk, sd, sv);
}

template <int VB, int MAXF>
__global__ __launch_bounds__(SS_T) void ss_scatter_l2_kernel(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ vals,
                                                             unsigned long long *__restrict__ okeys, uint8_t *__restrict__ ovals, int ms, int fb,
                                                             const unsigned long long *__restrict__ fine, const uint32_t *__restrict__ base1,
                                                             uint32_t parts, const uint32_t *__restrict__ cnt2) {
	SS_SCATTER_LDS(VB, MAXF)
	const uint32_t F = 1u << fb, s = blockIdx.x / parts, p = blockIdx.x % parts;
	for (uint32_t j = threadIdx.x; j < F; j += SS_T) {
		sp[j] = j + 1 < F ? fine[size_t(s) * F + j] : ~0ull;
		goff[j] = cnt2[(size_t(s) * F + j) * parts + p];
	}
	__syncthreads();
	uint32_t begin, end;
	ss_l2_range(base1, s, p, parts, begin, end);
	if (begin < end) ss_scatter_range<VB, MAXF>(keys, vals, okeys, ovals, begin, end, ms, fb, sp, goff, cnt, tstart, gdelta, scratch, sk, sd, sv);
}

// ---- partition by RESERVATION: no histogram pass ---------------------------------------------------------------------
// The splitters are quantiles of a sample, so every bucket of a level expects the same number of records: n / F1 per coarse bucket
// (16 384 sample points each: +- 0.8 %), n / F2 per fine bucket (64 sample points: +- 12 %).  Instead of counting first (one more read of
// all keys per level: ss_hist_l1 / ss_hist_l2, 0.28 ms each per 1e8 reads) every bucket gets a REGION of fixed capacity -- the mean plus
// 1/16 at the first level, 1.75 x the mean at the second -- and a tile takes its places in the regions it feeds with one atomic add per
// bucket on the bucket's cursor (issued before the tile is regrouped in LDS, needed only when it is written out).  A bucket that
// outgrows its region raises a flag (records beyond the region are dropped, nothing is overwritten): the host then rebuilds the keys
// and takes the counting path -- a stream with a heavy key (one molecule with a good share of all reads) does that.  What follows the
// partition (ss_local, ss_compact) addresses buckets by (base, count) anyway; the regions' gaps cost address space, not traffic.
struct SsReserve {
	uint32_t *cursor;       // per bucket: records placed so far
	uint32_t cstride;       // words between the cursors of neighbouring buckets (32 at the first level: one cache line each)
	uint32_t cap;           // records a region holds
	uint32_t first;         // index of this block's first bucket among all regions of the level
	uint32_t *overflow;
	uint32_t dbg;           // timing probes (DROPEST_SS_PROBE; results unusable): 1 no splitter search, 2 no LDS regrouping
	uint32_t rebase_bits;   // REBASE: bits a rebased key may take (61; tests: fewer, so that the fall-back runs)
};
// REBASE (first level of the key + mark byte layout, a key that fills all 64 bits: C3): a record leaves as ONE word,
//   ((key - lo(bucket)) << 3) | mark,   lo(d) = the bucket's lower splitter with the UMI field cleared (0 for bucket 0),
// so the second level and ss_local run on keys only (9 -> 8 bytes per record, no scattered byte stores): inside a coarse bucket the order
// of the differences is the order of the keys, and differences above the UMI field are equal exactly when the fields are (lo has no UMI
// bits).  ss_local adds lo back when it writes a molecule's key.  A difference that does not fit 61 bits (a coarse bucket spanning an
// eighth of the key space: never with quantile splitters over used cell ids) raises the overflow flag: the pass is then redone with the
// counting partitions, which keep the byte column.
template <int VB, int MAXF, bool REBASE = false>
__device__ inline void ss_scatter_res_range(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ vals,
                                            unsigned long long *__restrict__ okeys, uint8_t *__restrict__ ovals, uint32_t begin, uint32_t end,
                                            int ms, int fb, const unsigned long long *sp, const SsReserve rs, uint32_t *cnt, uint32_t *tstart,
                                            uint32_t *gdelta, uint32_t *scratch, unsigned long long *sk, uint16_t *sd, uint8_t *sv,
                                            unsigned long long rebase_mask = 0) {
	static_assert(!REBASE || VB == 1, "rebased keys carry the mark byte in their low bits");
	constexpr int PER = (MAXF + SS_T - 1) / SS_T;   // table entries per thread (MAXF = 256: the upper half of the threads has none)
	const uint32_t F = 1u << fb, tid = threadIdx.x;
	// The records of the NEXT tile are requested as soon as this tile's have moved from registers to LDS: they are on their way while this
	// tile is written out and the next one's buckets are counted.  (Every kernel of the pass spends most of its wave cycles parked on
	// memory -- rocprofv3 SQ_WAIT_ANY 59-66 % here --: what helps is more requests in flight per CU, not fewer instructions.)
	unsigned long long key[SS_I];
	uint8_t val[VB ? SS_I : 1];
	auto load_tile = [&](uint32_t t0) {
		const uint32_t in_tile = end - t0 < uint32_t(SS_TILE) ? end - t0 : uint32_t(SS_TILE);
#pragma unroll
		for (int i = 0; i < SS_I; ++i) {
			const uint32_t p = i * SS_T + tid;
			key[i] = p < in_tile ? keys[t0 + p] : 0ull;
			if (VB) val[i] = p < in_tile ? vals[t0 + p] : uint8_t(0);
		}
	};
	if (begin < end) load_tile(begin);
	for (uint32_t t0 = begin; t0 < end; t0 += SS_TILE) {
		const uint32_t in_tile = end - t0 < uint32_t(SS_TILE) ? end - t0 : uint32_t(SS_TILE);
		for (uint32_t j = tid; j < F; j += SS_T) cnt[j] = 0;
		lds_barrier();
		uint32_t pos[SS_I], rk[SS_I];
#pragma unroll
		for (int i = 0; i < SS_I; ++i) pos[i] = 0;
		for (int b = fb - 1; b >= 0; --b) {
			const uint32_t step = 1u << b;
#pragma unroll
			for (int i = 0; i < SS_I; ++i) if (sp[pos[i] + step - 1] <= (key[i] >> ms)) pos[i] += step;
		}
		if (rs.dbg & 1u) {   // probe: the search a second time (what it costs = what the launch gains in time)
			uint32_t pos2[SS_I];
#pragma unroll
			for (int i = 0; i < SS_I; ++i) pos2[i] = 0;
			for (int b = fb - 1; b >= 0; --b) {
				const uint32_t step = 1u << b;
#pragma unroll
				for (int i = 0; i < SS_I; ++i) if (sp[pos2[i] + step - 1] <= ((key[i] ^ (rs.dbg >> 8)) >> ms)) pos2[i] += step;
			}
#pragma unroll
			for (int i = 0; i < SS_I; ++i) pos[i] = (pos[i] + pos2[i]) >> 1;
		}
#pragma unroll
		for (int i = 0; i < SS_I; ++i) rk[i] = (i * SS_T + tid) < in_tile ? atomicAdd(&cnt[pos[i]], 1u) : 0u;
		lds_barrier();
		uint32_t c[PER], got[PER], mine = 0;
#pragma unroll
		for (int k = 0; k < PER; ++k) {   // the tile's places in the regions: in flight while the tile is regrouped
			const uint32_t e = tid * PER + k;
			c[k] = e < F ? cnt[e] : 0u; mine += c[k];
			got[k] = c[k] ? atomicAdd(&rs.cursor[size_t(rs.first + e) * rs.cstride], c[k]) : 0u;
		}
		uint32_t total;
		uint32_t ex = block_excl_scan_u32<SS_T, true>(mine, scratch, total);
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const uint32_t e = tid * PER + k;
			if (e < F) tstart[e] = ex;
			ex += c[k];
		}
		lds_barrier();
#pragma unroll
		for (int i = 0; i < SS_I; ++i)
			if ((i * SS_T + tid) < in_tile) {
				const uint32_t q = tstart[pos[i]] + rk[i];
				sk[q] = key[i]; sd[q] = uint16_t(pos[i]);
				if (VB) sv[q] = val[i];
			}
		if (!(rs.dbg & 2u) && t0 + SS_TILE < end) load_tile(t0 + SS_TILE);   // (probe 2: no prefetch)
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const uint32_t e = tid * PER + k;
			if (e < F) {
				gdelta[e] = (rs.first + e) * rs.cap + got[k] - tstart[e];
				if (got[k] + c[k] > rs.cap) atomicOr(rs.overflow, 1u);
			}
		}
		lds_barrier();
		for (uint32_t q = tid; q < in_tile; q += SS_T) {
			const uint32_t d = sd[q], g = gdelta[d] + q;
			if (g < (rs.first + d + 1u) * rs.cap) {
				if constexpr (REBASE) {
					const unsigned long long diff = sk[q] - (d ? (sp[d - 1u] & rebase_mask) : 0ull);
					if (diff >> rs.rebase_bits) atomicOr(rs.overflow, 1u);
					okeys[g] = (diff << 3) | (unsigned long long)(sv[q] & 7u);
				} else {
					okeys[g] = sk[q];
					if (VB) ovals[g] = sv[q];
				}
			}
		}
		if ((rs.dbg & 2u) && t0 + SS_TILE < end) load_tile(t0 + SS_TILE);
		lds_barrier();
	}
}

#define SS_SCATTER_RES_LDS(VB, MAXF)                                                                        \
	__shared__ unsigned long long sp[MAXF];                                                                 \
	__shared__ uint32_t cnt[MAXF], tstart[MAXF], gdelta[MAXF], scratch[SS_T / 64 + 1];                      \
	__shared__ unsigned long long sk[SS_TILE];                                                              \
	__shared__ uint16_t sd[SS_TILE];                                                                        \
	__shared__ uint8_t sv[VB ? SS_TILE : 1];

// first level: block blk owns tiles [blk * tpb, (blk + 1) * tpb); bucket j's region = [j * cap, (j + 1) * cap) of okeys
template <int VB, int MAXF, bool REBASE = false>
__global__ __launch_bounds__(SS_T) void ss_scatter_res_l1_kernel(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ vals,
                                                                 unsigned long long *__restrict__ okeys, uint8_t *__restrict__ ovals, uint32_t n,
                                                                 int ms, int fb, const unsigned long long *__restrict__ coarse,
                                                                 uint32_t tiles_per_block, SsReserve rs, unsigned long long rebase_mask) {
	SS_SCATTER_RES_LDS(VB, MAXF)
	const uint32_t F = 1u << fb;
	for (uint32_t j = threadIdx.x; j < F; j += SS_T) sp[j] = j + 1 < F ? coarse[j] : ~0ull;
	__syncthreads();
	const uint64_t b64 = uint64_t(blockIdx.x) * tiles_per_block * SS_TILE;
	uint64_t e64 = b64 + uint64_t(tiles_per_block) * SS_TILE;
	if (e64 > n) e64 = n;
	if (b64 < n) ss_scatter_res_range<VB, MAXF, REBASE>(keys, vals, okeys, ovals, uint32_t(b64), uint32_t(e64), ms, fb, sp, rs, cnt, tstart, gdelta, scratch, sk, sd, sv, rebase_mask);
}
// second level: block (s, p) takes part p of coarse region s (cur1 = the first level's cursors = the regions' fills)
template <int VB, int MAXF>
__global__ __launch_bounds__(SS_T) void ss_scatter_res_l2_kernel(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ vals,
                                                                 unsigned long long *__restrict__ okeys, uint8_t *__restrict__ ovals, int ms, int fb,
                                                                 const unsigned long long *__restrict__ fine, const uint32_t *__restrict__ cur1,
                                                                 uint32_t cstride1, uint32_t cap1, uint32_t parts, SsReserve rs,
                                                                 const unsigned long long *__restrict__ rebase_coarse, unsigned long long rebase_mask) {
	SS_SCATTER_RES_LDS(VB, MAXF)
	const uint32_t F = 1u << fb, s = blockIdx.x / parts, p = blockIdx.x % parts;
	// (rebased keys, see ss_scatter_res_range: the region's fine splitters move down by its lo as well)
	const unsigned long long lo = (rebase_coarse && s) ? (rebase_coarse[s - 1] & rebase_mask) : 0ull;
	for (uint32_t j = threadIdx.x; j < F; j += SS_T) {
		const unsigned long long f = j + 1 < F ? fine[size_t(s) * F + j] : ~0ull;
		sp[j] = j + 1 < F ? (f > lo ? f - lo : 0ull) : ~0ull;
	}
	__syncthreads();
	const uint32_t fill = cur1[size_t(s) * cstride1], sb = s * cap1, se = sb + (fill < cap1 ? fill : cap1);
	const uint32_t tiles = (se - sb + SS_TILE - 1) / SS_TILE, tpp = (tiles + parts - 1) / parts;
	const uint64_t b = uint64_t(sb) + uint64_t(p) * tpp * SS_TILE, e = b + uint64_t(tpp) * SS_TILE;
	const uint32_t begin = b < se ? uint32_t(b) : se, end = e < se ? uint32_t(e) : se;
	SsReserve mine = rs;
	mine.first = s * F;
	if (begin < end) ss_scatter_res_range<VB, MAXF>(keys, vals, okeys, ovals, begin, end, ms, fb, sp, mine, cnt, tstart, gdelta, scratch, sk, sd, sv);
}
// the fine regions as ss_local's buckets: base = b * cap, count = the region's fill; the largest count
__global__ __launch_bounds__(256) void ss_res_buckets_kernel(const uint32_t *__restrict__ cur2, uint32_t n_buckets, uint32_t cap,
                                                             uint32_t *__restrict__ bucket_base, uint32_t *__restrict__ bucket_cnt, uint32_t *__restrict__ max_cnt) {
	const uint32_t b = blockIdx.x * 256 + threadIdx.x;
	uint32_t c = 0;
	if (b < n_buckets) { c = cur2[b] < cap ? cur2[b] : cap; bucket_base[b] = b * cap; bucket_cnt[b] = c; }
	const unsigned long long m = wave_reduce_max_u64(c);
	if (lane_id() == 0 && m) atomicMax(max_cnt, uint32_t(m));
}

// ---- finishing sort + reads -> molecules ---------------------------------------------------------------------------
// One workgroup per fine bucket (blockIdx = bucket).  A bucket of c records yields at most c molecules, so its rows are
// written at the bucket's OWN record offset into scratch arrays (sparse) together with the row count; ss_compact then
// scans the counts and moves the rows to the dense table.  (Writing dense rows straight away needs every bucket's
// molecule count before its rows can be placed: a decoupled look-back was built and measured -- a bucket only knows its
// count after its sort, so every workgroup waited for its slowest resident predecessor and the single ticket word
// alone capped the launch at 88 workgroups per microsecond; 3.1-3.5 ms against 1.2 + 0.45 ms for sparse rows + compaction.)
struct SsLocalArgs {
	const unsigned long long *keys;      // grouped by fine bucket (output of the L2 scatter)
	const uint8_t *vals;                 // mark bytes (VB = 1)
	const uint32_t *bucket_base, *bucket_cnt;
	const uint32_t *big_list;            // (big launch) bucket ids
	uint32_t n_buckets;
	int ms;                              // mark bits folded under the key (VB = 0: 3)
	uint32_t cap;                        // records the LDS of this launch holds
	uint32_t skip_above;                 // small launch: buckets with more records are left to the big launch
	uint32_t debug;                      // DROPEST_SS_DEBUG: 2 = skip the sort passes (the order check below must then catch it: tests)
	uint32_t atomic_below;               // ATOMIC_RANK: passes whose digit starts below this bit rank by the LDS atomic, the others by ballots
	uint32_t *order_flag;                // set to 1 when a bucket is found out of order after its sort (checked on EVERY pass)
	unsigned long long *t_key;           // sparse rows: molecule key, reads, agg (bit 0 not-annotated, exon << 1, intron << 16)
	uint32_t *t_reads, *t_agg, *n_loc;
	// (cell, gene) heads INSIDE the bucket: records p > 0 whose key above the UMI field differs from their predecessor's (cg_shift = ms + UMI bits).
	// ss_compact_cg turns them into the (cell, gene) table while it moves the molecule rows (zeroed before the launches: every wave adds its share)
	uint32_t *cg_loc;
	int cg_shift;
	// rebased keys (ss_scatter_res_range<..., REBASE>): fine bucket b belongs to coarse bucket b / rebase_div, whose lo comes back onto the molecule keys
	const unsigned long long *rebase_coarse;
	unsigned long long rebase_mask;
	uint32_t rebase_div;
};

// LDS of ss_local (dynamic, 16-byte aligned): sk[cap] keys -- later aliased by agg[cap] + headpos[cap] (u32 each) --,
// wcnt[WAVES][256], tstart[256], scratch, red[], sv[cap] bytes
// ATOMIC_RANK: the rank of a record among the records of its wave with the same digit comes from ONE LDS atomic add (returning the
// old count) instead of 8 ballots + ~50 ALU instructions.  That is stable only because the LDS applies the lanes of one instruction
// that hit the same address in lane order -- true on gfx950, checked once per device before it is relied on
// (dropest_amd.hip: lds_atomics_lane_ordered); the ballot ranking of k_radix.h is the fall-back.
template <int THREADS, int ITEMS, int VB, bool ATOMIC_RANK = false>
__device__ inline void ss_local_run(const SsLocalArgs &a, uint32_t bucket, uint32_t base, uint32_t cnt, unsigned char *smem) {
	constexpr uint32_t WAVES = THREADS / 64;
	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
	unsigned long long *sk = reinterpret_cast<unsigned long long *>(smem);
	uint32_t *agg = reinterpret_cast<uint32_t *>(smem);
	uint32_t *headpos = agg + a.cap;
	uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + size_t(a.cap) * 8);   // [WAVES][256]
	uint32_t *tstart = wcnt + WAVES * 256;                                     // [256]
	uint32_t *scratch = tstart + 256;                                          // [THREADS / 64 + 1], then wtot[WAVES], misc[4]
	uint32_t *wtot = scratch + (THREADS / 64 + 1);
	uint32_t *misc = wtot + WAVES;
	unsigned long long *red = reinterpret_cast<unsigned long long *>(misc + 4 + ((WAVES + 1) & 1u));   // [2 * WAVES], 8-byte aligned
	uint8_t *sv = reinterpret_cast<uint8_t *>(red + 2 * WAVES);
	const int ms = a.ms;
	const uint32_t coarse_b = a.rebase_coarse ? bucket / a.rebase_div : 0u;
	const unsigned long long kbase = coarse_b ? (a.rebase_coarse[coarse_b - 1u] & a.rebase_mask) : 0ull;

	const uint32_t lane_off = w * (64 * ITEMS) + lane;   // position of item 0; item i sits at lane_off + 64 i: order = (wave, item, lane)
	unsigned long long key[ITEMS];
	uint8_t val[VB ? ITEMS : 1];
	unsigned long long k_or = 0, k_and = ~0ull;
#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		const uint32_t p = lane_off + i * 64;
		const bool valid = p < cnt;
		key[i] = valid ? a.keys[base + p] : 0ull;
		if (VB) val[i] = valid ? a.vals[base + p] : uint8_t(0);
	}
#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		const uint32_t p = lane_off + i * 64;
		if (p < cnt) { k_or |= key[i]; k_and &= key[i]; sk[p] = key[i]; if (VB) sv[p] = val[i]; }
	}
	k_or = wave_reduce_or_u64(k_or); k_and = wave_reduce_and_u64(k_and);
	if (lane == 0) { red[2 * w] = k_or; red[2 * w + 1] = k_and; }
	lds_barrier();
	k_or = 0; k_and = ~0ull;
#pragma unroll
	for (uint32_t k = 0; k < WAVES; ++k) { k_or |= red[2 * k]; k_and &= red[2 * k + 1]; }
	const unsigned long long vary = (k_or ^ k_and) & ~((1ull << ms) - 1ull);

	// LSD over the varying bits, 8 per pass, ranks by the wave-ballot multisplit of k_radix.h (stable)
	for (int shift = vary ? __builtin_ctzll(vary) : 64; shift < 64 && (vary >> shift) != 0; shift += 8) {
		if (((vary >> shift) & 0xFFull) == 0 || (a.debug & 2u)) continue;
		for (uint32_t j = tid; j < WAVES * 256; j += THREADS) wcnt[j] = 0;
		lds_barrier();
		uint32_t lrank[ITEMS];
		// The one-atomic rank costs a cycle per lane that shares the digit: nothing on the random UMI bits, up to 64 on the digits above
		// them (a bucket is ~1500 consecutive records of the sorted order: one or two cells, a window of genes).  Those passes rank by
		// ballots -- over the VARYING bits of the digit only, often two or three.
		const bool by_atomic = ATOMIC_RANK && uint32_t(shift) < a.atomic_below;
		const uint32_t vbits = uint32_t(vary >> shift) & 0xFFu;
#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			const bool valid = (lane_off + i * 64) < cnt;
			const uint32_t d = uint32_t(key[i] >> shift) & 0xFFu;
			if (by_atomic) { lrank[i] = valid ? atomicAdd(&wcnt[w * 256 + d], 1u) : 0u; continue; }
			uint32_t diff_lo = 0, diff_hi = 0;
			for (uint32_t left = vbits; left; left &= left - 1u) {
				const int b = __builtin_ctz(left);
				const int32_t mine = int32_t(d << (31 - b)) >> 31;
				const unsigned long long bal = __ballot(mine != 0);
				diff_lo |= uint32_t(bal) ^ uint32_t(mine);
				diff_hi |= uint32_t(bal >> 32) ^ uint32_t(mine);
			}
			unsigned long long m = ~(((unsigned long long)diff_hi << 32) | diff_lo);
			m &= __ballot(valid);
			const uint32_t before = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
			const uint32_t old = wcnt[w * 256 + d];
			__builtin_amdgcn_wave_barrier();
			if (valid && before == 0) wcnt[w * 256 + d] = old + __popcll(m);
			__builtin_amdgcn_wave_barrier();
			lrank[i] = old + before;
		}
		lds_barrier();
		if constexpr (THREADS >= 256) {
			uint32_t run = 0;
			if (tid < 256) {
#pragma unroll
				for (uint32_t k = 0; k < WAVES; ++k) { const uint32_t c = wcnt[k * 256 + tid]; wcnt[k * 256 + tid] = run; run += c; }
			}
			uint32_t total;
			const uint32_t ex = block_excl_scan_u32<THREADS, true>(tid < 256 ? run : 0u, scratch, total);
			if (tid < 256) tstart[tid] = ex;
		} else {   // fewer threads than digits: 256 / THREADS consecutive digits per thread
			constexpr uint32_t PER = 256 / THREADS;
			uint32_t tot[PER], sum = 0;
#pragma unroll
			for (uint32_t j = 0; j < PER; ++j) {
				const uint32_t d = tid * PER + j;
				uint32_t run = 0;
#pragma unroll
				for (uint32_t k = 0; k < WAVES; ++k) { const uint32_t c = wcnt[k * 256 + d]; wcnt[k * 256 + d] = run; run += c; }
				tot[j] = run; sum += run;
			}
			uint32_t total;
			uint32_t ex = block_excl_scan_u32<THREADS, true>(sum, scratch, total);
#pragma unroll
			for (uint32_t j = 0; j < PER; ++j) { tstart[tid * PER + j] = ex; ex += tot[j]; }
		}
		lds_barrier();
#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if ((lane_off + i * 64) < cnt) {
				const uint32_t d = uint32_t(key[i] >> shift) & 0xFFu;
				const uint32_t q = tstart[d] + wcnt[w * 256 + d] + lrank[i];
				sk[q] = key[i];
				if (VB) sv[q] = val[i];
			}
		lds_barrier();
#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			const uint32_t p = lane_off + i * 64;
			if (p < cnt) { key[i] = sk[p]; if (VB) val[i] = sv[p]; }
		}
	}

	// heads: a record whose molecule key differs from its predecessor's; position order = (wave, item, lane).
	// The same comparison verifies the sort: a bucket is sorted iff no record is smaller than its predecessor -- one compare per
	// record, OR-reduced into a device flag the host reads together with the molecule count.  The one-atomic ranking above leans on
	// the LDS applying same-address lanes of one instruction in lane order (undocumented); should that ever fail, molecules would
	// silently split into several runs.  With this check it cannot be silent: the host falls back to the LSD sort.
	uint32_t run = 0, pre[ITEMS], head_bits = 0, cg_heads = 0;
	bool out_of_order = false;
#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		const uint32_t p = lane_off + i * 64;
		const bool valid = p < cnt;
		const unsigned long long prev = (valid && p) ? sk[p - 1] : 0ull;
		out_of_order |= valid && p && (prev >> ms) > (key[i] >> ms);
		const bool head = valid && (p == 0 || (prev >> ms) != (key[i] >> ms));
		const unsigned long long bal = __ballot(head);
		pre[i] = run + __builtin_amdgcn_mbcnt_hi(uint32_t(bal >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bal), 0u));
		run += uint32_t(__popcll(bal));
		if (head) head_bits |= 1u << i;
		if (a.cg_loc) cg_heads += uint32_t(__popcll(__ballot(valid && p && (prev >> a.cg_shift) != (key[i] >> a.cg_shift))));
	}
	if (lane == 0) wtot[w] = run;
	if (a.cg_loc && lane == 0 && cg_heads) atomicAdd(&a.cg_loc[bucket], cg_heads);
	if (__ballot(out_of_order) && lane == 0) atomicOr(a.order_flag, 1u);
	lds_barrier();   // every sk[p - 1] is read: the key area may now hold the aggregates
	uint32_t woff = 0, n_loc = 0;
#pragma unroll
	for (uint32_t k = 0; k < WAVES; ++k) { const uint32_t t = wtot[k]; if (k < w) woff += t; n_loc += t; }
	for (uint32_t m = tid; m < n_loc; m += THREADS) agg[m] = 0;
	if (tid == 0) a.n_loc[bucket] = n_loc;
	lds_barrier();
	// fold: agg = any not-annotated read (bit 0, OR) | exon reads << 1 (15 bits, +) | intron reads << 16 (15 bits, +)
#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		const uint32_t p = lane_off + i * 64;
		if (p >= cnt) continue;
		const uint32_t is_head = (head_bits >> i) & 1u;
		const uint32_t m = woff + pre[i] + is_head - 1u;
		const uint32_t mark = VB ? uint32_t(val[i]) & 7u : uint32_t(key[i]) & 7u;
		if (is_head) { headpos[m] = p; a.t_key[base + m] = (key[i] >> ms) + kbase; }
		const uint32_t add = (((mark >> 1) & 1u) << 1) | (((mark >> 2) & 1u) << 16);
		if (add) atomicAdd(&agg[m], add);
		if (mark & 1u) atomicOr(&agg[m], 1u);
	}
	lds_barrier();
	for (uint32_t m = tid; m < n_loc; m += THREADS) {
		const uint32_t e = m + 1 < n_loc ? headpos[m + 1] : cnt;
		a.t_reads[base + m] = e - headpos[m];
		a.t_agg[base + m] = agg[m];
	}
}

// small launch: grid = all buckets, 256 threads, buckets of up to 2048 records (larger ones are listed for the big launch)
template <int VB, bool ATOMIC_RANK = false>
__global__ __launch_bounds__(256) void ss_local_kernel(SsLocalArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];
	const uint32_t b = blockIdx.x;
	const uint32_t cnt = a.bucket_cnt[b], base = a.bucket_base[b];
	if (cnt == 0) { if (threadIdx.x == 0) a.n_loc[b] = 0; return; }
	if (cnt > a.skip_above) return;
	if (cnt <= 1024) ss_local_run<256, 4, VB, ATOMIC_RANK>(a, b, base, cnt, ss_smem);
	else ss_local_run<256, 8, VB, ATOMIC_RANK>(a, b, base, cnt, ss_smem);
}
// (experiment, DROPEST_SS_LOCAL_WAVE=64 / 128) the same finishing sort with one or two waves per bucket: no / cheaper workgroup barriers
template <int THREADS, int VB, bool ATOMIC_RANK = false>
__global__ __launch_bounds__(THREADS) void ss_local_wave_kernel(SsLocalArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];
	const uint32_t b = blockIdx.x;
	const uint32_t cnt = a.bucket_cnt[b], base = a.bucket_base[b];
	if (cnt == 0) { if (threadIdx.x == 0) a.n_loc[b] = 0; return; }
	if (cnt > a.skip_above) return;
	if (cnt <= 1024) ss_local_run<THREADS, 1024 / THREADS, VB, ATOMIC_RANK>(a, b, base, cnt, ss_smem);
	else ss_local_run<THREADS, 2048 / THREADS, VB, ATOMIC_RANK>(a, b, base, cnt, ss_smem);
}
// listed buckets: medium launch 256 threads x 16 records (up to 4096), big launch 512 x 16 (up to 8192)
template <int THREADS, int VB, bool ATOMIC_RANK = false>
__global__ __launch_bounds__(THREADS) void ss_local_big_kernel(SsLocalArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];
	const uint32_t b = a.big_list[blockIdx.x];
	ss_local_run<THREADS, 16, VB, ATOMIC_RANK>(a, b, a.bucket_base[b], a.bucket_cnt[b], ss_smem);
}

// bytes of dynamic LDS ss_local needs for `cap` records
inline size_t ss_local_lds_bytes(uint32_t cap, int threads) {
	const size_t waves = size_t(threads) / 64;
	size_t words = waves * 256 + 256 + (size_t(threads) / 64 + 1) + waves + 4 + ((waves + 1) & 1u);
	return size_t(cap) * 8 + words * 4 + 2 * waves * 8 + size_t(cap) + 16;
}

// probe of the property ATOMIC_RANK relies on: 64 lanes add 1 to counters picked by `digit`; out = what each lane got back
__global__ __launch_bounds__(64) void ss_lds_order_probe_kernel(const uint32_t *__restrict__ digit, uint32_t rounds, uint32_t *__restrict__ out) {
	__shared__ uint32_t cnt[64];
	for (uint32_t r = 0; r < rounds; ++r) {
		cnt[threadIdx.x] = 0;
		__builtin_amdgcn_wave_barrier();
		out[r * 64 + threadIdx.x] = atomicAdd(&cnt[digit[r * 64 + threadIdx.x] & 63u], 1u);
		__builtin_amdgcn_wave_barrier();
	}
}

// ---- compaction of the sparse rows ---------------------------------------------------------------------------------
// sums of n_loc over chunks of MAXF buckets
template <int MAXF>
__global__ __launch_bounds__(MAXF) void ss_chunk_sums_kernel(const uint32_t *__restrict__ n_loc, uint32_t n_buckets, uint32_t *__restrict__ chunk_sum) {
	__shared__ uint32_t scratch[MAXF / 64 + 1];
	const uint32_t i = blockIdx.x * MAXF + threadIdx.x;
	uint32_t total;
	block_excl_scan_u32<MAXF>(i < n_buckets ? n_loc[i] : 0u, scratch, total);
	if (threadIdx.x == 0) chunk_sum[blockIdx.x] = total;
}
// exclusive prefix of n_loc (chunk c adds the sums of the chunks before it: at most 512 of them); total -> *n_mol
template <int MAXF>
__global__ __launch_bounds__(MAXF) void ss_prefix_kernel(const uint32_t *__restrict__ n_loc, uint32_t n_buckets, const uint32_t *__restrict__ chunk_sum,
                                                          uint32_t n_chunks, uint32_t *__restrict__ prefix, uint32_t *__restrict__ n_mol) {
	__shared__ uint32_t scratch[MAXF / 64 + 1];
	__shared__ uint32_t chunk_base;
	uint32_t t1;
	const uint32_t before = block_excl_scan_u32<MAXF>(threadIdx.x < n_chunks ? chunk_sum[threadIdx.x] : 0u, scratch, t1);
	if (threadIdx.x == blockIdx.x) chunk_base = before;
	__syncthreads();
	const uint32_t i = blockIdx.x * MAXF + threadIdx.x;
	uint32_t t2;
	const uint32_t ex = block_excl_scan_u32<MAXF>(i < n_buckets ? n_loc[i] : 0u, scratch, t2);
	if (i < n_buckets) prefix[i] = chunk_base + ex;
	if (blockIdx.x == 0 && threadIdx.x == 0) *n_mol = t1;
}
struct SsCompactArgs {
	const uint32_t *bucket_base, *n_loc, *prefix;
	uint32_t n_buckets;
	const unsigned long long *t_key;
	const uint32_t *t_reads, *t_agg;
	unsigned long long *mol_key;
	uint32_t *mol_reads, *mol_mark, *mol_exon, *mol_intron;
};
// one wave per bucket: its rows move from the sparse scratch to their place in the molecule table
__global__ __launch_bounds__(256) void ss_compact_kernel(SsCompactArgs a) {
	const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (b >= a.n_buckets) return;
	const uint32_t n = a.n_loc[b], src = a.bucket_base[b], dst = a.prefix[b];
	for (uint32_t j = lane; j < n; j += 64) {
		const uint32_t v = a.t_agg[src + j], exon = (v >> 1) & 0x7FFFu, intron = (v >> 16) & 0x7FFFu;
		a.mol_key[dst + j] = a.t_key[src + j];
		a.mol_reads[dst + j] = a.t_reads[src + j];
		a.mol_mark[dst + j] = (v & 1u) | (exon ? 2u : 0u) | (intron ? 4u : 0u);
		a.mol_exon[dst + j] = exon;
		a.mol_intron[dst + j] = intron;
	}
}

// ---- compaction fused with molecules -> (cell, gene) ---------------------------------------------------------------------
// ss_compact reads every sparse molecule row once to move it; the (cell, gene) table was then made by seg_count + seg_reduce<MoleculesTo
// CellGeneX>, which read the dense table again (8 + 24 bytes per molecule).  The rows a wave moves are in key order, so it folds them into
// (cell, gene) rows on the way: Gene::number_of_umis / number_of_requested_umis / reads (Gene.cpp:60-93) per row, exactly what the
// policy MoleculesToCellGeneX (k_segreduce.h) computes.  Where a row goes needs the number of (cell, gene) heads in the buckets before:
// ss_local counted the heads INSIDE each bucket (cg_loc), ss_cg_counts adds whether a bucket's first molecule opens a new pair (its
// predecessor is the last molecule of the nearest non-empty bucket before it), the scan gives cg_prefix.
__global__ __launch_bounds__(256) void ss_cg_counts_kernel(const uint32_t *__restrict__ bucket_base, const uint32_t *__restrict__ n_loc, const unsigned long long *__restrict__ t_key,
                                                           int umi_bits, uint32_t n_buckets, const uint32_t *__restrict__ cg_loc, uint32_t *__restrict__ cg_cnt) {
	const uint32_t b = blockIdx.x * 256 + threadIdx.x;
	if (b >= n_buckets) return;
	uint32_t c = 0;
	if (n_loc[b]) {
		uint32_t head0 = 1;
		for (uint32_t q = b; q-- > 0;)
			if (n_loc[q]) { head0 = (t_key[size_t(bucket_base[q]) + n_loc[q] - 1] >> umi_bits) != (t_key[bucket_base[b]] >> umi_bits); break; }
		c = cg_loc[b] + head0;
	}
	cg_cnt[b] = c;
}
struct SsCompactCgArgs {
	const uint32_t *bucket_base, *n_loc, *prefix, *cg_loc, *cg_cnt, *cg_prefix;
	uint32_t n_buckets;
	const unsigned long long *t_key;
	const uint32_t *t_reads, *t_agg;
	unsigned long long *mol_key;
	uint32_t *mol_reads, *mol_mark, *mol_exon, *mol_intron;
	unsigned long long *cg_key;
	uint32_t *cg_mol_begin;
	uint32_t *out[6];        // n_all, n_req, reads_all, reads_req, exon reads, intron reads (total + 1 rows each)
	int umi_bits;
	uint32_t query_mask, n_cg;
	uint32_t dbg;            // timing probes (DROPEST_CG_DBG; results unusable): 1 no segmented sums, 2 no (cell, gene) output, 4 no molecule rows
	// the re-keyed form (ss_compact_cg_kernel<true>, after a merge): the rows are the sorted (new key, old row) pairs of the molecule table, equal
	// keys fold into one molecule (reads add, marks OR: Gene::merge / UMI::merge, Gene.cpp:26-58, UMI.cpp:15-19) on the way.  A "bucket" is then a
	// tile of tile_rows pairs whose ends are moved forward to the next change of key (rk_tile_begin), n_loc is unused, prefix / cg_cnt / cg_prefix
	// come from rk_counts_kernel + the scans.
	const unsigned long long *rk_key;
	const uint32_t *rk_idx, *old_reads, *old_mark, *old_exon, *old_intron;   // (old_exon / old_intron may be null: no exon / intron counts kept)
	uint32_t n_rows, tile_rows;
};
// first row of tile t of the re-keyed pairs: t * tile_rows moved forward until a new key starts there (no molecule straddles two tiles)
__device__ __forceinline__ uint32_t rk_tile_begin(const unsigned long long *__restrict__ key, uint32_t n, uint32_t tile_rows, uint32_t t) {
	const unsigned long long x64 = (unsigned long long)t * tile_rows;
	if (x64 >= n) return n;
	uint32_t x = uint32_t(x64);
	while (x && x < n && key[x] == key[x - 1]) ++x;
	return x;
}
// per tile: the molecules (key heads) and the (cell, gene) heads in it; one wave per tile
__global__ __launch_bounds__(256) void rk_counts_kernel(const unsigned long long *__restrict__ key, uint32_t n, uint32_t tile_rows, int umi_bits, uint32_t n_tiles,
                                                        uint32_t *__restrict__ mol_cnt, uint32_t *__restrict__ cg_cnt) {
	const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (t >= n_tiles) return;
	const uint32_t s = rk_tile_begin(key, n, tile_rows, t), e = rk_tile_begin(key, n, tile_rows, t + 1);
	uint32_t m = 0, c = 0;
	for (uint32_t r = s + lane; r < e; r += 64) {
		const unsigned long long k = __builtin_nontemporal_load(key + r), pk = r ? key[r - 1] : ~k;
		m += k != pk;
		c += r == 0 || (k >> umi_bits) != (pk >> umi_bits);
	}
#pragma unroll
	for (int d = 32; d; d >>= 1) { m += uint32_t(__shfl_down(int(m), d, 64)); c += uint32_t(__shfl_down(int(c), d, 64)); }
	if (lane == 0) { mol_cnt[t] = m; cg_cnt[t] = c; }
}
// The rows the fused kernel adds into with atomics -- the last (cell, gene) run of every bucket (the next bucket may continue it) -- and the
// sentinel row behind the table are cleared first; every other row is written with plain stores.
__global__ __launch_bounds__(256) void ss_cg_zero_borders_kernel(SsCompactCgArgs a) {
	const uint32_t b = blockIdx.x * 256 + threadIdx.x;
	if (b == 0) {
#pragma unroll
		for (int c = 0; c < 6; ++c) a.out[c][a.n_cg] = 0;
	}
	if (b >= a.n_buckets || !a.cg_cnt[b]) return;
	const uint32_t row = a.cg_prefix[b] + a.cg_cnt[b] - 1;   // (buckets or tiles alike)
#pragma unroll
	for (int c = 0; c < 6; ++c) a.out[c][row] = 0;
}
// One wave per bucket.  The (cell, gene) rows a wave completes go through a ring in LDS (128 rows per wave) and leave 64 rows at a time,
// one row per lane: whole lines on all eight output arrays.  (Written straight from the lanes where the runs end -- about every second
// lane -- the same rows cost 0.93 ms per 1e8 reads instead of ~0.3: half-filled store instructions, partial lines.)
template <bool RK>
__global__ __launch_bounds__(256) void ss_compact_cg_kernel(SsCompactCgArgs a) {
	constexpr uint32_t RING = 128;
	__shared__ unsigned long long s_key[4][RING];
	__shared__ uint32_t s_begin[4][RING], s_v[4][6][RING];
	const uint32_t w = threadIdx.x >> 6, b = blockIdx.x * 4 + w, lane = threadIdx.x & 63u;
	if (b >= a.n_buckets) return;
	uint32_t n, src, head0;
	const uint32_t dst = a.prefix[b], cgp = a.cg_prefix[b], runs_total = a.cg_cnt[b];
	if constexpr (RK) {
		src = rk_tile_begin(a.rk_key, a.n_rows, a.tile_rows, b);
		n = rk_tile_begin(a.rk_key, a.n_rows, a.tile_rows, b + 1) - src;
		if (!n) return;
		head0 = src == 0 || (a.rk_key[src - 1] >> a.umi_bits) != (a.rk_key[src] >> a.umi_bits);
	} else {
		n = a.n_loc[b];
		if (!n) return;
		src = a.bucket_base[b];
		head0 = runs_total - a.cg_loc[b];
	}
	const unsigned long long le = lane == 63u ? ~0ull : ((2ull << lane) - 1ull);   // lanes <= this one
	// the (cell, gene) run that is open at the end of the previous 64 rows: its sums so far (wave-uniform), run_base = heads seen before this chunk
	// (run id r: 0 = the run the previous bucket left open, else the r-th head of this bucket; its row = cgp + r - 1)
	uint32_t carry[6] = {0, 0, 0, 0, 0, 0}, run_base = 0, flushed = 0;   // flushed: rows of this bucket already written out (rows 0 .. flushed - 1)
	uint32_t m_base = 0;   // (re-keyed form) molecules of this tile before the chunk
	unsigned long long prev_key = 0;
	// the sums of a run that has ended: the run the previous bucket left open (r = 0) and this bucket's last run (which the next bucket may
	// continue) are added into their rows with atomics; every other run goes to the ring
	auto emit = [&](uint32_t r, bool border, const uint32_t (&v)[6]) {
		if (a.dbg & 2u) return;
		if (border) {
			const uint32_t row = cgp + r - 1u;
#pragma unroll
			for (int c = 0; c < 6; ++c) if (v[c]) atomicAdd(&a.out[c][row], v[c]);
		} else {
#pragma unroll
			for (int c = 0; c < 6; ++c) s_v[w][c][(r - 1u) & (RING - 1u)] = v[c];
		}
	};
	for (uint32_t j0 = 0; j0 < n; j0 += 64) {
		const uint32_t j = j0 + lane;
		const bool valid = j < n;
		unsigned long long key = 0; uint32_t reads = 0, exon = 0, intron = 0, mark = 0;
		if constexpr (RK) { if (valid) key = a.rk_key[size_t(src) + j]; }
		else if (valid) key = a.t_key[size_t(src) + j];
		unsigned long long before_key = (unsigned long long)__shfl_up((long long)key, 1, 64);
		if (lane == 0) before_key = prev_key;
		bool counts = valid;    // the row is a molecule of the output (re-keyed form: the first row of a run of equal keys, which takes the others in)
		uint32_t mi = j;        // its place in the bucket's / tile's part of the molecule table
		if constexpr (RK) {
			counts = valid && (j == 0 || key != before_key);
			const unsigned long long mm = __ballot(counts);
			mi = m_base + uint32_t(__popcll(mm & le)) - 1u;
			m_base += uint32_t(__popcll(mm));
			if (counts) {
				uint32_t o = a.rk_idx[size_t(src) + j];
				reads = a.old_reads[o]; mark = a.old_mark[o];
				if (a.old_exon) { exon = a.old_exon[o]; intron = a.old_intron[o]; }
				for (uint32_t q = j + 1; q < n && a.rk_key[size_t(src) + q] == key; ++q) {
					o = a.rk_idx[size_t(src) + q];
					reads += a.old_reads[o]; mark |= a.old_mark[o];
					if (a.old_exon) { exon += a.old_exon[o]; intron += a.old_intron[o]; }
				}
			}
		} else {
			uint32_t agg = 0;
			if (valid) { reads = a.t_reads[size_t(src) + j]; agg = a.t_agg[size_t(src) + j]; }
			exon = (agg >> 1) & 0x7FFFu; intron = (agg >> 16) & 0x7FFFu; mark = (agg & 1u) | (exon ? 2u : 0u) | (intron ? 4u : 0u);
		}
		if (counts && !(a.dbg & 4u)) {
			a.mol_key[size_t(dst) + mi] = key; a.mol_reads[size_t(dst) + mi] = reads; a.mol_mark[size_t(dst) + mi] = mark;
			if (!RK || a.mol_exon) { a.mol_exon[size_t(dst) + mi] = exon; a.mol_intron[size_t(dst) + mi] = intron; }
		}
		const unsigned long long cg = key >> a.umi_bits, before = before_key >> a.umi_bits;
		const bool head = valid && (j == 0 ? head0 != 0 : cg != before);
		const unsigned long long hm = __ballot(head), vm = __ballot(valid);
		const unsigned long long mine = hm & le;
		const uint32_t r = run_base + uint32_t(__popcll(mine));
		const uint32_t start = mine ? 63u - uint32_t(__builtin_clzll(mine)) : 0u;   // first lane of this lane's run inside the chunk
		const uint32_t req = (a.query_mask >> (mark & 7u)) & 1u;
		// segmented inclusive sums over the lanes of one run: n_all | n_req << 16 (at most 64 each per chunk), reads, requested reads, exon, intron
		uint32_t t[5] = {counts ? 1u | (req << 16) : 0u, counts ? reads : 0u, (counts && req) ? reads : 0u, counts ? exon : 0u, counts ? intron : 0u};
		if (!(a.dbg & 1u))
#pragma unroll
		for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
			const bool take = lane >= dlt && lane - dlt >= start;
#pragma unroll
			for (int c = 0; c < 5; ++c) { const uint32_t o = uint32_t(__shfl_up(int(t[c]), dlt, 64)); if (take) t[c] += o; }
		}
		uint32_t tot[6] = {t[0] & 0xFFFFu, t[0] >> 16, t[1], t[2], t[3], t[4]};   // (the two counts shared a word inside the chunk only: a run of any length)
		if (!mine) {   // the run came into the chunk open: what the earlier chunks saw of it
#pragma unroll
			for (int c = 0; c < 6; ++c) tot[c] += carry[c];
		}
		// the run that was open at the end of the previous chunk ended there if this chunk starts with a head: lane 0 hands it on
		if (j0 && lane == 0 && head) emit(run_base, run_base == 0, carry);
		if (head && !(a.dbg & 2u)) {
			if (r == runs_total) { a.cg_key[cgp + r - 1u] = cg; a.cg_mol_begin[cgp + r - 1u] = dst + mi; }   // the bucket's last run never passes the ring
			else { s_key[w][(r - 1u) & (RING - 1u)] = cg; s_begin[w][(r - 1u) & (RING - 1u)] = dst + mi; }
		}
		const bool next_valid = lane < 63u && ((vm >> (lane + 1u)) & 1ull), next_head = lane < 63u && ((hm >> (lane + 1u)) & 1ull);
		const bool last_row = valid && j == n - 1u;
		if (valid && (last_row || (next_valid && next_head))) emit(r, last_row || r == 0, tot);   // a run's last row inside the chunk
		// what stays open behind lane 63 (only when the bucket goes on)
#pragma unroll
		for (int c = 0; c < 6; ++c) carry[c] = uint32_t(__shfl(int(tot[c]), 63, 64));
		run_base += uint32_t(__popcll(hm));
		prev_key = (unsigned long long)__shfl((long long)key, 63, 64);
		// rows complete: the runs before the open one, and never the bucket's last run (run ids 1 .. done are rows 0 .. done - 1)
		const bool final = j0 + 64 >= n;
		uint32_t done = run_base ? run_base - 1u : 0u;
		if (final && runs_total && done > runs_total - 1u) done = runs_total - 1u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		while (!(a.dbg & 2u) && (done - flushed >= 64u || (final && done > flushed))) {
			const uint32_t cnt = done - flushed < 64u ? done - flushed : 64u;
			if (lane < cnt) {
				const uint32_t row = flushed + lane, sl = row & (RING - 1u);
				const unsigned long long k = s_key[w][sl];
				const uint32_t mb = s_begin[w][sl];
				uint32_t v[6];
#pragma unroll
				for (int c = 0; c < 6; ++c) v[c] = s_v[w][c][sl];
				a.cg_key[cgp + row] = k; a.cg_mol_begin[cgp + row] = mb;
#pragma unroll
				for (int c = 0; c < 6; ++c) a.out[c][cgp + row] = v[c];
			}
			flushed += cnt;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
}

}  // namespace dropest
