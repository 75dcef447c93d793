// k_cbhash.h -- cell-barcode table: open-addressed hash build, first-seen ordinals, first-seen cell ids.
//
// Replaces the per-read `_cell_ids_by_cb.emplace(cb, size)` of CellsDataContainer::add_record
// (Estimation/CellsDataContainer.cpp:64-69): cell id = rank of the barcode's first occurrence in stream
// order, including reads without a gene.  On the device that is
//   (1) insert every barcode into an open-addressed table and atomicMin the read ordinal per slot,
//   (2) flag the reads that ARE their barcode's first occurrence and prefix-sum the flags: the exclusive
//       prefix at a first occurrence is exactly the first-seen rank.
// Integer/HBM-latency work, no MFMA.  Table keys: 0 = empty (packed codes always carry a sentinel bit).
#pragma once

#include "util.h"

namespace dropest {

// One 16-byte slot per barcode: the key, its first-seen ordinal and its cell id share a cache line, so a probe
// that hits costs ONE random access.  `nfirst` stores ~ordinal: the all-zero pattern of a fresh table then means
// "empty key, no ordinal yet" and atomicMax(nfirst, ~r) is atomicMin(first, r).
struct CbSlot {
	unsigned long long key;   // packed barcode, 0 = empty
	uint32_t nfirst;          // ~(min read ordinal), 0 = unset
	uint32_t cell_id;         // first-seen rank (filled by cb_assign_ids)
};
static_assert(sizeof(CbSlot) == 16, "slot layout");

struct CbTable {
	CbSlot *slots;
	uint64_t mask;   // capacity - 1
	__device__ uint32_t first(uint32_t s) const { return ~slots[s].nfirst; }
};

// Statistics gathered while streaming the reads once (sizes the sort key; see pipeline).
struct IngestStats {
	unsigned long long umi_clean_min, umi_clean_max;   // over non-escaped UMI codes
	unsigned long long umi_escape_max_plus1;           // 1 + max escape id among UMIs, 0 if none
	unsigned long long cb_escape_count;                // number of reads with an escaped barcode
	unsigned int gene_max_plus1;                       // 1 + max gene id, 0 if no read has a gene
	unsigned int overflow;                             // probe chain exceeded the limit -> grow & retry
	unsigned int chr_max_plus1;                        // 1 + max chromosome id over the reads that count per chromosome
	unsigned int gene_chr_conflict;                    // some gene was seen on two chromosomes (or its id exceeds the table)
};

constexpr uint32_t GENE_CHR_UNSET = 0xFFFFFFFFu;

// Reads as they come out of a sharded run's exchange (k_misc.h: ExchangePack): w0 = barcode | UMI << cb_bits, w1 = gene | mark <<
// gene_bits | chromosome << (gene_bits + 3).  The kernels of the hot path read such records as they are (the `cb` / `umi` arguments
// both point at w0, `gene` at w1, `aux` is not read): unpacking them into four columns first was a pass of 36 bytes per read.
// cb_bits < 0: plain columns.
struct ReadPack {
	int cb_bits = -1, gene_bits = 0;
	__host__ __device__ bool on() const { return cb_bits >= 0; }
	__device__ unsigned long long cb(unsigned long long w0) const { return cb_bits >= 64 || cb_bits < 0 ? w0 : (w0 & ((1ull << cb_bits) - 1ull)); }
	__device__ unsigned long long umi(unsigned long long w0) const { return cb_bits >= 64 ? 0ull : (w0 >> cb_bits); }
	__device__ uint32_t gene(uint32_t w1) const { const uint32_t gm = (1u << gene_bits) - 1u, g = w1 & gm; return g == gm ? 0xFFFFFFFFu : g; }
	__device__ uint32_t aux(uint32_t w1) const { return (w1 >> (gene_bits + 3)) | (((w1 >> gene_bits) & 7u) << 16); }
};

constexpr uint32_t CB_MAX_PROBE = 8192;
constexpr uint64_t ESCAPE_BIT = 0x8000000000000000ull;
constexpr uint32_t NO_GENE = 0xFFFFFFFFu;

__device__ inline uint32_t cb_find_or_insert(const CbTable &t, unsigned long long k, uint64_t h, bool &ok) {
	for (uint32_t probe = 0; probe < CB_MAX_PROBE; ++probe) {
		// Plain load as a hint: a slot only ever goes 0 -> key, so a stale 0 merely sends us to the CAS,
		// whose return value is authoritative.
		unsigned long long cur = t.slots[h].key;
		if (cur == k) return uint32_t(h);
		if (cur == 0ull) {
			unsigned long long prev = atomicCAS(&t.slots[h].key, 0ull, k);
			if (prev == 0ull || prev == k) return uint32_t(h);
		}
		h = (h + 1) & t.mask;
	}
	ok = false;
	return 0;
}

// The same for several keys of one thread at once: every round puts ONE request per unresolved key in flight before it waits
// for any of them (one after the other, the keys of a thread cost a memory round trip each -- and nearly every wave has a
// lane that needs one).  seen[j] = what the caller's own first load found at h[j] (0: the slot looked empty, anything else:
// another barcode).  Occupied slots are walked with plain loads -- a compare-and-swap as the probe serialises the reads of a
// popular barcode that sits behind a collision on one cache line (measured: 1.5 -> 2.7 ms without the hot list) -- and only a
// slot that looked empty gets the compare-and-swap, whose return value is authoritative.
template <int ILP>
__device__ inline void cb_resolve_together(const CbTable &t, const unsigned long long (&k)[ILP], uint64_t (&h)[ILP],
                                           unsigned long long (&seen)[ILP], uint32_t pending, uint32_t (&slot)[ILP], bool &ok) {
	for (uint32_t probe = 0; pending && probe < CB_MAX_PROBE; ++probe) {
		unsigned long long cur[ILP];
		uint32_t claim = 0;
#pragma unroll
		for (int j = 0; j < ILP; ++j) {   // next slot of every key that stands on an occupied one
			cur[j] = 0ull;
			if (((pending >> j) & 1u) && seen[j] != 0ull) { h[j] = (h[j] + 1) & t.mask; cur[j] = t.slots[h[j]].key; }
		}
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			if (!((pending >> j) & 1u)) continue;
			if (cur[j] == k[j]) { slot[j] = uint32_t(h[j]); pending &= ~(1u << j); }
			else if (cur[j] == 0ull) claim |= 1u << j;
			else seen[j] = cur[j];
		}
		unsigned long long prev[ILP];
#pragma unroll
		for (int j = 0; j < ILP; ++j) if ((claim >> j) & 1u) prev[j] = atomicCAS(&t.slots[h[j]].key, 0ull, k[j]);
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			if (!((claim >> j) & 1u)) continue;
			if (prev[j] == 0ull || prev[j] == k[j]) { slot[j] = uint32_t(h[j]); pending &= ~(1u << j); }
			else seen[j] = prev[j];   // somebody else's barcode got there first: walk on
		}
	}
	if (pending) ok = false;
}

__device__ inline uint32_t cb_find(const CbTable &t, unsigned long long k) {   // 0xFFFFFFFF if absent
	uint64_t h = mix64(k) & t.mask;
	for (uint32_t probe = 0; probe < CB_MAX_PROBE; ++probe) {
		unsigned long long cur = t.slots[h].key;
		if (cur == k) return uint32_t(h);
		if (cur == 0ull) return 0xFFFFFFFFu;
		h = (h + 1) & t.mask;
	}
	return 0xFFFFFFFFu;
}

// Per-thread accumulator of the ingest statistics: the part of a read's (umi, gene, aux) that sizes the sort key and decides
// whether the chromosome is a function of the gene.  Used by build_keys when the key layout was planned from a sample and the exact
// statistics ride along with the key pass (cb_insert then reads the barcodes only), and by the sample pass itself.
struct IngestAcc {
	unsigned long long umin = ~0ull, umax = 0ull, uesc = 0ull;
	uint32_t gmax = 0, cmax = 0;
	bool chr_conflict = false;
	__device__ inline void add(unsigned long long u, uint32_t g, uint32_t a) {
		if (u & ESCAPE_BIT) { const unsigned long long id1 = (u & ~ESCAPE_BIT) + 1; uesc = id1 > uesc ? id1 : uesc; }
		else { umin = u < umin ? u : umin; umax = u > umax ? u : umax; }
		if (g != NO_GENE && g + 1 > gmax) gmax = g + 1;
		const uint32_t chr = a & 0xFFFFu, mark = (a >> 16) & 0xFFu;
		if ((g == NO_GENE || (mark & 6u)) && chr + 1 > cmax) cmax = chr + 1;
	}
	// Is the chromosome a function of the gene?  (Only reads that are counted per chromosome matter: reads with a gene and an
	// exon / intron mark, CellsDataContainer.cpp:73-78, :312-321.)  One table look-up per read; the CAS runs once per gene.
	// `loaded`: the entry as the caller read it earlier (with other loads in flight)
	__device__ inline void check_chromosome(uint32_t g, uint32_t a, uint32_t *__restrict__ gene_chr, uint32_t gene_chr_cap, uint32_t loaded) {
		if (g == NO_GENE || !((a >> 16) & 6u)) return;
		if (g >= gene_chr_cap) { chr_conflict = true; return; }
		const uint32_t chr = a & 0xFFFFu;
		uint32_t cur = loaded;
		if (cur == GENE_CHR_UNSET) cur = atomicCAS(&gene_chr[g], GENE_CHR_UNSET, chr), cur = cur == GENE_CHR_UNSET ? chr : cur;
		if (cur != chr) chr_conflict = true;
	}
	__device__ inline void commit(IngestStats *stats) {
		const unsigned long long mn = wave_reduce_min_u64(umin), mx = wave_reduce_max_u64(umax), es = wave_reduce_max_u64(uesc);
		const unsigned long long g64 = wave_reduce_max_u64(gmax), c64 = wave_reduce_max_u64(cmax);
		const unsigned long long conf = wave_reduce_max_u64(chr_conflict ? 1ull : 0ull);
		if (lane_id() == 0) {
			if (mn != ~0ull) atomicMin(&stats->umi_clean_min, mn);
			if (mx != 0ull) atomicMax(&stats->umi_clean_max, mx);
			if (es) atomicMax(&stats->umi_escape_max_plus1, es);
			if (g64) atomicMax(&stats->gene_max_plus1, uint32_t(g64));
			if (c64) atomicMax(&stats->chr_max_plus1, uint32_t(c64));
			if (conf) atomicMax(&stats->gene_chr_conflict, 1u);
		}
	}
};

// The statistics of every `stride`-th read: what the key layout is planned from when the exact ones are gathered by the key pass.
__global__ __launch_bounds__(256) void ingest_sample_stats_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene,
                                                                  const uint32_t *__restrict__ aux, uint32_t n, uint32_t stride, IngestStats *stats,
                                                                  ReadPack pk = ReadPack{}) {
	IngestAcc acc;
	for (uint64_t j = uint64_t(blockIdx.x) * 256 + threadIdx.x; j * stride < n; j += uint64_t(gridDim.x) * 256) {
		if (pk.on()) { const uint32_t w1 = gene[j * stride]; acc.add(pk.umi(umi[j * stride]), pk.gene(w1), pk.aux(w1)); }
		else acc.add(umi[j * stride], gene[j * stride], aux[j * stride]);
	}
	// the four waves of a workgroup meet in LDS first: one set of atomics per workgroup on the six shared words
	__shared__ unsigned long long w_min[4], w_max[4], w_esc[4];
	__shared__ uint32_t w_g[4], w_c[4];
	const unsigned long long mn = wave_reduce_min_u64(acc.umin), mx = wave_reduce_max_u64(acc.umax), es = wave_reduce_max_u64(acc.uesc);
	const unsigned long long g64 = wave_reduce_max_u64(acc.gmax), c64 = wave_reduce_max_u64(acc.cmax);
	const uint32_t w = threadIdx.x >> 6;
	if (lane_id() == 0) { w_min[w] = mn; w_max[w] = mx; w_esc[w] = es; w_g[w] = uint32_t(g64); w_c[w] = uint32_t(c64); }
	__syncthreads();
	if (threadIdx.x == 0) {
		IngestAcc all;
		for (int k = 0; k < 4; ++k) {
			all.umin = w_min[k] < all.umin ? w_min[k] : all.umin; all.umax = w_max[k] > all.umax ? w_max[k] : all.umax;
			all.uesc = w_esc[k] > all.uesc ? w_esc[k] : all.uesc; all.gmax = w_g[k] > all.gmax ? w_g[k] : all.gmax; all.cmax = w_c[k] > all.cmax ? w_c[k] : all.cmax;
		}
		if (all.umin != ~0ull) atomicMin(&stats->umi_clean_min, all.umin);
		if (all.umax != 0ull) atomicMax(&stats->umi_clean_max, all.umax);
		if (all.uesc) atomicMax(&stats->umi_escape_max_plus1, all.uesc);
		if (all.gmax) atomicMax(&stats->gene_max_plus1, all.gmax);
		if (all.cmax) atomicMax(&stats->chr_max_plus1, all.cmax);
	}
}

// The ranges of the read arrays one launch covers.  One range = the whole stream (a plain context).  A sharded run that receives its reads in
// chunks (csrc/shard_run.h: the all-to-all overlapped with the table build) hands the kernels chunk c of every source's block: ranges of
// the SAME arrays, so a read's index stays its first-seen ordinal.
struct CbRanges {
	uint32_t n = 0;
	uint32_t off[64], cnt[64];
	static CbRanges whole(uint32_t n_reads) { CbRanges r; r.n = 1; r.off[0] = 0; r.cnt[0] = n_reads; return r; }
	__host__ __device__ uint64_t total() const { uint64_t t = 0; for (uint32_t q = 0; q < n; ++q) t += cnt[q]; return t; }
};
// per-thread accumulators of the insert kernels (statistics + flags), committed once per launch
struct CbInsertAcc {
	unsigned long long umin = ~0ull, umax = 0ull, uesc = 0ull, cbesc = 0ull;
	uint32_t gmax = 0, cmax = 0;
	bool ok = true, chr_conflict = false;
	__device__ inline void commit(IngestStats *stats) {
		umin = wave_reduce_min_u64(umin); umax = wave_reduce_max_u64(umax); uesc = wave_reduce_max_u64(uesc);
		cbesc = wave_reduce_add_u64(cbesc);
		const unsigned long long g64 = wave_reduce_max_u64(gmax), c64 = wave_reduce_max_u64(cmax);
		const unsigned long long conf = wave_reduce_max_u64(chr_conflict ? 1ull : 0ull), bad = wave_reduce_max_u64(ok ? 0ull : 1ull);
		if (lane_id() == 0) {
			if (umin != ~0ull) atomicMin(&stats->umi_clean_min, umin);
			if (umax != 0ull) atomicMax(&stats->umi_clean_max, umax);
			if (uesc) atomicMax(&stats->umi_escape_max_plus1, uesc);
			if (cbesc) atomicAdd(&stats->cb_escape_count, cbesc);
			if (g64) atomicMax(&stats->gene_max_plus1, uint32_t(g64));
			if (bad) atomicMax(&stats->overflow, 1u);
			if (c64) atomicMax(&stats->chr_max_plus1, uint32_t(c64));
			if (conf) atomicMax(&stats->gene_chr_conflict, 1u);
		}
	}
	// the part of a read's (umi, gene, aux) that sizes the sort key and decides whether the chromosome is a function of the gene
	__device__ inline void add_read(unsigned long long u, uint32_t g, uint32_t a, uint32_t *__restrict__ gene_chr, uint32_t gene_chr_cap) {
		if (u & ESCAPE_BIT) { const unsigned long long id1 = (u & ~ESCAPE_BIT) + 1; uesc = id1 > uesc ? id1 : uesc; }
		else { umin = u < umin ? u : umin; umax = u > umax ? u : umax; }
		if (g != NO_GENE && g + 1 > gmax) gmax = g + 1;
		// Is the chromosome a function of the gene?  (Only reads that are counted per chromosome matter: gene-less reads and reads with an
		// exon / intron mark, CellsDataContainer.cpp:73-78, :312-321.)
		const uint32_t chr = a & 0xFFFFu, mark = (a >> 16) & 0xFFu;
		if (g == NO_GENE) { if (chr + 1 > cmax) cmax = chr + 1; }
		else if (mark & 6u) {
			if (chr + 1 > cmax) cmax = chr + 1;
			if (g >= gene_chr_cap) chr_conflict = true;
			else {
				uint32_t cur2 = gene_chr[g];            // L1/L2-hot table; the CAS runs once per gene
				if (cur2 == GENE_CHR_UNSET) cur2 = atomicCAS(&gene_chr[g], GENE_CHR_UNSET, chr), cur2 = cur2 == GENE_CHR_UNSET ? chr : cur2;
				if (cur2 != chr) chr_conflict = true;
			}
		}
	}
};

// One pass over (cb, umi, gene): table insert + first ordinal + ingest statistics.  Each thread takes FOUR CONSECUTIVE
// reads per iteration: with 16-byte-aligned arrays (VEC) every streaming access is a 16-byte load / store per lane (two
// barcodes, two UMIs, four gene ids, four aux words, four slot indices) -- with one read per lane the 4- and 8-byte
// accesses left the kernel at 1.4 TB/s whatever else it did (the table probes cost nothing: measured with the probe
// switched off) -- and the four table probes are independent 16-byte loads in flight together.
// One range of reads: the arrays point at the range's first read, whose first-seen ordinal is ord_base.
template <int THREADS, bool VEC, bool STATS>
__device__ inline void cb_insert_range(const unsigned long long *__restrict__ cb, const unsigned long long *__restrict__ umi,
                                       const uint32_t *__restrict__ gene, const uint32_t *__restrict__ aux, uint32_t n, uint32_t ord_base, const CbTable &t,
                                       uint32_t *__restrict__ slot_out, uint32_t *__restrict__ gene_chr, uint32_t gene_chr_cap, const ReadPack &pk, CbInsertAcc &acc) {
	constexpr int ILP = 4;
	const uint64_t stride = uint64_t(gridDim.x) * THREADS * ILP;
	for (uint64_t r0 = (uint64_t(blockIdx.x) * THREADS + threadIdx.x) * ILP; r0 < n; r0 += stride) {
		unsigned long long k[ILP], u[ILP];
		uint64_t h[ILP];
		uint4 v[ILP];
		uint32_t g[ILP], a[ILP], sl[ILP];
		const bool full = r0 + ILP <= n;
		if (VEC && full) {
			const ulonglong2 k01 = *reinterpret_cast<const ulonglong2 *>(cb + r0), k23 = *reinterpret_cast<const ulonglong2 *>(cb + r0 + 2);
			k[0] = k01.x; k[1] = k01.y; k[2] = k23.x; k[3] = k23.y;
		} else {
#pragma unroll
			for (int j = 0; j < ILP; ++j) k[j] = r0 + j < n ? cb[r0 + j] : 0ull;
		}
		unsigned long long kraw[ILP];   // (packed records: the UMI sits above the barcode in the same word)
#pragma unroll
		for (int j = 0; j < ILP; ++j) { kraw[j] = k[j]; k[j] = pk.cb(k[j]); }
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			h[j] = mix64(k[j]) & t.mask;
			v[j] = *reinterpret_cast<const uint4 *>(&t.slots[h[j]]);   // key + ~first in one access
		}
		if (STATS) {
			if (pk.on()) {
#pragma unroll
				for (int j = 0; j < ILP; ++j) {
					const uint32_t w1 = r0 + j < n ? gene[r0 + j] : 0xFFFFFFFFu;
					u[j] = pk.umi(kraw[j]); g[j] = r0 + j < n ? pk.gene(w1) : NO_GENE; a[j] = r0 + j < n ? pk.aux(w1) : 0u;
				}
			} else if (VEC && full) {
				const ulonglong2 u01 = *reinterpret_cast<const ulonglong2 *>(umi + r0), u23 = *reinterpret_cast<const ulonglong2 *>(umi + r0 + 2);
				const uint4 g4 = *reinterpret_cast<const uint4 *>(gene + r0), a4 = *reinterpret_cast<const uint4 *>(aux + r0);
				u[0] = u01.x; u[1] = u01.y; u[2] = u23.x; u[3] = u23.y;
				g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
				a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
			} else {
#pragma unroll
				for (int j = 0; j < ILP; ++j) {
					const uint64_t r = r0 + j;
					u[j] = r < n ? umi[r] : 0ull;
					g[j] = r < n ? gene[r] : NO_GENE;
					a[j] = r < n ? aux[r] : 0u;
				}
			}
		}
		uint32_t pending = 0, hinted = 0;
		unsigned long long seen[ILP];
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			seen[j] = (unsigned long long)v[j].x | ((unsigned long long)v[j].y << 32);
			sl[j] = uint32_t(h[j]);
			if (r0 + j < n) {
				if (seen[j] == k[j]) hinted |= 1u << j;   // found at once: v[j].z is a recent value of its first ordinal
				else pending |= 1u << j;
			}
		}
		if (pending) cb_resolve_together<ILP>(t, k, h, seen, pending, sl, acc.ok);
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			const uint64_t r = r0 + j;
			if (r >= n) continue;
			const uint32_t s = sl[j], ord = ord_base + uint32_t(r);
			// stale (too large) hints only cost an extra atomic; ordinals only ever decrease
			const uint32_t first_hint = ((hinted >> j) & 1u) ? ~v[j].z : 0xFFFFFFFFu;
			if (ord < first_hint) atomicMax(&t.slots[s].nfirst, ~ord);
			if (k[j] & ESCAPE_BIT) ++acc.cbesc;
			if (STATS) acc.add_read(u[j], g[j], a[j], gene_chr, gene_chr_cap);
		}
		if (VEC && full) stream_store_u32x4(slot_out + r0, sl[0], sl[1], sl[2], sl[3]);
		else {
#pragma unroll
			for (int j = 0; j < ILP; ++j) if (r0 + j < n) slot_out[r0 + j] = sl[j];
		}
	}
}
template <int THREADS, bool VEC, bool STATS = true>
__global__ __launch_bounds__(THREADS) void cb_insert_kernel(const unsigned long long *__restrict__ cb,
                                                            const unsigned long long *__restrict__ umi,
                                                            const uint32_t *__restrict__ gene,
                                                            const uint32_t *__restrict__ aux, CbRanges rg, CbTable t,
                                                            uint32_t *__restrict__ slot_out, uint32_t *__restrict__ gene_chr,
                                                            uint32_t gene_chr_cap, IngestStats *stats, ReadPack pk = ReadPack{}) {
	CbInsertAcc acc;
	for (uint32_t q = 0; q < rg.n; ++q) {
		// (a range need not start on a 16-byte boundary of the arrays: its first reads up to one take the scalar accesses)
		uint32_t o = rg.off[q], cnt = rg.cnt[q];
		const uint32_t head = VEC ? (cnt < ((4u - (o & 3u)) & 3u) ? cnt : ((4u - (o & 3u)) & 3u)) : cnt;
		if (head) cb_insert_range<THREADS, false, STATS>(cb + o, umi + o, gene + o, aux + o, head, o, t, slot_out + o, gene_chr, gene_chr_cap, pk, acc);
		o += head; cnt -= head;
		if (VEC && cnt) cb_insert_range<THREADS, true, STATS>(cb + o, umi + o, gene + o, aux + o, cnt, o, t, slot_out + o, gene_chr, gene_chr_cap, pk, acc);
	}
	acc.commit(stats);
}

// The gene -> chromosome table seeded from every `stride`-th read (same protocol as cb_insert's own check).  A gene's entry
// goes UNSET -> chromosome through one compare-and-swap, but every read that still SEES it unset sends one: when the big pass
// starts cold, the first tiles of all waves do that for the popular genes at once -- tens of thousands of atomics on one
// address.  A few thousand reads ahead of it set the entries of all genes that matter for that.
__global__ __launch_bounds__(256) void gene_chr_seed_kernel(const uint32_t *__restrict__ gene, const uint32_t *__restrict__ aux, uint32_t n, uint32_t stride,
                                                            uint32_t *__restrict__ gene_chr, uint32_t gene_chr_cap, IngestStats *stats, ReadPack pk = ReadPack{}) {
	bool conflict = false;
	for (uint64_t j = uint64_t(blockIdx.x) * 256 + threadIdx.x; j * stride < n; j += uint64_t(gridDim.x) * 256) {
		const uint32_t w = gene[j * stride];
		const uint32_t g = pk.on() ? pk.gene(w) : w, a = pk.on() ? pk.aux(w) : aux[j * stride];
		if (g == NO_GENE || !((a >> 16) & 6u)) continue;
		if (g >= gene_chr_cap) { conflict = true; continue; }
		const uint32_t chr = a & 0xFFFFu;
		uint32_t cur = gene_chr[g];
		if (cur == GENE_CHR_UNSET) cur = atomicCAS(&gene_chr[g], GENE_CHR_UNSET, chr), cur = cur == GENE_CHR_UNSET ? chr : cur;
		if (cur != chr) conflict = true;
	}
	if (conflict) atomicMax(&stats->gene_chr_conflict, 1u);
}

// Distinct barcodes among every `stride`-th read (inserted into a scratch table): sizes the real table.  A table of
// n / 2 slots for a stream whose 1e8 reads carry 3e6 barcodes is 1 GB of 16-byte slots at 5 % load -- every probe of a
// rare barcode a certain HBM miss, 1 GB to clear and 1 GB to scan for the occupied slots; sized from the sample it is
// 128 MB and lives in the 256 MB Infinity Cache.
// count_every: only every count_every-th sampled read adds (count_every) to its barcode's occurrence count -- the counts pick the hot list,
// nothing else; a real cell of a 10x run collects ~300 of these atomics on ONE word, and same-address atomics from all over the device
// run one after the other (cb_sample was 0.2 ms per 1e8 reads for a 20 us stream).
__global__ __launch_bounds__(256) void cb_sample_distinct_kernel(const unsigned long long *__restrict__ cb, uint32_t n, uint32_t stride,
                                                                 CbTable t, uint32_t *__restrict__ distinct, ReadPack pk = ReadPack{}, uint32_t count_every = 1) {
	uint32_t mine = 0;
	for (uint64_t j = uint64_t(blockIdx.x) * 256 + threadIdx.x; j * stride < n; j += uint64_t(gridDim.x) * 256) {
		const unsigned long long k = pk.cb(cb[j * stride]);
		uint64_t h = mix64(k) & t.mask;
		bool found = false;
		for (uint32_t probe = 0; probe < CB_MAX_PROBE && !found; ++probe) {
			const unsigned long long cur = t.slots[h].key;
			if (cur == k) found = true;
			else if (cur == 0ull) {
				const unsigned long long prev = atomicCAS(&t.slots[h].key, 0ull, k);
				if (prev == 0ull) { ++mine; found = true; }
				else if (prev == k) found = true;
			}
			if (!found) h = (h + 1) & t.mask;
		}
		if (found && j % count_every == 0) atomicAdd(&t.slots[h].nfirst, count_every);   // occurrences in the sample: the hot barcodes are picked by it
	}
	const unsigned long long tot = wave_reduce_add_u64(mine);
	__shared__ uint32_t block_new;   // (one atomic per workgroup on the one counter, not one per wave)
	if (threadIdx.x == 0) block_new = 0;
	__syncthreads();
	if (lane_id() == 0 && tot) atomicAdd(&block_new, uint32_t(tot));
	__syncthreads();
	if (threadIdx.x == 0 && block_new) atomicAdd(distinct, block_new);
}

// ---- hot barcodes -------------------------------------------------------------------------------------------------
// A 10x / inDrop stream puts > 90 % of its reads on a few thousand barcodes.  cb_insert and build_keys are bound by the
// number of L2 requests -- every table probe is its own request, 1.5e8 of them for 1e8 reads -- so the barcodes the sample
// saw most often (at most CB_HOT_MAX) get a read-only hash table in LDS: a hit costs no memory request at all.
constexpr uint32_t CB_HOT_MAX = 4096, CB_HOT_LDS = 8192, CB_HOT_FLAG = 0x80000000u;
constexpr int CB_HOT_LEVELS = 24;
// 4, 6, 8, 12, 16, 24, ... sample hits (steps of 1.33-1.5x; with the steps of 4x of earlier rounds the C2 stream listed 2 262 of its
// 5 000 real cells, now 3 240; C3 at 1e9 reads 401 -> 1 365 of 50 000).  DROPEST_CB_TRACE=1 prints the counts per threshold.
__device__ __host__ inline uint32_t cb_hot_threshold(int level) { return ((level & 1) ? 6u : 4u) << (level >> 1); }
// how many sampled barcodes reach each threshold
__global__ __launch_bounds__(256) void cb_hot_count_kernel(CbTable ts, uint32_t *__restrict__ counts) {
	uint32_t c[CB_HOT_LEVELS] = {};
	for (uint64_t s = uint64_t(blockIdx.x) * 256 + threadIdx.x; s <= ts.mask; s += uint64_t(gridDim.x) * 256) {
		if (ts.slots[s].key == 0ull) continue;
		const uint32_t n = ts.slots[s].nfirst;
#pragma unroll
		for (int l = 0; l < CB_HOT_LEVELS; ++l) c[l] += n >= cb_hot_threshold(l);
	}
	// one global atomic per level and WORKGROUP: 24 counters took 98 000 atomics from 4 096 waves -- 0.40 ms for a 16 us scan
	__shared__ uint32_t block_c[CB_HOT_LEVELS];
	if (threadIdx.x < CB_HOT_LEVELS) block_c[threadIdx.x] = 0;
	__syncthreads();
#pragma unroll
	for (int l = 0; l < CB_HOT_LEVELS; ++l) {
		const unsigned long long tot = wave_reduce_add_u64(c[l]);
		if (lane_id() == 0 && tot) atomicAdd(&block_c[l], uint32_t(tot));
	}
	__syncthreads();
	if (threadIdx.x < CB_HOT_LEVELS && block_c[threadIdx.x]) atomicAdd(&counts[threadIdx.x], block_c[threadIdx.x]);
}
__global__ __launch_bounds__(256) void cb_hot_collect_kernel(CbTable ts, uint32_t threshold, unsigned long long *__restrict__ hot_key,
                                                             uint32_t *__restrict__ n_hot) {
	for (uint64_t s = uint64_t(blockIdx.x) * 256 + threadIdx.x; s <= ts.mask; s += uint64_t(gridDim.x) * 256) {
		const unsigned long long k = ts.slots[s].key;
		if (k == 0ull || ts.slots[s].nfirst < threshold) continue;
		const uint32_t at = atomicAdd(n_hot, 1u);
		if (at < CB_HOT_MAX) hot_key[at] = k;
	}
}
// the hot barcodes take their slots in the real table first
__global__ __launch_bounds__(256) void cb_hot_preinsert_kernel(const unsigned long long *__restrict__ hot_key, uint32_t n_hot, CbTable t,
                                                               uint32_t *__restrict__ hot_slot, uint32_t *__restrict__ overflow) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_hot) return;
	bool ok = true;
	hot_slot[i] = cb_find_or_insert(t, hot_key[i], mix64(hot_key[i]) & t.mask, ok);
	if (!ok) atomicMax(overflow, 1u);
}

// cb_insert with the LDS table of the hot barcodes.  One workgroup of 1024 threads per CU (the table takes 128 KB of the
// CU's 160 KB); slot_out[r] = CB_HOT_FLAG | hot index for a hit, the table slot otherwise.  Per-entry minimum of the read
// ordinals seen by this workgroup keeps the atomics on the table to the few reads that lower it.
struct CbHot { const unsigned long long *key; const uint32_t *slot; uint32_t n; };
// EXP (timing probes, results not usable; launched only by builds with -DDROPEST_CBI_PROBE, see dropest_ctx::build_cb_table):
// 1 no table probe, 2 no LDS look-up, 4 no atomics
struct CbHotLds { unsigned long long *lk; uint32_t *li, *lf; };   // [CB_HOT_LDS] key (0 = empty), hot index, smallest ordinal seen by this workgroup
// one range of reads (see cb_insert_range): the LDS table is built once per launch, the ranges share it
template <bool VEC, bool STATS, int EXP>
__device__ inline void cb_insert_hot_range(const unsigned long long *__restrict__ cb, const unsigned long long *__restrict__ umi,
                                           const uint32_t *__restrict__ gene, const uint32_t *__restrict__ aux, uint32_t n, uint32_t ord_base, const CbTable &t,
                                           const CbHot &hot, const CbHotLds &L, uint32_t *__restrict__ slot_out, uint32_t *__restrict__ gene_chr,
                                           uint32_t gene_chr_cap, const ReadPack &pk, CbInsertAcc &acc) {
	constexpr int ILP = 4, THREADS = 1024;
	unsigned long long *lk = L.lk; uint32_t *li = L.li, *lf = L.lf;
	const uint64_t stride = uint64_t(gridDim.x) * THREADS * ILP;
	// the barcodes of the NEXT tile are in flight while this one is worked on (4 waves per SIMD do not hide the load by themselves)
	auto load_cb = [&](uint64_t r0, unsigned long long (&kk)[ILP]) {
		if (VEC && r0 + ILP <= n) {
			const ulonglong2 k01 = stream_load_u64x2(cb + r0), k23 = stream_load_u64x2(cb + r0 + 2);   // (the table lines stay in L2, the stream passes)
			kk[0] = k01.x; kk[1] = k01.y; kk[2] = k23.x; kk[3] = k23.y;
		} else {
#pragma unroll
			for (int j = 0; j < ILP; ++j) kk[j] = r0 + j < n ? cb[r0 + j] : 0ull;
		}
	};
	unsigned long long k_next[ILP] = {0ull, 0ull, 0ull, 0ull};
	{
		const uint64_t first = (uint64_t(blockIdx.x) * THREADS + threadIdx.x) * ILP;
		if (first < n) load_cb(first, k_next);
	}
	for (uint64_t r0 = (uint64_t(blockIdx.x) * THREADS + threadIdx.x) * ILP; r0 < n; r0 += stride) {
		unsigned long long k[ILP], u[ILP];
		uint64_t h[ILP];
		uint32_t g[ILP], a[ILP], sl[ILP], hit[ILP];
		const bool full = r0 + ILP <= n;
		unsigned long long kraw[ILP];   // (packed records: the UMI sits above the barcode in the same word)
#pragma unroll
		for (int j = 0; j < ILP; ++j) { kraw[j] = k_next[j]; k[j] = pk.cb(k_next[j]); }
		if (r0 + stride < n) load_cb(r0 + stride, k_next);
		if (STATS && pk.on()) {
#pragma unroll
			for (int j = 0; j < ILP; ++j) {
				const uint32_t w1 = r0 + j < n ? gene[r0 + j] : 0xFFFFFFFFu;
				u[j] = pk.umi(kraw[j]); g[j] = r0 + j < n ? pk.gene(w1) : NO_GENE; a[j] = r0 + j < n ? pk.aux(w1) : 0u;
			}
		} else if (VEC && full) {
			if (STATS) {
				const ulonglong2 u01 = *reinterpret_cast<const ulonglong2 *>(umi + r0), u23 = *reinterpret_cast<const ulonglong2 *>(umi + r0 + 2);
				const uint4 g4 = *reinterpret_cast<const uint4 *>(gene + r0), a4 = *reinterpret_cast<const uint4 *>(aux + r0);
				u[0] = u01.x; u[1] = u01.y; u[2] = u23.x; u[3] = u23.y;
				g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
				a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
			}
		} else {
#pragma unroll
			for (int j = 0; j < ILP; ++j) {
				const uint64_t r = r0 + j;
				if (STATS) { u[j] = r < n ? umi[r] : 0ull; g[j] = r < n ? gene[r] : NO_GENE; a[j] = r < n ? aux[r] : 0u; }
			}
		}
#pragma unroll
		for (int j = 0; j < ILP; ++j) {   // LDS look-ups of the four barcodes
			h[j] = mix64(k[j]);
			hit[j] = 0xFFFFFFFFu;
			if (r0 + j >= n || (EXP & 2)) continue;
			uint32_t i = uint32_t(h[j] >> 40) & (CB_HOT_LDS - 1);
			for (;;) {
				const unsigned long long e = lk[i];
				if (e == k[j]) { hit[j] = i; break; }
				if (e == 0ull) break;
				i = (i + 1) & (CB_HOT_LDS - 1);
			}
		}
		uint4 v[ILP];
#pragma unroll
		for (int j = 0; j < ILP; ++j) {   // the others probe the table: independent 16-byte loads in flight together
			h[j] &= t.mask;
			v[j] = make_uint4(0u, 0u, 0u, 0u);
			if (!(EXP & 1) && r0 + j < n && hit[j] == 0xFFFFFFFFu) v[j] = *reinterpret_cast<const uint4 *>(&t.slots[h[j]]);
		}
		uint32_t pending = 0, hinted = 0;
		unsigned long long seen[ILP];
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			seen[j] = (unsigned long long)v[j].x | ((unsigned long long)v[j].y << 32);
			sl[j] = uint32_t(h[j]);
			if (!(EXP & 1) && r0 + j < n && hit[j] == 0xFFFFFFFFu) {
				if (seen[j] == k[j]) hinted |= 1u << j;   // found at once: v[j].z is a recent value of its first ordinal
				else pending |= 1u << j;
			}
		}
		if (pending) cb_resolve_together<ILP>(t, k, h, seen, pending, sl, acc.ok);
#pragma unroll
		for (int j = 0; j < ILP; ++j) {
			const uint64_t r = r0 + j;
			if (r >= n) continue;
			const uint32_t ord = ord_base + uint32_t(r);
			if (hit[j] != 0xFFFFFFFFu) {
				const uint32_t e = hit[j], hi = li[e];
				sl[j] = CB_HOT_FLAG | hi;
				if (!(EXP & 4) && ord < lf[e]) {
					const uint32_t old = atomicMin(&lf[e], ord);
					if (ord < old) atomicMax(&t.slots[hot.slot[hi]].nfirst, ~ord);
				}
			} else {
				const uint32_t first_hint = ((hinted >> j) & 1u) ? ~v[j].z : 0xFFFFFFFFu;
				if (!(EXP & 5) && ord < first_hint) atomicMax(&t.slots[sl[j]].nfirst, ~ord);
			}
			if (k[j] & ESCAPE_BIT) ++acc.cbesc;
			if (STATS) acc.add_read(u[j], g[j], a[j], gene_chr, gene_chr_cap);
		}
		if (VEC && full) stream_store_u32x4(slot_out + r0, sl[0], sl[1], sl[2], sl[3]);
		else {
#pragma unroll
			for (int j = 0; j < ILP; ++j) if (r0 + j < n) slot_out[r0 + j] = sl[j];
		}
	}
}
template <bool VEC, bool STATS = true, int EXP = 0>
__global__ __launch_bounds__(1024) void cb_insert_hot_kernel(const unsigned long long *__restrict__ cb,
                                                             const unsigned long long *__restrict__ umi,
                                                             const uint32_t *__restrict__ gene,
                                                             const uint32_t *__restrict__ aux, CbRanges rg, CbTable t, CbHot hot,
                                                             uint32_t *__restrict__ slot_out, uint32_t *__restrict__ gene_chr,
                                                             uint32_t gene_chr_cap, IngestStats *stats, ReadPack pk = ReadPack{}) {
	constexpr int THREADS = 1024;
	extern __shared__ __attribute__((aligned(16))) unsigned char cb_smem[];
	CbHotLds L;
	L.lk = reinterpret_cast<unsigned long long *>(cb_smem);
	L.li = reinterpret_cast<uint32_t *>(L.lk + CB_HOT_LDS);
	L.lf = L.li + CB_HOT_LDS;
	for (uint32_t j = threadIdx.x; j < CB_HOT_LDS; j += THREADS) { L.lk[j] = 0ull; L.lf[j] = 0xFFFFFFFFu; }
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < hot.n; j += THREADS) {
		const unsigned long long k = hot.key[j];
		uint32_t i = uint32_t(mix64(k) >> 40) & (CB_HOT_LDS - 1);
		for (;;) {
			const unsigned long long prev = atomicCAS(&L.lk[i], 0ull, k);
			if (prev == 0ull) { L.li[i] = j; break; }
			i = (i + 1) & (CB_HOT_LDS - 1);   // (hot keys are distinct: no equal key to find)
		}
	}
	__syncthreads();
	CbInsertAcc acc;
	for (uint32_t q = 0; q < rg.n; ++q) {
		uint32_t o = rg.off[q], cnt = rg.cnt[q];
		const uint32_t head = VEC ? (cnt < ((4u - (o & 3u)) & 3u) ? cnt : ((4u - (o & 3u)) & 3u)) : cnt;
		if (head) cb_insert_hot_range<false, STATS, EXP>(cb + o, umi + o, gene + o, aux + o, head, o, t, hot, L, slot_out + o, gene_chr, gene_chr_cap, pk, acc);
		o += head; cnt -= head;
		if (VEC && cnt) cb_insert_hot_range<true, STATS, EXP>(cb + o, umi + o, gene + o, aux + o, cnt, o, t, hot, L, slot_out + o, gene_chr, gene_chr_cap, pk, acc);
	}
	acc.commit(stats);
}

// ---- cell ids from the table alone ---------------------------------------------------------------------------
// The first-seen rank of a barcode is the rank of its first read ordinal among the first ordinals of all barcodes:
// compact the occupied slots into (first ordinal << 32 | slot) records, radix-sort them (a few million records instead
// of two more passes over all reads), and the position in the sorted list IS the cell id.
constexpr int CS_ITEMS = 16;   // slots per thread and iteration: one atomic per 4096 slots
__global__ __launch_bounds__(256) void cb_compact_slots_kernel(CbTable t, unsigned long long *__restrict__ out, uint32_t *__restrict__ count) {
	__shared__ uint32_t scratch[256 / 64 + 1];
	__shared__ uint32_t block_base;
	const uint64_t cap = t.mask + 1, chunk = uint64_t(256) * CS_ITEMS;
	for (uint64_t c0 = uint64_t(blockIdx.x) * chunk; c0 < cap; c0 += uint64_t(gridDim.x) * chunk) {
		uint32_t hits = 0, mine = 0;
		uint32_t first[CS_ITEMS];   // the whole 16-byte slot in one load: key and first ordinal together (the second pass read the line again)
#pragma unroll
		for (int j = 0; j < CS_ITEMS; ++j) {
			const uint64_t s = c0 + uint64_t(j) * 256 + threadIdx.x;
			first[j] = 0;
			if (s < cap) {
				const uint4 q = *reinterpret_cast<const uint4 *>(&t.slots[s]);
				if ((q.x | q.y) != 0u) { hits |= 1u << j; ++mine; first[j] = ~q.z; }
			}
		}
		uint32_t total;
		const uint32_t ex = block_excl_scan_u32<256>(mine, scratch, total);
		if (threadIdx.x == 0) block_base = total ? atomicAdd(count, total) : 0u;
		__syncthreads();
		uint32_t o = block_base + ex;
#pragma unroll
		for (int j = 0; j < CS_ITEMS; ++j)
			if (hits & (1u << j)) {
				const uint32_t s = uint32_t(c0 + uint64_t(j) * 256 + threadIdx.x);
				out[o++] = ((unsigned long long)first[j] << 32) | s;
			}
		__syncthreads();
	}
}
__global__ __launch_bounds__(256) void cb_assign_sorted_kernel(const unsigned long long *__restrict__ sorted, uint32_t n_cells, CbTable t,
                                                               unsigned long long *__restrict__ cell_cb, uint32_t *__restrict__ cell_first) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_cells) return;
	const uint32_t s = uint32_t(sorted[i]);
	t.slots[s].cell_id = i;
	cell_cb[i] = t.slots[s].key;
	cell_first[i] = uint32_t(sorted[i] >> 32);
}

// ---- StringIndexer _umi_indexer (CellsDataContainer.h:66, Gene.cpp:19): UMIs in the order of first appearance among the
// gene-bearing reads.  The same table protocol with the UMI code as the key; produced on demand (dropest_umi_first_seen).
__global__ __launch_bounds__(256) void umi_insert_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene, uint32_t n,
                                                         CbTable t, uint32_t *__restrict__ overflow) {
	bool ok = true;
	for (uint64_t r = uint64_t(blockIdx.x) * 256 + threadIdx.x; r < n; r += uint64_t(gridDim.x) * 256) {
		if (gene[r] == NO_GENE) continue;   // reads without a gene never reach Gene::add_umi (CellsDataContainer.cpp:73-78)
		const unsigned long long k = umi[r];
		const uint32_t s = cb_find_or_insert(t, k, mix64(k) & t.mask, ok);
		if (ok && uint32_t(r) < ~t.slots[s].nfirst) atomicMax(&t.slots[s].nfirst, ~uint32_t(r));
	}
	if (!ok) atomicMax(overflow, 1u);
}
__global__ __launch_bounds__(256) void gather_slot_keys_kernel(const unsigned long long *__restrict__ sorted, uint32_t n, CbTable t,
                                                               unsigned long long *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = t.slots[uint32_t(sorted[i])].key;
}

}  // namespace dropest
