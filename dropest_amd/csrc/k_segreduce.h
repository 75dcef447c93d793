// k_segreduce.h -- segmented reduce over key-sorted rows ("runs of equal segment key -> one output row").
//
// Three instantiations build the reference's nested containers bottom-up from the sorted read records:
//   reads      -> molecules      (Gene::add_umi / UMI::add_read: read_count++, mark |= ; Gene.cpp:17-24, UMI.cpp:21-34)
//   molecules  -> (cell, gene)   (Gene::number_of_requested_umis / number_of_umis; Gene.cpp:60-93)
//   (cell,gene)-> cells          (Cell::update_requested_size, Cell::size, Stats TOTAL_*; Cell.cpp:130-143)
// plus reads -> (cell, chromosome) partial rows for Stats' per-chromosome counters
// (CellsDataContainer::update_cell_stats, CellsDataContainer.cpp:309-327, :73-78).
//
// Scheme per instantiation (no inter-workgroup communication inside a launch):
//   seg_count   heads per tile (a head = row whose segment key differs from its predecessor's)
//   scan        exclusive scan of the tile counts (single block, tiles are few)
//   seg_zero_borders   clears the output rows of the runs that touch a tile border (the only rows added into with
//               atomics) and the sentinel row behind the table -- not a memset of every output channel
//   seg_reduce  persistent workgroups walk the tiles, the next tile's rows are fetched (striped, coalesced) into
//               registers while the current one is processed; each thread folds its 8 consecutive rows per run, run
//               aggregates of the tile are combined with LDS atomics indexed by the run's rank inside the tile (the
//               aggregates re-use the LDS of the staging area), then written coalesced; only the first/last run of a
//               tile may straddle a tile border: those two use global atomics, every interior run is a plain store.
//               The write-out loop covers the interior runs only and is branch-free (all LDS reads of a slot issued
//               together, then the stores); the two border runs are handled by two threads afterwards.
//               Measured on C2 (1e8 reads -> 4.2e7 molecules): reads -> molecules 1.00 -> 0.70 ms, molecules ->
//               (cell, gene) 0.90 -> 0.60 ms, and 0.35 ms of memsets gone.
// Works for any run length (one molecule with 10^6 reads, a cell with one chromosome, ...).
#pragma once

#include "util.h"

namespace dropest {

constexpr int SR_THREADS = 256;   // rows per thread = P::ITEMS, tile = SR_THREADS * P::ITEMS

// padded LDS index: 8 logical entries occupy 9 physical ones, so that a thread reading its consecutive rows
// (stride ITEMS entries across lanes) hits distinct banks
__device__ inline uint32_t sr_pad(uint32_t i) { return i + (i >> 3); }

template <class P>
__global__ __launch_bounds__(SR_THREADS) void seg_count_kernel(P p, uint32_t n, uint32_t *__restrict__ tile_heads) {
	constexpr int SR_ITEMS = P::ITEMS, SR_TILE = SR_THREADS * SR_ITEMS;
	__shared__ uint32_t scratch[SR_THREADS / 64 + 1];
	// striped (coalesced) accesses: a head is a row whose segment key differs from its predecessor's
	const uint32_t t0 = blockIdx.x * SR_TILE;
	uint32_t c = 0;
#pragma unroll
	for (int j = 0; j < SR_ITEMS; ++j) {
		const uint32_t i = t0 + j * SR_THREADS + threadIdx.x;
		if (i < n) c += (i == 0) || (p.seg_key(i) != p.seg_key(i - 1));
	}
	uint32_t total;
	block_excl_scan_u32<SR_THREADS>(c, scratch, total);
	if (threadIdx.x == 0) tile_heads[blockIdx.x] = total;
}

// The only output rows seg_reduce accumulates into with atomics are the runs that touch a tile border: the last run of
// every tile (row tile_prefix[t + 1] - 1; it may continue as the carry-in run of the following tiles).  Every other row
// is written with one plain store.  So instead of a memset of all output channels (0.7 GB for the molecules of C2)
// only those rows are cleared, plus the sentinel row `total` behind the table.
template <class P>
__global__ __launch_bounds__(256) void seg_zero_borders_kernel(P p, const uint32_t *__restrict__ tile_prefix, uint32_t tiles, uint32_t total) {
	const uint32_t t = blockIdx.x * 256 + threadIdx.x + 1;   // 1 .. tiles
	if (t > tiles) return;
	const uint32_t row = t < tiles ? tile_prefix[t] - 1 : total - 1;
#pragma unroll
	for (int c = 0; c < P::NV; ++c) p.out[c][row] = 0;
	if (t == tiles) {
#pragma unroll
		for (int c = 0; c < P::NV; ++c) p.out[c][total] = 0;
	}
}

// P must provide:
//   static constexpr int ITEMS;                      rows per thread (tile = 256 * ITEMS)
//   static constexpr int NV;                         number of u32 channels
//   static constexpr unsigned OR_MASK;               bit c set: channel c combines with OR, else with +
//   __device__ unsigned long long seg_key(uint32_t i) const;
//   __device__ void load(uint32_t i, uint32_t (&v)[NV]) const;      per-row contribution
//   __device__ void write_head(uint32_t out, uint32_t i, unsigned long long key) const;
//   uint32_t *out[NV];                               output channels, total + 1 rows each (DIRECT: zero-initialised by the
//                                                    caller; otherwise run_segmented_reduce clears what needs clearing)
//   static constexpr bool PACKED;                    true: the row's contribution is staged as ONE word (pack / unpack)
//                                                    instead of NV words (load) -- less LDS for wide reductions
//   static constexpr bool DIRECT;                    false: output row = rank of the run (needs seg_count + scan);
//                                                    true: output row = direct_index(segment key), no count pass
//   __device__ uint32_t direct_index(unsigned long long key) const;   (DIRECT only)
template <class P>
__global__ __launch_bounds__(SR_THREADS) void seg_reduce_kernel(P p, uint32_t n,
                                                                const uint32_t *__restrict__ tile_prefix) {
	constexpr int NV = P::NV;
	constexpr int SR_ITEMS = P::ITEMS, SR_TILE = SR_THREADS * SR_ITEMS;
	constexpr int PADDED = SR_TILE + SR_TILE / 8 + 2;
	constexpr int NSV = P::PACKED ? 1 : NV;               // staged words per row
	// One LDS buffer, used twice: first as the staging area of the coalesced loads (keys + per-row contributions), then --
	// once every thread holds its rows in registers -- as the per-run aggregates.  max() instead of the sum keeps the
	// kernel at 4 workgroups per CU instead of 2 (33-37 KB instead of 60 KB), i.e. twice the bytes in flight during the
	// load and the write-out phases, which is what bounds this kernel (measured: the load phase alone took 0.43 ms of the
	// 1.0 ms for 0.8 GB at two workgroups per CU).
	constexpr int KEY_WORDS = 2 * PADDED, VAL_WORDS = NSV * PADDED, AGG_STRIDE = SR_TILE + 1, AGG_WORDS = NV * AGG_STRIDE;
	constexpr int RAW_WORDS = (KEY_WORDS + VAL_WORDS) > AGG_WORDS ? (KEY_WORDS + VAL_WORDS) : AGG_WORDS;
	__shared__ uint32_t scratch[SR_THREADS / 64 + 1];
	__shared__ unsigned long long raw[(RAW_WORDS + 1) / 2];
	__shared__ uint32_t slot_row[P::DIRECT ? SR_TILE + 1 : 1];   // DIRECT: output row of each slot
	unsigned long long *skey = raw;                              // logical index 0 = predecessor of the tile, 1.. = rows
	uint32_t *sval = reinterpret_cast<uint32_t *>(raw) + KEY_WORDS;   // [NSV][PADDED]: per-row contributions
	uint32_t *agg = reinterpret_cast<uint32_t *>(raw);                // [NV][AGG_STRIDE]: slot 0 = run continuing from the previous tile

	// Persistent workgroups: each walks tiles blockIdx.x, + gridDim.x, ...; the NEXT tile's rows are fetched into registers
	// (striped, coalesced) before the current tile is processed, so the HBM latency hides behind the LDS phases.
	const uint32_t n_tiles = (n + SR_TILE - 1) / SR_TILE;
	unsigned long long pk[SR_ITEMS], ppred = 0;
	uint32_t pv[SR_ITEMS][NSV];
	auto fetch = [&](uint32_t tile) {
		const uint32_t t0 = tile * SR_TILE;
#pragma unroll
		for (int j = 0; j < SR_ITEMS; ++j) {
			const uint32_t i = t0 + j * SR_THREADS + threadIdx.x;
			if (i < n) {
				pk[j] = p.seg_key(i);
				if constexpr (P::PACKED) pv[j][0] = p.pack(i); else p.load(i, pv[j]);
			}
		}
		if (threadIdx.x == 0) ppred = t0 ? p.seg_key(t0 - 1) : ~p.seg_key(0);
	};
	if (blockIdx.x < n_tiles) fetch(blockIdx.x);

	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		// phase 1: the fetched rows go to LDS
		const uint32_t t0 = tile * SR_TILE;
#pragma unroll
		for (int j = 0; j < SR_ITEMS; ++j) {
			const uint32_t r = j * SR_THREADS + threadIdx.x, i = t0 + r;
			if (i < n) {
				skey[sr_pad(r + 1)] = pk[j];
#pragma unroll
				for (int c2 = 0; c2 < NSV; ++c2) sval[c2 * PADDED + sr_pad(r)] = pv[j][c2];
			}
		}
		if (threadIdx.x == 0) {
			skey[sr_pad(0)] = ppred;
			if (P::DIRECT) slot_row[0] = t0 ? p.direct_index(ppred) : 0u;
		}
		if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
		lds_barrier();

		// phase 2: each thread owns SR_ITEMS consecutive rows and takes them (keys and contributions) into registers
		const uint32_t r0 = threadIdx.x * SR_ITEMS, i0 = t0 + r0;
		unsigned long long key[SR_ITEMS];
		uint32_t rv[SR_ITEMS][NSV];
		uint32_t heads = 0, c = 0;
		if (i0 < n) {
			unsigned long long prev = skey[sr_pad(r0)];
#pragma unroll
			for (int j = 0; j < SR_ITEMS; ++j) {
				if (i0 + j < n) {
					key[j] = skey[sr_pad(r0 + j + 1)];
					if (key[j] != prev) { heads |= 1u << j; ++c; }
					prev = key[j];
#pragma unroll
					for (int c2 = 0; c2 < NSV; ++c2) rv[j][c2] = sval[c2 * PADDED + sr_pad(r0 + j)];
				}
			}
		}
		lds_barrier();   // every thread has its rows: the staging area is dead, the aggregates may overwrite it
		uint32_t total;
		const uint32_t ex = block_excl_scan_u32<SR_THREADS, true>(c, scratch, total);
		const uint32_t tp = P::DIRECT ? 1u : tile_prefix[tile];   // DIRECT: only "tp != 0" matters below
		// only the slots this tile uses (0 = carry-in run, 1..total = its heads) are cleared and, later, read back
		for (uint32_t s = threadIdx.x; s <= total; s += SR_THREADS) {
#pragma unroll
			for (int c2 = 0; c2 < NV; ++c2) agg[c2 * AGG_STRIDE + s] = 0;
		}
		lds_barrier();

		uint32_t slot = ex;   // rows before this thread's first head belong to the last head seen so far
		uint32_t acc[NV];
#pragma unroll
		for (int c2 = 0; c2 < NV; ++c2) acc[c2] = 0;
		bool dirty = false;
		auto flush = [&]() {
			if (!dirty) return;
#pragma unroll
			for (int c2 = 0; c2 < NV; ++c2) {
				if (acc[c2]) {
					if (P::OR_MASK & (1u << c2)) atomicOr(&agg[c2 * AGG_STRIDE + slot], acc[c2]);
					else atomicAdd(&agg[c2 * AGG_STRIDE + slot], acc[c2]);
				}
				acc[c2] = 0;
			}
			dirty = false;
		};
		if (i0 < n) {
#pragma unroll
			for (int j = 0; j < SR_ITEMS; ++j) {
				if (i0 + j < n) {
					if (heads & (1u << j)) {
						flush();
						++slot;
						if (P::DIRECT) {
							const uint32_t row = p.direct_index(key[j]);
							slot_row[slot] = row;
							p.write_head(row, i0 + j, key[j]);
						} else {
							p.write_head(tp + slot - 1, i0 + j, key[j]);
						}
					}
					if constexpr (P::PACKED) {
						uint32_t v[NV];
						p.unpack(rv[j][0], v);
#pragma unroll
						for (int c2 = 0; c2 < NV; ++c2) { if (P::OR_MASK & (1u << c2)) acc[c2] |= v[c2]; else acc[c2] += v[c2]; }
					} else {
#pragma unroll
						for (int c2 = 0; c2 < NV; ++c2) {
							const uint32_t v = rv[j][c2];
							if (P::OR_MASK & (1u << c2)) acc[c2] |= v; else acc[c2] += v;
						}
					}
					dirty = true;
				}
			}
			flush();
		}
		lds_barrier();

		// interior runs (slots 1 .. total - 1): one plain store per channel, all LDS reads of a slot issued together
		for (uint32_t s = threadIdx.x + 1; s < total; s += SR_THREADS) {
			uint32_t v[NV];
#pragma unroll
			for (int c2 = 0; c2 < NV; ++c2) v[c2] = agg[c2 * AGG_STRIDE + s];
			const uint32_t g = P::DIRECT ? slot_row[s] : tp + s - 1;
#pragma unroll
			for (int c2 = 0; c2 < NV; ++c2) p.out[c2][g] = v[c2];
		}
		// the two runs that may straddle the tile border (slot 0 = carry-in run, slot total = last run): global atomics.
		// Row 0 is always a head, so the first tile has no carry-in run.
		if (threadIdx.x < 2) {
			const uint32_t s = threadIdx.x == 0 ? 0u : total;
			const bool skip = (threadIdx.x == 1 && total == 0) || (s == 0 && (P::DIRECT ? tile == 0 : tp == 0));
			if (!skip) {
				const uint32_t g = P::DIRECT ? slot_row[s] : tp + s - 1;
#pragma unroll
				for (int c2 = 0; c2 < NV; ++c2) {
					const uint32_t v = agg[c2 * AGG_STRIDE + s];
					if (v) { if (P::OR_MASK & (1u << c2)) atomicOr(&p.out[c2][g], v); else atomicAdd(&p.out[c2][g], v); }
				}
			}
		}
		lds_barrier();   // the aggregates (and slot_row) are read: the next tile may stage over them
	}
}

// ---------------------------------------------------------------------------------------------------
// Policies
// ---------------------------------------------------------------------------------------------------

// sorted read records (key = cell|gene|umi, val = chr | mark<<16)  ->  molecules
struct ReadsToMolecules {
	static constexpr int ITEMS = 8;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 2;
	static constexpr unsigned OR_MASK = 0x2;   // ch0 read_count (+), ch1 mark (|)
	const unsigned long long *keys;
	const uint32_t *vals;
	unsigned long long *mol_key;
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const { return keys[i]; }
	__device__ void load(uint32_t i, uint32_t (&v)[NV]) const { v[0] = 1; v[1] = (vals[i] >> 16) & 0xFFu; }
	__device__ void write_head(uint32_t o, uint32_t, unsigned long long k) const { mol_key[o] = k; }
};

// sorted read records -> (cell, chromosome) partial rows with exon / intron / intergenic read counts
struct ReadsToChrRows {
	static constexpr int ITEMS = 8;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 3;
	static constexpr unsigned OR_MASK = 0;
	const unsigned long long *keys;
	const uint32_t *vals;
	int cell_shift;                  // gene_bits + umi_bits
	int umi_bits;
	unsigned long long gene_mask;    // (1 << gene_bits) - 1  == code of "no gene"
	unsigned long long *row_key;     // cell << 16 | chr
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const {
		return ((keys[i] >> cell_shift) << 16) | (vals[i] & 0xFFFFu);
	}
	__device__ void load(uint32_t i, uint32_t (&v)[NV]) const {
		const bool nogene = ((keys[i] >> umi_bits) & gene_mask) == gene_mask;
		const uint32_t mark = (vals[i] >> 16) & 0xFFu;
		v[0] = (!nogene && (mark & 2u)) ? 1u : 0u;
		v[1] = (!nogene && (mark & 4u)) ? 1u : 0u;
		v[2] = nogene ? 1u : 0u;
	}
	__device__ void write_head(uint32_t o, uint32_t, unsigned long long k) const { row_key[o] = k; }
};

// molecules -> (cell, gene) rows
struct MoleculesToCellGene {
	static constexpr int ITEMS = 4;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 4;   // n_all, n_req, reads_all, reads_req
	static constexpr unsigned OR_MASK = 0;
	const unsigned long long *mol_key;
	const uint32_t *mol_reads, *mol_mark;
	int umi_bits;
	uint32_t query_mask;             // bit m set: mark value m is requested (UMI::Mark::match, UMI.cpp:76-85)
	unsigned long long *cg_key;      // cell << gene_bits | gene
	uint32_t *cg_mol_begin;
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const { return mol_key[i] >> umi_bits; }
	__device__ void load(uint32_t i, uint32_t (&v)[NV]) const {
		const uint32_t r = mol_reads[i];
		const uint32_t req = (query_mask >> (mol_mark[i] & 7u)) & 1u;
		v[0] = 1; v[1] = req; v[2] = r; v[3] = req ? r : 0u;
	}
	__device__ void write_head(uint32_t o, uint32_t i, unsigned long long k) const { cg_key[o] = k; cg_mol_begin[o] = i; }
};

// (cell, gene) rows -> cells.  DIRECT: the output row is the cell id itself, so cells that lost all their rows
// to a merge simply stay zero.
struct CellGeneToCells {
	static constexpr int ITEMS = 4;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = true;
	static constexpr int NV = 6;   // n_genes, req_genes, req_umis, total_umis, total_reads, n_rows
	static constexpr unsigned OR_MASK = 0;
	const unsigned long long *cg_key;
	const uint32_t *n_all, *n_req, *reads_all;
	int gene_bits;
	unsigned long long gene_mask;
	uint32_t *cell_cg_begin;
	uint32_t *out[NV];
	__device__ uint32_t direct_index(unsigned long long key) const { return uint32_t(key); }
	__device__ unsigned long long seg_key(uint32_t i) const { return cg_key[i] >> gene_bits; }
	__device__ void load(uint32_t i, uint32_t (&v)[NV]) const {
		const bool nogene = (cg_key[i] & gene_mask) == gene_mask;
		const uint32_t rq = n_req[i];
		v[0] = nogene ? 0u : 1u;
		v[1] = (!nogene && rq) ? 1u : 0u;
		v[2] = nogene ? 0u : rq;
		v[3] = nogene ? 0u : n_all[i];
		v[4] = nogene ? 0u : reads_all[i];
		v[5] = 1u;
	}
	__device__ void write_head(uint32_t o, uint32_t i, unsigned long long) const { cell_cg_begin[o] = i; }
};

// re-keyed molecules (sorted by their new key; value = index of the molecule in the old table) -> molecules
struct RekeyedToMolecules {
	static constexpr int ITEMS = 8;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 2;
	static constexpr unsigned OR_MASK = 0x2;
	const unsigned long long *keys;
	const uint32_t *idx;
	const uint32_t *old_reads, *old_mark;
	unsigned long long *mol_key;
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const { return keys[i]; }
	__device__ void load(uint32_t i, uint32_t (&v)[NV]) const { const uint32_t j = idx[i]; v[0] = old_reads[j]; v[1] = old_mark[j]; }
	__device__ void write_head(uint32_t o, uint32_t, unsigned long long k) const { mol_key[o] = k; }
};

// sorted read records -> molecules, chromosome derived from the gene (sort layouts VB = 0 / 1): besides read count and
// mark the molecule keeps how many of its reads carry the exon / intron bit (UMI::Mark::HAS_EXONS / HAS_INTRONS), which
// is what Stats counts per chromosome (CellsDataContainer.cpp:312-321).  Only the mark byte is staged.
template <int VB>
struct ReadsToMoleculesX {
	static constexpr int ITEMS = 8;
	static constexpr bool PACKED = true;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 4;               // read_count (+), mark (|), exon reads (+), intron reads (+)
	static constexpr unsigned OR_MASK = 0x2;
	const unsigned long long *keys;
	const uint8_t *marks;                      // VB == 1
	unsigned long long *mol_key;
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const { return VB == 0 ? keys[i] >> 3 : keys[i]; }
	__device__ uint32_t pack(uint32_t i) const { return VB == 0 ? uint32_t(keys[i] & 7u) : uint32_t(marks[i]); }
	__device__ void unpack(uint32_t m, uint32_t (&v)[NV]) const { v[0] = 1; v[1] = m; v[2] = (m >> 1) & 1u; v[3] = (m >> 2) & 1u; }
	__device__ void load(uint32_t, uint32_t (&)[NV]) const {}
	__device__ void write_head(uint32_t o, uint32_t, unsigned long long k) const { mol_key[o] = k; }
};

// molecules -> (cell, gene) rows, with the exon / intron read counts
struct MoleculesToCellGeneX {
	static constexpr int ITEMS = 4;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 6;   // n_all, n_req, reads_all, reads_req, exon reads, intron reads
	static constexpr unsigned OR_MASK = 0;
	const unsigned long long *mol_key;
	const uint32_t *mol_reads, *mol_mark, *mol_exon, *mol_intron;
	int umi_bits;
	uint32_t query_mask;
	unsigned long long *cg_key;
	uint32_t *cg_mol_begin;
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const { return mol_key[i] >> umi_bits; }
	__device__ void load(uint32_t i, uint32_t (&v)[NV]) const {
		const uint32_t r = mol_reads[i];
		const uint32_t req = (query_mask >> (mol_mark[i] & 7u)) & 1u;
		v[0] = 1; v[1] = req; v[2] = r; v[3] = req ? r : 0u; v[4] = mol_exon[i]; v[5] = mol_intron[i];
	}
	__device__ void write_head(uint32_t o, uint32_t i, unsigned long long k) const { cg_key[o] = k; cg_mol_begin[o] = i; }
};

// re-keyed molecules -> molecules, carrying the exon / intron read counts through a CB merge
struct RekeyedToMoleculesX {
	static constexpr int ITEMS = 4;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 4;
	static constexpr unsigned OR_MASK = 0x2;
	const unsigned long long *keys;
	const uint32_t *idx;
	const uint32_t *old_reads, *old_mark, *old_exon, *old_intron;
	unsigned long long *mol_key;
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const { return keys[i]; }
	__device__ void load(uint32_t i, uint32_t (&v)[NV]) const {
		const uint32_t j = idx[i];
		v[0] = old_reads[j]; v[1] = old_mark[j]; v[2] = old_exon[j]; v[3] = old_intron[j];
	}
	__device__ void write_head(uint32_t o, uint32_t, unsigned long long k) const { mol_key[o] = k; }
};

}  // namespace dropest
