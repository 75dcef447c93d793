// k_misc.h -- key construction, single-block scan, count-matrix emission, per-chromosome accumulation.
#pragma once

#include "k_cbhash.h"
#include "util.h"

namespace dropest {

// ---- exclusive scan of a short u32 array by ONE block (tile counts: N/2048 entries) ----
__global__ __launch_bounds__(1024) void scan_small_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                          uint32_t n, uint32_t *__restrict__ total_out) {
	__shared__ uint32_t scratch[1024 / 64 + 1];
	uint32_t carry = 0;
	for (uint32_t base = 0; base < n; base += 1024) {
		uint32_t i = base + threadIdx.x;
		uint32_t v = i < n ? in[i] : 0;
		uint32_t total;
		uint32_t ex = block_excl_scan_u32<1024>(v, scratch, total);
		if (i < n) out[i] = carry + ex;
		carry += total;
	}
	if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// ---- sort-key construction -------------------------------------------------------------------------
// molecule key = cell_id << (gene_bits + umi_bits) | gene_code << umi_bits | umi_code.
// gene_code of a read without a gene = all ones in gene_bits (sorts last inside its cell).
//
// Three sort-record layouts, chosen from the data (dropest_ctx::plan_key_layout):
//   VB = 0  key = molecule key << 3 | mark            keys only; needs 3 spare bits; chromosome derived from the gene
//   VB = 1  key = molecule key, value = mark (1 byte)  chromosome derived from the gene
//   VB = 4  key = molecule key, value = chr | mark<<16 general case (a gene seen on several chromosomes)
// With the chromosome derived from the gene (VB 0 / 1), a gene-less read stores its chromosome in the UMI field (it
// never enters a Gene, CellsDataContainer.cpp:73-78), so the per-(cell, chromosome) intergenic counts fall out of
// the molecule reduce; with VB = 4 gene-less reads carry umi_code 0.
struct KeyLayout {
	int umi_bits, gene_bits, cell_bits;
	int mark_shift;                      // 3 when the mark lives in the low key bits (VB = 0), else 0
	int val_bytes;                       // 0, 1 or 4
	unsigned long long umi_strip_mask;   // applied to clean UMI codes (drops the sentinel when all lengths agree)
	unsigned long long umi_escape_base;  // escaped UMI k -> umi_escape_base + k
	unsigned long long gene_none;        // (1 << gene_bits) - 1
};

struct GlobalCounters {   // CellsDataContainer.cpp:73-78, :309-327
	unsigned long long intergenic, exon, intron, not_annotated;
	unsigned long long key_or, key_and;
};

// STATS: the exact ingest statistics (k_cbhash.h: IngestAcc) ride along -- the layout L was planned from a sample of the reads and is
// checked against them afterwards (dropest_ctx::run_set_initialized); cb_insert then read the barcodes only.
// GCL (with STATS): the gene -> chromosome check reads a byte table in LDS (the first `lds_genes` genes, chromosomes below 255; 0xFF =
// "ask the table in memory") instead of one gather per read from gene_chr[] -- the gather was a quarter of the kernel at the C2 shape
// (scripts/probe/bk_probe.hip: 0.97 -> 0.73 ms without it).  Anything the byte table cannot answer takes the exact path below.
constexpr uint32_t BK_LDS_GENES_MAX = 49152;   // with the 16 KB of hot cell ids: 64 KB of LDS per workgroup at most
// PK: the reads are the packed 12-byte records of a sharded run (k_cbhash.h: ReadPack; umi -> w0, gene -> w1) -- a template parameter, not a
// run-time test: the plain kernel lost 0.12 ms per 1e8 reads with the test inside its loop.
template <int THREADS, int VB, bool VEC, bool HOT = false, bool STATS = false, bool GCL = false, bool PK = false>
__global__ __launch_bounds__(THREADS) void build_keys_kernel(const unsigned long long *__restrict__ umi,
                                                             const uint32_t *__restrict__ gene,
                                                             const uint32_t *__restrict__ aux,
                                                             const uint32_t *__restrict__ slot, uint32_t n, CbTable t,
                                                             KeyLayout L, unsigned long long *__restrict__ keys,
                                                             void *__restrict__ vals_, GlobalCounters *gc, CbHot hot = CbHot{nullptr, nullptr, 0},
                                                             uint32_t *__restrict__ gene_chr = nullptr, uint32_t gene_chr_cap = 0, IngestStats *stats = nullptr,
                                                             uint32_t lds_genes = 0, ReadPack pk = ReadPack{}) {
	static_assert(!GCL || STATS, "the LDS gene table serves the statistics only");
	// the cell ids of the hot barcodes (k_cbhash.h): 16 KB of LDS instead of one L2 request per read
	__shared__ uint32_t hot_cell[HOT ? CB_HOT_MAX : 1];
	extern __shared__ uint8_t bk_gene_chr8[];   // [lds_genes] (GCL)
	if (HOT)
		for (uint32_t j = threadIdx.x; j < hot.n; j += THREADS) hot_cell[j] = t.slots[hot.slot[j]].cell_id;
	if (GCL) {
		for (uint32_t g0 = threadIdx.x * 4; g0 < lds_genes; g0 += THREADS * 4) {   // (lds_genes is a multiple of 4, gene_chr_cap >= it)
			const uint4 c = *reinterpret_cast<const uint4 *>(gene_chr + g0);
			*reinterpret_cast<uint32_t *>(bk_gene_chr8 + g0) = (c.x < 255u ? c.x : 255u) | ((c.y < 255u ? c.y : 255u) << 8) | ((c.z < 255u ? c.z : 255u) << 16) |
			                                                   ((c.w < 255u ? c.w : 255u) << 24);
		}
	}
	if (HOT || GCL) __syncthreads();
	unsigned long long c_inter = 0, c_exon = 0, c_intron = 0, c_na = 0, k_or = 0, k_and = ~0ull;
	IngestAcc acc;
	// four CONSECUTIVE records per thread and iteration: 16-byte accesses per lane on every stream when the arrays are
	// 16-byte aligned (VEC), then the four dependent gathers of the cell ids
	constexpr int U = 4;
	const uint64_t stride = uint64_t(gridDim.x) * THREADS * U;
	for (uint64_t base = (uint64_t(blockIdx.x) * THREADS + threadIdx.x) * U; base < n; base += stride) {
		uint32_t sl[U], g[U], a[U];
		unsigned long long u[U], cell[U], kk[U];
		uint32_t vv[U] = {0, 0, 0, 0};
		const bool full = base + U <= n;
		if (PK && VEC && full) {   // packed records of a sharded run: 16 bytes of stream per read instead of 20 (umi -> w0, gene -> w1)
			const uint4 s4 = stream_load_u32x4(slot + base), w4 = stream_load_u32x4(gene + base);
			const ulonglong2 u01 = stream_load_u64x2(umi + base), u23 = stream_load_u64x2(umi + base + 2);
			sl[0] = s4.x; sl[1] = s4.y; sl[2] = s4.z; sl[3] = s4.w;
			g[0] = pk.gene(w4.x); g[1] = pk.gene(w4.y); g[2] = pk.gene(w4.z); g[3] = pk.gene(w4.w);
			a[0] = pk.aux(w4.x); a[1] = pk.aux(w4.y); a[2] = pk.aux(w4.z); a[3] = pk.aux(w4.w);
			u[0] = pk.umi(u01.x); u[1] = pk.umi(u01.y); u[2] = pk.umi(u23.x); u[3] = pk.umi(u23.y);
		} else if (VEC && full) {
			const uint4 s4 = stream_load_u32x4(slot + base), g4 = stream_load_u32x4(gene + base), a4 = stream_load_u32x4(aux + base);
			const ulonglong2 u01 = stream_load_u64x2(umi + base), u23 = stream_load_u64x2(umi + base + 2);
			sl[0] = s4.x; sl[1] = s4.y; sl[2] = s4.z; sl[3] = s4.w;
			g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
			a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
			u[0] = u01.x; u[1] = u01.y; u[2] = u23.x; u[3] = u23.y;
		} else {
#pragma unroll
			for (int q = 0; q < U; ++q) {
				const uint64_t r = base + q;
				sl[q] = 0; g[q] = NO_GENE; a[q] = 0; u[q] = 0;
				if (PK && r < n) { const uint32_t w1 = gene[r]; sl[q] = slot[r]; g[q] = pk.gene(w1); a[q] = pk.aux(w1); u[q] = pk.umi(umi[r]); }
				else if (r < n) { sl[q] = slot[r]; g[q] = gene[r]; a[q] = aux[r]; u[q] = umi[r]; }
			}
		}
		uint32_t gc[U];   // STATS: the gene -> chromosome entries, in flight together with the cell-id look-ups
#pragma unroll
		for (int q = 0; q < U; ++q) {
			if (base + q >= n) cell[q] = 0u;
			else if (HOT && (sl[q] & CB_HOT_FLAG)) cell[q] = hot_cell[sl[q] & ~CB_HOT_FLAG];
			else cell[q] = t.slots[sl[q]].cell_id;
			gc[q] = 0u;
			if (STATS && !GCL && base + q < n && g[q] != NO_GENE && ((a[q] >> 16) & 6u) && g[q] < gene_chr_cap) gc[q] = gene_chr[g[q]];
		}
#pragma unroll
		for (int q = 0; q < U; ++q) {
			const uint64_t r = base + q;
			kk[q] = 0;
			if (r >= n) continue;
			if (STATS) {
				acc.add(u[q], g[q], a[q]);
				if (!GCL) acc.check_chromosome(g[q], a[q], gene_chr, gene_chr_cap, gc[q]);
				else if (g[q] != NO_GENE && ((a[q] >> 16) & 6u)) {
					const uint32_t c8 = g[q] < lds_genes ? bk_gene_chr8[g[q]] : 255u;
					if (c8 != 255u) acc.chr_conflict |= c8 != (a[q] & 0xFFFFu);
					else {   // a gene the sample did not see, a chromosome id past 254, a gene past the byte table: the exact protocol
						acc.check_chromosome(g[q], a[q], gene_chr, gene_chr_cap, g[q] < gene_chr_cap ? gene_chr[g[q]] : 0u);
						if (g[q] < lds_genes) { const uint32_t now = gene_chr[g[q]]; if (now < 255u) bk_gene_chr8[g[q]] = uint8_t(now); }
					}
				}
			}
			uint32_t mark = (a[q] >> 16) & 0xFFu;
			unsigned long long gcode, ucode;
			if (g[q] == NO_GENE) {
				gcode = L.gene_none; ++c_inter;
				ucode = VB == 4 ? 0ull : (unsigned long long)(a[q] & 0xFFFFu);   // chromosome of the gene-less read
				if (VB != 4) mark = 0;
			} else {
				gcode = g[q];
				ucode = (u[q] & ESCAPE_BIT) ? (L.umi_escape_base + (u[q] & ~ESCAPE_BIT)) : (u[q] & L.umi_strip_mask);
				c_exon += (mark >> 1) & 1u; c_intron += (mark >> 2) & 1u; c_na += mark & 1u;
			}
			unsigned long long k = (cell[q] << (L.gene_bits + L.umi_bits)) | (gcode << L.umi_bits) | ucode;
			if (VB == 0) k = (k << 3) | (mark & 7u);
			kk[q] = k;
			vv[q] = VB == 1 ? mark & 0xFFu : a[q] & 0x00FFFFFFu;
			if (!(VEC && full)) {
				if (VB == 1) static_cast<uint8_t *>(vals_)[r] = uint8_t(mark);
				if (VB == 4) static_cast<uint32_t *>(vals_)[r] = a[q] & 0x00FFFFFFu;
			}
			k_or |= k; k_and &= k;
		}
		if (VEC && full) {
			if (VB == 1) stream_store_u32(reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(vals_) + base), vv[0] | (vv[1] << 8) | (vv[2] << 16) | (vv[3] << 24));
			if (VB == 4) stream_store_u32x4(static_cast<uint32_t *>(vals_) + base, vv[0], vv[1], vv[2], vv[3]);
			stream_store_u64x2(keys + base, kk[0], kk[1]);
			stream_store_u64x2(keys + base + 2, kk[2], kk[3]);
		} else {
#pragma unroll
			for (int q = 0; q < U; ++q) if (base + q < n) keys[base + q] = kk[q];
		}
	}
	// The twelve shared words (six counters of the pass, six ingest statistics) sit in two cache lines: an atomic per word and WAVE was
	// 49 000 atomics on those two lines at the tail of the kernel.  The waves of a workgroup meet in LDS first.
	constexpr int WAVES = THREADS / 64;
	__shared__ unsigned long long red[WAVES][12];
	c_inter = wave_reduce_add_u64(c_inter); c_exon = wave_reduce_add_u64(c_exon);
	c_intron = wave_reduce_add_u64(c_intron); c_na = wave_reduce_add_u64(c_na);
	k_or = wave_reduce_or_u64(k_or); k_and = wave_reduce_and_u64(k_and);
	unsigned long long s_min = ~0ull, s_max = 0, s_esc = 0, s_g = 0, s_c = 0, s_conf = 0;
	if (STATS) {
		s_min = wave_reduce_min_u64(acc.umin); s_max = wave_reduce_max_u64(acc.umax); s_esc = wave_reduce_max_u64(acc.uesc);
		s_g = wave_reduce_max_u64(acc.gmax); s_c = wave_reduce_max_u64(acc.cmax); s_conf = wave_reduce_max_u64(acc.chr_conflict ? 1ull : 0ull);
	}
	const uint32_t wv = threadIdx.x >> 6;
	if (lane_id() == 0) {
		red[wv][0] = c_inter; red[wv][1] = c_exon; red[wv][2] = c_intron; red[wv][3] = c_na; red[wv][4] = k_or; red[wv][5] = k_and;
		red[wv][6] = s_min; red[wv][7] = s_max; red[wv][8] = s_esc; red[wv][9] = s_g; red[wv][10] = s_c; red[wv][11] = s_conf;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w2 = 1; w2 < WAVES; ++w2) {
			for (int k = 0; k < 4; ++k) red[0][k] += red[w2][k];
			red[0][4] |= red[w2][4]; red[0][5] &= red[w2][5];
			red[0][6] = red[w2][6] < red[0][6] ? red[w2][6] : red[0][6];
			for (int k = 7; k < 12; ++k) red[0][k] = red[w2][k] > red[0][k] ? red[w2][k] : red[0][k];
		}
		if (red[0][0]) atomicAdd(&gc->intergenic, red[0][0]);
		if (red[0][1]) atomicAdd(&gc->exon, red[0][1]);
		if (red[0][2]) atomicAdd(&gc->intron, red[0][2]);
		if (red[0][3]) atomicAdd(&gc->not_annotated, red[0][3]);
		atomicOr(&gc->key_or, red[0][4]);
		atomicAnd(&gc->key_and, red[0][5]);
		if (STATS) {
			if (red[0][6] != ~0ull) atomicMin(&stats->umi_clean_min, red[0][6]);
			if (red[0][7] != 0ull) atomicMax(&stats->umi_clean_max, red[0][7]);
			if (red[0][8]) atomicMax(&stats->umi_escape_max_plus1, red[0][8]);
			if (red[0][9]) atomicMax(&stats->gene_max_plus1, uint32_t(red[0][9]));
			if (red[0][10]) atomicMax(&stats->chr_max_plus1, uint32_t(red[0][10]));
			if (red[0][11]) atomicMax(&stats->gene_chr_conflict, 1u);
		}
	}
}

// ---- real cells: Cell::is_real before any merge (Cell.cpp:125-128) -> id list in ascending order ---------
// ordered stream compaction: per-tile counts, scan_small, ordered write (the host needs the ids ascending)
constexpr int RC_THREADS = 256, RC_ITEMS = 8, RC_TILE = RC_THREADS * RC_ITEMS;
__global__ __launch_bounds__(RC_THREADS) void count_real_kernel(const uint32_t *__restrict__ n_genes, uint32_t n_cells,
                                                                uint32_t min_genes, uint32_t *__restrict__ tile_counts) {
	__shared__ uint32_t scratch[RC_THREADS / 64 + 1];
	const uint32_t i0 = blockIdx.x * RC_TILE + threadIdx.x * RC_ITEMS;
	uint32_t c = 0;
#pragma unroll
	for (int j = 0; j < RC_ITEMS; ++j) c += (i0 + j < n_cells && n_genes[i0 + j] >= min_genes);
	uint32_t total;
	block_excl_scan_u32<RC_THREADS>(c, scratch, total);
	if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(RC_THREADS) void write_real_kernel(const uint32_t *__restrict__ n_genes, uint32_t n_cells,
                                                                uint32_t min_genes, const uint32_t *__restrict__ tile_prefix,
                                                                uint32_t *__restrict__ real_list) {
	__shared__ uint32_t scratch[RC_THREADS / 64 + 1];
	const uint32_t i0 = blockIdx.x * RC_TILE + threadIdx.x * RC_ITEMS;
	uint32_t flags = 0, c = 0;
#pragma unroll
	for (int j = 0; j < RC_ITEMS; ++j)
		if (i0 + j < n_cells && n_genes[i0 + j] >= min_genes) { flags |= 1u << j; ++c; }
	uint32_t total;
	uint32_t o = tile_prefix[blockIdx.x] + block_excl_scan_u32<RC_THREADS>(c, scratch, total);
#pragma unroll
	for (int j = 0; j < RC_ITEMS; ++j)
		if (flags & (1u << j)) real_list[o++] = i0 + j;
}

// gathers the per-cell header + size rows of a list of cells (device -> staging for one D2H copy)
struct CellArrays {
	const unsigned long long *cb;
	const uint32_t *first, *n_genes, *req_genes, *req_umis, *total_umis, *total_reads;
};
struct CellRowPod {   // == dropest_cell_row (include/dropest_amd.h)
	unsigned long long barcode;
	uint32_t first_read, n_genes, requested_genes, requested_umis;
	int32_t total_reads, total_umis;
	uint8_t is_merged, is_excluded, is_real, pad;
};
__global__ __launch_bounds__(256) void gather_cell_rows_kernel(CellArrays a, const uint32_t *__restrict__ ids,
                                                               uint32_t first_id, uint32_t count,
                                                               CellRowPod *__restrict__ out) {
	uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j >= count) return;
	uint32_t i = ids ? ids[j] : first_id + j;
	CellRowPod r;
	r.barcode = a.cb[i]; r.first_read = a.first[i]; r.n_genes = a.n_genes[i]; r.requested_genes = a.req_genes[i];
	r.requested_umis = a.req_umis[i]; r.total_reads = int32_t(a.total_reads[i]); r.total_umis = int32_t(a.total_umis[i]);
	r.is_merged = r.is_excluded = r.is_real = r.pad = 0;
	out[j] = r;
}

// the three sizes a merge changes (n_genes, requested_genes, requested_umis) of the listed cells: 12 bytes per cell instead of a whole row
__global__ __launch_bounds__(256) void gather_cell_sizes_kernel(CellArrays a, const uint32_t *__restrict__ ids, uint32_t count, uint32_t *__restrict__ out) {
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j >= count) return;
	const uint32_t i = ids[j];
	out[3 * size_t(j)] = a.n_genes[i]; out[3 * size_t(j) + 1] = a.req_genes[i]; out[3 * size_t(j) + 2] = a.req_umis[i];
}

// ---- count matrix (ResultsPrinter::get_count_matrix_filtered / _raw, ResultsPrinter.cpp:334-396) -------
// One block per matrix column.  The column's cell owns the contiguous (cell, gene) rows
// [cg_begin[cell], cg_begin[cell+1]); rows are already gene-ascending.  `col_start` is the exclusive
// prefix of the per-column non-zero counts (requested_genes resp. n_genes), computed by the caller.
struct MatrixArgs {
	const uint32_t *col_cell;       // [ncols] cell id of each column
	const uint32_t *col_start;      // [ncols] first triplet of the column
	const uint32_t *cell_cg_begin;  // [n_cells] first (cell, gene) row of the cell
	const uint32_t *cell_cg_count;  // [n_cells] number of rows
	const unsigned long long *cg_key;
	const uint32_t *value;          // per (cell, gene) row: n_req | reads_req | n_all | reads_all
	unsigned long long gene_mask;
	int skip_zero;                  // filtered matrix omits zero entries (Cell.cpp:59-62)
	uint32_t *t_gene, *t_val;
	// NARROW output (gene ids below 65536): 16-bit row indices and values, 4 bytes per entry over PCIe instead of 8.  A value
	// beyond 65534 is stored as 0xFFFF and listed exactly in (ovf_pos, ovf_val) -- at most ovf_cap entries; ovf_count keeps counting.
	uint16_t *t_gene16, *t_val16;
	uint32_t *ovf_count, *ovf_pos, *ovf_val;
	uint32_t ovf_cap;
	// BYTE output (FORM 2; any gene id): per entry one byte of row DELTA (row - previous row of the column, the first entry against -1;
	// 255 = "listed": the exact ROW stands in (rovf_pos, rovf_row)) and one byte of value (255 = listed in (ovf_pos, ovf_val)): 2 bytes per
	// entry over PCIe.  Rows of a column ascend, a cell with a few thousand of 30 000 genes has gaps of ~10 and counts of a few UMIs:
	// well under 1 % of the entries are listed.
	uint8_t *t_drow8, *t_val8;
	uint32_t *rovf_count, *rovf_pos, *rovf_row;
	uint32_t rovf_cap;              // capacity of the row list (the value list: ovf_cap)
	const uint32_t *col_list;       // non-null: the launch covers these columns only (n_list of them)
	uint32_t n_list;
};
// appends (pos, val) of the lanes with `listed` to a list: ONE atomic per wave (all lists of a launch share one counter word: with an atomic per
// entry the 4e5 listed rows of cm_raw at C2 were most of the kernel's time)
__device__ inline void matrix_list_append(bool listed, uint32_t pos, uint32_t val, uint32_t *count, uint32_t *list_pos, uint32_t *list_val, uint32_t cap) {
	const unsigned long long m = __ballot(listed);
	if (!m) return;
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t base = 0;
	if (lane == uint32_t(__builtin_ctzll(m))) base = atomicAdd(count, uint32_t(__popcll(m)));
	base = uint32_t(__shfl(int(base), __builtin_ctzll(m), 64));
	if (listed) {
		const uint32_t at = base + uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
		if (at < cap) { list_pos[at] = pos; list_val[at] = val; }
	}
}
template <int FORM>   // 0: 32-bit, 1: 16-bit (NARROW), 2: bytes
__global__ __launch_bounds__(256) void emit_matrix_kernel(MatrixArgs a) {
	constexpr bool NARROW = FORM == 1;
	__shared__ uint32_t scratch[256 / 64 + 1];
	__shared__ uint32_t kept_row[FORM == 2 ? 257 : 1];   // [0] last kept row of the previous round (+1), [1 + ex] the rows kept in this one
	__shared__ uint8_t stage_d[FORM == 2 ? 256 : 1], stage_v[FORM == 2 ? 256 : 1];
	const uint32_t col = a.col_list ? a.col_list[blockIdx.x] : blockIdx.x;
	const uint32_t cell = a.col_cell[col];
	const uint32_t b = a.cell_cg_begin[cell], e = b + a.cell_cg_count[cell];
	uint32_t out = a.col_start[col];
	if (FORM == 2) { if (threadIdx.x == 0) kept_row[0] = 0u; }   // (row + 1 of "no entry yet": the first delta is row - (-1))
	for (uint32_t base = b; base < e; base += 256) {
		const uint32_t i = base + threadIdx.x;
		bool keep = false;
		uint32_t g = 0, v = 0;
		if (i < e) {
			g = uint32_t(a.cg_key[i] & a.gene_mask);
			v = a.value[i];
			keep = (a.cg_key[i] & a.gene_mask) != a.gene_mask && !(a.skip_zero && v == 0);
		}
		uint32_t total;
		const uint32_t ex = block_excl_scan_u32<256>(keep ? 1u : 0u, scratch, total);
		if (FORM == 2) {
			if (keep) kept_row[1 + ex] = g + 1u;
			__syncthreads();
			const uint32_t delta = keep ? g + 1u - kept_row[ex] : 0u;   // kept_row[ex]: the entry before this one (+1), kept_row[0] from the round before
			if (keep) {
				stage_d[ex] = delta >= 255u ? uint8_t(255u) : uint8_t(delta);
				stage_v[ex] = v >= 255u ? uint8_t(255u) : uint8_t(v);
			}
			matrix_list_append(keep && delta >= 255u, out + ex, g, a.rovf_count, a.rovf_pos, a.rovf_row, a.rovf_cap);
			matrix_list_append(keep && v >= 255u, out + ex, v, a.ovf_count, a.ovf_pos, a.ovf_val, a.ovf_cap);
			__syncthreads();
			{   // the round's bytes leave as aligned 4-byte words (one-byte stores of 256 threads cost 3x the 16-bit form's kernel time)
				const uint32_t head = min(total, (4u - (out & 3u)) & 3u), nw = (total - head) >> 2, tail0 = head + 4u * nw, t = threadIdx.x;
				if (t < head) { a.t_drow8[out + t] = stage_d[t]; a.t_val8[out + t] = stage_v[t]; }
				if (t < nw) {
					const uint32_t i = head + 4u * t;
					*reinterpret_cast<uint32_t *>(a.t_drow8 + out + i) = uint32_t(stage_d[i]) | (uint32_t(stage_d[i + 1]) << 8) | (uint32_t(stage_d[i + 2]) << 16) | (uint32_t(stage_d[i + 3]) << 24);
					*reinterpret_cast<uint32_t *>(a.t_val8 + out + i) = uint32_t(stage_v[i]) | (uint32_t(stage_v[i + 1]) << 8) | (uint32_t(stage_v[i + 2]) << 16) | (uint32_t(stage_v[i + 3]) << 24);
				}
				if (t < total - tail0) { a.t_drow8[out + tail0 + t] = stage_d[tail0 + t]; a.t_val8[out + tail0 + t] = stage_v[tail0 + t]; }
			}
			if (threadIdx.x == 0 && total) kept_row[0] = kept_row[total];
			// (the next round's block scan synchronises before anyone touches kept_row / the stages again)
		} else if (keep) {
			if (NARROW) {
				a.t_gene16[out + ex] = uint16_t(g);
				a.t_val16[out + ex] = v >= 0xFFFFu ? uint16_t(0xFFFFu) : uint16_t(v);
				if (v >= 0xFFFFu) {
					const uint32_t at = atomicAdd(a.ovf_count, 1u);
					if (at < a.ovf_cap) { a.ovf_pos[at] = out + ex; a.ovf_val[at] = v; }
				}
			} else { a.t_gene[out + ex] = g; a.t_val[out + ex] = v; }
		}
		out += total;
	}
}

// Byte form, short columns: one WAVE per column (eight columns per workgroup), no workgroup barrier while the columns are walked.  cm_raw
// of a 10x run has 5 000 cells of thousands of genes and 10^5 of a few dozen, whose sparse rows are mostly LISTED (gaps beyond 254):
// 4e5 list entries at C2.  Appending them with one atomic each -- or one per wave -- on the list's single counter word was the kernel's
// time (0.87 / 0.74 ms against 0.16 ms for the same matrix in the 16-bit form): the listed entries of a workgroup are staged in LDS and
// leave with ONE atomic.
constexpr uint32_t EMS_COLS = 8, EMS_STAGE = EMS_COLS * 255u;   // columns per workgroup; a short column lists at most 255 entries of a kind
__global__ __launch_bounds__(256) void emit_matrix_bytes_short_kernel(MatrixArgs a) {
	__shared__ uint32_t st_pos[2][EMS_STAGE], st_val[2][EMS_STAGE];   // [0] rows, [1] values
	__shared__ uint32_t st_n[2], st_base[2];
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	if (threadIdx.x < 2) st_n[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t c = w; c < EMS_COLS; c += 4) {
		const uint32_t idx = blockIdx.x * EMS_COLS + c;
		if (idx >= a.n_list) break;
		const uint32_t col = a.col_list[idx];
		const uint32_t cell = a.col_cell[col];
		const uint32_t b = a.cell_cg_begin[cell], e = b + a.cell_cg_count[cell];
		uint32_t out = a.col_start[col], prev1 = 0;   // prev1: row + 1 of the last entry kept so far (wave-uniform)
		for (uint32_t base = b; base < e; base += 64) {
			const uint32_t i = base + lane;
			bool keep = false;
			uint32_t g = 0, v = 0;
			if (i < e) {
				g = uint32_t(a.cg_key[i] & a.gene_mask);
				v = a.value[i];
				keep = (a.cg_key[i] & a.gene_mask) != a.gene_mask && !(a.skip_zero && v == 0);
			}
			const unsigned long long bal = __ballot(keep);
			if (!bal) continue;
			const unsigned long long below = bal & ((1ull << lane) - 1ull);
			const uint32_t ex = uint32_t(__popcll(below));
			const int src = below ? 63 - __builtin_clzll(below) : int(lane);   // the kept lane before this one
			const uint32_t before1 = uint32_t(__shfl(int(g + 1u), src, 64));
			const uint32_t delta = keep ? g + 1u - (below ? before1 : prev1) : 0u;
			if (keep) {
				a.t_drow8[out + ex] = delta >= 255u ? uint8_t(255u) : uint8_t(delta);
				a.t_val8[out + ex] = v >= 255u ? uint8_t(255u) : uint8_t(v);
				if (delta >= 255u) { const uint32_t at = atomicAdd(&st_n[0], 1u); st_pos[0][at] = out + ex; st_val[0][at] = g; }
				if (v >= 255u) { const uint32_t at = atomicAdd(&st_n[1], 1u); st_pos[1][at] = out + ex; st_val[1][at] = v; }
			}
			prev1 = uint32_t(__shfl(int(g + 1u), 63 - __builtin_clzll(bal), 64));
			out += uint32_t(__popcll(bal));
		}
	}
	__syncthreads();
	if (threadIdx.x < 2 && st_n[threadIdx.x]) st_base[threadIdx.x] = atomicAdd(threadIdx.x ? a.ovf_count : a.rovf_count, st_n[threadIdx.x]);
	__syncthreads();
	for (uint32_t k = 0; k < 2; ++k) {
		uint32_t *lp = k ? a.ovf_pos : a.rovf_pos, *lv = k ? a.ovf_val : a.rovf_row;
		const uint32_t cap = k ? a.ovf_cap : a.rovf_cap;
		for (uint32_t j = threadIdx.x; j < st_n[k]; j += 256) { const uint32_t at = st_base[k] + j; if (at < cap) { lp[at] = st_pos[k][j]; lv[at] = st_val[k][j]; } }
	}
}

// The two lists of a byte-form emit -> pinned host memory, with their exact lengths read on the device: no host round trip between the
// emit and the copies that follow it on the stream.  A list longer than its capacity sends its count only (the host then asks for a wider form).
// d / h: [0] count, [1 ..] positions, [1 + cap ..] rows / values.
__global__ __launch_bounds__(256) void matrix_lists_out_kernel(const uint32_t *__restrict__ d_r, uint32_t rcap, uint32_t *__restrict__ h_r,
                                                               const uint32_t *__restrict__ d_v, uint32_t vcap, uint32_t *__restrict__ h_v) {
	const uint32_t t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
	const uint32_t nr = d_r[0], nv = d_v[0];
	if (t == 0) { h_r[0] = nr; h_v[0] = nv; }
	if (nr <= rcap) for (uint32_t i = t; i < nr; i += stride) { h_r[1 + i] = d_r[1 + i]; h_r[1 + size_t(rcap) + i] = d_r[1 + size_t(rcap) + i]; }
	if (nv <= vcap) for (uint32_t i = t; i < nv; i += stride) { h_v[1 + i] = d_v[1 + i]; h_v[1 + size_t(vcap) + i] = d_v[1 + size_t(vcap) + i]; }
}

// The same for a shard's columns (sharded runs): the lists name entries of the shard's LOCAL matrix; on their way out every position becomes
// the entry's place in the GLOBAL matrix (desc[3c .. 3c + 2] = local start, global start, length of local column c, local starts
// ascending: a binary search per listed entry -- on the host it cost 18 cache misses for each of cm_raw's 4e5 listed rows).
__global__ __launch_bounds__(256) void matrix_lists_out_global_kernel(const uint32_t *__restrict__ d_r, uint32_t rcap, uint32_t *__restrict__ h_r,
                                                                      const uint32_t *__restrict__ d_v, uint32_t vcap, uint32_t *__restrict__ h_v,
                                                                      const unsigned long long *__restrict__ desc, uint32_t ncols) {
	const uint32_t t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
	const uint32_t nr = d_r[0], nv = d_v[0];
	if (t == 0) { h_r[0] = nr; h_v[0] = nv; }
	auto to_global = [&](uint32_t pos) {
		uint32_t lo = 0, hi = ncols;            // last column whose local start is <= pos
		while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (desc[3ull * mid] <= pos) lo = mid; else hi = mid; }
		return uint32_t(desc[3ull * lo + 1] + (pos - desc[3ull * lo]));
	};
	if (nr <= rcap) for (uint32_t i = t; i < nr; i += stride) { h_r[1 + i] = to_global(d_r[1 + i]); h_r[1 + size_t(rcap) + i] = d_r[1 + size_t(rcap) + i]; }
	if (nv <= vcap) for (uint32_t i = t; i < nv; i += stride) { h_v[1 + i] = to_global(d_v[1 + i]); h_v[1 + size_t(vcap) + i] = d_v[1 + size_t(vcap) + i]; }
}

// One chunk of a byte-form matrix (entries [k0, k1) of both byte arrays) from device memory to pinned host memory, by a kernel: a
// device-to-host hipMemcpyAsync costs ~20 us of copy-engine set-up whatever its size, and the chunked copy (matrix_decode.h) makes two
// dozen of them per matrix -- 0.5 ms on a 10 ms pass.  Source and destination are equally aligned (same index into arrays that both
// start on a page), so the body moves 16 bytes per lane; the ragged ends go byte by byte.
// flag / epoch: the arrival flag of what came BEFORE this kernel on the stream (the lists, the previous chunk) -- this kernel runs, so that is
// complete and visible to the host; the host's decoding threads watch the flags (matrix_decode.h) instead of asking the runtime about events.
__global__ __launch_bounds__(256) void matrix_chunk_to_host_kernel(const uint8_t *__restrict__ d_a, const uint8_t *__restrict__ d_b, uint8_t *__restrict__ h_a,
                                                                   uint8_t *__restrict__ h_b, size_t k0, size_t k1, uint32_t *flag, uint32_t epoch) {
	if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	const size_t a0 = (k0 + 15) & ~size_t(15), a1 = k1 & ~size_t(15);
	const size_t t = size_t(blockIdx.x) * 256 + threadIdx.x, stride = size_t(gridDim.x) * 256;
	const bool two = d_b != nullptr;   // (one array: the value bytes of a matrix that rides on another one's rows, emit_values_on_rows_kernel)
	if (a0 >= a1) {   // shorter than a line: bytes
		for (size_t k = k0 + t; k < k1; k += stride) { h_a[k] = d_a[k]; if (two) h_b[k] = d_b[k]; }
		return;
	}
	for (size_t k = k0 + t; k < a0; k += stride) { h_a[k] = d_a[k]; if (two) h_b[k] = d_b[k]; }
	for (size_t k = a1 + t; k < k1; k += stride) { h_a[k] = d_a[k]; if (two) h_b[k] = d_b[k]; }
	const uint4 *sa = reinterpret_cast<const uint4 *>(d_a + a0), *sb = reinterpret_cast<const uint4 *>(d_b + a0);
	uint4 *da = reinterpret_cast<uint4 *>(h_a + a0), *db = reinterpret_cast<uint4 *>(h_b + a0);
	const size_t n16 = (a1 - a0) >> 4;
	if (two) for (size_t i = t; i < n16; i += stride) { da[i] = sa[i]; db[i] = sb[i]; }
	else for (size_t i = t; i < n16; i += stride) da[i] = sa[i];
}

// cm as a rider on cm_raw (round 6): cm's columns are a subset of cm_raw's (filtered cells are real cells) and a column's
// entries are cm_raw's entries of that cell with values that are equal or smaller (requested UMIs against all UMIs of the same (cell, gene)
// row: Cell.cpp:54-68, ResultsPrinter.cpp:334-396), 0 = the entry is not in cm.  So cm needs no row bytes of its own: ONE byte per entry of
// cm_raw -- the value in cm, aligned with cm_raw's entries -- crosses the link, and the host reads the rows from cm_raw's delta bytes
// (csrc/matrix_decode.h: widen_derived) and leaves the zeros out.  col_out[c]: where column c of cm_raw starts in cm's slots, 0xFFFFFFFF = the
// cell is not a column of cm (its bytes are zeros).  A value beyond 254 is listed with its place in cm's SLOTS (the wave counts the entries it keeps).
// One wave per column of cm_raw; rows of a cell ascend by gene, the gene-less row (all ones) comes last and is no entry.
__global__ __launch_bounds__(256) void emit_values_on_rows_kernel(const uint32_t *__restrict__ col_cell, const uint32_t *__restrict__ col_start, uint32_t ncols,
                                                                  const uint32_t *__restrict__ cell_cg_begin, const uint32_t *__restrict__ cell_cg_count,
                                                                  const unsigned long long *__restrict__ cg_key, unsigned long long gene_mask,
                                                                  const uint32_t *__restrict__ value, const uint32_t *__restrict__ col_out,
                                                                  uint8_t *__restrict__ t_val8, uint32_t *ovf_count, uint32_t *ovf_pos, uint32_t *ovf_val, uint32_t ovf_cap) {
	const uint32_t lane = threadIdx.x & 63u;
	for (uint32_t col = blockIdx.x * 4u + (threadIdx.x >> 6); col < ncols; col += gridDim.x * 4u) {
		const uint32_t cell = col_cell[col];
		const uint32_t b = cell_cg_begin[cell], e = b + cell_cg_count[cell];
		const uint32_t first = col_start[col], oc = col_out[col];
		uint32_t kept = 0;
		for (uint32_t base = b; base < e; base += 64) {
			const uint32_t i = base + lane;
			const bool entry = i < e && (cg_key[i] & gene_mask) != gene_mask;
			const uint32_t v = entry && oc != 0xFFFFFFFFu ? value[i] : 0u;
			if (entry) t_val8[first + (i - b)] = v >= 255u ? uint8_t(255u) : uint8_t(v);
			const unsigned long long keep = __ballot(v != 0u);
			matrix_list_append(v >= 255u, oc + kept + uint32_t(__popcll(keep & ((1ull << lane) - 1ull))), v, ovf_count, ovf_pos, ovf_val, ovf_cap);
			kept += uint32_t(__popcll(keep));
		}
	}
}

__global__ void matrix_flag_kernel(uint32_t *flag, uint32_t epoch) { __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// requested UMIs / reads of every (cell, gene) row under ANOTHER mark query than the container's own
// (ResultsPrinter::save_intron_exon_matrices asks for "e", "i" and "BA", ResultsPrinter.cpp:455-474)
__global__ __launch_bounds__(256) void cg_requested_by_mask_kernel(const uint32_t *__restrict__ cg_mol_begin, uint32_t n_cg,
                                                                   const uint32_t *__restrict__ mol_mark, const uint32_t *__restrict__ mol_reads,
                                                                   uint32_t query_mask, int reads_output, uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_cg) return;
	uint32_t v = 0;
	for (uint32_t m = cg_mol_begin[i]; m < cg_mol_begin[i + 1]; ++m)
		if ((query_mask >> (mol_mark[m] & 7u)) & 1u) v += reads_output ? mol_reads[m] : 1u;
	out[i] = v;
}

// non-zero gene rows of each column's cell
__global__ __launch_bounds__(256) void count_nonzero_rows_kernel(const uint32_t *__restrict__ col_cell, uint32_t ncols,
                                                                 const uint32_t *__restrict__ cell_cg_begin,
                                                                 const uint32_t *__restrict__ cell_cg_count,
                                                                 const unsigned long long *__restrict__ cg_key, unsigned long long gene_mask,
                                                                 const uint32_t *__restrict__ value, uint32_t *__restrict__ count) {
	const uint32_t c = blockIdx.x * 256 + threadIdx.x;
	if (c >= ncols) return;
	const uint32_t cell = col_cell[c];
	uint32_t n = 0;
	for (uint32_t i = cell_cg_begin[cell], e = i + cell_cg_count[cell]; i < e; ++i)
		n += (cg_key[i] & gene_mask) != gene_mask && value[i] != 0;
	count[c] = n;
}

// ---- per-chromosome counters of real cells -------------------------------------------------------------
// partial rows (cell << 16 | chr ; exon, intron, intergenic) are folded into a dense
// [n_real][3][n_chr] table through the cell -> (merged) real-cell index map; rows are pre-aggregated per
// run of equal (cell, chr), so the atomics are few and mostly to distinct addresses.
__global__ __launch_bounds__(256) void chr_accumulate_kernel(const unsigned long long *__restrict__ row_key,
                                                             const uint32_t *__restrict__ exon,
                                                             const uint32_t *__restrict__ intron,
                                                             const uint32_t *__restrict__ intergenic, uint32_t n_rows,
                                                             const uint32_t *__restrict__ real_index /* [n_cells] */,
                                                             uint32_t n_chr, uint32_t *__restrict__ table) {
	uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_rows) return;
	const unsigned long long k = row_key[i];
	const uint32_t ri = real_index[uint32_t(k >> 16)];
	if (ri == 0xFFFFFFFFu) return;
	const uint32_t chr = uint32_t(k & 0xFFFFu);
	uint32_t *base = table + size_t(ri) * 3 * n_chr;
	if (exon[i]) atomicAdd(base + chr, exon[i]);
	if (intron[i]) atomicAdd(base + n_chr + chr, intron[i]);
	if (intergenic[i]) atomicAdd(base + 2 * n_chr + chr, intergenic[i]);
}

// ---- multi-GPU: stable partition of the reads by owner(cb) = mix64(cb) mod n_parts, in one pass over the five arrays ------
// owner_hist: per workgroup (a contiguous range of tiles) the number of reads of every owner; after the scans
// owner_scatter walks the same tiles carrying one output cursor per owner and writes every read (24 B in) with its
// position (28 B out) straight to its place: ranks inside a tile by the wave-ballot multisplit over the owner bits, combined
// over the waves through LDS -- the order (wave, item, lane) is the order of the positions, so reads of one owner keep stream
// order.  6 GB per 1e8 reads instead of the 10 GB of (position, owner) keys + radix pass + gather it replaces.
constexpr int OP_T = 512, OP_I = 8, OP_TILE = OP_T * OP_I;
// x mod n for a 64-bit x and a small n (<= 65536) without the 64-bit division (~100 instructions per read: the histogram and the scatter of
// the owner partition were bound by it, not by memory): x = hi 2^32 + lo, so x mod n = ((hi mod n)(2^32 mod n) + lo mod n) mod n, and a
// 32-bit a mod n = mulhi64(M a, n) with M = floor((2^64 - 1) / n) + 1 (Lemire, Kaser, Kurz: "Faster remainder by direct computation", 2019).
struct OwnerMod {
	unsigned long long M; uint32_t n, c;
	__host__ __device__ explicit OwnerMod(uint32_t n_) : M(~0ull / n_ + 1ull), n(n_), c(uint32_t((1ull << 32) % n_)) {}
	__device__ uint32_t mod32(uint32_t a) const { return uint32_t(__umul64hi(M * a, n)); }
	__device__ uint32_t operator()(unsigned long long x) const { return mod32(mod32(uint32_t(x >> 32)) * c + mod32(uint32_t(x))); }
};
__global__ __launch_bounds__(OP_T) void owner_hist_kernel(const unsigned long long *__restrict__ cb, uint32_t n, uint32_t n_parts,
                                                          uint32_t tiles_per_block, uint32_t *__restrict__ hist /* [256][gridDim.x] */) {
	__shared__ uint32_t h[256];
	if (threadIdx.x < 256) h[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t begin = uint64_t(blockIdx.x) * tiles_per_block * OP_TILE;
	uint64_t end = begin + uint64_t(tiles_per_block) * OP_TILE;
	if (end > n) end = n;
	const OwnerMod owner_of(n_parts);
	for (uint64_t i = begin + threadIdx.x; i < end; i += OP_T) atomicAdd(&h[owner_of(mix64(cb[i]))], 1u);
	__syncthreads();
	if (threadIdx.x < 256) hist[threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}
// The same histogram, and on the way the widths of the four fields over ALL resident reads (the sharded pass packs a read into
// 12 bytes for the exchange when they allow it): stats[0] largest barcode code, [1] largest UMI code, [2] 1 + largest gene id,
// [3] largest chromosome id, [4] OR of the aux bits above chromosome and 3-bit mark (must be zero).
// EVERY: the three other columns are read for every EVERY-th row of OP_T reads only (1 = all: exact statistics).  A sampled pass costs the
// barcodes' 8 bytes per read instead of 24; what the sample missed is caught by owner_scatter, which sees every field anyway (OwnerSelf::bad).
template <uint32_t EVERY>
__global__ __launch_bounds__(OP_T) void owner_hist_stats_kernel(const unsigned long long *__restrict__ cb, const unsigned long long *__restrict__ umi,
                                                                const uint32_t *__restrict__ gene, const uint32_t *__restrict__ aux, uint32_t n, uint32_t n_parts,
                                                                uint32_t tiles_per_block, uint32_t *__restrict__ hist /* [256][gridDim.x] */,
                                                                unsigned long long *__restrict__ stats) {
	__shared__ uint32_t h[256];
	if (threadIdx.x < 256) h[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t begin = uint64_t(blockIdx.x) * tiles_per_block * OP_TILE;
	uint64_t end = begin + uint64_t(tiles_per_block) * OP_TILE;
	if (end > n) end = n;
	unsigned long long cmax = 0, umax = 0, gmax = 0, chmax = 0, hi = 0;
	// up to eight owners (one node): counted in registers -- an LDS atomic per read on so few addresses serialises the lanes of a wave
	// (with ONE owner, the forced exchange of the bench, all 64 of them)
	uint32_t c8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	const bool few = n_parts <= 8;
	uint32_t row = 0;
	const OwnerMod owner_of(n_parts);
	for (uint64_t i = begin + threadIdx.x; i < end; i += OP_T, ++row) {
		const unsigned long long k = cb[i];
		const uint32_t own = owner_of(mix64(k));
		if (few) {
#pragma unroll
			for (uint32_t p = 0; p < 8; ++p) c8[p] += own == p;
		} else atomicAdd(&h[own], 1u);
		cmax = k > cmax ? k : cmax;
		if (EVERY > 1 && row % EVERY) continue;
		const unsigned long long u = umi[i];
		const uint32_t g = gene[i], a = aux[i];
		umax = u > umax ? u : umax;
		if (g != 0xFFFFFFFFu && (unsigned long long)g + 1 > gmax) gmax = (unsigned long long)g + 1;
		const unsigned long long ch = a & 0xFFFFu;
		chmax = ch > chmax ? ch : chmax;
		hi |= a >> 19;
	}
	if (few) {
#pragma unroll
		for (uint32_t p = 0; p < 8; ++p) {
			const unsigned long long tot = wave_reduce_add_u64(c8[p]);
			if (lane_id() == 0 && tot) atomicAdd(&h[p], uint32_t(tot));
		}
	}
	cmax = wave_reduce_max_u64(cmax); umax = wave_reduce_max_u64(umax); gmax = wave_reduce_max_u64(gmax); chmax = wave_reduce_max_u64(chmax);
	hi = wave_reduce_or_u64(hi);
	// the five statistics words: the waves of the workgroup meet in LDS, one set of atomics per workgroup
	__shared__ unsigned long long wst[OP_T / 64][5];
	if (lane_id() == 0) { const uint32_t wv = threadIdx.x >> 6; wst[wv][0] = cmax; wst[wv][1] = umax; wst[wv][2] = gmax; wst[wv][3] = chmax; wst[wv][4] = hi; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (uint32_t q = 1; q < OP_T / 64; ++q) {
			for (int z = 0; z < 4; ++z) wst[0][z] = wst[q][z] > wst[0][z] ? wst[q][z] : wst[0][z];
			wst[0][4] |= wst[q][4];
		}
		if (wst[0][0]) atomicMax(&stats[0], wst[0][0]);
		if (wst[0][1]) atomicMax(&stats[1], wst[0][1]);
		if (wst[0][2]) atomicMax(&stats[2], wst[0][2]);
		if (wst[0][3]) atomicMax(&stats[3], wst[0][3]);
		if (wst[0][4]) atomicOr(&stats[4], wst[0][4]);
	}
	if (threadIdx.x < 256) hist[threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// reads of every owner inside every chunk of blocks (bounds[0 .. chunks]: first block of each chunk): out[p * chunks + k], from the scanned
// per-block histogram (hist[p * nblocks + b] = reads of owner p before block b) and the owners' totals
__global__ __launch_bounds__(256) void owner_chunk_counts_kernel(const uint32_t *__restrict__ hist, const uint32_t *__restrict__ row_total, uint32_t nblocks,
                                                                 uint32_t n_parts, uint32_t chunks, const uint32_t *__restrict__ bounds, uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_parts * chunks) return;
	const uint32_t p = i / chunks, k = i % chunks;
	const uint32_t lo = bounds[k], hi = bounds[k + 1];
	const uint32_t before_lo = lo < nblocks ? hist[size_t(p) * nblocks + lo] : row_total[p], before_hi = hi < nblocks ? hist[size_t(p) * nblocks + hi] : row_total[p];
	out[i] = before_hi - before_lo;
}

// A read packed for the exchange: w0 = barcode code | UMI code << cb_bits (64 bits), w1 = gene | mark << gene_bits | chromosome
// << (gene_bits + 3) (32 bits; "no gene" = all ones of the gene field) -- 12 bytes where the five arrays are 28.
struct ExchangePack { int cb_bits, gene_bits; };
__device__ inline void exchange_pack(const ExchangePack &p, unsigned long long cb, unsigned long long umi, uint32_t gene, uint32_t aux, unsigned long long &w0, uint32_t &w1) {
	const uint32_t gmask = (1u << p.gene_bits) - 1u;
	w0 = cb | (umi << p.cb_bits);
	w1 = (gene == 0xFFFFFFFFu ? gmask : gene) | (((aux >> 16) & 7u) << p.gene_bits) | ((aux & 0xFFFFu) << (p.gene_bits + 3));
}
// records [own_lo, own_hi) are the block this shard kept: it never moved -- read where the partition left it (own_w0 / own_w1 point at
// the record that stands at own_lo)
__global__ __launch_bounds__(256) void exchange_unpack_kernel(const unsigned long long *__restrict__ w0, const uint32_t *__restrict__ w1, uint32_t n, ExchangePack p,
                                                              unsigned long long *__restrict__ cb, unsigned long long *__restrict__ umi,
                                                              uint32_t *__restrict__ gene, uint32_t *__restrict__ aux,
                                                              uint32_t own_lo, uint32_t own_hi, const unsigned long long *__restrict__ own_w0,
                                                              const uint32_t *__restrict__ own_w1) {
	const uint32_t gmask = (1u << p.gene_bits) - 1u;
	const unsigned long long cmask = p.cb_bits >= 64 ? ~0ull : ((1ull << p.cb_bits) - 1ull);
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
		const bool own = i >= own_lo && i < own_hi;
		const unsigned long long a = own ? own_w0[i - own_lo] : w0[i];
		const uint32_t b = own ? own_w1[i - own_lo] : w1[i];
		cb[i] = a & cmask;
		umi[i] = p.cb_bits >= 64 ? 0ull : a >> p.cb_bits;
		const uint32_t g = b & gmask;
		gene[i] = g == gmask ? 0xFFFFFFFFu : g;
		aux[i] = (b >> (p.gene_bits + 3)) | (((b >> p.gene_bits) & 7u) << 16);
	}
}

// PACKED: o_cb receives w0, o_gene receives w1 (o_umi / o_aux unused); the index array is written either way.
// self / self_w0 / self_w1 (PACKED): the records this shard keeps go straight to their place in the RECEIVE arrays (pointers shifted so
// that the partition's own index lands there): the receive arrays are then complete without a copy of the kept block.
// bad (PACKED): set when a read does not fit the packed record (the field widths came from a SAMPLE of the reads): the caller repeats the
// partition with exact widths.
struct OwnerSelf { uint32_t owner = 0xFFFFFFFFu; unsigned long long *w0 = nullptr; uint32_t *w1 = nullptr; uint32_t *bad = nullptr; };
template <bool PACKED>
__global__ __launch_bounds__(OP_T) void owner_scatter_kernel(const unsigned long long *__restrict__ cb, const unsigned long long *__restrict__ umi,
                                                             const uint32_t *__restrict__ gene, const uint32_t *__restrict__ aux, uint32_t n,
                                                             uint32_t n_parts, int owner_bits, uint32_t tiles_per_block,
                                                             const uint32_t *__restrict__ hist, const uint32_t *__restrict__ owner_base,
                                                             unsigned long long *__restrict__ o_cb, unsigned long long *__restrict__ o_umi,
                                                             uint32_t *__restrict__ o_gene, uint32_t *__restrict__ o_aux, uint32_t *__restrict__ o_idx,
                                                             ExchangePack pack, OwnerSelf self = OwnerSelf{}, uint32_t block0 = 0, uint32_t blocks_total = 0) {
	// block0 / blocks_total: the launch covers blocks [block0, block0 + gridDim.x) of a partition of blocks_total blocks (0: all of them) --
	// a sharded run sends the blocks of an owner in chunks as soon as they are written (csrc/shard_run.h)
	constexpr uint32_t WAVES = OP_T / 64;
	__shared__ uint32_t wcnt[WAVES][256], goff[256], tcnt[256];
	const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63u;
	const uint32_t blk = block0 + blockIdx.x, nblk = blocks_total ? blocks_total : gridDim.x;
	if (tid < 256) goff[tid] = owner_base[tid] + hist[tid * nblk + blk];
	const uint32_t n_tiles = (n + OP_TILE - 1) / OP_TILE, first_tile = blk * tiles_per_block;
	const uint32_t lane_off = w * (64 * OP_I) + lane;
	const OwnerMod owner_of(n_parts);
	for (uint32_t tt = 0; tt < tiles_per_block; ++tt) {
		const uint32_t tile = first_tile + tt;
		if (tile >= n_tiles) break;
		const uint32_t t0 = tile * OP_TILE;
		for (uint32_t j = tid; j < WAVES * 256; j += OP_T) (&wcnt[0][0])[j] = 0;
		lds_barrier();
		unsigned long long k[OP_I], u[OP_I];
		uint32_t g[OP_I], a[OP_I], own[OP_I], lrank[OP_I];
#pragma unroll
		for (int i = 0; i < OP_I; ++i) {
			const uint32_t r = t0 + lane_off + i * 64;
			const bool valid = r < n;
			k[i] = valid ? cb[r] : 0ull; u[i] = valid ? umi[r] : 0ull; g[i] = valid ? gene[r] : 0u; a[i] = valid ? aux[r] : 0u;
		}
#pragma unroll
		for (int i = 0; i < OP_I; ++i) {
			const bool valid = (t0 + lane_off + i * 64) < n;
			const uint32_t d = owner_of(mix64(k[i]));
			own[i] = d;
			uint32_t diff_lo = 0, diff_hi = 0;
			for (int b = 0; b < owner_bits; ++b) {
				const int32_t mine = int32_t(d << (31 - b)) >> 31;
				const unsigned long long bal = __ballot(mine != 0);
				diff_lo |= uint32_t(bal) ^ uint32_t(mine);
				diff_hi |= uint32_t(bal >> 32) ^ uint32_t(mine);
			}
			unsigned long long m = ~(((unsigned long long)diff_hi << 32) | diff_lo);
			m &= __ballot(valid);
			const uint32_t before = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
			const uint32_t old = wcnt[w][d];
			__builtin_amdgcn_wave_barrier();
			if (valid && before == 0) wcnt[w][d] = old + __popcll(m);
			__builtin_amdgcn_wave_barrier();
			lrank[i] = old + before;
		}
		lds_barrier();
		if (tid < 256) {
			uint32_t run = 0;
#pragma unroll
			for (uint32_t q = 0; q < WAVES; ++q) { const uint32_t c = wcnt[q][tid]; wcnt[q][tid] = run; run += c; }
			tcnt[tid] = run;
		}
		lds_barrier();
#pragma unroll
		for (int i = 0; i < OP_I; ++i) {
			const uint32_t r = t0 + lane_off + i * 64;
			if (r >= n) continue;
			const uint32_t dst = goff[own[i]] + wcnt[w][own[i]] + lrank[i];
			if (PACKED) {
				unsigned long long w0; uint32_t w1;
				exchange_pack(pack, k[i], u[i], g[i], a[i], w0, w1);
				if (self.bad) {
					const uint32_t gmask = (1u << pack.gene_bits) - 1u;
					const int chr_bits = 32 - 3 - pack.gene_bits;
					const bool fits = pack.cb_bits < 64 && (k[i] >> pack.cb_bits) == 0ull && (u[i] >> (64 - pack.cb_bits)) == 0ull &&
					                  (g[i] == 0xFFFFFFFFu || g[i] < gmask) && (a[i] >> 19) == 0u && (chr_bits >= 16 || ((a[i] & 0xFFFFu) >> chr_bits) == 0u);
					if (!fits) atomicOr(self.bad, 1u);
				}
				if (own[i] == self.owner) { self.w0[dst] = w0; self.w1[dst] = w1; }
				else { o_cb[dst] = w0; o_gene[dst] = w1; }
			} else { o_cb[dst] = k[i]; o_umi[dst] = u[i]; o_gene[dst] = g[i]; o_aux[dst] = a[i]; }
			o_idx[dst] = r;
		}
		lds_barrier();
		if (tid < 256) goff[tid] += tcnt[tid];
		// (the next iteration's first barrier orders this update before goff is read again)
	}
}

// ---- multi-GPU: owner keys, stable gather after the owner partition, column assembly ---------------------
// record = (position << 8) | owner: ONE keys-only radix pass on the low digit groups the reads by owner, stably
__global__ __launch_bounds__(256) void owner_keys_kernel(const unsigned long long *__restrict__ cb, uint32_t n, uint32_t n_parts,
                                                         unsigned long long *__restrict__ keys) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) keys[i] = ((unsigned long long)i << 8) | (mix64(cb[i]) % n_parts);
}
__global__ __launch_bounds__(256) void gather_reads_kernel(const unsigned long long *__restrict__ rec, uint32_t n,
                                                           const unsigned long long *__restrict__ cb, const unsigned long long *__restrict__ umi,
                                                           const uint32_t *__restrict__ gene, const uint32_t *__restrict__ aux,
                                                           unsigned long long *__restrict__ o_cb, unsigned long long *__restrict__ o_umi,
                                                           uint32_t *__restrict__ o_gene, uint32_t *__restrict__ o_aux,
                                                           uint32_t *__restrict__ o_idx) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		const uint32_t j = uint32_t(rec[i] >> 8);
		o_cb[i] = cb[j]; o_umi[i] = umi[j]; o_gene[i] = gene[j]; o_aux[i] = aux[j]; o_idx[i] = j;
	}
}
// desc[3c .. 3c+2] = (src_start, dst_start, len) of column c
__global__ __launch_bounds__(256) void assemble_columns_kernel(const unsigned long long *__restrict__ desc,
                                                               const uint32_t *__restrict__ src_rows, const uint32_t *__restrict__ src_vals,
                                                               uint32_t *__restrict__ dst_rows, uint32_t *__restrict__ dst_vals) {
	const unsigned long long s = desc[3ull * blockIdx.x], d = desc[3ull * blockIdx.x + 1], len = desc[3ull * blockIdx.x + 2];
	for (unsigned long long t = threadIdx.x; t < len; t += 256) { dst_rows[d + t] = src_rows[s + t]; dst_vals[d + t] = src_vals[s + t]; }
}

// keys_out[i] = table[idx[i]]: re-keys an index permutation for the next stable sort pass of a multi-key sort
__global__ __launch_bounds__(256) void gather_u64_kernel(const unsigned long long *__restrict__ table, const uint32_t *__restrict__ idx,
                                                         uint32_t n, unsigned long long *__restrict__ keys_out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) keys_out[i] = table[idx[i]];
}

// A small result written STRAIGHT into pinned host memory by the compute queue (the pointer is the host buffer's device view):
// for read-backs that must not queue behind a large device-to-host copy on the DMA engine (the permutation of sort_filtered
// while cm_raw's prefetch holds the copy engine: 15 ms of waiting for 10 MB at C3 size).
// Ordering the real-candidate cells entirely on the device (dropest_ctx::sort_filtered, pristine list): the three sort columns straight from
// the rows fetch_real_cells gathered -- barcode code, TOTAL_UMIS as Cell::umis_number casts it, (requested genes << 32 | requested UMIs) --
// with the OR / AND of every column (constant digits are skipped by the sorts) and the smallest / largest barcode code length.
// stats: [0..5] OR, AND of the three columns, [6] min bit length, [7] max bit length of the barcode codes (OR-ed with the escape bit's presence in [0])
__global__ __launch_bounds__(256) void sortf_columns_kernel(const CellRowPod *__restrict__ rows, uint32_t n, unsigned long long *__restrict__ code,
                                                            unsigned long long *__restrict__ umis, unsigned long long *__restrict__ sizes,
                                                            unsigned long long *__restrict__ stats) {
	unsigned long long o[3] = {0, 0, 0}, a[3] = {~0ull, ~0ull, ~0ull}, lmin = 64, lmax = 0;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
		const CellRowPod r = rows[i];
		const unsigned long long c = r.barcode, u = (unsigned long long)(size_t)(r.total_umis), z = ((unsigned long long)r.requested_genes << 32) | r.requested_umis;
		code[i] = c; umis[i] = u; sizes[i] = z;
		o[0] |= c; a[0] &= c; o[1] |= u; a[1] &= u; o[2] |= z; a[2] &= z;
		const unsigned long long bl = c ? 64 - __builtin_clzll(c) : 0;
		lmin = bl < lmin ? bl : lmin; lmax = bl > lmax ? bl : lmax;
	}
#pragma unroll
	for (int k = 0; k < 3; ++k) { o[k] = wave_reduce_or_u64(o[k]); a[k] = wave_reduce_and_u64(a[k]); }
	lmin = wave_reduce_min_u64(lmin); lmax = wave_reduce_max_u64(lmax);
	if (lane_id() == 0) {
#pragma unroll
		for (int k = 0; k < 3; ++k) { atomicOr(&stats[2 * k], o[k]); atomicAnd(&stats[2 * k + 1], a[k]); }
		atomicMin(&stats[6], lmin); atomicMax(&stats[7], lmax);
	}
}
// the ordered list to pinned host memory: out[0..m) = cell id, out[m..2m) = index in the real list, out[2m..3m) = TOTAL_UMIS of the cell at place i
__global__ __launch_bounds__(256) void sortf_out_kernel(const uint32_t *__restrict__ perm, const CellRowPod *__restrict__ rows, const uint32_t *__restrict__ ids, uint32_t m,
                                                        uint32_t *__restrict__ host_out) {
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
		const uint32_t at = perm[i];
		host_out[i] = ids[at]; host_out[size_t(m) + i] = at; host_out[2 * size_t(m) + i] = uint32_t(rows[at].total_umis);
	}
}
__global__ __launch_bounds__(256) void store_u32_to_host_kernel(const uint32_t *__restrict__ src, uint32_t n, uint32_t *__restrict__ host_dst) {
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) host_dst[i] = src[i];
}

__global__ __launch_bounds__(256) void load_u64_from_host_kernel(const unsigned long long *__restrict__ host_src, uint32_t n, unsigned long long *__restrict__ dst) {
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = host_src[i];
}

// Same table, chromosome derived from the gene: exon / intron read counts come from the (cell, gene) rows, the
// gene-less reads of a cell from its pseudo-molecules (cell, NONE, chromosome) whose read count is the answer.
struct ChrFromGeneArgs {
	const unsigned long long *cg_key; const uint32_t *cg_exon, *cg_intron, *cg_mol_begin; uint32_t n_cg;
	const unsigned long long *mol_key; const uint32_t *mol_reads;
	int gene_bits, umi_bits; unsigned long long gene_none;
	const uint32_t *gene_chr, *real_index; uint32_t n_chr; uint32_t *table;
};
__global__ __launch_bounds__(256) void chr_from_gene_kernel(ChrFromGeneArgs a) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= a.n_cg) return;
	const unsigned long long k = a.cg_key[i];
	const uint32_t ri = a.real_index[uint32_t(k >> a.gene_bits)];
	if (ri == 0xFFFFFFFFu) return;
	uint32_t *base = a.table + size_t(ri) * 3 * a.n_chr;
	const unsigned long long g = k & a.gene_none;
	if (g == a.gene_none) {
		const unsigned long long umask = (1ull << a.umi_bits) - 1ull;
		for (uint32_t m = a.cg_mol_begin[i]; m < a.cg_mol_begin[i + 1]; ++m)
			atomicAdd(base + 2 * a.n_chr + uint32_t(a.mol_key[m] & umask), a.mol_reads[m]);
		return;
	}
	const uint32_t e = a.cg_exon[i], n = a.cg_intron[i];
	if (!(e | n)) return;
	const uint32_t chr = a.gene_chr[uint32_t(g)];
	if (e) atomicAdd(base + chr, e);
	if (n) atomicAdd(base + a.n_chr + chr, n);
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *p, uint32_t v, size_t n) {
	size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
	if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void scatter_index_kernel(const uint32_t *__restrict__ ids, uint32_t n,
                                                            uint32_t *__restrict__ map) {
	uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) map[ids[i]] = i;
}

}  // namespace dropest
