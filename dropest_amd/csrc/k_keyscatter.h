// k_keyscatter.h -- the sort keys built AND partitioned in one pass: build_keys (k_misc.h) fused with the first level of the splitter sort
// (k_ssort.h: ss_scatter_res_l1).  Hand-written for gfx950; integer / HBM-bound work, no MFMA.
//
// build_keys wrote 8 (+ VB) bytes per read that the first partition read back a moment later: 1.6 GB per 1e8 reads of pure hand-over
// ("three of these are passes that exist only to hand data to the next kernel").  The partitions place their records by
// RESERVATION (no histogram pass needs the keys first), so a tile can be regrouped where it is built -- once the coarse splitters exist.
// They used to come from a sample of the finished KEYS; here the sample is taken from the READS before the key pass (ss_sample_reads:
// clusters of 16 consecutive reads, whose keys are as independent as those of reads 24 apart -- the stream is not ordered by barcode --
// and cost one line per array instead of sixteen), sorted as before, and gives both levels their splitters.
//
// Replaces, like the two kernels it fuses: the key path of CellsDataContainer::add_record (CellsDataContainer.cpp:356-364), the read-type
// counters (:73-78, :309-327), and the first descent of the std::map inserts of Cell::genes() / Gene::_umis (Estimation/Cell.h:19, Gene.h:19).
#pragma once

#include "k_cbhash.h"
#include "k_misc.h"
#include "k_ssort.h"

namespace dropest {

// molecule key of one read (what build_keys computes): cell | gene | UMI; a read without a gene carries its chromosome in the UMI field
// when the chromosome is derived from the gene (VB != 4)
__device__ inline unsigned long long mol_key_of(const KeyLayout &L, unsigned long long cell, unsigned long long u, uint32_t g, uint32_t a) {
	unsigned long long gcode, ucode;
	if (g == NO_GENE) { gcode = L.gene_none; ucode = (unsigned long long)(a & 0xFFFFu); }
	else { gcode = g; ucode = (u & ESCAPE_BIT) ? (L.umi_escape_base + (u & ~ESCAPE_BIT)) : (u & L.umi_strip_mask); }
	return (cell << (L.gene_bits + L.umi_bits)) | (gcode << L.umi_bits) | ucode;
}

// n_sample molecule keys from the reads: cluster c = 16 consecutive reads starting at a multiple of 16 near (c + 1/2) n / n_clusters
template <bool PK>
__global__ __launch_bounds__(256) void ss_sample_reads_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene, const uint32_t *__restrict__ aux,
                                                              const uint32_t *__restrict__ slot, uint32_t n, CbTable t, KeyLayout L, CbHot hot, ReadPack pk,
                                                              uint32_t n_sample, unsigned long long *__restrict__ out) {
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n_sample) return;
	const uint32_t n_clusters = (n_sample + 15u) / 16u, c = j >> 4;
	uint64_t pos = ((uint64_t(c) * n + n / 2) / n_clusters) & ~uint64_t(15);
	pos += j & 15u;
	if (pos >= n) pos = n - 1;
	const uint32_t sl = slot[pos];
	const unsigned long long cell = (sl & CB_HOT_FLAG) && hot.n ? t.slots[hot.slot[sl & ~CB_HOT_FLAG]].cell_id : t.slots[sl].cell_id;
	unsigned long long u; uint32_t g, a;
	if (PK) { const uint32_t w1 = gene[pos]; u = pk.umi(umi[pos]); g = pk.gene(w1); a = pk.aux(w1); }
	else { u = umi[pos]; g = gene[pos]; a = g == NO_GENE ? aux[pos] : 0u; }
	out[j] = mol_key_of(L, cell, u, g, a);
}

constexpr int KS_T = 1024, KS_I = 4, KS_TILE = KS_T * KS_I;   // one workgroup per CU: 4 096 reads per tile, four CONSECUTIVE reads per thread
constexpr uint32_t KS_MAXF = 1024;

// STATS: 0 none, 1 exact ingest statistics with the gene -> chromosome check by gather, 2 the same from the LDS byte table (GCL of build_keys)
template <int VB, int STATS, bool PK>
__global__ __launch_bounds__(KS_T) void build_keys_scatter_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene,
                                                                  const uint32_t *__restrict__ aux, const uint32_t *__restrict__ slot, uint32_t n, CbTable t,
                                                                  KeyLayout L, GlobalCounters *gc, CbHot hot, uint32_t *__restrict__ gene_chr, uint32_t gene_chr_cap,
                                                                  IngestStats *stats, uint32_t lds_genes, ReadPack pk,
                                                                  unsigned long long *__restrict__ okeys, uint8_t *__restrict__ ovals, int ms, int fb,
                                                                  const unsigned long long *__restrict__ coarse, SsReserve rs) {
	static_assert(VB == 0 || VB == 1, "layouts whose chromosome is derived from the gene");
	__shared__ uint32_t hot_cell[CB_HOT_MAX];
	__shared__ unsigned long long sp[KS_MAXF];
	__shared__ uint32_t cnt[KS_MAXF], tstart[KS_MAXF], gdelta[KS_MAXF], scratch[KS_T / 64 + 1];
	// dynamic LDS (beyond the 64 KB a kernel may declare): the tile -- keys, bucket of each, mark bytes -- and behind it the gene -> chromosome
	// byte table [lds_genes] (STATS == 2); ks_dynamic_lds() says how much
	extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
	unsigned long long *sk = reinterpret_cast<unsigned long long *>(ks_smem);
	uint16_t *sd = reinterpret_cast<uint16_t *>(ks_smem + size_t(KS_TILE) * 8);
	uint8_t *sv = ks_smem + size_t(KS_TILE) * 10;
	uint8_t *ks_gene_chr8 = ks_smem + size_t(KS_TILE) * 11;
	const uint32_t F = 1u << fb, tid = threadIdx.x;
	for (uint32_t j = tid; j < hot.n; j += KS_T) hot_cell[j] = t.slots[hot.slot[j]].cell_id;
	for (uint32_t j = tid; j < F; j += KS_T) sp[j] = j + 1 < F ? coarse[j] : ~0ull;
	if (STATS == 2)
		for (uint32_t g0 = tid * 4; g0 < lds_genes; g0 += KS_T * 4) {
			const uint4 c = *reinterpret_cast<const uint4 *>(gene_chr + g0);
			*reinterpret_cast<uint32_t *>(ks_gene_chr8 + g0) = (c.x < 255u ? c.x : 255u) | ((c.y < 255u ? c.y : 255u) << 8) | ((c.z < 255u ? c.z : 255u) << 16) |
			                                                   ((c.w < 255u ? c.w : 255u) << 24);
		}
	__syncthreads();
	unsigned long long c_inter = 0, c_exon = 0, c_intron = 0, c_na = 0, k_or = 0, k_and = ~0ull;
	IngestAcc acc;
	const uint32_t n_tiles = (n + KS_TILE - 1) / KS_TILE;
	bool have_next = false;
	uint4 nx_s4 = make_uint4(0u, 0u, 0u, 0u), nx_g4 = nx_s4, nx_a4 = nx_s4;
	ulonglong2 nx_u01 = make_ulonglong2(0ull, 0ull), nx_u23 = nx_u01;
	unsigned long long nx_cell[KS_I] = {0, 0, 0, 0};
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t base = uint64_t(tile) * KS_TILE + uint64_t(tid) * KS_I;
		const uint32_t in_tile = n - tile * uint32_t(KS_TILE) < uint32_t(KS_TILE) ? n - tile * uint32_t(KS_TILE) : uint32_t(KS_TILE);
		// ---- the keys of this thread's four reads (build_keys) ----
		// (One workgroup per CU marches through its tiles in lockstep: nothing else hides a tile's memory latency.  So the NEXT tile's reads are
		// requested while this tile is counted and regrouped, and its cell ids -- the gathers of the reads off the hot list, the longest wait of
		// the key pass -- while this tile is written out.)
		uint32_t sl[KS_I], g[KS_I], a[KS_I];
		unsigned long long u[KS_I], key[KS_I];
		uint8_t val[KS_I];
		const bool full = base + KS_I <= n;
		uint4 s4, g4, a4 = make_uint4(0u, 0u, 0u, 0u);
		ulonglong2 u01, u23;
		if (have_next) { s4 = nx_s4; g4 = nx_g4; a4 = nx_a4; u01 = nx_u01; u23 = nx_u23; }
		else if (full) {
			s4 = stream_load_u32x4(slot + base); g4 = stream_load_u32x4(gene + base);
			u01 = stream_load_u64x2(umi + base); u23 = stream_load_u64x2(umi + base + 2);
			if (!PK) a4 = stream_load_u32x4(aux + base);
		}
		if (full) {
			sl[0] = s4.x; sl[1] = s4.y; sl[2] = s4.z; sl[3] = s4.w;
			if (PK) {
				g[0] = pk.gene(g4.x); g[1] = pk.gene(g4.y); g[2] = pk.gene(g4.z); g[3] = pk.gene(g4.w);
				a[0] = pk.aux(g4.x); a[1] = pk.aux(g4.y); a[2] = pk.aux(g4.z); a[3] = pk.aux(g4.w);
				u[0] = pk.umi(u01.x); u[1] = pk.umi(u01.y); u[2] = pk.umi(u23.x); u[3] = pk.umi(u23.y);
			} else {
				g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
				a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
				u[0] = u01.x; u[1] = u01.y; u[2] = u23.x; u[3] = u23.y;
			}
		} else {
#pragma unroll
			for (int q = 0; q < KS_I; ++q) {
				const uint64_t r = base + q;
				sl[q] = 0; g[q] = NO_GENE; a[q] = 0; u[q] = 0;
				if (PK && r < n) { const uint32_t w1 = gene[r]; sl[q] = slot[r]; g[q] = pk.gene(w1); a[q] = pk.aux(w1); u[q] = pk.umi(umi[r]); }
				else if (r < n) { sl[q] = slot[r]; g[q] = gene[r]; a[q] = aux[r]; u[q] = umi[r]; }
			}
		}
		unsigned long long cell[KS_I];
		uint32_t gcv[KS_I];
#pragma unroll
		for (int q = 0; q < KS_I; ++q) {
			if (base + q >= n) cell[q] = 0u;
			else if (have_next) cell[q] = nx_cell[q];
			else if (sl[q] & CB_HOT_FLAG) cell[q] = hot_cell[sl[q] & ~CB_HOT_FLAG];
			else cell[q] = t.slots[sl[q]].cell_id;
			gcv[q] = 0u;
			if (STATS == 1 && base + q < n && g[q] != NO_GENE && ((a[q] >> 16) & 6u) && g[q] < gene_chr_cap) gcv[q] = gene_chr[g[q]];
		}
#pragma unroll
		for (int q = 0; q < KS_I; ++q) {
			key[q] = 0; val[q] = 0;
			if (base + q >= n) continue;
			if (STATS) {
				acc.add(u[q], g[q], a[q]);
				if (STATS == 1) acc.check_chromosome(g[q], a[q], gene_chr, gene_chr_cap, gcv[q]);
				else if (g[q] != NO_GENE && ((a[q] >> 16) & 6u)) {
					const uint32_t c8 = g[q] < lds_genes ? ks_gene_chr8[g[q]] : 255u;
					if (c8 != 255u) acc.chr_conflict |= c8 != (a[q] & 0xFFFFu);
					else {
						acc.check_chromosome(g[q], a[q], gene_chr, gene_chr_cap, g[q] < gene_chr_cap ? gene_chr[g[q]] : 0u);
						if (g[q] < lds_genes) { const uint32_t now = gene_chr[g[q]]; if (now < 255u) ks_gene_chr8[g[q]] = uint8_t(now); }
					}
				}
			}
			uint32_t mark = (a[q] >> 16) & 0xFFu;
			if (g[q] == NO_GENE) { ++c_inter; mark = 0; }
			else { c_exon += (mark >> 1) & 1u; c_intron += (mark >> 2) & 1u; c_na += mark & 1u; }
			unsigned long long k = mol_key_of(L, cell[q], u[q], g[q], a[q]);
			if (VB == 0) k = (k << 3) | (mark & 7u);
			key[q] = k; val[q] = uint8_t(mark);
			k_or |= k; k_and &= k;
		}
		// ---- the tile into the coarse regions (ss_scatter_res_range, four records per thread) ----
		for (uint32_t j = tid; j < F; j += KS_T) cnt[j] = 0;
		lds_barrier();
		uint32_t pos[KS_I], rk[KS_I];
#pragma unroll
		for (int q = 0; q < KS_I; ++q) pos[q] = 0;
		for (int b = fb - 1; b >= 0; --b) {
			const uint32_t step = 1u << b;
#pragma unroll
			for (int q = 0; q < KS_I; ++q) if (sp[pos[q] + step - 1] <= (key[q] >> ms)) pos[q] += step;
		}
#pragma unroll
		for (int q = 0; q < KS_I; ++q) rk[q] = base + q < n ? atomicAdd(&cnt[pos[q]], 1u) : 0u;
		// the next tile of this workgroup, when it lies wholly inside the stream
		const uint32_t next_tile = tile + gridDim.x;
		const bool next_ok = next_tile < n_tiles && (uint64_t(next_tile) + 1) * KS_TILE <= n;
		const uint64_t nbase = uint64_t(next_tile) * KS_TILE + uint64_t(tid) * KS_I;
		if (next_ok) {
			nx_s4 = stream_load_u32x4(slot + nbase); nx_g4 = stream_load_u32x4(gene + nbase);
			nx_u01 = stream_load_u64x2(umi + nbase); nx_u23 = stream_load_u64x2(umi + nbase + 2);
			if (!PK) nx_a4 = stream_load_u32x4(aux + nbase);
		}
		lds_barrier();
		const uint32_t c = tid < F ? cnt[tid] : 0u;
		const uint32_t got = c ? atomicAdd(&rs.cursor[size_t(rs.first + tid) * rs.cstride], c) : 0u;   // the tile's places in the region: in flight while it is regrouped
		uint32_t total;
		const uint32_t ex = block_excl_scan_u32<KS_T, true>(c, scratch, total);
		if (tid < F) tstart[tid] = ex;
		lds_barrier();
#pragma unroll
		for (int q = 0; q < KS_I; ++q)
			if (base + q < n) {
				const uint32_t at = tstart[pos[q]] + rk[q];
				sk[at] = key[q]; sd[at] = uint16_t(pos[q]);
				if (VB) sv[at] = val[q];
			}
		if (tid < F) {
			gdelta[tid] = (rs.first + tid) * rs.cap + got - ex;
			if (got + c > rs.cap) atomicOr(rs.overflow, 1u);
		}
		lds_barrier();
		if (next_ok) {
			const uint32_t ns[KS_I] = {nx_s4.x, nx_s4.y, nx_s4.z, nx_s4.w};
#pragma unroll
			for (int q = 0; q < KS_I; ++q) nx_cell[q] = (ns[q] & CB_HOT_FLAG) ? hot_cell[ns[q] & ~CB_HOT_FLAG] : t.slots[ns[q]].cell_id;
		}
		have_next = next_ok;
		for (uint32_t q = tid; q < in_tile; q += KS_T) {
			const uint32_t d = sd[q], gpos = gdelta[d] + q;
			if (gpos < (rs.first + d + 1u) * rs.cap) {
				okeys[gpos] = sk[q];
				if (VB) ovals[gpos] = sv[q];
			}
		}
		lds_barrier();
	}
	// the shared counters: the waves of the workgroup meet in LDS first (as build_keys does)
	constexpr int WAVES = KS_T / 64;
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

inline size_t ks_dynamic_lds(uint32_t lds_genes) { return size_t(KS_TILE) * 11 + lds_genes; }

}  // namespace dropest
