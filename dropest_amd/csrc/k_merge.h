// k_merge.h -- device side of the cell-barcode merge against a whitelist ("real barcodes").
//
// Replaces the read-only LOOP B of MergeStrategyBase::merge_inited (Estimation/Merge/MergeStrategyBase.cpp:20-27)
// for RealBarcodesMergeStrategy (RealBarcodesMergeStrategy.cpp:22-109):
//   wl_neighbours   per filtered cell: edit distance of each barcode part to EVERY whitelist part
//                   (BarcodesParser::get_distances_to_barcode, BarcodesParser.cpp:21-39), enumeration of the part
//                   combinations by increasing total distance <= 5 (push_remaining_dists, :52-74), barcode-table
//                   lookup of each combination, and the candidate test of RealBarcodesMergeStrategy.cpp:93-103.
//                   The scan stops after the first distance level that yields a candidate (:82-106).
//   umig_intersect  |UMI-genes(base) n UMI-genes(candidate)| (MergeStrategyBase::get_umigs_intersect_size, :100-147)
//                   as a binary-search join of the two cells' sorted molecule ranges.
//   rekey_molecules molecules of merged cells take their target's cell id (Gene::merge, Gene.cpp:26-36, applied by
//                   re-sorting + re-reducing the molecule table).
// The merge decision itself (double arithmetic, strict arg-max, sequential smallest-first application with
// re-targeting, MergeStrategyBase.cpp:30-51,:64-82) runs on the host over the few filtered cells.
// Integer work; no MFMA.
#pragma once

#include "k_cbhash.h"
#include "util.h"

namespace dropest {

constexpr int WL_MAX_LEN = 31;          // bases per barcode part
constexpr int WL_MAX_DIST = 5;          // BarcodesParser::MAX_REAL_MERGE_EDIT_DISTANCE (BarcodesParser.h:57)
constexpr int WL_CAND_CAP = 128;        // candidates kept per cell (exceeding it is reported, never truncated silently)
constexpr int WL_THREADS = 256;
constexpr int WL_MAX_PARTS = 4;         // whitelist parts (lines of a const-length file; the reference ships files of 2 and 3)

struct WlEntry {            // one whitelist part entry
	char seq[32];           // ASCII, NUL padded
};

struct WlBase {             // one filtered cell's barcode, split into the whitelist's parts (BarcodesParser::split_barcode)
	char part[WL_MAX_PARTS][32];
	uint8_t len[WL_MAX_PARTS];
	uint32_t cell;          // cell id of the base
};

// Levenshtein distance with 'N' as a wildcard on either side == Tools::edit_distance(a, b) with its default
// arguments (Tools/UtilFunctions.cpp:32-65; the band is inactive for max_ed = 10000).  Bit-parallel
// (Myers / Hyyro) over the pattern `pat` (length m <= 31) against text `txt` (NUL terminated).
__device__ inline uint32_t wl_peq(const uint32_t peq[5], char c) {
	switch (c) {
		case 'A': return peq[0];
		case 'C': return peq[1];
		case 'G': return peq[2];
		case 'T': return peq[3];
		case 'N': return peq[4];
		default: return peq[4] & 0u;   // foreign letter: matches only pattern wildcards (set below)
	}
}
__device__ inline void wl_build_peq(const char *pat, int m, uint32_t peq[5], uint32_t &wild) {
	peq[0] = peq[1] = peq[2] = peq[3] = 0; wild = 0;
	for (int i = 0; i < m; ++i) {
		const char c = pat[i];
		const uint32_t bit = 1u << i;
		if (c == 'A') peq[0] |= bit; else if (c == 'C') peq[1] |= bit; else if (c == 'G') peq[2] |= bit;
		else if (c == 'T') peq[3] |= bit; else if (c == 'N') wild |= bit;
	}
	peq[0] |= wild; peq[1] |= wild; peq[2] |= wild; peq[3] |= wild;
	peq[4] = m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u);   // a text 'N' matches every pattern position
}
__device__ inline uint32_t wl_edit_distance(const uint32_t peq[5], uint32_t wild, int m, const char *txt) {
	if (m == 0) { int n = 0; while (n < 32 && txt[n]) ++n; return uint32_t(n); }
	uint32_t pv = m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u), mv = 0, score = uint32_t(m);
	const uint32_t last = 1u << (m - 1);
	for (int j = 0; j < 32 && txt[j]; ++j) {
		const char c = txt[j];
		uint32_t eq;
		if (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') eq = wl_peq(peq, c); else eq = wild;
		const uint32_t xv = eq | mv;
		const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
		uint32_t ph = mv | ~(xh | pv);
		uint32_t mh = pv & xh;
		if (ph & last) ++score; else if (mh & last) --score;
		ph = (ph << 1) | 1u;
		mh <<= 1;
		pv = mh | ~(xv | ph);
		mv = ph & xv;
	}
	return score;
}

// the same against a text given as a 2-bit code of `n` bases (most significant base first): no per-character branches
__device__ inline uint32_t wl_edit_distance_code(const uint32_t peq[5], int m, unsigned long long code, int n) {
	if (m == 0) return uint32_t(n);
	uint32_t pv = m >= 32 ? 0xFFFFFFFFu : ((1u << m) - 1u), mv = 0, score = uint32_t(m);
	const uint32_t last = 1u << (m - 1);
	for (int j = n - 1; j >= 0; --j) {
		const uint32_t eq = peq[uint32_t(code >> (2 * j)) & 3u];
		const uint32_t xv = eq | mv;
		const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
		uint32_t ph = mv | ~(xh | pv);
		uint32_t mh = pv & xh;
		if (ph & last) ++score; else if (mh & last) --score;
		ph = (ph << 1) | 1u;
		mh <<= 1;
		pv = mh | ~(xv | ph);
		mv = ph & xv;
	}
	return score;
}

// appends the 2-bit payload of a NUL-terminated ACGT string to `c` (clean strings only)
__device__ inline unsigned long long wl_append(unsigned long long c, const char *s) {
	for (int i = 0; i < 32 && s[i]; ++i) {
		const char ch = s[i];
		c = (c << 2) | (ch == 'C' ? 1ull : ch == 'G' ? 2ull : ch == 'T' ? 3ull : 0ull);
	}
	return c;
}

struct WlArgs {
	const WlBase *bases; uint32_t n_bases;
	const WlEntry *part[WL_MAX_PARTS]; uint32_t part_size[WL_MAX_PARTS];   // whitelist parts
	uint32_t n_parts;
	const unsigned long long *part_code[WL_MAX_PARTS];                     // per entry: length << 58 | 2-bit code, ~0 = not a clean ACGT string
	CbTable table;
	const uint32_t *cell_n_genes, *cell_total_umis;
	uint32_t min_genes;
	uint32_t *cand_count;    // [n_bases] qualifying candidates found (may exceed WL_CAND_CAP: overflow)
	uint32_t *cand_level;    // [n_bases] total distance of the level that produced them
	uint32_t *cand_off;      // [n_bases] first entry of the cell's candidates in the flat lists
	uint32_t *flat_cell;     // [flat_cap] candidate cell ids, all cells back to back (order of cells arbitrary)
	uint32_t *flat_umis;     // [flat_cap] TOTAL_UMIS of each candidate (un-merged state)
	uint32_t *flat_ridx;     // [flat_cap] index of each candidate in the host's real-candidate list
	const uint32_t *cell_real_index;   // [n_cells] cell id -> that index (0xFFFFFFFF for the others)
	uint32_t *flat_total;    // running size of the flat lists (atomic)
	uint32_t flat_cap;
	const uint32_t *base_list;   // optional: block i works on base base_list[i] (the bases the table search left over)
	uint8_t *dist_dump;      // optional [n_bases][part_size[0] + part_size[1]] per-part distances (tie replay), or null
	int poisson;             // PoissonRealBarcodesMergeStrategy::get_max_merge_dist: all levels up to (min == 0 ? 2 : min + 1)
};

// one block per filtered cell; dynamic LDS: dist bytes [ntot] + index lists u16 [ntot], ntot = entries of all parts
// (the number of parts is a template parameter: the per-part arrays then live in registers, not in scratch)
template <uint32_t P>
__global__ __launch_bounds__(WL_THREADS) void wl_neighbours_kernel(WlArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t poff[WL_MAX_PARTS + 1];                  // first entry of each part in dist / lists
	poff[0] = 0;
#pragma unroll
	for (uint32_t p = 0; p < P; ++p) poff[p + 1] = poff[p] + a.part_size[p];
	const uint32_t ntot = poff[P];
	uint8_t *dist = smem;                                             // [ntot]
	uint16_t *lists = reinterpret_cast<uint16_t *>(smem + ((ntot + 15u) & ~15u));   // [ntot], grouped by (part, distance)
	__shared__ uint32_t cnt[WL_MAX_PARTS][WL_MAX_DIST + 2];      // entries per distance 0..5, [6] = farther
	__shared__ uint32_t start[WL_MAX_PARTS][WL_MAX_DIST + 2];
	__shared__ uint32_t fill[WL_MAX_PARTS][WL_MAX_DIST + 2];
	__shared__ uint32_t n_found;
	__shared__ uint32_t found[WL_CAND_CAP];
	__shared__ uint32_t flat_base;

	const uint32_t bi = a.base_list ? a.base_list[blockIdx.x] : blockIdx.x;
	const WlBase &b = a.bases[bi];
	const uint32_t tid = threadIdx.x;
	if (tid < WL_MAX_PARTS * (WL_MAX_DIST + 2)) { (&cnt[0][0])[tid] = 0; (&fill[0][0])[tid] = 0; }
	if (tid == 0) n_found = 0;
	__syncthreads();

	// 1. distances of every part to every whitelist entry of that part
#pragma unroll
	for (uint32_t p = 0; p < P; ++p) {
		uint32_t peq[5], wild;
		wl_build_peq(b.part[p], b.len[p], peq, wild);
		const uint32_t np = a.part_size[p], off = poff[p];
		for (uint32_t i = tid; i < np; i += WL_THREADS) {
			const unsigned long long pc = a.part_code[p][i];
			const uint32_t d = pc != ~0ull ? wl_edit_distance_code(peq, b.len[p], pc & ((1ull << 58) - 1ull), int(pc >> 58))
			                               : wl_edit_distance(peq, wild, b.len[p], a.part[p][i].seq);
			dist[off + i] = uint8_t(d > 255 ? 255 : d);
			atomicAdd(&cnt[p][d > WL_MAX_DIST ? WL_MAX_DIST + 1 : d], 1u);
		}
	}
	__syncthreads();
	if (a.dist_dump) for (uint32_t i = tid; i < ntot; i += WL_THREADS) a.dist_dump[size_t(blockIdx.x) * ntot + i] = dist[i];
	if (tid < P) {
		uint32_t run = poff[tid];
		for (int k = 0; k <= WL_MAX_DIST + 1; ++k) { start[tid][k] = run; run += cnt[tid][k]; }
	}
	__syncthreads();
	// 2. index lists grouped by distance (order inside a group is irrelevant: candidates form a set)
#pragma unroll
	for (uint32_t p = 0; p < P; ++p) {
		const uint32_t np = a.part_size[p], off = poff[p];
		for (uint32_t i = tid; i < np; i += WL_THREADS) {
			const uint32_t d = dist[off + i];
			if (d <= WL_MAX_DIST) lists[start[p][d] + atomicAdd(&fill[p][d], 1u)] = uint16_t(i);
		}
	}
	__syncthreads();

	// 3. levels of increasing total distance (RealBarcodesMergeStrategy::get_real_neighbour_cbs, :63-109): every level
	//    up to max_dist = get_max_merge_dist(smallest distance of ANY whitelist combination), then further levels one
	//    at a time while no candidate has been found; a level is taken whole.  A level = every tuple of per-part
	//    distances with that sum; a tuple contributes the product of its parts' lists.
	const uint32_t base_umis = a.cell_total_umis[b.cell];
	uint32_t min_level = 0;
#pragma unroll
	for (uint32_t p = 0; p < P; ++p) {
		uint32_t m = WL_MAX_DIST + 1;
		for (uint32_t d = 0; d <= WL_MAX_DIST; ++d) if (cnt[p][d]) { m = d; break; }
		min_level += m;                                   // (> WL_MAX_DIST when a part has no entry that close)
	}
	uint32_t n_tuples = 1;
#pragma unroll
	for (uint32_t p = 0; p < P; ++p) n_tuples *= WL_MAX_DIST + 1;
	uint32_t max_dist = a.poisson ? (min_level == 0 ? 2u : min_level + 1u) : min_level;
	uint32_t level = min_level, last_level = min_level;
	for (; level <= WL_MAX_DIST; ++level) {
		last_level = level;
		for (uint32_t t = 0; t < n_tuples; ++t) {
			uint32_t dd[WL_MAX_PARTS], cc[WL_MAX_PARTS], sum = 0, x = t;
			unsigned long long combos = 1;
#pragma unroll
			for (uint32_t p = 0; p < P; ++p) { dd[p] = x % (WL_MAX_DIST + 1); x /= WL_MAX_DIST + 1; sum += dd[p]; cc[p] = cnt[p][dd[p]]; combos *= cc[p]; }
			if (sum != level || combos == 0) continue;
			auto probe = [&](unsigned long long code) {
				const uint32_t s = cb_find(a.table, code);
				if (s == 0xFFFFFFFFu) return;
				const uint32_t c = a.table.slots[s].cell_id;
				if (a.cell_n_genes[c] >= a.min_genes && a.cell_total_umis[c] >= base_umis) {
					const uint32_t k = atomicAdd(&n_found, 1u);
					if (k < WL_CAND_CAP) found[k] = c;
				}
			};
			// packed code of the concatenation (sentinel bit first; entries of a part may differ in length); mixed-radix
			// decoding of the combination number in 32-bit arithmetic whenever it fits (64-bit divisions are slow)
			if (combos <= 0xFFFFFFFFull) {
				const uint32_t n32 = uint32_t(combos);
				for (uint32_t q = tid; q < n32; q += WL_THREADS) {
					unsigned long long code = 1ull;
					uint32_t r = q;
#pragma unroll
					for (uint32_t p = 0; p + 1 < P; ++p) {
						const uint32_t quot = r / cc[p];
						code = wl_append(code, a.part[p][lists[start[p][dd[p]] + (r - quot * cc[p])]].seq);
						r = quot;
					}
					code = wl_append(code, a.part[P - 1][lists[start[P - 1][dd[P - 1]] + r]].seq);
					probe(code);
				}
			} else {
				for (unsigned long long q = tid; q < combos; q += WL_THREADS) {
					unsigned long long code = 1ull, r = q;
#pragma unroll
					for (uint32_t p = 0; p < P; ++p) {
						code = wl_append(code, a.part[p][lists[start[p][dd[p]] + uint32_t(r % cc[p])]].seq);
						r /= cc[p];
					}
					probe(code);
				}
			}
		}
		__syncthreads();                  // n_found of this level is final for everybody
		const uint32_t found_so_far = n_found;
		__syncthreads();                  // ... and read by everybody before the next level adds to it
		if (level > max_dist) max_dist = level;
		if (level + 1 > max_dist && found_so_far) break;
	}
	level = last_level;
	// append this cell's candidates to the flat lists
	const uint32_t keep = n_found < uint32_t(WL_CAND_CAP) ? n_found : uint32_t(WL_CAND_CAP);
	if (tid == 0) {
		a.cand_count[bi] = n_found; a.cand_level[bi] = level;
		flat_base = keep ? atomicAdd(a.flat_total, keep) : 0u;
		a.cand_off[bi] = flat_base;
	}
	__syncthreads();
	for (uint32_t k = tid; k < keep; k += WL_THREADS) {
		const uint32_t o = flat_base + k;
		if (o < a.flat_cap) {
			a.flat_cell[o] = found[k]; a.flat_umis[o] = a.cell_total_umis[found[k]]; a.flat_ridx[o] = a.cell_real_index[found[k]];
		}
	}
}

inline void wl_neighbours_launch(const WlArgs &a, uint32_t n_blocks, size_t lds, hipStream_t stream) {
	switch (a.n_parts) {
		case 1: hipLaunchKernelGGL(wl_neighbours_kernel<1>, dim3(n_blocks), dim3(WL_THREADS), lds, stream, a); break;
		case 2: hipLaunchKernelGGL(wl_neighbours_kernel<2>, dim3(n_blocks), dim3(WL_THREADS), lds, stream, a); break;
		case 3: hipLaunchKernelGGL(wl_neighbours_kernel<3>, dim3(n_blocks), dim3(WL_THREADS), lds, stream, a); break;
		default: hipLaunchKernelGGL(wl_neighbours_kernel<4>, dim3(n_blocks), dim3(WL_THREADS), lds, stream, a); break;
	}
}

// ---- neighbour tables ------------------------------------------------------------------------------------------------
// wl_neighbours_kernel measures every base against every whitelist entry: 2.5 M bases x 2 016 entries at C3 size, 13 ms --
// yet a barcode part of L bases takes one of only 4^L values (10x: 4^7 and 4^9), and nearly every base is decided by its
// CLOSE neighbours (a real barcode: itself at distance 0; a sequencing error of one: its origin at distance 1).  So, once per
// whitelist, every possible value of a part gets a row: how many entries lie at distance 0, 1, 2, 3 and which (WL_TAB_LIST of
// them at most).  A base whose levels end at total distance <= WL_TAB_DMAX with few candidates is then decided by one THREAD
// from two rows; everything else -- parts of another length, N, a needed distance group that did not fit its row, levels beyond
// WL_TAB_DMAX, more than WL_TAB_FOUND candidates -- is left, exactly as before, to wl_neighbours_kernel (cand_count =
// WL_TAB_TODO marks them).
constexpr int WL_TAB_DMAX = 3, WL_TAB_LIST = 60, WL_TAB_MAX_LEN = 9, WL_TAB_FOUND = 12;
constexpr uint32_t WL_TAB_TODO = 0xFFFFFFFFu;
struct WlTabRow { uint16_t cnt[WL_TAB_DMAX + 1]; uint16_t list[WL_TAB_LIST]; };   // group d is listed iff cnt[0] + .. + cnt[d] <= WL_TAB_LIST
static_assert(sizeof(WlTabRow) == 128, "row layout");

// one wave per value v of a part of length L: exact counts of the entries at distance 0 .. WL_TAB_DMAX, and the entries
// themselves grouped by distance for as many of the CLOSEST groups as fit the row (a short part has dozens of entries within
// three edits of anything: its rows then list distances 0 .. 2 only, and a base that needs the third group takes the full search)
__global__ __launch_bounds__(256) void wl_table_build_kernel(const unsigned long long *__restrict__ part_code, uint32_t np, int L, uint32_t n_values,
                                                             WlTabRow *__restrict__ rows) {
	__shared__ uint32_t s_cnt[4][WL_TAB_DMAX + 1];
	__shared__ uint32_t s_item[4][WL_TAB_LIST];
	__shared__ uint32_t s_n[4];
	const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u, v = blockIdx.x * 4 + w;
	if (lane <= WL_TAB_DMAX) s_cnt[w][lane] = 0;
	if (lane == 0) s_n[w] = 0;
	__builtin_amdgcn_wave_barrier();
	if (v >= n_values) return;
	uint32_t peq[5] = {0, 0, 0, 0, 0};
	for (int i = 0; i < L; ++i) peq[(v >> (2 * (L - 1 - i))) & 3u] |= 1u << i;   // pattern = the value's bases, first base = bit 0
	peq[4] = (1u << L) - 1u;
	for (uint32_t i = lane; i < np; i += 64) {
		const unsigned long long pc = part_code[i];
		if (pc == ~0ull) continue;                                                  // (tables are only built for clean whitelists)
		const uint32_t d = wl_edit_distance_code(peq, L, pc & ((1ull << 58) - 1ull), int(pc >> 58));
		if (d <= uint32_t(WL_TAB_DMAX)) atomicAdd(&s_cnt[w][d], 1u);
	}
	__builtin_amdgcn_s_waitcnt(0xC07F);
	__builtin_amdgcn_wave_barrier();
	uint32_t keep_d = 0, run = 0;                  // groups 0 .. keep_d - 1 fit the list
	for (uint32_t d = 0; d <= uint32_t(WL_TAB_DMAX); ++d) { run += s_cnt[w][d]; if (run > uint32_t(WL_TAB_LIST)) break; keep_d = d + 1; }
	for (uint32_t i = lane; i < np && keep_d; i += 64) {
		const unsigned long long pc = part_code[i];
		if (pc == ~0ull) continue;
		const uint32_t d = wl_edit_distance_code(peq, L, pc & ((1ull << 58) - 1ull), int(pc >> 58));
		if (d < keep_d) s_item[w][atomicAdd(&s_n[w], 1u)] = (d << 16) | i;
	}
	__builtin_amdgcn_s_waitcnt(0xC07F);
	__builtin_amdgcn_wave_barrier();
	WlTabRow &row = rows[v];
	const uint32_t n = s_n[w];
	if (lane <= uint32_t(WL_TAB_DMAX)) row.cnt[lane] = uint16_t(s_cnt[w][lane] > 0xFFFEu ? 0xFFFEu : s_cnt[w][lane]);
	// grouped by distance, ascending entry index inside a group (any order would do: candidates form a set)
	if (lane == 0) {
		uint32_t at = 0;
		for (uint32_t d = 0; d < keep_d; ++d) {
			const uint32_t first = at;
			for (uint32_t k = 0; k < n; ++k) if ((s_item[w][k] >> 16) == d) row.list[at++] = uint16_t(s_item[w][k]);
			for (uint32_t x = first + 1; x < at; ++x) {   // insertion sort of a handful of indices
				const uint16_t key = row.list[x]; uint32_t y = x;
				while (y > first && row.list[y - 1] > key) { row.list[y] = row.list[y - 1]; --y; }
				row.list[y] = key;
			}
		}
	}
}

// the bases the table search left over, as a list for wl_neighbours_kernel (order irrelevant)
__global__ __launch_bounds__(256) void wl_collect_todo_kernel(const uint32_t *__restrict__ cand_count, uint32_t n, uint32_t *__restrict__ list, uint32_t *__restrict__ n_todo) {
	const uint32_t f = blockIdx.x * 256 + threadIdx.x;
	const bool mine = f < n && cand_count[f] == WL_TAB_TODO;
	const unsigned long long bal = __ballot(mine);
	if (!bal) return;
	uint32_t base = 0;
	if ((threadIdx.x & 63u) == 0) base = atomicAdd(n_todo, uint32_t(__popcll(bal)));
	base = __shfl(base, 0, 64);
	if (mine) list[base + __builtin_amdgcn_mbcnt_hi(uint32_t(bal >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bal), 0u))] = f;
}

struct WlTabArgs {
	const WlTabRow *rows[WL_MAX_PARTS];
	int len[WL_MAX_PARTS];
};

// one thread per base; same decisions as wl_neighbours_kernel for the bases it takes
template <uint32_t P>
__global__ __launch_bounds__(256) void wl_table_search_kernel(WlArgs a, WlTabArgs t) {
	const uint32_t f = blockIdx.x * 256 + threadIdx.x;
	if (f >= a.n_bases) return;
	const WlBase &b = a.bases[f];
	auto todo = [&]() { a.cand_count[f] = WL_TAB_TODO; };
	const WlTabRow *row[P];
	uint32_t start[P][WL_TAB_DMAX + 2];
#pragma unroll
	for (uint32_t p = 0; p < P; ++p) {
		if (int(b.len[p]) != t.len[p]) return todo();
		uint32_t v = 0;
		for (int i = 0; i < t.len[p]; ++i) {
			const char c = b.part[p][i];
			const uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
			if (code > 3u) return todo();
			v = (v << 2) | code;
		}
		row[p] = t.rows[p] + v;
		uint32_t run = 0;
		for (int d = 0; d <= WL_TAB_DMAX; ++d) { start[p][d] = run; run += row[p]->cnt[d]; }
		start[p][WL_TAB_DMAX + 1] = run;
	}
	const uint32_t base_umis = a.cell_total_umis[b.cell];
	uint32_t min_level = 0;
#pragma unroll
	for (uint32_t p = 0; p < P; ++p) {
		uint32_t m = WL_TAB_DMAX + 1;
		for (int d = 0; d <= WL_TAB_DMAX; ++d) if (row[p]->cnt[d]) { m = uint32_t(d); break; }
		if (m > uint32_t(WL_TAB_DMAX)) return todo();     // nothing that close in this part: the exact minimum is not in the table
		min_level += m;
	}
	uint32_t max_dist = a.poisson ? (min_level == 0 ? 2u : min_level + 1u) : min_level;
	uint32_t found[WL_TAB_FOUND], n_found = 0, level = min_level, last_level = min_level;
	uint32_t n_tuples = 1;
#pragma unroll
	for (uint32_t p = 0; p < P; ++p) n_tuples *= WL_TAB_DMAX + 1;
	for (;; ++level) {
		if (level > uint32_t(WL_TAB_DMAX)) return todo();   // a level the table does not describe completely
		last_level = level;
		for (uint32_t tup = 0; tup < n_tuples; ++tup) {
			uint32_t dd[P], cc[P], sum = 0, x = tup, combos = 1;
			bool empty = false;
#pragma unroll
			for (uint32_t p = 0; p < P; ++p) { dd[p] = x % (WL_TAB_DMAX + 1); x /= WL_TAB_DMAX + 1; sum += dd[p]; cc[p] = row[p]->cnt[dd[p]]; empty |= cc[p] == 0; }
			if (sum != level || empty) continue;
#pragma unroll
			for (uint32_t p = 0; p < P; ++p) if (start[p][dd[p] + 1] > uint32_t(WL_TAB_LIST)) return todo();   // a group the row does not list
#pragma unroll
			for (uint32_t p = 0; p < P; ++p) combos *= cc[p];   // every factor <= WL_TAB_LIST now
			for (uint32_t q = 0; q < combos; ++q) {
				unsigned long long code = 1ull;
				uint32_t r = q;
#pragma unroll
				for (uint32_t p = 0; p < P; ++p) {
					const uint32_t pick = r % cc[p]; r /= cc[p];
					const unsigned long long pc = a.part_code[p][row[p]->list[start[p][dd[p]] + pick]];
					const int n = int(pc >> 58);
					code = (code << (2 * n)) | (pc & ((1ull << (2 * n)) - 1ull));
				}
				const uint32_t s = cb_find(a.table, code);
				if (s == 0xFFFFFFFFu) continue;
				const uint32_t c = a.table.slots[s].cell_id;
				if (a.cell_n_genes[c] >= a.min_genes && a.cell_total_umis[c] >= base_umis) {
					if (n_found >= uint32_t(WL_TAB_FOUND)) return todo();
					found[n_found++] = c;
				}
			}
		}
		if (level > max_dist) max_dist = level;
		if (level + 1 > max_dist && n_found) break;
		if (level == uint32_t(WL_MAX_DIST)) break;          // (unreachable: WL_TAB_DMAX < WL_MAX_DIST sends such bases to the full search)
	}
	a.cand_count[f] = n_found; a.cand_level[f] = last_level;
	const uint32_t o = n_found ? atomicAdd(a.flat_total, n_found) : 0u;
	a.cand_off[f] = o;
	for (uint32_t k = 0; k < n_found; ++k)
		if (o + k < a.flat_cap) { a.flat_cell[o + k] = found[k]; a.flat_umis[o + k] = a.cell_total_umis[found[k]]; a.flat_ridx[o + k] = a.cell_real_index[found[k]]; }
}

// Splits the barcodes of a list of cells into the whitelist's parts on the device
// (BarcodesParser::split_barcode: InDropBarcodesParser.cpp:32-39 -- two parts, the second of fixed length --
// / ConstLengthBarcodesParser.cpp:33-48 -- any number of parts of fixed lengths).
// Escaped barcodes (with N) are left for the host (len[0] = 0xFF); a barcode whose length does not fit sets *bad.
struct WlSplit { uint32_t n_parts; uint32_t len[WL_MAX_PARTS]; int const_kind; };
__global__ __launch_bounds__(256) void make_bases_kernel(const uint32_t *__restrict__ cells, uint32_t n,
                                                         const unsigned long long *__restrict__ cell_cb, WlSplit sp,
                                                         WlBase *__restrict__ out, uint32_t *__restrict__ bad) {
	const uint32_t f = blockIdx.x * 256 + threadIdx.x;
	if (f >= n) return;
	WlBase b;
	for (int p = 0; p < WL_MAX_PARTS; ++p) { for (int i = 0; i < 32; ++i) b.part[p][i] = 0; b.len[p] = 0; }
	b.cell = cells[f];
	const unsigned long long code = cell_cb[b.cell];
	if (code & ESCAPE_BIT) { b.len[0] = 0xFF; out[f] = b; return; }
	const uint32_t len = uint32_t(bit_length(code) - 1) / 2;
	uint32_t lens[WL_MAX_PARTS];
	bool ok = true;
	if (sp.const_kind) {
		uint32_t total = 0;
		for (uint32_t p = 0; p < sp.n_parts; ++p) { lens[p] = sp.len[p]; total += lens[p]; }
		if (len != total) { atomicMax(bad, 1u); ok = false; }
	} else {
		lens[1] = sp.len[1];
		if (len < lens[1]) { atomicMax(bad, 1u); ok = false; } else lens[0] = len - lens[1];
	}
	if (ok) for (uint32_t p = 0; p < sp.n_parts; ++p) if (lens[p] > uint32_t(WL_MAX_LEN)) { atomicMax(bad, 2u); ok = false; }
	if (ok) {
		uint32_t at = 0;                                   // bases consumed so far, from the first base of the barcode
		for (uint32_t p = 0; p < sp.n_parts; ++p) {
			for (uint32_t i = 0; i < lens[p]; ++i) b.part[p][i] = "ACGT"[(code >> (2 * (len - 1 - (at + i)))) & 3];
			b.len[p] = uint8_t(lens[p]);
			at += lens[p];
		}
	}
	out[f] = b;
}

// remap[cell] = cell, then the (source -> target) pairs of the merge are scattered over it
__global__ __launch_bounds__(256) void iota_kernel(uint32_t *p, uint32_t n) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) p[i] = i;
}
__global__ __launch_bounds__(256) void scatter_pairs_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ tgt,
                                                            uint32_t n, uint32_t *__restrict__ remap) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) remap[src[i]] = tgt[i];
}

// ---- UMI-gene intersection sizes ---------------------------------------------------------------------
struct PairRange { uint32_t base_begin, base_end, cand_begin, cand_end; };

__global__ __launch_bounds__(256) void umig_intersect_kernel(const PairRange *__restrict__ pairs, uint32_t n_pairs,
                                                             const unsigned long long *__restrict__ base_key,
                                                             const unsigned long long *__restrict__ mol_key,
                                                             unsigned long long low_mask, int umi_bits,
                                                             unsigned long long gene_none, uint32_t *__restrict__ out) {
	__shared__ uint32_t scratch[256 / 64 + 1];
	const PairRange r = pairs[blockIdx.x];
	uint32_t c = 0;
	for (uint32_t i = r.base_begin + threadIdx.x; i < r.base_end; i += 256) {
		const unsigned long long k = base_key[i] & low_mask;   // the base's rows may live in another table (sharded runs)
		if ((k >> umi_bits) == gene_none) continue;          // reads without a gene are not UMI-genes
		uint32_t lo = r.cand_begin, hi = r.cand_end;
		while (lo < hi) {
			const uint32_t mid = lo + ((hi - lo) >> 1);
			if ((mol_key[mid] & low_mask) < k) lo = mid + 1; else hi = mid;
		}
		if (lo < r.cand_end && (mol_key[lo] & low_mask) == k) ++c;
	}
	uint32_t total;
	block_excl_scan_u32<256>(c, scratch, total);
	if (threadIdx.x == 0) out[blockIdx.x] = total;
}

// ---- re-keying after the merge decisions --------------------------------------------------------------
// remap[cell] = final target of a merged cell, identity otherwise
__global__ __launch_bounds__(256) void rekey_molecules_kernel(const unsigned long long *__restrict__ mol_key, uint32_t n,
                                                              int cell_shift, const uint32_t *__restrict__ remap,
                                                              unsigned long long *__restrict__ keys,
                                                              uint32_t *__restrict__ vals, unsigned long long *key_or_and) {
	unsigned long long k_or = 0, k_and = ~0ull;
	const uint32_t stride = gridDim.x * 256;
	const unsigned long long low = (1ull << cell_shift) - 1ull;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		const unsigned long long k = mol_key[i];
		const unsigned long long nk = ((unsigned long long)remap[uint32_t(k >> cell_shift)] << cell_shift) | (k & low);
		keys[i] = nk; vals[i] = i;
		k_or |= nk; k_and &= nk;
	}
	k_or = wave_reduce_or_u64(k_or); k_and = wave_reduce_and_u64(k_and);
	__shared__ unsigned long long w_or[4], w_and[4];   // (256 threads: the four waves meet here, one pair of atomics per workgroup)
	if (lane_id() == 0) { w_or[threadIdx.x >> 6] = k_or; w_and[threadIdx.x >> 6] = k_and; }
	__syncthreads();
	if (threadIdx.x == 0) { atomicOr(&key_or_and[0], w_or[0] | w_or[1] | w_or[2] | w_or[3]); atomicAnd(&key_or_and[1], w_and[0] & w_and[1] & w_and[2] & w_and[3]); }
}

}  // namespace dropest
