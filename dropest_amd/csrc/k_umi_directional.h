// k_umi_directional.h -- device side of the "directional" UMI correction (-u):
// MergeUMIsStrategyDirectional::merge / find_targets / find_target
// (Estimation/Merge/UMIs/MergeUMIsStrategyDirectional.cpp:18-116).
//
// Per (real cell, gene) group the reference builds the vector of (UMI, read count) in UMI-INDEX order (= order of
// the UMI's first occurrence among gene-bearing reads, StringIndexer), std::sorts it by read count, and lets every
// UMI look -- from the most-read UMI downwards, while `reads(src) * mult <= reads(dst)` -- for the first UMI within
// the edit distance; chains are then shortened once from the top.  On the device:
//   umi_first_table   first read ordinal of every UMI code (one pass over the reads, atomicMin into a table indexed
//                     by the key's UMI field)
//   directional       one thread per group of up to 16 clean UMIs of one length: libstdc++'s std::sort IS an insertion
//                     sort (stable) for <= 16 elements, so the order is (reads, first occurrence); targets, chain
//                     shortening and the re-keyed molecule keys are written in place
// Groups with an N-UMI (random fills draw from glibc rand() in cell order), with more than 16 UMIs (introsort: order
// of equal read counts is implementation-defined) or with UMIs of several lengths are listed for the host, which
// replays the reference literally (umi_directional_host.h).  Integer work; no MFMA.
#pragma once

#include "k_cbhash.h"
#include "k_misc.h"
#include "util.h"

namespace dropest {

constexpr int DIR_MAX_GROUP = 16;   // libstdc++ _S_threshold: std::sort degenerates to insertion sort up to here

__global__ __launch_bounds__(256) void umi_first_table_kernel(const unsigned long long *__restrict__ umi,
                                                              const uint32_t *__restrict__ gene, uint32_t n, KeyLayout L,
                                                              uint32_t *__restrict__ first) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
		if (gene[r] == NO_GENE) continue;
		const unsigned long long u = umi[r];
		const unsigned long long code = (u & ESCAPE_BIT) ? (L.umi_escape_base + (u & ~ESCAPE_BIT)) : (u & L.umi_strip_mask);
		if (r < first[code]) atomicMin(&first[code], r);
	}
}

// Levenshtein distance of two clean 2-bit codes of `len` bases (== Tools::edit_distance whenever the result is
// <= max_ed and both strings have one length: its band then never cuts an optimal path, Tools/UtilFunctions.cpp:32-65)
__device__ inline uint32_t umi_code_distance(unsigned long long a, unsigned long long b, int len, uint32_t max_ed) {
	const unsigned long long x = a ^ b;
	const uint32_t ham = uint32_t(__popcll((x | (x >> 1)) & 0x5555555555555555ull));
	if (ham <= 1u || max_ed <= 1u) return ham;          // distance <= 1 <=> Hamming <= 1 for equal lengths
	uint8_t row[32];
	for (int i = 0; i <= len; ++i) row[i] = uint8_t(i);
	for (int j = 1; j <= len; ++j) {
		const uint32_t cb = uint32_t(b >> (2 * (len - j))) & 3u;
		uint8_t diag = row[0];
		row[0] = uint8_t(j);
		for (int i = 1; i <= len; ++i) {
			const uint32_t ca = uint32_t(a >> (2 * (len - i))) & 3u;
			const uint8_t up = row[i];
			uint8_t v = uint8_t(diag + (ca != cb));
			if (uint8_t(up + 1) < v) v = uint8_t(up + 1);
			if (uint8_t(row[i - 1] + 1) < v) v = uint8_t(row[i - 1] + 1);
			row[i] = v; diag = up;
		}
	}
	return row[len];
}

struct DirArgs {
	const unsigned long long *cg_key; const uint32_t *cg_mol_begin; uint32_t n_cg;
	const unsigned long long *mol_key; const uint32_t *mol_reads;
	const uint32_t *real_flag;          // [n_cells] 1 = Cell::is_real now
	int gene_bits, umi_bits, umi_len;   // umi_len = bases of a clean UMI, 0 = UMIs of several lengths (everything goes to the host)
	unsigned long long gene_none, escape_base;
	const uint32_t *umi_first;          // [2^umi_bits]
	double mult; uint32_t max_ed;
	unsigned long long *new_key;        // [n_mol] pre-filled with mol_key; merged sources get their root's UMI
	uint32_t *cell_removed;             // [n_cells] += re-keyed UMIs (Cell::merge_umis decrements TOTAL_UMIS once per pair)
	uint32_t *host_list, *host_count;   // groups left to the host
	uint32_t *n_changed;                // re-keyed molecules in total
};

__global__ __launch_bounds__(256) void directional_kernel(DirArgs a) {
	const uint32_t g = blockIdx.x * 256 + threadIdx.x;
	if (g >= a.n_cg) return;
	const unsigned long long ck = a.cg_key[g];
	if ((ck & a.gene_none) == a.gene_none) return;                       // reads without a gene
	const uint32_t cell = uint32_t(ck >> a.gene_bits);
	if (!a.real_flag[cell]) return;
	const uint32_t b = a.cg_mol_begin[g], k = a.cg_mol_begin[g + 1] - b;
	const unsigned long long umask = (1ull << a.umi_bits) - 1ull;
	const bool has_n = (a.mol_key[b + k - 1] & umask) >= a.escape_base;  // escaped codes sort last in the group
	if (k < 2 && !has_n) return;
	if (has_n || k > uint32_t(DIR_MAX_GROUP) || a.umi_len == 0) {
		a.host_list[atomicAdd(a.host_count, 1u)] = g;
		return;
	}
	unsigned long long code[DIR_MAX_GROUP];
	uint32_t reads[DIR_MAX_GROUP], first[DIR_MAX_GROUP];
	int8_t pos[DIR_MAX_GROUP], tgt[DIR_MAX_GROUP];   // pos[s] = molecule (offset in the group) at sorted position s
	for (uint32_t j = 0; j < k; ++j) {
		code[j] = a.mol_key[b + j] & umask; reads[j] = a.mol_reads[b + j]; first[j] = a.umi_first[code[j]];
	}
	// insertion sort by (reads, first occurrence)
	for (uint32_t j = 0; j < k; ++j) {
		int s = int(j);
		while (s > 0) {
			const int p = pos[s - 1];
			if (reads[p] < reads[j] || (reads[p] == reads[j] && first[p] < first[j])) break;
			pos[s] = pos[s - 1]; --s;
		}
		pos[s] = int8_t(j);
	}
	// find_target (:83-116) for every source, then the chain shortening of find_targets (:66-78)
	bool any = false;
	for (uint32_t s = 0; s < k; ++s) {
		tgt[s] = -1;
		const int ps = pos[s];
		uint32_t min_ed = 0xFFFFFFFFu;
		for (int d = int(k) - 1; d > int(s); --d) {
			const int pd = pos[d];
			if (double(reads[ps]) * a.mult > double(reads[pd])) break;
			const uint32_t ed = umi_code_distance(code[ps], code[pd], a.umi_len, a.max_ed);
			if (ed > a.max_ed) continue;
			if (ed < min_ed) {
				tgt[s] = int8_t(d);
				if (ed <= 1u) break;
				min_ed = ed;
			}
		}
		any |= tgt[s] >= 0;
	}
	if (!any) return;
	for (int s = int(k) - 1; s >= 0; --s)
		if (tgt[s] >= 0 && tgt[int(tgt[s])] >= 0) tgt[s] = tgt[int(tgt[s])];
	uint32_t removed = 0;
	for (uint32_t s = 0; s < k; ++s) {
		if (tgt[s] < 0) continue;
		const int ps = pos[s], pt = pos[int(tgt[s])];
		a.new_key[b + ps] = (a.mol_key[b + ps] & ~umask) | code[pt];
		++removed;
	}
	atomicAdd(&a.cell_removed[cell], removed);
	atomicAdd(a.n_changed, removed);
}

__global__ __launch_bounds__(256) void iota_or_and_kernel(const unsigned long long *__restrict__ keys, uint32_t n,
                                                          uint32_t *__restrict__ vals, unsigned long long *key_or_and) {
	unsigned long long k_or = 0, k_and = ~0ull;
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		const unsigned long long k = keys[i];
		vals[i] = i; k_or |= k; k_and &= k;
	}
	k_or = wave_reduce_or_u64(k_or); k_and = wave_reduce_and_u64(k_and);
	if (lane_id() == 0) { atomicOr(&key_or_and[0], k_or); atomicAnd(&key_or_and[1], k_and); }
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ idx,
                                                         uint32_t n, uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = table[idx[i]];
}

}  // namespace dropest
