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
//   directional_big   one wave per group of 17..4096 clean UMIs: libstdc++'s introsort restated step for step on an index
//                     array (the arrangement of equal read counts is a property of that algorithm), targets in parallel
//   directional_huge  the same with the work arrays in global scratch, for the handful of groups beyond 4096 UMIs
// Groups with an N-UMI (random fills draw from glibc rand() in cell order) or with UMIs of several lengths are listed
// for the host, which replays the reference literally (umi_directional_host.h).
// Integer work; no MFMA.
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
	uint32_t *big_list, *big_count;     // groups of 17 .. DIR_BIG_MAX clean UMIs: directional_big_kernel
	uint32_t *huge_list, *huge_count;   // larger clean groups: directional_huge_kernel
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
	if (has_n || a.umi_len == 0 || a.umi_bits > 32) {
		a.host_list[atomicAdd(a.host_count, 1u)] = g;
		return;
	}
	if (k > 4096u /* DIR_BIG_MAX */) {
		a.huge_list[atomicAdd(a.huge_count, 1u)] = g;
		return;
	}
	if (k > uint32_t(DIR_MAX_GROUP)) {
		a.big_list[atomicAdd(a.big_count, 1u)] = g;
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

// ---- groups of 17 .. DIR_BIG_MAX UMIs: one wave per group ------------------------------------------------------
// For more than 16 elements std::sort is libstdc++'s introsort, whose arrangement of EQUAL read counts is a property of
// the algorithm, not of the data.  It is restated here step for step (bits/stl_algo.h: __introsort_loop with the
// median-of-three pivot and the unguarded partition, depth limit 2 * floor(log2 n) with the heapsort fallback of
// __partial_sort, then __final_insertion_sort; bits/stl_heap.h: __make_heap / __adjust_heap / __push_heap / __pop_heap)
// over an index array whose initial order is the UMI-index order, comparing read counts only -- the same comparisons
// and moves the reference's std::sort performs on its vector of (sequence, reads).
constexpr int DIR_BIG_MAX = 4096;

template <typename IdxT>
struct StdSort {
	IdxT *a;                    // element i of the "vector" = molecule a[i]
	const uint32_t *reads;
	__device__ bool less(IdxT x, IdxT y) const { return reads[x] < reads[y]; }
	__device__ void swap(int i, int j) { const IdxT t = a[i]; a[i] = a[j]; a[j] = t; }

	__device__ void push_heap(int first, int hole, int top, IdxT value) {
		int parent = (hole - 1) / 2;
		while (hole > top && less(a[first + parent], value)) { a[first + hole] = a[first + parent]; hole = parent; parent = (hole - 1) / 2; }
		a[first + hole] = value;
	}
	__device__ void adjust_heap(int first, int hole, int len, IdxT value) {
		const int top = hole;
		int child = hole;
		while (child < (len - 1) / 2) {
			child = 2 * (child + 1);
			if (less(a[first + child], a[first + child - 1])) --child;
			a[first + hole] = a[first + child];
			hole = child;
		}
		if ((len & 1) == 0 && child == (len - 2) / 2) {
			child = 2 * (child + 1);
			a[first + hole] = a[first + child - 1];
			hole = child - 1;
		}
		push_heap(first, hole, top, value);
	}
	__device__ void heap_sort(int first, int last) {          // __partial_sort(first, last, last)
		const int len = last - first;
		if (len >= 2)
			for (int parent = (len - 2) / 2;; --parent) { adjust_heap(first, parent, len, a[first + parent]); if (parent == 0) break; }
		while (last - first > 1) {
			--last;
			const IdxT value = a[last];
			a[last] = a[first];
			adjust_heap(first, 0, last - first, value);
		}
	}
	__device__ void move_median_to_first(int result, int x, int y, int z) {
		if (less(a[x], a[y])) {
			if (less(a[y], a[z])) swap(result, y);
			else if (less(a[x], a[z])) swap(result, z);
			else swap(result, x);
		} else if (less(a[x], a[z])) swap(result, x);
		else if (less(a[y], a[z])) swap(result, z);
		else swap(result, y);
	}
	__device__ int unguarded_partition(int first, int last, int pivot) {
		for (;;) {
			while (less(a[first], a[pivot])) ++first;
			--last;
			while (less(a[pivot], a[last])) --last;
			if (!(first < last)) return first;
			swap(first, last);
			++first;
		}
	}
	__device__ void unguarded_linear_insert(int last) {
		const IdxT value = a[last];
		int next = last - 1;
		while (less(value, a[next])) { a[last] = a[next]; last = next; --next; }
		a[last] = value;
	}
	__device__ void insertion_sort(int first, int last) {
		if (first == last) return;
		for (int i = first + 1; i != last; ++i) {
			if (less(a[i], a[first])) {
				const IdxT value = a[i];
				for (int j = i; j > first; --j) a[j] = a[j - 1];
				a[first] = value;
			} else unguarded_linear_insert(i);
		}
	}
	__device__ void sort(int n) {
		if (n == 0) return;
		int depth = 0;
		for (int m = n; m > 1; m >>= 1) ++depth;
		depth *= 2;
		// __introsort_loop: the recursion on [cut, last) becomes an explicit stack (the ranges are disjoint, so the
		// order in which they are finished does not change the result)
		int st_first[64], st_last[64], st_depth[64], sp = 0;
		st_first[0] = 0; st_last[0] = n; st_depth[0] = depth; sp = 1;
		while (sp) {
			--sp;
			int first = st_first[sp], last = st_last[sp], d = st_depth[sp];
			while (last - first > 16) {
				if (d == 0) { heap_sort(first, last); break; }
				--d;
				const int mid = first + (last - first) / 2;
				move_median_to_first(first, first + 1, mid, last - 1);
				const int cut = unguarded_partition(first + 1, last, first);
				st_first[sp] = cut; st_last[sp] = last; st_depth[sp] = d; ++sp;
				last = cut;
			}
		}
		if (n > 16) { insertion_sort(0, 16); for (int i = 16; i < n; ++i) unguarded_linear_insert(i); }
		else insertion_sort(0, n);
	}
};

struct DirBigArgs {
	const uint32_t *groups; uint32_t n_groups;     // (cell, gene) rows with more than 16 clean UMIs
	const uint32_t *scratch_off;                   // huge groups only: start of each group's slice of the scratch arrays
	uint32_t *s_code, *s_reads, *s_first, *s_ord; int32_t *s_tgt;
	const uint32_t *cg_mol_begin; const unsigned long long *cg_key;
	const unsigned long long *mol_key; const uint32_t *mol_reads;
	int gene_bits, umi_bits, umi_len;
	const uint32_t *umi_first;
	double mult; uint32_t max_ed;
	unsigned long long *new_key; uint32_t *cell_removed; uint32_t *n_changed;
};

// One workgroup per group.  Work arrays: code / reads / first occurrence per UMI, the index vector being sorted, targets.
template <typename IdxT, typename TgtT>
__device__ inline void directional_group(const DirBigArgs &a, uint32_t g, uint32_t *code, uint32_t *reads, uint32_t *first, IdxT *ord,
                                         TgtT *tgt) {
	const uint32_t b = a.cg_mol_begin[g], k = a.cg_mol_begin[g + 1] - b;
	const uint32_t tid = threadIdx.x, nt = blockDim.x;
	const unsigned long long umask = (1ull << a.umi_bits) - 1ull;
	for (uint32_t j = tid; j < k; j += nt) {
		const unsigned long long c = a.mol_key[b + j] & umask;
		code[j] = uint32_t(c); reads[j] = a.mol_reads[b + j]; first[j] = a.umi_first[c];
	}
	__syncthreads();
	// UMI-index order: rank of each UMI by its first occurrence (distinct per UMI)
	for (uint32_t j = tid; j < k; j += nt) {
		uint32_t r = 0;
		const uint32_t f = first[j];
		for (uint32_t t = 0; t < k; ++t) r += first[t] < f;
		ord[r] = IdxT(j);
	}
	__syncthreads();
	if (tid == 0) { StdSort<IdxT> s{ord, reads}; s.sort(int(k)); }
	__syncthreads();
	// find_target for every source position (threads take positions round-robin)
	for (uint32_t s = tid; s < k; s += nt) {
		const uint32_t ps = ord[s];
		int t = -1;
		uint32_t min_ed = 0xFFFFFFFFu;
		for (int d = int(k) - 1; d > int(s); --d) {
			const uint32_t pd = ord[d];
			if (double(reads[ps]) * a.mult > double(reads[pd])) break;
			const uint32_t ed = umi_code_distance(code[ps], code[pd], a.umi_len, a.max_ed);
			if (ed > a.max_ed) continue;
			if (ed < min_ed) { t = d; if (ed <= 1u) break; min_ed = ed; }
		}
		tgt[s] = TgtT(t);
	}
	__syncthreads();
	if (tid == 0)
		for (int s = int(k) - 1; s >= 0; --s)
			if (tgt[s] >= 0 && tgt[tgt[s]] >= 0) tgt[s] = tgt[tgt[s]];
	__syncthreads();
	uint32_t removed = 0;
	for (uint32_t s = tid; s < k; s += nt) {
		if (tgt[s] < 0) continue;
		const uint32_t ps = ord[s], pt = ord[tgt[s]];
		a.new_key[b + ps] = (a.mol_key[b + ps] & ~umask) | code[pt];
		++removed;
	}
	removed = uint32_t(wave_reduce_add_u64(removed));
	if (lane_id() == 0 && removed) {
		atomicAdd(&a.cell_removed[uint32_t(a.cg_key[g] >> a.gene_bits)], removed);
		atomicAdd(a.n_changed, removed);
	}
}

// 17 .. DIR_BIG_MAX UMIs: one wave per group, work arrays in LDS (64 KB)
__global__ __launch_bounds__(64) void directional_big_kernel(DirBigArgs a) {
	__shared__ uint32_t code[DIR_BIG_MAX], reads[DIR_BIG_MAX], first[DIR_BIG_MAX];
	__shared__ uint16_t ord[DIR_BIG_MAX];
	__shared__ int16_t tgt[DIR_BIG_MAX];
	directional_group<uint16_t, int16_t>(a, a.groups[blockIdx.x], code, reads, first, ord, tgt);
}
// more than DIR_BIG_MAX UMIs (a handful of groups: one hot gene in one big cell): 256 threads per group, work arrays
// in a global scratch slice
__global__ __launch_bounds__(256) void directional_huge_kernel(DirBigArgs a) {
	const uint32_t o = a.scratch_off[blockIdx.x];
	directional_group<uint32_t, int32_t>(a, a.groups[blockIdx.x], a.s_code + o, a.s_reads + o, a.s_first + o, a.s_ord + o, a.s_tgt + o);
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
	__shared__ unsigned long long w_or[4], w_and[4];   // (256 threads: the four waves meet here, one pair of atomics per workgroup)
	if (lane_id() == 0) { w_or[threadIdx.x >> 6] = k_or; w_and[threadIdx.x >> 6] = k_and; }
	__syncthreads();
	if (threadIdx.x == 0) { atomicOr(&key_or_and[0], w_or[0] | w_or[1] | w_or[2] | w_or[3]); atomicAnd(&key_or_and[1], w_and[0] & w_and[1] & w_and[2] & w_and[3]); }
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ idx,
                                                         uint32_t n, uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = table[idx[i]];
}

}  // namespace dropest
