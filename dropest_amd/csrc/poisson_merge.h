// poisson_merge.h -- PoissonRealBarcodesMergeStrategy (-M with a barcodes file): the neighbour search of the
// whitelist merge with a wider distance rule (k_merge.h, WlArgs::poisson) and PoissonTargetEstimator's decision
// (Estimation/Merge/PoissonTargetEstimator.cpp:14-127): for a base cell and each neighbour
//   intersection = |UMI-gene pairs in common|
//   expected     = SUM over genes in common of  est(adj(size1), adj(size2)),  sizes = UMIs of the gene in either cell,
//                  adj = Tools::CollisionsAdjuster table over the UMI distribution of the filtered cells,
//                  est(a1 <= a2) = SUM_u (1 - (1-p_u)^a1) * (1 - (1-p_u)^a1 (1-p_u)^(a2-a1))
//   prob         = ppois(intersection - 1, expected, lower = false) = P(Poisson(expected) >= intersection)
// and the target is the neighbour of smallest prob if that is <= max_(real_)merge_prob / |neighbours|.
//
// Device: the UMI distribution (sort + run lengths, then classes of equally frequent UMIs), the adjuster table, the genes in common of every
// (base, neighbour) pair, one est() per DISTINCT pair of adjusted sizes (the reference caches the same way,
// :104-108), and the per-pair sums.  Host: the Poisson tail and the few comparisons per cell.
// Floating point: est() and the adjuster's sums are reductions over the UMI distribution; the reference adds them
// in the iteration order of an unordered_map of strings, the kernels per class of equally frequent UMIs (count x term)
// in a fixed tree order, so `expected` agrees to rounding (measured <= 1e-12 relative) and the decisions agree unless
// a probability lies that close to its threshold.
#pragma once

#include "context.h"

namespace dropest {

constexpr int PM_THREADS = 256;

// genes in common of pair p (merge join of the two cells' (cell, gene) rows, which are sorted by gene): COUNT pass
// writes the number, WRITE pass the keys (adj(min size) << 32 | adj(max size)) at off[p]..
template <bool WRITE>
__global__ __launch_bounds__(256) void common_genes_kernel(const uint32_t *__restrict__ base_cell, const uint32_t *__restrict__ cand_cell,
                                                           uint32_t n_pairs, const uint32_t *__restrict__ cell_cg_begin,
                                                           const uint32_t *__restrict__ cell_cg_count,
                                                           const unsigned long long *__restrict__ cg_key,
                                                           const uint32_t *__restrict__ cg_mol_begin, unsigned long long gene_mask,
                                                           const unsigned long long *__restrict__ adjusted,
                                                           uint32_t *__restrict__ count, const uint32_t *__restrict__ off,
                                                           unsigned long long *__restrict__ out) {
	const uint32_t p = blockIdx.x * 256 + threadIdx.x;
	if (p >= n_pairs) return;
	const uint32_t b = base_cell[p], c = cand_cell[p];
	uint32_t i = cell_cg_begin[b], ie = i + cell_cg_count[b], j = cell_cg_begin[c], je = j + cell_cg_count[c];
	uint32_t n = 0;
	const uint32_t o = WRITE ? off[p] : 0u;
	while (i < ie && j < je) {
		const unsigned long long gi = cg_key[i] & gene_mask, gj = cg_key[j] & gene_mask;
		if (gi < gj) ++i;
		else if (gj < gi) ++j;
		else {
			if (gi != gene_mask) {   // reads without a gene are not a gene of the cell
				if (WRITE) {
					unsigned long long s1 = cg_mol_begin[i + 1] - cg_mol_begin[i], s2 = cg_mol_begin[j + 1] - cg_mol_begin[j];
					if (s1 > s2) { const unsigned long long t = s1; s1 = s2; s2 = t; }
					out[o + n] = (adjusted[s1 - 1] << 32) | adjusted[s2 - 1];
				}
				++n;
			}
			++i; ++j;
		}
	}
	if (!WRITE) count[p] = n;
}

__global__ __launch_bounds__(256) void max_gene_size_kernel(const unsigned long long *__restrict__ cg_key, const uint32_t *__restrict__ cg_mol_begin,
                                                            uint32_t n_cg, unsigned long long gene_mask, uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	uint32_t v = 0;
	if (i < n_cg && (cg_key[i] & gene_mask) != gene_mask) v = cg_mol_begin[i + 1] - cg_mol_begin[i];
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v = max(v, uint32_t(__shfl_down(v, d, 64)));
	if (lane_id() == 0 && v) atomicMax(out, v);
}

__global__ __launch_bounds__(256) void widen_counts_kernel(const uint32_t *__restrict__ count, uint32_t n, unsigned long long *__restrict__ out,
                                                           uint32_t *__restrict__ max_out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	uint32_t v = i < n ? count[i] : 0u;
	if (i < n) out[i] = v;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v = max(v, uint32_t(__shfl_down(v, d, 64)));
	if (lane_id() == 0 && v) atomicMax(max_out, v);
}

// UMIs that were seen equally often have the same probability and go through every formula below together: the
// distribution is kept as (probability, number of UMIs with it) classes -- a few dozen to a few thousand entries
// where the reference walks millions of UMIs
__global__ __launch_bounds__(256) void classes_to_probs_kernel(const unsigned long long *__restrict__ count_value,
                                                               const uint32_t *__restrict__ multiplicity, uint32_t n, double total,
                                                               double *__restrict__ p, double *__restrict__ mult, double *__restrict__ ones) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) { p[i] = double(count_value[i]) / total; mult[i] = double(multiplicity[i]); ones[i] = 1.0; }
}

// Tools::CollisionsAdjuster table (CollisionsAdjuster.cpp:12-49) over the probability classes, one block: the
// recurrence over s is sequential, the work per s a reduction over the classes (fixed order: strided per thread,
// wave shuffle tree, waves in index order)
__global__ __launch_bounds__(PM_THREADS) void collisions_table_kernel(const double *__restrict__ p, const double *__restrict__ mult,
                                                                      double *__restrict__ neg_prod, uint32_t n_classes, uint32_t max_size,
                                                                      unsigned long long *__restrict__ adjusted) {
	__shared__ double wave_sum[PM_THREADS / 64];
	__shared__ unsigned long long delta_s;
	double sum_collisions = 0;                 // meaningful in thread 0
	unsigned long long last_total = 0;
	for (uint32_t s = 1; s <= max_size; ++s) {
		if (threadIdx.x == 0) {
			const unsigned long long total = s + (unsigned long long)sum_collisions;
			delta_s = total - last_total;
			last_total = total;
		}
		__syncthreads();
		const unsigned long long delta = delta_s;
		double acc = 0;
		for (uint32_t i = threadIdx.x; i < n_classes; i += PM_THREADS) {
			const double np = neg_prod[i] * dev_fpow(1 - p[i], delta);
			neg_prod[i] = np;
			acc += mult[i] * (p[i] * (1 - np));
		}
#pragma unroll
		for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
		if (lane_id() == 0) wave_sum[wave_id()] = acc;
		__syncthreads();
		if (threadIdx.x == 0) {
			double new_prob = 0;
			for (int w = 0; w < PM_THREADS / 64; ++w) new_prob += wave_sum[w];
			sum_collisions += 1.0 / (1.0 - new_prob) - 1.0;
			adjusted[s - 1] = isfinite(sum_collisions) && sum_collisions < 4e9 ? (unsigned long long)lround(double(s) + sum_collisions) : ~0ull;
		}
	}
}

// est(a1, a2) of one distinct size pair per block (PoissonTargetEstimator::estimate_genes_intersection_size, :110-123)
__global__ __launch_bounds__(PM_THREADS) void genes_intersection_kernel(const unsigned long long *__restrict__ size_pair, uint32_t n_keys,
                                                                        const double *__restrict__ p, const double *__restrict__ mult,
                                                                        uint32_t n_umis, double *__restrict__ est) {
	__shared__ double wave_sum[PM_THREADS / 64];
	const unsigned long long key = size_pair[blockIdx.x];
	const unsigned long long a1 = key >> 32, d = (key & 0xFFFFFFFFull) - a1;
	double acc = 0;
	for (uint32_t i = threadIdx.x; i < n_umis; i += PM_THREADS) {
		const double q = 1 - p[i];
		const double mn = dev_fpow(q, a1);
		const double mx = mn * dev_fpow(q, d);
		acc += mult[i] * ((1 - mn) * (1 - mx));
	}
#pragma unroll
	for (int s = 32; s > 0; s >>= 1) acc += __shfl_down(acc, s, 64);
	if (lane_id() == 0) wave_sum[wave_id()] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		double s = 0;
		for (int w = 0; w < PM_THREADS / 64; ++w) s += wave_sum[w];
		est[blockIdx.x] = s;
	}
	(void)n_keys;
}

// expected intersection of pair p: its genes' est() values added in gene order (:79-88)
__global__ __launch_bounds__(256) void expected_intersection_kernel(const uint32_t *__restrict__ off, uint32_t n_pairs,
                                                                    const unsigned long long *__restrict__ pair_keys,
                                                                    const unsigned long long *__restrict__ uniq, uint32_t n_uniq,
                                                                    const double *__restrict__ est, double *__restrict__ expected) {
	const uint32_t p = blockIdx.x * 256 + threadIdx.x;
	if (p >= n_pairs) return;
	double e = 0;
	for (uint32_t i = off[p]; i < off[p + 1]; ++i) {
		const unsigned long long k = pair_keys[i];
		uint32_t lo = 0, hi = n_uniq;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (uniq[mid] < k) lo = mid + 1; else hi = mid; }
		e += est[lo];
	}
	expected[p] = e;
}

// ---- sharded -M: the base's molecule rows arrive from another shard (merge_shard.h) ---------------------------------------
// genes in common of pair p = (rows [bb, be) of base_low: gene << umi_bits | UMI, ascending; local cell cand): same keys, same gene
// order as common_genes_kernel
template <bool WRITE>
__global__ __launch_bounds__(256) void common_genes_ext_kernel(const uint32_t *__restrict__ bb, const uint32_t *__restrict__ be,
                                                               const uint32_t *__restrict__ cand_cell, uint32_t n_pairs,
                                                               const unsigned long long *__restrict__ base_low, int umi_bits,
                                                               const uint32_t *__restrict__ cell_cg_begin, const uint32_t *__restrict__ cell_cg_count,
                                                               const unsigned long long *__restrict__ cg_key, const uint32_t *__restrict__ cg_mol_begin,
                                                               unsigned long long gene_mask, const unsigned long long *__restrict__ adjusted,
                                                               uint32_t adjusted_n, uint32_t *__restrict__ count, const uint32_t *__restrict__ off,
                                                               unsigned long long *__restrict__ out, uint32_t *__restrict__ too_large) {
	const uint32_t p = blockIdx.x * 256 + threadIdx.x;
	if (p >= n_pairs) return;
	const uint32_t c = cand_cell[p];
	uint32_t i = bb[p];
	const uint32_t ie = be[p];
	uint32_t j = cell_cg_begin[c];
	const uint32_t je = j + cell_cg_count[c];
	uint32_t n = 0;
	const uint32_t o = WRITE ? off[p] : 0u;
	while (i < ie && j < je) {
		const unsigned long long gi = (base_low[i] >> umi_bits) & gene_mask, gj = cg_key[j] & gene_mask;
		uint32_t run = i + 1;
		while (run < ie && ((base_low[run] >> umi_bits) & gene_mask) == gi) ++run;   // the base's molecules of gene gi
		if (gi < gj) i = run;
		else if (gj < gi) ++j;
		else {
			if (gi != gene_mask) {
				if (WRITE) {
					unsigned long long s1 = run - i, s2 = cg_mol_begin[j + 1] - cg_mol_begin[j];
					if (s1 > s2) { const unsigned long long t = s1; s1 = s2; s2 = t; }
					if (s2 > adjusted_n) { atomicMax(too_large, 1u); s1 = s2 = 1; }
					out[o + n] = (adjusted[s1 - 1] << 32) | adjusted[s2 - 1];
				}
				++n;
			}
			i = run; ++j;
		}
	}
	if (!WRITE) count[p] = n;
}
// dense histogram of UMI codes
__global__ __launch_bounds__(256) void umi_histogram_kernel(const unsigned long long *__restrict__ codes, uint32_t n, uint32_t *__restrict__ hist) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) atomicAdd(&hist[codes[i]], 1u);
}

// P(X >= k), X ~ Poisson(lambda) = the regularised lower incomplete gamma function P(k, lambda)
// (what Rcpp::ppois(k - 1, lambda, false) returns, PoissonTargetEstimator.cpp:91): its series for lambda < k + 1,
// else one minus the continued fraction of the upper function (modified Lentz).
inline double poisson_upper_tail(long k, double lambda) {
	if (k <= 0) return 1.0;
	if (!(lambda > 0)) return 0.0;
	const double a = double(k), x = lambda;
	const double lead = std::exp(-x + a * std::log(x) - std::lgamma(a));
	if (x < a + 1.0) {
		double ap = a, del = 1.0 / a, sum = del;
		for (int n = 0; n < 100000; ++n) {
			ap += 1.0;
			del *= x / ap;
			sum += del;
			if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
		}
		const double r = sum * lead;
		return r > 1.0 ? 1.0 : r;
	}
	const double tiny = 1e-300;
	double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
	for (int i = 1; i < 100000; ++i) {
		const double an = -double(i) * (double(i) - a);
		b += 2.0;
		d = an * d + b; if (std::fabs(d) < tiny) d = tiny;
		c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
		d = 1.0 / d;
		const double del = d * c;
		h *= del;
		if (std::fabs(del - 1.0) < 1e-16) break;
	}
	const double q = lead * h;
	return q >= 1.0 ? 0.0 : 1.0 - q;
}

}  // namespace dropest

// UMI codes of the gene-bearing molecules of the filtered cells (CellsDataContainer::umi_distribution, :182-197) -> keys_a[0 .. kept);
// max_size = the largest number of molecules of one gene in one cell.
void dropest_ctx::poisson_local_umis(u32 &kept, u32 &max_size) {
	using namespace dropest;
	const KeyLayout &L = layout;
	kept = max_size = 0;
	if (!n_mol) return;
	std::vector<u32> flags(n_cells, 0);
	for (uint64_t id : filtered_cells()) flags[id] = 1;
	remap.ensure(n_cells);
	HIP_CHECK(hipMemcpyAsync(remap.p, flags.data(), size_t(n_cells) * 4, hipMemcpyHostToDevice, stream));
	keys_a.ensure(n_mol); keys_b.ensure(n_mol); vals_a.ensure(n_mol); vals_b.ensure(n_mol);
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 16, stream));
	hipLaunchKernelGGL(emit_filtered_umis_kernel, dim3(div_up(n_mol, 256)), dim3(256), 0, stream, mol_key.p, n_mol, L.umi_bits,
	                   L.gene_bits, L.gene_none, remap.p, keys_a.p, scalars.p);
	hipLaunchKernelGGL(max_gene_size_kernel, dim3(div_up(n_cg, 256)), dim3(256), 0, stream, cg_key.p, cg_mol_begin.p, n_cg, L.gene_none,
	                   scalars.p + 1);
	HIP_CHECK(hipGetLastError());
	u32 head[2] = {0, 0};
	fetch(head, scalars.p, 8);
	kept = head[0]; max_size = head[1];
}

// The estimator's tables from "how often was each UMI seen" (d_counts: any order, zeros allowed and ignored): classes of equally
// frequent UMIs (probability, multiplicity) and the CollisionsAdjuster table for sizes 1 .. max_size.  The same call on the same
// counts gives the same bits: a sharded run hands every shard the summed histogram.
void dropest_ctx::poisson_build_tables(const u32 *d_counts, u32 n_counts, double total, u32 max_size) {
	using namespace dropest;
	PoissonTables &T = ptab;
	T = PoissonTables{};
	if (!n_counts || !max_size || !(total > 0)) return;
	DevBuf<u64> class_count; DevBuf<u32> class_mult;
	u32 n_classes = 0;
	{
		keys_a.ensure(n_counts); keys_b.ensure(n_counts); vals_a.ensure(n_counts); vals_b.ensure(n_counts);
		scalars.ensure(16);
		HIP_CHECK(hipMemsetAsync(scalars.p, 0, 4, stream));
		hipLaunchKernelGGL(widen_counts_kernel, dim3(div_up(n_counts, 256)), dim3(256), 0, stream, d_counts, n_counts, keys_a.p, scalars.p);
		HIP_CHECK(hipGetLastError());
		u32 max_count = 0;
		fetch(&max_count, scalars.p, 4);
		if (!max_count) return;
		u64 mask = 1;
		while (mask <= max_count) mask <<= 1;
		u64 *k = keys_a.p, *k_alt = keys_b.p;
		u32 *v = vals_a.p, *v_alt = vals_b.p;
		radix_sort(k, v, k_alt, v_alt, n_counts, mask - 1);
		UmiRuns class_policy{};
		class_policy.keys = k;
		n_classes = run_segmented_reduce(*this, "umi_classes", class_policy, n_counts, 8, [&](u32 t) {
			class_count.alloc(t + 1); class_mult.alloc(t + 1);
			zero_async(*this, class_mult.p, size_t(t + 1) * 4);
			class_policy.run_key = class_count.p; class_policy.out[0] = class_mult.p;
		});
	}
	// UMIs nobody saw (a dense histogram lists them) form the class of count 0, the first one: it is not a class of the reference's map
	u64 first_count = 1;
	fetch(&first_count, class_count.p, 8);
	const u32 skip = first_count == 0 ? 1u : 0u;
	n_classes -= skip;
	if (!n_classes) return;
	T.p.alloc(n_classes); T.mult.alloc(n_classes); T.np.alloc(n_classes);
	hipLaunchKernelGGL(classes_to_probs_kernel, dim3(div_up(n_classes, 256)), dim3(256), 0, stream, class_count.p + skip, class_mult.p + skip, n_classes,
	                   total, T.p.p, T.mult.p, T.np.p);
	T.adj.alloc(max_size);
	timed("collisions_table", double(max_size) * n_classes * 24, [&] {
		hipLaunchKernelGGL(collisions_table_kernel, dim3(1), dim3(PM_THREADS), 0, stream, T.p.p, T.mult.p, T.np.p, n_classes, max_size, T.adj.p);
	});
	u64 top = 0;
	fetch(&top, T.adj.p + (max_size - 1), 8);
	// the table diverges when a gene's size approaches the number of distinct UMIs (the reference's fpow then gets a
	// negative exponent and does not terminate)
	if (top >= (1ull << 32)) throw UnsupportedError("collisions adjustment diverged (gene size close to the number of distinct UMIs)");
	T.n_classes = n_classes; T.max_size = max_size; T.ready = true;
}

// expected[p] = SUM of est() over the keys d_keys[off[p] .. off[p + 1]) (one est() per DISTINCT key), added in the order they stand
void dropest_ctx::poisson_expected_from_keys(const u32 *d_off, u32 NP, const u64 *d_keys, u32 NK, double *expected_host) {
	using namespace dropest;
	PoissonTables &T = ptab;
	keys_a.ensure(NK); keys_b.ensure(NK); vals_a.ensure(NK); vals_b.ensure(NK);
	HIP_CHECK(hipMemcpyAsync(keys_a.p, d_keys, size_t(NK) * 8, hipMemcpyDeviceToDevice, stream));
	u64 *k = keys_a.p, *k_alt = keys_b.p;
	u32 *v = vals_a.p, *v_alt = vals_b.p;
	radix_sort(k, v, k_alt, v_alt, NK, ~0ull);
	UmiRuns uniq_policy{};
	uniq_policy.keys = k;
	DevBuf<u64> uniq; DevBuf<u32> uniq_cnt;
	const u32 n_uniq = run_segmented_reduce(*this, "size_pairs", uniq_policy, NK, 8, [&](u32 t) {
		uniq.alloc(t + 1); uniq_cnt.alloc(t + 1);
		zero_async(*this, uniq_cnt.p, size_t(t + 1) * 4);
		uniq_policy.run_key = uniq.p; uniq_policy.out[0] = uniq_cnt.p;
	});
	DevBuf<double> d_est, d_expected;
	d_est.alloc(n_uniq); d_expected.alloc(NP);
	timed("genes_intersection", double(n_uniq) * T.n_classes * 16, [&] {
		hipLaunchKernelGGL(genes_intersection_kernel, dim3(n_uniq), dim3(PM_THREADS), 0, stream, uniq.p, n_uniq, T.p.p, T.mult.p, T.n_classes,
		                   d_est.p);
	});
	hipLaunchKernelGGL(expected_intersection_kernel, dim3(div_up(NP, 256)), dim3(256), 0, stream, d_off, NP, d_keys, uniq.p, n_uniq,
	                   d_est.p, d_expected.p);
	HIP_CHECK(hipGetLastError());
	fetch(expected_host, d_expected.p, size_t(NP) * 8);
}

// Expected intersection sizes of S's pairs on the current device state (single context).
std::vector<double> dropest_ctx::poisson_expected_intersections(const std::vector<u32> &pair_base_cell, const std::vector<u32> &pair_cand_cell) {
	using namespace dropest;
	const u32 NP = u32(pair_base_cell.size());
	std::vector<double> expected(NP, 0.0);
	if (!NP) return expected;
	const KeyLayout &L = layout;
	const u64 gene_mask = L.gene_none;

	// 1. UMI distribution of the filtered cells: sorted codes -> one count per distinct UMI
	u32 kept = 0, max_size = 0;
	poisson_local_umis(kept, max_size);
	if (!kept || !max_size) return expected;
	u64 *k = keys_a.p, *k_alt = keys_b.p;
	u32 *v = vals_a.p, *v_alt = vals_b.p;
	radix_sort(k, v, k_alt, v_alt, kept, L.umi_bits >= 64 ? ~0ull : ((1ull << L.umi_bits) - 1ull));
	UmiRuns runs_policy{};
	runs_policy.keys = k;
	DevBuf<u64> run_key; DevBuf<u32> run_cnt;
	const u32 n_umis = run_segmented_reduce(*this, "umi_runs", runs_policy, kept, 8, [&](u32 total) {
		run_key.alloc(total + 1); run_cnt.alloc(total + 1);
		zero_async(*this, run_cnt.p, size_t(total + 1) * 4);
		runs_policy.run_key = run_key.p; runs_policy.out[0] = run_cnt.p;
	});
	// 2. classes of equally frequent UMIs, adjusted sizes 1..max_size (Tools::CollisionsAdjuster)
	poisson_build_tables(run_cnt.p, n_umis, double(kept), max_size);
	if (!ptab.ready) return expected;

	// 3. genes in common of every pair -> keys of adjusted sizes
	DevBuf<u32> d_pb, d_pc, d_cnt, d_off;
	d_pb.alloc(NP); d_pc.alloc(NP); d_cnt.alloc(NP); d_off.alloc(size_t(NP) + 1);
	HIP_CHECK(hipMemcpyAsync(d_pb.p, pair_base_cell.data(), size_t(NP) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_pc.p, pair_cand_cell.data(), size_t(NP) * 4, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(common_genes_kernel<false>, dim3(div_up(NP, 256)), dim3(256), 0, stream, d_pb.p, d_pc.p, NP, cell_cg_begin.p,
	                   cell_cg_count.p, cg_key.p, cg_mol_begin.p, gene_mask, ptab.adj.p, d_cnt.p, nullptr, nullptr);
	HIP_CHECK(hipGetLastError());
	std::vector<u32> cnt(NP), off(size_t(NP) + 1, 0);
	fetch(cnt.data(), d_cnt.p, size_t(NP) * 4);
	u64 total = 0;
	for (u32 p = 0; p < NP; ++p) { off[p] = u32(total); total += cnt[p]; }
	if (total >= 0xFFFFFFFFull) throw UnsupportedError("too many common genes over the merge candidates");
	off[NP] = u32(total);
	if (!total) return expected;
	const u32 NK = u32(total);
	HIP_CHECK(hipMemcpyAsync(d_off.p, off.data(), (size_t(NP) + 1) * 4, hipMemcpyHostToDevice, stream));
	DevBuf<u64> d_keys; d_keys.alloc(NK);
	hipLaunchKernelGGL(common_genes_kernel<true>, dim3(div_up(NP, 256)), dim3(256), 0, stream, d_pb.p, d_pc.p, NP, cell_cg_begin.p,
	                   cell_cg_count.p, cg_key.p, cg_mol_begin.p, gene_mask, ptab.adj.p, d_cnt.p, d_off.p, d_keys.p);
	HIP_CHECK(hipGetLastError());

	// 4. one est() per distinct key, summed per pair
	poisson_expected_from_keys(d_off.p, NP, d_keys.p, NK, expected.data());
	return expected;
}

// PoissonTargetEstimator::get_best_merge_target (:14-44) for every base of S
void dropest_ctx::decide_poisson_targets(const dropest::MergeUniverse &U, dropest::MergeSearch &S, const std::vector<u32> &inter,
                                         const std::vector<double> &expected, std::vector<long> &targets, std::vector<u32> &target_ridx) {
	using namespace dropest;
	const u32 F = S.F;
	targets.assign(F, -1);
	target_ridx.assign(F, 0xFFFFFFFFu);
	std::vector<double> prob(inter.size());
	for (size_t p = 0; p < inter.size(); ++p) prob[p] = inter[p] == 0 ? 1.0 : poisson_upper_tail(long(inter[p]), expected[p]);   // :69-75, :91
	std::vector<u32> need_order;
	std::vector<double> limit(F);
	for (u32 f = 0; f < F; ++f) {
		if (S.cnt[f] == 0) continue;                                   // no neighbours: -1 (RealBarcodesMergeStrategy.cpp:25-28)
		const bool base_real = S.self_ridx[f] != 0xFFFFFFFFu;
		limit[f] = (base_real ? cfg.max_merge_prob : cfg.max_real_merge_prob) / double(S.cnt[f]);
		const u32 p0 = S.pair_first[f], p1 = S.pair_first[f + 1];
		double min_prob = 2; u32 n_min = 0, min_p = 0;
		for (u32 p = p0; p < p1; ++p) {
			if (prob[p] < min_prob) { min_prob = prob[p]; n_min = 1; min_p = p; }
			else if (prob[p] == min_prob) ++n_min;
		}
		if (min_prob > limit[f]) {
			if (base_real) { targets[f] = long(S.cells[f]); target_ridx[f] = S.self_ridx[f]; }
			continue;
		}
		if (n_min == 1) { targets[f] = long(S.pair_cand[min_p]); target_ridx[f] = S.pair_ridx[min_p]; continue; }
		need_order.push_back(f);   // several neighbours at the minimum: the first in the reference's order wins
	}
	if (need_order.empty()) return;
	const std::vector<std::vector<u32>> orders = replay_candidate_orders(U, S, need_order);
	for (u32 r = 0; r < u32(need_order.size()); ++r) {
		const u32 f = need_order[r];
		std::unordered_map<u32, u32> pair_of;
		for (u32 p = S.pair_first[f]; p < S.pair_first[f + 1]; ++p) pair_of[S.pair_cand[p]] = p;
		double min_prob = 2; long best = -1;
		for (u32 c : orders[r]) {
			const double pr = prob[pair_of.at(c)];
			if (pr < min_prob) { min_prob = pr; best = long(c); }
		}
		if (best < 0) throw DeviceError("internal: candidate replay found no candidate");
		targets[f] = best;
		target_ridx[f] = S.pair_ridx[pair_of.at(u32(best))];
	}
}

// ---- -M across shards (PoissonRealBarcodesMergeStrategy; the phases of merge_shard.h) ------------------------------------------------------------
// The estimator's UMI distribution is the one of ALL filtered cells: every shard counts its own molecules into a dense histogram over the
// UMI field (identical layout everywhere), the driver adds the histograms, and every shard builds the same tables from the sum.
void dropest_ctx::shard_merge_umi_histogram(dropest::DevBuf<u32> &hist, uint64_t &kept, u32 &max_size) {
	using namespace dropest;
	if (!initialized) throw InvalidError("You must initialize container");
	if (layout.umi_bits > 26) throw UnsupportedError("-M across shards needs a UMI field of at most 26 bits (UMIs of 13 bases)");
	HostStage hs(this, "shard_merge:umi_histogram");
	const size_t n = size_t(1) << layout.umi_bits;
	hist.ensure(n);
	HIP_CHECK(hipMemsetAsync(hist.p, 0, n * 4, stream));
	u32 k = 0, ms = 0;
	poisson_local_umis(k, ms);
	if (k) { hipLaunchKernelGGL(umi_histogram_kernel, dim3(div_up(k, 256)), dim3(256), 0, stream, keys_a.p, k, hist.p); HIP_CHECK(hipGetLastError()); }
	HIP_CHECK(stream_wait(stream));
	kept = k; max_size = ms;
}

void dropest_ctx::shard_merge_set_umi_distribution(const u32 *d_counts, uint64_t n_counts, uint64_t kept_total, u32 max_size) {
	if (n_counts > 0xFFFFFFF0ull) throw UnsupportedError("UMI histogram too large");
	HostStage hs(this, "shard_merge:umi_tables");
	poisson_build_tables(d_counts, u32(n_counts), double(kept_total), max_size);
}

// expected[p] of the pairs whose candidate lives here (same arguments as shard_merge_intersect)
void dropest_ctx::shard_merge_expected(uint64_t n_pairs, const uint32_t *cand_local, const uint64_t *base_begin, const uint64_t *base_end,
                                       const uint64_t *d_base_low, double *expected) {
	using namespace dropest;
	if (!initialized) throw InvalidError("You must initialize container");
	for (uint64_t p = 0; p < n_pairs; ++p) expected[p] = 0.0;
	if (n_pairs == 0 || !ptab.ready) return;
	if (n_pairs > 0xFFFFFFF0ull) throw UnsupportedError("too many pairs");
	HostStage hs(this, "shard_merge:expected");
	const u32 NP = u32(n_pairs);
	std::vector<u32> bb(NP), be(NP);
	for (u32 p = 0; p < NP; ++p) {
		if (cand_local[p] >= n_cells) throw RangeError("candidate is not a cell of this shard");
		if (base_end[p] > 0xFFFFFFF0ull || base_begin[p] > base_end[p]) throw RangeError("bad base row range");
		bb[p] = u32(base_begin[p]); be[p] = u32(base_end[p]);
	}
	DevBuf<u32> d_c, d_bb, d_be, d_cnt, d_off;
	d_c.alloc(NP); d_bb.alloc(NP); d_be.alloc(NP); d_cnt.alloc(NP); d_off.alloc(size_t(NP) + 1);
	HIP_CHECK(hipMemcpyAsync(d_c.p, cand_local, size_t(NP) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_bb.p, bb.data(), size_t(NP) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_be.p, be.data(), size_t(NP) * 4, hipMemcpyHostToDevice, stream));
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 4, stream));
	const unsigned long long *low = reinterpret_cast<const unsigned long long *>(d_base_low);
	hipLaunchKernelGGL(common_genes_ext_kernel<false>, dim3(div_up(NP, 256)), dim3(256), 0, stream, d_bb.p, d_be.p, d_c.p, NP, low, layout.umi_bits,
	                   cell_cg_begin.p, cell_cg_count.p, cg_key.p, cg_mol_begin.p, layout.gene_none, ptab.adj.p, ptab.max_size, d_cnt.p, nullptr, nullptr, scalars.p);
	HIP_CHECK(hipGetLastError());
	std::vector<u32> cnt(NP), off(size_t(NP) + 1, 0);
	fetch(cnt.data(), d_cnt.p, size_t(NP) * 4);
	u64 total = 0;
	for (u32 p = 0; p < NP; ++p) { off[p] = u32(total); total += cnt[p]; }
	if (total >= 0xFFFFFFFFull) throw UnsupportedError("too many common genes over the merge candidates");
	off[NP] = u32(total);
	if (!total) return;
	const u32 NK = u32(total);
	HIP_CHECK(hipMemcpyAsync(d_off.p, off.data(), (size_t(NP) + 1) * 4, hipMemcpyHostToDevice, stream));
	DevBuf<u64> d_keys; d_keys.alloc(NK);
	hipLaunchKernelGGL(common_genes_ext_kernel<true>, dim3(div_up(NP, 256)), dim3(256), 0, stream, d_bb.p, d_be.p, d_c.p, NP, low, layout.umi_bits,
	                   cell_cg_begin.p, cell_cg_count.p, cg_key.p, cg_mol_begin.p, layout.gene_none, ptab.adj.p, ptab.max_size, d_cnt.p, d_off.p, d_keys.p, scalars.p);
	HIP_CHECK(hipGetLastError());
	u32 too_large = 0;
	fetch(&too_large, scalars.p, 4);
	if (too_large) throw DeviceError("internal: a gene larger than the adjuster table (the shards did not agree on the largest gene)");
	poisson_expected_from_keys(d_off.p, NP, d_keys.p, NK, expected);
	collect_timings();
}

void dropest_ctx::shard_merge_decide_poisson(const uint32_t *inter, const double *expected, int64_t *target_g) {
	if (!shard) throw InvalidError("dropest_shard_merge_search was not run");
	HostStage hs(this, "shard_merge:decide");
	ShardMerge &M = *shard;
	if (M.S.F == 0) return;
	const size_t np = M.S.pair_base.size();
	std::vector<u32> in(inter, inter + np);
	std::vector<double> ex(expected, expected + np);
	std::vector<long> targets;
	std::vector<u32> tr;
	decide_poisson_targets(M.U, M.S, in, ex, targets, tr);
	for (u32 f = 0; f < M.S.F; ++f) target_g[f] = targets[f];
}

