// simple_merge.h -- "-m" without a barcode whitelist: SimpleMergeStrategy
// (Estimation/Merge/SimpleMergeStrategy.cpp:16-108), included by dropest_amd.hip.
//
// The reference builds an inverted index (UMI, gene) -> set of filtered cells, and for every filtered cell counts,
// per other cell that is at least as large (in genes), the UMI-genes they share; the target is the cell with the
// largest 0.5 * shared * (1/umis(base) + 1/umis(other)) among those within the barcode edit distance, with an EPS band
// in which the larger cell wins.  On the device:
//   umig_keys       molecules of the filtered cells re-keyed (gene | UMI | cell) and radix-sorted: a run = the cells that
//                   share one UMI-gene (the inverted index, as a sorted table)
//   umig_pairs      every ordered pair (base, other != base, size(other) >= size(base)) of a run -> (base << 32 | other);
//                   sorted and run-length encoded this is common_umigs_per_cell for ALL bases at once
//   pair_distance   Levenshtein distance (N wildcards) of the two barcodes of each distinct pair
// The decision is taken on the host.  It is order-free whenever the best admissible fraction is more than 2 EPS clear
// of the others; otherwise the reference's answer depends on the iteration order of two std::unordered containers
// keyed by cell id, and those bases are replayed with the same containers filled in the same order (the members of
// each UMI-gene come from the sorted table, the base's UMIs in (gene index, UMI first occurrence) order).
// No reference test pins this strategy; DESIGN.md §2b says how it is checked.
#pragma once

namespace {

struct SimpleKeyArgs {
	const unsigned long long *mol_key; uint32_t n_mol;
	int umi_bits, gene_bits, cell_bits; unsigned long long gene_none;
	const uint32_t *flag;             // [n_cells] cell takes part (filtered cell)
	unsigned long long *keys; uint32_t *n_valid;
};
__global__ __launch_bounds__(256) void umig_keys_kernel(SimpleKeyArgs a) {
	const uint32_t stride = gridDim.x * 256;
	uint32_t valid = 0;
	const int low_bits = a.umi_bits + a.gene_bits;
	const unsigned long long low_mask = (1ull << low_bits) - 1ull;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.n_mol; i += stride) {
		const unsigned long long k = a.mol_key[i];
		const uint32_t cell = uint32_t(k >> low_bits);
		const bool ok = a.flag[cell] && ((k >> a.umi_bits) & a.gene_none) != a.gene_none;
		a.keys[i] = ok ? (((k & low_mask) << a.cell_bits) | cell) : ~0ull;
		valid += ok;
	}
	valid = uint32_t(dropest::wave_reduce_add_u64(valid));
	if (dropest::lane_id() == 0 && valid) atomicAdd(a.n_valid, valid);
}

struct SimplePairArgs {
	const unsigned long long *keys; uint32_t n;    // sorted (UMI-gene, cell) records
	int cell_bits;
	const uint32_t *cell_size;                     // Cell::size() = number of genes
	const uint32_t *tile_prefix;                   // pass 2
	uint32_t *tile_counts;                         // pass 1
	unsigned long long *pairs;                     // pass 2
};
constexpr int SP_THREADS = 256;
// pairs a record takes part in as the BASE: the other cells of its run that are at least as large
__device__ inline uint32_t simple_run_bounds(const SimplePairArgs &a, uint32_t i, uint32_t &lo, uint32_t &hi) {
	const unsigned long long low = a.keys[i] >> a.cell_bits;
	uint32_t l = 0, h = i;                         // first record of the run
	while (l < h) { const uint32_t m = (l + h) >> 1; if ((a.keys[m] >> a.cell_bits) < low) l = m + 1; else h = m; }
	lo = l;
	l = i + 1; h = a.n;
	while (l < h) { const uint32_t m = (l + h) >> 1; if ((a.keys[m] >> a.cell_bits) <= low) l = m + 1; else h = m; }
	hi = l;
	return hi - lo;
}
template <bool WRITE>
__global__ __launch_bounds__(SP_THREADS) void umig_pairs_kernel(SimplePairArgs a) {
	__shared__ uint32_t scratch[SP_THREADS / 64 + 1];
	const uint32_t i = blockIdx.x * SP_THREADS + threadIdx.x;
	const unsigned long long cmask = (1ull << a.cell_bits) - 1ull;
	uint32_t lo = 0, hi = 0, cnt = 0, x = 0, sx = 0;
	if (i < a.n && simple_run_bounds(a, i, lo, hi) > 1) {
		x = uint32_t(a.keys[i] & cmask); sx = a.cell_size[x];
		for (uint32_t j = lo; j < hi; ++j) cnt += (j != i) && a.cell_size[uint32_t(a.keys[j] & cmask)] >= sx;
	}
	uint32_t total;
	const uint32_t ex = dropest::block_excl_scan_u32<SP_THREADS>(cnt, scratch, total);
	if (!WRITE) { if (threadIdx.x == 0) a.tile_counts[blockIdx.x] = total; return; }
	uint32_t o = a.tile_prefix[blockIdx.x] + ex;
	if (cnt)
		for (uint32_t j = lo; j < hi; ++j) {
			const uint32_t y = uint32_t(a.keys[j] & cmask);
			if (j != i && a.cell_size[y] >= sx) a.pairs[o++] = ((unsigned long long)x << 32) | y;
		}
}

// Tools::edit_distance(base barcode, other barcode) with its default arguments (N wildcards, no band);
// 0xFFFFFFFF when a barcode is escaped or longer than 31 bases (the host computes those)
__global__ __launch_bounds__(256) void pair_distance_kernel(const unsigned long long *__restrict__ pairs, uint32_t n,
                                                            const unsigned long long *__restrict__ cell_cb, uint32_t *__restrict__ ed) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const unsigned long long ca = cell_cb[uint32_t(pairs[i] >> 32)], cb = cell_cb[uint32_t(pairs[i])];
	if ((ca | cb) & dropest::ESCAPE_BIT) { ed[i] = 0xFFFFFFFFu; return; }
	const int la = (dropest::bit_length(ca) - 1) / 2, lb = (dropest::bit_length(cb) - 1) / 2;
	if (la > 31 || lb > 31) { ed[i] = 0xFFFFFFFFu; return; }
	char sa[32], sb[33];
	for (int k = 0; k < la; ++k) sa[k] = "ACGT"[(ca >> (2 * (la - 1 - k))) & 3];
	for (int k = 0; k < lb; ++k) sb[k] = "ACGT"[(cb >> (2 * (lb - 1 - k))) & 3];
	sb[lb] = 0;
	uint32_t peq[5], wild;
	dropest::wl_build_peq(sa, la, peq, wild);
	ed[i] = dropest::wl_edit_distance(peq, wild, la, sb);
}

// members of the UMI-genes of the replayed bases: query = (gene | UMI) field of one molecule of a base
__global__ __launch_bounds__(256) void umig_members_kernel(const unsigned long long *__restrict__ keys, uint32_t n, int cell_bits,
                                                           const unsigned long long *__restrict__ query, uint32_t nq,
                                                           const uint32_t *__restrict__ out_off, uint32_t *__restrict__ count,
                                                           uint32_t *__restrict__ out_cells) {
	const uint32_t q = blockIdx.x * 256 + threadIdx.x;
	if (q >= nq) return;
	const unsigned long long low = query[q];
	uint32_t l = 0, h = n;
	while (l < h) { const uint32_t m = (l + h) >> 1; if ((keys[m] >> cell_bits) < low) l = m + 1; else h = m; }
	uint32_t e = l;
	while (e < n && (keys[e] >> cell_bits) == low) ++e;
	if (!out_cells) { count[q] = e - l; return; }
	const unsigned long long cmask = (1ull << cell_bits) - 1ull;
	for (uint32_t j = l; j < e; ++j) out_cells[out_off[q] + (j - l)] = uint32_t(keys[j] & cmask);
}

// Tools::edit_distance with default arguments on the host (escaped barcodes)
unsigned plain_edit_distance(const std::string &a, const std::string &b) {
	std::vector<unsigned> col(a.size() + 1);
	for (size_t i = 0; i <= a.size(); ++i) col[i] = unsigned(i);
	for (size_t j = 1; j <= b.size(); ++j) {
		unsigned diag = col[0];
		col[0] = unsigned(j);
		for (size_t i = 1; i <= a.size(); ++i) {
			const unsigned up = col[i];
			const bool match = a[i - 1] == b[j - 1] || a[i - 1] == 'N' || b[j - 1] == 'N';
			col[i] = std::min(std::min(col[i] + 1, col[i - 1] + 1), diag + unsigned(!match));
			diag = up;
		}
	}
	return col[a.size()];
}

// ---- decisions shared by one context and the shards of a sharded run (shard_merge_free.h) --------------------------------------------------
// What the decisions need to know about a cell.  `id` is whatever the pair table holds: the cell id of a context, the place in the
// global cell list of a sharded run.
struct SimpleCells {
	std::function<size_t(u32)> umis, genes;           // Stats::TOTAL_UMIS, Cell::size()
	std::function<std::string(u32)> barcode;          // text (edit distance of escaped codes; replay)
	std::function<u32(u32)> slot;                     // base -> index into the caller's per-base arrays
	std::function<u32(u32)> filtered_pos;             // place in the filtered order (a UMI-gene's cells are emplaced in that order)
	std::function<size_t(u32)> container_id;          // index of the cell in the reference's container: the key of its unordered containers
	std::function<u32(size_t)> from_container_id;
};

const double SIMPLE_EPS = 0.00001;                    // SimpleMergeStrategy::EPS

int simple_pair_distance(const SimplePairs &P, size_t p, const SimpleCells &C) {
	if (P.ed[p] != 0xFFFFFFFFu) return int(P.ed[p]);
	return int(plain_edit_distance(C.barcode(u32(P.key[p] >> 32)), C.barcode(u32(P.key[p]))));
}

// SimpleMergeStrategy::get_merge_target (:47-86) / PoissonSimpleMergeStrategy::get_merge_target (PoissonSimpleMergeStrategy.cpp:15-43)
// for the bases whose answer does not depend on the iteration order of common_umigs_per_cell; the others land in `replay`.
// target[slot] is preset to the base itself.  expected(bases, others) = the estimator's expected intersections (-M only).
void simple_decide(const dropest_cfg &cfg, bool poisson, SimplePairs &P, const SimpleCells &C,
                   const std::function<std::vector<double>(const std::vector<u32> &, const std::vector<u32> &)> &expected,
                   std::vector<u32> &target, std::vector<u32> &nb_count, std::vector<u32> &replay) {
	const double EPS = SIMPLE_EPS;
	const int max_ed = cfg.max_cb_merge_edit_distance;
	// PoissonSimple: the neighbours are the cells with common UMI-genes within the edit distance (<=, not < as below); the shared
	// count IS the intersection size
	if (poisson) {
		std::vector<u32> pb, pc; std::vector<size_t> at;
		for (size_t p = 0; p < P.key.size(); ++p) {
			const u32 base = u32(P.key[p] >> 32), other = u32(P.key[p]);
			if (simple_pair_distance(P, p, C) > max_ed) continue;
			pb.push_back(base); pc.push_back(other); at.push_back(p);
			++nb_count[C.slot(base)];
		}
		const std::vector<double> ex = expected(pb, pc);
		P.prob.assign(P.key.size(), 2.0);
		for (size_t i = 0; i < at.size(); ++i) P.prob[at[i]] = poisson_upper_tail(long(P.cnt[at[i]]), ex[i]);
	}
	for (size_t p = 0; poisson && p < P.key.size();) {
		const u32 base = u32(P.key[p] >> 32);
		size_t e = p;
		while (e < P.key.size() && u32(P.key[e] >> 32) == base) ++e;
		const u32 f = C.slot(base);
		if (nb_count[f]) {
			// the base is never its own neighbour: PoissonTargetEstimator.cpp:17-22 takes max_real_cb_merge_prob / |neighbours|
			const double limit = cfg.max_real_merge_prob / double(nb_count[f]);
			double min_prob = 2; size_t n_min = 0, min_p = p;
			for (size_t q = p; q < e; ++q) {
				if (P.prob[q] < min_prob) { min_prob = P.prob[q]; n_min = 1; min_p = q; }
				else if (P.prob[q] == min_prob && min_prob < 2) ++n_min;
			}
			if (min_prob > limit) { /* -1 -> the base itself (:39-42) */ }
			else if (n_min == 1) target[f] = u32(P.key[min_p]);
			else replay.push_back(base);                                     // first of the minima in the map's iteration order
		}
		p = e;
	}
	for (size_t p = 0; !poisson && p < P.key.size();) {
		const u32 base = u32(P.key[p] >> 32);
		size_t e = p;
		while (e < P.key.size() && u32(P.key[e] >> 32) == base) ++e;
		const u32 f = C.slot(base);
		double best = -1; size_t best_p = p;
		std::vector<double> frac(e - p, -1.0);                               // admissible candidates only
		for (size_t q = p; q < e; ++q) {
			const u32 other = u32(P.key[q]);
			if (simple_pair_distance(P, q, C) >= max_ed) continue;
			frac[q - p] = 0.5 * P.cnt[q] * (1. / C.umis(base) + 1. / C.umis(other));
			if (frac[q - p] > best) { best = frac[q - p]; best_p = q; }
		}
		size_t near = 0;
		for (double x : frac) near += x >= 0 && best - x <= 2 * EPS;
		if (best < 0) { /* nobody within the edit distance: top stays -1, returns the base */ }
		else if (near == 1) { if (!(best < cfg.min_merge_fraction)) target[f] = u32(P.key[best_p]); }
		else replay.push_back(base);
		p = e;
	}
}

// One base with a near-tie, replayed with the reference's containers filled in the reference's order.  q_in_order = the base's
// gene-bearing molecules in (gene index, UMI index) order -- the reference's nested std::map walk -- as indices into q_off / q_cnt;
// members[q_off[q] .. + q_cnt[q]) = the cells of that molecule's UMI-gene.  Returns the target (the base itself: none).
u32 simple_replay_base(const dropest_cfg &cfg, bool poisson, u32 base, const std::vector<u32> &q_in_order, const std::vector<u32> &members,
                       const std::vector<u32> &q_off, const std::vector<u32> &q_cnt, const SimpleCells &C, const SimplePairs &P, u32 nb_count) {
	const double EPS = SIMPLE_EPS;
	const int max_ed = cfg.max_cb_merge_edit_distance;
	const size_t base_size = C.genes(base), base_cid = C.container_id(base);
	std::unordered_map<size_t, size_t> common;                       // u_u_hash_t common_umigs_per_cell (:19)
	for (u32 q : q_in_order) {
		// sul_set_t of this UMI-gene: cells emplaced in filtered order by init() (:88-102)
		std::vector<u32> mem(members.begin() + q_off[q], members.begin() + q_off[q] + q_cnt[q]);
		std::sort(mem.begin(), mem.end(), [&](u32 x, u32 y) { return C.filtered_pos(x) < C.filtered_pos(y); });
		std::unordered_set<size_t> set;
		for (u32 c : mem) set.emplace(C.container_id(c));
		for (size_t other : set) {
			if (other == base_cid) continue;
			if (C.genes(C.from_container_id(other)) >= base_size) common[other]++;
		}
	}
	if (poisson) {   // neighbours in the map's order; probabilities from the pair table (sorted by (base, other))
		double min_prob = 2; long best = -1;
		for (auto const &c : common) {
			const u32 other = C.from_container_id(c.first);
			const u64 key = (u64(base) << 32) | u64(other);
			const size_t q = size_t(std::lower_bound(P.key.begin(), P.key.end(), key) - P.key.begin());
			if (q >= P.key.size() || P.key[q] != key) throw DeviceError("internal: replayed neighbour without a pair record");
			if (P.prob[q] >= 2) continue;                                // beyond the edit distance
			if (P.prob[q] < min_prob) { min_prob = P.prob[q]; best = long(other); }
		}
		const double limit = cfg.max_real_merge_prob / double(nb_count);
		return (best < 0 || min_prob > limit) ? base : u32(best);
	}
	long top = -1, top_genes = -1;
	double top_frac = -1;
	for (auto const &c : common) {
		const u32 ind = C.from_container_id(c.first);
		const double fr = 0.5 * c.second * (1. / C.umis(base) + 1. / C.umis(ind));
		if (fr - top_frac > EPS || (std::abs(fr - top_frac) < EPS && long(C.genes(ind)) > top_genes)) {
			const int ed = int(plain_edit_distance(C.barcode(base), C.barcode(ind)));
			if (ed >= max_ed) continue;
			top = long(ind); top_frac = fr; top_genes = long(C.genes(ind));
		}
	}
	return (top_frac < cfg.min_merge_fraction || top < 0) ? base : u32(top);
}

}  // namespace

// common_umigs_per_cell of every base from a sorted (UMI-gene, cell) table: ordered pairs per run, sorted, run-length encoded; distances
// of the distinct pairs.  cell_size / cell_code are indexed like the cell field of the table.  With `d_run_key` the table stays on the
// device (a sharded run routes the partial counts to the owners of the bases) and P is left empty.
void dropest_ctx::simple_pair_table(const u64 *sorted, u32 n_valid, int cell_bits, const u32 *d_cell_size, const u64 *d_cell_code, SimplePairs &P,
                                    dropest::DevBuf<u64> *d_run_key, dropest::DevBuf<u32> *d_run_cnt, u32 *n_runs) {
	if (n_runs) *n_runs = 0;
	if (!n_valid) return;
	const u32 tiles = div_up(n_valid, SP_THREADS);
	tile_counts.ensure(tiles); tile_prefix.ensure(tiles);
	scalars.ensure(16);
	SimplePairArgs pa{sorted, n_valid, cell_bits, d_cell_size, tile_prefix.p, tile_counts.p, nullptr};
	timed("simple:umig_pairs", double(n_valid) * 16, [&] {
		hipLaunchKernelGGL(umig_pairs_kernel<false>, dim3(tiles), dim3(SP_THREADS), 0, stream, pa);
		scan_counts(tile_counts.p, tile_prefix.p, tiles, scalars.p);
	});
	u32 NP = 0;
	fetch(&NP, scalars.p, 4);
	// (scan_small sums in 32 bits: refuse inputs whose pair count could have wrapped)
	if (NP > 0xF0000000u) throw UnsupportedError("more than 2^32 shared UMI-gene pairs");
	if (!NP) return;
	DevBuf<u64> p_a, p_b; DevBuf<u32> dummy_a, dummy_b;
	p_a.alloc(NP); p_b.alloc(NP); dummy_a.alloc(1); dummy_b.alloc(1);
	pa.pairs = p_a.p;
	timed("simple:umig_pairs", double(NP) * 8, [&] {
		hipLaunchKernelGGL(umig_pairs_kernel<true>, dim3(tiles), dim3(SP_THREADS), 0, stream, pa);
	});
	u64 *pk = p_a.p, *pk_alt = p_b.p;
	u32 *pv = dummy_a.p, *pv_alt = dummy_b.p;
	radix_sort(pk, pv, pk_alt, pv_alt, NP, (((1ull << cell_bits) - 1ull) << 32) | ((1ull << cell_bits) - 1ull), 0);
	UmiRuns rp{};
	rp.keys = pk;
	DevBuf<u64> run_key_own; DevBuf<u32> run_cnt_own, run_ed;
	DevBuf<u64> &run_key = d_run_key ? *d_run_key : run_key_own;
	DevBuf<u32> &run_cnt = d_run_cnt ? *d_run_cnt : run_cnt_own;
	const u32 runs = run_segmented_reduce(*this, "simple:pair_runs", rp, NP, 8, [&](u32 total) {
		run_key.alloc(total + 1); run_cnt.alloc(total + 1);
		zero_async(*this, run_cnt.p, size_t(total + 1) * 4);
		rp.run_key = run_key.p; rp.out[0] = run_cnt.p;
	});
	if (n_runs) *n_runs = runs;
	if (d_run_key) { HIP_CHECK(stream_wait(stream)); return; }   // (p_a / p_b die with this scope)
	simple_pairs_to_host(run_key.p, run_cnt.p, runs, d_cell_code, P);
}

// distances of the distinct pairs on the device, then everything to the host
void dropest_ctx::simple_pairs_to_host(const u64 *d_run_key, const u32 *d_run_cnt, u32 runs, const u64 *d_cell_code, SimplePairs &P) {
	P.key.resize(runs); P.cnt.resize(runs); P.ed.resize(runs);
	if (!runs) return;
	DevBuf<u32> run_ed;
	run_ed.alloc(runs);
	timed("simple:pair_distance", double(runs) * 28, [&] {
		hipLaunchKernelGGL(pair_distance_kernel, dim3(div_up(runs, 256)), dim3(256), 0, stream, reinterpret_cast<const unsigned long long *>(d_run_key), runs,
		                   reinterpret_cast<const unsigned long long *>(d_cell_code), run_ed.p);
	});
	fetch(P.key.data(), d_run_key, size_t(runs) * 8);
	fetch(P.cnt.data(), d_run_cnt, size_t(runs) * 4);
	fetch(P.ed.data(), run_ed.p, size_t(runs) * 4);
}

// The gene-bearing molecules of the replayed bases (local cells) in the reference's walk order: query[] = their (gene | UMI) fields,
// in_order[r] = indices into query of base r's molecules in (gene index, UMI index) order.  The UMI index is the order of first
// occurrence in the WHOLE stream: a sharded run (globalize) turns the table of first positions into global ranks -- a collective, so
// every shard calls this, with an empty list when it has nothing to replay.
void dropest_ctx::simple_replay_local(const std::vector<u32> &bases, SimpleReplayInput &R, bool globalize) {
	if (layout.umi_bits > 28) throw UnsupportedError("tie replay of the simple merge needs a UMI field of at most 28 bits");
	const size_t table = size_t(1) << layout.umi_bits;
	umi_first.ensure(table);
	HIP_CHECK(hipMemsetAsync(umi_first.p, 0xFF, table * 4, stream));
	const u32 n = u32(n_reads);
	need_columns();   // (a sharded run's reads may still be packed records)
	if (n) timed("umi_first_table", double(n) * 12, [&] {
		hipLaunchKernelGGL(umi_first_table_kernel, dim3(std::min<u32>(div_up(n, 256), 8192u)), dim3(256), 0, stream, umi_key_column(), d_gene, n, layout, umi_first.p);
	});
	if (globalize) {
		if (!hooks || !hooks->globalize_umi_first) throw InvalidError("internal: a sharded replay without the shard hooks");
		hooks->globalize_umi_first(umi_first.p, table);
	}
	R.query.clear(); R.in_order.assign(bases.size(), {});
	if (bases.empty()) return;
	// molecules of the replayed bases: (cell, gene) rows of each base are contiguous
	std::vector<u32> groups;                                             // cg rows of all replayed bases
	std::vector<u32> cgb(bases.size()), cgc(bases.size());
	{
		DevBuf<u32> d_ids, d_b, d_c;
		d_ids.alloc(bases.size()); d_b.alloc(bases.size()); d_c.alloc(bases.size());
		HIP_CHECK(hipMemcpyAsync(d_ids.p, bases.data(), bases.size() * 4, hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(gather_u32_kernel, dim3(div_up(u32(bases.size()), 256)), dim3(256), 0, stream, cell_cg_begin.p, d_ids.p, u32(bases.size()), d_b.p);
		hipLaunchKernelGGL(gather_u32_kernel, dim3(div_up(u32(bases.size()), 256)), dim3(256), 0, stream, cell_cg_count.p, d_ids.p, u32(bases.size()), d_c.p);
		HIP_CHECK(hipGetLastError());
		fetch(cgb.data(), d_b.p, bases.size() * 4);
		fetch(cgc.data(), d_c.p, bases.size() * 4);
	}
	std::vector<size_t> first_group(bases.size() + 1, 0);
	for (size_t r = 0; r < bases.size(); ++r) {
		for (u32 j = 0; j < cgc[r]; ++j) groups.push_back(cgb[r] + j);
		first_group[r + 1] = groups.size();
	}
	GatheredGroups GG;
	umi_gather_groups(groups, GG, umi_first.p);
	// queries: the (gene | UMI) fields of those molecules (gene-less rows are skipped below)
	const u64 low_mask = (1ull << (layout.umi_bits + layout.gene_bits)) - 1ull;
	R.query.resize(GG.hk.size());
	for (size_t i = 0; i < R.query.size(); ++i) R.query[i] = GG.hk[i] & low_mask;
	for (size_t r = 0; r < bases.size(); ++r) {
		// molecules of the base in (gene index, UMI index) order = the reference's nested std::map walk
		struct Q { u64 gene; u32 first; u32 q; };
		std::vector<Q> qs;
		for (size_t gi = first_group[r]; gi < first_group[r + 1]; ++gi)
			for (u32 t = 0; t < GG.size[gi]; ++t) {
				const u32 q = GG.off[gi] + t;
				const u64 gene = (GG.hk[q] >> layout.umi_bits) & layout.gene_none;
				if (gene == layout.gene_none) continue;
				qs.push_back(Q{gene, GG.hfirst[q], q});
			}
		std::sort(qs.begin(), qs.end(), [](const Q &x, const Q &y) { return x.gene != y.gene ? x.gene < y.gene : x.first < y.first; });
		R.in_order[r].reserve(qs.size());
		for (const Q &q : qs) R.in_order[r].push_back(q.q);
	}
}


void dropest_ctx::run_cb_merge_simple() {
	HostStage hs(this, "cb_merge");
	const std::vector<uint64_t> &order = filtered_cells();
	std::vector<u32> cells(order.begin(), order.end());
	const std::vector<u32> ridx = filtered_ridx;
	const u32 F = u32(cells.size()), nR = u32(real.size());
	clear_strategy_pairs();
	if (F == 0 || n_mol == 0) return;
	const int low_bits = layout.umi_bits + layout.gene_bits;

	// 1. the inverted index as a sorted table of (UMI-gene, cell) records of the filtered cells
	std::vector<u32> flags(n_cells, 0), rank_of(n_cells, 0xFFFFFFFFu);
	for (u32 f = 0; f < F; ++f) { flags[cells[f]] = 1; rank_of[cells[f]] = f; }
	remap.ensure(n_cells);
	HIP_CHECK(hipMemcpyAsync(remap.p, flags.data(), size_t(n_cells) * 4, hipMemcpyHostToDevice, stream));
	keys_a.ensure(n_mol); keys_b.ensure(n_mol); vals_a.ensure(n_mol); vals_b.ensure(n_mol);
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 16, stream));
	SimpleKeyArgs ka{mol_key.p, n_mol, layout.umi_bits, layout.gene_bits, layout.cell_bits, layout.gene_none, remap.p, keys_a.p, scalars.p};
	timed("simple:umig_keys", double(n_mol) * 16, [&] {
		hipLaunchKernelGGL(umig_keys_kernel, dim3(std::min<u32>(div_up(n_mol, 256), 4096u)), dim3(256), 0, stream, ka);
	});
	u32 n_valid = 0;
	fetch(&n_valid, scalars.p, 4);
	u64 *k = keys_a.p, *k_alt = keys_b.p;
	u32 *v = vals_a.p, *v_alt = vals_b.p;
	const int key_bits = low_bits + layout.cell_bits;
	radix_sort(k, v, k_alt, v_alt, n_mol, key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull) | (1ull << 63), 0);   // sentinels (all ones) sort last
	const u64 *sorted = k;
	DevBuf<u64> sorted_keep;   // -M: the estimator below reuses the sort buffers; the tie replay still needs this table
	if (cfg.merge_kind == DROPEST_MERGE_POISSON_SIMPLE && n_valid) {
		sorted_keep.alloc(n_valid);
		HIP_CHECK(hipMemcpyAsync(sorted_keep.p, k, size_t(n_valid) * 8, hipMemcpyDeviceToDevice, stream));
	}

	// 2. shared UMI-genes: ordered pairs per run, sorted, run-length encoded
	SimplePairs P;
	simple_pair_table(sorted, n_valid, layout.cell_bits, cell_n_genes.p, cell_cb.p, P, nullptr, nullptr);

	// 3. decisions for the bases whose answer does not depend on the iteration order of common_umigs_per_cell
	SimpleCells C;
	C.umis = [&](u32 cell) { return size_t(real[real_at(cell)].row.total_umis); };
	C.genes = [&](u32 cell) { return size_t(real[real_at(cell)].row.n_genes); };
	C.barcode = [&](u32 cell) { return barcode_of(real[real_at(cell)]); };
	C.slot = [&](u32 cell) { return rank_of[cell]; };
	C.filtered_pos = [&](u32 cell) { return rank_of[cell]; };
	C.container_id = [](u32 cell) { return size_t(cell); };
	C.from_container_id = [](size_t id) { return u32(id); };
	const bool poisson = cfg.merge_kind == DROPEST_MERGE_POISSON_SIMPLE;
	std::vector<u32> tgt_id(cells), nb_count(F, 0), replay;
	simple_decide(cfg, poisson, P, C, [&](const std::vector<u32> &pb, const std::vector<u32> &pc) {
		std::vector<double> ex = poisson_expected_intersections(pb, pc);
		if (sorted_keep.p) sorted = sorted_keep.p;
		return ex;
	}, tgt_id, nb_count, replay);

	// 4. replay of the bases with near-ties: same containers, same insertion order as the reference
	if (!replay.empty()) {
		HostStage hs2(this, "cb_merge:replay");
		SimpleReplayInput R;
		simple_replay_local(replay, R, false);
		// members of every queried UMI-gene
		const u32 nq = u32(R.query.size());
		DevBuf<u64> d_q; DevBuf<u32> d_cnt, d_off, d_cells;
		d_q.alloc(nq); d_cnt.alloc(nq); d_off.alloc(nq);
		HIP_CHECK(hipMemcpyAsync(d_q.p, R.query.data(), size_t(nq) * 8, hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(umig_members_kernel, dim3(div_up(nq, 256)), dim3(256), 0, stream, sorted, n_valid, layout.cell_bits, d_q.p, nq,
		                   static_cast<const u32 *>(nullptr), d_cnt.p, static_cast<u32 *>(nullptr));
		HIP_CHECK(hipGetLastError());
		std::vector<u32> q_cnt(nq), q_off(nq);
		fetch(q_cnt.data(), d_cnt.p, size_t(nq) * 4);
		uint64_t total = 0;
		for (u32 i = 0; i < nq; ++i) { q_off[i] = u32(total); total += q_cnt[i]; }
		if (total > 0xFFFFFFF0ull) throw UnsupportedError("too many UMI-gene members in the tie replay");
		d_cells.alloc(total);
		HIP_CHECK(hipMemcpyAsync(d_off.p, q_off.data(), size_t(nq) * 4, hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(umig_members_kernel, dim3(div_up(nq, 256)), dim3(256), 0, stream, sorted, n_valid, layout.cell_bits, d_q.p, nq,
		                   d_off.p, d_cnt.p, d_cells.p);
		HIP_CHECK(hipGetLastError());
		std::vector<u32> members(total);
		fetch(members.data(), d_cells.p, size_t(total) * 4);
		for (size_t r = 0; r < replay.size(); ++r)
			tgt_id[rank_of[replay[r]]] = simple_replay_base(cfg, poisson, replay[r], R.in_order[r], members, q_off, q_cnt, C, P, nb_count[rank_of[replay[r]]]);
	}
	std::vector<int64_t> target(F);
	for (u32 f = 0; f < F; ++f) target[f] = int64_t(real_at(tgt_id[f]));

	// 5. MergeStrategyBase::merge_inited second loop on the same flat arrays as the whitelist merge
	HostStage hs3(this, "cb_merge:apply");
	std::vector<int32_t> reads(nR), umis(nR);
	for (u32 i = 0; i < nR; ++i) { reads[i] = real[i].row.total_reads; umis[i] = real[i].row.total_umis; }
	std::vector<u32> cur(nR);
	std::vector<uint8_t> excl(nR);
	std::vector<u32> rank(nR);
	const bool any_merge = apply_merge_order(nR, F, ridx.data(), target.data(), reads.data(), umis.data(), cur.data(), excl.data(), rank.data());
	if (have_qual) {   // only the quality sums need the merge order (quality.h)
		merge_rank.assign(n_cells, 0);
		for (u32 i = 0; i < nR; ++i) merge_rank[real[i].id] = rank[i];
	}
	real_pristine = false;
	for (u32 i = 0; i < nR; ++i) {
		real[i].row.total_reads = reads[i]; real[i].row.total_umis = umis[i];
		if (cur[i] != i) { real[i].merged = true; merge_pairs.emplace_back(real[i].id, real[cur[i]].id); }
	}
	if (any_merge) reaggregate_after_merge();
}
