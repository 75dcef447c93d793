// merge_host.h -- host orchestration of the whitelist CB merge (included by dropest_amd.hip).
//
// MergeStrategyAbstract::merge -> MergeStrategyBase::merge_inited (Estimation/Merge/MergeStrategyBase.cpp:11-57):
//   pass 1 (read-only, on the device): a target for every filtered cell on the UNMERGED state
//   pass 2 (sequential, here): smallest-first application with exclusion, chasing of re-targeted cells
//          (reassign, :64-82), Stats::merge bookkeeping (Stats.cpp:29-43)
//   then the molecule table is re-keyed and re-reduced on the device (= the unions done by Gene::merge).
#pragma once
#include <unordered_set>
#include <thread>

namespace {

// Tools::edit_distance (Tools/UtilFunctions.cpp:32-65): banded dynamic programme with N wildcards, restated with its
// band handling (cells outside the band keep what earlier columns left there; the lower band edge is set to the
// column number) and its early exit, because values at and above max_ed decide which UMIs are skipped.
unsigned banded_edit_distance(const std::string &s1, const std::string &s2, unsigned max_ed) {
	const int n1 = int(s1.size()), n2 = int(s2.size());
	std::vector<int> column(size_t(n1) + 1);
	for (int i = 0; i <= n1; ++i) column[size_t(i)] = i;
	for (int j = 1; j <= n2; ++j) {
		const int lower = std::max(0, j - int(max_ed)), upper = std::min(n1, j + int(max_ed));
		int lastdiag = column[size_t(lower)];
		column[size_t(lower)] = j;
		int min_ed = j;
		for (int i = lower + 1; i <= upper; ++i) {
			const int olddiag = column[size_t(i)];
			const bool match = s1[size_t(i - 1)] == s2[size_t(j - 1)] || s1[size_t(i - 1)] == 'N' || s2[size_t(j - 1)] == 'N';
			const int v = std::min(std::min(column[size_t(i)] + 1, column[size_t(i - 1)] + 1), lastdiag + int(!match));
			min_ed = std::min(min_ed, v + std::abs(i - j));
			column[size_t(i)] = v;
			lastdiag = olddiag;
		}
		if (min_ed > int(max_ed)) return unsigned(min_ed);
	}
	return unsigned(column[size_t(n1)]);
}

struct PartDistH { size_t index; long value; };            // Tools/IndexedValue.h
struct ComboH { size_t idx[dropest::WL_MAX_PARTS]; unsigned ed; };   // BarcodesParser::BarcodesDistance

// BarcodesParser::push_remaining_dists (BarcodesParser.cpp:52-74): nested walk of the per-part lists in part order;
// the first entry that takes the running distance beyond the limit ends the walk of that list (the lists are sorted)
void push_remaining_dists(const std::vector<std::vector<PartDistH>> &d, size_t part, unsigned ed, ComboH &cur, std::vector<ComboH> &res) {
	if (part == d.size()) { cur.ed = ed; res.push_back(cur); return; }
	for (const PartDistH &x : d[part]) {
		const unsigned cur_ed = ed + unsigned(x.value);
		if (cur_ed > unsigned(dropest::WL_MAX_DIST)) return;
		cur.idx[part] = x.index;
		push_remaining_dists(d, part + 1, cur_ed, cur, res);
	}
}

// Reference ORDER of the candidate list for one cell, rebuilt from the device-computed per-part distances with
// the same std::sort calls on the same sequences (BarcodesParser.cpp:21-74, RealBarcodesMergeStrategy.cpp:63-109).
// Only needed when the arg-max of the merge fraction is tied (the reference's result then depends on this order).
std::vector<u32> reference_candidate_order(const dropest::Whitelist &wl, const uint8_t *dist,
                                           const std::unordered_map<u64, u32> &qualifying_by_code, bool poisson) {
	const size_t P = wl.parts.size();
	std::vector<std::vector<PartDistH>> d(P);
	size_t off = 0;
	for (size_t p = 0; p < P; ++p) {
		for (size_t i = 0; i < wl.parts[p].size(); ++i) d[p].push_back(PartDistH{i, long(dist[off + i])});
		std::sort(d[p].begin(), d[p].end(), [](const PartDistH &x, const PartDistH &y) { return x.value < y.value; });
		off += wl.parts[p].size();
	}
	std::vector<ComboH> combos;
	ComboH cur{};
	push_remaining_dists(d, 0, 0, cur, combos);
	std::vector<u32> out;
	if (combos.empty()) return out;
	std::sort(combos.begin(), combos.end(), [](const ComboH &x, const ComboH &y) { return x.ed < y.ed; });
	unsigned max_dist = combos.front().ed;
	if (poisson) max_dist = max_dist == 0 ? 2 : max_dist + 1;   // PoissonRealBarcodesMergeStrategy::get_max_merge_dist (:20-23)
	for (const ComboH &c : combos) {
		if (c.ed > max_dist && !out.empty()) break;
		std::string text;
		for (size_t p = 0; p < P; ++p) text += wl.parts[p][c.idx[p]];
		u64 code = 0;
		if (dropest::encode_code(text, code)) {
			auto it = qualifying_by_code.find(code);
			if (it != qualifying_by_code.end()) out.push_back(it->second);
		}
		max_dist = std::max(max_dist, c.ed);
	}
	return out;
}

__global__ __launch_bounds__(256) void pair_ranges_kernel(const uint32_t *__restrict__ base_cell, const uint32_t *__restrict__ cand_cell,
                                                          uint32_t n, const uint32_t *__restrict__ cell_cg_begin,
                                                          const uint32_t *__restrict__ cell_cg_count,
                                                          const uint32_t *__restrict__ cg_mol_begin, dropest::PairRange *out) {
	uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const uint32_t b = base_cell[i], c = cand_cell[i];
	dropest::PairRange r;
	r.base_begin = cg_mol_begin[cell_cg_begin[b]]; r.base_end = cg_mol_begin[cell_cg_begin[b] + cell_cg_count[b]];
	r.cand_begin = cg_mol_begin[cell_cg_begin[c]]; r.cand_end = cg_mol_begin[cell_cg_begin[c] + cell_cg_count[c]];
	out[i] = r;
}

}  // namespace

void dropest_ctx::upload_whitelist() {
	if (!wl.loaded) {
		if (barcodes_file.empty()) throw InvalidError("merge_kind = REAL_BARCODES needs barcodes_file");
		wl.load(cfg.barcodes_kind, barcodes_file);
	}
	for (size_t p = 0; p < wl.parts.size(); ++p) {
		if (d_wl[p].p) continue;
		std::vector<WlEntry> h(wl.parts[size_t(p)].size());
		for (size_t i = 0; i < h.size(); ++i) {
			std::memset(h[i].seq, 0, sizeof(h[i].seq));
			std::memcpy(h[i].seq, wl.parts[size_t(p)][i].data(), wl.parts[size_t(p)][i].size());
		}
		d_wl[p].alloc(h.size()); d_wl[p].mark_persistent();
		HIP_CHECK(hipMemcpy(d_wl[p].p, h.data(), h.size() * sizeof(WlEntry), hipMemcpyHostToDevice));
		// packed copies for the branch-free distance loop (entries of up to 29 clean bases; the others keep the string path)
		std::vector<u64> codes(h.size(), ~0ull);
		for (size_t i = 0; i < h.size(); ++i) {
			const std::string &e = wl.parts[size_t(p)][i];
			if (e.size() > 29) continue;
			u64 c = 0; bool clean = true;
			for (char ch : e) {
				if (ch == 'A') c = (c << 2); else if (ch == 'C') c = (c << 2) | 1; else if (ch == 'G') c = (c << 2) | 2; else if (ch == 'T') c = (c << 2) | 3;
				else { clean = false; break; }
			}
			if (clean) codes[i] = (u64(e.size()) << 58) | c;
		}
		d_wl_code[p].alloc(codes.size()); d_wl_code[p].mark_persistent();
		HIP_CHECK(hipMemcpy(d_wl_code[p].p, codes.data(), codes.size() * 8, hipMemcpyHostToDevice));
	}
	// neighbour tables (k_merge.h), once per whitelist: every part of one length of at most WL_TAB_MAX_LEN clean bases
	if (!wl_tab_ok && !d_wl_tab[0].p && !getenv("DROPEST_WL_NO_TABLE")) {
		bool ok = wl.parts.size() <= size_t(WL_MAX_PARTS);
		for (auto const &part : wl.parts) {
			const size_t L = part[0].size();
			ok = ok && L >= 1 && L <= size_t(WL_TAB_MAX_LEN);
			for (auto const &e : part) ok = ok && e.size() == L && e.find_first_not_of("ACGT") == std::string::npos;
		}
		if (ok) {
			for (size_t p = 0; p < wl.parts.size(); ++p) {
				const int L = int(wl.parts[p][0].size());
				const u32 n_values = 1u << (2 * L);
				d_wl_tab[p].alloc(n_values); d_wl_tab[p].mark_persistent();
				timed("wl_table_build", double(n_values) * wl.parts[p].size() * 8, [&] {
					hipLaunchKernelGGL(wl_table_build_kernel, dim3(div_up(n_values, 4)), dim3(256), 0, stream, d_wl_code[p].p, u32(wl.parts[p].size()), L, n_values, d_wl_tab[p].p);
				});
			}
			HIP_CHECK(stream_wait(stream));
			wl_tab_ok = true;
		}
	}
}

// Neighbour search of RealBarcodesMergeStrategy::get_merge_target for a list of base cells, over a universe of
// cells (the context's own, or in sharded runs the real cells of every shard).  Everything per-cell that scales
// with the number of filtered cells (10^5..10^6 at BASELINE sizes) is done on the device; the host only walks flat
// arrays.  Fills the candidate lists and the (base, candidate) pairs whose UMI-gene intersection is needed.
void dropest_ctx::search_merge_candidates(const std::vector<u32> &cells, const MergeUniverse &U, MergeSearch &S) {
	if (!wl.loaded) {
		if (barcodes_file.empty()) throw InvalidError("merge_kind = REAL_BARCODES needs barcodes_file");
		wl.load(cfg.barcodes_kind, barcodes_file);
	}
	if (wl.parts.size() > size_t(WL_MAX_PARTS)) return search_merge_candidates_host(cells, U, S);
	upload_whitelist();
	const u32 F = u32(cells.size());
	S.F = F;
	S.cells = cells;

	// 1. barcodes split into the two parts (device; escaped barcodes patched by the host)
	auto st_bases = std::make_unique<HostStage>(this, "cb_merge:targets:bases");
	DevBuf<u32> &d_cells = S.d_cells; d_cells.ensure(F);
	S.d_bases.ensure(F);
	scalars.ensure(16);
	upload(d_cells.p, cells.data(), size_t(F) * 4);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 16, stream));
	const u32 P = u32(wl.parts.size());
	WlSplit sp{};
	sp.n_parts = P; sp.const_kind = cfg.barcodes_kind == DROPEST_BARCODES_CONST ? 1 : 0;
	for (u32 p = 0; p < P; ++p) sp.len[p] = u32(wl.part_lengths[p]);
	hipLaunchKernelGGL(make_bases_kernel, dim3(div_up(F, 256)), dim3(256), 0, stream, d_cells.p, F, U.cell_cb, sp, S.d_bases.p, scalars.p);
	HIP_CHECK(hipGetLastError());
	u32 bad = 0;
	fetch(&bad, scalars.p, 4);
	if (bad == 2) throw UnsupportedError("barcode part longer than 31 bases");
	if (bad) {   // reproduce the reference's message for the first offending barcode
		for (u32 f = 0; f < F; ++f) wl.split(U.base_barcode_text(f));
		throw InvalidError("barcode length does not fit the whitelist");
	}
	if (U.any_escaped) {
		for (u32 f = 0; f < F; ++f) {
			if (!(U.barcode_code(cells[f]) & ESCAPE_BIT)) continue;
			const std::vector<std::string> pieces = wl.split(U.base_barcode_text(f));
			WlBase wb;
			std::memset(&wb, 0, sizeof(wb));
			for (size_t p = 0; p < pieces.size(); ++p) { std::memcpy(wb.part[p], pieces[p].data(), pieces[p].size()); wb.len[p] = uint8_t(pieces[p].size()); }
			wb.cell = cells[f];
			HIP_CHECK(hipMemcpy(S.d_bases.p + f, &wb, sizeof(wb), hipMemcpyHostToDevice));
		}
	}

	// 2. neighbour search; candidates land in flat lists
	st_bases.reset();
	auto st_search = std::make_unique<HostStage>(this, "cb_merge:targets:search");
	DevBuf<u32> &d_cnt = S.d_cnt, &d_lvl = S.d_lvl, &d_off = S.d_off, &d_fcell = S.d_fcell, &d_fumis = S.d_fumis, &d_fridx = S.d_fridx, &d_lvl_todo = S.d_todo;
	u32 flat_cap = std::max<u32>(F * 2u, 1024u);
	S.cnt.resize(F); S.off.resize(F);
	WlArgs &a = S.args;
	S.ntot = 0;
	for (auto const &part : wl.parts) S.ntot += u32(part.size());
	S.lds = ((S.ntot + 15u) & ~15u) + size_t(S.ntot) * 2;
	for (;;) {
		d_cnt.ensure(F); d_lvl.ensure(F); d_off.ensure(F); d_fcell.ensure(flat_cap); d_fumis.ensure(flat_cap); d_fridx.ensure(flat_cap);
		HIP_CHECK(hipMemsetAsync(scalars.p, 0, 16, stream));
		a = WlArgs{};
		a.bases = S.d_bases.p; a.n_bases = F;
		a.n_parts = P;
		for (u32 p = 0; p < P; ++p) { a.part[p] = d_wl[p].p; a.part_code[p] = d_wl_code[p].p; a.part_size[p] = u32(wl.parts[p].size()); }
		a.table = U.table; a.cell_n_genes = U.n_genes; a.cell_total_umis = U.total_umis; a.min_genes = min_before;
		a.cand_count = d_cnt.p; a.cand_level = d_lvl.p; a.cand_off = d_off.p; a.flat_cell = d_fcell.p; a.flat_umis = d_fumis.p;
		a.flat_ridx = d_fridx.p; a.cell_real_index = U.real_index;
		a.flat_total = scalars.p; a.flat_cap = flat_cap; a.dist_dump = nullptr;
		a.poisson = cfg.merge_kind == DROPEST_MERGE_POISSON_REAL ? 1 : 0;
		a.base_list = nullptr;
		if (wl_tab_ok) {
			// most bases are decided from the neighbour tables by one thread each; the rest goes through the full search
			WlTabArgs ta{};
			for (u32 p = 0; p < P; ++p) { ta.rows[p] = d_wl_tab[p].p; ta.len[p] = int(wl.part_lengths[p]); }
			timed("wl_table_search", double(F) * (sizeof(WlBase) + 2 * sizeof(WlTabRow)), [&] {
				switch (P) {
					case 1: hipLaunchKernelGGL(wl_table_search_kernel<1>, dim3(div_up(F, 256)), dim3(256), 0, stream, a, ta); break;
					case 2: hipLaunchKernelGGL(wl_table_search_kernel<2>, dim3(div_up(F, 256)), dim3(256), 0, stream, a, ta); break;
					case 3: hipLaunchKernelGGL(wl_table_search_kernel<3>, dim3(div_up(F, 256)), dim3(256), 0, stream, a, ta); break;
					default: hipLaunchKernelGGL(wl_table_search_kernel<4>, dim3(div_up(F, 256)), dim3(256), 0, stream, a, ta); break;
				}
			});
			DevBuf<u32> &todo = d_lvl_todo;
			todo.ensure(F);
			hipLaunchKernelGGL(wl_collect_todo_kernel, dim3(div_up(F, 256)), dim3(256), 0, stream, d_cnt.p, F, todo.p, scalars.p + 1);
			HIP_CHECK(hipGetLastError());
			u32 n_todo = 0;
			fetch(&n_todo, scalars.p + 1, 4);
			if (profiling) stats["count:wl_bases_full_search"].launches += n_todo;
			if (n_todo) {
				a.base_list = todo.p;
				timed("wl_neighbours", double(n_todo) * S.ntot * 32, [&] { wl_neighbours_launch(a, n_todo, S.lds, stream); });
			}
		} else {
			timed("wl_neighbours", double(F) * S.ntot * 32, [&] {
				wl_neighbours_launch(a, F, S.lds, stream);
			});
		}
		a.base_list = nullptr;   // (S.args outlives this call: the tie replay launches from it)
		u32 total = 0;
		fetch(&total, scalars.p, 4);
		if (total <= flat_cap) {
			fetch(S.cnt.data(), d_cnt.p, size_t(F) * 4);
			fetch(S.off.data(), d_off.p, size_t(F) * 4);
			S.fcell.resize(total); S.fumis.resize(total); S.fridx.resize(total);
			fetch(S.fcell.data(), d_fcell.p, size_t(total) * 4);
			fetch(S.fumis.data(), d_fumis.p, size_t(total) * 4);
			fetch(S.fridx.data(), d_fridx.p, size_t(total) * 4);
			break;
		}
		flat_cap = total + 1024;   // rare: more candidates than twice the number of cells; run again with room
	}

	// 3. pairs (base, candidate) whose UMI-gene intersection is needed
	st_search.reset();
	build_merge_pairs(cells, S);
}

void dropest_ctx::build_merge_pairs(const std::vector<u32> &cells, MergeSearch &S) {
	const u32 F = S.F;
	HostStage st_pairs(this, "cb_merge:targets:pairs");
	S.pair_first.assign(size_t(F) + 1, 0); S.self_ridx.assign(F, 0xFFFFFFFFu);
	// two passes over contiguous ranges on a few threads (2.4 M bases at C3 size): pairs per base, then the flat lists
	constexpr unsigned W = dropest::HostPool::MAX;
	size_t per_worker[W + 1] = {0};
	bool too_many = false;
	std::vector<u32> n_pairs(F);
	const unsigned workers = parallel_ranges(F, [&](size_t b, size_t e, unsigned w) {
		size_t mine = 0; bool over = false;
		for (size_t f = b; f < e; ++f) {
			if (S.cnt[f] > u32(WL_CAND_CAP)) { over = true; continue; }
			bool self = false;
			for (u32 k = 0; k < S.cnt[f]; ++k)
				if (S.fcell[S.off[f] + k] == cells[f]) { self = true; S.self_ridx[f] = S.fridx[S.off[f] + k]; }
			// the base is itself a whitelist barcode: neighbour_cells[0] == base (RealBarcodesMergeStrategy.cpp:34-35) ends the
			// decision there; the Poisson estimator goes on to its other neighbours (PoissonTargetEstimator.cpp:26-29)
			u32 np = 0;
			if (!(self && cfg.merge_kind != DROPEST_MERGE_POISSON_REAL)) np = S.cnt[f] - (self ? 1u : 0u);
			n_pairs[f] = np; mine += np;
		}
		per_worker[w] = mine;
		if (over) too_many = true;
	}, 100000, W);
	if (too_many) throw UnsupportedError("more than " + std::to_string(WL_CAND_CAP) + " merge candidates for one barcode");
	size_t start[W + 1] = {0};
	for (unsigned w = 0; w < workers; ++w) start[w + 1] = start[w] + per_worker[w];
	const size_t NP = start[workers];
	if (NP > 0xFFFFFFF0ull) throw UnsupportedError("more than 2^32 (base, candidate) pairs");
	S.pair_base.resize(NP); S.pair_cand.resize(NP); S.pair_umis.resize(NP); S.pair_ridx.resize(NP);
	parallel_ranges(F, [&](size_t b, size_t e, unsigned w) {   // same n and limits as above: the same ranges
		size_t at = start[w];
		for (size_t f = b; f < e; ++f) {
			S.pair_first[f] = u32(at);
			if (!n_pairs[f]) continue;
			for (u32 k = 0; k < S.cnt[f]; ++k) {
				if (S.fcell[S.off[f] + k] == cells[f]) continue;
				S.pair_base[at] = u32(f); S.pair_cand[at] = S.fcell[S.off[f] + k]; S.pair_umis[at] = S.fumis[S.off[f] + k];
				S.pair_ridx[at] = S.fridx[S.off[f] + k];
				++at;
			}
		}
	}, 100000, W);
	S.pair_first[F] = u32(NP);
}

// The same search on the host, literally as the reference runs it (BarcodesParser::get_distances_to_barcode / push_remaining_dists,
// BarcodesParser.cpp:21-74; RealBarcodesMergeStrategy::get_real_neighbour_cbs, RealBarcodesMergeStrategy.cpp:63-109) with the same
// libstdc++ sorts on the same sequences: for whitelists of MORE parts than the device kernel is built for (WL_MAX_PARTS; the
// reference has no limit, ConstLengthBarcodesParser.cpp:50-68).  A shard searches ITS bases against the all-gathered cell list.
void dropest_ctx::search_merge_candidates_host(const std::vector<u32> &cells, const MergeUniverse &U, MergeSearch &S) {
	if (!U.find_cell) throw InvalidError("internal: the host search of a whitelist needs a barcode look-up of the universe");
	HostStage st(this, "cb_merge:targets:host_search");
	const u32 F = u32(cells.size());
	const size_t P = wl.parts.size();
	const bool poisson = cfg.merge_kind == DROPEST_MERGE_POISSON_REAL;
	S.F = F; S.cells = cells;
	S.cnt.assign(F, 0); S.off.assign(F, 0); S.fcell.clear(); S.fumis.clear(); S.fridx.clear();
	S.host_order.assign(F, {});
	S.ntot = 0;
	for (auto const &part : wl.parts) S.ntot += u32(part.size());
	std::vector<std::vector<std::string>> all_pieces(F);
	for (u32 f = 0; f < F; ++f) all_pieces[f] = wl.split(U.base_barcode_text(f));   // throws the reference's length errors (caller's thread)
	auto search_base = [&](u32 f, std::vector<u32> &cand, std::vector<u32> &umis, std::vector<u32> &ridx) {
		const std::vector<std::string> &pieces = all_pieces[f];
		std::vector<std::vector<PartDistH>> d(P);
		for (size_t p = 0; p < P; ++p) {
			for (size_t i = 0; i < wl.parts[p].size(); ++i)
				d[p].push_back(PartDistH{i, long(banded_edit_distance(pieces[p], wl.parts[p][i], 10000u))});   // Tools::edit_distance, default max_ed
			std::sort(d[p].begin(), d[p].end(), [](const PartDistH &x, const PartDistH &y) { return x.value < y.value; });
		}
		std::vector<size_t> idx(P, 0);
		// push_remaining_dists over any number of parts (the fixed-size ComboH of the tie replay holds WL_MAX_PARTS)
		struct Combo { std::vector<size_t> idx; unsigned ed; };
		std::vector<Combo> all;
		std::function<void(size_t, unsigned)> walk = [&](size_t part, unsigned ed) {
			if (part == P) { all.push_back(Combo{idx, ed}); return; }
			for (const PartDistH &x : d[part]) {
				const unsigned cur_ed = ed + unsigned(x.value);
				if (cur_ed > unsigned(WL_MAX_DIST)) return;
				idx[part] = x.index;
				walk(part + 1, cur_ed);
			}
		};
		walk(0, 0);
		if (all.empty()) return;
		std::sort(all.begin(), all.end(), [](const Combo &x, const Combo &y) { return x.ed < y.ed; });
		unsigned max_dist = all.front().ed;
		if (poisson) max_dist = max_dist == 0 ? 2 : max_dist + 1;   // PoissonRealBarcodesMergeStrategy::get_max_merge_dist (:20-23)
		const int32_t base_umis = U.base_total_umis(f);
		for (const Combo &c : all) {
			if (c.ed > max_dist && !cand.empty()) break;
			std::string cb;
			for (size_t p = 0; p < P; ++p) cb += wl.parts[p][c.idx[p]];
			u64 code = 0;
			if (encode_code(cb, code)) {
				u32 ng = 0, ri = 0; int32_t tu = 0;
				const long cell = U.find_cell(code, ng, tu, ri);
				if (cell >= 0 && ng >= min_before && size_t(tu) >= size_t(base_umis)) { cand.push_back(u32(cell)); umis.push_back(u32(tu)); ridx.push_back(ri); }
			}
			max_dist = std::max(max_dist, c.ed);
		}
	};
	std::vector<std::vector<u32>> cand(F), umis(F), ridx(F);
	parallel_ranges(F, [&](size_t b, size_t e, unsigned) { for (size_t f = b; f < e; ++f) search_base(u32(f), cand[f], umis[f], ridx[f]); }, 64, 16);
	for (u32 f = 0; f < F; ++f) {
		S.off[f] = u32(S.fcell.size()); S.cnt[f] = u32(cand[f].size());
		S.fcell.insert(S.fcell.end(), cand[f].begin(), cand[f].end());
		S.fumis.insert(S.fumis.end(), umis[f].begin(), umis[f].end());
		S.fridx.insert(S.fridx.end(), ridx[f].begin(), ridx[f].end());
		S.host_order[f] = std::move(cand[f]);
	}
	build_merge_pairs(cells, S);
}

// The reference's candidate order (a replay of its two unstable std::sorts) for the bases in `need_order`: the
// kernel runs again for those bases only and dumps the per-part distances.  Only candidates that are part of S's
// pairs appear (the base itself is skipped like PoissonTargetEstimator.cpp:28-29 does).
std::vector<std::vector<u32>> dropest_ctx::replay_candidate_orders(const MergeUniverse &U, MergeSearch &S, const std::vector<u32> &need_order) {
	if (!S.host_order.empty()) {   // the host search already walked the reference's order; the base itself is not a pair
		std::vector<std::vector<u32>> orders(need_order.size());
		for (size_t r = 0; r < need_order.size(); ++r)
			for (u32 c : S.host_order[need_order[r]]) if (c != S.cells[need_order[r]]) orders[r].push_back(c);
		return orders;
	}
	const u32 R = u32(need_order.size()), ntot = S.ntot;
	std::vector<WlBase> rb(R);
	for (u32 r = 0; r < R; ++r) HIP_CHECK(hipMemcpy(&rb[r], S.d_bases.p + need_order[r], sizeof(WlBase), hipMemcpyDeviceToHost));
	DevBuf<WlBase> d_rb; d_rb.alloc(R);
	DevBuf<uint8_t> d_dump; d_dump.alloc(size_t(R) * ntot);
	DevBuf<u32> d_c2, d_l2, d_o2, d_f2, d_u2, d_r2;
	d_c2.alloc(R); d_l2.alloc(R); d_o2.alloc(R); d_f2.alloc(size_t(R) * WL_CAND_CAP); d_u2.alloc(size_t(R) * WL_CAND_CAP);
	d_r2.alloc(size_t(R) * WL_CAND_CAP);
	HIP_CHECK(hipMemcpyAsync(d_rb.p, rb.data(), size_t(R) * sizeof(WlBase), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 16, stream));
	WlArgs a2 = S.args;
	a2.bases = d_rb.p; a2.n_bases = R; a2.cand_count = d_c2.p; a2.cand_level = d_l2.p; a2.cand_off = d_o2.p;
	a2.flat_cell = d_f2.p; a2.flat_umis = d_u2.p; a2.flat_ridx = d_r2.p; a2.flat_cap = R * u32(WL_CAND_CAP); a2.dist_dump = d_dump.p;
	a2.base_list = nullptr;   // block r works on replay base r (the list of the main search is gone)
	wl_neighbours_launch(a2, R, S.lds, stream);
	HIP_CHECK(hipGetLastError());
	std::vector<uint8_t> dump(size_t(R) * ntot);
	fetch(dump.data(), d_dump.p, dump.size());
	std::vector<std::vector<u32>> orders(R);
	for (u32 r = 0; r < R; ++r) {
		const u32 f = need_order[r];
		std::unordered_map<u64, u32> by_code;
		for (u32 p = S.pair_first[f]; p < S.pair_first[f + 1]; ++p) by_code[U.barcode_code(S.pair_cand[p])] = S.pair_cand[p];
		orders[r] = reference_candidate_order(wl, dump.data() + size_t(r) * ntot, by_code, S.args.poisson != 0);
	}
	return orders;
}

// Decisions (RealBarcodesMergeStrategy::get_best_merge_target, :31-61) from the intersection sizes of S's pairs;
// all inputs are integers, the fraction is evaluated exactly as the reference writes it (double, no contraction on
// the host).  targets[f] = candidate cell (universe index) or -1; target_ridx[f] = its real_index.
void dropest_ctx::decide_merge_targets(const MergeUniverse &U, MergeSearch &S, const std::vector<u32> &inter,
                                       std::vector<long> &targets, std::vector<u32> &target_ridx) {
	const u32 F = S.F;
	targets.assign(F, -1);
	target_ridx.assign(F, 0xFFFFFFFFu);
	std::vector<u32> need_order;   // cells whose result depends on the reference's candidate order
	auto frac_of = [&](u32 f, u32 p) {
		return 0.5 * inter[p] * (1. / size_t(U.base_total_umis(f)) + 1. / size_t(int32_t(S.pair_umis[p])));
	};
	// independent per base: a few worker threads over contiguous ranges (2.5 M bases at C3 size, each a cache-missing
	// look-up of its row); the bases that need the replay are collected per range and concatenated in order
	auto decide_range = [&](u32 f0, u32 f1, std::vector<u32> &ties) {
		for (u32 f = f0; f < f1; ++f) {
			if (S.cnt[f] == 0) { targets[f] = -1; continue; }
			const u32 p0 = S.pair_first[f], p1 = S.pair_first[f + 1];
			if (p0 == p1) { targets[f] = long(S.cells[f]); target_ridx[f] = S.self_ridx[f]; continue; }   // self
			double best = 0; u32 n_best = 0, best_p = 0;
			for (u32 p = p0; p < p1; ++p) {
				const double fr = frac_of(f, p);
				if (fr > best) { best = fr; n_best = 1; best_p = p; }
				else if (fr == best) ++n_best;
			}
			if (best < cfg.min_merge_fraction) { targets[f] = -1; continue; }   // holds for any order
			if (best > 0 && n_best == 1) { targets[f] = long(S.pair_cand[best_p]); target_ridx[f] = S.pair_ridx[best_p]; continue; }
			ties.push_back(f);   // tie at the maximum (or all fractions zero with a non-positive threshold)
		}
	};
	{
		std::vector<std::vector<u32>> ties(dropest::HostPool::MAX);
		const unsigned workers = parallel_ranges(F, [&](size_t b, size_t e, unsigned w) {
			std::vector<u32> mine;   // local: the slots' vector headers share cache lines
			decide_range(u32(b), u32(e), mine);
			ties[w] = std::move(mine);
		}, 50000, dropest::HostPool::MAX);
		for (unsigned w = 0; w < workers; ++w) need_order.insert(need_order.end(), ties[w].begin(), ties[w].end());
	}
	if (need_order.empty()) return;
	HostStage st_replay(this, "cb_merge:targets:replay");
	if (profiling) stats["count:merge_ties_replayed"].launches += need_order.size();
	const std::vector<std::vector<u32>> orders = replay_candidate_orders(U, S, need_order);
	for (u32 r = 0; r < u32(need_order.size()); ++r) {
		const u32 f = need_order[r];
		std::unordered_map<u32, u32> pair_of;
		for (u32 p = S.pair_first[f]; p < S.pair_first[f + 1]; ++p) pair_of[S.pair_cand[p]] = p;
		const std::vector<u32> &order = orders[r];
		if (order.empty()) throw DeviceError("internal: candidate replay found no candidate");
		double best = 0; u32 best_cell = order[0];
		for (u32 c : order) {
			const double fr = frac_of(f, pair_of.at(c));
			if (best < fr) { best = fr; best_cell = c; }
		}
		targets[f] = best < cfg.min_merge_fraction ? -1 : long(best_cell);
		if (targets[f] >= 0) target_ridx[f] = S.pair_ridx[pair_of.at(best_cell)];
	}
}

// CellsDataContainer.cpp:320-350 (get_umigs_intersect_size) for a list of cell pairs of this context
std::vector<u32> dropest_ctx::pair_intersections(const std::vector<u32> &pb, const std::vector<u32> &pc) {
	std::vector<u32> inter;
	pair_intersections(pb, pc, inter);
	return inter;
}
void dropest_ctx::pair_intersections(const std::vector<u32> &pb, const std::vector<u32> &pc, std::vector<u32> &inter) {
	const u32 NP = u32(pb.size());
	inter.resize(NP);
	if (!NP) return;
	DevBuf<u32> &d_pb = ms.d_pb, &d_pc = ms.d_pc, &d_inter = ms.d_inter; DevBuf<PairRange> &d_pr = ms.d_pr;
	d_pb.ensure(NP); d_pc.ensure(NP); d_inter.ensure(NP); d_pr.ensure(NP);
	upload(d_pb.p, pb.data(), size_t(NP) * 4);
	upload(d_pc.p, pc.data(), size_t(NP) * 4);
	hipLaunchKernelGGL(pair_ranges_kernel, dim3(div_up(NP, 256)), dim3(256), 0, stream, d_pb.p, d_pc.p, NP, cell_cg_begin.p,
	                   cell_cg_count.p, cg_mol_begin.p, d_pr.p);
	HIP_CHECK(hipGetLastError());
	const int low_bits = layout.gene_bits + layout.umi_bits;
	timed("umig_intersect", double(NP) * 64, [&] {
		hipLaunchKernelGGL(umig_intersect_kernel, dim3(NP), dim3(256), 0, stream, d_pr.p, NP, mol_key.p, mol_key.p,
		                   (1ull << low_bits) - 1ull, layout.umi_bits, layout.gene_none, d_inter.p);
	});
	fetch(inter.data(), d_inter.p, size_t(NP) * 4);
}

// RealBarcodesMergeStrategy::get_merge_target for a list of this context's cells, on the current (unmerged) device
// state.  `ridx[f]` = index of cells[f] in `real`.
std::vector<long> dropest_ctx::compute_merge_targets(const std::vector<u32> &cells, const std::vector<u32> &ridx, std::vector<u32> *target_ridx) {
	std::vector<long> targets;
	compute_merge_targets(cells, ridx, targets, target_ridx);
	return targets;
}
void dropest_ctx::compute_merge_targets(const std::vector<u32> &cells, const std::vector<u32> &ridx, std::vector<long> &targets,
                                        std::vector<u32> *target_ridx) {
	targets.assign(cells.size(), -1);
	if (target_ridx) target_ridx->assign(cells.size(), 0xFFFFFFFFu);
	if (cells.empty()) return;

	// cell id -> index in `real`
	{
		const u32 nr = u32(real.size());
		std::vector<u32> ids;
		cell_real_index.ensure(n_cells);
		if (!real_list_current) {   // (fetch_real_cells left the ids on the device: usually nothing to upload)
			ids.resize(nr);
			for (u32 i = 0; i < nr; ++i) ids[i] = real[i].id;
			real_list.ensure(nr);
			HIP_CHECK(hipMemcpyAsync(real_list.p, ids.data(), size_t(nr) * 4, hipMemcpyHostToDevice, stream));
			real_list_current = true;
		}
		HIP_CHECK(hipMemsetAsync(cell_real_index.p, 0xFF, size_t(n_cells) * 4, stream));
		hipLaunchKernelGGL(scatter_index_kernel, dim3(div_up(nr, 256)), dim3(256), 0, stream, real_list.p, nr, cell_real_index.p);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(stream_wait(stream));
	}
	HostStage st_universe(this, "cb_merge:targets:universe");
	MergeUniverse U;
	U.table = table; U.cell_cb = cell_cb.p; U.n_genes = cell_n_genes.p; U.total_umis = cell_total_umis.p;
	U.real_index = cell_real_index.p; U.any_escaped = ingest.cb_escape_count != 0;
	// (the bases are the filtered cells in their order: their TOTAL_UMIS stand in a dense array beside the list -- 2.4e6 look-ups of
	// real[ridx[f]] were a cache miss each in an array of fat rows, most of the 4.3 ms the decisions took at C3 size)
	const bool dense_umis = filtered_valid && filtered_umis.size() == cells.size() && filtered_ridx.size() == ridx.size() && (ridx.empty() || (filtered_ridx.front() == ridx.front() && filtered_ridx.back() == ridx.back()));
	const std::vector<int32_t> base_umis = dense_umis ? filtered_umis : std::vector<int32_t>();
	U.base_total_umis = [&](u32 f) { return base_umis.empty() ? real[ridx[f]].row.total_umis : base_umis[f]; };
	U.barcode_code = [&](u32 cell) { return u64(real[real_at(cell)].row.barcode); };
	U.base_barcode_text = [&](u32 f) { return barcode_of(real[ridx[f]]); };
	std::unordered_map<u64, u32> real_by_code;   // filled on first use (host search only)
	U.find_cell = [&](u64 code, u32 &ng, int32_t &tu, u32 &ri) -> long {
		static std::mutex mu;
		{
			std::lock_guard<std::mutex> lk(mu);
			if (real_by_code.empty()) for (u32 i = 0; i < real.size(); ++i) real_by_code.emplace(u64(real[i].row.barcode), i);
		}
		auto it = real_by_code.find(code);
		if (it == real_by_code.end()) return -1;   // not a real-candidate cell: fewer genes than min_genes_before_merge, never qualifies
		const HostCell &h = real[it->second];
		ng = h.row.n_genes; tu = h.row.total_umis; ri = it->second;
		return long(h.id);
	};
	MergeSearch &S = ms.S;            // (kept across passes: see MergeScratch)
	S.host_order.clear();
	search_merge_candidates(cells, U, S);

	const u32 NP = u32(S.pair_base.size());
	std::vector<u32> &pb = ms.pb;
	pb.resize(NP);
	parallel_ranges(NP, [&](size_t b, size_t e, unsigned) { for (size_t p = b; p < e; ++p) pb[p] = cells[S.pair_base[p]]; }, 100000, dropest::HostPool::MAX);
	std::vector<u32> &inter = ms.inter;
	{ HostStage st(this, "cb_merge:targets:intersect"); pair_intersections(pb, S.pair_cand, inter); }
	HostStage st_decide(this, "cb_merge:targets:decide");
	std::vector<u32> &tr = ms.tr;
	if (cfg.merge_kind == DROPEST_MERGE_POISSON_REAL) {
		const std::vector<double> expected = poisson_expected_intersections(pb, S.pair_cand);
		decide_poisson_targets(U, S, inter, expected, targets, tr);
	} else
		decide_merge_targets(U, S, inter, targets, tr);
	if (target_ridx) target_ridx->assign(tr.begin(), tr.end());
}

// MergeStrategyBase::merge_inited second loop (:30-51) + reassign (:64-82), on flat arrays indexed by the position
// in the caller's cell list: order[i] = base of step i (ascending compare_cells order), target[i] = its target
// position or -1.  final_target[x] = where x's molecules end up (x itself if never merged); the cells merged into
// a target form an intrusive list so that a later merge of that target moves them along.  Stats::merge adds
// every int counter, TOTAL_UMIS included (Stats.cpp:29-43).  Returns whether anything was merged.
static bool apply_merge_order(u32 n_cells, size_t n_order, const u32 *order, const int64_t *target, int32_t *total_reads,
                              int32_t *total_umis, u32 *final_target, uint8_t *excluded, u32 *rank = nullptr,
                              std::vector<u32> *scratch = nullptr /* 3 x n_cells, kept by the caller across passes */) {
	const u32 NIL = 0xFFFFFFFFu;
	std::vector<u32> own;
	std::vector<u32> &lists = scratch ? *scratch : own;
	// The common shape (every whitelist merge): no target is itself merged away, so no step sees the result of another -- the steps
	// commute (integer sums, disjoint flags) and run on worker threads.  2.5 M steps at C3 size: 1 ms instead of 8.  The intrusive
	// lists exist only for chains (a target merged later takes its cells along) and for the quality ranks.
	size_t parallel_min = 100000;
	if (const char *e = getenv("DROPEST_PARALLEL_MERGE_ORDER_MIN")) parallel_min = size_t(std::max(1, atoi(e)));   // (tests)
	if (!rank && n_order >= parallel_min && !getenv("DROPEST_SERIAL_MERGE_ORDER")) {
		const size_t per_worker = std::max<size_t>(1, std::min<size_t>(100000, parallel_min));
		constexpr u32 NONE = 0xFFFFFFFEu, EXCLUDED = 0xFFFFFFFDu;
		lists.resize(size_t(n_cells) * 3);
		u32 *tgt_of = lists.data();
		constexpr unsigned W = dropest::HostPool::MAX;
		dropest::parallel_ranges(n_cells, [&](size_t b, size_t e, unsigned) { for (size_t i = b; i < e; ++i) { tgt_of[i] = NONE; final_target[i] = u32(i); excluded[i] = 0; } }, per_worker, W);
		unsigned bad[W + 1] = {0};
		dropest::parallel_ranges(n_order, [&](size_t b, size_t e, unsigned w) {
			for (size_t i = b; i < e; ++i) {
				const u32 c = order[i];
				if (c >= n_cells || (target[i] >= 0 && u64(target[i]) >= n_cells)) { bad[w] = 1; return; }
				tgt_of[c] = target[i] < 0 ? EXCLUDED : u32(target[i]);   // (a cell stands in the order once)
			}
		}, per_worker, W);
		unsigned chains[W + 1] = {0};
		bool any_bad = false;
		for (unsigned w = 0; w <= W; ++w) any_bad |= bad[w] != 0;
		if (!any_bad) dropest::parallel_ranges(n_order, [&](size_t b, size_t e, unsigned w) {
			for (size_t i = b; i < e; ++i) {
				if (target[i] < 0 || u32(target[i]) == order[i]) continue;
				const u32 tt = tgt_of[u32(target[i])];
				if (tt != NONE && tt != EXCLUDED && tt != u32(target[i])) { chains[w] = 1; return; }
			}
		}, per_worker, W);
		bool any_chain = any_bad;   // (bad indices: the serial loop below throws the reference-shaped error)
		for (unsigned w = 0; w <= W; ++w) any_chain |= chains[w] != 0;
		if (!any_chain) {
			// every worker owns a range of TARGETS and reads the whole order: sums without atomics (50 000 targets taking 2.4 M
			// additions from sixteen threads would pass their cache lines around), every cell written by exactly one worker
			unsigned merged[W + 1] = {0};
			// ... walking the CELLS in index order (tgt_of holds every cell's step): the per-cell arrays stream through, only the few
			// targets are touched out of order.  (Walking the order instead read total_reads[c] / total_umis[c] of cells that stand anywhere
			// in arrays of millions of entries: 150 000 cache misses per worker, 5.5 ms at C3 size.)
			dropest::parallel_ranges(n_cells, [&](size_t t0, size_t t1, unsigned w) {
				for (size_t c = 0; c < n_cells; ++c) {
					const u32 tg = tgt_of[c];
					if (tg == NONE) continue;
					if (tg == EXCLUDED) { if (c >= t0 && c < t1) excluded[c] = 1; continue; }
					if (tg < t0 || tg >= t1 || tg == c) continue;
					total_reads[tg] += total_reads[c];   // (c is nobody's target: its own sums are final)
					total_umis[tg] += total_umis[c];
					final_target[c] = tg;
					merged[w] = 1;
				}
			}, per_worker, W);
			bool any = false;
			for (unsigned w = 0; w <= W; ++w) any |= merged[w] != 0;
			return any;
		}
	}
	lists.assign(size_t(n_cells) * 3, NIL);
	u32 *head = lists.data(), *tail = head + n_cells, *next = tail + n_cells;
	u32 *cur = final_target;
	for (u32 i = 0; i < n_cells; ++i) { cur[i] = i; excluded[i] = 0; }
	auto append = [&](u32 tgt, u32 x) { if (head[tgt] == NIL) head[tgt] = x; else next[tail[tgt]] = x; tail[tgt] = x; next[x] = NIL; };
	bool any_merge = false;
	for (size_t i = 0; i < n_order; ++i) {
		const u32 b = order[i];
		if (b >= n_cells) throw RangeError("merge order refers to a cell outside the list");
		if (target[i] < 0) { excluded[b] = 1; continue; }
		if (u64(target[i]) >= n_cells) throw RangeError("merge target outside the list");
		const u32 tr = cur[u32(target[i])];        // "real barcodes could be merged too" (:43-46)
		if (tr == b) continue;
		// CellsDataContainer::merge_cells (:90-104)
		total_reads[tr] += total_reads[b];
		total_umis[tr] += total_umis[b];
		any_merge = true;
		cur[b] = tr;
		u32 moved = head[b];                         // cells previously merged into b follow it (reassign, :64-82)
		head[b] = tail[b] = NIL;
		append(tr, b);
		while (moved != NIL) { const u32 nx = next[moved]; cur[moved] = tr; append(tr, moved); moved = nx; }
	}
	if (rank) {   // order in which Gene::merge offered the cells' molecules to the final target (it keeps the first it saw)
		for (u32 i = 0; i < n_cells; ++i) rank[i] = 0;
		for (u32 t = 0; t < n_cells; ++t) {
			u32 r = 0;
			for (u32 x = head[t]; x != NIL; x = next[x]) rank[x] = ++r;
		}
	}
	return any_merge;
}

void dropest_ctx::run_cb_merge_real() {
	HostStage hs(this, "cb_merge");
	const std::vector<uint64_t> &order = filtered_cells();
	std::vector<u32> &cells = ms.cells;
	cells.assign(order.begin(), order.end());
	std::vector<u32> &ridx = ms.ridx;
	ridx.assign(filtered_ridx.begin(), filtered_ridx.end());
	std::vector<long> &targets = ms.targets;
	std::vector<u32> &target_ridx = ms.target_ridx;
	{ HostStage hs2(this, "cb_merge:targets"); compute_merge_targets(cells, ridx, targets, &target_ridx); }

	HostStage hs3(this, "cb_merge:apply");
	const u32 nR = u32(real.size());
	std::vector<int64_t> &tgt = ms.tgt;
	tgt.resize(cells.size());
	parallel_ranges(cells.size(), [&](size_t b, size_t e, unsigned) {
		for (size_t i = b; i < e; ++i) tgt[i] = targets[i] < 0 ? -1 : int64_t(target_ridx[i]);
	}, 100000, dropest::HostPool::MAX);
	std::vector<int32_t> &reads = ms.reads, &umis = ms.umis;
	reads.resize(nR); umis.resize(nR);
	parallel_ranges(nR, [&](size_t b, size_t e, unsigned) {
		for (size_t i = b; i < e; ++i) { reads[i] = real[i].row.total_reads; umis[i] = real[i].row.total_umis; }
	}, 100000, dropest::HostPool::MAX);
	std::vector<u32> &cur = ms.cur;
	std::vector<uint8_t> &excl = ms.excl;
	std::vector<u32> &rank = ms.rank;   // only the quality sums need the merge order (quality.h)
	cur.resize(nR); excl.resize(nR); rank.resize(have_qual ? nR : 0u);
	bool any_merge;
	{ HostStage hs4(this, "cb_merge:apply:order");
	any_merge = apply_merge_order(nR, cells.size(), ridx.data(), tgt.data(), reads.data(), umis.data(), cur.data(), excl.data(),
	                              have_qual ? rank.data() : nullptr, &ms.lists); }
	if (have_qual) {
		merge_rank.assign(n_cells, 0);
		for (u32 i = 0; i < nR; ++i) merge_rank[real[i].id] = rank[i];
	}
	reassign.clear();
	clear_strategy_pairs();
	// the (source, target) pairs in ascending source id: counted and filled over the same contiguous ranges
	constexpr unsigned W = dropest::HostPool::MAX;
	size_t n_of[W + 1] = {0};
	real_pristine = false;
	const unsigned workers = parallel_ranges(nR, [&](size_t b, size_t e, unsigned w) {
		size_t c = 0;
		for (size_t i = b; i < e; ++i) {
			real[i].row.total_reads = reads[i]; real[i].row.total_umis = umis[i];
			if (excl[i]) real[i].excluded = true;
			if (cur[i] != i) { real[i].merged = true; ++c; }
		}
		n_of[w] = c;
	}, 100000, W);
	size_t start[W + 1] = {0};
	const size_t kept = merge_pairs.size();   // pairs merged by hand stay in front (clear_strategy_pairs)
	for (unsigned w = 0; w < workers; ++w) start[w + 1] = start[w] + n_of[w];
	merge_pairs.resize(kept + start[workers]);
	// (the targets' ids from a dense array: real[cur[i]].id is a cache miss per merged cell in an array of 48-byte rows, 2.4e6 of them at C3)
	std::vector<u32> &ids_dense = ms.ids_dense;
	ids_dense.resize(nR);
	parallel_ranges(nR, [&](size_t b, size_t e, unsigned) { for (size_t i = b; i < e; ++i) ids_dense[i] = real[i].id; }, 100000, W);
	parallel_ranges(nR, [&](size_t b, size_t e, unsigned w) {
		size_t at = kept + start[w];
		for (size_t i = b; i < e; ++i) if (cur[i] != i) merge_pairs[at++] = {ids_dense[i], ids_dense[cur[i]]};
	}, 100000, W);
	if (any_merge) reaggregate_after_merge();
}

// Unions of the merged cells' molecule sets: re-key, re-sort, re-reduce (Gene::merge, Gene.cpp:26-36).
void dropest_ctx::reaggregate_after_merge() {
	invalidate_prefetch();   // a cm_raw prefetch in flight reads the tables this call rewrites
	HostStage hs(this, "cb_merge:reaggregate");
	if (getenv("DROPEST_MP_TRACE")) {   // the cells that RECEIVE rows and the share of the molecule table they hold (NOTES_r06 section 4)
		std::unordered_set<u32> targets;
		for (auto const &pr : merge_pairs) targets.insert(u32(pr.second));
		uint64_t mol_targets = 0;
		for (auto const &r : real) if (targets.count(r.id)) mol_targets += uint64_t(std::max(0, r.row.total_umis));
		fprintf(stderr, "[mp] %zu merged cells into %zu targets; the targets hold %llu of %u molecule rows (%.1f %%)\n", merge_pairs.size(), targets.size(),
		        (unsigned long long)mol_targets, n_mol, 100.0 * double(mol_targets) / std::max<u32>(1u, n_mol));
	}
	remap.ensure(n_cells);
	{
		std::vector<u32> &src = ms.src, &tgt = ms.tgt32;
		src.resize(merge_pairs.size()); tgt.resize(merge_pairs.size());
		parallel_ranges(merge_pairs.size(), [&](size_t b, size_t e, unsigned) {
			for (size_t i = b; i < e; ++i) { src[i] = u32(merge_pairs[i].first); tgt[i] = u32(merge_pairs[i].second); }
		});
		DevBuf<u32> &d_src = ms.d_src, &d_tgt = ms.d_tgt; d_src.ensure(std::max<size_t>(src.size(), 1)); d_tgt.ensure(std::max<size_t>(tgt.size(), 1));
		upload(d_src.p, src.data(), src.size() * 4);
		upload(d_tgt.p, tgt.data(), tgt.size() * 4);
		hipLaunchKernelGGL(iota_kernel, dim3(div_up(n_cells, 256)), dim3(256), 0, stream, remap.p, n_cells);
		if (!src.empty())
			hipLaunchKernelGGL(scatter_pairs_kernel, dim3(div_up(u32(src.size()), 256)), dim3(256), 0, stream, d_src.p, d_tgt.p,
			                   u32(src.size()), remap.p);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(stream_wait(stream));   // src / tgt are host vectors
	}
	if (have_qual && qual_len && n_mol) {   // which member's quality sums a folded molecule keeps (quality.h)
		if (merge_rank.size() != n_cells) merge_rank.assign(n_cells, 0);
		DevBuf<u32> d_rank; d_rank.alloc(n_cells);
		HIP_CHECK(hipMemcpyAsync(d_rank.p, merge_rank.data(), size_t(n_cells) * 4, hipMemcpyHostToDevice, stream));
		reagg_prio_buf.ensure(n_mol);
		hipLaunchKernelGGL(prio_from_cell_kernel, dim3(div_up(n_mol, 256)), dim3(256), 0, stream, mol_key.p, n_mol,
		                   layout.gene_bits + layout.umi_bits, d_rank.p, reagg_prio_buf.p);
		HIP_CHECK(hipGetLastError());
		if (reagg_import_prio && reagg_import_n && reagg_import_from + reagg_import_n <= n_mol)   // rows a sharded merge brought in: keyed to their TARGET already
			HIP_CHECK(hipMemcpyAsync(reagg_prio_buf.p + reagg_import_from, reagg_import_prio, size_t(reagg_import_n) * 4, hipMemcpyDeviceToDevice, stream));
		reagg_import_prio = nullptr; reagg_import_n = 0;
		HIP_CHECK(stream_wait(stream));
		reagg_prio = reagg_prio_buf.p;
	}
	keys_a.ensure(n_mol); keys_b.ensure(n_mol); vals_a.ensure(n_mol); vals_b.ensure(n_mol);
	scalars.ensure(16);
	u64 init[2] = {0ull, ~0ull};
	u64 *d_or_and = reinterpret_cast<u64 *>(scalars.p + 4);
	HIP_CHECK(hipMemcpyAsync(d_or_and, init, 16, hipMemcpyHostToDevice, stream));
	// Only the cell field of a key changes: when few rows change, the new keys are made on the fly by the split kernels and the re-keyed
	// array is never written (k_mergepath.h: MpRekey) -- at C3 size the rekey kernel was 1.7 ms, the arrays 6.4 GB of traffic.
	{
		u64 varying = 0;
		if (resort_changed_rows(0, remap.p, layout.gene_bits + layout.umi_bits, &varying)) { reaggregate_from_keys(varying, true); return; }
		HIP_CHECK(hipMemcpyAsync(d_or_and, init, 16, hipMemcpyHostToDevice, stream));   // (the attempt used the words)
	}
	const u32 blocks = std::min<u32>(div_up(n_mol, 256), 4096u);
	timed("rekey_molecules", double(n_mol) * 28, [&] {
		hipLaunchKernelGGL(rekey_molecules_kernel, dim3(blocks), dim3(256), 0, stream, mol_key.p, n_mol,
		                   layout.gene_bits + layout.umi_bits, remap.p, keys_a.p, vals_a.p, d_or_and);
	});
	u64 or_and[2];
	HIP_CHECK(hipMemcpyAsync(or_and, d_or_and, 16, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
	reaggregate_from_keys(or_and[0] ^ or_and[1]);
}

// keys_a holds the re-keyed molecule keys, vals_a the row each one came from: sort, fold equal keys (read counts
// add, marks OR: Gene::merge / UMI::merge, Gene.cpp:26-58, UMI.cpp:15-19), rebuild the (cell, gene) and cell levels.
// keys_a holds the new key of every molecule row: when only a small part of them differs from the current (sorted) keys,
// split / sort the changed part / merge (k_mergepath.h) instead of radix-sorting everything.  Returns false when the
// radix sort should run (many changes); else leaves the sorted (key, old row) pairs in keys_a / vals_a.
bool dropest_ctx::resort_changed_rows(u64 varying_mask, const u32 *d_remap, int cell_shift, u64 *varying_out) {
	if (n_mol < 4 * u32(MP_TILE)) return false;
	const u32 tiles = div_up(n_mol, u32(MP_TILE));
	tile_counts.ensure(tiles); tile_prefix.ensure(tiles); scalars.ensure(16);
	const MpRekey rk{keys_a.p, d_remap, cell_shift};
	u64 *d_or_and = d_remap ? reinterpret_cast<u64 *>(scalars.p + 4) : nullptr;   // (reaggregate_after_merge initialised the two words)
	timed("mp_split_count", double(n_mol) * (d_remap ? 12 : 16), [&] {
		hipLaunchKernelGGL(mp_split_count_kernel, dim3(tiles), dim3(MP_THREADS), 0, stream, mol_key.p, rk, n_mol, mol_sorted_rows, tile_counts.p,
		                   reinterpret_cast<unsigned long long *>(d_or_and));
		scan_counts(tile_counts.p, tile_prefix.p, tiles, scalars.p);
	});
	u32 head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	fetch(head, scalars.p, 32);
	const u32 nb = head[0];
	if (d_remap) {
		u64 or_and[2];
		std::memcpy(or_and, head + 4, 16);
		varying_mask = or_and[0] ^ or_and[1];
		if (varying_out) *varying_out = varying_mask;
	}
	if (getenv("DROPEST_MP_TRACE")) {   // how the changed rows spread over the tiles of 4 096 (what "re-aggregate only what a merge changed" could skip: NOTES_r06 section 4)
		std::vector<u32> per_tile(tiles);
		fetch(per_tile.data(), tile_counts.p, size_t(tiles) * 4);
		size_t touched = 0;
		for (u32 c : per_tile) touched += c != 0;
		fprintf(stderr, "[mp] %u molecule rows in %u tiles of %u: %u rows change their key (%.1f %%), in %zu tiles (%.1f %% of the tiles)\n", n_mol, tiles, u32(MP_TILE), nb,
		        100.0 * nb / std::max<u32>(1u, n_mol), touched, 100.0 * double(touched) / std::max<u32>(1u, tiles));
	}
	if (nb == 0 || nb > n_mol / 4) return false;
	const u32 na = n_mol - nb;
	mp_bk.ensure(nb); mp_bk2.ensure(nb); mp_bv.ensure(nb); mp_bv2.ensure(nb);
	timed("mp_split_write", double(n_mol) * 28, [&] {
		hipLaunchKernelGGL(mp_split_write_kernel, dim3(tiles), dim3(MP_THREADS), 0, stream, mol_key.p, rk, n_mol, mol_sorted_rows, tile_prefix.p,
		                   keys_b.p, vals_b.p, mp_bk.p, mp_bv.p);
	});
	u64 *bk = mp_bk.p, *bk_alt = mp_bk2.p;
	u32 *bv = mp_bv.p, *bv_alt = mp_bv2.p;
	radix_sort(bk, bv, bk_alt, bv_alt, nb, varying_mask, 4, "changed:");
	const u32 out_tiles = div_up(n_mol, u32(MP_TILE));
	mp_astart.ensure(size_t(out_tiles) + 1);
	timed("mp_merge", double(n_mol) * 24, [&] {
		hipLaunchKernelGGL(mp_partition_kernel, dim3(div_up(out_tiles + 1, 256u)), dim3(256), 0, stream, keys_b.p, na, bk, nb, out_tiles, mp_astart.p);
		hipLaunchKernelGGL(mp_merge_kernel, dim3(out_tiles), dim3(MP_THREADS), 0, stream, keys_b.p, vals_b.p, na, bk, bv, nb, mp_astart.p,
		                   keys_a.p, vals_a.p);
	});
	HIP_CHECK(hipGetLastError());
	return true;
}

void dropest_ctx::reaggregate_from_keys(u64 varying_mask, bool sorted_already) {
	invalidate_prefetch();   // a cm_raw prefetch in flight reads the tables this call rewrites
	u64 *keys = keys_a.p, *keys_alt = keys_b.p;
	u32 *vals = vals_a.p, *vals_alt = vals_b.p;
	if (!sorted_already && !resort_changed_rows(varying_mask)) radix_sort(keys, vals, keys_alt, vals_alt, n_mol, varying_mask);
	u32 new_n = 0;
	// fold + (cell, gene) rows in one pass over the sorted pairs (k_ssort.h: ss_compact_cg_kernel<true>) instead of seg_count + seg_reduce twice
	const bool fused_fold = getenv("DROPEST_NO_FUSED_FOLD") == nullptr && n_mol != 0;
	if (fused_fold) {
		constexpr u32 T = 4096;
		const u32 tiles = div_up(n_mol, T), n_chunks = div_up(tiles, 1024u);
		ss_n_loc.ensure(tiles); ss_prefix.ensure(tiles); ss_cg_cnt.ensure(tiles); ss_cg_prefix.ensure(tiles); ss_chunk.ensure(1024); scalars.ensure(16);
		timed("fold_counts", double(n_mol) * 8, [&] {
			hipLaunchKernelGGL(rk_counts_kernel, dim3(div_up(tiles, 4u)), dim3(256), 0, stream, keys, n_mol, T, layout.umi_bits, tiles, ss_n_loc.p, ss_cg_cnt.p);
			hipLaunchKernelGGL(ss_chunk_sums_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, ss_n_loc.p, tiles, ss_chunk.p);
			hipLaunchKernelGGL(ss_prefix_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, ss_n_loc.p, tiles, ss_chunk.p, n_chunks, ss_prefix.p, scalars.p + 1);
			hipLaunchKernelGGL(ss_chunk_sums_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, ss_cg_cnt.p, tiles, ss_chunk.p);
			hipLaunchKernelGGL(ss_prefix_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, ss_cg_cnt.p, tiles, ss_chunk.p, n_chunks, ss_cg_prefix.p, scalars.p + 2);
		});
		u32 totals[2] = {0, 0};
		fetch(totals, scalars.p + 1, 8);
		new_n = totals[0];
		const u32 new_cg = totals[1];
		const size_t cap = std::max<size_t>(size_t(new_n) + 1, mol_key.n);   // (the twins keep the capacity of the tables they are swapped with, see below)
		mol_key2.ensure(cap); mol_reads2.ensure(cap); mol_mark2.ensure(cap);
		if (chr_from_gene) { mol_exon2.ensure(cap); mol_intron2.ensure(cap); }
		cg_key.ensure(size_t(new_cg) + 1); cg_mol_begin.ensure(size_t(new_cg) + 1);
		for (DevBuf<u32> *b : {&cg_n_all, &cg_n_req, &cg_reads_all, &cg_reads_req, &cg_exon, &cg_intron}) b->ensure(size_t(new_cg) + 1);
		SsCompactCgArgs g{};
		g.prefix = ss_prefix.p; g.cg_cnt = ss_cg_cnt.p; g.cg_prefix = ss_cg_prefix.p; g.n_buckets = tiles;
		g.rk_key = keys; g.rk_idx = vals; g.n_rows = n_mol; g.tile_rows = T;
		g.old_reads = mol_reads.p; g.old_mark = mol_mark.p;
		if (chr_from_gene) { g.old_exon = mol_exon.p; g.old_intron = mol_intron.p; g.mol_exon = mol_exon2.p; g.mol_intron = mol_intron2.p; }
		g.mol_key = mol_key2.p; g.mol_reads = mol_reads2.p; g.mol_mark = mol_mark2.p;
		g.cg_key = cg_key.p; g.cg_mol_begin = cg_mol_begin.p;
		g.out[0] = cg_n_all.p; g.out[1] = cg_n_req.p; g.out[2] = cg_reads_all.p; g.out[3] = cg_reads_req.p; g.out[4] = cg_exon.p; g.out[5] = cg_intron.p;
		g.umi_bits = layout.umi_bits; g.query_mask = query_mask; g.n_cg = new_cg;
		timed("fold:cell_gene", double(n_mol) * (12 + (chr_from_gene ? 16 : 8)) + double(new_n) * (chr_from_gene ? 24 : 16) + double(new_cg) * 36, [&] {
			hipLaunchKernelGGL(ss_cg_zero_borders_kernel, dim3(div_up(tiles, 256u)), dim3(256), 0, stream, g);
			hipLaunchKernelGGL(ss_compact_cg_kernel<true>, dim3(div_up(tiles, 4u)), dim3(256), 0, stream, g);
		});
		HIP_CHECK(hipMemcpyAsync(cg_mol_begin.p + new_cg, &new_n, 4, hipMemcpyHostToDevice, stream));   // row i owns molecules [cg_mol_begin[i], cg_mol_begin[i + 1])
		for (DevBuf<u32> *b : {&mol_reads2, &mol_mark2}) HIP_CHECK(hipMemsetAsync(b->p + new_n, 0, 4, stream));   // sentinel row
		if (chr_from_gene) for (DevBuf<u32> *b : {&mol_exon2, &mol_intron2}) HIP_CHECK(hipMemsetAsync(b->p + new_n, 0, 4, stream));
		HIP_CHECK(stream_wait(stream));   // (new_n is a stack word)
		if (chr_from_gene) { std::swap(mol_exon, mol_exon2); std::swap(mol_intron, mol_intron2); }
		n_cg = new_cg;
	} else if (chr_from_gene) {
		RekeyedToMoleculesX p{};
		p.keys = keys; p.idx = vals; p.old_reads = mol_reads.p; p.old_mark = mol_mark.p; p.old_exon = mol_exon.p; p.old_intron = mol_intron.p;
		new_n = run_segmented_reduce(*this, "molecules_rekeyed", p, n_mol, 12 + 16, [&](u32 total) {
			// the tables are swapped with their twins below: give the twins the same capacity, or the next pass's first
			// reduce finds a smaller buffer and pays a hipFree + hipMalloc of gigabytes (measured: +300 ms at 1e9 reads)
			const size_t cap = std::max<size_t>(size_t(total) + 1, mol_key.n);
			mol_key2.ensure(cap);
			for (DevBuf<u32> *b : {&mol_reads2, &mol_mark2, &mol_exon2, &mol_intron2}) b->ensure(cap);
			p.mol_key = mol_key2.p; p.out[0] = mol_reads2.p; p.out[1] = mol_mark2.p; p.out[2] = mol_exon2.p; p.out[3] = mol_intron2.p;
		});
		HIP_CHECK(stream_wait(stream));
		std::swap(mol_exon, mol_exon2); std::swap(mol_intron, mol_intron2);
	} else {
		RekeyedToMolecules p{};
		p.keys = keys; p.idx = vals; p.old_reads = mol_reads.p; p.old_mark = mol_mark.p;
		new_n = run_segmented_reduce(*this, "molecules_rekeyed", p, n_mol, 12 + 8, [&](u32 total) {
			const size_t cap = std::max<size_t>(size_t(total) + 1, mol_key.n);
			mol_key2.ensure(cap); mol_reads2.ensure(cap); mol_mark2.ensure(cap);
			p.mol_key = mol_key2.p; p.out[0] = mol_reads2.p; p.out[1] = mol_mark2.p;
		});
		HIP_CHECK(stream_wait(stream));
	}
	mol_sorted_rows = 0xFFFFFFFFu;   // the folded table is sorted throughout
	requality_after_fold(keys, vals, n_mol, mol_key2.p, new_n);
	std::swap(mol_key, mol_key2); std::swap(mol_reads, mol_reads2); std::swap(mol_mark, mol_mark2);
	n_mol = new_n;
	if (!fused_fold) reduce_molecules_to_cell_gene();
	reduce_cell_gene_to_cells();
	HIP_CHECK(stream_wait(stream));
	refresh_real_rows();
}

// Re-reads the device sizes of the real-candidate cells; the int stats (TOTAL_READS / TOTAL_UMIS) and the flags
// are host-tracked through the merges (Stats::merge sums, it does not recount) and are kept.
void dropest_ctx::refresh_real_rows() {
	invalidate_prefetch();
	const u32 count = u32(real.size());
	if (!count) return;
	HostStage hs(this, "refresh_real_rows");
	// the ids of the real-candidate cells sit on the device since fetch_real_cells (real_list; compute_merge_targets re-uploads the
	// same list): only the three sizes a merge changes come back, 12 bytes per cell
	if (!real_list_current) {
		std::vector<u32> ids(count);
		parallel_ranges(count, [&](size_t b, size_t e, unsigned) { for (size_t i = b; i < e; ++i) ids[i] = real[i].id; });
		real_list.ensure(count);
		HIP_CHECK(hipMemcpyAsync(real_list.p, ids.data(), size_t(count) * 4, hipMemcpyHostToDevice, stream));
		HIP_CHECK(stream_wait(stream));
		real_list_current = true;
	}
	sizes_dev.ensure(size_t(count) * 3);
	CellArrays a{cell_cb.p, cell_first.p, cell_n_genes.p, cell_req_genes.p, cell_req_umis.p, cell_total_umis.p, cell_total_reads.p};
	hipLaunchKernelGGL(gather_cell_sizes_kernel, dim3(div_up(count, 256)), dim3(256), 0, stream, a, real_list.p, count, sizes_dev.p);
	HIP_CHECK(hipGetLastError());
	h_stage.ensure(size_t(count) * 12);
	HIP_CHECK(hipMemcpyAsync(h_stage.p, sizes_dev.p, size_t(count) * 12, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
	real_pristine = false;
	const u32 *sz = reinterpret_cast<const u32 *>(h_stage.p);
	parallel_ranges(count, [&](size_t b, size_t e, unsigned) {
		for (size_t i = b; i < e; ++i) {
			if (real[i].merged) continue;   // the reference keeps a merged source's stale sizes; nothing reads them again
			real[i].row.n_genes = sz[3 * i]; real[i].row.requested_genes = sz[3 * i + 1]; real[i].row.requested_umis = sz[3 * i + 2];
		}
	});
}
