// merge_host.h -- host orchestration of the whitelist CB merge (included by dropest_amd.hip).
//
// MergeStrategyAbstract::merge -> MergeStrategyBase::merge_inited (Estimation/Merge/MergeStrategyBase.cpp:11-57):
//   pass 1 (read-only, on the device): a target for every filtered cell on the UNMERGED state
//   pass 2 (sequential, here): smallest-first application with exclusion, chasing of re-targeted cells
//          (reassign, :64-82), Stats::merge bookkeeping (Stats.cpp:29-43)
//   then the molecule table is re-keyed and re-reduced on the device (= the unions done by Gene::merge).
#pragma once

namespace {

struct PartDistH { size_t index; long value; };            // Tools/IndexedValue.h
struct ComboH { size_t i0, i1; unsigned ed; };

// Reference ORDER of the candidate list for one cell, rebuilt from the device-computed per-part distances with
// the same std::sort calls on the same sequences (BarcodesParser.cpp:21-74, RealBarcodesMergeStrategy.cpp:63-109).
// Only needed when the arg-max of the merge fraction is tied (the reference's result then depends on this order).
std::vector<u32> reference_candidate_order(const dropest::Whitelist &wl, const uint8_t *dist,
                                           const std::unordered_map<u64, u32> &qualifying_by_code) {
	std::vector<std::vector<PartDistH>> d(2);
	size_t off = 0;
	for (int p = 0; p < 2; ++p) {
		for (size_t i = 0; i < wl.parts[size_t(p)].size(); ++i) d[size_t(p)].push_back(PartDistH{i, long(dist[off + i])});
		std::sort(d[size_t(p)].begin(), d[size_t(p)].end(), [](const PartDistH &x, const PartDistH &y) { return x.value < y.value; });
		off += wl.parts[size_t(p)].size();
	}
	std::vector<ComboH> combos;
	for (const PartDistH &a : d[0]) {
		if (unsigned(a.value) > unsigned(dropest::WL_MAX_DIST)) break;
		for (const PartDistH &b : d[1]) {
			const unsigned ed = unsigned(a.value) + unsigned(b.value);
			if (ed > unsigned(dropest::WL_MAX_DIST)) break;
			combos.push_back(ComboH{a.index, b.index, ed});
		}
	}
	std::vector<u32> out;
	if (combos.empty()) return out;
	std::sort(combos.begin(), combos.end(), [](const ComboH &x, const ComboH &y) { return x.ed < y.ed; });
	unsigned max_dist = combos.front().ed;
	for (const ComboH &c : combos) {
		if (c.ed > max_dist && !out.empty()) break;
		u64 code = 0;
		if (dropest::encode_code(wl.parts[0][c.i0] + wl.parts[1][c.i1], code)) {
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
	for (int p = 0; p < 2; ++p) {
		if (d_wl[p].p) continue;
		std::vector<WlEntry> h(wl.parts[size_t(p)].size());
		for (size_t i = 0; i < h.size(); ++i) {
			std::memset(h[i].seq, 0, sizeof(h[i].seq));
			std::memcpy(h[i].seq, wl.parts[size_t(p)][i].data(), wl.parts[size_t(p)][i].size());
		}
		d_wl[p].alloc(h.size());
		HIP_CHECK(hipMemcpy(d_wl[p].p, h.data(), h.size() * sizeof(WlEntry), hipMemcpyHostToDevice));
	}
}

// RealBarcodesMergeStrategy::get_merge_target for a list of cells, on the current (unmerged) device state.
std::vector<long> dropest_ctx::compute_merge_targets(const std::vector<u32> &cells) {
	std::vector<long> targets(cells.size(), -1);
	if (cells.empty()) return targets;
	upload_whitelist();
	const u32 F = u32(cells.size());

	std::vector<WlBase> bases(F);
	for (u32 f = 0; f < F; ++f) {
		const long ri = real_find(cells[f]);
		if (ri < 0) throw InvalidError("merge target requested for a cell below min_genes_before_merge");
		std::string a, b;
		wl.split(barcode_of(real[size_t(ri)]), a, b);
		WlBase &wb = bases[f];
		std::memset(&wb, 0, sizeof(wb));
		std::memcpy(wb.part[0], a.data(), a.size()); std::memcpy(wb.part[1], b.data(), b.size());
		wb.len[0] = uint8_t(a.size()); wb.len[1] = uint8_t(b.size());
		wb.cell = cells[f];
	}
	DevBuf<WlBase> d_bases; d_bases.alloc(F);
	DevBuf<u32> d_cnt, d_lvl, d_cand;
	d_cnt.alloc(F); d_lvl.alloc(F); d_cand.alloc(size_t(F) * WL_CAND_CAP);
	HIP_CHECK(hipMemcpyAsync(d_bases.p, bases.data(), size_t(F) * sizeof(WlBase), hipMemcpyHostToDevice, stream));
	WlArgs a{};
	a.bases = d_bases.p; a.n_bases = F;
	a.part[0] = d_wl[0].p; a.part[1] = d_wl[1].p;
	a.part_size[0] = u32(wl.parts[0].size()); a.part_size[1] = u32(wl.parts[1].size());
	a.table = table; a.cell_n_genes = cell_n_genes.p; a.cell_total_umis = cell_total_umis.p; a.min_genes = min_before;
	a.cand_count = d_cnt.p; a.cand_level = d_lvl.p; a.cand_cell = d_cand.p; a.dist_dump = nullptr;
	const u32 ntot = a.part_size[0] + a.part_size[1];
	const size_t lds = ((ntot + 15u) & ~15u) + size_t(ntot) * 2;
	timed("wl_neighbours", double(F) * ntot * 32, [&] {
		hipLaunchKernelGGL(wl_neighbours_kernel, dim3(F), dim3(WL_THREADS), lds, stream, a);
	});
	std::vector<u32> cnt(F), cand(size_t(F) * WL_CAND_CAP);
	HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt.p, size_t(F) * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(cand.data(), d_cand.p, cand.size() * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));

	// pairs (base, candidate) whose UMI-gene intersection is needed
	std::vector<u32> pair_base, pair_cand, pair_first(F + 1, 0);
	for (u32 f = 0; f < F; ++f) {
		pair_first[f] = u32(pair_base.size());
		if (cnt[f] > u32(WL_CAND_CAP))
			throw UnsupportedError("more than " + std::to_string(WL_CAND_CAP) + " merge candidates for one barcode");
		bool self = false;
		for (u32 k = 0; k < cnt[f]; ++k) self |= cand[size_t(f) * WL_CAND_CAP + k] == cells[f];
		if (self) continue;   // the base is itself a whitelist barcode: neighbour_cells[0] == base (RealBarcodesMergeStrategy.cpp:34-35)
		for (u32 k = 0; k < cnt[f]; ++k) { pair_base.push_back(cells[f]); pair_cand.push_back(cand[size_t(f) * WL_CAND_CAP + k]); }
	}
	pair_first[F] = u32(pair_base.size());
	const u32 NP = u32(pair_base.size());
	std::vector<u32> inter(NP);
	if (NP) {
		DevBuf<u32> d_pb, d_pc, d_inter; DevBuf<PairRange> d_pr;
		d_pb.alloc(NP); d_pc.alloc(NP); d_inter.alloc(NP); d_pr.alloc(NP);
		HIP_CHECK(hipMemcpyAsync(d_pb.p, pair_base.data(), size_t(NP) * 4, hipMemcpyHostToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(d_pc.p, pair_cand.data(), size_t(NP) * 4, hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(pair_ranges_kernel, dim3(div_up(NP, 256)), dim3(256), 0, stream, d_pb.p, d_pc.p, NP, cell_cg_begin.p,
		                   cell_cg_count.p, cg_mol_begin.p, d_pr.p);
		HIP_CHECK(hipGetLastError());
		const int low_bits = layout.gene_bits + layout.umi_bits;
		timed("umig_intersect", double(NP) * 64, [&] {
			hipLaunchKernelGGL(umig_intersect_kernel, dim3(NP), dim3(256), 0, stream, d_pr.p, NP, mol_key.p, (1ull << low_bits) - 1ull,
			                   layout.umi_bits, layout.gene_none, d_inter.p);
		});
		HIP_CHECK(hipMemcpyAsync(inter.data(), d_inter.p, size_t(NP) * 4, hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
	}

	// decisions (RealBarcodesMergeStrategy::get_best_merge_target, :31-61)
	std::vector<u32> need_order;   // cells whose result depends on the reference's candidate order
	auto umis_of = [&](u32 cell) { return size_t(real[real_at(cell)].row.total_umis); };
	auto frac_of = [&](u32 base, u32 other, u32 n) { return 0.5 * n * (1. / umis_of(base) + 1. / umis_of(other)); };
	for (u32 f = 0; f < F; ++f) {
		if (cnt[f] == 0) { targets[f] = -1; continue; }
		const u32 p0 = pair_first[f], p1 = pair_first[f + 1];
		if (p0 == p1) { targets[f] = long(cells[f]); continue; }   // self
		double best = 0; u32 n_best = 0, best_cell = 0;
		for (u32 p = p0; p < p1; ++p) {
			const double fr = frac_of(cells[f], pair_cand[p], inter[p]);
			if (fr > best) { best = fr; n_best = 1; best_cell = pair_cand[p]; }
			else if (fr == best) ++n_best;
		}
		if (best < cfg.min_merge_fraction) { targets[f] = -1; continue; }   // holds for any order
		if (best > 0 && n_best == 1) { targets[f] = long(best_cell); continue; }
		need_order.push_back(f);   // tie at the maximum (or all fractions zero with a non-positive threshold)
	}
	if (!need_order.empty()) {
		const u32 R = u32(need_order.size());
		std::vector<WlBase> rb(R);
		for (u32 r = 0; r < R; ++r) rb[r] = bases[need_order[r]];
		DevBuf<WlBase> d_rb; d_rb.alloc(R);
		DevBuf<uint8_t> d_dump; d_dump.alloc(size_t(R) * ntot);
		DevBuf<u32> d_c2, d_l2, d_k2; d_c2.alloc(R); d_l2.alloc(R); d_k2.alloc(size_t(R) * WL_CAND_CAP);
		HIP_CHECK(hipMemcpyAsync(d_rb.p, rb.data(), size_t(R) * sizeof(WlBase), hipMemcpyHostToDevice, stream));
		WlArgs a2 = a;
		a2.bases = d_rb.p; a2.n_bases = R; a2.cand_count = d_c2.p; a2.cand_level = d_l2.p; a2.cand_cell = d_k2.p; a2.dist_dump = d_dump.p;
		hipLaunchKernelGGL(wl_neighbours_kernel, dim3(R), dim3(WL_THREADS), lds, stream, a2);
		HIP_CHECK(hipGetLastError());
		std::vector<uint8_t> dump(size_t(R) * ntot);
		HIP_CHECK(hipMemcpyAsync(dump.data(), d_dump.p, dump.size(), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		for (u32 r = 0; r < R; ++r) {
			const u32 f = need_order[r];
			std::unordered_map<u64, u32> by_code;
			std::unordered_map<u32, u32> inter_of;
			for (u32 p = pair_first[f]; p < pair_first[f + 1]; ++p) {
				by_code[real[real_at(pair_cand[p])].row.barcode] = pair_cand[p];
				inter_of[pair_cand[p]] = inter[p];
			}
			const std::vector<u32> order = reference_candidate_order(wl, dump.data() + size_t(r) * ntot, by_code);
			if (order.empty()) throw DeviceError("internal: candidate replay found no candidate");
			double best = 0; u32 best_cell = order[0];
			for (u32 c : order) {
				const double fr = frac_of(cells[f], c, inter_of.at(c));
				if (best < fr) { best = fr; best_cell = c; }
			}
			targets[f] = best < cfg.min_merge_fraction ? -1 : long(best_cell);
		}
	}
	return targets;
}

void dropest_ctx::run_cb_merge_real() {
	const std::vector<uint64_t> &order = filtered_cells();
	std::vector<u32> cells(order.begin(), order.end());
	const std::vector<long> targets = compute_merge_targets(cells);

	// MergeStrategyBase::merge_inited second loop (:30-51) + reassign (:64-82)
	std::unordered_map<u32, std::unordered_set<u32>> reassigned_to;
	reassign.clear();
	auto current = [&](u32 c) { auto it = reassign.find(c); return it == reassign.end() ? c : it->second; };
	bool any_merge = false;
	for (size_t i = 0; i < cells.size(); ++i) {
		const u32 base = cells[i];
		HostCell &hb = real[real_at(base)];
		long t = targets[i];
		if (t < 0) { hb.excluded = true; continue; }
		u32 tgt = current(u32(t));
		if (tgt == base) continue;
		HostCell &ht = real[real_at(tgt)];
		// CellsDataContainer::merge_cells (:90-104): Stats::merge adds every counter, TOTAL_UMIS included
		ht.row.total_reads += hb.row.total_reads;
		ht.row.total_umis += hb.row.total_umis;
		hb.merged = true;
		any_merge = true;
		reassign[base] = tgt;
		reassigned_to[tgt].insert(base);
		auto it = reassigned_to.find(base);
		if (it != reassigned_to.end()) {
			for (u32 moved : it->second) { reassign[moved] = tgt; reassigned_to[tgt].insert(moved); }
			reassigned_to.find(base)->second.clear();
		}
	}
	merge_pairs.clear();
	for (auto &kv : reassign) merge_pairs.emplace_back(kv.first, kv.second);
	std::sort(merge_pairs.begin(), merge_pairs.end());
	if (any_merge) reaggregate_after_merge();
}

// Unions of the merged cells' molecule sets: re-key, re-sort, re-reduce (Gene::merge, Gene.cpp:26-36).
void dropest_ctx::reaggregate_after_merge() {
	std::vector<u32> h_remap(n_cells);
	for (u32 i = 0; i < n_cells; ++i) h_remap[i] = i;
	for (auto &kv : reassign) h_remap[kv.first] = kv.second;
	remap.ensure(n_cells);
	HIP_CHECK(hipMemcpyAsync(remap.p, h_remap.data(), size_t(n_cells) * 4, hipMemcpyHostToDevice, stream));
	keys_a.ensure(n_mol); keys_b.ensure(n_mol); vals_a.ensure(n_mol); vals_b.ensure(n_mol);
	scalars.ensure(16);
	u64 init[2] = {0ull, ~0ull};
	u64 *d_or_and = reinterpret_cast<u64 *>(scalars.p + 4);
	HIP_CHECK(hipMemcpyAsync(d_or_and, init, 16, hipMemcpyHostToDevice, stream));
	const u32 blocks = std::min<u32>(div_up(n_mol, 256), 4096u);
	timed("rekey_molecules", double(n_mol) * 28, [&] {
		hipLaunchKernelGGL(rekey_molecules_kernel, dim3(blocks), dim3(256), 0, stream, mol_key.p, n_mol,
		                   layout.gene_bits + layout.umi_bits, remap.p, keys_a.p, vals_a.p, d_or_and);
	});
	u64 or_and[2];
	HIP_CHECK(hipMemcpyAsync(or_and, d_or_and, 16, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	u64 *keys = keys_a.p, *keys_alt = keys_b.p;
	u32 *vals = vals_a.p, *vals_alt = vals_b.p;
	radix_sort(keys, vals, keys_alt, vals_alt, n_mol, or_and[0] ^ or_and[1]);
	RekeyedToMolecules p{};
	p.keys = keys; p.idx = vals; p.old_reads = mol_reads.p; p.old_mark = mol_mark.p;
	const u32 new_n = run_segmented_reduce(*this, "molecules_rekeyed", p, n_mol, 12 + 8, [&](u32 total) {
		mol_key2.ensure(total + 1); mol_reads2.ensure(total + 1); mol_mark2.ensure(total + 1);
		zero_async(*this, mol_reads2.p, size_t(total + 1) * 4); zero_async(*this, mol_mark2.p, size_t(total + 1) * 4);
		p.mol_key = mol_key2.p; p.out[0] = mol_reads2.p; p.out[1] = mol_mark2.p;
	});
	HIP_CHECK(hipStreamSynchronize(stream));
	std::swap(mol_key, mol_key2); std::swap(mol_reads, mol_reads2); std::swap(mol_mark, mol_mark2);
	n_mol = new_n;
	reduce_molecules_to_cell_gene();
	reduce_cell_gene_to_cells();
	HIP_CHECK(hipStreamSynchronize(stream));
	refresh_real_rows();
}

// Re-reads the device sizes of the real-candidate cells; the int stats (TOTAL_READS / TOTAL_UMIS) and the flags
// are host-tracked through the merges (Stats::merge sums, it does not recount) and are kept.
void dropest_ctx::refresh_real_rows() {
	const u32 count = u32(real.size());
	if (!count) return;
	std::vector<u32> ids(count);
	for (u32 i = 0; i < count; ++i) ids[i] = real[i].id;
	real_list.ensure(count); real_rows_dev.ensure(count);
	HIP_CHECK(hipMemcpyAsync(real_list.p, ids.data(), size_t(count) * 4, hipMemcpyHostToDevice, stream));
	CellArrays a{cell_cb.p, cell_first.p, cell_n_genes.p, cell_req_genes.p, cell_req_umis.p, cell_total_umis.p, cell_total_reads.p};
	hipLaunchKernelGGL(gather_cell_rows_kernel, dim3(div_up(count, 256)), dim3(256), 0, stream, a, real_list.p, 0u, count,
	                   real_rows_dev.p);
	HIP_CHECK(hipGetLastError());
	std::vector<CellRowPod> rows(count);
	HIP_CHECK(hipMemcpyAsync(rows.data(), real_rows_dev.p, size_t(count) * sizeof(CellRowPod), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	for (u32 i = 0; i < count; ++i) {
		if (real[i].merged) continue;   // the reference keeps a merged source's stale sizes; nothing reads them again
		real[i].row.n_genes = rows[i].n_genes;
		real[i].row.requested_genes = rows[i].requested_genes;
		real[i].row.requested_umis = rows[i].requested_umis;
	}
}
