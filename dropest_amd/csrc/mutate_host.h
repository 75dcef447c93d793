// mutate_host.h -- the container's public mutators, which the reference's strategies (and its tests) call directly:
// CellsDataContainer::exclude_cell (CellsDataContainer.cpp:106-109), merge_cells (:90-104), merge_umis (:209-213 ->
// Cell::merge_umis, Cell.cpp:31-42 -> Gene::merge(src, tgt), Gene.cpp:38-58).  They act on the built (initialised) state
// with the machinery of the strategies: a cell merge is one more (source, target) pair re-aggregated on the device, a
// UMI merge rewrites one (cell, gene) group through the host override map.  Included by dropest_amd.hip.
#pragma once

void dropest_ctx::mutate_exclude_cell(u32 cell) {
	invalidate_prefetch();   // a cm_raw prefetch in flight reads the tables this call rewrites
	invalidate_prefetch();
	const long ri = real_find(cell);
	if (ri < 0) { extra_excluded.insert(cell); return; }   // not a real-candidate cell: only the flag is observable
	real[size_t(ri)].excluded = true;
	request_filtered(filtered_threshold, filtered_max_cells);
}

void dropest_ctx::mutate_merge_cells(u32 src, u32 tgt) {
	invalidate_prefetch();   // a cm_raw prefetch in flight reads the tables this call rewrites
	using namespace dropest;
	if (src == tgt) throw InvalidError("merge_cells: source and target are the same cell");
	const long rs = real_find(src), rt = real_find(tgt);
	if (rs < 0 || rt < 0)
		throw UnsupportedError("merge_cells is supported between real-candidate cells (at least min_genes_before_merge genes at initialisation)");
	HostCell &s = real[size_t(rs)], &t = real[size_t(rt)];
	if (s.merged) throw InvalidError("merge_cells: the source cell was merged before");
	// Stats::merge adds every counter (Stats.cpp:29-43); the per-chromosome counters follow the molecules on the device
	t.row.total_reads += s.row.total_reads; t.row.total_umis += s.row.total_umis;
	s.merged = true;
	merge_pairs.emplace_back(src, tgt);
	std::sort(merge_pairs.begin(), merge_pairs.end());
	explicit_sources.insert(src);
	// cells merged into the source earlier stay where they are recorded; their molecules moved with the source's then
	if (have_qual) { merge_rank.assign(n_cells, 0); merge_rank[src] = 1; }
	reaggregate_after_merge();
	request_filtered(filtered_threshold, filtered_max_cells);
}

void dropest_ctx::mutate_merge_umis(u32 cell, u32 gene, uint64_t n, const uint64_t *src, const uint64_t *tgt) {
	invalidate_prefetch();   // a cm_raw prefetch in flight reads the tables this call rewrites
	using namespace dropest;
	if (gene >= layout.gene_none) throw RangeError("gene index out of range");
	// the (cell, gene) row
	u32 cgb = 0, cgc = 0;
	HIP_CHECK(hipMemcpy(&cgb, cell_cg_begin.p + cell, 4, hipMemcpyDeviceToHost));
	HIP_CHECK(hipMemcpy(&cgc, cell_cg_count.p + cell, 4, hipMemcpyDeviceToHost));
	std::vector<u64> keys(cgc);
	if (cgc) HIP_CHECK(hipMemcpy(keys.data(), cg_key.p + cgb, size_t(cgc) * 8, hipMemcpyDeviceToHost));
	const u64 want = (u64(cell) << layout.gene_bits) | gene;
	u32 row = 0xFFFFFFFFu;
	for (u32 j = 0; j < cgc; ++j) if (keys[j] == want) row = cgb + j;
	if (row == 0xFFFFFFFFu) throw RangeError("the cell has no such gene");   // genes_t::at throws std::out_of_range
	if (umi_overrides.count(want)) throw UnsupportedError("merge_umis on a group that the UMI merge strategy rewrote");
	const long ri = real_find(cell);
	if (ri < 0) throw UnsupportedError("merge_umis is supported on real-candidate cells");

	// the group's molecules; the pairs are replayed on them in the caller's order (Cell::merge_umis walks its map), each
	// molecule row ending up under the UMI it would belong to -- the re-keyed rows are then folded on the device, so
	// whatever runs later (the merge strategies) sees the result like any other molecule
	GatheredGroups GG;
	umi_gather_groups(std::vector<u32>{row}, GG, nullptr);
	const u64 umask = layout.umi_bits ? ((1ull << layout.umi_bits) - 1ull) : 0ull;
	std::map<u64, std::vector<u32>> rows_of;                 // API code -> molecule rows now under it
	std::unordered_map<u64, u64> device_code;                // API code -> UMI field of the key (existing molecules)
	for (u32 t = 0; t < GG.size[0]; ++t) {
		const u64 api = unmap_umi(GG.hk[t] & umask);
		rows_of[api].push_back(GG.begin[0] + t);
		device_code[api] = GG.hk[t] & umask;
	}
	int removed = 0;
	for (uint64_t i = 0; i < n; ++i) {
		if (src[i] == tgt[i]) continue;
		auto s = rows_of.find(src[i]);
		if (s == rows_of.end()) throw InvalidError("Source UMI doesn't belong to the gene: " + decode_code(src[i], side));
		std::vector<u32> moved = std::move(s->second);
		rows_of.erase(s);
		std::vector<u32> &dst = rows_of[tgt[i]];
		dst.insert(dst.end(), moved.begin(), moved.end());
		++removed;                                           // TOTAL_UMIS_PER_CB-- per pair (Cell.cpp:39)
	}
	if (!removed) return;
	// new key of every row that moved
	std::vector<std::pair<u32, u64>> patch;
	for (auto const &kv : rows_of) {
		u64 field;
		auto known = device_code.find(kv.first);
		if (known != device_code.end()) field = known->second;
		else {
			if (kv.first & ESCAPE_BIT) throw UnsupportedError("merge_umis to a new UMI that contains N");
			if (umi_sentinel_stripped && bit_length(kv.first) - 1 != umi_clean_bits) throw UnsupportedError("merge_umis to a UMI of another length");
			field = kv.first & layout.umi_strip_mask;
			if (field > umask || (ingest.umi_escape_max_plus1 && field >= layout.umi_escape_base))
				throw UnsupportedError("merge_umis to a UMI outside the key layout");
		}
		for (u32 r : kv.second) patch.emplace_back(r, (want << layout.umi_bits) | field);
	}
	keys_a.ensure(n_mol); keys_b.ensure(n_mol); vals_a.ensure(n_mol); vals_b.ensure(n_mol);
	HIP_CHECK(hipMemcpyAsync(keys_a.p, mol_key.p, size_t(n_mol) * 8, hipMemcpyDeviceToDevice, stream));
	HIP_CHECK(stream_wait(stream));
	for (auto const &pr : patch) HIP_CHECK(hipMemcpy(keys_a.p + pr.first, &pr.second, 8, hipMemcpyHostToDevice));
	scalars.ensure(16);
	u64 init[2] = {0ull, ~0ull};
	u64 *d_or_and = reinterpret_cast<u64 *>(scalars.p + 4);
	HIP_CHECK(hipMemcpyAsync(d_or_and, init, 16, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(iota_or_and_kernel, dim3(std::min<u32>(div_up(n_mol, 256), 4096u)), dim3(256), 0, stream, keys_a.p, n_mol, vals_a.p, d_or_and);
	if (have_qual && qual_len) {
		reagg_prio_buf.ensure(n_mol);
		hipLaunchKernelGGL(prio_from_rekey_kernel, dim3(div_up(n_mol, 256)), dim3(256), 0, stream, mol_key.p, keys_a.p, n_mol, reagg_prio_buf.p);
		reagg_prio = reagg_prio_buf.p;
	}
	HIP_CHECK(hipGetLastError());
	u64 or_and[2];
	fetch(or_and, d_or_and, 16);
	reaggregate_from_keys(or_and[0] ^ or_and[1]);
	real[size_t(ri)].row.total_umis -= removed;
	request_filtered(filtered_threshold, filtered_max_cells);
}
