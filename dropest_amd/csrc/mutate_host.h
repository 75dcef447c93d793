// mutate_host.h -- the container's public mutators, which the reference's strategies (and its tests) call directly:
// CellsDataContainer::exclude_cell (CellsDataContainer.cpp:106-109), merge_cells (:90-104), merge_umis (:209-213 ->
// Cell::merge_umis, Cell.cpp:31-42 -> Gene::merge(src, tgt), Gene.cpp:38-58).  They act on the built (initialised) state
// with the machinery of the strategies: a cell merge is one more (source, target) pair re-aggregated on the device, a
// UMI merge rewrites one (cell, gene) group through the host override map.  Included by dropest_amd.hip.
#pragma once

void dropest_ctx::mutate_exclude_cell(u32 cell) {
	invalidate_prefetch();   // a cm_raw prefetch in flight reads the tables this call rewrites
	const long ri = real_find(cell);
	if (ri < 0) { extra_excluded.insert(cell); return; }   // not a real-candidate cell: only the flag is observable
	real_pristine = false;
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
	real_pristine = false;
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
			if (!umi_dict_on && umi_sentinel_stripped && bit_length(kv.first) - 1 != umi_clean_bits) throw UnsupportedError("merge_umis to a UMI of another length");
			if (!map_umi_or_add(kv.first, field))
				throw UnsupportedError(umi_dict_on ? "merge_umis to a UMI outside the container's UMI dictionary" : "merge_umis to a UMI outside the key layout");
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
	real_pristine = false;
	real[size_t(ri)].row.total_umis -= removed;
	request_filtered(filtered_threshold, filtered_max_cells);
}

// CellsDataContainer::add_umi_to_cell (CellsDataContainer.cpp:356-364) on the initialised container: one more read of UMI
// `umi_code` for gene `gene` in cell `cell` -- Gene::add_umi: a new molecule (TOTAL_UMIS_PER_CB++) or read_count++ / mark |=
// of the existing one (UMI::add_read: the read's quality bytes are added to the molecule's sums, its length checked against
// them, UMI.cpp:21-34).  Like the reference's member it touches nothing else: no read counters, no chromosome statistics.
// The read becomes one row appended behind the sorted molecule table, folded by the machinery every merge uses.
void dropest_ctx::mutate_add_umi_to_cell(u32 cell, u32 gene, uint64_t umi_code, u32 mark, const uint8_t *quality, u32 quality_length) {
	invalidate_prefetch();
	using namespace dropest;
	const bool with_qual = have_qual && qual_len;
	// UMI.cpp:26-28: the length of a molecule's quality sums is fixed by the read that created it (checked below, once the row is known);
	// a NEW molecule may bring any length the rows of this container can hold
	if (have_qual && quality_length > qual_len)
		throw UnsupportedError("add_umi_to_cell: a quality string of " + std::to_string(quality_length) + " characters, the rows of this container hold " +
		                       std::to_string(qual_len));
	if (with_qual && quality_length && !quality) throw InvalidError("null quality string");
	if (!chr_from_gene) throw UnsupportedError("add_umi_to_cell needs the chromosome-from-gene record layout (a gene on two chromosomes was seen)");
	if (gene >= layout.gene_none) throw UnsupportedError("add_umi_to_cell: the gene index does not fit the key layout of this container");
	if (mark > 7u) throw InvalidError("add_umi_to_cell: mark out of range");
	u64 field;
	if (!map_umi_or_add(umi_code, field)) {
		if (umi_code & ESCAPE_BIT) throw UnsupportedError("add_umi_to_cell: a UMI with N that no read of the container carries");
		if (umi_dict_on) throw UnsupportedError("add_umi_to_cell: a UMI outside the container's UMI dictionary (no gene-bearing read carries it)");
		if (umi_sentinel_stripped && bit_length(umi_code) - 1 != umi_clean_bits) throw UnsupportedError("add_umi_to_cell: a UMI of another length than the container's");
		throw UnsupportedError("add_umi_to_cell: the UMI does not fit the key layout");
	}
	const u64 cg = (u64(cell) << layout.gene_bits) | gene;
	if (umi_overrides.count(cg)) throw UnsupportedError("add_umi_to_cell on a group that the UMI merge strategy rewrote");
	if (n_mol >= 0xFFFFFFF0u) throw UnsupportedError("molecule table would exceed 2^32 rows");
	const u64 key = (cg << layout.umi_bits) | field;
	if (with_qual) {   // an existing molecule: UMI::add_read's length check against the length it was created with
		scalars.ensure(16);
		hipLaunchKernelGGL(find_molecule_row_kernel, dim3(1), dim3(1), 0, stream, mol_key.p, n_mol, key, scalars.p);
		hipLaunchKernelGGL(molecule_quality_length_kernel, dim3(1), dim3(1), 0, stream, mol_qsum.p, mol_qrow.p, scalars.p, qual_stride(), scalars.p + 1);
		HIP_CHECK(hipGetLastError());
		u32 expected = 0;
		fetch(&expected, scalars.p + 1, 4);
		if (expected != 0xFFFFFFFFu && expected != quality_length)
			throw InvalidError("Wrong quality length: " + std::to_string(quality_length) + ", expected: " + std::to_string(expected));
	}
	const u32 row_vals[4] = {1u, mark, (mark >> 1) & 1u, (mark >> 2) & 1u};
	const u32 total = n_mol + 1;
	grow_preserving(mol_key, n_mol, size_t(total) + 1, stream);
	DevBuf<u32> *cols[4] = {&mol_reads, &mol_mark, &mol_exon, &mol_intron};
	for (int k = 0; k < 4; ++k) grow_preserving(*cols[k], n_mol, size_t(total) + 1, stream);
	HIP_CHECK(hipMemcpyAsync(mol_key.p + n_mol, &key, 8, hipMemcpyHostToDevice, stream));
	for (int k = 0; k < 4; ++k) HIP_CHECK(hipMemcpyAsync(cols[k]->p + n_mol, &row_vals[k], 4, hipMemcpyHostToDevice, stream));
	const size_t qstride = qual_stride();
	DevBuf<u32> d_q;
	std::vector<u32> q32(qstride, 0);
	if (with_qual) {
		// the read arrives with a sums row of its own (a new molecule keeps it; folded into an existing one, its bytes are added to
		// that molecule's row afterwards: the fold itself keeps the row of the member with the smaller index, the existing one)
		for (u32 i = 0; i < quality_length; ++i) q32[i] = quality[i];
		q32[qstride - 1] = quality_length;
		grow_preserving(mol_qsum, size_t(n_qsum_rows) * qstride, size_t(n_qsum_rows + 1) * qstride, stream);
		HIP_CHECK(hipMemcpyAsync(mol_qsum.p + size_t(n_qsum_rows) * qstride, q32.data(), qstride * 4, hipMemcpyHostToDevice, stream));
		grow_preserving(mol_qrow, n_mol, size_t(total), stream);
		HIP_CHECK(hipMemcpyAsync(mol_qrow.p + n_mol, &n_qsum_rows, 4, hipMemcpyHostToDevice, stream));
		d_q.alloc(qstride);
		HIP_CHECK(hipMemcpyAsync(d_q.p, q32.data(), qstride * 4, hipMemcpyHostToDevice, stream));
	}
	HIP_CHECK(stream_wait(stream));
	if (with_qual) ++n_qsum_rows;
	const u32 before = n_mol;
	if (mol_sorted_rows > n_mol) mol_sorted_rows = n_mol;   // the new row sits behind the sorted table
	n_mol = total;
	reaggregate_after_merge();          // identity remap apart from the recorded merges: sorts the row in, folds equal keys
	const bool is_new = n_mol == before + 1;
	if (with_qual && !is_new) {
		scalars.ensure(16);
		hipLaunchKernelGGL(find_molecule_row_kernel, dim3(1), dim3(1), 0, stream, mol_key.p, n_mol, key, scalars.p);
		hipLaunchKernelGGL(add_quality_row_kernel, dim3(1), dim3(256), 0, stream, mol_qsum.p, mol_qrow.p, scalars.p, u32(qstride), d_q.p, quality_length);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(stream_wait(stream));
	}
	real_pristine = false;
	if (is_new) { const long ri = real_find(cell); if (ri >= 0) real[size_t(ri)].row.total_umis += 1; }   // TOTAL_UMIS_PER_CB (:360-363)
	request_filtered(filtered_threshold, filtered_max_cells);
}
