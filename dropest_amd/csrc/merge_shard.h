// merge_shard.h -- whitelist CB merge when the cells are sharded over several GPUs (included by dropest_amd.hip).
//
// The reference has one container (MergeStrategyBase::merge_inited, Estimation/Merge/MergeStrategyBase.cpp:11-57);
// here a barcode's merge target can live on another shard.  The merge is cut into phases with small collectives
// between them (driven by dropest_amd/multi.py):
//   search     every shard: neighbour search of ITS real cells against the real cells of ALL shards
//              (their barcodes / gene counts / TOTAL_UMIS are all-gathered first; ~10^5..10^6 rows)
//   export     molecule rows of the bases that need an intersection size (non-whitelist barcodes with a candidate:
//              a few % of all molecules) -> all-gathered
//   intersect  the shard owning the candidate joins the shipped base rows with its own molecule table
//   decide     the shard owning the base picks the target (same double arithmetic and tie replay as one GPU)
//   apply      every shard replays the sequential smallest-first application over the GLOBAL order
//              (apply_merge_order; pure host arithmetic on small arrays, identical on every rank)
//   finish     shipped rows of bases whose final target lives here are appended to the molecule table, local
//              sources are re-keyed, the table is re-sorted and re-reduced
// The key fields (gene | UMI) have the same layout on every shard because the ingest summary is all-reduced before
// the keys are planned (dropest_ingest_summary_set).
#pragma once

namespace {

__global__ __launch_bounds__(256) void universe_table_kernel(const unsigned long long *__restrict__ cb, uint32_t n,
                                                             dropest::CbTable t, uint32_t *__restrict__ bad) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	bool ok = true;
	const uint32_t s = dropest::cb_find_or_insert(t, cb[i], dropest::mix64(cb[i]) & t.mask, ok);
	if (!ok) { atomicMax(bad, 1u); return; }
	if (atomicCAS(&t.slots[s].nfirst, 0u, 1u) != 0u) { atomicMax(bad, 2u); return; }   // the same barcode twice
	t.slots[s].cell_id = i;
}

// molecule range of each listed cell
__global__ __launch_bounds__(256) void cell_ranges_kernel(const uint32_t *__restrict__ cells, uint32_t n,
                                                          const uint32_t *__restrict__ cell_cg_begin,
                                                          const uint32_t *__restrict__ cell_cg_count,
                                                          const uint32_t *__restrict__ cg_mol_begin, uint32_t *__restrict__ begin,
                                                          uint32_t *__restrict__ end) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const uint32_t c = cells[i];
	begin[i] = cg_mol_begin[cell_cg_begin[c]];
	end[i] = cg_mol_begin[cell_cg_begin[c] + cell_cg_count[c]];
}

struct ExportArgs {
	const uint32_t *begin, *out_off; uint32_t n_cells;
	const unsigned long long *mol_key; const uint32_t *col[4];   // reads, mark, exon, intron (exon / intron may be null)
	unsigned long long low_mask;
	unsigned long long *o_low; uint32_t *o_col[4];
};
// one block per listed cell: copies its molecule rows to the contiguous export buffers
__global__ __launch_bounds__(256) void export_rows_kernel(ExportArgs a) {
	const uint32_t c = blockIdx.x;
	const uint32_t b = a.begin[c], o = a.out_off[c], len = a.out_off[c + 1] - o;
	for (uint32_t i = threadIdx.x; i < len; i += 256) {
		a.o_low[o + i] = a.mol_key[b + i] & a.low_mask;
#pragma unroll
		for (int k = 0; k < 4; ++k) a.o_col[k][o + i] = a.col[k] ? a.col[k][b + i] : 0u;
	}
}

// the quality sums rows (quality.h: qstride words, the molecule's length last) of the exported molecule rows
__global__ __launch_bounds__(256) void export_quality_kernel(const uint32_t *__restrict__ begin, const uint32_t *__restrict__ out_off, const uint32_t *__restrict__ mol_qrow,
                                                             const uint32_t *__restrict__ mol_qsum, uint32_t qstride, uint32_t *__restrict__ o_q) {
	const uint32_t c = blockIdx.x;
	const uint32_t b = begin[c], o = out_off[c], len = out_off[c + 1] - o;
	for (uint32_t k = threadIdx.x; k < len * qstride; k += 256) {
		const uint32_t i = k / qstride, w = k % qstride;
		o_q[size_t(o + i) * qstride + w] = mol_qsum[size_t(mol_qrow[b + i]) * qstride + w];
	}
}
__global__ __launch_bounds__(256) void iota_from_kernel(uint32_t *out, uint32_t n, uint32_t first) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = first + i;
}

__global__ __launch_bounds__(256) void pair_ranges_ext_kernel(const uint32_t *__restrict__ cand_cell, const uint32_t *__restrict__ base_begin,
                                                              const uint32_t *__restrict__ base_end, uint32_t n,
                                                              const uint32_t *__restrict__ cell_cg_begin,
                                                              const uint32_t *__restrict__ cell_cg_count,
                                                              const uint32_t *__restrict__ cg_mol_begin, dropest::PairRange *out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const uint32_t c = cand_cell[i];
	dropest::PairRange r;
	r.base_begin = base_begin[i]; r.base_end = base_end[i];
	r.cand_begin = cg_mol_begin[cell_cg_begin[c]]; r.cand_end = cg_mol_begin[cell_cg_begin[c] + cell_cg_count[c]];
	out[i] = r;
}

__global__ __launch_bounds__(256) void import_keys_kernel(const uint32_t *__restrict__ cell, const unsigned long long *__restrict__ low,
                                                          uint32_t n, int cell_shift, unsigned long long *__restrict__ mol_key) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) mol_key[i] = ((unsigned long long)cell[i] << cell_shift) | low[i];
}

template <class T>
void grow_preserving(dropest::DevBuf<T> &b, size_t keep, size_t want, hipStream_t st) {
	if (want <= b.n) return;
	dropest::DevBuf<T> nb;
	nb.alloc(want + want / 8);
	if (keep) HIP_CHECK(hipMemcpyAsync(nb.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, st));
	HIP_CHECK(stream_wait(st));
	b = std::move(nb);
}

}  // namespace

struct dropest_ctx::ShardMerge {
	u32 n_global = 0;
	std::vector<u64> g_barcode;
	std::vector<int32_t> g_total_umis;
	std::vector<u32> g_n_genes;
	std::unordered_map<u64, u32> by_code;   // whitelists searched on the host (more than WL_MAX_PARTS parts): barcode -> place in the global list
	dropest::DevBuf<u64> d_cb;
	dropest::DevBuf<u32> d_n_genes, d_total_umis, d_iota;
	dropest::DevBuf<dropest::CbSlot> d_slots;
	dropest::MergeUniverse U;
	dropest::MergeSearch S;
	std::vector<u32> base_g, base_local;
	// export
	std::vector<u32> listed_f;          // positions (in base_g) of the bases whose rows are exported
	std::vector<u32> listed_g;          // their places in the global cell list
	std::vector<uint64_t> row_offset;   // [listed + 1]
	dropest::DevBuf<u64> x_low;
	dropest::DevBuf<u32> x_col[4];
	dropest::DevBuf<u32> x_q;           // UMI qualities: the sums rows of the exported molecule rows
	// UMI qualities, set by the driver before shard_merge_finish (shard_merge_quality_import): the place of every listed local cell in its
	// target's merge order, and of every imported row's source; the sums rows of the imported rows
	std::vector<u32> local_rank;
	const u32 *d_import_rank = nullptr, *d_import_q = nullptr;
	bool quality_import_set = false;
};

// molecule rows (and their quality sums rows) of the listed local cells -> the contiguous export buffers of the shard merge
void dropest_ctx::shard_export_rows(const std::vector<u32> &cells) {
	if (!shard) throw InvalidError("internal: no shard merge in progress");
	ShardMerge &M = *shard;
	const u32 nl = u32(cells.size());
	M.row_offset.assign(size_t(nl) + 1, 0);
	if (nl) {
		std::vector<u32> b(nl), e(nl);
		for (u32 i = 0; i < nl; ++i) if (cells[i] >= n_cells) throw RangeError("exported cell is not a cell of this shard");
		DevBuf<u32> d_cells, d_b, d_e, d_off;
		d_cells.alloc(nl); d_b.alloc(nl); d_e.alloc(nl); d_off.alloc(size_t(nl) + 1);
		HIP_CHECK(hipMemcpyAsync(d_cells.p, cells.data(), size_t(nl) * 4, hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(cell_ranges_kernel, dim3(div_up(nl, 256)), dim3(256), 0, stream, d_cells.p, nl, cell_cg_begin.p,
		                   cell_cg_count.p, cg_mol_begin.p, d_b.p, d_e.p);
		HIP_CHECK(hipGetLastError());
		fetch(b.data(), d_b.p, size_t(nl) * 4);
		fetch(e.data(), d_e.p, size_t(nl) * 4);
		std::vector<u32> off(size_t(nl) + 1, 0);
		for (u32 i = 0; i < nl; ++i) {
			M.row_offset[i + 1] = M.row_offset[i] + (e[i] - b[i]);
			if (M.row_offset[i + 1] > 0xFFFFFFF0ull) throw UnsupportedError("more than 2^32 exported molecule rows");
			off[i + 1] = u32(M.row_offset[i + 1]);
		}
		const u32 rows = off[nl];
		M.x_low.alloc(std::max<u32>(rows, 1));
		for (auto &c : M.x_col) c.alloc(std::max<u32>(rows, 1));
		HIP_CHECK(hipMemcpyAsync(d_off.p, off.data(), (size_t(nl) + 1) * 4, hipMemcpyHostToDevice, stream));
		ExportArgs a{};
		a.begin = d_b.p; a.out_off = d_off.p; a.n_cells = nl; a.mol_key = mol_key.p;
		a.col[0] = mol_reads.p; a.col[1] = mol_mark.p;
		a.col[2] = chr_from_gene ? mol_exon.p : nullptr; a.col[3] = chr_from_gene ? mol_intron.p : nullptr;
		a.low_mask = (1ull << (layout.gene_bits + layout.umi_bits)) - 1ull;
		a.o_low = M.x_low.p;
		for (int k = 0; k < 4; ++k) a.o_col[k] = M.x_col[k].p;
		timed("shard_merge:export", double(rows) * 48, [&] {
			hipLaunchKernelGGL(export_rows_kernel, dim3(nl), dim3(256), 0, stream, a);
		});
		if (have_qual && qual_len) {
			M.x_q.alloc(std::max<size_t>(size_t(rows) * qual_stride(), 1));
			hipLaunchKernelGGL(export_quality_kernel, dim3(nl), dim3(256), 0, stream, d_b.p, d_off.p, mol_qrow.p, mol_qsum.p, qual_stride(), M.x_q.p);
			HIP_CHECK(hipGetLastError());
		}
		HIP_CHECK(stream_wait(stream));
	}
}

void dropest_ctx::shard_merge_search(uint64_t n_global, const uint64_t *g_barcode, const uint32_t *g_n_genes, const int32_t *g_total_umis,
                                     uint64_t n_bases, const uint32_t *base_g, const uint32_t *base_local, uint64_t *n_pairs) {
	if (!initialized) throw InvalidError("You must initialize container");
	if (merged) throw InvalidError("merge_and_filter was already run");
	if (cfg.merge_kind != DROPEST_MERGE_REAL_BARCODES && cfg.merge_kind != DROPEST_MERGE_POISSON_REAL)
		throw InvalidError("shard merge needs a barcode whitelist (merge_kind REAL_BARCODES or POISSON_REAL)");
	if (n_global >= 0x7FFFFFFFull) throw UnsupportedError("more than 2^31 real cells");
	HostStage hs(this, "shard_merge:search");
	shard.reset(new ShardMerge());
	ShardMerge &M = *shard;
	const u32 nG = u32(n_global);
	M.n_global = nG;
	M.g_barcode.assign(g_barcode, g_barcode + nG);
	M.g_total_umis.assign(g_total_umis, g_total_umis + nG);
	M.g_n_genes.assign(g_n_genes, g_n_genes + nG);
	M.base_g.assign(base_g, base_g + n_bases);
	M.base_local.assign(base_local, base_local + n_bases);
	for (u32 f = 0; f < n_bases; ++f) {
		if (base_g[f] >= nG) throw RangeError("base outside the global cell list");
		if (base_local[f] >= n_cells) throw RangeError("base is not a cell of this shard");
	}
	*n_pairs = 0;
	if (nG == 0 || n_bases == 0) { M.S.F = 0; M.S.pair_first.assign(1, 0); return; }

	M.d_cb.alloc(nG); M.d_n_genes.alloc(nG); M.d_total_umis.alloc(nG); M.d_iota.alloc(nG);
	HIP_CHECK(hipMemcpyAsync(M.d_cb.p, g_barcode, size_t(nG) * 8, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(M.d_n_genes.p, g_n_genes, size_t(nG) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(M.d_total_umis.p, g_total_umis, size_t(nG) * 4, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(iota_kernel, dim3(div_up(nG, 256)), dim3(256), 0, stream, M.d_iota.p, nG);
	uint64_t cap = 1024;
	while (cap < uint64_t(nG) * 2) cap <<= 1;
	M.d_slots.alloc(cap);
	HIP_CHECK(hipMemsetAsync(M.d_slots.p, 0, cap * sizeof(CbSlot), stream));
	CbTable t{M.d_slots.p, cap - 1};
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 16, stream));
	hipLaunchKernelGGL(universe_table_kernel, dim3(div_up(nG, 256)), dim3(256), 0, stream, M.d_cb.p, nG, t, scalars.p);
	HIP_CHECK(hipGetLastError());
	u32 bad = 0;
	fetch(&bad, scalars.p, 4);
	if (bad == 2) throw InvalidError("the global cell list holds one barcode twice (cells must be sharded by barcode)");
	if (bad) throw DeviceError("global barcode table overflow");

	MergeUniverse &U = M.U;
	U.table = t; U.cell_cb = M.d_cb.p; U.n_genes = M.d_n_genes.p;
	U.total_umis = M.d_total_umis.p; U.real_index = M.d_iota.p;
	// barcodes with N (escaped codes; every shard holds the whole side-string table): such a BASE is split into the whitelist's
	// parts on the host like on one GPU (Tools::edit_distance treats N as a wildcard); candidates are whitelist barcodes, never escaped
	U.any_escaped = false;
	for (u32 f = 0; f < n_bases; ++f) if (g_barcode[base_g[f]] & ESCAPE_BIT) U.any_escaped = true;
	ShardMerge *pm = &M;
	U.base_total_umis = [pm](u32 f) { return pm->g_total_umis[pm->base_g[f]]; };
	U.barcode_code = [pm](u32 g) { return pm->g_barcode[g]; };
	const std::vector<std::string> *pside = &side;
	U.base_barcode_text = [pm, pside](u32 f) { return dropest::decode_code(pm->g_barcode[pm->base_g[f]], *pside); };
	if (!wl.loaded) {
		if (barcodes_file.empty()) throw InvalidError("merge_kind = REAL_BARCODES needs barcodes_file");
		wl.load(cfg.barcodes_kind, barcodes_file);
	}
	if (wl.parts.size() > size_t(WL_MAX_PARTS)) {   // the host search (merge_host.h) asks for cells by barcode: the global list, on the host
		M.by_code.reserve(size_t(nG) * 2);
		for (u32 g = 0; g < nG; ++g) M.by_code.emplace(M.g_barcode[g], g);
		U.find_cell = [pm](u64 code, u32 &ng, int32_t &tu, u32 &ri) -> long {
			auto it = pm->by_code.find(code);
			if (it == pm->by_code.end()) return -1;
			ng = pm->g_n_genes[it->second]; tu = pm->g_total_umis[it->second]; ri = it->second;
			return long(it->second);
		};
	}
	search_merge_candidates(M.base_g, U, M.S);
	*n_pairs = M.S.pair_base.size();

	// bases whose rows travel: those with at least one pair
	M.listed_f.clear();
	for (u32 f = 0; f < M.S.F; ++f) if (M.S.pair_first[f] != M.S.pair_first[f + 1]) M.listed_f.push_back(f);
	std::vector<u32> cells(M.listed_f.size());
	M.listed_g.resize(M.listed_f.size());
	for (size_t i = 0; i < M.listed_f.size(); ++i) { cells[i] = M.base_local[M.listed_f[i]]; M.listed_g[i] = M.base_g[M.listed_f[i]]; }
	shard_export_rows(cells);
	collect_timings();
}

void dropest_ctx::shard_merge_intersect(uint64_t n_pairs, const uint32_t *cand_local, const uint64_t *base_begin, const uint64_t *base_end,
                                        const uint64_t *d_base_low, uint32_t *inter) {
	if (!initialized) throw InvalidError("You must initialize container");
	if (n_pairs == 0) return;
	if (n_pairs > 0xFFFFFFF0ull) throw UnsupportedError("too many pairs");
	HostStage hs(this, "shard_merge:intersect");
	const u32 NP = u32(n_pairs);
	std::vector<u32> bb(NP), be(NP);
	for (u32 p = 0; p < NP; ++p) {
		if (cand_local[p] >= n_cells) throw RangeError("candidate is not a cell of this shard");
		if (base_end[p] > 0xFFFFFFF0ull || base_begin[p] > base_end[p]) throw RangeError("bad base row range");
		bb[p] = u32(base_begin[p]); be[p] = u32(base_end[p]);
	}
	DevBuf<u32> d_c, d_bb, d_be, d_inter; DevBuf<PairRange> d_pr;
	d_c.alloc(NP); d_bb.alloc(NP); d_be.alloc(NP); d_inter.alloc(NP); d_pr.alloc(NP);
	HIP_CHECK(hipMemcpyAsync(d_c.p, cand_local, size_t(NP) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_bb.p, bb.data(), size_t(NP) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_be.p, be.data(), size_t(NP) * 4, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(pair_ranges_ext_kernel, dim3(div_up(NP, 256)), dim3(256), 0, stream, d_c.p, d_bb.p, d_be.p, NP, cell_cg_begin.p,
	                   cell_cg_count.p, cg_mol_begin.p, d_pr.p);
	HIP_CHECK(hipGetLastError());
	const int low_bits = layout.gene_bits + layout.umi_bits;
	timed("umig_intersect", double(NP) * 64, [&] {
		hipLaunchKernelGGL(umig_intersect_kernel, dim3(NP), dim3(256), 0, stream, d_pr.p, NP,
		                   reinterpret_cast<const unsigned long long *>(d_base_low), mol_key.p, (1ull << low_bits) - 1ull,
		                   layout.umi_bits, layout.gene_none, d_inter.p);
	});
	fetch(inter, d_inter.p, size_t(NP) * 4);
	collect_timings();
}

void dropest_ctx::shard_merge_decide(const uint32_t *inter, int64_t *target_g) {
	if (!shard) throw InvalidError("dropest_shard_merge_search was not run");
	HostStage hs(this, "shard_merge:decide");
	ShardMerge &M = *shard;
	if (M.S.F == 0) return;
	std::vector<u32> in(inter, inter + M.S.pair_base.size());
	std::vector<long> targets;
	std::vector<u32> tr;
	decide_merge_targets(M.U, M.S, in, targets, tr);
	for (u32 f = 0; f < M.S.F; ++f) target_g[f] = targets[f];
}

// UMI qualities across shards (quality.h): a molecule the target has keeps the target's sums, one that only merged cells had takes the
// sums of the FIRST of them in merge order (Gene::merge, Gene.cpp:26-36) -- wherever those cells lived.  local_rank[i] belongs to
// local_id[i] of the shard_merge_finish call that follows; import_rank / import_q (device) to its imported rows.
void dropest_ctx::shard_merge_quality_import(uint64_t n_local, const uint32_t *local_rank, const uint32_t *d_import_rank, const uint32_t *d_import_q) {
	if (!shard) throw InvalidError("dropest_shard_merge_search was not run");
	shard->local_rank.assign(local_rank, local_rank + n_local);
	shard->d_import_rank = d_import_rank; shard->d_import_q = d_import_q;
	shard->quality_import_set = true;
}

void dropest_ctx::shard_merge_finish(uint64_t n_local, const uint32_t *local_id, const uint8_t *excluded, const uint8_t *merged_away,
                                     const int32_t *total_reads, const int32_t *total_umis, uint64_t n_moves,
                                     const uint32_t *move_src, const uint32_t *move_tgt, uint64_t n_import,
                                     const uint32_t *d_cell, const uint64_t *d_low, const uint32_t *const d_cols[4]) {
	if (!initialized) throw InvalidError("You must initialize container");
	if (merged) throw InvalidError("merge_and_filter was already run");
	HostStage hs(this, "shard_merge:finish");
	real_pristine = false;
	for (uint64_t i = 0; i < n_local; ++i) {
		HostCell &h = real[real_at(local_id[i])];
		h.excluded = excluded[i] != 0; h.merged = merged_away[i] != 0;
		h.row.total_reads = total_reads[i]; h.row.total_umis = total_umis[i];
	}
	const bool with_qual = have_qual && qual_len;
	if (with_qual) {
		if (!shard || !shard->quality_import_set || shard->local_rank.size() != n_local)
			throw UnsupportedError("a sharded barcode merge on a container with UMI qualities needs shard_merge_quality_import first");
		merge_rank.assign(n_cells, 0);
		for (uint64_t i = 0; i < n_local; ++i) merge_rank[local_id[i]] = shard->local_rank[i];
	}
	clear_strategy_pairs();
	for (uint64_t i = 0; i < n_moves; ++i) {
		if (move_src[i] >= n_cells || move_tgt[i] >= n_cells) throw RangeError("merge move outside this shard");
		merge_pairs.emplace_back(move_src[i], move_tgt[i]);
	}
	std::sort(merge_pairs.begin(), merge_pairs.end());
	if (uint64_t(n_mol) + n_import > 0xFFFFFFF0ull) throw UnsupportedError("molecule table would exceed 2^32 rows");
	if (n_import) {
		const u32 ni = u32(n_import), total = n_mol + ni;
		grow_preserving(mol_key, n_mol, size_t(total) + 1, stream);
		grow_preserving(mol_reads, n_mol, size_t(total) + 1, stream);
		grow_preserving(mol_mark, n_mol, size_t(total) + 1, stream);
		hipLaunchKernelGGL(import_keys_kernel, dim3(div_up(ni, 256)), dim3(256), 0, stream, d_cell,
		                   reinterpret_cast<const unsigned long long *>(d_low), ni, layout.gene_bits + layout.umi_bits, mol_key.p + n_mol);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipMemcpyAsync(mol_reads.p + n_mol, d_cols[0], size_t(ni) * 4, hipMemcpyDeviceToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(mol_mark.p + n_mol, d_cols[1], size_t(ni) * 4, hipMemcpyDeviceToDevice, stream));
		if (chr_from_gene) {
			grow_preserving(mol_exon, n_mol, size_t(total) + 1, stream);
			grow_preserving(mol_intron, n_mol, size_t(total) + 1, stream);
			HIP_CHECK(hipMemcpyAsync(mol_exon.p + n_mol, d_cols[2], size_t(ni) * 4, hipMemcpyDeviceToDevice, stream));
			HIP_CHECK(hipMemcpyAsync(mol_intron.p + n_mol, d_cols[3], size_t(ni) * 4, hipMemcpyDeviceToDevice, stream));
		}
		if (with_qual) {   // the imported rows bring their sums rows along; their place in the merge order decides who keeps what in the fold
			const size_t qs = qual_stride();
			grow_preserving(mol_qsum, size_t(n_qsum_rows) * qs, size_t(n_qsum_rows + ni) * qs, stream);
			HIP_CHECK(hipMemcpyAsync(mol_qsum.p + size_t(n_qsum_rows) * qs, shard->d_import_q, size_t(ni) * qs * 4, hipMemcpyDeviceToDevice, stream));
			grow_preserving(mol_qrow, n_mol, size_t(total), stream);
			hipLaunchKernelGGL(iota_from_kernel, dim3(div_up(ni, 256)), dim3(256), 0, stream, mol_qrow.p + n_mol, ni, n_qsum_rows);
			HIP_CHECK(hipGetLastError());
			n_qsum_rows += ni;
			reagg_import_prio = shard->d_import_rank; reagg_import_from = n_mol; reagg_import_n = ni;
		}
		HIP_CHECK(stream_wait(stream));
		mol_sorted_rows = n_mol;   // the imported rows sit behind the sorted table
		n_mol = total;
	}
	if (n_import || !merge_pairs.empty()) reaggregate_after_merge();
	external_merge_done = true;
	shard.reset();
	collect_timings();
}
