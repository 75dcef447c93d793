// umi_merge_host.h -- UMIs containing 'N' (MergeUMIsStrategySimple, Estimation/Merge/UMIs/MergeUMIsStrategySimple.cpp:21-102).
// Included by dropest_amd.hip.
//
// Such UMIs are rare (<0.1 % of real data) and cannot live in a 2-bit code, so they travel as escaped codes and
// sort LAST inside their (cell, gene) group.  The device finds the affected groups of real cells and gathers
// their molecules; the decision per group follows the reference literally on the host, INCLUDING the
// implementation-defined orders it depends on: the bad UMIs are held in a std::unordered_set<std::string>
// filled in UMI-index order and random fills draw from glibc rand() (srand(42) when the strategy -- here the
// context -- is created), so the same libstdc++/glibc give the same fills.  Ties between clean candidates
// (equal distance and read count) go to the lowest UMI index = earliest first occurrence among gene-bearing
// reads; the device computes those first ordinals on demand.
#pragma once

namespace {

__global__ __launch_bounds__(256) void flag_n_groups_kernel(const unsigned long long *__restrict__ cg_key,
                                                            const uint32_t *__restrict__ cg_mol_begin,
                                                            const unsigned long long *__restrict__ mol_key, uint32_t n_cg,
                                                            const uint32_t *__restrict__ real_flag, int gene_bits,
                                                            unsigned long long gene_none, unsigned long long umi_mask,
                                                            unsigned long long escape_base, uint32_t *__restrict__ list,
                                                            uint32_t *__restrict__ count) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	bool hit = false;
	if (i < n_cg) {
		const unsigned long long k = cg_key[i];
		if ((k & gene_none) != gene_none && real_flag[uint32_t(k >> gene_bits)]) {
			const uint32_t last = cg_mol_begin[i + 1] - 1;          // escaped codes sort last in the group
			hit = (mol_key[last] & umi_mask) >= escape_base;
		}
	}
	const unsigned long long m = __ballot(hit);
	uint32_t base = 0;
	if (dropest::lane_id() == 0 && m) base = atomicAdd(count, uint32_t(__popcll(m)));
	base = __shfl(base, 0, 64);
	if (hit) list[base + __popcll(m & ((1ull << dropest::lane_id()) - 1ull))] = i;
}

__global__ __launch_bounds__(256) void group_extents_kernel(const uint32_t *__restrict__ idx, uint32_t n,
                                                            const uint32_t *__restrict__ cg_mol_begin,
                                                            uint32_t *__restrict__ begin, uint32_t *__restrict__ size) {
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	const uint32_t i = idx[j];
	begin[j] = cg_mol_begin[i];
	size[j] = cg_mol_begin[i + 1] - cg_mol_begin[i];
}

// one block per group: copies its molecules to the contiguous staging area
__global__ __launch_bounds__(64) void gather_groups_kernel(const uint32_t *__restrict__ begin, const uint32_t *__restrict__ size,
                                                           const uint32_t *__restrict__ offset,
                                                           const unsigned long long *__restrict__ mol_key,
                                                           const uint32_t *__restrict__ mol_reads, const uint32_t *__restrict__ mol_mark,
                                                           unsigned long long *__restrict__ o_key, uint32_t *__restrict__ o_reads,
                                                           uint32_t *__restrict__ o_mark) {
	const uint32_t g = blockIdx.x, b = begin[g], n = size[g], o = offset[g];
	for (uint32_t t = threadIdx.x; t < n; t += 64) { o_key[o + t] = mol_key[b + t]; o_reads[o + t] = mol_reads[b + t]; o_mark[o + t] = mol_mark[b + t]; }
}

// first ordinal among gene-bearing reads of each UMI code in the sorted query set
__global__ __launch_bounds__(256) void umi_first_seen_kernel(const unsigned long long *__restrict__ umi,
                                                             const uint32_t *__restrict__ gene, uint32_t n,
                                                             const unsigned long long *__restrict__ query, uint32_t nq,
                                                             uint32_t *__restrict__ first) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
		if (gene[r] == dropest::NO_GENE) continue;
		const unsigned long long u = umi[r];
		uint32_t lo = 0, hi = nq;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (query[mid] < u) lo = mid + 1; else hi = mid; }
		if (lo < nq && query[lo] == u && r < first[lo]) atomicMin(&first[lo], r);
	}
}

__global__ __launch_bounds__(256) void first_of_keys_kernel(const unsigned long long *__restrict__ keys, uint32_t n,
                                                           unsigned long long umi_mask, const uint32_t *__restrict__ table,
                                                           uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = table[keys[i] & umi_mask];
}

__global__ __launch_bounds__(256) void patch_cg_kernel(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ n_all,
                                                       const uint32_t *__restrict__ n_req, const uint32_t *__restrict__ reads_req,
                                                       uint32_t n, uint32_t *cg_n_all, uint32_t *cg_n_req, uint32_t *cg_reads_req) {
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	cg_n_all[idx[j]] = n_all[j]; cg_n_req[idx[j]] = n_req[j]; cg_reads_req[idx[j]] = reads_req[j];
}

// Tools::hamming_distance with N wildcards (Tools/UtilFunctions.cpp:67-82)
unsigned hamming_n(const std::string &a, const std::string &b) {
	if (a.size() != b.size()) throw dropest::InvalidError("Strings should have equal length");
	unsigned d = 0;
	for (size_t i = 0; i < a.size(); ++i)
		if (a[i] != b[i] && !(a[i] == 'N' || b[i] == 'N')) ++d;
	return d;
}

// MergeUMIsStrategyAbstract::fix_n_umi_with_random (MergeUMIsStrategyAbstract.cpp:11-23), glibc rand() restated (context.h)
std::string fix_n_with_random(const std::string &umi, dropest::GlibcRand &rng) {
	static const char nt[] = "ACGT";
	std::string t(umi);
	for (char &c : t) if (c == 'N') c = nt[rng.next() % 4];
	return t;
}

}  // namespace

// Molecules of a sorted list of (cell, gene) groups, copied to the host (+ optionally the first read ordinal of
// each molecule's UMI from a umi_first table).
void dropest_ctx::umi_gather_groups(const std::vector<u32> &groups, GatheredGroups &G, const u32 *d_first_table) {
	const u32 n_groups = u32(groups.size());
	G.size.assign(n_groups, 0); G.off.assign(n_groups, 0);
	if (!n_groups) return;
	DevBuf<u32> d_idx, d_begin, d_size, d_off;
	d_idx.alloc(n_groups); d_begin.alloc(n_groups); d_size.alloc(n_groups); d_off.alloc(n_groups);
	HIP_CHECK(hipMemcpyAsync(d_idx.p, groups.data(), size_t(n_groups) * 4, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(group_extents_kernel, dim3(div_up(n_groups, 256)), dim3(256), 0, stream, d_idx.p, n_groups, cg_mol_begin.p,
	                   d_begin.p, d_size.p);
	HIP_CHECK(hipGetLastError());
	G.begin.assign(n_groups, 0);
	HIP_CHECK(hipMemcpyAsync(G.size.data(), d_size.p, size_t(n_groups) * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(G.begin.data(), d_begin.p, size_t(n_groups) * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
	uint64_t total = 0;
	for (u32 g = 0; g < n_groups; ++g) { G.off[g] = u32(total); total += G.size[g]; }
	if (total > 0xFFFFFFF0ull) throw UnsupportedError("too many molecules in the groups handled on the host");
	DevBuf<u64> s_key; DevBuf<u32> s_reads, s_mark, s_first;
	s_key.alloc(total); s_reads.alloc(total); s_mark.alloc(total);
	HIP_CHECK(hipMemcpyAsync(d_off.p, G.off.data(), size_t(n_groups) * 4, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(gather_groups_kernel, dim3(n_groups), dim3(64), 0, stream, d_begin.p, d_size.p, d_off.p, mol_key.p,
	                   mol_reads.p, mol_mark.p, s_key.p, s_reads.p, s_mark.p);
	HIP_CHECK(hipGetLastError());
	G.hk.resize(total); G.hr.resize(total); G.hm.resize(total);
	HIP_CHECK(hipMemcpyAsync(G.hk.data(), s_key.p, total * 8, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(G.hr.data(), s_reads.p, total * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(G.hm.data(), s_mark.p, total * 4, hipMemcpyDeviceToHost, stream));
	if (d_first_table) {
		s_first.alloc(total);
		const u64 umask = layout.umi_bits ? ((1ull << layout.umi_bits) - 1ull) : 0ull;
		hipLaunchKernelGGL(first_of_keys_kernel, dim3(div_up(u32(total), 256)), dim3(256), 0, stream, s_key.p, u32(total), umask,
		                   d_first_table, s_first.p);
		HIP_CHECK(hipGetLastError());
		G.hfirst.resize(total);
		HIP_CHECK(hipMemcpyAsync(G.hfirst.data(), s_first.p, total * 4, hipMemcpyDeviceToHost, stream));
	}
	HIP_CHECK(stream_wait(stream));
}

// Writes the host-decided contents of re-keyed groups back: patched (cell, gene) rows, recomputed cell sizes, and the
// TOTAL_UMIS decrements of Cell::merge_umis (Cell.cpp:31-42).
void dropest_ctx::umi_patch_groups(const std::vector<u32> &p_idx, const std::vector<u32> &p_all, const std::vector<u32> &p_req,
                                   const std::vector<u32> &p_rreq, const std::unordered_map<u32, int> &umis_removed) {
	const u32 n_groups = u32(p_idx.size());
	if (n_groups) {
		DevBuf<u32> d_idx, d_pa, d_pr, d_prr;
		d_idx.alloc(n_groups); d_pa.alloc(n_groups); d_pr.alloc(n_groups); d_prr.alloc(n_groups);
		HIP_CHECK(hipMemcpyAsync(d_idx.p, p_idx.data(), size_t(n_groups) * 4, hipMemcpyHostToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(d_pa.p, p_all.data(), size_t(n_groups) * 4, hipMemcpyHostToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(d_pr.p, p_req.data(), size_t(n_groups) * 4, hipMemcpyHostToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(d_prr.p, p_rreq.data(), size_t(n_groups) * 4, hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(patch_cg_kernel, dim3(div_up(n_groups, 256)), dim3(256), 0, stream, d_idx.p, d_pa.p, d_pr.p, d_prr.p, n_groups,
		                   cg_n_all.p, cg_n_req.p, cg_reads_req.p);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(stream_wait(stream));   // the host vectors must outlive the copies
	}
	reduce_cell_gene_to_cells();
	HIP_CHECK(stream_wait(stream));
	refresh_real_rows();
	real_pristine = false;
	for (auto &kv : umis_removed) real[real_at(kv.first)].row.total_umis -= kv.second;
}

// position (in the resident reads) of the first gene-bearing read of each UMI code of a sorted list, 0xFFFFFFFF if none
std::vector<dropest::u32> dropest_ctx::umi_first_positions(const std::vector<u64> &sorted_codes) {
	const u32 nq = u32(sorted_codes.size());
	std::vector<u32> f(nq, 0xFFFFFFFFu);
	if (!nq || !n_reads) return f;
	DevBuf<u64> d_q; DevBuf<u32> d_first;
	d_q.alloc(nq); d_first.alloc(nq);
	HIP_CHECK(hipMemcpyAsync(d_q.p, sorted_codes.data(), size_t(nq) * 8, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemsetAsync(d_first.p, 0xFF, size_t(nq) * 4, stream));
	const u32 n = u32(n_reads);
	timed("umi_first_seen", double(n) * 12, [&] {
		need_columns();   // (a sharded run's reads may still be packed records)
		hipLaunchKernelGGL(umi_first_seen_kernel, dim3(std::min<u32>(div_up(n, 256), 8192u)), dim3(256), 0, stream, d_umi, d_gene, n,
		                   d_q.p, nq, d_first.p);
	});
	HIP_CHECK(hipMemcpyAsync(f.data(), d_first.p, size_t(nq) * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
	return f;
}

void dropest_ctx::run_umi_merge_simple() {
	umi_overrides.clear();
	if (ingest.umi_escape_max_plus1 == 0) return;   // no escaped UMI anywhere (on any shard): nothing can contain an N
	if (n_cg == 0 && !hooks) return;

	// 1. affected (cell, gene) groups of the cells that are real NOW (after the CB merge)
	std::vector<u32> flags(n_cells, 0);
	for (const HostCell &h : real)
		if (!h.merged && !h.excluded && h.row.n_genes >= min_before) flags[h.id] = 1;
	remap.ensure(n_cells);
	HIP_CHECK(hipMemcpyAsync(remap.p, flags.data(), size_t(n_cells) * 4, hipMemcpyHostToDevice, stream));
	DevBuf<u32> d_list; d_list.alloc(n_cg);
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 4, stream));
	const u64 umask = layout.umi_bits ? ((1ull << layout.umi_bits) - 1ull) : 0ull;
	if (n_cg) timed("flag_n_groups", double(n_cg) * 24, [&] {
		hipLaunchKernelGGL(flag_n_groups_kernel, dim3(div_up(n_cg, 256)), dim3(256), 0, stream, cg_key.p, cg_mol_begin.p, mol_key.p,
		                   n_cg, remap.p, layout.gene_bits, layout.gene_none, umask, layout.umi_escape_base, d_list.p, scalars.p);
	});
	u32 n_groups = 0;
	HIP_CHECK(hipMemcpyAsync(&n_groups, scalars.p, 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
	if (n_groups == 0 && !hooks) return;
	std::vector<u32> groups(n_groups);
	if (n_groups) HIP_CHECK(hipMemcpy(groups.data(), d_list.p, size_t(n_groups) * 4, hipMemcpyDeviceToHost));
	std::sort(groups.begin(), groups.end());   // (cell id, gene id) ascending == the reference's iteration order

	// 2. their molecules
	GatheredGroups GG;
	umi_gather_groups(groups, GG, nullptr);
	const std::vector<u32> &size = GG.size, &off = GG.off;
	const std::vector<u64> &hk = GG.hk;
	const std::vector<u32> &hr = GG.hr, &hm = GG.hm;

	struct Mol { u64 code; std::string seq; u32 reads, mark, row; bool bad; };
	std::vector<std::vector<Mol>> G(n_groups);
	for (u32 g = 0; g < n_groups; ++g) {
		for (u32 t = 0; t < size[g]; ++t) {
			Mol m;
			m.row = GG.begin[g] + t;
			m.code = unmap_umi(hk[off[g] + t] & umask);
			m.seq = decode_code(m.code, side);
			m.reads = hr[off[g] + t]; m.mark = hm[off[g] + t];
			m.bad = m.seq.find('N') != std::string::npos;   // MergeUMIsStrategySimple::is_umi_real (:61-64)
			G[g].push_back(std::move(m));
		}
	}

	// 3. clean UMIs that can tie as a target need their global first-seen order: query the device once
	std::vector<u64> tie_codes;
	auto candidates_of = [&](const std::vector<Mol> &mols, const Mol &bad, unsigned &min_ed, u32 &best_reads, std::vector<size_t> &best) {
		min_ed = ~0u; best_reads = 0; best.clear();
		for (size_t i = 0; i < mols.size(); ++i) {
			if (mols[i].bad) continue;
			const unsigned ed = hamming_n(mols[i].seq, bad.seq);
			if (ed < min_ed || (ed == min_ed && mols[i].reads > best_reads)) { min_ed = ed; best_reads = mols[i].reads; best.assign(1, i); }
			else if (ed == min_ed && mols[i].reads == best_reads) best.push_back(i);
		}
	};
	for (u32 g = 0; g < n_groups; ++g)
		for (const Mol &b : G[g]) {
			if (!b.bad) continue;
			unsigned min_ed; u32 br; std::vector<size_t> best;
			candidates_of(G[g], b, min_ed, br, best);
			if (best.size() > 1 && min_ed <= u32(cfg.max_umi_merge_edit_distance))
				for (size_t i : best) tie_codes.push_back(G[g][i].code);
		}
	// (one container: positions in the resident reads order like stream ordinals; sharded runs: the smallest GLOBAL ordinal
	// over all shards -- a UMI's first occurrence may sit on any of them)
	std::unordered_map<u64, u64> first_seen;
	if (!tie_codes.empty() || hooks) {
		std::sort(tie_codes.begin(), tie_codes.end());
		tie_codes.erase(std::unique(tie_codes.begin(), tie_codes.end()), tie_codes.end());
		if (hooks) {
			const std::vector<u64> f = hooks->first_seen_global(tie_codes);
			for (size_t i = 0; i < tie_codes.size(); ++i) first_seen[tie_codes[i]] = f[i];
		} else {
			const std::vector<u32> f = umi_first_positions(tie_codes);
			for (size_t i = 0; i < tie_codes.size(); ++i) first_seen[tie_codes[i]] = f[i];
		}
	}

	// 4. the reference's per-group decision and re-keying (MergeUMIsStrategySimple::merge / find_targets)
	std::vector<u32> p_idx, p_all, p_req, p_rreq;
	std::unordered_map<u32, int> umis_removed;   // per cell: TOTAL_UMIS decrements (Cell::merge_umis, Cell.cpp:31-42)
	// the bad UMIs of every group, in the container the reference iterates: they enter the unordered_set in UMI-index order
	// = first-seen order = ascending escape id
	std::vector<std::unordered_set<std::string>> bad_of(n_groups);
	std::vector<std::unordered_map<std::string, size_t>> mol_of_g(n_groups);
	for (u32 g = 0; g < n_groups; ++g) {
		std::vector<Mol> &mols = G[g];
		std::vector<size_t> bad_idx;
		for (size_t i = 0; i < mols.size(); ++i) if (mols[i].bad) bad_idx.push_back(i);
		std::sort(bad_idx.begin(), bad_idx.end(), [&](size_t x, size_t y) { return (mols[x].code & ~ESCAPE_BIT) < (mols[y].code & ~ESCAPE_BIT); });
		for (size_t i : bad_idx) { bad_of[g].insert(mols[i].seq); mol_of_g[g][mols[i].seq] = i; }
	}
	// Sharded runs: the reference draws every random fill from ONE rand() sequence, walking the cells in cell-id order and
	// the genes in gene-index order (MergeUMIsStrategySimple.cpp:25-53).  Which UMIs fall to a random fill, and with how
	// many draws, is decided by the group alone; the shards exchange (first ordinal of the cell, gene, draws) and every
	// group starts at its offset in the one sequence.
	std::vector<u64> rng_offset;
	if (hooks) {
		std::vector<u32> cell_first(n_groups), gene_of(n_groups), draws(n_groups, 0);
		for (u32 g = 0; g < n_groups; ++g) {
			const u64 cgk = hk[off[g]] >> layout.umi_bits;
			const u32 cell = u32(cgk >> layout.gene_bits);
			cell_first[g] = real[real_at(cell)].row.first_read; gene_of[g] = u32(cgk & layout.gene_none);
			for (const std::string &b : bad_of[g]) {
				unsigned min_ed; u32 br; std::vector<size_t> best;
				candidates_of(G[g], G[g][mol_of_g[g].at(b)], min_ed, br, best);
				if (best.empty() || min_ed > u32(cfg.max_umi_merge_edit_distance)) draws[g] += u32(std::count(b.begin(), b.end(), 'N'));
			}
		}
		rng_offset = hooks->rng_offsets(cell_first, gene_of, draws);
	}
	if (n_groups == 0) return;   // (a shard without such groups still took part in the two exchanges above)
	for (u32 g = 0; g < n_groups; ++g) {
		std::vector<Mol> &mols = G[g];
		const std::unordered_set<std::string> &bad = bad_of[g];
		const std::unordered_map<std::string, size_t> &mol_of = mol_of_g[g];
		if (hooks) rng.skip_to(rng_offset[g]);
		std::unordered_map<std::string, std::string> targets;
		for (const std::string &b : bad) {
			const Mol &bm = mols[mol_of.at(b)];
			unsigned min_ed; u32 br; std::vector<size_t> best;
			candidates_of(mols, bm, min_ed, br, best);
			if (best.empty() || min_ed > u32(cfg.max_umi_merge_edit_distance)) { targets[b] = fix_n_with_random(b, rng); continue; }
			size_t pick = best[0];
			if (best.size() > 1)
				for (size_t i : best) if (first_seen.at(mols[i].code) < first_seen.at(mols[pick].code)) pick = i;
			targets[b] = mols[pick].seq;
		}
		// Cell::merge_umis + Gene::merge(src, tgt) (Gene.cpp:38-58): counts add, marks OR
		struct Folded { u32 reads, mark, row; };              // row: the molecule whose quality sums this one shows
		std::map<std::string, Folded> merged;                 // clean sequence -> molecule
		std::unordered_map<std::string, u64> code_of;         // codes of the molecules that already exist
		for (const Mol &m : mols) if (!m.bad) { merged[m.seq] = Folded{m.reads, m.mark, m.row}; code_of[m.seq] = m.code; }
		const u32 cell = u32(hk[off[g]] >> (layout.umi_bits + layout.gene_bits));
		for (auto const &t : targets) {
			if (t.second == t.first) continue;
			const Mol &src = mols[mol_of.at(t.first)];
			auto it = merged.find(t.second);
			if (it == merged.end()) merged[t.second] = Folded{src.reads, src.mark, src.row};   // copy of the source (Gene.cpp:49)
			else { it->second.reads += src.reads; it->second.mark |= src.mark; }
			umis_removed[cell] += 1;
		}
		std::vector<UmiOverride> ov;
		u32 n_req = 0, reads_req = 0;
		for (auto const &kv : merged) {
			UmiOverride o;
			u64 code;
			auto known = code_of.find(kv.first);
			if (known != code_of.end()) code = known->second;
			else if (!encode_code(kv.first, code)) throw UnsupportedError("re-keyed UMI does not fit a 2-bit code: " + kv.first);
			o.umi = code; o.reads = kv.second.reads; o.mark = uint8_t(kv.second.mark); o.src_row = kv.second.row;
			ov.push_back(o);
			if ((query_mask >> (o.mark & 7u)) & 1u) { ++n_req; reads_req += o.reads; }
		}
		std::sort(ov.begin(), ov.end(), [](const UmiOverride &a, const UmiOverride &b) { return a.umi < b.umi; });
		p_idx.push_back(groups[g]); p_all.push_back(u32(ov.size())); p_req.push_back(n_req); p_rreq.push_back(reads_req);
		umi_overrides[hk[off[g]] >> layout.umi_bits] = std::move(ov);
	}

	// 5. patch the (cell, gene) rows, recompute the cell sizes, apply the TOTAL_UMIS decrements
	umi_patch_groups(p_idx, p_all, p_req, p_rreq, umis_removed);
}
