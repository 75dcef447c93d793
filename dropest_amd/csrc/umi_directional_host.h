// umi_directional_host.h -- host orchestration of the "directional" UMI correction (-u), included by dropest_amd.hip.
// Device part: k_umi_directional.h.  What stays here is the part of
// MergeUMIsStrategyDirectional (Estimation/Merge/UMIs/MergeUMIsStrategyDirectional.cpp:18-116) whose result depends
// on library behaviour: groups holding an N-UMI (glibc rand() fills drawn in cell order; the std::unordered_map
// iteration order of Cell::merge_umis matters once a fill collides with another source) and UMIs of several lengths
// (the reference's banded edit distance is not a plain Levenshtein there).  Those groups are replayed literally with the same libstdc++
// containers and calls.
#pragma once

namespace {

struct DirUmi { std::string seq; size_t n_reads; };

// MergeUMIsStrategyDirectional::find_target (:83-116)
std::string directional_find_target(size_t src, const std::vector<DirUmi> &v, double mult, unsigned max_ed, dropest::GlibcRand &rng) {
	const DirUmi &s = v[src];
	const bool has_n = s.seq.find('N') != std::string::npos;
	std::string target;
	unsigned min_ed = std::numeric_limits<unsigned>::max();
	for (long d = long(v.size()) - 1; d > long(src); --d) {
		const DirUmi &t = v[size_t(d)];
		if (s.n_reads * mult > t.n_reads) break;
		const unsigned ed = banded_edit_distance(s.seq, t.seq, max_ed);
		if (ed > max_ed) continue;
		if (ed < min_ed) {
			target = t.seq;
			if ((!has_n && ed <= 1) || ed == 0) break;
			min_ed = ed;
		}
	}
	if (has_n && target.empty()) return fix_n_with_random(s.seq, rng);
	return target;
}

// MergeUMIsStrategyDirectional::find_targets (:55-81); `v` arrives in UMI-index order
std::unordered_map<std::string, std::string> directional_find_targets(std::vector<DirUmi> &v, double mult, unsigned max_ed, dropest::GlibcRand &rng) {
	std::sort(v.begin(), v.end(), [](const DirUmi &a, const DirUmi &b) { return a.n_reads < b.n_reads; });
	std::unordered_map<std::string, std::string> out;
	for (size_t i = 0; i < v.size(); ++i) {
		std::string t = directional_find_target(i, v, mult, max_ed, rng);
		if (!t.empty()) out[v[i].seq] = t;
	}
	for (long i = long(v.size()) - 1; i >= 0; --i) {
		auto it = out.find(v[size_t(i)].seq);
		if (it == out.end()) continue;
		auto it2 = out.find(it->second);
		if (it2 == out.end()) continue;
		out[v[size_t(i)].seq] = it2->second;
	}
	return out;
}

}  // namespace

void dropest_ctx::run_umi_merge_directional() {
	umi_overrides.clear();
	if (layout.umi_bits > 28) throw UnsupportedError("-u needs a UMI field of at most 28 bits (table of first occurrences)");
	if (n_cg == 0 || n_mol == 0) {
		if (hooks) {   // a shard without molecules still takes part in the two exchanges of the others
			const size_t table = size_t(1) << layout.umi_bits;
			umi_first.ensure(table);
			HIP_CHECK(hipMemsetAsync(umi_first.p, 0xFF, table * 4, stream));
			hooks->globalize_umi_first(umi_first.p, table);
			(void)hooks->rng_offsets({}, {}, {});
		}
		return;
	}
	HostStage hs(this, "umi_directional");
	const u64 umask = layout.umi_bits ? ((1ull << layout.umi_bits) - 1ull) : 0ull;

	// 1. cells that are real NOW (after the CB merge); first read ordinal of every UMI = UMI-index order
	std::vector<u32> flags(n_cells, 0);
	for (const HostCell &h : real)
		if (!h.merged && !h.excluded && h.row.n_genes >= min_before) flags[h.id] = 1;
	remap.ensure(n_cells);
	HIP_CHECK(hipMemcpyAsync(remap.p, flags.data(), size_t(n_cells) * 4, hipMemcpyHostToDevice, stream));
	const size_t table = size_t(1) << layout.umi_bits;
	umi_first.ensure(table);
	HIP_CHECK(hipMemsetAsync(umi_first.p, 0xFF, table * 4, stream));
	const u32 n = u32(n_reads);
	timed("umi_first_table", double(n) * 12, [&] {
		need_columns();   // (a sharded run's reads may still be packed records)
		hipLaunchKernelGGL(umi_first_table_kernel, dim3(std::min<u32>(div_up(n, 256), 8192u)), dim3(256), 0, stream, umi_key_column(), d_gene, n,
		                   layout, umi_first.p);
	});
	if (hooks) hooks->globalize_umi_first(umi_first.p, table);   // sharded run: UMI index order of the whole stream

	// 2. device decision for the groups it can decide; re-keyed keys land in keys_a
	keys_a.ensure(n_mol); keys_b.ensure(n_mol); vals_a.ensure(n_mol); vals_b.ensure(n_mol);
	HIP_CHECK(hipMemcpyAsync(keys_a.p, mol_key.p, size_t(n_mol) * 8, hipMemcpyDeviceToDevice, stream));
	DevBuf<u32> d_removed, d_list, d_big, d_huge;
	d_removed.alloc(n_cells); d_list.alloc(n_cg); d_big.alloc(n_cg); d_huge.alloc(n_cg);
	HIP_CHECK(hipMemsetAsync(d_removed.p, 0, size_t(n_cells) * 4, stream));
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 16, stream));
	DirArgs a{};
	a.cg_key = cg_key.p; a.cg_mol_begin = cg_mol_begin.p; a.n_cg = n_cg; a.mol_key = mol_key.p; a.mol_reads = mol_reads.p;
	a.real_flag = remap.p; a.gene_bits = layout.gene_bits; a.umi_bits = layout.umi_bits;
	a.umi_len = umi_sentinel_stripped && !umi_dict_on ? umi_clean_bits / 2 : 0;   // (ranks of a UMI dictionary say nothing about bases: the host decides those groups)
	a.gene_none = layout.gene_none; a.escape_base = ingest.umi_escape_max_plus1 ? layout.umi_escape_base : ~0ull;
	a.umi_first = umi_first.p; a.mult = cfg.umi_merge_multiplier; a.max_ed = u32(cfg.max_umi_merge_edit_distance);
	a.new_key = keys_a.p; a.cell_removed = d_removed.p; a.host_list = d_list.p; a.host_count = scalars.p; a.n_changed = scalars.p + 1;
	a.big_list = d_big.p; a.big_count = scalars.p + 2; a.huge_list = d_huge.p; a.huge_count = scalars.p + 3;
	timed("umi_directional", double(n_mol) * 24, [&] {
		hipLaunchKernelGGL(directional_kernel, dim3(div_up(n_cg, 256)), dim3(256), 0, stream, a);
	});
	u32 counts[4] = {0, 0, 0, 0};
	fetch(counts, scalars.p, 16);
	const u32 n_big = counts[2], n_huge = counts[3];
	DirBigArgs bg{};
	bg.cg_mol_begin = cg_mol_begin.p; bg.cg_key = cg_key.p; bg.mol_key = mol_key.p; bg.mol_reads = mol_reads.p;
	bg.gene_bits = layout.gene_bits; bg.umi_bits = layout.umi_bits; bg.umi_len = a.umi_len; bg.umi_first = umi_first.p;
	bg.mult = a.mult; bg.max_ed = a.max_ed; bg.new_key = keys_a.p; bg.cell_removed = d_removed.p; bg.n_changed = scalars.p + 1;
	if (n_big) {   // groups of 17 .. 4096 UMIs: one wave each, work arrays in LDS
		bg.groups = d_big.p; bg.n_groups = n_big;
		timed("umi_directional:big", double(n_big) * 64 * 14, [&] {
			hipLaunchKernelGGL(directional_big_kernel, dim3(n_big), dim3(64), 0, stream, bg);
		});
	}
	DevBuf<u32> h_off, h_scratch[4]; DevBuf<int32_t> h_tgt;
	if (n_huge) {  // beyond that: 256 threads each, work arrays in global scratch
		std::vector<u32> hg(n_huge), ext_b(n_huge), ext_s(n_huge), off(n_huge);
		fetch(hg.data(), d_huge.p, size_t(n_huge) * 4);
		DevBuf<u32> d_b, d_s;
		d_b.alloc(n_huge); d_s.alloc(n_huge); h_off.alloc(n_huge);
		hipLaunchKernelGGL(group_extents_kernel, dim3(div_up(n_huge, 256)), dim3(256), 0, stream, d_huge.p, n_huge, cg_mol_begin.p, d_b.p, d_s.p);
		HIP_CHECK(hipGetLastError());
		fetch(ext_s.data(), d_s.p, size_t(n_huge) * 4);
		uint64_t total = 0;
		for (u32 i = 0; i < n_huge; ++i) { off[i] = u32(total); total += ext_s[i]; }
		if (total > 0x7FFFFFF0ull) throw UnsupportedError("too many UMIs in very large (cell, gene) groups");
		for (auto &b : h_scratch) b.alloc(total);
		h_tgt.alloc(total);
		HIP_CHECK(hipMemcpyAsync(h_off.p, off.data(), size_t(n_huge) * 4, hipMemcpyHostToDevice, stream));
		bg.groups = d_huge.p; bg.n_groups = n_huge; bg.scratch_off = h_off.p;
		bg.s_code = h_scratch[0].p; bg.s_reads = h_scratch[1].p; bg.s_first = h_scratch[2].p; bg.s_ord = h_scratch[3].p; bg.s_tgt = h_tgt.p;
		timed("umi_directional:huge", double(total) * 20, [&] {
			hipLaunchKernelGGL(directional_huge_kernel, dim3(n_huge), dim3(256), 0, stream, bg);
		});
		HIP_CHECK(stream_wait(stream));   // off (host vector) and the scratch buffers outlive the launch
	}
	if (n_big || n_huge) fetch(counts, scalars.p, 16);
	const u32 n_host = counts[0], n_changed = counts[1];
	if (profiling) {   // group counts by path, reported next to the kernel timings
		stats["count:umi_groups_host"].launches += n_host; stats["count:umi_groups_wave"].launches += n_big;
		stats["count:umi_groups_huge"].launches += n_huge;
		stats["count:umi_rekeyed"].launches += n_changed;
	}
	std::vector<u32> groups(n_host);
	if (n_host) fetch(groups.data(), d_list.p, size_t(n_host) * 4);
	std::sort(groups.begin(), groups.end());   // (cell id, gene id) ascending == the reference's iteration order

	std::unordered_map<u32, int> umis_removed;
	if (n_changed) {
		// TOTAL_UMIS decrements of the real cells (Cell::merge_umis, Cell.cpp:31-42)
		const u32 nr = u32(real.size());
		std::vector<u32> ids(nr), rem(nr);
		for (u32 i = 0; i < nr; ++i) ids[i] = real[i].id;
		real_list.ensure(nr);
		DevBuf<u32> d_rem; d_rem.alloc(nr);
		HIP_CHECK(hipMemcpyAsync(real_list.p, ids.data(), size_t(nr) * 4, hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(gather_u32_kernel, dim3(div_up(nr, 256)), dim3(256), 0, stream, d_removed.p, real_list.p, nr, d_rem.p);
		HIP_CHECK(hipGetLastError());
		fetch(rem.data(), d_rem.p, size_t(nr) * 4);
		// fold the re-keyed molecules: sort by the new keys, add read counts, OR marks
		u64 init[2] = {0ull, ~0ull};
		u64 *d_or_and = reinterpret_cast<u64 *>(scalars.p + 4);
		HIP_CHECK(hipMemcpyAsync(d_or_and, init, 16, hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(iota_or_and_kernel, dim3(std::min<u32>(div_up(n_mol, 256), 4096u)), dim3(256), 0, stream, keys_a.p, n_mol,
		                   vals_a.p, d_or_and);
		if (have_qual && qual_len) {   // a molecule that receives others keeps its own quality sums (Gene.cpp:50-54)
			reagg_prio_buf.ensure(n_mol);
			hipLaunchKernelGGL(prio_from_rekey_kernel, dim3(div_up(n_mol, 256)), dim3(256), 0, stream, mol_key.p, keys_a.p, n_mol, reagg_prio_buf.p);
			reagg_prio = reagg_prio_buf.p;
		}
		HIP_CHECK(hipGetLastError());
		u64 or_and[2];
		fetch(or_and, d_or_and, 16);
		reaggregate_from_keys(or_and[0] ^ or_and[1]);   // the (cell, gene) rows keep their indices: groups never vanish
		real_pristine = false;
		for (u32 i = 0; i < nr; ++i) if (rem[i]) real[i].row.total_umis -= int(rem[i]);
	}
	if (!n_host && !hooks) return;   // (a shard without such groups still takes part in the exchange of the offsets)

	// 3. the remaining groups, replayed literally on the host
	GatheredGroups GG;
	umi_gather_groups(groups, GG, umi_first.p);
	std::vector<u32> p_idx, p_all, p_req, p_rreq;
	struct Mol { u64 code; std::string seq; u32 reads, mark, first, row; };
	auto group_molecules = [&](u32 g, std::vector<Mol> &mols, std::vector<DirUmi> &v) {
		mols.assign(GG.size[g], Mol{});
		for (u32 t = 0; t < GG.size[g]; ++t) {
			Mol &m = mols[t];
			m.row = GG.begin[g] + t;
			m.code = unmap_umi(GG.hk[GG.off[g] + t] & umask);
			m.seq = decode_code(m.code, side);
			m.reads = GG.hr[GG.off[g] + t]; m.mark = GG.hm[GG.off[g] + t]; m.first = GG.hfirst[GG.off[g] + t];
		}
		std::vector<size_t> by_index(mols.size());
		for (size_t i = 0; i < by_index.size(); ++i) by_index[i] = i;
		std::sort(by_index.begin(), by_index.end(), [&](size_t x, size_t y) { return mols[x].first < mols[y].first; });
		v.clear();
		for (size_t i : by_index) v.push_back(DirUmi{mols[i].seq, size_t(mols[i].reads)});
	};
	// Sharded run: the random fills of ALL shards come from one rand() sequence, drawn group by group in (cell id, gene)
	// order.  How many numbers a group draws does not depend on their values (a fill happens when an N-UMI finds no
	// target): count them in a dry run, let the shards agree on every group's offset, then replay from there.
	std::vector<u64> rng_offset;
	if (hooks) {
		std::vector<u32> draws(n_host, 0), cell_first_pos(n_host), gene_of(n_host), cells_of(n_host);
		std::vector<Mol> mols; std::vector<DirUmi> v;
		for (u32 g = 0; g < n_host; ++g) {
			group_molecules(g, mols, v);
			dropest::GlibcRand dry(1);
			(void)directional_find_targets(v, cfg.umi_merge_multiplier, unsigned(cfg.max_umi_merge_edit_distance), dry);
			draws[g] = u32(dry.drawn);
			const u64 cgk = GG.hk[GG.off[g]] >> layout.umi_bits;
			cells_of[g] = u32(cgk >> layout.gene_bits); gene_of[g] = u32(cgk & layout.gene_none);
		}
		for (u32 g = 0; g < n_host; ++g) cell_first_pos[g] = real[real_at(cells_of[g])].row.first_read;
		rng_offset = hooks->rng_offsets(cell_first_pos, gene_of, draws);
	}
	for (u32 g = 0; g < n_host; ++g) {
		std::vector<Mol> mols;
		std::vector<DirUmi> v;
		group_molecules(g, mols, v);
		if (hooks) rng.skip_to(rng_offset[g]);
		const auto targets = directional_find_targets(v, cfg.umi_merge_multiplier, unsigned(cfg.max_umi_merge_edit_distance), rng);
		if (targets.empty()) continue;
		// Cell::merge_umis + Gene::merge(src, tgt) (Cell.cpp:31-42, Gene.cpp:38-58), in the map's own iteration order
		struct Folded { u32 reads, mark, row; };               // row: the molecule whose quality sums this one shows
		std::map<std::string, Folded> merged;                  // sequence -> molecule
		std::unordered_map<std::string, u64> code_of;
		for (const Mol &m : mols) { merged[m.seq] = Folded{m.reads, m.mark, m.row}; code_of[m.seq] = m.code; }
		const u32 cell = u32(GG.hk[GG.off[g]] >> (layout.umi_bits + layout.gene_bits));
		for (auto const &t : targets) {
			if (t.second == t.first) continue;
			auto s = merged.find(t.first);
			if (s == merged.end()) throw InvalidError("Source UMI doesn't belong to the gene: " + t.first);
			const Folded moved = s->second;
			auto it = merged.find(t.second);
			if (it == merged.end()) merged[t.second] = moved;   // a copy of the source, quality sums included (Gene.cpp:49)
			else { it->second.reads += moved.reads; it->second.mark |= moved.mark; }
			merged.erase(t.first);
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
		std::sort(ov.begin(), ov.end(), [](const UmiOverride &x, const UmiOverride &y) { return x.umi < y.umi; });
		p_idx.push_back(groups[g]); p_all.push_back(u32(ov.size())); p_req.push_back(n_req); p_rreq.push_back(reads_req);
		umi_overrides[GG.hk[GG.off[g]] >> layout.umi_bits] = std::move(ov);
	}
	umi_patch_groups(p_idx, p_all, p_req, p_rreq, umis_removed);
}
