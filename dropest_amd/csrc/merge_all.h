// merge_all.h -- MergeAllMergeStrategy (Estimation/Merge/MergeAllMergeStrategy.h:16-50, Estimation.Merge.merge_type = "all"):
// every filtered cell merges into the filtered cell with MORE UMIs whose barcode is nearest (Tools::edit_distance with
// skip_n = false and the band max_cb_merge_edit_distance, distance <= that), ties by the larger UMI count, then by the
// earlier position in the filtered order.  All pairs of filtered cells are compared -- on the device one block per base
// cell when every barcode is a clean code of one length (the banded distance then equals the Levenshtein distance
// whenever it is <= the band, k_umi_directional.h:42-43); otherwise on the host with the restated banded function.
// Included by dropest_amd.hip.
#pragma once

namespace {

constexpr int MA_THREADS = 256;

// bases: the filtered positions this launch decides (null = all of them; a shard of a sharded run decides its own cells)
__global__ __launch_bounds__(MA_THREADS) void merge_all_kernel(const unsigned long long *__restrict__ code, const int32_t *__restrict__ umis,
                                                               uint32_t n, int len, uint32_t max_ed, const uint32_t *__restrict__ bases,
                                                               uint32_t *__restrict__ target_pos) {
	__shared__ unsigned long long wave_best[MA_THREADS / 64];
	const uint32_t f = bases ? bases[blockIdx.x] : blockIdx.x;
	const unsigned long long base = code[f];
	const int32_t base_umis = umis[f];
	unsigned long long best = ~0ull;
	for (uint32_t j = threadIdx.x; j < n; j += MA_THREADS) {
		const int32_t u = umis[j];
		if (u <= base_umis) continue;                                    // (:24-26)
		const uint32_t ed = dropest::umi_code_distance(base, code[j], len, max_ed);
		if (ed > max_ed) continue;                                       // (:31-32)
		// smaller distance, then more UMIs, then earlier in the filtered order (:34-44; updates are strict)
		const unsigned long long key = ((unsigned long long)ed << 56) | ((unsigned long long)(0x7FFFFFFF - u) << 25) | j;
		if (key < best) best = key;
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_down(best, d, 64); if (o < best) best = o; }
	if (dropest::lane_id() == 0) wave_best[dropest::wave_id()] = best;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < MA_THREADS / 64; ++w) if (wave_best[w] < best) best = wave_best[w];
		target_pos[blockIdx.x] = best == ~0ull ? 0xFFFFFFFFu : uint32_t(best & 0x1FFFFFFull);
	}
}

// Tools::edit_distance(s1, s2, false, max_ed) (Tools/UtilFunctions.cpp:32-65)
unsigned banded_edit_distance_exact(const std::string &s1, const std::string &s2, unsigned max_ed) {
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
			const bool match = s1[size_t(i - 1)] == s2[size_t(j - 1)];
			const int v = std::min(std::min(column[size_t(i)] + 1, column[size_t(i - 1)] + 1), lastdiag + int(!match));
			min_ed = std::min(min_ed, v + std::abs(i - j));
			column[size_t(i)] = v;
			lastdiag = olddiag;
		}
		if (min_ed > int(max_ed)) return unsigned(min_ed);
	}
	return unsigned(column[size_t(n1)]);
}

}  // namespace

// target_pos[k] = filtered position of the target of base bases[k] (all F cells when bases is null), 0xFFFFFFFF = none.
// code / umis: barcode codes and TOTAL_UMIS in filtered order; text(f) = the barcode string of position f.
void dropest_ctx::merge_all_targets(std::vector<u64> code, const std::vector<int32_t> &umis, const std::function<std::string(u32)> &text,
                                    const std::vector<u32> *bases, std::vector<u32> &target_pos) {
	using namespace dropest;
	const u32 F = u32(code.size()), NB = bases ? u32(bases->size()) : F;
	if (F >= (1u << 25)) throw UnsupportedError("merge_type = all over more than 2^25 filtered cells");
	const u32 max_ed = u32(std::max(cfg.max_cb_merge_edit_distance, 0));
	bool uniform = true;
	int len = -1;
	for (u32 f = 0; f < F; ++f) {
		if (code[f] & ESCAPE_BIT) { uniform = false; continue; }
		const int l = (63 - __builtin_clzll(code[f])) / 2;              // bases below the sentinel bit
		if (len < 0) len = l;
		uniform &= l == len;
	}
	target_pos.assign(NB, 0xFFFFFFFFu);
	if (!NB) return;
	if (uniform && len > 0 && len <= 31 && max_ed < 200) {
		HostStage hs2(this, "cb_merge:targets");
		const u64 strip = (1ull << (2 * len)) - 1ull;
		for (u64 &c : code) c &= strip;
		DevBuf<u64> d_code; DevBuf<int32_t> d_umis; DevBuf<u32> d_tgt, d_bases;
		d_code.alloc(F); d_umis.alloc(F); d_tgt.alloc(NB);
		HIP_CHECK(hipMemcpyAsync(d_code.p, code.data(), size_t(F) * 8, hipMemcpyHostToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(d_umis.p, umis.data(), size_t(F) * 4, hipMemcpyHostToDevice, stream));
		if (bases) { d_bases.alloc(NB); HIP_CHECK(hipMemcpyAsync(d_bases.p, bases->data(), size_t(NB) * 4, hipMemcpyHostToDevice, stream)); }
		timed("merge_all", double(NB) * F * 12, [&] {
			hipLaunchKernelGGL(merge_all_kernel, dim3(NB), dim3(MA_THREADS), 0, stream, d_code.p, d_umis.p, F, len, max_ed,
			                   bases ? d_bases.p : static_cast<const u32 *>(nullptr), d_tgt.p);
		});
		fetch(target_pos.data(), d_tgt.p, size_t(NB) * 4);
	} else {
		HostStage hs2(this, "cb_merge:targets_host");
		if (F > 30000) throw UnsupportedError("merge_type = all with barcodes of several lengths / with N over more than 30000 filtered cells");
		std::vector<std::string> txt(F);
		for (u32 f = 0; f < F; ++f) txt[f] = text(f);
		for (u32 k = 0; k < NB; ++k) {
			const u32 f = bases ? (*bases)[k] : k;
			int min_ed = std::numeric_limits<int>::max(), max_umi = 0;
			for (u32 j = 0; j < F; ++j) {
				if (umis[j] <= umis[f]) continue;
				const int ed = int(banded_edit_distance_exact(txt[f], txt[j], max_ed));
				if (ed > int(max_ed)) continue;
				if (min_ed > ed) { min_ed = ed; max_umi = umis[j]; target_pos[k] = j; }
				else if ((min_ed == ed) & (max_umi < umis[j])) { max_umi = umis[j]; target_pos[k] = j; }
			}
		}
	}
}

void dropest_ctx::run_cb_merge_all() {
	using namespace dropest;
	HostStage hs(this, "cb_merge");
	const std::vector<uint64_t> &order = filtered_cells();
	std::vector<u32> cells(order.begin(), order.end());
	const std::vector<u32> ridx = filtered_ridx;
	const u32 F = u32(cells.size()), nR = u32(real.size());
	clear_strategy_pairs();
	if (F == 0) return;
	std::vector<u64> code(F);
	std::vector<int32_t> umis(F);
	for (u32 f = 0; f < F; ++f) { code[f] = u64(real[ridx[f]].row.barcode); umis[f] = real[ridx[f]].row.total_umis; }
	std::vector<u32> target_pos;
	merge_all_targets(code, umis, [&](u32 f) { return barcode_of(real[ridx[f]]); }, nullptr, target_pos);

	// MergeStrategyBase::merge_inited second loop
	HostStage hs3(this, "cb_merge:apply");
	std::vector<int64_t> target(F);
	for (u32 f = 0; f < F; ++f) target[f] = int64_t(target_pos[f] == 0xFFFFFFFFu ? ridx[f] : ridx[target_pos[f]]);
	std::vector<int32_t> reads(nR), tu(nR);
	for (u32 i = 0; i < nR; ++i) { reads[i] = real[i].row.total_reads; tu[i] = real[i].row.total_umis; }
	std::vector<u32> cur(nR), rank(nR);
	std::vector<uint8_t> excl(nR);
	const bool any_merge = apply_merge_order(nR, F, ridx.data(), target.data(), reads.data(), tu.data(), cur.data(), excl.data(), rank.data());
	if (have_qual) {   // only the quality sums need the merge order (quality.h)
		merge_rank.assign(n_cells, 0);
		for (u32 i = 0; i < nR; ++i) merge_rank[real[i].id] = rank[i];
	}
	real_pristine = false;
	for (u32 i = 0; i < nR; ++i) {
		real[i].row.total_reads = reads[i]; real[i].row.total_umis = tu[i];
		if (cur[i] != i) { real[i].merged = true; merge_pairs.emplace_back(real[i].id, real[cur[i]].id); }
	}
	if (any_merge) reaggregate_after_merge();
}
