// quality.h -- per-position UMI quality sums of the molecules (UMI::add_read, Estimation/UMI.cpp:21-34; read back
// as UMI::mean_quality, :46-55, by ResultsPrinter::get_reads_per_umi_per_cell, ResultsPrinter.cpp:261-314).
// Included by dropest_amd.hip.  Only active when the caller handed over qualities (dropest_set_umi_qualities).
//
// The sums are accumulated ONCE, per molecule row of the un-merged container (quality_sums_kernel: the read's key is
// looked up in the sorted molecule keys, its bytes are added with atomics).  Everything later only moves a row
// reference around, because the reference never adds qualities again after add_read:
//   * UMI::merge (UMI.cpp:15-19) adds read counts and marks only -- a molecule that receives another keeps ITS sums;
//   * Gene::merge(gene) (Gene.cpp:26-36) inserts a copy of a molecule the target lacks -- sums travel with it;
//   * Gene::merge(src, tgt) (Gene.cpp:38-58) creates a missing target as a copy of the source, else keeps the target's.
// So every current molecule carries `mol_qrow` = the original row whose sums it shows; a fold of several molecules
// (CB merge, directional UMI merge) picks the member the reference would have kept: the one with the smallest
// priority (reagg_prio: position of the member's cell in the merge order of its target, or "key unchanged" first).
#pragma once

namespace {

// The molecule row of a gene-bearing read (0xFFFFFFFF: none found -- an internal error the caller reports).
__device__ inline uint32_t quality_row_of_read(uint32_t r, const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene,
                                               const uint32_t *__restrict__ slot, const dropest::CbTable &t, const dropest::KeyLayout &L,
                                               const unsigned long long *__restrict__ mol_key, uint32_t n_mol, const uint32_t *__restrict__ hot_slot) {
	const uint32_t sl = slot[r];   // the table slot, or CB_HOT_FLAG | index into the hot list (k_cbhash.h)
	const unsigned long long cell = t.slots[(sl & dropest::CB_HOT_FLAG) ? hot_slot[sl & ~dropest::CB_HOT_FLAG] : sl].cell_id;
	const unsigned long long u = umi[r];
	const unsigned long long ucode = (u & dropest::ESCAPE_BIT) ? (L.umi_escape_base + (u & ~dropest::ESCAPE_BIT)) : (u & L.umi_strip_mask);
	const unsigned long long k = (cell << (L.gene_bits + L.umi_bits)) | ((unsigned long long)gene[r] << L.umi_bits) | ucode;
	uint32_t lo = 0, hi = n_mol;
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (mol_key[mid] < k) lo = mid + 1; else hi = mid; }
	return (lo < n_mol && mol_key[lo] == k) ? lo : 0xFFFFFFFFu;
}

// Quality strings of several lengths (dropest_set_umi_qualities_var): the length of a molecule is the length of the read that created it
// (Gene::add_umi, Gene.cpp:20: UMI(read_info.params.umi_quality().length(), 0)) -- its first read in stream order.  First pass: row per
// read and first read per row.
__global__ __launch_bounds__(256) void quality_first_read_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene,
                                                                 const uint32_t *__restrict__ slot, uint32_t n, dropest::CbTable t, dropest::KeyLayout L,
                                                                 const unsigned long long *__restrict__ mol_key, uint32_t n_mol,
                                                                 const uint32_t *__restrict__ hot_slot, uint32_t *__restrict__ read_row,
                                                                 uint32_t *__restrict__ first_read, uint32_t *__restrict__ missing) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
		read_row[r] = 0xFFFFFFFFu;
		if (gene[r] == dropest::NO_GENE) continue;
		const uint32_t row = quality_row_of_read(r, umi, gene, slot, t, L, mol_key, n_mol, hot_slot);
		if (row == 0xFFFFFFFFu) { atomicAdd(missing, 1u); continue; }
		read_row[r] = row;
		atomicMin(first_read + row, r);
	}
}

// Sums per molecule row.  lens == nullptr: every string has qlen characters.  Otherwise (read_row, first_read from the pass above) a read
// whose length differs from its molecule's is UMI::add_read's "Wrong quality length" (UMI.cpp:26-28): the earliest such read is reported.
__global__ __launch_bounds__(256) void quality_sums_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene,
                                                           const uint32_t *__restrict__ slot, uint32_t n, dropest::CbTable t,
                                                           dropest::KeyLayout L, const uint8_t *__restrict__ qual, uint32_t qlen, uint32_t qstride,
                                                           const unsigned long long *__restrict__ mol_key, uint32_t n_mol,
                                                           uint32_t *__restrict__ qsum, uint32_t *__restrict__ missing,
                                                           const uint32_t *__restrict__ hot_slot, const uint8_t *__restrict__ lens,
                                                           const uint32_t *__restrict__ read_row, const uint32_t *__restrict__ first_read,
                                                           uint32_t *__restrict__ wrong_length_read) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
		if (gene[r] == dropest::NO_GENE) continue;                       // reads without a gene never reach Gene::add_umi
		uint32_t lo, mine = qlen;
		if (lens) {
			lo = read_row[r];
			if (lo == 0xFFFFFFFFu) continue;
			mine = lens[r];
			if (mine != lens[first_read[lo]]) { atomicMin(wrong_length_read, r); continue; }
		} else {
			lo = quality_row_of_read(r, umi, gene, slot, t, L, mol_key, n_mol, hot_slot);
			if (lo == 0xFFFFFFFFu) { atomicAdd(missing, 1u); continue; }
		}
		// two positions per 64-bit atomic (rows are padded to an even number of words; a sum stays below 2^32, so the low
		// half never carries into the high one): the kernel is bound by the number of L2 atomics
		const uint8_t *q = qual + size_t(r) * qlen;
		unsigned long long *s = reinterpret_cast<unsigned long long *>(qsum + size_t(lo) * qstride);
		for (uint32_t i = 0; i < mine; i += 2)
			atomicAdd(s + (i >> 1), (unsigned long long)q[i] | (i + 1 < mine ? (unsigned long long)q[i + 1] << 32 : 0ull));
	}
}
// the last word of a sums row: the quality length of its molecule
__global__ __launch_bounds__(256) void quality_row_lengths_kernel(uint32_t *__restrict__ qsum, uint32_t n_rows, uint32_t qstride, uint32_t qlen,
                                                                  const uint8_t *__restrict__ lens, const uint32_t *__restrict__ first_read) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_rows) return;
	// (rows that no gene-bearing read maps to -- the per-chromosome rows of gene-less reads -- have no first read)
	qsum[size_t(i) * qstride + qstride - 1] = !lens ? qlen : first_read[i] == 0xFFFFFFFFu ? 0u : lens[first_read[i]];
}

__global__ __launch_bounds__(256) void iota_u32_kernel(uint32_t *out, uint32_t n) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = i;
}

// priority of an old molecule row in a CB merge = rank of its cell inside its target (0 = the target itself)
__global__ __launch_bounds__(256) void prio_from_cell_kernel(const unsigned long long *__restrict__ mol_key, uint32_t n, int cell_shift,
                                                             const uint32_t *__restrict__ cell_rank, uint32_t *__restrict__ prio) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) prio[i] = cell_rank[uint32_t(mol_key[i] >> cell_shift)];
}

// UMI re-keying inside a gene: a molecule whose key did not change is the existing target (keeps its sums)
__global__ __launch_bounds__(256) void prio_from_rekey_kernel(const unsigned long long *__restrict__ old_key,
                                                              const unsigned long long *__restrict__ new_key, uint32_t n,
                                                              uint32_t *__restrict__ prio) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) prio[i] = old_key[i] == new_key[i] ? 0u : 1u;
}

// sorted re-keyed members -> best (priority, old row) per new molecule row
__global__ __launch_bounds__(256) void member_best_kernel(const unsigned long long *__restrict__ sorted_key, const uint32_t *__restrict__ old_row,
                                                          uint32_t n_old, const unsigned long long *__restrict__ new_key, uint32_t n_new,
                                                          const uint32_t *__restrict__ prio, unsigned long long *__restrict__ best) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_old) return;
	const unsigned long long k = sorted_key[i];
	uint32_t lo = 0, hi = n_new;
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (new_key[mid] < k) lo = mid + 1; else hi = mid; }
	const uint32_t row = old_row[i];
	atomicMin(best + lo, ((unsigned long long)(prio ? prio[row] : 0u) << 32) | row);
}

__global__ __launch_bounds__(256) void take_best_qrow_kernel(const unsigned long long *__restrict__ best, uint32_t n_new,
                                                             const uint32_t *__restrict__ old_qrow, uint32_t *__restrict__ new_qrow) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n_new) new_qrow[i] = old_qrow[uint32_t(best[i] & 0xFFFFFFFFull)];
}

// out[m][0 .. qlen) = the sums of molecule row rows[m]; out_len[m] = its quality length (the last word of the sums row)
__global__ __launch_bounds__(256) void gather_quality_rows_kernel(const uint32_t *__restrict__ rows, uint32_t n, const uint32_t *__restrict__ qrow,
                                                                  const uint32_t *__restrict__ qsum, uint32_t qlen, uint32_t qstride,
                                                                  uint32_t *__restrict__ out, uint32_t *__restrict__ out_len) {
	const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
	if (i >= size_t(n) * (qlen + 1u)) return;
	const uint32_t m = uint32_t(i / (qlen + 1u)), p = uint32_t(i % (qlen + 1u));
	const uint32_t *row = qsum + size_t(qrow[rows[m]]) * qstride;
	if (p < qlen) { if (out) out[size_t(m) * qlen + p] = row[p]; }
	else if (out_len) out_len[m] = row[qstride - 1];
}

// add_umi_to_cell on a container with qualities: the molecule row that holds `key`, and a read's bytes added to a sums row
__global__ void find_molecule_row_kernel(const unsigned long long *__restrict__ mol_key, uint32_t n_mol, unsigned long long key, uint32_t *__restrict__ out) {
	uint32_t lo = 0, hi = n_mol;
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (mol_key[mid] < key) lo = mid + 1; else hi = mid; }
	*out = (lo < n_mol && mol_key[lo] == key) ? lo : 0xFFFFFFFFu;
}
__global__ void molecule_quality_length_kernel(const uint32_t *__restrict__ qsum, const uint32_t *__restrict__ qrow, const uint32_t *__restrict__ row,
                                               uint32_t qstride, uint32_t *__restrict__ out) {
	*out = *row == 0xFFFFFFFFu ? 0xFFFFFFFFu : qsum[size_t(qrow[*row]) * qstride + qstride - 1];
}
__global__ void add_quality_row_kernel(uint32_t *__restrict__ qsum, const uint32_t *__restrict__ qrow, const uint32_t *__restrict__ row, uint32_t qstride,
                                       const uint32_t *__restrict__ add, uint32_t qlen) {
	const uint32_t i = threadIdx.x;
	if (i < qlen && *row != 0xFFFFFFFFu) qsum[size_t(qrow[*row]) * qstride + i] += add[i];
}

}  // namespace

// After the first reduce of set_initialized: sums per molecule row, identity row references.
void dropest_ctx::accumulate_umi_qualities() {
	using namespace dropest;
	if (!have_qual) return;
	if (qual_reads != n_reads) throw InvalidError("UMI qualities were given for " + std::to_string(qual_reads) + " reads, the container holds " +
	                                              std::to_string(n_reads));
	n_mol_at_init = n_mol; n_qsum_rows = n_mol;
	if (!n_mol || !qual_len) return;
	HostStage hs(this, "umi_qualities");
	need_columns();   // (a sharded run's reads may still be packed records)
	const u32 qstride = qual_stride();   // the sums padded to whole 64-bit pairs, the molecule's quality length in the last word
	mol_qsum.ensure(size_t(n_mol) * qstride);
	HIP_CHECK(hipMemsetAsync(mol_qsum.p, 0, size_t(n_mol) * qstride * 4, stream));
	mol_qrow.ensure(n_mol);
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 4, stream));
	HIP_CHECK(hipMemsetAsync(scalars.p + 1, 0xFF, 4, stream));   // the earliest read with a wrong quality length
	const u32 n = u32(n_reads);
	const u32 grid = std::min<u32>(div_up(n, 256), 16384u);
	const uint8_t *lens = qual_var ? umi_qual_lens.p : nullptr;
	DevBuf<u32> read_row, first_read;
	if (qual_var) {
		read_row.alloc(n); first_read.alloc(n_mol);
		HIP_CHECK(hipMemsetAsync(first_read.p, 0xFF, size_t(n_mol) * 4, stream));
		timed("quality_first_read", double(n) * 24, [&] {
			hipLaunchKernelGGL(quality_first_read_kernel, dim3(grid), dim3(256), 0, stream, umi_key_column(), d_gene, slot.p, n, table, layout, mol_key.p, n_mol, hot_slot.p,
			                   read_row.p, first_read.p, scalars.p);
		});
	}
	timed("quality_sums", double(n) * (20 + 5 * qual_len), [&] {
		hipLaunchKernelGGL(quality_sums_kernel, dim3(grid), dim3(256), 0, stream, umi_key_column(), d_gene, slot.p, n, table,
		                   layout, umi_qual.p, qual_len, qstride, mol_key.p, n_mol, mol_qsum.p, scalars.p, hot_slot.p, lens, read_row.p, first_read.p, scalars.p + 1);
	});
	hipLaunchKernelGGL(quality_row_lengths_kernel, dim3(div_up(n_mol, 256)), dim3(256), 0, stream, mol_qsum.p, n_mol, qstride, qual_len, lens, first_read.p);
	hipLaunchKernelGGL(iota_u32_kernel, dim3(div_up(n_mol, 256)), dim3(256), 0, stream, mol_qrow.p, n_mol);
	HIP_CHECK(hipGetLastError());
	u32 res[2] = {0, 0};
	fetch(res, scalars.p, 8);
	if (res[0]) throw DeviceError("internal: " + std::to_string(res[0]) + " reads found no molecule row for their quality");
	if (res[1] != 0xFFFFFFFFu) {   // UMI::add_read, UMI.cpp:26-28: the reference throws at this read's add_record
		u32 row = 0, first = 0; uint8_t got = 0, expected = 0;
		fetch(&row, read_row.p + res[1], 4); fetch(&first, first_read.p + row, 4);
		fetch(&got, umi_qual_lens.p + res[1], 1); fetch(&expected, umi_qual_lens.p + first, 1);
		throw InvalidError("Wrong quality length: " + std::to_string(unsigned(got)) + ", expected: " + std::to_string(unsigned(expected)));
	}
}

// Called by reaggregate_from_keys with the sorted re-keyed members (sorted_key, old_row) and the new unique keys.
void dropest_ctx::requality_after_fold(const u64 *sorted_key, const u32 *old_row, u32 n_old, const u64 *new_key, u32 n_new) {
	using namespace dropest;
	if (!have_qual || !qual_len || !n_new) return;
	DevBuf<u64> best; best.alloc(n_new);
	HIP_CHECK(hipMemsetAsync(best.p, 0xFF, size_t(n_new) * 8, stream));
	mol_qrow2.ensure(std::max<size_t>(n_new, mol_qrow.n));
	hipLaunchKernelGGL(member_best_kernel, dim3(div_up(n_old, 256)), dim3(256), 0, stream, sorted_key, old_row, n_old, new_key, n_new,
	                   reagg_prio, best.p);
	hipLaunchKernelGGL(take_best_qrow_kernel, dim3(div_up(n_new, 256)), dim3(256), 0, stream, best.p, n_new, mol_qrow.p, mol_qrow2.p);
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(stream_wait(stream));
	std::swap(mol_qrow, mol_qrow2);
	reagg_prio = nullptr;
}

// Quality sums of a list of CURRENT molecule rows -> host, qual_len values per row (positions beyond a molecule's own length are 0),
// and / or the quality length of every row.
void dropest_ctx::fetch_quality_rows(const std::vector<u32> &rows, uint32_t *out, uint32_t *out_len) {
	using namespace dropest;
	const u32 n = u32(rows.size());
	if (!n || !qual_len) return;
	DevBuf<u32> d_rows, d_out, d_len;
	d_rows.alloc(n);
	if (out) d_out.alloc(size_t(n) * qual_len);
	if (out_len) d_len.alloc(n);
	HIP_CHECK(hipMemcpyAsync(d_rows.p, rows.data(), size_t(n) * 4, hipMemcpyHostToDevice, stream));
	const size_t total = size_t(n) * (qual_len + 1u);
	hipLaunchKernelGGL(gather_quality_rows_kernel, dim3(u32((total + 255) / 256)), dim3(256), 0, stream, d_rows.p, n, mol_qrow.p, mol_qsum.p,
	                   qual_len, qual_stride(), out ? d_out.p : nullptr, out_len ? d_len.p : nullptr);
	HIP_CHECK(hipGetLastError());
	if (out) fetch(out, d_out.p, size_t(n) * qual_len * 4);
	if (out_len) fetch(out_len, d_len.p, size_t(n) * 4);
}
