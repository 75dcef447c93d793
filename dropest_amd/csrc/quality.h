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

__global__ __launch_bounds__(256) void quality_sums_kernel(const unsigned long long *__restrict__ umi, const uint32_t *__restrict__ gene,
                                                           const uint32_t *__restrict__ slot, uint32_t n, dropest::CbTable t,
                                                           dropest::KeyLayout L, const uint8_t *__restrict__ qual, uint32_t qlen,
                                                           const unsigned long long *__restrict__ mol_key, uint32_t n_mol,
                                                           uint32_t *__restrict__ qsum, uint32_t *__restrict__ missing,
                                                           const uint32_t *__restrict__ hot_slot) {
	const uint32_t stride = gridDim.x * 256;
	for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
		const uint32_t g = gene[r];
		if (g == dropest::NO_GENE) continue;                       // reads without a gene never reach Gene::add_umi
		const uint32_t sl = slot[r];   // the table slot, or CB_HOT_FLAG | index into the hot list (k_cbhash.h)
		const unsigned long long cell = t.slots[(sl & dropest::CB_HOT_FLAG) ? hot_slot[sl & ~dropest::CB_HOT_FLAG] : sl].cell_id;
		const unsigned long long u = umi[r];
		const unsigned long long ucode = (u & dropest::ESCAPE_BIT) ? (L.umi_escape_base + (u & ~dropest::ESCAPE_BIT)) : (u & L.umi_strip_mask);
		const unsigned long long k = (cell << (L.gene_bits + L.umi_bits)) | ((unsigned long long)g << L.umi_bits) | ucode;
		uint32_t lo = 0, hi = n_mol;
		while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (mol_key[mid] < k) lo = mid + 1; else hi = mid; }
		if (lo >= n_mol || mol_key[lo] != k) { atomicAdd(missing, 1u); continue; }
		// two positions per 64-bit atomic (rows are padded to an even number of sums; a sum stays below 2^32, so the low
		// half never carries into the high one): the kernel is bound by the number of L2 atomics
		const uint8_t *q = qual + size_t(r) * qlen;
		const uint32_t qstride = (qlen + 1u) & ~1u;
		unsigned long long *s = reinterpret_cast<unsigned long long *>(qsum + size_t(lo) * qstride);
		for (uint32_t i = 0; i < qlen; i += 2)
			atomicAdd(s + (i >> 1), (unsigned long long)q[i] | (i + 1 < qlen ? (unsigned long long)q[i + 1] << 32 : 0ull));
	}
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

__global__ __launch_bounds__(256) void gather_quality_rows_kernel(const uint32_t *__restrict__ rows, uint32_t n, const uint32_t *__restrict__ qrow,
                                                                  const uint32_t *__restrict__ qsum, uint32_t qlen, uint32_t *__restrict__ out) {
	const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
	if (i >= size_t(n) * qlen) return;
	const uint32_t m = uint32_t(i / qlen), p = uint32_t(i % qlen);
	out[i] = qsum[size_t(qrow[rows[m]]) * ((qlen + 1u) & ~1u) + p];
}

// add_umi_to_cell on a container with qualities: the molecule row that holds `key`, and a read's bytes added to a sums row
__global__ void find_molecule_row_kernel(const unsigned long long *__restrict__ mol_key, uint32_t n_mol, unsigned long long key, uint32_t *__restrict__ out) {
	uint32_t lo = 0, hi = n_mol;
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (mol_key[mid] < key) lo = mid + 1; else hi = mid; }
	*out = (lo < n_mol && mol_key[lo] == key) ? lo : 0xFFFFFFFFu;
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
	const size_t qstride = (size_t(qual_len) + 1) & ~size_t(1);   // padded to whole 64-bit pairs
	mol_qsum.ensure(size_t(n_mol) * qstride);
	HIP_CHECK(hipMemsetAsync(mol_qsum.p, 0, size_t(n_mol) * qstride * 4, stream));
	mol_qrow.ensure(n_mol);
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 4, stream));
	const u32 n = u32(n_reads);
	timed("quality_sums", double(n) * (20 + 5 * qual_len), [&] {
		hipLaunchKernelGGL(quality_sums_kernel, dim3(std::min<u32>(div_up(n, 256), 16384u)), dim3(256), 0, stream, d_umi, d_gene, slot.p, n, table,
		                   layout, umi_qual.p, qual_len, mol_key.p, n_mol, mol_qsum.p, scalars.p, hot_slot.p);
	});
	hipLaunchKernelGGL(iota_u32_kernel, dim3(div_up(n_mol, 256)), dim3(256), 0, stream, mol_qrow.p, n_mol);
	HIP_CHECK(hipGetLastError());
	u32 missing = 0;
	fetch(&missing, scalars.p, 4);
	if (missing) throw DeviceError("internal: " + std::to_string(missing) + " reads found no molecule row for their quality");
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

// Quality sums of a list of CURRENT molecule rows -> host, qual_len values per row.
void dropest_ctx::fetch_quality_rows(const std::vector<u32> &rows, uint32_t *out) {
	using namespace dropest;
	const u32 n = u32(rows.size());
	if (!n || !qual_len) return;
	DevBuf<u32> d_rows, d_out;
	d_rows.alloc(n); d_out.alloc(size_t(n) * qual_len);
	HIP_CHECK(hipMemcpyAsync(d_rows.p, rows.data(), size_t(n) * 4, hipMemcpyHostToDevice, stream));
	const size_t total = size_t(n) * qual_len;
	hipLaunchKernelGGL(gather_quality_rows_kernel, dim3(u32((total + 255) / 256)), dim3(256), 0, stream, d_rows.p, n, mol_qrow.p, mol_qsum.p,
	                   qual_len, d_out.p);
	HIP_CHECK(hipGetLastError());
	fetch(out, d_out.p, total * 4);
}
