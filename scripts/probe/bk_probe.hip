// probe: what bounds build_keys?  Runs the product kernel (k_misc.h) and stripped variants of it over a synthetic stream with a chosen
// share of hot-barcode reads.  hipcc --offload-arch=gfx950 -O3 -I dropest_amd/csrc scripts/probe/bk_probe.hip -o scripts/probe/bk_probe
#include "k_misc.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace dropest;
using u64 = unsigned long long;
using u32 = uint32_t;

// variant A: pure stream -- same loads and stores, the cell id is the slot word itself, no statistics
template <int U>
__global__ __launch_bounds__(256) void stream_only(const u64 *__restrict__ umi, const u32 *__restrict__ gene, const u32 *__restrict__ aux,
                                                   const u32 *__restrict__ slot, u32 n, u64 *__restrict__ keys, uint8_t *__restrict__ vals) {
	const uint64_t stride = uint64_t(gridDim.x) * 256 * U;
	for (uint64_t base = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * U; base + U <= n; base += stride) {
#pragma unroll
		for (int q = 0; q < U; q += 4) {
			const uint4 s4 = *reinterpret_cast<const uint4 *>(slot + base + q), g4 = *reinterpret_cast<const uint4 *>(gene + base + q),
			            a4 = *reinterpret_cast<const uint4 *>(aux + base + q);
			const ulonglong2 u01 = *reinterpret_cast<const ulonglong2 *>(umi + base + q), u23 = *reinterpret_cast<const ulonglong2 *>(umi + base + q + 2);
			u64 k0 = (u64(s4.x) << 40) | (u64(g4.x) << 24) | (u01.x & 0xFFFFFF), k1 = (u64(s4.y) << 40) | (u64(g4.y) << 24) | (u01.y & 0xFFFFFF);
			u64 k2 = (u64(s4.z) << 40) | (u64(g4.z) << 24) | (u23.x & 0xFFFFFF), k3 = (u64(s4.w) << 40) | (u64(g4.w) << 24) | (u23.y & 0xFFFFFF);
			*reinterpret_cast<uint32_t *>(vals + base + q) = ((a4.x >> 16) & 0xFF) | (((a4.y >> 16) & 0xFF) << 8) | (((a4.z >> 16) & 0xFF) << 16) | (((a4.w >> 16) & 0xFF) << 24);
			*reinterpret_cast<ulonglong2 *>(keys + base + q) = make_ulonglong2(k0, k1);
			*reinterpret_cast<ulonglong2 *>(keys + base + q + 2) = make_ulonglong2(k2, k3);
		}
	}
}

int main(int argc, char **argv) {
	const u32 n = argc > 1 ? u32(atof(argv[1])) : 100000000u;
	const double hot_share = argc > 2 ? atof(argv[2]) : 0.93;
	const u32 n_cells = argc > 3 ? u32(atof(argv[3])) : 300000u;
	const u32 n_genes = 30000, n_hot = 4096; const u32 skew10 = argc > 4 ? u32(atoi(argv[4])) : 8;   // tenths of the other reads that fall on 50 000 cells
	u64 cap = 1; while (cap < 2ull * n_cells) cap <<= 1;
	std::vector<u64> h_umi(n); std::vector<u32> h_gene(n), h_aux(n), h_slot(n);
	u64 x = 88172645463325252ull;
	auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
	std::vector<u32> cell_slot(n_cells);
	for (auto &s : cell_slot) s = u32(rnd() & (cap - 1));
	for (u32 r = 0; r < n; ++r) {
		const u64 v = rnd();
		h_umi[r] = (1ull << 20) | (v & 0xFFFFF);
		const u32 g = u32((v >> 20) % n_genes);
		h_gene[r] = ((v >> 40) & 7) == 0 ? NO_GENE : g;
		h_aux[r] = (g % 25) | (2u << 16);
		const double p = double((v >> 43) & 0xFFFFF) / double(0x100000);
		h_slot[r] = p < hot_share ? (CB_HOT_FLAG | u32((v >> 8) % n_hot)) : cell_slot[((v >> 5) % 10 < skew10) ? (v >> 11) % (n_cells < 50000u ? n_cells : 50000u) : (v >> 11) % n_cells];
	}
	u64 *d_umi, *d_keys, *d_hot_key; u32 *d_gene, *d_aux, *d_slot, *d_hot_slot, *d_gene_chr; uint8_t *d_vals; CbSlot *d_slots; GlobalCounters *d_gc; IngestStats *d_st;
	hipMalloc(&d_umi, size_t(n) * 8); hipMalloc(&d_keys, size_t(n) * 8 + 64); hipMalloc(&d_gene, size_t(n) * 4); hipMalloc(&d_aux, size_t(n) * 4);
	hipMalloc(&d_slot, size_t(n) * 4); hipMalloc(&d_vals, size_t(n) * 4 + 64); hipMalloc(&d_slots, cap * sizeof(CbSlot)); hipMalloc(&d_gc, sizeof(GlobalCounters));
	hipMalloc(&d_st, sizeof(IngestStats)); hipMalloc(&d_hot_key, n_hot * 8); hipMalloc(&d_hot_slot, n_hot * 4); hipMalloc(&d_gene_chr, (1u << 20) * 4);
	hipMemcpy(d_umi, h_umi.data(), size_t(n) * 8, hipMemcpyHostToDevice); hipMemcpy(d_gene, h_gene.data(), size_t(n) * 4, hipMemcpyHostToDevice);
	hipMemcpy(d_aux, h_aux.data(), size_t(n) * 4, hipMemcpyHostToDevice); hipMemcpy(d_slot, h_slot.data(), size_t(n) * 4, hipMemcpyHostToDevice);
	hipMemset(d_slots, 0, cap * sizeof(CbSlot)); hipMemset(d_gc, 0, sizeof(GlobalCounters)); hipMemset(d_st, 0, sizeof(IngestStats));
	std::vector<u32> hs(n_hot); for (u32 j = 0; j < n_hot; ++j) hs[j] = cell_slot[j % n_cells];
	hipMemcpy(d_hot_slot, hs.data(), n_hot * 4, hipMemcpyHostToDevice);
	std::vector<u32> gc(1u << 20); for (u32 g = 0; g < gc.size(); ++g) gc[g] = g % 25;
	hipMemcpy(d_gene_chr, gc.data(), gc.size() * 4, hipMemcpyHostToDevice);
	KeyLayout L{}; L.umi_bits = 20; L.gene_bits = 15; L.cell_bits = 20; L.mark_shift = 0; L.val_bytes = 1; L.umi_strip_mask = 0xFFFFF; L.umi_escape_base = 1ull << 20; L.gene_none = (1u << 15) - 1;
	CbTable t{d_slots, cap - 1};
	CbHot hot{d_hot_key, d_hot_slot, n_hot};
	int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	auto time_it = [&](const char *name, double bytes, auto launch) {
		for (int w = 0; w < 2; ++w) launch();
		hipEventRecord(e0, 0);
		const int reps = 5;
		for (int i = 0; i < reps; ++i) launch();
		hipEventRecord(e1, 0); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
		printf("%-44s %8.3f ms  %7.0f GB/s\n", name, ms, bytes / ms / 1e6);
	};
	const double bytes = double(n) * (8 + 4 + 4 + 4 + 8 + 1);
	auto grid_of = [&](auto kernel, int mult) { int per = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, 256, 0); return u32(cus * per * mult); };
	printf("n=%u hot_share=%.2f cells=%u cap=%llu\n", n, hot_share, n_cells, cap);
	{ auto k = build_keys_kernel<256, 1, true, true, true>; const u32 g = grid_of(k, 1);
	  time_it("product: HOT STATS", bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d_umi, d_gene, d_aux, d_slot, n, t, L, d_keys, (void *)d_vals, d_gc, hot, d_gene_chr, 1u << 20, d_st, 0u); }); }
	for (u32 lg : {30000u}) {
		{ auto k = build_keys_kernel<256, 1, true, true, true, true>; int per = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, 256, lg); const u32 g = u32(cus * per);
		  char nm[64]; snprintf(nm, 64, "HOT STATS GCL 256thr lds_genes=%u (%d/CU)", lg, per);
		  time_it(nm, bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), lg, 0, d_umi, d_gene, d_aux, d_slot, n, t, L, d_keys, (void *)d_vals, d_gc, hot, d_gene_chr, 1u << 20, d_st, lg); }); }
		for (int per_cu : {1, 2}) { auto k = build_keys_kernel<256, 1, true, true, true, true>; const u32 g = u32(cus * per_cu);
		  char nm[64]; snprintf(nm, 64, "HOT STATS GCL 256thr %d/CU", per_cu);
		  time_it(nm, bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), lg, 0, d_umi, d_gene, d_aux, d_slot, n, t, L, d_keys, (void *)d_vals, d_gc, hot, d_gene_chr, 1u << 20, d_st, lg); }); }
		{ auto k = build_keys_kernel<512, 1, true, true, true, true>; const u32 g = u32(cus);
		  time_it("HOT STATS GCL 512thr 1/CU", bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(512), lg, 0, d_umi, d_gene, d_aux, d_slot, n, t, L, d_keys, (void *)d_vals, d_gc, hot, d_gene_chr, 1u << 20, d_st, lg); }); }
		{ auto k = build_keys_kernel<512, 1, true, true, true, true>; int per = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, 512, lg); const u32 g = u32(cus * per);
		  char nm[64]; snprintf(nm, 64, "HOT STATS GCL 512thr lds_genes=%u (%d/CU)", lg, per);
		  time_it(nm, bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(512), lg, 0, d_umi, d_gene, d_aux, d_slot, n, t, L, d_keys, (void *)d_vals, d_gc, hot, d_gene_chr, 1u << 20, d_st, lg); }); }
		{ auto k = build_keys_kernel<1024, 1, true, true, true, true>; int per = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, 1024, lg); const u32 g = u32(cus * per);
		  char nm[64]; snprintf(nm, 64, "HOT STATS GCL 1024thr lds_genes=%u (%d/CU)", lg, per);
		  time_it(nm, bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(1024), lg, 0, d_umi, d_gene, d_aux, d_slot, n, t, L, d_keys, (void *)d_vals, d_gc, hot, d_gene_chr, 1u << 20, d_st, lg); }); }
	}
	for (int div : {2, 4}) { auto k = build_keys_kernel<256, 1, true, true, false>; const u32 g = grid_of(k, 1) / div; char nm[64]; snprintf(nm, 64, "product: HOT grid/%d", div);
	  time_it(nm, bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d_umi, d_gene, d_aux, d_slot, n, t, L, d_keys, (void *)d_vals, d_gc, hot, d_gene_chr, 1u << 20, d_st, 0u); }); }
	{ auto k = build_keys_kernel<256, 1, true, true, false>; const u32 g = grid_of(k, 1);
	  time_it("product: HOT", bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d_umi, d_gene, d_aux, d_slot, n, t, L, d_keys, (void *)d_vals, d_gc, hot, d_gene_chr, 1u << 20, d_st, 0u); }); }
	{ auto k = stream_only<4>; for (int mult : {1, 2, 4}) { const u32 g = grid_of(k, mult); char nm[64]; snprintf(nm, 64, "stream only U=4 grid x%d (%u)", mult, g);
	  time_it(nm, bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d_umi, d_gene, d_aux, d_slot, n, d_keys, d_vals); }); } }
	{ auto k = stream_only<8>; const u32 g = grid_of(k, 1);
	  time_it("stream only U=8", bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d_umi, d_gene, d_aux, d_slot, n, d_keys, d_vals); }); }
	{ auto k = stream_only<4>; const u32 g = (n / 4 + 255) / 256;
	  time_it("stream only U=4 one tile per thread", bytes, [&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d_umi, d_gene, d_aux, d_slot, n, d_keys, d_vals); }); }
	return 0;
}
