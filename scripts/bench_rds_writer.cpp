// Times Rds::save (dropest_amd/csrc/host/rds_writer.cpp) on a list shaped like a C2 result: two dgCMatrix of `nnz` entries in all (12 bytes each
// in the serialisation), saturation_info's three vectors of `mol` entries (reads, cbs, umis), per-cell vectors.  Host-only.
//   g++ -O2 -std=c++17 scripts/bench_rds_writer.cpp dropest_amd/csrc/host/rds_writer.cpp -o /tmp/bench_rds -lz -pthread && /tmp/bench_rds 38000000 4000000
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../dropest_amd/csrc/host/rds_writer.h"

int main(int argc, char **argv) {
	using namespace Rds;
	const size_t nnz = argc > 1 ? size_t(atof(argv[1])) : 38000000, mol = argc > 2 ? size_t(atof(argv[2])) : 4000000, cols = 5000, genes = 30000;
	const std::string out = argc > 3 ? argv[3] : "/tmp/bench_rds.rds";
	for (unsigned threads : {1u, 4u, 16u}) {
		auto matrix = [&](size_t n) {
			std::vector<uint32_t> p(cols + 1), i(n), x(n);
			for (size_t c = 0; c <= cols; ++c) p[c] = uint32_t(c * n / cols);
			uint32_t s = 12345;
			for (size_t c = 0; c < cols; ++c) { uint32_t row = 0; for (uint32_t k = p[c]; k < p[c + 1]; ++k) { s = s * 1664525u + 1013904223u; row += 1 + (s >> 29); i[k] = row; x[k] = 1 + ((s >> 20) & 3) * ((s >> 27) == 0); } }
			std::vector<std::string> rn(genes), cn(cols);
			for (size_t g = 0; g < genes; ++g) { char b[32]; snprintf(b, sizeof b, "ENSG%011zu", g); rn[g] = b; }
			for (size_t c = 0; c < cols; ++c) cn[c] = "ACGTACGTACGTACGT";
			return dgCMatrix(std::move(p), std::move(i), std::move(x), rn, cn);
		};
		std::vector<int32_t> reads(mol); std::vector<std::string> cbs(mol), umis(mol);
		uint32_t s = 99;
		for (size_t k = 0; k < mol; ++k) { s = s * 1664525u + 1013904223u; reads[k] = 1 + int32_t(s >> 29); cbs[k] = "ACGTACGTACGTACGT"; cbs[k][k % 16] = "ACGT"[(s >> 8) & 3]; umis[k] = "ACGTACGTAC"; umis[k][k % 10] = "ACGT"[(s >> 10) & 3]; }
		auto v = named_list({{"cm", matrix(nnz / 2)}, {"cm_raw", matrix(nnz - nnz / 2)},
		                     {"saturation_info", named_list({{"reads", integers(std::move(reads))}, {"cbs", strings(std::move(cbs))}, {"umis", strings(std::move(umis))}})}});
		const auto t0 = std::chrono::steady_clock::now();
		save(v, out, threads);
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		FILE *f = fopen(out.c_str(), "rb"); fseek(f, 0, SEEK_END); const long bytes = ftell(f); fclose(f);
		printf("{\"threads\": %u, \"matrix_entries\": %zu, \"molecules\": %zu, \"save_ms\": %.1f, \"file_MB\": %.1f}\n", threads, nnz, mol, ms, bytes / 1e6);
		fflush(stdout);
	}
	return 0;
}
