// Rate of the facade's add_record(ReadInfo) -- the call BamProcessor::save_read makes per read (BamProcessor.cpp:18-21): strings in,
// 2-bit codes + first-seen gene / chromosome ids + batches pushed to the device out.  bench.py quotes it under host_ingest.
//   add_record_rate <reads> [<cells> [<genes>]]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../../dropest_amd/csrc/host/facade.h"

using namespace Estimation;

int main(int argc, char **argv) {
	const size_t n = argc > 1 ? size_t(std::atof(argv[1])) : 4000000, n_cells = argc > 2 ? size_t(std::atoi(argv[2])) : 5000,
	             n_genes = argc > 3 ? size_t(std::atoi(argv[3])) : 30000;
	try {
		uint64_t x = 0x2545F4914F6CDD1Dull;
		auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
		auto seq = [&](uint64_t v, int len) { std::string s(size_t(len), 'A'); for (int i = 0; i < len; ++i) { s[size_t(i)] = "ACGT"[v & 3]; v >>= 2; } return s; };
		std::vector<std::string> cbs, genes, chrs;
		for (size_t i = 0; i < n_cells; ++i) cbs.push_back(seq(rnd(), 16));
		for (size_t i = 0; i < n_genes; ++i) { char b[32]; std::snprintf(b, sizeof b, "ENSG%011zu", i); genes.push_back(b); }
		for (int i = 0; i < 25; ++i) chrs.push_back("chr" + std::to_string(i + 1));
		auto merge = std::make_shared<Merge::DummyMergeStrategy>(20, 100);
		auto umis = std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1);
		CellsDataContainer c(merge, umis, UMI::Mark::get_by_code(UMI::Mark::DEFAULT_CODE));
		const auto t0 = std::chrono::steady_clock::now();
		for (size_t i = 0; i < n; ++i) {
			const uint64_t r = rnd();
			const size_t g = size_t((r >> 20) % n_genes) * size_t((r >> 40) % 7 + 1) / 7;      // skewed towards low gene ids
			c.add_record(ReadInfo(Tools::ReadParameters(cbs[size_t(r % n_cells)], seq(r >> 8, 10)), genes[g], chrs[g % 25], UMI::Mark(UMI::Mark::HAS_EXONS)));
		}
		const auto t1 = std::chrono::steady_clock::now();
		c.set_initialized();
		c.merge_and_filter();
		const auto t2 = std::chrono::steady_clock::now();
		auto s = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
		std::printf("{\"reads\": %zu, \"add_record_Mreads_per_s\": %.2f, \"add_record_plus_pass_Mreads_per_s\": %.2f, \"cells\": %zu}\n", n, double(n) / s(t0, t1) / 1e6,
		            double(n) / s(t0, t2) / 1e6, c.total_cells_number());
	} catch (const std::exception &e) {
		std::fprintf(stderr, "ERROR: %s\n", e.what());
		return 1;
	}
	return 0;
}
