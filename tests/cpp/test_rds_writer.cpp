// Writes one .rds with every node type the results use (tests/test_rds.py parses it back).  Host-only.
#include <cstdio>
#include <limits>
#include <string>
#include <vector>
#include "../../dropest_amd/csrc/host/rds_writer.h"

int main(int argc, char **argv) {
	if (argc < 2) return 2;
	using namespace Rds;
	auto m = dgCMatrix({0, 2, 2, 5}, {0, 3, 1, 2, 3}, {7, 1, 2, 300000, 5}, {"g0", "g1", "g2", "g3"}, {"AAAC", "AAAG", "AAAT"});
	auto df = data_frame({"chr1", "chrX"}, {"AAAC", "AAAT"}, {{1, 2}, {30, 40}});
	auto d = named_list({
		{"cm", m},
		{"frame", df},
		{"named_int", with_names(integers({5, -7, 2147483647}), {"a", "b", "c"})},
		{"named_real", with_names(reals({1.5, -0.25, std::numeric_limits<double>::quiet_NaN()}), {"x", "y", "z"})},
		{"chars", strings({"ACGT", "", "g\xC3\xA9ne"})},
		{"nested", list({list({integers({3}), reals({})}), null_value()})},
		{"targets", named_list({{"AAAG", strings({"AAAC"})}})},
		{"empty", named_list({})},
	});
	save(d, argv[1]);
	if (argc > 2) {
		// long vectors of every kind (referenced, cut into ~2 MB pieces, swapped and deflated side by side: concatenated gzip members), written
		// twice -- by one thread and by many: the two files must hold the same serialisation
		const size_t n = 700001;
		std::vector<uint32_t> colptr(1001), rows(n), vals(n);
		for (size_t c = 0; c <= 1000; ++c) colptr[c] = uint32_t(c * n / 1000);
		for (size_t c = 0; c < 1000; ++c) for (uint32_t k = colptr[c]; k < colptr[c + 1]; ++k) { rows[k] = k - colptr[c]; vals[k] = uint32_t((k * 2654435761u) >> 12) + 1u; }
		std::vector<std::string> genes(701), cells(1000), many(300001);
		for (size_t g = 0; g < genes.size(); ++g) genes[g] = "G" + std::to_string(g);
		for (size_t c = 0; c < cells.size(); ++c) cells[c] = "C" + std::to_string(c);
		for (size_t k = 0; k < many.size(); ++k) many[k] = std::string(k % 37, char('A' + k % 4)) + std::to_string(k);
		std::vector<int32_t> ints(1000003);
		std::vector<double> dbl(500009);
		for (size_t k = 0; k < ints.size(); ++k) ints[k] = int32_t(k * 7919u) - 1000000;
		for (size_t k = 0; k < dbl.size(); ++k) dbl[k] = double(k) * 0.37 - 11.0;
		auto big = [&] {
			std::vector<uint64_t> codes(200003);      // packed base strings (sentinel bit + 2 bits per base), lengths 0..31
			for (size_t k = 0; k < codes.size(); ++k) { const int len = int(k % 32); uint64_t c = len ? 1 : 0; for (int b = 0; b < len; ++b) c = (c << 2) | ((k >> (b % 17)) & 3); codes[k] = c; }
			return named_list({{"cm", dgCMatrix(colptr, rows, vals, genes, cells)}, {"ints", integers(ints)}, {"reals", reals(dbl)}, {"strs", strings(many)},
			                   {"packed", strings_from_packed(codes)}, {"tail", integers({1, 2, 3})}});
		};
		save(big(), std::string(argv[2]) + ".one.rds", 1);
		save(big(), std::string(argv[2]) + ".many.rds", 8);
	}
	std::printf("ok\n");
	return 0;
}
