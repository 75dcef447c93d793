// Writes one .rds with every node type the results use (tests/test_rds.py parses it back).  Host-only.
#include <cstdio>
#include <limits>
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
	std::printf("ok\n");
	return 0;
}
