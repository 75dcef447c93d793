// BAM file(s) -> count matrices through the facade: BamController (native BGZF/BAM reader) -> CellsDataContainer ->
// ResultsPrinter::save_results.  Used by tests/test_gpu_bam.py and as the timing harness of the ingest path.
//   bam_to_counts <out_base> <filled|name> <min_genes_before> <min_genes_after> <whitelist|-> <threads> <bam> [<bam> ...]
//   environment: DROPEST_DEVICES = "0,0" shards the container over these devices; DROPEST_GTF = annotation file for -g (genes from the alignment positions instead of the gene tag)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../../dropest_amd/csrc/host/bam_ingest.h"

using namespace Estimation;

int main(int argc, char **argv) {
	if (argc < 8) { std::fprintf(stderr, "usage: %s out_base filled|name min_before min_after whitelist|- threads bam...\n", argv[0]); return 2; }
	try {
		const std::string out = argv[1], mode = argv[2], wl = argv[5];
		const size_t min_before = size_t(std::atoi(argv[3])), min_after = size_t(std::atoi(argv[4]));
		const unsigned threads = unsigned(std::atoi(argv[6]));
		std::vector<std::string> bams(argv + 7, argv + argc);
		std::shared_ptr<Merge::MergeStrategyAbstract> merge;
		if (wl == "-") merge = std::make_shared<Merge::DummyMergeStrategy>(min_before, min_after);
		else merge = std::make_shared<Merge::RealBarcodesMergeStrategy>(Merge::RealBarcodesMergeStrategy::CONST_LENGTH, wl, min_before, min_after, 7, 0.2);
		auto umi = std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1);
		// DROPEST_DEVICES = "0,0,0": the container sharded over these devices (several shards may share one GPU)
		std::vector<int> devices;
		if (const char *d = std::getenv("DROPEST_DEVICES")) { std::string t(d); size_t at = 0; while (at < t.size()) { devices.push_back(std::atoi(t.c_str() + at)); at = t.find(',', at); if (at == std::string::npos) break; ++at; } }
		if (devices.empty()) devices.push_back(0);
		CellsDataContainer c(merge, umi, UMI::Mark::get_by_code(UMI::Mark::DEFAULT_CODE), false, -1, devices);
		if (const char *q = std::getenv("DROPEST_SHARD_QUOTA")) c.shard_quota = size_t(std::atoll(q));
		BamProcessing::BamTags tags;
		tags.read_type = "RE"; tags.intronic_read_value = "N"; tags.intergenic_read_value = "I"; tags.exonic_read_value = "E";   // configs/10x.xml style
		const char *gtf = std::getenv("DROPEST_GTF");
		// mode: filled | name | params:<read-parameter files of droptag, space separated> ; DROPEST_MIN_PHRED = min_barcode_quality
		const std::string param_files = mode.rfind("params:", 0) == 0 ? mode.substr(7) : "";
		const char *phred = std::getenv("DROPEST_MIN_PHRED");
		BamProcessing::BamController ctl(tags, mode == "filled", param_files, gtf ? gtf : "", false, phred ? std::atoi(phred) + 33 : 0, threads);
		const auto t0 = std::chrono::steady_clock::now();
		ctl.parse_bam_files(bams, c);
		const auto t1 = std::chrono::steady_clock::now();
		c.set_initialized();
		c.merge_and_filter();
		const auto t2 = std::chrono::steady_clock::now();
		ResultsPrinter(true, false, false, std::getenv("DROPEST_RPUPC") != nullptr).save_results(c, out + ".rds");   // + reads_per_umi_per_cell
		const auto t3 = std::chrono::steady_clock::now();
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		const auto &k = ctl.counters();
		std::printf("{\"total_reads\": %zu, \"cant_parse\": %zu, \"low_quality\": %zu, \"saved\": %zu, \"cells\": %zu, \"real_cells\": %zu, "
		            "\"ingest_ms\": %.3f, \"ingest_wait_ms\": %.3f, \"ingest_parse_ms\": %.3f, \"ingest_add_ms\": %.3f, \"estimate_ms\": %.3f, \"write_ms\": %.3f}\n",
		            k.total_reads, k.cant_parse, k.low_quality, k.saved, c.sharded() ? size_t(0) : c.total_cells_number(), c.sharded() ? c.real_cells().size() : c.real_cells_number(), ms(t0, t1), k.wait_ms, k.parse_ms, k.add_ms, ms(t1, t2), ms(t2, t3));
	} catch (const std::exception &e) {
		std::fprintf(stderr, "ERROR: %s\n", e.what());
		return 1;
	}
	return 0;
}
