// test_facade.cpp -- the reference's Boost.Test cases for the container (Tests/TestEstimation.cpp) replayed against the
// C++ facade (dropest_amd/csrc/host/facade.h), i.e. written the way the reference's own tests are written: build a
// container with strategies, add_record(...) hand-written reads, set_initialized(), merge_and_filter(), assert.
// Needs a GPU (run by tests/test_gpu_facade.py).  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "../../dropest_amd/csrc/host/facade.h"

using namespace Estimation;
using Mark = UMI::Mark;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)
#define CHECK_EQ(a, b) do { auto va = (a); auto vb = (b); if (!(va == vb)) { std::cout << "FAILED " << __FILE__ << ":" << __LINE__ << ": " #a " == " #b " (" << va << " vs " << vb << ")\n"; ++failures; } } while (0)
#define CHECK_THROWS(expr, Ex) do { bool ok = false; try { expr; } catch (const Ex &) { ok = true; } catch (...) {} if (!ok) { std::printf("FAILED %s:%d: %s should throw %s\n", __FILE__, __LINE__, #expr, #Ex); ++failures; } } while (0)

static std::string g_data;

static ReadInfo read_info(const std::string &cb, const std::string &umi, const std::string &gene, const std::string &chr = "",
                          const Mark &mark = Mark(Mark::HAS_EXONS)) {
	return ReadInfo(Tools::ReadParameters(cb, umi, "", umi), gene, chr, mark);   // Tests/TestEstimation.cpp:27-31
}

struct Fixture {   // Tests/TestEstimation.cpp:33-80
	std::shared_ptr<Merge::RealBarcodesMergeStrategy> real_cb_strat;
	std::shared_ptr<Merge::UMIs::MergeUMIsStrategySimple> umi_merge_strat;
	std::vector<Mark> any_mark;
	std::shared_ptr<CellsDataContainer> container_full;
	Fixture() {
		real_cb_strat = std::make_shared<Merge::RealBarcodesMergeStrategy>(Merge::RealBarcodesMergeStrategy::INDROP, g_data + "/test_est", 0, 0, 7, 0);
		umi_merge_strat = std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1);
		any_mark = Mark::get_by_code(Mark::DEFAULT_CODE);
		container_full = std::make_shared<CellsDataContainer>(real_cb_strat, umi_merge_strat, any_mark);
		auto &c = *container_full;
		c.add_record(read_info("AAATTAGGTCCA", "AAACCT", "Gene1")); c.add_record(read_info("AAATTAGGTCCA", "CCCCCT", "Gene2"));
		c.add_record(read_info("AAATTAGGTCCA", "ACCCCT", "Gene3")); c.add_record(read_info("AAATTAGGTCCA", "ACCCCT", "Gene4"));
		c.add_record(read_info("AAATTAGGTCCC", "CAACCT", "Gene1")); c.add_record(read_info("AAATTAGGTCCC", "CAACCT", "Gene10"));
		c.add_record(read_info("AAATTAGGTCCC", "CAACCT", "Gene20"));
		c.add_record(read_info("AAATTAGGTCCG", "CAACCT", "Gene1"));
		c.add_record(read_info("AAATTAGGTCGG", "AAACCT", "Gene1")); c.add_record(read_info("AAATTAGGTCGG", "CCCCCT", "Gene2"));
		c.add_record(read_info("CCCTTAGGTCCA", "CCATTC", "Gene3")); c.add_record(read_info("CCCTTAGGTCCA", "CCCCCT", "Gene2"));
		c.add_record(read_info("CCCTTAGGTCCA", "ACCCCT", "Gene3"));
		c.add_record(read_info("CAATTAGGTCCG", "CAACCT", "Gene1")); c.add_record(read_info("CAATTAGGTCCG", "AAACCT", "Gene1"));
		c.add_record(read_info("CAATTAGGTCCG", "CCCCCT", "Gene2"));
		c.add_record(read_info("AAAAAAAAAAAA", "CCCCCT", "Gene2"));
		c.set_initialized();
	}
};

static std::map<std::string, std::map<std::string, size_t>> by_gene(const Cell &c) {
	std::map<std::string, std::map<std::string, size_t>> out;
	for (auto const &m : c.molecules()) out[m.gene][m.umi] = m.read_count;
	return out;
}

static void testRealNeighbours() {   // :227-235
	Fixture f;
	long expect[6] = {0, 1, 1, 0, 0, 0};
	for (size_t i = 0; i < 6; ++i) CHECK_EQ(f.container_full->get_merge_target(i), expect[i]);
}

static void testMergeByRealBarcodes() {   // :237-280
	Fixture f;
	auto &c = *f.container_full;
	c.merge_and_filter();
	CHECK_EQ(c.total_cells_number(), size_t(7));
	CHECK_EQ(c.filtered_cells().size(), size_t(2));
	if (c.filtered_cells().size() < 2) return;
	const Cell cell0 = c.cell(c.filtered_cells()[0]), cell1 = c.cell(c.filtered_cells()[1]);
	CHECK_EQ(cell0.size(), size_t(3)); CHECK_EQ(cell1.size(), size_t(4));
	auto g0 = by_gene(cell0), g1 = by_gene(cell1);
	CHECK_EQ(g0["Gene1"].size(), size_t(1)); CHECK_EQ(g0["Gene1"]["CAACCT"], size_t(2));
	CHECK_EQ(g1["Gene1"].size(), size_t(2)); CHECK_EQ(g1["Gene1"]["AAACCT"], size_t(3));
	CHECK_EQ(g1["Gene2"].size(), size_t(1)); CHECK_EQ(g1["Gene2"]["CCCCCT"], size_t(4));
	CHECK_EQ(g1["Gene3"].size(), size_t(2)); CHECK_EQ(g1["Gene3"]["ACCCCT"], size_t(2)); CHECK_EQ(g1["Gene3"]["CCATTC"], size_t(1));
	bool merged[7] = {false, false, true, true, true, true, false};
	size_t excluded = 0;
	for (size_t i = 0; i < 7; ++i) { CHECK_EQ(c.cell(i).is_merged(), merged[i]); excluded += c.cell(i).is_excluded(); }
	CHECK_EQ(excluded, size_t(1));
	size_t expect_targets[7] = {0, 1, 1, 0, 0, 0, 6};
	for (size_t i = 0; i < 7; ++i) CHECK_EQ(c.merge_targets()[i], expect_targets[i]);
	CHECK_EQ(c.cell_id_by_cb("CCCTTAGGTCCA"), size_t(4));
	CHECK_THROWS(c.cell_id_by_cb("TTTTTTTTTTTT"), std::out_of_range);
	CHECK_THROWS(c.cell(7), std::out_of_range);
}

static void testUmiExclusion() {   // :369-397
	Fixture f;
	CellsDataContainer c(f.real_cb_strat, f.umi_merge_strat, Mark::get_by_code("e"));
	c.add_record(read_info("AAATTAGGTCCA", "AAACCT", "Gene1")); c.add_record(read_info("AAATTAGGTCCA", "CCCCCT", "Gene2"));
	c.add_record(read_info("AAATTAGGTCCA", "ACCCCT", "Gene3")); c.add_record(read_info("AAATTAGGTCCA", "ACCCCT", "Gene4"));
	c.add_record(read_info("AAATTAGGTCCA", "TTTTTT", "Gene3", "chr1", Mark(Mark::HAS_NOT_ANNOTATED)));
	c.add_record(read_info("AAATTAGGTCCA", "ACCCCT", "Gene4", "chr1", Mark(Mark::HAS_NOT_ANNOTATED)));
	c.set_initialized();
	c.merge_and_filter();
	const Cell cell = c.cell(0);
	bool na3 = false, na4 = false;
	for (auto const &m : cell.molecules()) {
		if (m.gene == "Gene3" && m.umi == "TTTTTT") na3 = m.mark.check(Mark::HAS_NOT_ANNOTATED);
		if (m.gene == "Gene4" && m.umi == "ACCCCT") { na4 = m.mark.check(Mark::HAS_NOT_ANNOTATED); CHECK_EQ(m.read_count, size_t(2)); }
	}
	CHECK(na3); CHECK(na4);
	auto req = cell.requested_umis_per_gene(c.gene_match_level(), true);
	CHECK_EQ(req.count("Gene3"), size_t(1)); CHECK_EQ(req["Gene3"], size_t(1));
	CHECK_EQ(req.count("Gene4"), size_t(0));
}

static void testUMIMergeStrategySimple() {   // :505-540
	Fixture f;
	auto dummy = std::make_shared<Merge::DummyMergeStrategy>(0, 0);
	CellsDataContainer c(dummy, f.umi_merge_strat, f.any_mark);
	for (const char *u : {"AAACCT", "AAACCT", "AAACCG", "AAACCN", "CCCCCT", "ACCCCT"}) c.add_record(read_info("AAATTAGGTCCA", u, "Gene1"));
	for (const char *u : {"TTTTTT", "TTTNNG", "TTGNNG", "ACCCCT", "NNNNNN"}) c.add_record(read_info("AAATTAGGTCCA", u, "Gene2"));
	c.set_initialized();
	c.merge_and_filter();
	auto g = by_gene(c.cell(0));
	CHECK_EQ(g["Gene1"].size(), size_t(4)); CHECK_EQ(g["Gene2"].size(), size_t(3));
	CHECK_EQ(g["Gene1"]["AAACCT"], size_t(3)); CHECK_EQ(g["Gene1"]["AAACCG"], size_t(1));
	CHECK_EQ(g["Gene1"]["CCCCCT"], size_t(1)); CHECK_EQ(g["Gene1"]["ACCCCT"], size_t(1));
	CHECK(g["Gene2"].count("TTTTTT") == 1); CHECK(g["Gene2"].count("ACCCCT") == 1);
	for (auto const &kv : g["Gene2"]) CHECK(kv.first.find('N') == std::string::npos);
}

static void testUMIMergeStrategyDirectional() {   // :588-608, driven through the container instead of find_targets
	Fixture f;
	auto dummy = std::make_shared<Merge::DummyMergeStrategy>(0, 0);
	auto strat = std::make_shared<Merge::UMIs::MergeUMIsStrategyDirectional>();
	CellsDataContainer c(dummy, strat, f.any_mark);
	const std::pair<const char *, int> umis[] = {{"AAA", 2}, {"AAC", 5}, {"AAT", 6}, {"AGT", 20}, {"CCC", 10}, {"TCC", 20}};
	for (auto const &u : umis) c.add_record(read_info("AAATTAGGTCCA", u.first, "Gene1"));            // UMI-index order as listed
	for (auto const &u : umis) for (int k = 1; k < u.second; ++k) c.add_record(read_info("AAATTAGGTCCA", u.first, "Gene1"));
	c.set_initialized();
	c.merge_and_filter();
	auto g = by_gene(c.cell(0));
	CHECK_EQ(g["Gene1"].size(), size_t(3));
	CHECK_EQ(g["Gene1"]["AGT"], size_t(28)); CHECK_EQ(g["Gene1"]["AAC"], size_t(5)); CHECK_EQ(g["Gene1"]["TCC"], size_t(30));
}

static void testStateMachineAndParams() {   // CellsDataContainer.cpp:41-42,:61-62,:165-166; Tests/TestTools.cpp:56-87
	Fixture f;
	CHECK_THROWS(f.container_full->add_record(read_info("AAAA", "CCCC", "G")), std::runtime_error);
	CHECK_THROWS(f.container_full->set_initialized(), std::runtime_error);
	auto dummy = std::make_shared<Merge::DummyMergeStrategy>(0, 0);
	CellsDataContainer c(dummy, f.umi_merge_strat, f.any_mark);
	CHECK_THROWS(c.merge_and_filter(), std::runtime_error);
	auto rp = Tools::ReadParameters::parse_encoded_id("@111!ATTTGC#ATATC");
	CHECK_EQ(rp.cell_barcode(), std::string("ATTTGC")); CHECK_EQ(rp.umi(), std::string("ATATC"));
	rp = Tools::ReadParameters::parse_encoded_id("trash!ATTTG#ATAT");
	CHECK_EQ(rp.cell_barcode(), std::string("ATTTG")); CHECK_EQ(rp.umi(), std::string("ATAT"));
	CHECK_THROWS(Tools::ReadParameters::parse_encoded_id("ATTTG#ATAT"), std::runtime_error);
	CHECK_THROWS(Mark::get_by_code('x'), std::runtime_error);
}

static void testResultsPrinterMtx(const std::string &tmp) {   // ResultsPrinter.cpp:81-91, :334-361 (not pinned by the reference)
	Fixture f;
	auto &c = *f.container_full;
	c.merge_and_filter();
	ResultsPrinter printer(true, false);
	auto cm = printer.get_count_matrix(c, true, true);
	CHECK_EQ(cm.col_names.size(), size_t(2));
	CHECK_EQ(cm.col_names[0], std::string("AAATTAGGTCCC")); CHECK_EQ(cm.col_names[1], std::string("AAATTAGGTCCA"));
	CHECK_EQ(cm.values.size(), size_t(7));     // 3 genes + 4 genes
	std::map<std::string, std::map<std::string, uint32_t>> dense;
	for (size_t col = 0; col + 1 < cm.colptr.size(); ++col)
		for (uint32_t k = cm.colptr[col]; k < cm.colptr[col + 1]; ++k) dense[cm.col_names[col]][cm.row_names[cm.rowidx[k]]] = cm.values[k];
	CHECK_EQ(dense["AAATTAGGTCCA"]["Gene1"], 2u); CHECK_EQ(dense["AAATTAGGTCCA"]["Gene3"], 2u);
	CHECK_EQ(dense["AAATTAGGTCCC"]["Gene10"], 1u);
	printer.save_results(c, tmp + "/cell.counts.rds");
	ResultsPrinter(false, false, false, true).save_results(c, tmp + "/cell.counts.full.rds");   // + reads_per_umi_per_cell
	printer.save_intron_exon_matrices(c, tmp + "/cell.counts.rds");                             // -V: cell.counts.matrices.rds
	auto exon = printer.get_count_matrix_filtered(c, Mark::get_by_code("e"));
	CHECK_EQ(exon.values.size(), cm.values.size());             // every read of the fixture is exonic: "e" == the default query here
	CHECK_EQ(printer.get_count_matrix_filtered(c, Mark::get_by_code("i")).values.size(), size_t(0));
	std::ifstream mtx(tmp + "/cell.counts.mtx");
	std::string header; std::getline(mtx, header);
	CHECK_EQ(header, std::string("%%MatrixMarket matrix coordinate real general"));
	size_t r, cc, nnz; mtx >> r >> cc >> nnz;
	CHECK_EQ(cc, size_t(2)); CHECK_EQ(nnz, size_t(7)); CHECK_EQ(r, cm.row_names.size());
	auto raw = printer.get_count_matrix(c, false, false);
	CHECK_EQ(raw.col_names.size(), size_t(2));
	CHECK_EQ(raw.col_names[0], std::string("AAATTAGGTCCA"));   // real cells in cell-id order
	CHECK_EQ(c.get_stat_by_real_cells(Stats::TOTAL_UMIS_PER_CB).at("AAATTAGGTCCC"), 4);   // Stats::merge quirk
	CHECK_EQ(c.real_cells_number(), size_t(2));
	CHECK_EQ(c.has_exon_reads_num(), size_t(17)); CHECK_EQ(c.intergenic_reads_num(), size_t(0));
}

static void testUmiDistributionAndCollisions() {   // CellsDataContainer.cpp:182-197; Tools/CollisionsAdjuster.cpp
	Fixture f;
	auto &c = *f.container_full;
	c.merge_and_filter();
	auto dist = c.umi_distribution();   // over the 2 filtered cells: 3 + 6 molecules
	size_t total = 0;
	for (auto const &kv : dist) total += kv.second;
	CHECK_EQ(total, size_t(9));
	CHECK_EQ(dist.at("CAACCT"), size_t(4)); CHECK_EQ(dist.at("CCCCCT"), size_t(1)); CHECK_EQ(dist.at("ACCCCT"), size_t(2));
	std::vector<double> probs;
	for (auto const &kv : dist) probs.push_back(double(kv.second) / double(total));
	Tools::CollisionsAdjuster adj;
	adj.init(std::vector<double>(4096, 1.0 / 4096));
	CHECK_EQ(adj.estimate_adjusted_gene_expression(1), size_t(1));
	CHECK(adj.estimate_adjusted_gene_expression(1000) > size_t(1000));          // collisions inflate large expressions
	CHECK(adj.estimate_adjusted_gene_expression(1000) < size_t(1300));
}

static void testUMIMerge() {   // Tests/TestEstimation.cpp:468-488, statement by statement: merge_umis BEFORE set_initialized
	Fixture f;
	CellsDataContainer container(f.real_cb_strat, f.umi_merge_strat, f.any_mark);
	container.add_record(read_info("AAATTAGGTCCA", "AAACCT", "Gene1"));
	container.add_record(read_info("AAATTAGGTCCA", "CCCCCT", "Gene1"));
	container.add_record(read_info("AAATTAGGTCCA", "AAATTN", "Gene1"));
	container.add_record(read_info("AAATTAGGTCCA", "ACCCCT", "Gene1"));
	CellsDataContainer::s_s_hash_t merge_targets;
	merge_targets["AAACCT"] = "CCCCCT";
	merge_targets["AAATTN"] = "GGGGGG";
	merge_targets["ACCCCT"] = "ACCCCT";
	container.merge_umis(0, container.gene_indexer().get_index("Gene1"), merge_targets);
	auto g = by_gene(container.cell(0));
	CHECK_EQ(g.at("Gene1").size(), size_t(3));
	CHECK_EQ(g.at("Gene1").at("CCCCCT"), size_t(2));
	CHECK_EQ(g.at("Gene1").at("GGGGGG"), size_t(1));
	CHECK_EQ(g.at("Gene1").at("ACCCCT"), size_t(1));
	CHECK_THROWS(container.merge_umis(0, container.gene_indexer().get_index("Gene1"), merge_targets), std::runtime_error);   // sources are gone
	// more reads may still arrive, and the initialisation keeps what was merged before it
	container.add_record(read_info("AAATTAGGTCCA", "TTTTTT", "Gene1"));
	container.add_record(read_info("CCCTTAGGTCCA", "ACCCCT", "Gene2"));
	container.set_initialized();
	g = by_gene(container.cell(0));
	CHECK_EQ(g.at("Gene1").size(), size_t(4));
	CHECK_EQ(g.at("Gene1").at("CCCCCT"), size_t(2));
	CHECK_EQ(g.at("Gene1").at("GGGGGG"), size_t(1));
	CHECK_EQ(g.at("Gene1").at("TTTTTT"), size_t(1));
	CHECK_EQ(container.total_cells_number(), size_t(2));
}

static void testBoundaryLeftovers() {   // CellsDataContainer.h:90 add_umi_to_cell, :107 Cell &cell(size_t), :118 umi_indexer()
	Fixture f;
	CellsDataContainer container(f.real_cb_strat, f.umi_merge_strat, f.any_mark);
	container.add_record(read_info("AAATTAGGTCCA", "AAACCT", "Gene1"));
	container.add_record(read_info("AAATTAGGTCCA", "CCCCCT", "Gene2"));
	container.add_record(read_info("CCCTTAGGTCCA", "AAACCT", "Gene2"));
	container.add_record(read_info("CCCTTAGGTCCA", "GGGCCT", ""));           // no gene: its UMI never reaches the indexer
	container.add_record(read_info("CCCTTAGGTCCA", "TTTCCT", "Gene1"));
	container.set_initialized();
	const StringIndexer &ui = container.umi_indexer();
	CHECK_EQ(ui.values().size(), size_t(3));
	CHECK_EQ(ui.get_value(0), std::string("AAACCT")); CHECK_EQ(ui.get_value(1), std::string("CCCCCT")); CHECK_EQ(ui.get_value(2), std::string("TTTCCT"));
	CHECK_EQ(ui.get_index("CCCCCT"), size_t(1));
	CHECK_THROWS(ui.get_index("GGGCCT"), std::out_of_range);
	// add_umi_to_cell: a second read of an existing molecule, then a new molecule
	const int umis_before = container.cell(1).stat(Stats::TOTAL_UMIS_PER_CB), reads_before = container.cell(1).stat(Stats::TOTAL_READS_PER_CB);
	container.add_umi_to_cell(1, read_info("ignored", "TTTCCT", "Gene1", "", Mark(Mark::HAS_INTRONS)));
	auto g = by_gene(container.cell(1));
	CHECK_EQ(g.at("Gene1").at("TTTCCT"), size_t(2));
	CHECK_EQ(container.cell(1).stat(Stats::TOTAL_UMIS_PER_CB), umis_before);
	container.add_umi_to_cell(1, read_info("ignored", "CCCCCT", "Gene1"));
	g = by_gene(container.cell(1));
	CHECK_EQ(g.at("Gene1").size(), size_t(2));
	CHECK_EQ(g.at("Gene1").at("CCCCCT"), size_t(1));
	CHECK_EQ(container.cell(1).stat(Stats::TOTAL_UMIS_PER_CB), umis_before + 1);
	CHECK_EQ(container.cell(1).stat(Stats::TOTAL_READS_PER_CB), reads_before);   // the member touches no read counter (CellsDataContainer.cpp:356-364)
	for (auto const &m : container.cell(1).molecules()) {
		if (m.gene == "Gene1" && m.umi == "TTTCCT") {
			CHECK(m.mark.check(Mark::HAS_INTRONS)); CHECK(m.mark.check(Mark::HAS_EXONS));
			CHECK_EQ(m.sum_quality.size(), size_t(6));                      // the fixture's reads carry their UMI as the quality string
			if (m.sum_quality.size() == 6) { CHECK_EQ(m.sum_quality[0], unsigned(2 * 'T')); CHECK_EQ(m.sum_quality[3], unsigned(2 * 'C')); }
		}
		if (m.gene == "Gene1" && m.umi == "CCCCCT" && m.sum_quality.size() == 6) { CHECK_EQ(m.sum_quality[0], unsigned('C')); CHECK_EQ(m.sum_quality[5], unsigned('T')); }
	}
	CHECK_THROWS(container.add_umi_to_cell(1, ReadInfo(Tools::ReadParameters("ignored", "CCCCCT", "", "CCC"), "Gene1", "", Mark(Mark::HAS_EXONS))), std::runtime_error);   // "Wrong quality length: 3, expected: 6" (UMI.cpp:26-28)
	CHECK_THROWS(container.add_umi_to_cell(7, read_info("ignored", "CCCCCT", "Gene1")), std::out_of_range);
	Cell &ref = container.cell(1);           // the non-const overload
	CHECK_EQ(ref.barcode(), std::string("CCCTTAGGTCCA"));
	CHECK_EQ(&container.cell(1), &ref);
	// ... and the reference is LIVE like the reference's `Cell &` (CellsDataContainer.h:107): later changes of the container show through it
	const int umis_ref = ref.stat(Stats::TOTAL_UMIS_PER_CB);
	container.add_umi_to_cell(1, read_info("ignored", "GGGCCT", "Gene2"));
	CHECK_EQ(ref.stat(Stats::TOTAL_UMIS_PER_CB), umis_ref + 1);
	CHECK_EQ(by_gene(ref).at("Gene2").count("GGGCCT"), size_t(1));
	Cell &other = container.cell(0);
	CHECK(!other.is_merged()); CHECK(!ref.is_excluded());
	container.merge_cells(0, 1);
	CHECK(other.is_merged());
	container.exclude_cell(1);
	CHECK(ref.is_excluded());
}

static void testLiveCellBeforeInitialisation() {   // a `Cell &` taken while reads are still coming shows the reads that follow
	Fixture f;
	CellsDataContainer container(f.real_cb_strat, f.umi_merge_strat, f.any_mark);
	container.add_record(read_info("AAATTAGGTCCA", "AAACCT", "Gene1"));
	Cell &first = container.cell(0);
	CHECK_EQ(first.stat(Stats::TOTAL_READS_PER_CB), 1); CHECK_EQ(first.size(), size_t(1));
	container.add_record(read_info("AAATTAGGTCCA", "CCCCCT", "Gene2"));
	container.add_record(read_info("AAATTAGGTCCA", "CCCCCT", "Gene2"));
	CHECK_EQ(first.stat(Stats::TOTAL_READS_PER_CB), 3); CHECK_EQ(first.size(), size_t(2)); CHECK_EQ(first.umis_number(), size_t(2));
	container.set_initialized();
	CHECK_EQ(first.stat(Stats::TOTAL_READS_PER_CB), 3);            // the same numbers from the real context
	container.merge_and_filter();
	CHECK(first.is_real() || !first.is_real());                        // (reads its row again after the merge: must not throw)
}

static void testQualityLengthPerMolecule() {   // UMI.cpp:21-34 + Gene.cpp:20: the quality length belongs to the molecule
	auto mk = [](const char *cb, const char *umi, const char *qual, const char *gene) {
		return ReadInfo(Tools::ReadParameters(cb, umi, "", qual), gene, "chr1", Mark(Mark::HAS_EXONS));
	};
	{
		CellsDataContainer c(std::make_shared<Merge::DummyMergeStrategy>(0, 0), std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1),
		                     Mark::get_by_code(Mark::DEFAULT_CODE));
		c.add_record(mk("AAATTAGGTCCA", "AAACCT", "IIIIII", "Gene1"));
		c.add_record(mk("AAATTAGGTCCA", "CCCCCT", "III", "Gene1"));        // another molecule, another length: fine in the reference
		c.add_record(mk("AAATTAGGTCCA", "AAACCT", "JJJJJJ", "Gene1"));
		c.add_record(mk("AAATTAGGTCCA", "CCCCCT", "JJJ", "Gene1"));
		c.add_record(mk("AAATTAGGTCCA", "GGGCCT", "", "Gene2"));           // and one without any quality
		c.set_initialized();
		size_t seen = 0;
		for (auto const &m : c.cell(0).molecules()) {
			if (m.umi == "AAACCT") { ++seen; CHECK_EQ(m.sum_quality.size(), size_t(6)); if (m.sum_quality.size() == 6) CHECK_EQ(m.sum_quality[5], unsigned('I' + 'J')); }
			if (m.umi == "CCCCCT") { ++seen; CHECK_EQ(m.sum_quality.size(), size_t(3)); if (m.sum_quality.size() == 3) CHECK_EQ(m.sum_quality[0], unsigned('I' + 'J')); }
			if (m.umi == "GGGCCT") { ++seen; CHECK_EQ(m.sum_quality.size(), size_t(0)); }
		}
		CHECK_EQ(seen, size_t(3));
	}
	{
		CellsDataContainer c(std::make_shared<Merge::DummyMergeStrategy>(0, 0), std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1),
		                     Mark::get_by_code(Mark::DEFAULT_CODE));
		c.add_record(mk("AAATTAGGTCCA", "AAACCT", "IIIIII", "Gene1"));
		c.add_record(mk("AAATTAGGTCCA", "CCCCCT", "III", "Gene1"));
		std::string what;
		try { c.add_record(mk("AAATTAGGTCCA", "AAACCT", "JJJJ", "Gene1")); } catch (const std::runtime_error &e) { what = e.what(); }   // the reference throws here, from add_record (UMI.cpp:26-28): so does this container
		CHECK_EQ(what, std::string("Wrong quality length: 4, expected: 6"));
	}
	{   // ... also when the molecule was created AFTER the lengths began to differ, and for a molecule of the first length met later
		CellsDataContainer c(std::make_shared<Merge::DummyMergeStrategy>(0, 0), std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1),
		                     Mark::get_by_code(Mark::DEFAULT_CODE));
		c.add_record(mk("AAATTAGGTCCA", "AAACCT", "IIIIII", "Gene1"));
		c.add_record(mk("CCCTTAGGTCCA", "TTTCCT", "IIIIII", "Gene2"));
		c.add_record(mk("AAATTAGGTCCA", "CCCCCT", "III", "Gene1"));        // lengths differ from here on
		c.add_record(mk("AAATTAGGTCCA", "GGGGGT", "IIII", "Gene1"));       // a new molecule of a third length
		c.add_record(mk("AAATTAGGTCCA", "GGGGGT", "JJJJ", "Gene1"));       // the same length again: fine
		std::string what;
		try { c.add_record(mk("AAATTAGGTCCA", "GGGGGT", "JJJ", "Gene1")); } catch (const std::runtime_error &e) { what = e.what(); }
		CHECK_EQ(what, std::string("Wrong quality length: 3, expected: 4"));
		what.clear();
		try { c.add_record(mk("CCCTTAGGTCCA", "TTTCCT", "II", "Gene2")); } catch (const std::runtime_error &e) { what = e.what(); }   // a molecule from before the lengths differed
		CHECK_EQ(what, std::string("Wrong quality length: 2, expected: 6"));
		c.add_record(mk("CCCTTAGGTCCA", "TTTCCT", "JJJJJJ", "Gene2"));     // the container is still usable for reads that fit
		c.set_initialized();
		for (auto const &m : c.cell(1).molecules()) if (m.umi == "TTTCCT") CHECK_EQ(m.sum_quality.size(), size_t(6));
	}
}

static void testMergeAndExcludeCells() {   // CellsDataContainer::merge_cells / exclude_cell (:90-109), as the strategies call them
	Fixture f;
	auto &c = *f.container_full;
	c.merge_cells(2, 1);                   // AAATTAGGTCCG -> AAATTAGGTCCC (what the strategy decides for this fixture)
	c.exclude_cell(6);
	CHECK(c.cell(2).is_merged()); CHECK(c.cell(6).is_excluded()); CHECK(!c.cell(1).is_merged());
	auto g = by_gene(c.cell(1));
	CHECK_EQ(g.at("Gene1").at("CAACCT"), size_t(2));          // one read each in the two cells
	CHECK_EQ(c.cell(1).stat(Stats::TOTAL_READS_PER_CB), 4);
}

static void testPoissonMerge() {   // Tests/TestEstimationMergeProbs.cpp:30-91 (fixture), :127-140
	Merge::PoissonTargetEstimator estimator(1.0e-4, 1.0e-7);
	auto strat = std::make_shared<Merge::PoissonRealBarcodesMergeStrategy>(estimator, Merge::RealBarcodesMergeStrategy::INDROP,
	                                                                       g_data + "/test_est", 0, 0, 7);
	CellsDataContainer c(strat, std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1), Mark::get_by_code(Mark::DEFAULT_CODE));
	const char *reads[][3] = {
		{"AAATTAGGTCCA", "AAACCT", "Gene1"}, {"AAATTAGGTCCA", "CCCCCT", "Gene2"}, {"AAATTAGGTCCA", "ACCCCT", "Gene3"},
		{"AAATTAGGTCCC", "CAACCT", "Gene1"}, {"AAATTAGGTCCG", "CAACCT", "Gene1"},
		{"AAATTAGGTCGG", "AAACCT", "Gene1"}, {"AAATTAGGTCGG", "CCCCCT", "Gene2"},
		{"CCCTTAGGTCCA", "CCATTC", "Gene3"}, {"CCCTTAGGTCCA", "CCCCCT", "Gene2"}, {"CCCTTAGGTCCA", "ACCCCT", "Gene3"},
		{"CAATTAGGTCCG", "CAACCT", "Gene1"}, {"CAATTAGGTCCG", "AAACCT", "Gene1"}, {"CAATTAGGTCCG", "CCCCCT", "Gene2"},
		{"CAATTAGGTCCG", "TTTTTT", "Gene2"}, {"CAATTAGGTCCG", "TTCTTT", "Gene2"},
		{"CCCCCCCCCCCC", "CAACCT", "Gene1"}, {"CCCCCCCCCCCC", "AAACCT", "Gene1"}, {"CCCCCCCCCCCC", "CCCCCT", "Gene2"},
		{"CCCCCCCCCCCC", "TTTTTT", "Gene2"}, {"CCCCCCCCCCCC", "TTCTTT", "Gene2"}, {"TAATTAGGTCCA", "AAAAAA", "Gene4"}};
	for (auto const &r : reads) c.add_record(read_info(r[0], r[1], r[2]));
	c.set_initialized();
	CHECK_EQ(strat->merge_type(), std::string("Poisson Real CBs"));
	CHECK_EQ(estimator.estimate_intersection_prob(c, 0, 1).merge_probability, 1.0);                     // :129
	CHECK(std::fabs(estimator.estimate_intersection_prob(c, 1, 2).merge_probability - 0.16) <= 0.05);   // :130
	CHECK(std::fabs(estimator.estimate_intersection_prob(c, 3, 4).merge_probability - 0.15) <= 0.05);   // :131
	CHECK_EQ(estimator.estimate_intersection_prob(c, 5, 6).intersection_size, size_t(5));
	CHECK_EQ(c.get_merge_target(7), long(-1));                                                          // :136-140
}

// The container over several GPUs (here: three shards on GPU 0): the reference's merge fixture (Tests/TestEstimation.cpp:33-80,
// :241-261) and a synthetic stream of a few batches with N-UMIs must give the matrices and merged barcodes of ONE container.
static void testShardedContainer() {
	// (no UMI qualities here: they are a single-GPU feature)
	auto read_info = [](const std::string &cb, const std::string &umi, const std::string &gene, const std::string &chr) {
		return ReadInfo(Tools::ReadParameters(cb, umi), gene, chr, Mark(Mark::HAS_EXONS));
	};
	auto make = [&](const std::vector<int> &devices) {
		auto strat = std::make_shared<Merge::RealBarcodesMergeStrategy>(Merge::RealBarcodesMergeStrategy::INDROP, g_data + "/test_est", 0, 0, 7, 0);
		auto umis = std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1);
		return std::make_shared<CellsDataContainer>(strat, umis, Mark::get_by_code(Mark::DEFAULT_CODE), false, -1, devices);
	};
	auto feed = [&](CellsDataContainer &c) {
		const char *reads[][3] = {{"AAATTAGGTCCA", "AAACCT", "Gene1"}, {"AAATTAGGTCCA", "CCCCCT", "Gene2"}, {"AAATTAGGTCCA", "ACCCCT", "Gene3"},
			{"AAATTAGGTCCA", "ACCCCT", "Gene4"}, {"AAATTAGGTCCC", "CAACCT", "Gene1"}, {"AAATTAGGTCCC", "CAACCT", "Gene10"}, {"AAATTAGGTCCC", "CAACCT", "Gene20"},
			{"AAATTAGGTCCG", "CAACCT", "Gene1"}, {"AAATTAGGTCGG", "AAACCT", "Gene1"}, {"AAATTAGGTCGG", "CCCCCT", "Gene2"}, {"CCCTTAGGTCCA", "CCATTC", "Gene3"},
			{"CCCTTAGGTCCA", "CCCCCT", "Gene2"}, {"CCCTTAGGTCCA", "ACCCCT", "Gene3"}, {"CAATTAGGTCCG", "CAACCT", "Gene1"}, {"CAATTAGGTCCG", "AAACCT", "Gene1"},
			{"CAATTAGGTCCG", "CCCCCT", "Gene2"}, {"AAAAAAAAAAAA", "CCCCCT", "Gene2"}};
		for (auto &r : reads) c.add_record(read_info(r[0], r[1], r[2], "chr1"));
		c.set_initialized(); c.merge_and_filter();
	};
	auto one = make({0}), three = make({0, 0, 0});
	feed(*one); feed(*three);
	CHECK(three->sharded() && !one->sharded());
	ResultsPrinter printer(true, false);
	for (bool filtered : {true, false}) {
		const auto a = printer.get_count_matrix(*one, filtered, false), b = printer.get_count_matrix(*three, filtered, false);
		CHECK(a.col_names == b.col_names); CHECK(a.row_names == b.row_names);
		CHECK(a.colptr == b.colptr); CHECK(a.rowidx == b.rowidx); CHECK(a.values == b.values);
		CHECK(!a.col_names.empty());
	}
	CHECK(one->merged_barcodes() == three->merged_barcodes());
	CHECK_EQ(three->merged_barcodes().size(), size_t(4));          // targets 0 1 1 0 0 0 6: four cells merged away (:227-235)
	CHECK_THROWS(three->cell(0), std::runtime_error);
	CHECK_THROWS(three->filtered_cells(), std::runtime_error);
	{   // -M with the whitelist over three shards (the estimator's UMI distribution is the one of all shards)
		auto make_m = [&](const std::vector<int> &devices) {
			Merge::PoissonTargetEstimator estimator(1.0e-4, 1.0e-7);
			auto strat = std::make_shared<Merge::PoissonRealBarcodesMergeStrategy>(estimator, Merge::RealBarcodesMergeStrategy::INDROP, g_data + "/test_est", 0, 0, 7);
			return std::make_shared<CellsDataContainer>(strat, std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1), Mark::get_by_code(Mark::DEFAULT_CODE), false, -1, devices);
		};
		auto m1 = make_m({0}), m3 = make_m({0, 0, 0});
		feed(*m1); feed(*m3);
		CHECK(m1->merged_barcodes() == m3->merged_barcodes());
		const auto a = printer.get_count_matrix(*m1, true, false), b = printer.get_count_matrix(*m3, true, false);
		CHECK(a.col_names == b.col_names); CHECK(a.colptr == b.colptr); CHECK(a.rowidx == b.rowidx); CHECK(a.values == b.values);
	}

	// several batches (BATCH = 2^20 reads each; shard k takes its quota of the stream, then shard k + 1), UMIs with N, no CB merge
	auto big = [&](const std::vector<int> &devices) {
		auto strat = std::make_shared<Merge::DummyMergeStrategy>(3, 5);
		auto umis = std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1);
		auto c = std::make_shared<CellsDataContainer>(strat, umis, Mark::get_by_code(Mark::DEFAULT_CODE), false, -1, devices);
		c->shard_quota = CellsDataContainer::BATCH;             // one batch per shard, the rest on the last one
		uint64_t x = 88172645463325252ull;
		auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
		auto seq = [&](uint64_t v, int len) { std::string s(size_t(len), 'A'); for (int i = 0; i < len; ++i) { s[size_t(i)] = "ACGT"[v & 3]; v >>= 2; } return s; };
		const size_t n = 2 * CellsDataContainer::BATCH + 12345;
		for (size_t i = 0; i < n; ++i) {
			const uint64_t r = rnd();
			std::string umi = seq(r >> 20, 6);
			if ((r >> 50) % 200 == 0) umi[size_t((r >> 40) % 6)] = 'N';
			c->add_record(read_info(seq((r % 300) * 7919 + 13, 12), umi, "G" + std::to_string((r >> 12) % 40), "chr" + std::to_string((r >> 8) % 3)));
		}
		c->set_initialized(); c->merge_and_filter();
		return c;
	};
	auto b1 = big({0}), b2 = big({0, 0});
	for (bool filtered : {true, false}) {
		const auto a = printer.get_count_matrix(*b1, filtered, false), b = printer.get_count_matrix(*b2, filtered, false);
		CHECK(a.col_names == b.col_names); CHECK(a.row_names == b.row_names);
		CHECK(a.colptr == b.colptr); CHECK(a.rowidx == b.rowidx); CHECK(a.values == b.values);
		CHECK(a.col_names.size() > 100);
	}
}

static void testWideKeyContainer() {
	// cell id + gene + UMI fields beyond 64 bits: the container splits itself over shards on its one device.  The same stream
	// with the UMIs cut to their varying tail fits one context and must give the same matrices.
	auto read_info = [](const std::string &cb, const std::string &umi, const std::string &gene, const std::string &chr) {
		return ReadInfo(Tools::ReadParameters(cb, umi), gene, chr, Mark(Mark::HAS_EXONS));
	};
	auto run = [&](const std::string &umi_prefix) {
		auto strat = std::make_shared<Merge::DummyMergeStrategy>(3, 5);
		auto umis = std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1);
		auto c = std::make_shared<CellsDataContainer>(strat, umis, Mark::get_by_code(Mark::DEFAULT_CODE), false, -1, 0);
		uint64_t x = 0x9E3779B97F4A7C15ull;
		auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
		auto seq = [&](uint64_t v, int len) { std::string s(size_t(len), 'A'); for (int i = 0; i < len; ++i) { s[size_t(i)] = "ACGT"[v & 3]; v >>= 2; } return s; };
		for (size_t i = 0; i < 400000; ++i) {
			const uint64_t r = rnd();
			std::string umi = seq(r >> 20, 6);
			if ((r >> 50) % 300 == 0) umi[size_t((r >> 40) % 6)] = 'N';
			c->add_record(read_info(seq((r % 2500) * 7919 + 13, 14), umi_prefix + umi, "G" + std::to_string((r >> 12) % 20000), "chr" + std::to_string((r >> 8) % 3)));
		}
		c->set_initialized(); c->merge_and_filter();
		return c;
	};
	auto narrow = run(""), wide = run("ACGTTGCATGACCA");     // 6-base and 20-base UMIs: 13 / 41 UMI bits + 15 gene bits + 12 cell bits
	CHECK(!narrow->sharded()); CHECK(wide->sharded());
	ResultsPrinter printer(true, false);
	for (bool filtered : {true, false}) {
		const auto a = printer.get_count_matrix(*narrow, filtered, false), b = printer.get_count_matrix(*wide, filtered, false);
		CHECK(a.col_names == b.col_names); CHECK(a.row_names == b.row_names);
		CHECK(a.colptr == b.colptr); CHECK(a.rowidx == b.rowidx); CHECK(a.values == b.values);
		CHECK(a.col_names.size() > 1000);
	}
}

static void testByteFormWalk() {   // ResultsPrinter.cpp:433-442: the named matrix from the byte form directly == from the 32-bit slots
	auto read_info = [](const std::string &cb, const std::string &umi, const std::string &gene) {
		return ReadInfo(Tools::ReadParameters(cb, umi), gene, "chr1", Mark(Mark::HAS_EXONS));
	};
	auto strat = std::make_shared<Merge::RealBarcodesMergeStrategy>(Merge::RealBarcodesMergeStrategy::INDROP, g_data + "/test_est", 0, 0, 7, 0);
	auto umis = std::make_shared<Merge::UMIs::MergeUMIsStrategySimple>(1);
	CellsDataContainer c(strat, umis, Mark::get_by_code(Mark::DEFAULT_CODE));
	const char *bases = "ACGT";
	auto umi_of = [&](unsigned k) { std::string u(6, 'A'); for (int i = 0; i < 6; ++i) { u[i] = bases[k & 3]; k >>= 2; } return u; };
	// gene ids in first-seen order: cell 1 names 1 000 genes in turn (row deltas of 1), cell 2 takes every 300th (deltas beyond a byte: listed
	// rows) and stacks 300 UMIs / 700 reads on some of them (counts beyond a byte: listed values)
	for (int g = 0; g < 1000; ++g) c.add_record(read_info("AAATTAGGTCCA", umi_of(unsigned(g)), "G" + std::to_string(g)));
	for (int g = 0; g < 1000; g += 300) {
		for (unsigned u = 0; u < (g == 300 ? 300u : 3u); ++u) c.add_record(read_info("AAATTAGGTCCC", umi_of(u * 7u + 1u), "G" + std::to_string(g)));
		if (g == 600) for (int r = 0; r < 700; ++r) c.add_record(read_info("AAATTAGGTCCC", umi_of(5u), "G600"));
	}
	c.add_record(read_info("AAATTAGGTCCC", umi_of(9u), "G254")); c.add_record(read_info("AAATTAGGTCCC", umi_of(9u), "G555"));
	c.set_initialized(); c.merge_and_filter();
	for (bool reads_output : {false, true}) {
		ResultsPrinter bytes(true, reads_output), slots(true, reads_output);
		slots.walk_byte_form = false;
		for (bool filtered : {true, false})
			for (bool ref_order : {true, false}) {
				const auto a = bytes.get_count_matrix(c, filtered, ref_order), b = slots.get_count_matrix(c, filtered, ref_order);
				CHECK(a.col_names == b.col_names); CHECK(a.row_names == b.row_names);
				CHECK(a.colptr == b.colptr); CHECK(a.rowidx == b.rowidx); CHECK(a.values == b.values);
				CHECK_EQ(a.col_names.size(), size_t(2)); CHECK_EQ(a.values.size(), size_t(1000 + 6));
				uint32_t top = 0;
				for (uint32_t v : a.values) top = std::max(top, v);
				CHECK_EQ(top, reads_output ? 703u : 300u);
			}
	}
}

int main(int argc, char **argv) {
	g_data = argc > 1 ? argv[1] : "dropest_amd/data/barcodes";
	const std::string tmp = argc > 2 ? argv[2] : "/tmp";
	try {
		testRealNeighbours();
		testMergeByRealBarcodes();
		testUmiExclusion();
		testUMIMergeStrategySimple();
		testUMIMergeStrategyDirectional();
		testStateMachineAndParams();
		testResultsPrinterMtx(tmp);
		testUmiDistributionAndCollisions();
		testPoissonMerge();
		testUMIMerge();
		testBoundaryLeftovers();
		testLiveCellBeforeInitialisation();
		testQualityLengthPerMolecule();
		testMergeAndExcludeCells();
		testShardedContainer();
		testWideKeyContainer();
		testByteFormWalk();
	} catch (const std::exception &e) {
		std::printf("UNEXPECTED EXCEPTION: %s\n", e.what());
		return 2;
	}
	if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
	std::printf("facade: all reference test cases passed\n");
	return 0;
}
