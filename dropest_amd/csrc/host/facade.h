// facade.h -- the reference's C++ surface for this path, re-created on top of the C-ABI (include/dropest_amd.h).
//
// A dropEst maintainer swaps Estimation/CellsDataContainer.h + Estimation/ResultsPrinter.h for this header:
// same namespaces, class names, member names, argument meaning and exceptions (std::runtime_error for
// call-order errors, std::out_of_range for bad indices), so dropest.cpp:239-254 / BamProcessor.cpp:18-21 compile
// against it unchanged in spirit.  Everything below is plain host C++ that only PACKS
// records (2-bit codes, first-seen dictionaries for gene / chromosome names) and RELAYS results; all counting,
// de-duplication and merging happens on the GPU behind the C-ABI.
//
// Reference declarations mirrored here:
//   Tools::ReadParameters           Tools/ReadParameters.h:9-50
//   Estimation::UMI::Mark           Estimation/UMI.h:13-44
//   Estimation::ReadInfo            Estimation/ReadInfo.h:9-24
//   Estimation::Stats (enums)       Estimation/Stats.h:18-32
//   Estimation::Merge::*Strategy    Estimation/Merge/{MergeStrategyAbstract.h,DummyMergeStrategy.h,RealBarcodesMergeStrategy.h}
//   Estimation::CellsDataContainer  Estimation/CellsDataContainer.h:33-123
//   Estimation::ResultsPrinter      Estimation/ResultsPrinter.h:59-62 (count matrices + MatrixMarket; .rds is "next")
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../../include/dropest_amd.h"

namespace Rds { struct Value; }   // rds_writer.h

namespace Tools {
class ReadParameters {
	std::string _cb, _umi, _cb_quality, _umi_quality;
public:
	ReadParameters(const std::string &cell_barcode, const std::string &umi, const std::string &cell_barcode_quality = "",
	               const std::string &umi_quality = "")
		: _cb(cell_barcode), _umi(umi), _cb_quality(cell_barcode_quality), _umi_quality(umi_quality) {
		if (cell_barcode.empty() || umi.empty())
			throw std::runtime_error("Wrong read parameters: '" + cell_barcode + "' '" + umi + "'");
	}
	static const char quality_offset = 33;                                     // ReadParameters.h:21
	static ReadParameters parse_encoded_id(const std::string &encoded_id);   // "<id>!<CB>#<UMI>"
	const std::string &cell_barcode() const { return _cb; }
	const std::string &umi() const { return _umi; }
	const std::string &cell_barcode_quality() const { return _cb_quality; }
	const std::string &umi_quality() const { return _umi_quality; }
};
// Tools::CollisionsAdjuster (Tools/CollisionsAdjuster.h:8-33); the table is computed on the GPU
// (dropest_collisions_adjusted_sizes) and extended on demand like the reference does.
class CollisionsAdjuster {
	std::vector<size_t> _adjusted_sizes;
	std::vector<double> _umi_probabilities;
	int _device;
	void update_adjusted_sizes(size_t max_gene_expression);
public:
	using probs_vec_t = std::vector<double>;
	explicit CollisionsAdjuster(int device = 0) : _device(device) {}
	void init(const probs_vec_t &umi_probabilities, size_t max_gene_expression = 0);
	size_t estimate_adjusted_gene_expression(size_t expression);
};
}  // namespace Tools

namespace Estimation {

class UMI {
public:
	class Mark {
		char _mark;
	public:
		enum MarkType { NONE = 0, HAS_NOT_ANNOTATED = 1, HAS_EXONS = 2, HAS_INTRONS = 4 };
		using query_t = std::vector<Mark>;
		static const std::string DEFAULT_CODE;   // "eEBA"
		explicit Mark(MarkType type = NONE) : _mark(char(type)) {}
		void add(MarkType type) { _mark = char(_mark | type); }
		void add(const Mark &m) { _mark = char(_mark | m._mark); }
		bool check(MarkType type) const { return (_mark & type) != 0; }
		bool match(const query_t &levels) const { for (auto const &l : levels) if (l._mark == _mark) return true; return false; }
		bool operator==(const MarkType &o) const { return _mark == o; }
		bool operator==(const Mark &o) const { return _mark == o._mark; }
		char bits() const { return _mark; }
		static Mark get_by_code(char code);
		static query_t get_by_code(const std::string &code);
		static std::string to_code(const query_t &levels);
	};
};

class ReadInfo {
public:
	const Tools::ReadParameters params;
	const std::string gene, chromosome_name;
	const UMI::Mark umi_mark;
	ReadInfo(const Tools::ReadParameters &p, const std::string &g, const std::string &chr, const UMI::Mark &m)
		: params(p), gene(g), chromosome_name(chr), umi_mark(m) {}
};

struct Stats {
	enum CellStatType { TOTAL_READS_PER_CB, TOTAL_UMIS_PER_CB, CELL_STAT_SIZE };
	enum CellChrStatType { EXON_READS_PER_CHR_PER_CELL = 0, INTRON_READS_PER_CHR_PER_CELL, INTERGENIC_READS_PER_CHR_PER_CELL, CHROMOSOME_STAT_SIZE };
};

class StringIndexer {   // read-only view of a first-seen dictionary
	std::vector<std::string> _values;
	std::unordered_map<std::string, size_t> _index;
public:
	using index_t = size_t;
	index_t add(const std::string &v) { auto it = _index.emplace(v, _index.size()); if (it.second) _values.push_back(v); return it.first->second; }
	const std::string &get_value(index_t i) const { return _values.at(i); }
	index_t get_index(const std::string &v) const { return _index.at(v); }
	const std::vector<std::string> &values() const { return _values; }
};

class CellsDataContainer;

namespace Merge {
// Parameter carriers: the strategy objects of the reference own the merge algorithm; here the algorithm lives on the
// device and these classes only say WHICH strategy with WHICH thresholds (MergeStrategyFactory.cpp:61-111).
class MergeStrategyAbstract {
	size_t _min_before, _min_after;
public:
	MergeStrategyAbstract(size_t min_genes_before_merge, size_t min_genes_after_merge)
		: _min_before(min_genes_before_merge), _min_after(std::max(min_genes_after_merge, min_genes_before_merge)) {}
	virtual ~MergeStrategyAbstract() = default;
	virtual std::string merge_type() const = 0;
	virtual void fill(dropest_cfg &cfg) const = 0;
	size_t min_genes_before_merge() const { return _min_before; }
	size_t min_genes_after_merge() const { return _min_after; }
};
class DummyMergeStrategy : public MergeStrategyAbstract {
public:
	using MergeStrategyAbstract::MergeStrategyAbstract;
	std::string merge_type() const override { return "No"; }
	void fill(dropest_cfg &cfg) const override { cfg.merge_kind = DROPEST_MERGE_NONE; }
};
// Estimation/Merge/SimpleMergeStrategy.h (-m without a barcodes file)
class SimpleMergeStrategy : public MergeStrategyAbstract {
	unsigned _max_ed; double _min_fraction;
public:
	SimpleMergeStrategy(size_t min_genes_before_merge, size_t min_genes_after_merge, unsigned max_merge_edit_distance, double min_merge_fraction)
		: MergeStrategyAbstract(min_genes_before_merge, min_genes_after_merge), _max_ed(max_merge_edit_distance), _min_fraction(min_merge_fraction) {}
	std::string merge_type() const override { return "Simple"; }
	void fill(dropest_cfg &cfg) const override {
		cfg.merge_kind = DROPEST_MERGE_SIMPLE; cfg.max_cb_merge_edit_distance = int(_max_ed); cfg.min_merge_fraction = _min_fraction;
	}
};
// Estimation/Merge/MergeAllMergeStrategy.h (merge_type = "all")
class MergeAllMergeStrategy : public MergeStrategyAbstract {
	unsigned _max_ed;
public:
	MergeAllMergeStrategy(size_t min_genes_before_merge, size_t min_genes_after_merge, unsigned max_merge_edit_distance)
		: MergeStrategyAbstract(min_genes_before_merge, min_genes_after_merge), _max_ed(max_merge_edit_distance) {}
	std::string merge_type() const override { return "Merge all"; }
	void fill(dropest_cfg &cfg) const override { cfg.merge_kind = DROPEST_MERGE_ALL; cfg.max_cb_merge_edit_distance = int(_max_ed); cfg.min_merge_fraction = 0; }
};
class RealBarcodesMergeStrategy : public MergeStrategyAbstract {
	std::string _file; int _kind; unsigned _max_ed; double _min_fraction;
public:
	enum BarcodesType { INDROP = DROPEST_BARCODES_INDROP, CONST_LENGTH = DROPEST_BARCODES_CONST };
	RealBarcodesMergeStrategy(BarcodesType type, const std::string &barcodes_filename, size_t min_genes_before_merge,
	                          size_t min_genes_after_merge, unsigned max_merge_edit_distance, double min_merge_fraction)
		: MergeStrategyAbstract(min_genes_before_merge, min_genes_after_merge), _file(barcodes_filename), _kind(type),
		  _max_ed(max_merge_edit_distance), _min_fraction(min_merge_fraction) {}
	std::string merge_type() const override { return "Real CBs"; }
	void fill(dropest_cfg &cfg) const override {
		cfg.merge_kind = DROPEST_MERGE_REAL_BARCODES; cfg.barcodes_kind = _kind; cfg.barcodes_file = _file.c_str();
		cfg.max_cb_merge_edit_distance = int(_max_ed); cfg.min_merge_fraction = _min_fraction;
	}
};
// Estimation/Merge/PoissonTargetEstimator.h:57 + PoissonRealBarcodesMergeStrategy.h (-M with a barcodes file)
struct PoissonTargetEstimator {
	struct EstimationResult {                                  // PoissonTargetEstimator.h:21-33
		size_t intersection_size; double expected_intersection_size, merge_probability;
	};
	double max_merge_prob, max_real_cb_merge_prob;
	PoissonTargetEstimator(double max_merge_prob_, double max_real_cb_merge_prob_)
		: max_merge_prob(max_merge_prob_), max_real_cb_merge_prob(max_real_cb_merge_prob_) {}
	// PoissonTargetEstimator.cpp:67-94; the UMI distribution is taken from the container's filtered cells (init, :46-60)
	EstimationResult estimate_intersection_prob(const CellsDataContainer &container, size_t cell1_ind, size_t cell2_ind) const;
};
class PoissonRealBarcodesMergeStrategy : public MergeStrategyAbstract {
	PoissonTargetEstimator _estimator; std::string _file; int _kind; unsigned _max_ed;
public:
	PoissonRealBarcodesMergeStrategy(const PoissonTargetEstimator &target_estimator, RealBarcodesMergeStrategy::BarcodesType type,
	                                 const std::string &barcodes_filename, size_t min_genes_before_merge, size_t min_genes_after_merge,
	                                 unsigned max_merge_edit_distance)
		: MergeStrategyAbstract(min_genes_before_merge, min_genes_after_merge), _estimator(target_estimator), _file(barcodes_filename),
		  _kind(type), _max_ed(max_merge_edit_distance) {}
	std::string merge_type() const override { return "Poisson Real CBs"; }
	void fill(dropest_cfg &cfg) const override {
		cfg.merge_kind = DROPEST_MERGE_POISSON_REAL; cfg.barcodes_kind = _kind; cfg.barcodes_file = _file.c_str();
		cfg.max_cb_merge_edit_distance = int(_max_ed); cfg.min_merge_fraction = 0;   // PoissonRealBarcodesMergeStrategy.cpp:15-16
		cfg.max_merge_prob = _estimator.max_merge_prob; cfg.max_real_merge_prob = _estimator.max_real_cb_merge_prob;
	}
};
// Estimation/Merge/PoissonSimpleMergeStrategy.h (-M without a barcodes file)
class PoissonSimpleMergeStrategy : public MergeStrategyAbstract {
	PoissonTargetEstimator _estimator; unsigned _max_ed;
public:
	PoissonSimpleMergeStrategy(const PoissonTargetEstimator &target_estimator, unsigned min_genes_before_merge, unsigned min_genes_after_merge,
	                           unsigned max_merge_edit_distance)
		: MergeStrategyAbstract(min_genes_before_merge, min_genes_after_merge), _estimator(target_estimator), _max_ed(max_merge_edit_distance) {}
	std::string merge_type() const override { return "Poisson Simple"; }
	void fill(dropest_cfg &cfg) const override {
		cfg.merge_kind = DROPEST_MERGE_POISSON_SIMPLE; cfg.max_cb_merge_edit_distance = int(_max_ed); cfg.min_merge_fraction = 0;
		cfg.max_merge_prob = _estimator.max_merge_prob; cfg.max_real_merge_prob = _estimator.max_real_cb_merge_prob;
	}
};
namespace UMIs {
class MergeUMIsStrategyAbstract {
public:
	virtual ~MergeUMIsStrategyAbstract() = default;
	virtual void fill(dropest_cfg &cfg) const = 0;
};
class MergeUMIsStrategySimple : public MergeUMIsStrategyAbstract {
	unsigned _max_merge_distance;
public:
	explicit MergeUMIsStrategySimple(unsigned max_merge_distance) : _max_merge_distance(max_merge_distance) {}
	void fill(dropest_cfg &cfg) const override { cfg.umi_merge_kind = DROPEST_UMI_MERGE_SIMPLE; cfg.max_umi_merge_edit_distance = int(_max_merge_distance); }
};

// Estimation/Merge/UMIs/MergeUMIsStrategyDirectional.h:44 (-u)
class MergeUMIsStrategyDirectional : public MergeUMIsStrategyAbstract {
	double _mult; unsigned _max_edit_distance;
public:
	explicit MergeUMIsStrategyDirectional(double mult = 2, unsigned max_edit_distance = 1) : _mult(mult), _max_edit_distance(max_edit_distance) {}
	void fill(dropest_cfg &cfg) const override {
		cfg.umi_merge_kind = DROPEST_UMI_MERGE_DIRECTIONAL; cfg.umi_merge_multiplier = _mult;
		cfg.max_umi_merge_edit_distance = int(_max_edit_distance);
	}
};
}  // namespace UMIs
}  // namespace Merge

class CellsDataContainer;

// What Cell / Gene / UMI expose to ResultsPrinter and the tests: one cell's row, read again whenever the (unsharded) container has changed
// since -- a `Cell &` from CellsDataContainer::cell(index) shows later merges and mutations like the reference's.
class Cell {
	friend class CellsDataContainer;
	const CellsDataContainer *_owner = nullptr;
	mutable dropest_ctx *_ctx = nullptr;      // the context the cell lives in (a sharded container: the owner shard's)
	size_t _id = 0;
	mutable dropest_cell_row _row{};
	std::string _barcode;
	mutable uint64_t _gen = 0;                // the container's generation the row was read at (0: not tracked)
	void sync() const;
public:
	struct MoleculeRow {
		std::string gene, umi; size_t read_count; UMI::Mark mark;
		std::vector<unsigned> sum_quality;                                 // UMI::_sum_quality (UMI.h:51); empty without qualities
		std::vector<double> mean_quality() const {                         // UMI.cpp:46-55: unsigned integer arithmetic, as written
			std::vector<double> res(sum_quality.size());
			for (size_t i = 0; i < res.size(); ++i) res[i] = double((sum_quality[i] - unsigned(Tools::ReadParameters::quality_offset)) / read_count);
			return res;
		}
	};
	uint64_t barcode_code() const { return _row.barcode; }   // (a cell's barcode never changes)
	size_t id() const { return _id; }                        // index in the container (first-seen rank of the barcode)
	bool is_merged() const { sync(); return _row.is_merged; }
	bool is_excluded() const { sync(); return _row.is_excluded; }
	bool is_real() const { sync(); return _row.is_real; }
	std::string barcode() const { return _barcode; }
	size_t umis_number() const { sync(); return size_t(_row.total_umis); }
	size_t requested_genes_num() const { sync(); return _row.requested_genes; }
	size_t requested_umis_num() const { sync(); return _row.requested_umis; }
	size_t size() const { sync(); return _row.n_genes; }
	int stat(Stats::CellStatType t) const { sync(); return t == Stats::TOTAL_READS_PER_CB ? _row.total_reads : _row.total_umis; }
	std::vector<MoleculeRow> molecules() const;                                  // walk of genes() x umis()
	std::unordered_map<std::string, size_t> requested_umis_per_gene(const UMI::Mark::query_t &query, bool return_reads) const;
};

class CellsDataContainer {
public:
	using s_ul_hash_t = std::unordered_map<std::string, size_t>;
	using s_i_hash_t = std::unordered_map<std::string, int>;
	using ids_t = std::vector<size_t>;
	using counts_t = std::vector<int>;
	using names_t = std::vector<std::string>;

private:
	std::shared_ptr<Merge::MergeStrategyAbstract> _merge_strategy;
	std::shared_ptr<Merge::UMIs::MergeUMIsStrategyAbstract> _umi_merge_strategy;
	UMI::Mark::query_t _query_marks;
	dropest_ctx *_ctx = nullptr;
	// N GPUs behind the one container: every shard takes ONE contiguous range of the stream (shard_quota reads, the last shard
	// the rest; expect_reads() sizes the quota so that the ranges come out even); merge_and_filter runs the sharded pass
	// (include/dropest_amd.h: dropest_shard_*), one host thread per GPU
	std::vector<dropest_shard *> _shards;
	uint64_t _batches = 0;
	size_t _side_sent = 0;
	int _split_parts = 0;     // > 0: one device, the stream split over this many shards because the sort key passed 64 bits
	void split_for_wide_keys();
	[[noreturn]] void single_only(const char *what) const;
	bool _is_initialized = false;
	// host-side dictionaries (strings never reach the device)
	StringIndexer _gene_indexer, _chr_indexer;
	std::vector<std::string> _side;
	std::unordered_map<std::string, uint64_t> _side_cb, _side_umi;
	// pending batch
	std::vector<uint64_t> _cb, _umi;
	std::vector<uint32_t> _gene, _aux;
	size_t _umi_quality_length = size_t(-1);  // row width of _qual: the longest UMI quality string so far (-1: no gene-bearing read yet)
	std::vector<uint8_t> _qual;               // qualities of every read so far, _umi_quality_length bytes each
	size_t _qual_pending = 0;                 // gene-less reads seen before the length was known
	size_t _qual_reads = 0;                   // reads that went through append_quality
	std::vector<size_t> _shard_reads;         // sharded container: reads dealt to every shard so far (ranges of the stream, in shard order)
	std::vector<uint8_t> _qual_lens;          // per read, once two gene-bearing reads differed in length (UMI.cpp:26-28 is a per-molecule check)
	void note_quality_length(size_t ql);
	// UMI::add_read's "Wrong quality length" (UMI.cpp:26-28) from the add_record that meets it: once two gene-bearing reads have differed
	// in length, the length of every molecule is kept here (the molecules before that moment come from the preview: they all have the
	// first length)
	struct MolKey { uint64_t cb, umi; uint32_t gene; bool operator==(const MolKey &o) const { return cb == o.cb && umi == o.umi && gene == o.gene; } };
	struct MolKeyHash { size_t operator()(const MolKey &k) const { uint64_t h = k.cb * 0x9E3779B97F4A7C15ull ^ (k.umi + 0x632BE59BD9B4E019ull) * 0xC2B2AE3D27D4EB4Full ^ uint64_t(k.gene) * 0x165667B19E3779F9ull; return size_t(h ^ (h >> 29)); } };
	std::unordered_map<MolKey, uint8_t, MolKeyHash> _mol_qlen;
	bool _mol_qlen_tracking = false;
	void check_molecule_quality_length(uint64_t cb_code, uint32_t gene, uint64_t umi_code, size_t ql);
	void append_quality(const char *q, size_t len, bool has_gene);
	std::vector<std::string> _ref_names;                      // ParsedRead::ref_id -> chromosome name
	std::vector<int32_t> _ref_chr;                            // ... -> index in _chr_indexer, -1 = not met yet
	std::unordered_map<uint64_t, uint32_t> _gene_by_hash;     // name hash -> gene index (verified against the name)
	mutable ids_t _filtered_cache, _merge_targets_cache;
	// Before set_initialized the reference's container already answers accessors and takes mutators (its cells exist from the
	// first add_record).  Here the device tables exist only once the container is initialised, so such calls look at a PREVIEW:
	// a second context over the reads pushed so far (adopted in place), initialised on demand, with the mutators called so far
	// replayed on it; the same mutators are replayed on the real context when set_initialized() runs.
	struct PendingMutation { int kind; size_t a = 0, b = 0; std::unordered_map<std::string, std::string> targets; std::string umi, gene, quality; uint8_t mark = 0; };
	std::vector<PendingMutation> _pending;
	mutable dropest_ctx *_preview = nullptr;
	mutable bool _preview_valid = false;
	int _device = 0;
	int _max_cells_num = -1;
	void fill_cfg(dropest_cfg &cfg, std::string &levels) const;
	dropest_ctx *view() const;                                 // the context accessors read: the real one, or the preview
	void apply_mutation(dropest_ctx *ctx, const PendingMutation &m);
	void send_side_strings(dropest_ctx *ctx) const;
	mutable StringIndexer _umi_indexer_cache;                 // umi_indexer(): built on demand from dropest_umi_first_seen
	mutable bool _umi_indexer_valid = false;
	mutable std::unordered_map<size_t, Cell> _cell_cache;     // Cell &cell(index): container-owned, live (Cell::sync)
	mutable uint64_t _generation = 1;                         // bumped by whatever changes what a Cell shows
	friend class Cell;

	uint64_t encode(const std::string &s, std::unordered_map<std::string, uint64_t> &escapes);
	void flush();
	[[noreturn]] void fail(dropest_status st) const;
	void check(dropest_status st) const { if (st != DROPEST_OK) fail(st); }

public:
	static const size_t BATCH = size_t(1) << 20;
	size_t shard_quota = size_t(1) << 27;   // sharded container: reads (a multiple of BATCH) a shard takes before the next one starts
	// A caller that knows roughly how many reads will come (a BAM's size, a previous run) says so before the first add_record: the
	// quota becomes ceil(expected / shards) rounded up to whole batches, so that every GPU and every PCIe link carries its share
	// (with the default quota a stream shorter than 2^27 reads lands on shard 0 alone and a very long one piles onto the last).
	// An under-estimate only makes the last shard longer; results do not depend on where reads waited.
	void expect_reads(size_t expected) {
		if (!sharded()) {      // one context: its read store at that size now, instead of doubling (and copying itself) on the way there -- 10 ms per step on a BAM of 8 M reads
			if (_ctx && !_is_initialized && expected) (void)dropest_reserve_reads(_ctx, expected);
			return;
		}
		if (_batches) return;
		const size_t per = (expected + _shards.size() - 1) / _shards.size();
		shard_quota = std::max<size_t>(BATCH, (per + BATCH - 1) / BATCH * BATCH);
	}

	CellsDataContainer(const std::shared_ptr<Merge::MergeStrategyAbstract> &merge_strategy,
	                   const std::shared_ptr<Merge::UMIs::MergeUMIsStrategyAbstract> &umi_merge_strategy,
	                   const std::vector<UMI::Mark> &gene_match_levels, bool save_umi_merge_targets = false,
	                   int max_cells_num = -1, int device = 0);
	// The same container over several GPUs (the path shards by cell barcode): `devices` = one HIP ordinal per shard.  add_record,
	// set_initialized, merge_and_filter and ResultsPrinter::save_results (the whole R list: both matrices, per-chromosome frames,
	// saturation info, per-cell counts, reads_per_umi_per_cell) / get_count_matrix / save_mtx work as on one GPU -- results are
	// those of ONE container over the whole stream: real_cells() lists the real cells in that container's cell-id order, each
	// answering from the shard that owns it --; accessors BY CELL ID (cell(i), merge_targets(), filtered_cells(), ...) and the
	// strategies other than Dummy / RealBarcodes + the default UMI merge need the single-GPU container and throw
	// std::runtime_error here.
	CellsDataContainer(const std::shared_ptr<Merge::MergeStrategyAbstract> &merge_strategy,
	                   const std::shared_ptr<Merge::UMIs::MergeUMIsStrategyAbstract> &umi_merge_strategy,
	                   const std::vector<UMI::Mark> &gene_match_levels, bool save_umi_merge_targets, int max_cells_num,
	                   const std::vector<int> &devices);
	// A container on ONE device whose sort key (cell id + gene + UMI fields) passes 64 bits turns itself into such a sharded
	// container at set_initialized(): the same reads as 2^k shards on that device (dropest_ctx_split), same restrictions.
	bool sharded() const { return !_shards.empty(); }
	dropest_shard *shard0() const { return _shards.empty() ? nullptr : _shards[0]; }
	// (source, target) barcodes of the cells the CB merge folded: merge_targets() by barcode, also on a sharded container
	std::vector<std::pair<std::string, std::string>> merged_barcodes() const;
	~CellsDataContainer();
	CellsDataContainer(const CellsDataContainer &) = delete;              // pointer-stable, like the reference needs to be
	CellsDataContainer &operator=(const CellsDataContainer &) = delete;

	void add_record(const ReadInfo &read_info);
	// add_record for callers that parsed the read themselves (the BAM reader's worker threads): barcode / UMI arrive as
	// 2-bit codes when they are packable (0 = use the text), the gene name with its hash, the chromosome as a slot of
	// the caller's reference table.  Same effect as add_record(ReadInfo(...)) read by read.
	struct ParsedRead {
		uint64_t cb_code = 0, umi_code = 0;
		std::string_view cb, umi, gene;
		uint64_t gene_hash = 0;
		int64_t gene_id = -1;                 // from lookup_gene (a parser thread may resolve known genes ahead), -1 = unknown
		int32_t ref_id = 0;                   // index into the chromosome names given to set_reference_names
		uint8_t mark = 0;
		uint32_t umi_quality_length = 0;
		std::string_view umi_quality;         // umi_quality_length characters (phred+33)
	};
	void set_reference_names(const std::vector<std::string> &names);
	void add_record(const ParsedRead &read);
	// index of a gene that is ALREADY in the dictionary, -1 otherwise; safe to call from several threads as long as no
	// add_record runs at the same time
	int64_t lookup_gene(uint64_t gene_hash, std::string_view name) const;
	// by the hash alone: the index of the FIRST name with this FNV-1a value, -1 = unseen.  (The device BAM path confirms a hit against the
	// name's bytes since round 6 -- dropest_bam_decoder_set_gene_names -- so two names with one hash do not share an index there either.)
	int64_t lookup_gene_hash(uint64_t gene_hash) const { auto it = _gene_by_hash.find(gene_hash); return it == _gene_by_hash.end() ? -1 : int64_t(it->second); }
	int device() const { return _device; }
	// the gene dictionary as (hash, index) pairs and the chromosome index of every reference (-1 = none yet): what the device BAM path looks records up in
	void dictionary_snapshot(std::vector<uint64_t> &gene_hash, std::vector<uint32_t> &gene_id, std::vector<int32_t> &chr_of_ref) const {
		gene_hash.clear(); gene_id.clear();
		for (auto const &kv : _gene_by_hash) { gene_hash.push_back(kv.first); gene_id.push_back(kv.second); }
		chr_of_ref.assign(_ref_chr.begin(), _ref_chr.end());
	}
	// add_records_packed for columns that already live in the container's GPU memory (include/dropest_bgzf.h); any_gene: some read carries a gene
	void add_records_packed_device(const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene, const uint32_t *d_aux, size_t n, bool any_gene);
	// ... with one row of ql quality bytes per read in HOST memory (bulk_ingest_possible_with_quality(ql); every gene-bearing read's string is ql long)
	void add_records_packed_device(const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene, const uint32_t *d_aux, size_t n, bool any_gene, const uint8_t *quality_rows, size_t ql);
	// ---- bulk ingest (the BAM reader's fast path) ----------------------------------------------------------------------
	// add_record read by read costs a handful of vector appends and dictionary look-ups per read on ONE thread; a caller that
	// parses records on many threads resolves the dictionaries itself -- the few reads per window that bring something new
	// (an unseen gene name or chromosome, a barcode / UMI with N), in stream order, through the intern_* members -- and hands
	// whole arrays of packed records over.  Same effect as add_record(ParsedRead) read by read, in the same order.
	bool bulk_ingest_possible() const { return !sharded() && !_is_initialized && (_umi_quality_length == size_t(-1) || _umi_quality_length == 0); }
	uint64_t intern_barcode(const std::string &s) { return encode(s, _side_cb); }
	uint64_t intern_umi(const std::string &s) { return encode(s, _side_umi); }
	uint32_t intern_gene(std::string_view name, uint64_t hash);
	int32_t chromosome_of_ref(int32_t ref_id) const { return _ref_chr[size_t(ref_id)]; }        // -1: no read touched it yet
	uint32_t intern_chromosome_of_ref(int32_t ref_id);
	// n records, every id resolved (gene: dictionary index or DROPEST_NO_GENE; aux = chromosome index | mark << 16); the reads
	// carry no UMI qualities (quality length 0, like add_record of a read with an empty quality string)
	void add_records_packed(const uint64_t *cb, const uint64_t *umi, const uint32_t *gene, const uint32_t *aux, size_t n);
	// ... several such runs that follow one another in the stream, handed over in one piece (dropest_push_reads_gather)
	struct PackedRun { const uint64_t *cb, *umi; const uint32_t *gene, *aux; size_t n; };
	void add_records_packed(const std::vector<PackedRun> &runs);
	// The same for reads with UMI quality strings of ONE length: quality[k] = counts-of-run-k rows of `ql` bytes (the row of a read without a gene is
	// not looked at).  Allowed while every gene-bearing read so far had that length (bulk_ingest_possible_with_quality): what UMI::add_read's length
	// check (UMI.cpp:26-28) needs beyond that goes through add_record.
	bool bulk_ingest_possible_with_quality(size_t ql) const {
		return !sharded() && !_is_initialized && !_mol_qlen_tracking && _qual_lens.empty() && ql > 0 && ql <= 255 && (_umi_quality_length == size_t(-1) || _umi_quality_length == ql);
	}
	void add_records_packed(const std::vector<PackedRun> &runs, const std::vector<const uint8_t *> &quality, size_t ql);
	void reserve_quality_rows(size_t reads, size_t ql) { if (reads && ql && _qual.capacity() < reads * ql) _qual.reserve(reads * ql); }   // (a reader that knows how long the stream will be)
	bool bulk_ingest_possible_at_all() const { return !sharded() && !_is_initialized && !_mol_qlen_tracking && _qual_lens.empty(); }   // (with or without quality rows: the window decides)
	static bool pack_code(std::string_view s, uint64_t &code);
	static uint64_t hash_name(std::string_view s);
	void set_initialized();
	void merge_and_filter();

	size_t total_cells_number() const;
	size_t cell_id_by_cb(const std::string &barcode) const;             // throws std::out_of_range
	const ids_t &filtered_cells() const;
	const ids_t &merge_targets() const;
	const UMI::Mark::query_t &gene_match_level() const { return _query_marks; }
	// PoissonTargetEstimator::estimate_intersection_prob's numbers for two cells (PoissonTargetEstimator.cpp:67-94)
	void poisson_intersection(size_t cell1_ind, size_t cell2_ind, size_t &intersection, double &expected, double &probability) const;
	long get_merge_target(size_t base_cell_ind) const;
	// the public mutators (CellsDataContainer.h:88-95); here they act on the initialised container
	using s_s_hash_t = std::unordered_map<std::string, std::string>;
	void exclude_cell(size_t index);
	void merge_cells(size_t source_cell_ind, size_t target_cell_ind);
	void merge_umis(size_t cell_id, size_t gene, const s_s_hash_t &merge_targets);                  // RealBarcodesMergeStrategy::get_merge_target
	// CellsDataContainer.h:90 (CellsDataContainer.cpp:356-364): one more read of read_info's UMI for read_info's gene in cell
	// `cell_id` -- a new molecule (TOTAL_UMIS_PER_CB + 1) or read count + 1 / mark OR; no read counters, no chromosome statistics.
	// Cell ids exist once the container is initialised (they are assigned on the device): std::runtime_error before that.
	void add_umi_to_cell(size_t cell_id, const ReadInfo &read_info);

	s_i_hash_t get_stat_by_real_cells(Stats::CellStatType type) const;
	void get_stat_by_real_cells(Stats::CellChrStatType stat, names_t &cell_barcodes, names_t &chromosome_names,
	                            counts_t &counts) const;
	Cell cell(size_t index) const;                                      // throws std::out_of_range
	// CellsDataContainer.h:107: a reference to a container-owned Cell that stays current (it reads its row again after the container
	// changed; what Cell exposes here is read-only: changes go through the container's mutators above)
	Cell &cell(size_t index);
	// every real cell in cell-id order -- also on a sharded container, where that is the order of first appearance in the WHOLE
	// stream and every Cell answers from the shard that owns its barcode -- and the positions of the filtered cells in that list
	// (in filtered_cells() order).  ResultsPrinter::results_list is written on these two.
	std::vector<Cell> real_cells() const;
	std::vector<size_t> filtered_positions(const std::vector<Cell> &real) const;

	size_t intergenic_reads_num() const;
	size_t has_exon_reads_num() const;
	size_t has_intron_reads_num() const;
	size_t has_not_annotated_reads_num() const;
	size_t real_cells_number() const;
	std::string merge_type() const { return _merge_strategy->merge_type(); }
	const StringIndexer &gene_indexer() const { return _gene_indexer; }
	// CellsDataContainer.h:118: UMIs in index order (first appearance among the gene-bearing reads, then the UMIs merges
	// brought in); materialised on the first call after the container last changed
	const StringIndexer &umi_indexer() const;
	s_ul_hash_t umi_distribution() const;                               // CellsDataContainer.cpp:182-197
	const std::vector<std::string> &side_strings() const { return _side; }
	dropest_ctx *handle() const { return _ctx; }
	std::string decode(uint64_t code) const;
};

// Count-matrix assembly and the MatrixMarket writer of ResultsPrinter (Estimation/ResultsPrinter.cpp:334-396, :81-91).
class ResultsPrinter {
	const bool write_matrix, reads_output, umi_correction_info;
public:
	struct SparseMatrix {            // dgCMatrix layout (CSC), rows = genes, columns = cells
		std::vector<std::string> row_names, col_names;
		std::vector<uint32_t> colptr, rowidx, values;
	};
	ResultsPrinter(bool write_matrix_, bool reads_output_, bool /*validation_stats*/ = false, bool umi_correction_info_ = false)
		: write_matrix(write_matrix_), reads_output(reads_output_), umi_correction_info(umi_correction_info_) {}
	// get_count_matrix / save_mtx / save_results name the matrix straight from the byte form that crossed PCIe (two bytes per entry, decoded
	// column by column on the calling thread) instead of from the 32-bit slots; false takes the slots (dropest_count_matrix_csc)
	bool walk_byte_form = true;
	// reference_row_order = true reproduces the row order of the reference's dgCMatrix (rows numbered on first
	// encounter while iterating an unordered_map per cell, Cell.cpp:54-68 + ResultsPrinter.cpp:345-355); false keeps
	// rows in gene-index order.
	SparseMatrix get_count_matrix(const CellsDataContainer &container, bool filtered, bool reference_row_order = true) const;
	// get_count_matrix_filtered(container, query_marks) for an explicit query (ResultsPrinter.cpp:333-361)
	SparseMatrix get_count_matrix_filtered(const CellsDataContainer &container, const UMI::Mark::query_t &query_marks,
	                                       bool reference_row_order = true) const;
	// <base>.matrices.rds: list(exon, intron, spanning) of dgCMatrix (-V, ResultsPrinter.cpp:455-474)
	void save_intron_exon_matrices(const CellsDataContainer &container, const std::string &filename) const;
private:
	// where a column's (gene id, count) entries come from: the 32-bit arrays, or the byte form walked directly (dropest_matrix_bytes)
	struct ColumnSource {
		uint64_t ncols = 0, nnz = 0;
		const uint32_t *colptr = nullptr, *rowidx = nullptr, *values = nullptr;
		const dropest_matrix_bytes *bytes = nullptr;
		std::vector<std::pair<uint32_t, uint32_t>> row_listed, value_listed;   // (entry position, exact row / value), ascending
		ColumnSource(uint64_t ncols_, uint64_t nnz_, const uint32_t *cp, const uint32_t *ri, const uint32_t *v) : ncols(ncols_), nnz(nnz_), colptr(cp), rowidx(ri), values(v) {}
		explicit ColumnSource(const dropest_matrix_bytes &B);
		void column(uint64_t col, std::vector<std::pair<uint32_t, uint32_t>> &out) const;
	};
	SparseMatrix named_matrix(const CellsDataContainer &c, bool filtered, bool reference_row_order, uint64_t ncols, uint64_t nnz,
	                          const uint32_t *colptr, const uint32_t *rowidx, const uint32_t *values,
	                          const std::vector<std::string> *col_names = nullptr) const;
	SparseMatrix named_matrix(const CellsDataContainer &c, bool filtered, bool reference_row_order, const ColumnSource &src,
	                          const std::vector<std::string> *col_names = nullptr) const;
	SparseMatrix sharded_matrix(const CellsDataContainer &c, bool filtered, bool reference_row_order) const;
public:
	// <base>.mtx + <base>.cells.tsv + <base>.genes.tsv (what save_mtx writes through R's Matrix::writeMM)
	void save_mtx(const CellsDataContainer &container, const std::string &filename_base) const;
	// <base>.rds: the R list d = list(cm, cm_raw, reads_per_chr_per_cells, mean_reads_per_umi, saturation_info, merge_targets,
	// aligned_reads_per_cell, aligned_umis_per_cell, requested_umis_per_cb, requested_reads_per_cb[, reads_per_umi_per_cell]),
	// written natively in R's serialisation format (rds_writer.h); + the matrix triple with write_matrix
	void save_results(const CellsDataContainer &container, const std::string &filename) const;
	std::shared_ptr<Rds::Value> results_list(const CellsDataContainer &container) const;
};

}  // namespace Estimation
