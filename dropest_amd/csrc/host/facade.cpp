// facade.cpp -- see facade.h.  Host-side packing and result relaying only; no counting happens here.
#include "facade.h"
#include "rds_writer.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <fstream>

namespace Tools {
ReadParameters ReadParameters::parse_encoded_id(const std::string &encoded_id) {   // Tools/ReadParameters.cpp:42-56
	const size_t umi_pos = encoded_id.rfind('#');
	if (umi_pos == std::string::npos) throw std::runtime_error("ERROR: unable to parse out UMI in: " + encoded_id);
	const size_t cb_pos = encoded_id.rfind('!', umi_pos);
	if (cb_pos == std::string::npos) throw std::runtime_error("ERROR: unable to parse out cell barcode in: " + encoded_id);
	return ReadParameters(encoded_id.substr(cb_pos + 1, umi_pos - cb_pos - 1), encoded_id.substr(umi_pos + 1));
}
void CollisionsAdjuster::init(const probs_vec_t &umi_probabilities, size_t max_gene_expression) {   // CollisionsAdjuster.cpp:12-19
	_umi_probabilities = umi_probabilities;
	_adjusted_sizes.clear();
	update_adjusted_sizes(max_gene_expression);
}
void CollisionsAdjuster::update_adjusted_sizes(size_t max_gene_expression) {
	if (max_gene_expression <= _adjusted_sizes.size()) return;
	// the recurrence restarts from s = 1 on the device (deterministic: same prefix every time)
	std::vector<uint64_t> t(max_gene_expression);
	if (dropest_collisions_adjusted_sizes(_device, _umi_probabilities.data(), _umi_probabilities.size(), max_gene_expression, t.data()) != DROPEST_OK)
		throw std::runtime_error(dropest_last_error());
	_adjusted_sizes.assign(t.begin(), t.end());
}
size_t CollisionsAdjuster::estimate_adjusted_gene_expression(size_t expression) {   // :41-49
	if (expression > _adjusted_sizes.size()) update_adjusted_sizes(expression);
	return _adjusted_sizes.at(expression - 1);
}
}  // namespace Tools

namespace Estimation {

const std::string UMI::Mark::DEFAULT_CODE = "eEBA";

UMI::Mark UMI::Mark::get_by_code(char code) {   // UMI.cpp:123-154
	Mark m;
	switch (code) {
		case 'e': m.add(HAS_EXONS); break;
		case 'i': m.add(HAS_INTRONS); break;
		case 'E': m.add(HAS_EXONS); m.add(HAS_NOT_ANNOTATED); break;
		case 'I': m.add(HAS_INTRONS); m.add(HAS_NOT_ANNOTATED); break;
		case 'B': m.add(HAS_EXONS); m.add(HAS_INTRONS); break;
		case 'A': m.add(HAS_EXONS); m.add(HAS_INTRONS); m.add(HAS_NOT_ANNOTATED); break;
		default: throw std::runtime_error(std::string("Unexpected gene match levels: ") + code);
	}
	return m;
}
UMI::Mark::query_t UMI::Mark::get_by_code(const std::string &code) {
	query_t q;
	for (char c : code) q.push_back(get_by_code(c));
	return q;
}
std::string UMI::Mark::to_code(const query_t &levels) {
	std::string s;
	for (auto const &l : levels) {
		switch (l.bits()) {
			case 2: s += 'e'; break; case 4: s += 'i'; break; case 3: s += 'E'; break; case 5: s += 'I'; break;
			case 6: s += 'B'; break; case 7: s += 'A'; break;
			default: throw std::runtime_error("Unexpected gene match level");
		}
	}
	return s;
}

// ---- 2-bit codes ----
// A table look-up per base and no branch on its value: the bases of a barcode are random, and a four-way switch per character cost a
// misprediction every other base -- 235 ns for a 16 + 10 base pair of barcode and UMI against 22 ns this way, which was the larger
// half of what a BAM record costs to parse.
struct BaseTable { uint8_t v[256]; constexpr BaseTable() : v() { for (int i = 0; i < 256; ++i) v[i] = 0x80; v[int('A')] = 0; v[int('C')] = 1; v[int('G')] = 2; v[int('T')] = 3; } };
static constexpr BaseTable BASES{};
static inline bool pack_bases(const char *s, size_t n, uint64_t &code) {
	if (n == 0 || n > 31) return false;
	uint64_t c = 1;
	unsigned bad = 0;
	for (size_t i = 0; i < n; ++i) { const unsigned b = BASES.v[uint8_t(s[i])]; bad |= b; c = (c << 2) | (b & 3u); }
	if (bad & 0x80u) return false;
	code = c;
	return true;
}
static bool pack2(const std::string &s, uint64_t &code) { return pack_bases(s.data(), s.size(), code); }
std::string CellsDataContainer::decode(uint64_t code) const {
	if (code & DROPEST_ESCAPE) return _side.at(size_t(code & ~DROPEST_ESCAPE));
	if (!code) return std::string();
	const int len = (63 - __builtin_clzll(code)) / 2;
	std::string s(size_t(len), 'A');
	for (int i = 0; i < len; ++i) s[size_t(i)] = "ACGT"[(code >> (2 * (len - 1 - i))) & 3];
	return s;
}
uint64_t CellsDataContainer::encode(const std::string &s, std::unordered_map<std::string, uint64_t> &escapes) {
	uint64_t code;
	if (pack2(s, code)) return code;
	auto it = escapes.find(s);
	if (it != escapes.end()) return it->second;
	code = DROPEST_ESCAPE | uint64_t(_side.size());
	_side.push_back(s);
	escapes.emplace(s, code);
	return code;
}

void CellsDataContainer::fail(dropest_status st) const {
	const std::string msg = dropest_last_error();
	if (st == DROPEST_ERR_RANGE) throw std::out_of_range(msg);
	throw std::runtime_error(msg);
}

void CellsDataContainer::fill_cfg(dropest_cfg &cfg, std::string &levels) const {
	dropest_cfg_defaults(&cfg);
	cfg.device = _device;
	_merge_strategy->fill(cfg);
	_umi_merge_strategy->fill(cfg);
	cfg.min_genes_before_merge = int(_merge_strategy->min_genes_before_merge());
	cfg.min_genes_after_merge = int(_merge_strategy->min_genes_after_merge());
	levels = UMI::Mark::to_code(_query_marks);
	cfg.gene_match_levels = levels.c_str();
	cfg.max_cells = _max_cells_num;
}

CellsDataContainer::CellsDataContainer(const std::shared_ptr<Merge::MergeStrategyAbstract> &merge_strategy,
                                       const std::shared_ptr<Merge::UMIs::MergeUMIsStrategyAbstract> &umi_merge_strategy,
                                       const std::vector<UMI::Mark> &gene_match_levels, bool /*save_umi_merge_targets*/,
                                       int max_cells_num, int device)
	: _merge_strategy(merge_strategy), _umi_merge_strategy(umi_merge_strategy), _query_marks(gene_match_levels) {
	_device = device; _max_cells_num = max_cells_num;
	dropest_cfg cfg; std::string levels;
	fill_cfg(cfg, levels);
	check(dropest_ctx_create(&cfg, &_ctx));
}

CellsDataContainer::CellsDataContainer(const std::shared_ptr<Merge::MergeStrategyAbstract> &merge_strategy,
                                       const std::shared_ptr<Merge::UMIs::MergeUMIsStrategyAbstract> &umi_merge_strategy,
                                       const std::vector<UMI::Mark> &gene_match_levels, bool /*save_umi_merge_targets*/,
                                       int max_cells_num, const std::vector<int> &devices)
	: _merge_strategy(merge_strategy), _umi_merge_strategy(umi_merge_strategy), _query_marks(gene_match_levels) {
	if (devices.empty()) throw std::runtime_error("no device given");
	_device = devices[0]; _max_cells_num = max_cells_num;
	dropest_cfg cfg; std::string levels;
	fill_cfg(cfg, levels);
	if (devices.size() == 1) { check(dropest_ctx_create(&cfg, &_ctx)); return; }
	std::vector<int32_t> dev(devices.begin(), devices.end());
	_shards.assign(devices.size(), nullptr);
	check(dropest_shard_group_create(&cfg, int32_t(dev.size()), dev.data(), _shards.data()));
}

void CellsDataContainer::send_side_strings(dropest_ctx *ctx) const {
	std::vector<const char *> ptrs(_side.size());
	for (size_t i = 0; i < _side.size(); ++i) ptrs[i] = _side[i].c_str();
	check(dropest_set_side_strings(ctx, ptrs.data(), ptrs.size()));
}

// The context accessors read.  Initialised container: its own.  Before that: the preview (see facade.h).
dropest_ctx *CellsDataContainer::view() const {
	if (_is_initialized) return _ctx;
	if (sharded()) single_only("accessors before set_initialized");
	if (_preview_valid) return _preview;
	CellsDataContainer *self = const_cast<CellsDataContainer *>(this);   // flushing the pending batch changes nothing observable
	self->flush();
	if (_preview) { dropest_ctx_destroy(_preview); _preview = nullptr; }
	dropest_cfg cfg; std::string levels;
	fill_cfg(cfg, levels);
	check(dropest_ctx_create(&cfg, &_preview));
	send_side_strings(_preview);
	const uint64_t *cb = nullptr, *umi = nullptr; const uint32_t *gene = nullptr, *aux = nullptr; uint64_t n = 0;
	check(dropest_resident_reads(_ctx, &cb, &umi, &gene, &aux, &n));
	if (n) check(dropest_push_reads_device(_preview, cb, umi, gene, aux, n, 1));
	check(dropest_set_initialized(_preview));
	for (const PendingMutation &m : _pending) self->apply_mutation(_preview, m);
	_preview_valid = true;
	return _preview;
}

void CellsDataContainer::apply_mutation(dropest_ctx *ctx, const PendingMutation &m) {
	switch (m.kind) {
		case 0: check(dropest_exclude_cell(ctx, m.a)); break;
		case 1: check(dropest_merge_cells(ctx, m.a, m.b)); break;
		case 2: {
			std::vector<uint64_t> src, tgt;
			for (auto const &t : m.targets) {   // the map's own iteration order, like Cell::merge_umis
				src.push_back(encode(t.first, _side_umi));
				tgt.push_back(encode(t.second, _side_umi));
			}
			send_side_strings(ctx);          // a target with N registers a new side string
			check(dropest_merge_umis(ctx, m.a, uint32_t(m.b), src.size(), src.data(), tgt.data()));
			break;
		}
		case 3: {
			const size_t gene = _gene_indexer.add(m.gene);
			const uint64_t umi = encode(m.umi, _side_umi);
			send_side_strings(ctx);
			check(dropest_add_umi_to_cell(ctx, m.a, uint32_t(gene), umi, m.mark, reinterpret_cast<const uint8_t *>(m.quality.data()), uint32_t(m.quality.size())));
			break;
		}
		default: throw std::runtime_error("internal: unknown pending mutation");
	}
}

CellsDataContainer::~CellsDataContainer() {
	if (_preview) dropest_ctx_destroy(_preview);
	for (dropest_shard *s : _shards) dropest_shard_destroy(s);
	dropest_ctx_destroy(_ctx);
}

void CellsDataContainer::single_only(const char *what) const {
	throw std::runtime_error(std::string(what) + " is not available on a container sharded over several GPUs");
}

std::vector<std::pair<std::string, std::string>> CellsDataContainer::merged_barcodes() const {
	std::vector<std::pair<std::string, std::string>> out;
	if (sharded()) {
		uint64_t n = 0;
		check(dropest_shard_merged_barcodes(_shards[0], &n, nullptr, nullptr));
		std::vector<uint64_t> src(n), tgt(n);
		if (n) check(dropest_shard_merged_barcodes(_shards[0], &n, src.data(), tgt.data()));
		for (uint64_t i = 0; i < n; ++i) out.emplace_back(decode(src[i]), decode(tgt[i]));
		return out;
	}
	const ids_t &t = merge_targets();
	for (size_t i = 0; i < t.size(); ++i) if (t[i] != i) out.emplace_back(cell(i).barcode(), cell(t[i]).barcode());
	std::sort(out.begin(), out.end());
	return out;
}

void CellsDataContainer::add_record(const ReadInfo &r) {   // CellsDataContainer.cpp:59-88
	if (_is_initialized) throw std::runtime_error("Container is already initialized");
	_preview_valid = false; ++_generation;
	const uint8_t mark = uint8_t(r.umi_mark.bits());
	const bool has_gene = !r.gene.empty();
	const uint64_t cb_code = encode(r.params.cell_barcode(), _side_cb);
	uint32_t chr = 0;
	if (has_gene) {
		const size_t ql = r.params.umi_quality().size();
		const uint64_t umi_code = encode(r.params.umi(), _side_umi);   // UMI side strings: first seen on gene-bearing reads only
		const uint32_t gid = uint32_t(_gene_indexer.add(r.gene));
		check_molecule_quality_length(cb_code, gid, umi_code, ql);   // throws where UMI::add_read would (UMI.cpp:26-28)
		_cb.push_back(cb_code);
		note_quality_length(ql);
		append_quality(r.params.umi_quality().data(), ql, true);
		_umi.push_back(umi_code);
		_gene.push_back(gid);
		// Stats::inc(chr) is reached only for exon / intron reads (CellsDataContainer.cpp:312-321)
		if (mark & (UMI::Mark::HAS_EXONS | UMI::Mark::HAS_INTRONS)) chr = uint32_t(_chr_indexer.add(r.chromosome_name));
	} else {
		_cb.push_back(cb_code);
		append_quality(nullptr, 0, false);
		_umi.push_back(1);   // ignored by the device for gene-less reads
		_gene.push_back(DROPEST_NO_GENE);
		chr = uint32_t(_chr_indexer.add(r.chromosome_name));   // :75
	}
	if (chr > 0xFFFF) throw std::runtime_error("more than 65536 chromosome names");
	_aux.push_back(chr | (uint32_t(mark) << 16));
	if (_cb.size() >= BATCH) flush();
}

bool CellsDataContainer::pack_code(std::string_view s, uint64_t &code) { return pack_bases(s.data(), s.size(), code); }

uint64_t CellsDataContainer::hash_name(std::string_view s) {   // FNV-1a
	uint64_t h = 1469598103934665603ull;
	for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
	static const uint64_t mask = [] { const char *e = getenv("DROPEST_BAM_TEST_GENE_HASH_BITS"); const int b = e ? atoi(e) : 0; return b > 0 && b < 64 ? (1ull << b) - 1ull : ~0ull; }();   // (tests: names that collide)
	return h & mask;
}

int64_t CellsDataContainer::lookup_gene(uint64_t gene_hash, std::string_view name) const {
	auto it = _gene_by_hash.find(gene_hash);
	if (it != _gene_by_hash.end() && _gene_indexer.get_value(it->second) == name) return int64_t(it->second);
	return -1;
}

uint32_t CellsDataContainer::intern_gene(std::string_view name, uint64_t hash) {   // the gene branch of add_record(ParsedRead)
	auto it = _gene_by_hash.find(hash);
	if (it != _gene_by_hash.end() && _gene_indexer.get_value(it->second) == name) return it->second;
	const uint32_t gid = uint32_t(_gene_indexer.add(std::string(name)));
	if (it == _gene_by_hash.end()) _gene_by_hash.emplace(hash, gid);   // (a colliding name keeps taking the slow path)
	return gid;
}

uint32_t CellsDataContainer::intern_chromosome_of_ref(int32_t ref_id) {
	if (ref_id < 0 || size_t(ref_id) >= _ref_names.size()) throw std::out_of_range("reference id outside set_reference_names");
	int32_t &slot = _ref_chr[size_t(ref_id)];
	if (slot < 0) slot = int32_t(_chr_indexer.add(_ref_names[size_t(ref_id)]));
	if (slot > 0xFFFF) throw std::runtime_error("more than 65536 chromosome names");
	return uint32_t(slot);
}

void CellsDataContainer::add_records_packed(const std::vector<PackedRun> &runs) {
	if (_is_initialized) throw std::runtime_error("Container is already initialized");
	if (!bulk_ingest_possible()) throw std::runtime_error("add_records_packed: the container is sharded or carries UMI qualities (use add_record)");
	size_t n = 0;
	for (const PackedRun &r : runs) n += r.n;
	if (!n) return;
	_preview_valid = false; ++_generation;
	flush();                                   // whatever add_record collected comes first
	if (_umi_quality_length == size_t(-1))     // the first gene-bearing read fixes the quality length: 0 (no quality strings)
		for (const PackedRun &r : runs) { for (size_t i = 0; i < r.n; ++i) if (r.gene[i] != DROPEST_NO_GENE) { _umi_quality_length = 0; _qual_pending = 0; break; } if (_umi_quality_length == 0) break; }
	if (_umi_quality_length == size_t(-1)) _qual_pending += n;
	_qual_reads += n;
	if (!_qual_lens.empty()) _qual_lens.insert(_qual_lens.end(), n, uint8_t(0));
	send_side_strings(_ctx);
	std::vector<const uint64_t *> pc, pu; std::vector<const uint32_t *> pg, pa; std::vector<uint64_t> cnt;
	for (const PackedRun &r : runs) { pc.push_back(r.cb); pu.push_back(r.umi); pg.push_back(r.gene); pa.push_back(r.aux); cnt.push_back(r.n); }
	check(dropest_push_reads_gather(_ctx, runs.size(), pc.data(), pu.data(), pg.data(), pa.data(), cnt.data()));
}

void CellsDataContainer::add_records_packed(const std::vector<PackedRun> &runs, const std::vector<const uint8_t *> &quality, size_t ql) {
	if (_is_initialized) throw std::runtime_error("Container is already initialized");
	if (!bulk_ingest_possible_with_quality(ql) || quality.size() != runs.size())
		throw std::runtime_error("add_records_packed: UMI quality strings of this length cannot be taken in bulk here (use add_record)");
	size_t n = 0;
	for (const PackedRun &r : runs) n += r.n;
	if (!n) return;
	_preview_valid = false; ++_generation;
	flush();                                   // whatever add_record collected comes first
	if (_umi_quality_length == size_t(-1)) {   // the first gene-bearing read fixes the length (append_quality): the gene-less reads before it get their rows now
		bool any_gene = false;
		for (const PackedRun &r : runs) { for (size_t i = 0; i < r.n && !any_gene; ++i) any_gene = r.gene[i] != DROPEST_NO_GENE; if (any_gene) break; }
		if (any_gene) {
			_umi_quality_length = ql;
			_qual.insert(_qual.end(), _qual_pending * ql, uint8_t(0));
			_qual_pending = 0;
		}
	}
	if (_umi_quality_length == size_t(-1)) _qual_pending += n;
	else for (size_t k = 0; k < runs.size(); ++k) _qual.insert(_qual.end(), quality[k], quality[k] + runs[k].n * ql);
	_qual_reads += n;
	send_side_strings(_ctx);
	std::vector<const uint64_t *> pc, pu; std::vector<const uint32_t *> pg, pa; std::vector<uint64_t> cnt;
	for (const PackedRun &r : runs) { pc.push_back(r.cb); pu.push_back(r.umi); pg.push_back(r.gene); pa.push_back(r.aux); cnt.push_back(r.n); }
	check(dropest_push_reads_gather(_ctx, runs.size(), pc.data(), pu.data(), pg.data(), pa.data(), cnt.data()));
}

void CellsDataContainer::add_records_packed(const uint64_t *cb, const uint64_t *umi, const uint32_t *gene, const uint32_t *aux, size_t n) {
	if (_is_initialized) throw std::runtime_error("Container is already initialized");
	if (!bulk_ingest_possible()) throw std::runtime_error("add_records_packed: the container is sharded or carries UMI qualities (use add_record)");
	_preview_valid = false; ++_generation;
	if (!n) return;
	flush();                                   // whatever add_record collected comes first
	if (_umi_quality_length == size_t(-1))     // the first gene-bearing read fixes the quality length: 0 (no quality strings)
		for (size_t i = 0; i < n; ++i) if (gene[i] != DROPEST_NO_GENE) { _umi_quality_length = 0; _qual_pending = 0; break; }
	if (_umi_quality_length == size_t(-1)) _qual_pending += n;   // (only gene-less reads so far: still no length)
	_qual_reads += n;                                            // these reads carry no quality string: rows of length 0
	if (!_qual_lens.empty()) _qual_lens.insert(_qual_lens.end(), n, uint8_t(0));
	send_side_strings(_ctx);
	for (size_t at = 0; at < n; at += BATCH * 8) {
		const size_t m = std::min(n - at, BATCH * 8);
		check(dropest_push_reads(_ctx, cb + at, umi + at, gene + at, aux + at, m));
	}
}

void CellsDataContainer::add_records_packed_device(const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene, const uint32_t *d_aux, size_t n, bool any_gene) {
	if (_is_initialized) throw std::runtime_error("Container is already initialized");
	if (!bulk_ingest_possible()) throw std::runtime_error("add_records_packed: the container is sharded or carries UMI qualities (use add_record)");
	_preview_valid = false; ++_generation;
	if (!n) return;
	flush();                                   // whatever add_record collected comes first
	if (_umi_quality_length == size_t(-1) && any_gene) { _umi_quality_length = 0; _qual_pending = 0; }   // the first gene-bearing read fixes the quality length: none
	if (_umi_quality_length == size_t(-1)) _qual_pending += n;
	_qual_reads += n;
	static const bool trace = getenv("DROPEST_BAM_TRACE") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	if (!_qual_lens.empty()) _qual_lens.insert(_qual_lens.end(), n, uint8_t(0));
	const auto t1 = std::chrono::steady_clock::now();
	send_side_strings(_ctx);
	const auto t2 = std::chrono::steady_clock::now();
	check(dropest_push_reads_device(_ctx, d_cb, d_umi, d_gene, d_aux, n, 0));
	if (trace) {
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		std::fprintf(stderr, "[bam] container: %zu reads: quality lengths %.2f ms, side strings %.2f ms (%zu), push %.2f ms\n", n, ms(t0, t1), ms(t1, t2), _side.size(), ms(t2, std::chrono::steady_clock::now()));
	}
}

void CellsDataContainer::add_records_packed_device(const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene, const uint32_t *d_aux, size_t n, bool any_gene,
                                                   const uint8_t *quality_rows, size_t ql) {
	if (_is_initialized) throw std::runtime_error("Container is already initialized");
	if (!bulk_ingest_possible_with_quality(ql) || !quality_rows) throw std::runtime_error("add_records_packed: UMI quality strings of this length cannot be taken in bulk here (use add_record)");
	_preview_valid = false; ++_generation;
	if (!n) return;
	flush();
	if (_umi_quality_length == size_t(-1) && any_gene) {   // the first gene-bearing read fixes the length: the gene-less reads before it get their rows now
		_umi_quality_length = ql;
		_qual.insert(_qual.end(), _qual_pending * ql, uint8_t(0));
		_qual_pending = 0;
	}
	if (_umi_quality_length == size_t(-1)) _qual_pending += n;
	else _qual.insert(_qual.end(), quality_rows, quality_rows + n * ql);
	_qual_reads += n;
	send_side_strings(_ctx);
	check(dropest_push_reads_device(_ctx, d_cb, d_umi, d_gene, d_aux, n, 0));
}

void CellsDataContainer::set_reference_names(const std::vector<std::string> &names) {
	_ref_names = names;
	_ref_chr.assign(names.size(), -1);
}

void CellsDataContainer::add_record(const ParsedRead &r) {   // same statements as add_record(const ReadInfo&), on parsed fields
	if (_is_initialized) throw std::runtime_error("Container is already initialized");
	_preview_valid = false; ++_generation;
	if (r.ref_id < 0 || size_t(r.ref_id) >= _ref_names.size()) throw std::out_of_range("reference id outside set_reference_names");
	const bool has_gene = !r.gene.empty();
	const uint64_t cb_code = r.cb_code ? r.cb_code : encode(std::string(r.cb), _side_cb);
	auto chr_index = [&]() {
		int32_t &slot = _ref_chr[size_t(r.ref_id)];
		if (slot < 0) slot = int32_t(_chr_indexer.add(_ref_names[size_t(r.ref_id)]));
		return uint32_t(slot);
	};
	uint32_t chr = 0;
	if (has_gene) {
		const size_t ql = r.umi_quality_length;
		const uint64_t umi_code = r.umi_code ? r.umi_code : encode(std::string(r.umi), _side_umi);
		uint32_t gid;
		auto it = r.gene_id >= 0 ? _gene_by_hash.end() : _gene_by_hash.find(r.gene_hash);
		if (r.gene_id >= 0) gid = uint32_t(r.gene_id);
		else if (it != _gene_by_hash.end() && _gene_indexer.get_value(it->second) == r.gene) gid = it->second;
		else {
			gid = uint32_t(_gene_indexer.add(std::string(r.gene)));
			if (it == _gene_by_hash.end()) _gene_by_hash.emplace(r.gene_hash, gid);   // (a colliding name keeps taking the slow path)
		}
		check_molecule_quality_length(cb_code, gid, umi_code, ql);
		_cb.push_back(cb_code);
		note_quality_length(ql);
		append_quality(r.umi_quality.data(), ql, true);
		_umi.push_back(umi_code);
		_gene.push_back(gid);
		if (r.mark & (UMI::Mark::HAS_EXONS | UMI::Mark::HAS_INTRONS)) chr = chr_index();
	} else {
		_cb.push_back(cb_code);
		append_quality(nullptr, 0, false);
		_umi.push_back(1);
		_gene.push_back(DROPEST_NO_GENE);
		chr = chr_index();
	}
	if (chr > 0xFFFF) throw std::runtime_error("more than 65536 chromosome names");
	_aux.push_back(chr | (uint32_t(r.mark) << 16));
	if (_cb.size() >= BATCH) flush();
}

// One fixed-length quality row per read for dropest_set_umi_qualities: the bytes of gene-bearing reads, zeros for reads
// without a gene (never read: they do not reach Gene::add_umi).
void CellsDataContainer::append_quality(const char *q, size_t len, bool has_gene) {
	++_qual_reads;
	if (!_qual_lens.empty()) _qual_lens.push_back(uint8_t(has_gene ? len : 0));
	if (!has_gene) {
		if (_umi_quality_length == size_t(-1)) ++_qual_pending;
		else _qual.insert(_qual.end(), _umi_quality_length, uint8_t(0));
		return;
	}
	if (_qual_pending) { _qual.insert(_qual.end(), _qual_pending * _umi_quality_length, uint8_t(0)); _qual_pending = 0; }
	_qual.insert(_qual.end(), reinterpret_cast<const uint8_t *>(q), reinterpret_cast<const uint8_t *>(q) + len);
	if (len < _umi_quality_length) _qual.insert(_qual.end(), _umi_quality_length - len, uint8_t(0));   // (only with several lengths)
}

// The quality length is a property of the MOLECULE in the reference (fixed by the read that creates it, Gene.cpp:20; UMI::add_read throws
// "Wrong quality length" for a later read of the same molecule with another one, UMI.cpp:26-28).  Reads of different molecules may differ:
// from the first such read on, the length of every read is kept beside the rows (widened to the longest string), and the per-molecule
// check runs on the device in set_initialized (dropest_set_umi_qualities_var) -- same exception text, for the read the reference would have
// stopped at.
// The quality length of every molecule, from the moment two gene-bearing reads differ in it.  Until then every molecule has the one
// length seen so far, so nothing is kept; at that moment the molecules that exist are read from the preview (a second context over the
// reads pushed so far: one pass on the device -- the reference's per-read std::map look-ups are not restated on the host for the
// common case of one length).  From then on each gene-bearing read looks its molecule up here: an existing molecule with another
// length is UMI::add_read's exception (UMI.cpp:26-28), raised by the add_record that the reference raises it from.
void CellsDataContainer::check_molecule_quality_length(uint64_t cb_code, uint32_t gene, uint64_t umi_code, size_t ql) {
	if (_umi_quality_length == size_t(-1) || sharded()) return;          // no molecule yet / sharded containers check in the step
	if (!_mol_qlen_tracking) {
		if (ql == _umi_quality_length) return;
		if (ql > 255) throw std::runtime_error("UMI quality strings longer than 255");
		dropest_ctx *pv = view();                                          // flushes the pending batch; the current read is not in it
		uint64_t n_mol = 0, n_cells = 0;
		check(dropest_molecules(pv, &n_mol, nullptr, nullptr, nullptr, nullptr, nullptr));
		check(dropest_total_cells(pv, &n_cells));
		std::vector<uint32_t> cell(n_mol), g(n_mol), reads(n_mol);
		std::vector<uint64_t> umi(n_mol);
		std::vector<uint8_t> mark(n_mol);
		if (n_mol) check(dropest_molecules(pv, &n_mol, cell.data(), g.data(), umi.data(), reads.data(), mark.data()));
		std::vector<dropest_cell_row> rows(n_cells);
		if (n_cells) check(dropest_cell_rows(pv, 0, n_cells, rows.data()));
		_mol_qlen.reserve(size_t(n_mol) * 2);
		for (uint64_t i = 0; i < n_mol; ++i) _mol_qlen.emplace(MolKey{rows[cell[i]].barcode, umi[i], g[i]}, uint8_t(_umi_quality_length));
		_mol_qlen_tracking = true;
		_preview_valid = false; ++_generation;                             // (the next accessor sees the read that follows)
	}
	auto ins = _mol_qlen.emplace(MolKey{cb_code, umi_code, gene}, uint8_t(ql));
	if (!ins.second && ins.first->second != ql)
		throw std::runtime_error("Wrong quality length: " + std::to_string(ql) + ", expected: " + std::to_string(unsigned(ins.first->second)));
}

void CellsDataContainer::note_quality_length(size_t ql) {
	if (ql > 255) throw std::runtime_error("UMI quality strings longer than 255");
	if (_umi_quality_length == size_t(-1)) { _umi_quality_length = ql; return; }
	if (ql == _umi_quality_length && _qual_lens.empty()) return;
	if (_qual_lens.empty()) _qual_lens.assign(_qual_reads, uint8_t(_umi_quality_length));   // (rows of gene-less reads are never looked at)
	if (ql > _umi_quality_length) {   // widen the rows
		const size_t old = _umi_quality_length, rows = _qual_reads - _qual_pending;
		std::vector<uint8_t> wide(rows * ql, uint8_t(0));
		for (size_t i = 0; old && i < rows; ++i) std::memcpy(wide.data() + i * ql, _qual.data() + i * old, old);
		_qual.swap(wide);
		_umi_quality_length = ql;
	}
}

void CellsDataContainer::flush() {
	if (_cb.empty()) return;
	std::vector<const char *> ptrs(_side.size());
	for (size_t i = 0; i < _side.size(); ++i) ptrs[i] = _side[i].c_str();
	if (sharded()) {
		// Every shard holds ONE contiguous range of the stream, ascending with the shard: the first SHARD_QUOTA reads go to
		// shard 0, the next to shard 1, ... (the stream's length is not known while it arrives; the pass re-distributes the
		// reads by barcode owner anyway, so WHERE they wait only decides which PCIe link carried them).
		if (_side.size() != _side_sent) { for (dropest_shard *s : _shards) check(dropest_set_side_strings(dropest_shard_ctx(s), ptrs.data(), ptrs.size())); _side_sent = _side.size(); }
		const size_t shard = std::min<size_t>(_shards.size() - 1, size_t(_batches / std::max<size_t>(1, shard_quota / BATCH)));
		check(dropest_shard_push_reads(_shards[shard], _cb.data(), _umi.data(), _gene.data(), _aux.data(), _cb.size(), _batches * BATCH));
		if (_shard_reads.size() != _shards.size()) _shard_reads.assign(_shards.size(), 0);
		_shard_reads[shard] += _cb.size();
		++_batches;
		_cb.clear(); _umi.clear(); _gene.clear(); _aux.clear();
		return;
	}
	check(dropest_set_side_strings(_ctx, ptrs.data(), ptrs.size()));
	check(dropest_push_reads(_ctx, _cb.data(), _umi.data(), _gene.data(), _aux.data(), _cb.size()));
	_cb.clear(); _umi.clear(); _gene.clear(); _aux.clear();
}

void CellsDataContainer::set_initialized() {   // CellsDataContainer.cpp:163-175
	++_generation;
	if (_is_initialized) throw std::runtime_error("Container is already initialized");
	flush();
	if (sharded()) {
		if (_umi_quality_length != size_t(-1) && _umi_quality_length > 0) {
			// every shard gets the strings of ITS range of the stream (they travel with the reads in the exchange)
			if (!_qual_lens.empty()) single_only("UMI quality strings of several lengths");
			const size_t ql = _umi_quality_length;
			if (_shard_reads.size() != _shards.size()) _shard_reads.assign(_shards.size(), 0);
			size_t at = 0;
			for (size_t k = 0; k < _shards.size(); ++k) {
				if ((at + _shard_reads[k]) * ql > _qual.size()) throw std::runtime_error("internal: fewer UMI quality rows than reads");
				check(dropest_shard_set_umi_qualities(_shards[k], _qual.data() + at * ql, uint32_t(ql), _shard_reads[k]));
				at += _shard_reads[k];
			}
			std::vector<uint8_t>().swap(_qual);
		}
		_is_initialized = true;   // the sharded pass (partition, exchange, pipeline, merges) runs as one piece in merge_and_filter
		return;
	}
	if (_umi_quality_length != size_t(-1) && _umi_quality_length > 0) {
		if (_qual_lens.empty()) check(dropest_set_umi_qualities(_ctx, _qual.data(), uint32_t(_umi_quality_length), _qual.size() / _umi_quality_length));
		else check(dropest_set_umi_qualities_var(_ctx, _qual.data(), uint32_t(_umi_quality_length), _qual_lens.data(), _qual_lens.size()));
		std::vector<uint8_t>().swap(_qual);
	}
	const dropest_status st = dropest_set_initialized(_ctx);
	if (st == DROPEST_ERR_UNSUPPORTED && std::strstr(dropest_last_error(), "sort key needs")) {
		// cell id + gene + UMI do not fit the 64-bit key of one context: the same reads as 2^k shards on the same device
		// (dropest_ctx_split; every shard numbers 1 / 2^k of the barcodes).  merge_and_filter doubles k while a shard still fails.
		uint32_t cb = 0, gb = 0, ub = 0;
		check(dropest_key_width(_ctx, &cb, &gb, &ub));
		_split_parts = 1;
		while (_split_parts < 64 && int(cb + gb + ub) - 64 > int(__builtin_ctz(unsigned(_split_parts)))) _split_parts *= 2;
		split_for_wide_keys();
		_is_initialized = true;
		return;
	}
	check(st);
	_is_initialized = true;
	if (_preview) { dropest_ctx_destroy(_preview); _preview = nullptr; _preview_valid = false; }
	for (const PendingMutation &m : _pending) apply_mutation(_ctx, m);   // what was merged / excluded before the initialisation
	_pending.clear();
}

void CellsDataContainer::split_for_wide_keys() {
	for (dropest_shard *s : _shards) dropest_shard_destroy(s);
	_shards.assign(size_t(_split_parts), nullptr);
	const dropest_status st = dropest_ctx_split(_ctx, int32_t(_split_parts), _shards.data());
	if (st != DROPEST_OK) { _shards.clear(); fail(st); }
}

void CellsDataContainer::merge_and_filter() {   // CellsDataContainer.cpp:39-57
	if (!_is_initialized) throw std::runtime_error("You must initialize container");
	++_generation;
	if (sharded()) {
		for (;;) {
			const dropest_status st = dropest_shard_group_step(_shards.data(), int32_t(_shards.size()));
			if (st == DROPEST_ERR_UNSUPPORTED && _split_parts && _split_parts < 64 && std::strstr(dropest_last_error(), "sort key needs")) {
				_split_parts *= 2;   // owners are not perfectly even: one more bit
				split_for_wide_keys();
				continue;
			}
			check(st);
			return;
		}
	}
	_umi_indexer_valid = false;
	check(dropest_merge_and_filter(_ctx));
}

size_t CellsDataContainer::total_cells_number() const { if (sharded()) single_only("total_cells_number"); uint64_t n = 0; check(dropest_total_cells(view(), &n)); return size_t(n); }
size_t CellsDataContainer::real_cells_number() const { if (sharded()) single_only("real_cells_number"); uint64_t n = 0; check(dropest_real_cells(_ctx, &n)); return size_t(n); }

size_t CellsDataContainer::cell_id_by_cb(const std::string &barcode) const { if (sharded()) single_only("cell_id_by_cb");
	uint64_t code;
	if (!pack2(barcode, code)) {
		auto it = _side_cb.find(barcode);
		if (it == _side_cb.end()) throw std::out_of_range("unknown barcode: " + barcode);
		code = it->second;
	}
	int64_t id = -1;
	check(dropest_cell_id_by_cb(view(), code, &id));
	if (id < 0) throw std::out_of_range("unknown barcode: " + barcode);
	return size_t(id);
}

const CellsDataContainer::ids_t &CellsDataContainer::filtered_cells() const { if (sharded()) single_only("filtered_cells");
	uint64_t n = 0;
	check(dropest_filtered_cells(_ctx, &n, nullptr));
	std::vector<uint64_t> ids(n);
	if (n) check(dropest_filtered_cells(_ctx, &n, ids.data()));
	_filtered_cache.assign(ids.begin(), ids.end());
	return _filtered_cache;
}

const CellsDataContainer::ids_t &CellsDataContainer::merge_targets() const { if (sharded()) single_only("merge_targets");
	uint64_t n = 0;
	check(dropest_merge_targets(_ctx, &n, nullptr, nullptr));
	std::vector<uint64_t> src(n), tgt(n);
	if (n) check(dropest_merge_targets(_ctx, &n, src.data(), tgt.data()));
	_merge_targets_cache.resize(total_cells_number());
	for (size_t i = 0; i < _merge_targets_cache.size(); ++i) _merge_targets_cache[i] = i;
	for (size_t i = 0; i < n; ++i) _merge_targets_cache[size_t(src[i])] = size_t(tgt[i]);
	return _merge_targets_cache;
}

// The public mutators (CellsDataContainer.h:88-95).  On an initialised container they act on it; before that (the reference
// takes them from the first add_record on) they act on the preview -- so that errors surface where the reference throws them --
// and are replayed on the real context by set_initialized().
void CellsDataContainer::exclude_cell(size_t index) {   // CellsDataContainer.cpp:106-109
	if (sharded()) single_only("exclude_cell");
	PendingMutation m; m.kind = 0; m.a = index;
	apply_mutation(view(), m);
	if (!_is_initialized) _pending.push_back(m);
	++_generation;
}

void CellsDataContainer::merge_cells(size_t source_cell_ind, size_t target_cell_ind) {   // :90-104
	if (sharded()) single_only("merge_cells");
	PendingMutation m; m.kind = 1; m.a = source_cell_ind; m.b = target_cell_ind;
	_umi_indexer_valid = false;
	apply_mutation(view(), m);
	if (!_is_initialized) _pending.push_back(m);
	++_generation;
}

void CellsDataContainer::add_umi_to_cell(size_t cell_id, const ReadInfo &read_info) {   // :356-364
	if (sharded()) single_only("add_umi_to_cell");
	if (read_info.gene.empty()) throw std::runtime_error("add_umi_to_cell: the read has no gene");
	PendingMutation m; m.kind = 3; m.a = cell_id; m.gene = read_info.gene; m.umi = read_info.params.umi(); m.mark = uint8_t(read_info.umi_mark.bits());
	m.quality = read_info.params.umi_quality();
	_umi_indexer_valid = false;
	apply_mutation(view(), m);
	if (!_is_initialized) _pending.push_back(m);
	++_generation;
}

const StringIndexer &CellsDataContainer::umi_indexer() const {
	if (sharded()) single_only("umi_indexer");
	if (!_is_initialized) throw std::runtime_error("You must initialize container");
	if (!_umi_indexer_valid) {
		uint64_t n = 0;
		check(dropest_umi_first_seen(_ctx, &n, nullptr));
		std::vector<uint64_t> codes(n);
		if (n) check(dropest_umi_first_seen(_ctx, &n, codes.data()));
		_umi_indexer_cache = StringIndexer();
		for (uint64_t c : codes) _umi_indexer_cache.add(decode(c));
		_umi_indexer_valid = true;
	}
	return _umi_indexer_cache;
}

void CellsDataContainer::merge_umis(size_t cell_id, size_t gene, const s_s_hash_t &merge_targets) {   // :209-213, Cell.cpp:31-42
	if (sharded()) single_only("merge_umis");
	PendingMutation m; m.kind = 2; m.a = cell_id; m.b = gene; m.targets = merge_targets;
	_umi_indexer_valid = false;
	apply_mutation(view(), m);
	if (!_is_initialized) _pending.push_back(m);
	++_generation;
}

long CellsDataContainer::get_merge_target(size_t base_cell_ind) const {
	int64_t t = 0;
	check(dropest_merge_target(_ctx, base_cell_ind, &t));
	return long(t);
}

Merge::PoissonTargetEstimator::EstimationResult
Merge::PoissonTargetEstimator::estimate_intersection_prob(const CellsDataContainer &container, size_t cell1_ind, size_t cell2_ind) const {
	size_t n = 0; double expected = -1, prob = 1;
	container.poisson_intersection(cell1_ind, cell2_ind, n, expected, prob);
	return EstimationResult{n, expected, prob};
}

void CellsDataContainer::poisson_intersection(size_t cell1_ind, size_t cell2_ind, size_t &intersection, double &expected, double &probability) const {
	uint64_t n = 0;
	check(dropest_poisson_intersection_prob(_ctx, cell1_ind, cell2_ind, &n, &expected, &probability));
	intersection = size_t(n);
}

Cell CellsDataContainer::cell(size_t index) const {
	if (sharded()) single_only("cell(index)");
	Cell c;
	c._owner = this; c._ctx = view(); c._id = index;
	check(dropest_cell_rows(c._ctx, index, 1, &c._row));   // DROPEST_ERR_RANGE -> std::out_of_range (vector::at in the reference)
	c._barcode = decode(c._row.barcode);
	c._gen = _generation;
	return c;
}

// A Cell of an unsharded container is LIVE like the reference's `Cell &` (CellsDataContainer.h:107): whatever changed the container since
// the row was read (add_record before set_initialized, the merges, the public mutators) makes the next accessor read it again.
void Cell::sync() const {
	if (!_owner || _gen == 0 || _owner->sharded() || _gen == _owner->_generation) return;
	Cell *self = const_cast<Cell *>(this);
	self->_ctx = _owner->view();
	if (dropest_cell_rows(self->_ctx, _id, 1, &self->_row) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
	self->_gen = _owner->_generation;
}

Cell &CellsDataContainer::cell(size_t index) {
	Cell &slot = _cell_cache[index];
	slot = static_cast<const CellsDataContainer *>(this)->cell(index);
	return slot;
}

std::vector<Cell> CellsDataContainer::real_cells() const {
	std::vector<Cell> out;
	if (!sharded()) {
		const size_t n = total_cells_number();
		std::vector<dropest_cell_row> rows(n);
		if (n) check(dropest_cell_rows(_ctx, 0, n, rows.data()));
		for (size_t i = 0; i < n; ++i) {
			if (!rows[i].is_real) continue;
			Cell c;
			c._owner = this; c._ctx = _ctx; c._id = i; c._row = rows[i]; c._barcode = decode(rows[i].barcode);
			out.push_back(std::move(c));
		}
		return out;
	}
	// the columns of the global cm_raw ARE the real cells in the cell-id order of one container
	uint64_t ncols = 0, nnz = 0;
	const uint64_t *bc = nullptr;
	check(dropest_shard_matrix(_shards[0], 0, &ncols, &nnz, nullptr, nullptr, nullptr, &bc));
	for (uint64_t j = 0; j < ncols; ++j) {
		const uint32_t w = dropest_owner_of(bc[j], uint32_t(_shards.size()));
		Cell c;
		c._owner = this; c._ctx = dropest_shard_ctx(_shards[w]);
		int64_t id = -1;
		check(dropest_cell_id_by_cb(c._ctx, bc[j], &id));
		if (id < 0) throw std::runtime_error("internal: a column of the global matrix is unknown to the shard that owns its barcode");
		c._id = size_t(id);
		check(dropest_cell_rows(c._ctx, c._id, 1, &c._row));
		c._barcode = decode(bc[j]);
		out.push_back(std::move(c));
	}
	return out;
}

std::vector<size_t> CellsDataContainer::filtered_positions(const std::vector<Cell> &real) const {
	std::vector<size_t> out;
	if (!sharded()) {
		std::unordered_map<size_t, size_t> at;
		for (size_t k = 0; k < real.size(); ++k) at.emplace(real[k]._id, k);
		for (size_t id : filtered_cells()) out.push_back(at.at(id));
		return out;
	}
	std::unordered_map<uint64_t, size_t> at;
	for (size_t k = 0; k < real.size(); ++k) at.emplace(real[k]._row.barcode, k);
	uint64_t ncols = 0, nnz = 0;
	const uint64_t *bc = nullptr;
	check(dropest_shard_matrix(_shards[0], 1, &ncols, &nnz, nullptr, nullptr, nullptr, &bc));
	for (uint64_t j = 0; j < ncols; ++j) out.push_back(at.at(bc[j]));
	return out;
}

CellsDataContainer::s_i_hash_t CellsDataContainer::get_stat_by_real_cells(Stats::CellStatType type) const {   // :278-289
	s_i_hash_t res;
	if (sharded()) {
		for (const Cell &c : real_cells()) res[c.barcode()] = c.stat(type);
		return res;
	}
	const size_t n = total_cells_number();
	std::vector<dropest_cell_row> rows(n);
	if (n) check(dropest_cell_rows(_ctx, 0, n, rows.data()));
	for (auto const &r : rows)
		if (r.is_real) res[decode(r.barcode)] = type == Stats::TOTAL_READS_PER_CB ? r.total_reads : r.total_umis;
	return res;
}

void CellsDataContainer::get_stat_by_real_cells(Stats::CellChrStatType stat, names_t &cell_barcodes, names_t &chromosome_names,
                                                counts_t &counts) const {   // :291-307, chromosomes in first-seen order
	if (sharded()) {
		// every shard reports the cells it owns; rows come out in the cell-id order of one container (real_cells())
		struct Entry { uint64_t barcode; uint32_t chr; int32_t cnt; };
		std::vector<Entry> entries;
		std::vector<char> present(_chr_indexer.values().size(), 0);
		for (dropest_shard *s : _shards) {
			dropest_ctx *ctx = dropest_shard_ctx(s);
			uint64_t n = 0;
			check(dropest_chr_stats(ctx, &n, nullptr, nullptr, nullptr, nullptr));
			std::vector<uint32_t> cell(n), kind(n), chr(n);
			std::vector<int32_t> cnt(n);
			if (n) check(dropest_chr_stats(ctx, &n, cell.data(), kind.data(), chr.data(), cnt.data()));
			uint64_t nc = 0;
			check(dropest_total_cells(ctx, &nc));
			std::vector<dropest_cell_row> rows(nc);
			if (nc) check(dropest_cell_rows(ctx, 0, nc, rows.data()));
			for (size_t i = 0; i < n; ++i)
				if (kind[i] == uint32_t(stat)) { present[chr[i]] = 1; entries.push_back(Entry{rows[cell[i]].barcode, chr[i], cnt[i]}); }
		}
		std::vector<int> column(present.size(), -1);
		for (size_t c = 0; c < present.size(); ++c) if (present[c]) { column[c] = int(chromosome_names.size()); chromosome_names.push_back(_chr_indexer.get_value(c)); }
		const size_t width = chromosome_names.size();
		std::unordered_map<uint64_t, std::vector<int>> row_of;
		for (const Entry &e : entries) {
			auto it = row_of.find(e.barcode);
			if (it == row_of.end()) it = row_of.emplace(e.barcode, std::vector<int>(width, 0)).first;
			it->second[size_t(column[e.chr])] = e.cnt;
		}
		for (const Cell &c : real_cells()) {
			auto it = row_of.find(c.barcode_code());
			if (it == row_of.end()) continue;
			cell_barcodes.push_back(c.barcode());
			counts.insert(counts.end(), it->second.begin(), it->second.end());
		}
		return;
	}
	uint64_t n = 0;
	check(dropest_chr_stats(_ctx, &n, nullptr, nullptr, nullptr, nullptr));
	std::vector<uint32_t> cell(n), kind(n), chr(n);
	std::vector<int32_t> cnt(n);
	if (n) check(dropest_chr_stats(_ctx, &n, cell.data(), kind.data(), chr.data(), cnt.data()));
	std::vector<char> present(_chr_indexer.values().size(), 0);
	for (size_t i = 0; i < n; ++i) if (kind[i] == uint32_t(stat)) present[chr[i]] = 1;
	std::vector<int> column(present.size(), -1);
	for (size_t c = 0; c < present.size(); ++c) if (present[c]) { column[c] = int(chromosome_names.size()); chromosome_names.push_back(_chr_indexer.get_value(c)); }
	const size_t width = chromosome_names.size();
	// the cells' barcodes from ONE fetch of the rows (round 6: cell(index) twice per row here was two device round trips each -- 3 000 of
	// them, 59 of the 127 ms of save_results on a 500-cell sample)
	const size_t n_cells = total_cells_number();
	std::vector<dropest_cell_row> all_rows(n_cells);
	if (n_cells) check(dropest_cell_rows(_ctx, 0, n_cells, all_rows.data()));
	size_t i = 0;
	while (i < n) {
		const uint32_t cur = cell[i];
		std::vector<int> row(width, 0);
		bool any = false;
		for (; i < n && cell[i] == cur; ++i)
			if (kind[i] == uint32_t(stat)) { row[size_t(column[chr[i]])] = cnt[i]; any = true; }
		if (!any) continue;   // Stats::get returns false for a cell without entries of this kind (Stats.cpp:50-63)
		if (cur >= n_cells) throw std::out_of_range("cell index out of range");
		cell_barcodes.push_back(decode(all_rows[cur].barcode));
		counts.insert(counts.end(), row.begin(), row.end());
	}
}

CellsDataContainer::s_ul_hash_t CellsDataContainer::umi_distribution() const {
	uint64_t n = 0;
	check(dropest_umi_distribution(_ctx, &n, nullptr, nullptr));
	std::vector<uint64_t> umi(n), cnt(n);
	if (n) check(dropest_umi_distribution(_ctx, &n, umi.data(), cnt.data()));
	s_ul_hash_t res;
	for (size_t i = 0; i < n; ++i) res[decode(umi[i])] = size_t(cnt[i]);
	return res;
}

static uint64_t counter(dropest_ctx *ctx, int i) { uint64_t c[4] = {0, 0, 0, 0}; if (dropest_global_counters(ctx, c) != DROPEST_OK) throw std::runtime_error(dropest_last_error()); return c[i]; }
size_t CellsDataContainer::intergenic_reads_num() const { return size_t(counter(_ctx, 0)); }
size_t CellsDataContainer::has_exon_reads_num() const { return size_t(counter(_ctx, 1)); }
size_t CellsDataContainer::has_intron_reads_num() const { return size_t(counter(_ctx, 2)); }
size_t CellsDataContainer::has_not_annotated_reads_num() const { return size_t(counter(_ctx, 3)); }

std::vector<Cell::MoleculeRow> Cell::molecules() const {
	sync();
	uint64_t n = 0;
	dropest_ctx *h = _ctx ? _ctx : _owner->handle();
	if (dropest_cell_molecules(h, _id, &n, nullptr, nullptr, nullptr, nullptr) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
	std::vector<uint32_t> gene(n), reads(n);
	std::vector<uint64_t> umi(n);
	std::vector<uint8_t> mark(n);
	if (n && dropest_cell_molecules(h, _id, &n, gene.data(), umi.data(), reads.data(), mark.data()) != DROPEST_OK)
		throw std::runtime_error(dropest_last_error());
	uint32_t ql = 0;
	if (dropest_umi_quality_length(h, &ql) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
	std::vector<uint32_t> qsum(size_t(n) * ql), qlen(n, 0u);
	if (n && ql && dropest_cell_molecule_qualities(h, _id, n, qsum.data()) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
	if (n && ql && dropest_cell_molecule_quality_lengths(h, _id, n, qlen.data()) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
	std::vector<MoleculeRow> out;
	for (size_t i = 0; i < n; ++i) {
		UMI::Mark m;
		if (mark[i] & 1) m.add(UMI::Mark::HAS_NOT_ANNOTATED);
		if (mark[i] & 2) m.add(UMI::Mark::HAS_EXONS);
		if (mark[i] & 4) m.add(UMI::Mark::HAS_INTRONS);
		out.push_back(MoleculeRow{_owner->gene_indexer().get_value(gene[i]), _owner->decode(umi[i]), reads[i], m,
		                          std::vector<unsigned>(qsum.begin() + long(i * ql), qsum.begin() + long(i * ql + qlen[i]))});   // _sum_quality of ITS length
	}
	return out;
}

std::unordered_map<std::string, size_t> Cell::requested_umis_per_gene(const UMI::Mark::query_t &query, bool return_reads) const {
	std::unordered_map<std::string, size_t> res;   // Cell.cpp:54-68: inserted in gene-index order
	std::string cur; size_t acc = 0; bool open = false;
	auto close = [&]() { if (open && acc) res.emplace(cur, acc); };
	for (auto const &m : molecules()) {
		if (!open || m.gene != cur) { close(); cur = m.gene; acc = 0; open = true; }
		if (m.mark.match(query)) acc += return_reads ? m.read_count : 1;
	}
	close();
	return res;
}

// ---- ResultsPrinter ---------------------------------------------------------------------------------

ResultsPrinter::SparseMatrix ResultsPrinter::get_count_matrix(const CellsDataContainer &c, bool filtered, bool reference_row_order) const {
	if (c.sharded()) {
		if (reads_output) throw std::runtime_error("reads_output is not available on a container sharded over several GPUs");
		return sharded_matrix(c, filtered, reference_row_order);
	}
	// the dgCMatrix slots i / x (ResultsPrinter.cpp:433-442).  The library sends a large matrix over PCIe as bytes (2 per entry instead of 8)
	// and widens it into these slots on its host threads while the copy is still running; a matrix whose sparse columns do not suit the byte
	// form comes as 32-bit arrays directly -- either way this call cannot fail for the shape of the data.
	// walk_byte_form: the named matrix is built straight from the bytes that crossed PCIe (this thread decodes each column as it names it);
	// no u32 arrays in between and no decode threads of the library involved.  A matrix the byte form refuses comes as 32-bit arrays.
	if (walk_byte_form) {
		dropest_matrix_bytes B{};
		const dropest_status st = dropest_count_matrix_csc_bytes(c.handle(), filtered ? 1 : 0, reads_output ? 1 : 0, &B);
		if (st == DROPEST_OK) return named_matrix(c, filtered, reference_row_order, ColumnSource(B));
		if (st != DROPEST_ERR_UNSUPPORTED) throw std::runtime_error(dropest_last_error());
	}
	uint64_t ncols = 0, nnz = 0;
	const uint32_t *colptr = nullptr, *rowidx = nullptr, *values = nullptr;
	if (dropest_count_matrix_csc(c.handle(), filtered ? 1 : 0, reads_output ? 1 : 0, &ncols, &nnz, &colptr, &rowidx, &values) != DROPEST_OK)
		throw std::runtime_error(dropest_last_error());
	return named_matrix(c, filtered, reference_row_order, ncols, nnz, colptr, rowidx, values);
}

// the entries of one column, from either form of the matrix
ResultsPrinter::ColumnSource::ColumnSource(const dropest_matrix_bytes &B) : ncols(B.ncols), nnz(B.nnz), colptr(B.colptr), bytes(&B) {
	// the device appends to the lists as it goes: ordered by position here, once
	row_listed.reserve(B.n_row_listed); value_listed.reserve(B.n_value_listed);
	for (uint64_t k = 0; k < B.n_row_listed; ++k) row_listed.emplace_back(B.row_listed_pos[k], B.row_listed_row[k]);
	for (uint64_t k = 0; k < B.n_value_listed; ++k) value_listed.emplace_back(B.value_listed_pos[k], B.value_listed_value[k]);
	std::sort(row_listed.begin(), row_listed.end());
	std::sort(value_listed.begin(), value_listed.end());
}

void ResultsPrinter::ColumnSource::column(uint64_t col, std::vector<std::pair<uint32_t, uint32_t>> &out) const {
	out.clear();
	const uint32_t b = colptr[col], e = colptr[col + 1];
	if (!bytes) {
		for (uint32_t k = b; k < e; ++k) out.emplace_back(rowidx[k], values[k]);
		return;
	}
	auto rl = std::lower_bound(row_listed.begin(), row_listed.end(), std::make_pair(b, 0u));
	auto vl = std::lower_bound(value_listed.begin(), value_listed.end(), std::make_pair(b, 0u));
	uint32_t row = 0xFFFFFFFFu;   // deltas count from row -1 (dropest_amd.h: dropest_matrix_bytes)
	for (uint32_t k = b; k < e; ++k) {
		const uint8_t d = bytes->row_delta[k], v = bytes->value[k];
		if (d == 255) {
			if (rl == row_listed.end() || rl->first != k) throw std::runtime_error("byte-form matrix: a listed row is missing from the list");
			row = (rl++)->second;
		} else row += d;
		uint32_t val = v;
		if (v == 255) {
			if (vl == value_listed.end() || vl->first != k) throw std::runtime_error("byte-form matrix: a listed value is missing from the list");
			val = (vl++)->second;
		}
		out.emplace_back(row, val);
	}
}

static std::string levels_code(const UMI::Mark::query_t &query) {   // inverse of UMI::Mark::get_by_code (UMI.cpp:112-154)
	std::string code;
	for (const UMI::Mark &m : query) {
		const bool e = m.check(UMI::Mark::HAS_EXONS), i = m.check(UMI::Mark::HAS_INTRONS), n = m.check(UMI::Mark::HAS_NOT_ANNOTATED);
		if (e && !i && !n) code += 'e'; else if (!e && i && !n) code += 'i'; else if (e && !i && n) code += 'E';
		else if (!e && i && n) code += 'I'; else if (e && i && !n) code += 'B'; else if (e && i && n) code += 'A';
		else throw std::runtime_error("Unexpected gene match level");
	}
	return code;
}

ResultsPrinter::SparseMatrix ResultsPrinter::get_count_matrix_filtered(const CellsDataContainer &c, const UMI::Mark::query_t &query,
                                                                       bool reference_row_order) const {
	uint64_t ncols = 0, nnz = 0;
	const uint32_t *colptr = nullptr, *rowidx = nullptr, *values = nullptr;
	if (dropest_count_matrix_csc_levels(c.handle(), levels_code(query).c_str(), reads_output ? 1 : 0, &ncols, &nnz, &colptr, &rowidx, &values) != DROPEST_OK)
		throw std::runtime_error(dropest_last_error());
	return named_matrix(c, true, reference_row_order, ncols, nnz, colptr, rowidx, values);
}

// the global matrix of a sharded container: columns = the cells' barcodes in the order of ONE container; rows named like on
// one GPU
ResultsPrinter::SparseMatrix ResultsPrinter::sharded_matrix(const CellsDataContainer &c, bool filtered, bool reference_row_order) const {
	uint64_t ncols = 0, nnz = 0;
	const uint64_t *colptr = nullptr, *bc = nullptr;
	const uint32_t *rowidx = nullptr, *values = nullptr;
	if (dropest_shard_matrix(c.shard0(), filtered ? 1 : 0, &ncols, &nnz, &colptr, &rowidx, &values, &bc) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
	std::vector<std::string> names;
	for (uint64_t j = 0; j < ncols; ++j) names.push_back(c.decode(bc[j]));
	if (nnz > 0xFFFFFFF0ull) throw std::runtime_error("count matrix with more than 2^32 non-zeros");
	std::vector<uint32_t> cp(ncols + 1, 0);
	for (uint64_t j = 0; j <= ncols && colptr; ++j) cp[j] = uint32_t(colptr[j]);
	return named_matrix(c, filtered, reference_row_order, ncols, nnz, cp.data(), rowidx, values, &names);
}

ResultsPrinter::SparseMatrix ResultsPrinter::named_matrix(const CellsDataContainer &c, bool filtered, bool reference_row_order, uint64_t ncols,
                                                          uint64_t nnz, const uint32_t *colptr, const uint32_t *rowidx, const uint32_t *values,
                                                          const std::vector<std::string> *col_names) const {
	return named_matrix(c, filtered, reference_row_order, ColumnSource(ncols, nnz, colptr, rowidx, values), col_names);
}

ResultsPrinter::SparseMatrix ResultsPrinter::named_matrix(const CellsDataContainer &c, bool filtered, bool reference_row_order, const ColumnSource &src,
                                                          const std::vector<std::string> *col_names) const {
	const uint64_t ncols = src.ncols, nnz = src.nnz;
	const uint32_t *colptr = src.colptr;
	std::vector<std::pair<uint32_t, uint32_t>> entries;   // (gene, value) of the column at hand
	SparseMatrix M;
	// column names: filtered cells in their order / real cells in cell-id order
	if (col_names) M.col_names = *col_names;
	else {      // (one fetch of the rows for either list: cell(id) per filtered cell was a device round trip each)
		const size_t n = c.total_cells_number();
		std::vector<dropest_cell_row> rows(n);
		if (n && dropest_cell_rows(c.handle(), 0, n, rows.data()) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
		if (filtered) { for (size_t id : c.filtered_cells()) M.col_names.push_back(c.decode(rows.at(id).barcode)); }
		else for (auto const &r : rows) if (r.is_real) M.col_names.push_back(c.decode(r.barcode));
	}
	M.colptr.assign(colptr, colptr + ncols + 1);
	M.values.resize(nnz);
	M.rowidx.resize(nnz);
	const auto &genes = c.gene_indexer().values();
	std::vector<uint32_t> row_of_gene(genes.size(), 0xFFFFFFFFu);
	auto row_id = [&](uint32_t g) {
		if (row_of_gene[g] == 0xFFFFFFFFu) { row_of_gene[g] = uint32_t(M.row_names.size()); M.row_names.push_back(genes[g]); }
		return row_of_gene[g];
	};
	if (!reference_row_order) {
		// rows = genes that occur, in gene-index order
		std::vector<char> seen(genes.size(), 0);
		for (uint64_t col = 0; col < ncols; ++col) {   // (gene ids land in the slots first, their rows replace them below)
			src.column(col, entries);
			uint32_t k = colptr[col];
			for (auto const &e : entries) { seen[e.first] = 1; M.rowidx[k] = e.first; M.values[k] = e.second; ++k; }
		}
		for (uint32_t g = 0; g < genes.size(); ++g) if (seen[g]) row_id(g);
		for (uint64_t k = 0; k < nnz; ++k) M.rowidx[k] = row_of_gene[M.rowidx[k]];
		return M;
	}
	// the reference numbers rows on first encounter while walking, per cell, an unordered_map<string,size_t> that was
	// filled in gene-index order (filtered: Cell.cpp:54-68) -- replayed here with the same container type; the raw
	// matrix walks the std::map directly (gene-index order, ResultsPrinter.cpp:376-387).  Inside a column the entries
	// are then sorted by row id (Eigen::setFromTriplets builds a CSC with ascending inner indices).
	// Three steps (round 6: the one loop over the columns -- two string-keyed hash maps per filtered column -- was most of what save_results
	// still took): (1) every column's entries in the order the reference meets them, columns side by side on host threads (the iteration order
	// of a column's own unordered_map hangs on nothing outside it); (2) rows numbered on first encounter, one walk in column order; (3) rows in
	// place of the genes and every column sorted by row, side by side again.  M.rowidx holds gene ids between (1) and (3).
	const size_t n_pieces = std::max<size_t>(1, std::min<size_t>(size_t(ncols), 256));
	auto piece_cols = [&](size_t piece, uint64_t &c0, uint64_t &c1) { c0 = ncols * piece / n_pieces; c1 = ncols * (piece + 1) / n_pieces; };
	Rds::parallel_pieces(n_pieces, 0, [&](size_t piece) {
		uint64_t c0, c1; piece_cols(piece, c0, c1);
		std::vector<std::pair<uint32_t, uint32_t>> of_col;
		for (uint64_t col = c0; col < c1; ++col) {
			src.column(col, of_col);
			uint32_t k = colptr[col];
			if (filtered) {
				std::unordered_map<std::string, size_t> per_gene;
				std::unordered_map<std::string, uint32_t> gene_of;
				for (auto const &e : of_col) { per_gene.emplace(genes[e.first], e.second); gene_of.emplace(genes[e.first], e.first); }
				for (auto const &kv : per_gene) { M.rowidx[k] = gene_of.at(kv.first); M.values[k] = uint32_t(kv.second); ++k; }
			} else {
				for (auto const &e : of_col) { M.rowidx[k] = e.first; M.values[k] = e.second; ++k; }
			}
		}
	});
	for (uint64_t k = 0; k < nnz; ++k) row_id(M.rowidx[k]);
	Rds::parallel_pieces(n_pieces, 0, [&](size_t piece) {
		uint64_t c0, c1; piece_cols(piece, c0, c1);
		std::vector<std::pair<uint32_t, uint32_t>> ent;
		for (uint64_t col = c0; col < c1; ++col) {
			ent.clear();
			for (uint32_t k = colptr[col]; k < colptr[col + 1]; ++k) ent.emplace_back(row_of_gene[M.rowidx[k]], M.values[k]);
			std::sort(ent.begin(), ent.end());
			for (uint32_t k = colptr[col], j = 0; k < colptr[col + 1]; ++k, ++j) { M.rowidx[k] = ent[j].first; M.values[k] = ent[j].second; }
		}
	});
	return M;
}

void ResultsPrinter::save_mtx(const CellsDataContainer &c, const std::string &base) const {   // ResultsPrinter.cpp:81-91
	const SparseMatrix M = get_count_matrix(c, true, true);
	std::ofstream mtx(base + ".mtx");
	if (!mtx) throw std::runtime_error("Can't open file: " + base + ".mtx");
	// Matrix::writeMM of a dgCMatrix: coordinate / real / general, 1-based, column-major
	mtx << "%%MatrixMarket matrix coordinate real general\n";
	mtx << M.row_names.size() << ' ' << M.col_names.size() << ' ' << M.values.size() << '\n';
	// the entry lines are formatted by host threads, ~2^18 entries per piece (whole columns), and written in order
	const size_t ncols = M.colptr.empty() ? 0 : M.colptr.size() - 1;
	std::vector<size_t> cut{0};
	for (size_t col = 0; col < ncols; ++col)
		if (M.colptr[col + 1] - M.colptr[cut.back()] >= (1u << 18) || col + 1 == ncols) cut.push_back(col + 1);
	std::vector<std::string> text(cut.size() - 1);
	Rds::parallel_pieces(text.size(), 0, [&](size_t piece) {
		std::string &t = text[piece];
		t.reserve(size_t(M.colptr[cut[piece + 1]] - M.colptr[cut[piece]]) * 16);
		char buf[48];
		for (size_t col = cut[piece]; col < cut[piece + 1]; ++col)
			for (uint32_t k = M.colptr[col]; k < M.colptr[col + 1]; ++k) {
				char *e = buf + sizeof(buf), *p = e;
				*--p = '\n';
				for (uint32_t v = M.values[k];; v /= 10) { *--p = char('0' + v % 10); if (v < 10) break; }
				*--p = ' ';
				for (size_t v = col + 1;; v /= 10) { *--p = char('0' + v % 10); if (v < 10) break; }
				*--p = ' ';
				for (uint32_t v = M.rowidx[k] + 1;; v /= 10) { *--p = char('0' + v % 10); if (v < 10) break; }
				t.append(p, size_t(e - p));
			}
	});
	for (auto const &t : text) mtx.write(t.data(), std::streamsize(t.size()));
	std::ofstream cells(base + ".cells.tsv"), genes(base + ".genes.tsv");
	for (auto const &s : M.col_names) cells << s << '\n';
	for (auto const &s : M.row_names) genes << s << '\n';
}

// The R list `d` of ResultsPrinter::save_results (ResultsPrinter.cpp:23-79), written natively (rds_writer.h) instead
// of through RInside::saveRDS.  Element names and R types follow the reference; the ORDER of entries inside vectors
// that the reference fills from unordered_map iteration (merge_targets, aligned_*_per_cell, saturation_info,
// reads_per_umi_per_cell) is not pinned by anything and is deterministic here: cell-id order, then gene index, then UMI.
Rds::ValuePtr ResultsPrinter::results_list(const CellsDataContainer &c) const {
	using namespace Rds;
	// DROPEST_RDS_TRACE=1: where save_results spends its time (stderr)
	const bool rds_trace = getenv("DROPEST_RDS_TRACE") != nullptr;
	auto rds_t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!rds_trace) return;
		const auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "[rds] %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - rds_t0).count());
		rds_t0 = now;
	};
	// both matrices are part of the list: cm_raw's emit + copy to the host start now, on the device's second stream, and
	// run under the cell rows and cm (a sharded container has assembled both already)
	if (!c.sharded()) {
		dropest_status st = walk_byte_form ? dropest_prefetch_raw_matrix_bytes(c.handle(), reads_output ? 1 : 0) : DROPEST_ERR_UNSUPPORTED;
		if (st == DROPEST_ERR_UNSUPPORTED) st = dropest_prefetch_raw_matrix(c.handle(), reads_output ? 1 : 0);
		if (st != DROPEST_OK) throw std::runtime_error(dropest_last_error());
	}
	const std::vector<Cell> real = c.real_cells();                      // cell-id order (sharded: of ONE container over the stream)
	const std::vector<size_t> filtered_at = c.filtered_positions(real);
	std::vector<std::string> real_names;
	for (const Cell &cell : real) real_names.push_back(cell.barcode());
	lap("real cells and their names");

	auto matrix = [&](bool filtered) {
		SparseMatrix M = get_count_matrix(c, filtered, true);
		return dgCMatrix(std::move(M.colptr), std::move(M.rowidx), std::move(M.values), M.row_names, M.col_names);   // (the slots are taken over: the writer's threads swap them from where they are)
	};
	// reads_per_chr_per_cells: as.data.frame of the cells x chromosomes IntegerMatrix (:144-172, :117-126)
	auto chr_frame = [&](Stats::CellChrStatType stat) {
		CellsDataContainer::names_t cells, chrs;
		CellsDataContainer::counts_t counts;
		c.get_stat_by_real_cells(stat, cells, chrs, counts);
		std::vector<std::vector<int32_t>> cols(chrs.size(), std::vector<int32_t>(cells.size()));
		for (size_t ci = 0; ci < cells.size(); ++ci)
			for (size_t k = 0; k < chrs.size(); ++k) cols[k][ci] = counts[ci * chrs.size() + k];
		return data_frame(chrs, cells, std::move(cols));
	};

	// one walk over the molecules of the real cells feeds mean_reads_per_umi, saturation_info, requested reads and
	// (for the filtered cells) reads_per_umi_per_cell
	std::vector<double> mean_rpu(real.size());
	std::vector<int32_t> sat_reads, req_umis(real.size()), req_reads(real.size());
	std::vector<std::string> sat_cbs, sat_umis;
	std::unordered_map<size_t, std::vector<Cell::MoleculeRow>> filtered_mols;   // by position in `real`
	std::vector<char> is_filtered(real.size(), 0);
	for (size_t k : filtered_at) is_filtered[k] = 1;
	// One fetch of the whole molecule table (ascending cell id, gene id, UMI code: the order the per-cell walks below see) instead of three
	// device round trips per real cell, and the barcodes / UMIs of saturation_info as packed codes that the writer's threads turn into
	// strings (round 6: with 500 real cells and 4e6 molecules the walks and their strings were 200 of the 245 ms of save_results).
	// Not with umi_correction_info (per-molecule quality sums), a sharded container, or strings with N in play (escaped codes).
	ValuePtr sat_cbs_packed, sat_umis_packed;
	bool bulk = !c.sharded() && !umi_correction_info;
	if (bulk) {
		uint64_t n_mol = 0;
		if (dropest_molecules(c.handle(), &n_mol, nullptr, nullptr, nullptr, nullptr, nullptr) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
		std::vector<uint32_t> m_cell(n_mol), m_gene(n_mol), m_reads(n_mol);
		std::vector<uint64_t> m_umi(n_mol);
		std::vector<uint8_t> m_mark(n_mol);
		if (n_mol && dropest_molecules(c.handle(), &n_mol, m_cell.data(), m_gene.data(), m_umi.data(), m_reads.data(), m_mark.data()) != DROPEST_OK)
			throw std::runtime_error(dropest_last_error());
		uint8_t wanted[8] = {0, 0, 0, 0, 0, 0, 0, 0};                         // mark bits -> does it match the container's levels
		for (int b = 0; b < 8; ++b) {
			UMI::Mark m;
			if (b & 1) m.add(UMI::Mark::HAS_NOT_ANNOTATED);
			if (b & 2) m.add(UMI::Mark::HAS_EXONS);
			if (b & 4) m.add(UMI::Mark::HAS_INTRONS);
			wanted[b] = m.match(c.gene_match_level()) ? 1 : 0;
		}
		// every real cell's rows [lo, hi) of the table, its sums, and its place in the three saturation vectors
		std::vector<size_t> lo(real.size()), hi(real.size()), sat_at(real.size() + 1, 0);
		Rds::parallel_pieces(real.size(), 0, [&](size_t k) {
			const uint32_t id = uint32_t(real[k].id());
			lo[k] = size_t(std::lower_bound(m_cell.begin(), m_cell.end(), id) - m_cell.begin());
			hi[k] = size_t(std::upper_bound(m_cell.begin() + long(lo[k]), m_cell.end(), id) - m_cell.begin());
			double reads = 0; size_t rr = 0, ns = 0;
			for (size_t i = lo[k]; i < hi[k]; ++i) { reads += double(m_reads[i]); if (wanted[m_mark[i] & 7]) { rr += m_reads[i]; ++ns; } }
			mean_rpu[k] = reads / double(hi[k] - lo[k]);
			req_reads[k] = int32_t(rr);
			sat_at[k + 1] = ns;
		});
		for (size_t k = 0; k < real.size(); ++k) { sat_at[k + 1] += sat_at[k]; req_umis[k] = int32_t(real[k].requested_umis_num()); }
		std::vector<uint64_t> cb_codes(sat_at.back()), umi_codes(sat_at.back());
		sat_reads.resize(sat_at.back());
		std::atomic<bool> escaped{false};
		Rds::parallel_pieces(real.size(), 0, [&](size_t k) {
			size_t at = sat_at[k];
			const uint64_t cbc = real[k].barcode_code();
			if (cbc >> 63) escaped.store(true, std::memory_order_relaxed);
			for (size_t i = lo[k]; i < hi[k]; ++i)
				if (wanted[m_mark[i] & 7]) { sat_reads[at] = int32_t(m_reads[i]); cb_codes[at] = cbc; umi_codes[at] = m_umi[i]; if (m_umi[i] >> 63) escaped.store(true, std::memory_order_relaxed); ++at; }
		});
		if (escaped.load()) {      // strings with N: as strings (the side table lives in the container)
			sat_cbs.resize(cb_codes.size()); sat_umis.resize(umi_codes.size());
			Rds::parallel_pieces((cb_codes.size() + 65535) / 65536, 0, [&](size_t piece) {
				for (size_t i = piece * 65536; i < std::min(cb_codes.size(), (piece + 1) * 65536); ++i) { sat_cbs[i] = c.decode(cb_codes[i]); sat_umis[i] = c.decode(umi_codes[i]); }
			});
		} else { sat_cbs_packed = strings_from_packed(std::move(cb_codes)); sat_umis_packed = strings_from_packed(std::move(umi_codes)); }
	}
	lap("molecule table and saturation vectors (bulk)");
	for (size_t k = 0; !bulk && k < real.size(); ++k) {
		const Cell &cell = real[k];
		auto mols = cell.molecules();
		double reads = 0;
		size_t rr = 0;
		for (auto const &m : mols) {
			reads += double(m.read_count);
			if (m.mark.match(c.gene_match_level())) {
				sat_reads.push_back(int32_t(m.read_count)); sat_cbs.push_back(real_names[k]); sat_umis.push_back(m.umi);
				rr += m.read_count;
			}
		}
		mean_rpu[k] = reads / double(mols.size());                      // :225-249 (0/0 = NaN for a cell without UMIs, as in R)
		req_umis[k] = int32_t(cell.requested_umis_num());               // :398-431
		req_reads[k] = int32_t(rr);
		if (umi_correction_info && is_filtered[k]) filtered_mols.emplace(k, std::move(mols));
	}

	std::vector<std::pair<std::string, ValuePtr>> merged;               // :313-332: list(source barcode = target barcode)
	if (c.sharded()) {   // (sources in barcode order: their cell ids are not part of the global table)
		for (auto const &pr : c.merged_barcodes()) merged.emplace_back(pr.first, strings({pr.second}));
	} else {
		const size_t n_cells = c.total_cells_number();
		std::vector<dropest_cell_row> rows(n_cells);
		if (n_cells && dropest_cell_rows(c.handle(), 0, n_cells, rows.data()) != DROPEST_OK) throw std::runtime_error(dropest_last_error());
		const auto &mt = c.merge_targets();
		for (size_t i = 0; i < mt.size(); ++i)
			if (mt[i] != i) merged.emplace_back(c.decode(rows[i].barcode), strings({c.decode(rows[mt[i]].barcode)}));
	}
	std::vector<int32_t> aligned_reads(real.size()), aligned_umis(real.size());
	for (size_t k = 0; k < real.size(); ++k) { aligned_reads[k] = real[k].stat(Stats::TOTAL_READS_PER_CB); aligned_umis[k] = real[k].stat(Stats::TOTAL_UMIS_PER_CB); }

	lap("merge targets, per-cell vectors");
	ValuePtr v_cm = matrix(true);
	lap("cm named");
	ValuePtr v_raw = matrix(false);
	lap("cm_raw named");
	std::vector<std::pair<std::string, ValuePtr>> d = {
		{"cm", v_cm},
		{"cm_raw", v_raw},
		{"reads_per_chr_per_cells", named_list({{"Exon", chr_frame(Stats::EXON_READS_PER_CHR_PER_CELL)},
		                                        {"Intron", chr_frame(Stats::INTRON_READS_PER_CHR_PER_CELL)},
		                                        {"Intergenic", chr_frame(Stats::INTERGENIC_READS_PER_CHR_PER_CELL)}})},
		{"mean_reads_per_umi", with_names(reals(std::move(mean_rpu)), real_names)},
		{"saturation_info", named_list({{"reads", integers(std::move(sat_reads))}, {"cbs", sat_cbs_packed ? sat_cbs_packed : strings(std::move(sat_cbs))},
		                                {"umis", sat_umis_packed ? sat_umis_packed : strings(std::move(sat_umis))}})},
		{"merge_targets", named_list(std::move(merged))},
		{"aligned_reads_per_cell", with_names(integers(std::move(aligned_reads)), real_names)},
		{"aligned_umis_per_cell", with_names(integers(std::move(aligned_umis)), real_names)},
		{"requested_umis_per_cb", with_names(integers(std::move(req_umis)), real_names)},
		{"requested_reads_per_cb", with_names(integers(std::move(req_reads)), real_names)},
	};
	if (umi_correction_info) {
		// get_reads_per_umi_per_cell (:251-311): filtered cells, requested UMIs; per UMI list(reads, mean quality); the
		// mean quality is numeric(0) when the reads carried no UMI quality (as in the reference).
		StringIndexer cell_ix, gene_ix;
		std::vector<int32_t> cell_indexes, gene_indexes;
		std::vector<ValuePtr> per_gene;
		for (size_t k : filtered_at) {
			auto it = filtered_mols.find(k);
			if (it == filtered_mols.end()) continue;
			const int32_t ci = int32_t(cell_ix.add(real_names[k]));
			std::vector<ValuePtr> umis; std::vector<std::string> umi_names;
			std::string cur;
			auto close = [&]() {
				if (umis.empty()) return;
				per_gene.push_back(with_names(list(std::move(umis)), std::move(umi_names)));
				cell_indexes.push_back(ci); gene_indexes.push_back(int32_t(gene_ix.add(cur)));
				umis.clear(); umi_names.clear();
			};
			for (auto const &m : it->second) {
				if (m.gene != cur) { close(); cur = m.gene; }
				if (!m.mark.match(c.gene_match_level())) continue;
				umis.push_back(list({integers({int32_t(m.read_count)}), reals(m.mean_quality())}));
				umi_names.push_back(m.umi);
			}
			close();
		}
		d.emplace_back("reads_per_umi_per_cell", named_list({{"cells", strings(cell_ix.values())}, {"genes", strings(gene_ix.values())},
		                                                     {"cell_indexes", integers(std::move(cell_indexes))},
		                                                     {"gene_indexes", integers(std::move(gene_indexes))},
		                                                     {"reads_per_umi", list(std::move(per_gene))}}));
	}
	lap("chromosome frames and the list");
	return named_list(std::move(d));
}

void ResultsPrinter::save_intron_exon_matrices(const CellsDataContainer &c, const std::string &filename) const {   // ResultsPrinter.cpp:455-474
	using namespace Rds;
	auto m = [&](const char *code) {
		SparseMatrix M = get_count_matrix_filtered(c, UMI::Mark::get_by_code(code), true);
		return dgCMatrix(std::move(M.colptr), std::move(M.rowidx), std::move(M.values), M.row_names, M.col_names);
	};
	std::string base = filename;
	const size_t dot = filename.find_last_of('.');
	if (dot != std::string::npos && filename.substr(dot + 1) == "rds") base = filename.substr(0, dot);
	Rds::save(named_list({{"exon", m("e")}, {"intron", m("i")}, {"spanning", m("BA")}}), base + ".matrices.rds");
}

void ResultsPrinter::save_results(const CellsDataContainer &c, const std::string &filename) const {   // ResultsPrinter.cpp:23-79
	std::string base = filename;
	const size_t dot = filename.find_last_of('.');
	if (dot != std::string::npos && filename.substr(dot + 1) == "rds") base = filename.substr(0, dot);
	const Rds::ValuePtr v = results_list(c);
	const auto t_save = std::chrono::steady_clock::now();
	Rds::save(v, base + ".rds");                                        // save_rds (:442-452)
	if (getenv("DROPEST_RDS_TRACE")) std::fprintf(stderr, "[rds] serialise + deflate + write %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_save).count());
	if (write_matrix) save_mtx(c, base);
}

}  // namespace Estimation
