// dropest_oracle.cpp -- CPU ORACLE for the dropEst Estimation hot path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing under dropest_amd/ may include, link, import or
// execute this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
// and there only as the checker / timed CPU baseline -- never as the product path.
//
// What it is: a single-threaded C++ restatement of the reference algorithm (kharchenkolab/dropEst
// v0.8.6, mounted read-only at /root/reference while authoring), written from the reference's
// behaviour, using the same libstdc++ container kinds (std::unordered_map<std::string,...>,
// std::map, std::unordered_set) and the same std::sort calls on the same sequences wherever the
// reference's results depend on hash-iteration order or on the (unstable) sort, so that those
// implementation-defined orders are inherited instead of re-derived.
//
// Pinning: the reference itself is UNBUILDABLE in this image (every translation unit on the path
// includes Boost and/or RInside headers, which are absent, and stand-in headers are not allowed),
// so this oracle is pinned on the reference's own known-answer unit tests instead:
// tests/test_oracle_reference_kat.py replays every assertion of Tests/TestEstimation.cpp and
// Tests/TestTools.cpp that touches the path (see the test file for the line-by-line list).
//
// Reference citations are `path:line` under /root/reference.

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace orc {

// ----------------------------------------------------------------------------------------------
// Small tools (Tools/UtilFunctions.cpp)
// ----------------------------------------------------------------------------------------------

// Banded Levenshtein distance with 'N' as a wildcard; follows Tools/UtilFunctions.cpp:32-65
// (column DP, band [s2-max_ed, s2+max_ed], early exit when the best reachable value exceeds max_ed).
static unsigned edit_distance(const char *a, const char *b, bool skip_n = true, unsigned max_ed = 10000) {
	const int la = int(std::strlen(a)), lb = int(std::strlen(b));
	std::vector<int> col(size_t(la) + 1);
	for (int i = 0; i <= la; ++i) col[i] = i;
	for (int j = 1; j <= lb; ++j) {
		const int lo = std::max(0, j - int(max_ed));
		const int hi = std::min(la, j + int(max_ed));
		int diag = col[lo];
		col[lo] = j;
		int best = j;
		for (int i = lo + 1; i <= hi; ++i) {
			const int up = col[i];
			const bool same = (a[i - 1] == b[j - 1]) || (skip_n && (a[i - 1] == 'N' || b[j - 1] == 'N'));
			const int v = std::min(std::min(col[i] + 1, col[i - 1] + 1), diag + int(!same));
			best = std::min(best, v + std::abs(i - j));
			col[i] = v;
			diag = up;
		}
		if (best > int(max_ed)) return unsigned(best);
	}
	return unsigned(col[la]);
}

// Tools/UtilFunctions.cpp:67-82
static unsigned hamming_distance(const std::string &a, const std::string &b, bool skip_n = true) {
	if (a.size() != b.size()) throw std::runtime_error("Strings should have equal length");
	unsigned d = 0;
	for (size_t i = 0; i < a.size(); ++i)
		if (a[i] != b[i] && !(skip_n && (a[i] == 'N' || b[i] == 'N'))) ++d;
	return d;
}

// Tools/UtilFunctions.cpp:13-30
static double fpow(double base, long exp) {
	if (exp < 0) throw std::runtime_error("fpow: negative exponent (the reference does not terminate here: the collisions adjustment diverged)");
	if (exp == 1) return base;
	double r = 1;
	while (exp) {
		if (exp & 1) r *= base;
		exp >>= 1;
		base *= base;
	}
	return r;
}

// Tools/UtilFunctions.cpp:97-116 (only A/C/G/T/N are defined in the reference; others are UB there)
static std::string reverse_complement(const std::string &s) {
	std::string r(s);
	for (size_t i = 0; i < s.size(); ++i) {
		char c = s[s.size() - 1 - i], o;
		switch (c) {
			case 'A': o = 'T'; break;
			case 'T': o = 'A'; break;
			case 'G': o = 'C'; break;
			case 'C': o = 'G'; break;
			case 'N': o = 'N'; break;
			default: throw std::runtime_error("reverse_complement: unexpected base");
		}
		r[i] = o;
	}
	return r;
}

// Tools/ReadParameters.cpp:42-56  ("<id>!<CB>#<UMI>")
static bool parse_encoded_id(const std::string &id, std::string &cb, std::string &umi) {
	size_t u = id.rfind('#');
	if (u == std::string::npos) return false;
	size_t c = id.rfind('!', u);
	if (c == std::string::npos) return false;
	cb = id.substr(c + 1, u - c - 1);
	umi = id.substr(u + 1);
	return !cb.empty() && !umi.empty();  // ReadParameters ctor throws on empty parts (:14-15)
}

// Tools/CollisionsAdjuster.cpp:12-49
struct CollisionsAdjuster {
	std::vector<size_t> adjusted;
	std::vector<double> p, neg_prod;
	double sum_collisions = 0;
	size_t last_total = 0;
	void init(const std::vector<double> &probs, size_t max_expr) {
		sum_collisions = 0; last_total = 0; p = probs; neg_prod.assign(probs.size(), 1.0);
		extend(max_expr);
	}
	void extend(size_t max_expr) {
		for (size_t s = adjusted.size() + 1; s <= max_expr; ++s) {
			const size_t total = s + size_t(sum_collisions);
			double new_prob = 0;
			for (size_t i = 0; i < p.size(); ++i) {
				neg_prod[i] *= fpow(1 - p[i], long(total - last_total));
				new_prob += p[i] * (1 - neg_prod[i]);
			}
			last_total = total;
			sum_collisions += 1.0 / (1.0 - new_prob) - 1.0;
			adjusted.push_back(size_t(std::lround(double(s) + sum_collisions)));
		}
	}
	size_t estimate(size_t expr) {
		if (expr > adjusted.size()) extend(expr);
		return adjusted.at(expr - 1);
	}
};

// ----------------------------------------------------------------------------------------------
// Data model (Estimation/{StringIndexer,UMI,Gene,Cell,Stats}.*)
// ----------------------------------------------------------------------------------------------

enum : uint8_t { MARK_NOT_ANNOTATED = 1, MARK_EXON = 2, MARK_INTRON = 4 };  // UMI.h:16-22
enum { CHR_EXON = 0, CHR_INTRON = 1, CHR_INTERGENIC = 2, CHR_KINDS = 3 };   // Stats.h:26-32

// Estimation/StringIndexer.cpp:10-23 -- first-seen dense ids
struct Indexer {
	std::vector<std::string> values;
	std::unordered_map<std::string, size_t> index;
	size_t add(const std::string &v) {
		auto it = index.emplace(v, index.size());
		if (it.second) values.push_back(v);
		return it.first->second;
	}
	size_t get(const std::string &v) const { return index.at(v); }
};

struct Molecule {           // Estimation/UMI.h:46-64
	size_t reads = 0;
	uint8_t mark = 0;
	std::vector<unsigned> qual_sum;
};
typedef std::map<size_t, Molecule> umis_t;   // Gene.h:19, key = umi index
typedef std::map<size_t, umis_t> genes_t;    // Cell.h:19, key = gene index

struct Cell {
	std::string barcode;
	bool merged = false, excluded = false;
	size_t req_genes = 0, req_umis = 0;
	genes_t genes;
	int total_reads = 0, total_umis = 0;                  // Stats::_stat_data
	std::unordered_map<size_t, int> chr[CHR_KINDS];       // Stats::_chromosome_stat_data
};

static bool mark_matches(uint8_t mark, const std::vector<uint8_t> &query) {  // UMI.cpp:76-85 (equality!)
	for (uint8_t q : query) if (mark == q) return true;
	return false;
}

static std::vector<uint8_t> marks_by_code(const std::string &code) {  // UMI.cpp:112-154
	std::vector<uint8_t> out;
	for (char c : code) {
		switch (c) {
			case 'e': out.push_back(MARK_EXON); break;
			case 'i': out.push_back(MARK_INTRON); break;
			case 'E': out.push_back(MARK_EXON | MARK_NOT_ANNOTATED); break;
			case 'I': out.push_back(MARK_INTRON | MARK_NOT_ANNOTATED); break;
			case 'B': out.push_back(MARK_EXON | MARK_INTRON); break;
			case 'A': out.push_back(MARK_EXON | MARK_INTRON | MARK_NOT_ANNOTATED); break;
			default: throw std::runtime_error(std::string("Unexpected gene match levels: ") + c);
		}
	}
	return out;
}

// ----------------------------------------------------------------------------------------------
// Whitelist ("real barcodes") parsing -- Estimation/Merge/BarcodesParsing/*
// ----------------------------------------------------------------------------------------------

struct PartDist { size_t index; long value; };   // Tools/IndexedValue.h
struct ComboDist { std::vector<size_t> part_inds; unsigned ed; };

struct Whitelist {
	enum Kind { INDROP = 0, CONST_LEN = 1 };
	Kind kind = INDROP;
	std::vector<std::vector<std::string>> parts;
	std::vector<size_t> part_lengths;   // const-length parser
	size_t total_length = 0;
	size_t bc2_length = 0;              // indrop parser
	static const int MAX_REAL_MERGE_EDIT_DISTANCE = 5;   // BarcodesParser.h:57

	// BarcodesParser.cpp:117-144: one text line -> reverse-complemented tokens
	static bool read_line(std::istream &in, std::vector<std::string> &out, bool equal_len) {
		std::string line;
		if (!std::getline(in, line)) return false;
		std::istringstream ss(line);
		size_t len0 = 0;
		std::string tok;
		while (ss >> tok) {
			if (len0 == 0) len0 = tok.size();
			else if (equal_len && len0 != tok.size())
				throw std::runtime_error("All barcodes in one line must have the same length");
			out.push_back(reverse_complement(tok));
		}
		return true;
	}

	void load(Kind k, const std::string &file) {
		kind = k;
		std::ifstream f(file);
		if (f.fail()) throw std::runtime_error("Can't open barcodes file: '" + file + "'");
		parts.clear();
		if (k == INDROP) {   // InDropBarcodesParser.cpp:15-30 : exactly two lines
			parts.resize(2);
			for (int i = 0; i < 2; ++i)
				if (!read_line(f, parts[i], false) || parts[i].empty())
					throw std::runtime_error("File with barcodes (" + file + ") has wrong format");
		} else {             // ConstLengthBarcodesParser.cpp:50-68 : one part per line
			std::vector<std::string> part;
			while (read_line(f, part, true)) {
				if (part.empty()) throw std::runtime_error("File with barcodes (" + file + ") has wrong format");
				parts.push_back(part);
				part.clear();
			}
		}
		finish_init();
	}
	void finish_init() {
		if (parts.empty()) throw std::runtime_error("ERROR: empty barcodes list");   // BarcodesParser.cpp:92-103
		for (auto &p : parts) if (p.empty()) throw std::runtime_error("ERROR: empty barcodes list");
		part_lengths.clear(); total_length = 0;
		for (auto &p : parts) { part_lengths.push_back(p[0].size()); total_length += p[0].size(); }
		bc2_length = parts.size() > 1 ? parts[1][0].size() : 0;
	}

	std::vector<std::string> split(const std::string &cb) const {
		std::vector<std::string> r;
		if (kind == INDROP) {   // InDropBarcodesParser.cpp:32-39
			r.push_back(cb.substr(0, cb.size() - bc2_length));
			r.push_back(cb.substr(cb.size() - bc2_length));
		} else {                // ConstLengthBarcodesParser.cpp:33-48
			if (cb.size() != total_length)
				throw std::runtime_error("Barcode '" + cb + "' has wrong length");
			size_t pos = 0;
			for (size_t l : part_lengths) { r.push_back(cb.substr(pos, l)); pos += l; }
		}
		return r;
	}

	// BarcodesParser.cpp:21-39 : per part, (index, edit distance) sorted by distance with std::sort
	std::vector<std::vector<PartDist>> distances(const std::string &cb) const {
		std::vector<std::string> pieces = split(cb);
		std::vector<std::vector<PartDist>> res(parts.size());
		for (size_t p = 0; p < parts.size(); ++p) {
			for (size_t i = 0; i < parts[p].size(); ++i)
				res[p].push_back(PartDist{i, long(edit_distance(pieces[p].c_str(), parts[p][i].c_str()))});
			std::sort(res[p].begin(), res[p].end(),
			          [](const PartDist &x, const PartDist &y) { return x.value < y.value; });
		}
		return res;
	}

	// BarcodesParser.cpp:52-74 : depth-first enumeration of part combinations, total distance <= 5
	void enumerate(const std::vector<std::vector<PartDist>> &d, size_t level, unsigned ed,
	               std::vector<size_t> &inds, std::vector<ComboDist> &out) const {
		if (level == d.size()) { out.push_back(ComboDist{inds, ed}); return; }
		inds.push_back(0);
		for (const PartDist &pd : d[level]) {
			unsigned cur = ed + unsigned(pd.value);
			if (cur > unsigned(MAX_REAL_MERGE_EDIT_DISTANCE)) break;  // sorted: nothing further qualifies
			inds.back() = pd.index;
			enumerate(d, level + 1, cur, inds, out);
		}
		inds.pop_back();
	}
	std::vector<ComboDist> real_neighbours(const std::string &cb) const {   // BarcodesParser.cpp:41-50
		std::vector<ComboDist> out;
		auto d = distances(cb);
		if (d.empty()) return out;
		std::vector<size_t> inds;
		enumerate(d, 0, 0, inds, out);
		return out;
	}
	std::string barcode_of(const std::vector<size_t> &inds) const {        // BarcodesParser.cpp:76-87
		std::string r;
		for (size_t i = 0; i < inds.size(); ++i) r += parts[i].at(inds[i]);
		return r;
	}
};

// ----------------------------------------------------------------------------------------------
// Container (Estimation/CellsDataContainer.*) + merge strategies
// ----------------------------------------------------------------------------------------------

struct Config {
	int merge_kind = 0;          // 0 = none (DummyMergeStrategy), 1 = RealBarcodes, 2 = Simple (-m without a whitelist),
	                             // 3 = PoissonRealBarcodes (-M with a whitelist), 4 = PoissonSimple (-M without), 5 = MergeAll (merge_type all)
	double max_merge_prob = 1e-4, max_real_merge_prob = 1e-7;   // PreciseMerge.* (MergeStrategyFactory.cpp:52-55)
	int barcodes_kind = 0;       // Whitelist::Kind
	std::string barcodes_file;
	size_t min_genes_before = 10, min_genes_after = 10;   // MergeStrategyFactory.cpp:26-58 defaults
	double min_merge_fraction = 0.2;
	int max_cb_merge_ed = 0;     // ignored by RealBarcodes (RealBarcodesMergeStrategy.cpp:111-114); used by Simple
	int umi_merge_kind = 0;      // 0 = MergeUMIsStrategySimple (N fix), 1 = Directional
	unsigned max_umi_merge_ed = 1;
	double umi_mult = 2.0;       // MergeUMIsStrategyDirectional default multiplier
	std::string match_levels = "eEBA";
	int max_cells = -1;
};

struct Container {
	Config cfg;
	std::vector<uint8_t> query;
	std::vector<Cell> cells;
	std::unordered_map<std::string, size_t> cell_by_cb;
	std::vector<size_t> filtered, merge_targets;
	bool initialized = false;
	size_t intergenic_reads = 0, exon_reads = 0, intron_reads = 0, not_annotated_reads = 0, real_cells = 0;
	Indexer umi_ix, gene_ix;
	// Stats statics (Stats.cpp:5-7); one container per process in the reference, per-container here
	std::unordered_set<size_t> presented_chr[CHR_KINDS];
	Indexer chr_ix;
	Whitelist wl;

	explicit Container(const Config &c) : cfg(c), query(marks_by_code(c.match_levels)) {
		// MergeStrategyAbstract.cpp:8-11 : after >= before
		cfg.min_genes_after = std::max(cfg.min_genes_after, cfg.min_genes_before);
		if (cfg.merge_kind == 1 || cfg.merge_kind == 3) wl.load(Whitelist::Kind(cfg.barcodes_kind), cfg.barcodes_file);
		if (cfg.umi_merge_kind == 0) srand(42);   // MergeUMIsStrategySimple.cpp:15-19
	}

	size_t umis_number(const Cell &c) const { return size_t(c.total_umis); }   // Cell.cpp:110-113
	bool is_real(const Cell &c) const {                                        // Cell.cpp:125-128
		return !c.excluded && !c.merged && c.genes.size() >= cfg.min_genes_before;
	}
	void chr_inc(Cell &c, int kind, const std::string &chr) {                  // Stats.cpp:22-27
		size_t id = chr_ix.add(chr);
		presented_chr[kind].insert(id);
		c.chr[kind][id]++;
	}

	// CellsDataContainer.cpp:59-88, :356-364, :309-327; Gene.cpp:17-24; UMI.cpp:21-34
	void add_record(const std::string &cb, const std::string &umi, const std::string &umi_qual,
	                const std::string &gene, const std::string &chr, uint8_t mark) {
		if (initialized) throw std::runtime_error("Container is already initialized");
		auto res = cell_by_cb.emplace(cb, cell_by_cb.size());
		if (res.second) { cells.emplace_back(); cells.back().barcode = cb; }
		const size_t cid = res.first->second;
		if (gene.empty()) {
			chr_inc(cells[cid], CHR_INTERGENIC, chr);
			++intergenic_reads;
			return;
		}
		const size_t g = gene_ix.add(gene);
		umis_t &gu = cells[cid].genes[g];
		const size_t u = umi_ix.add(umi);
		auto ins = gu.emplace(u, Molecule());
		Molecule &m = ins.first->second;
		if (ins.second) m.qual_sum.assign(umi_qual.size(), 0);
		m.reads++;
		m.mark |= mark;
		if (umi_qual.size() != m.qual_sum.size())
			throw std::runtime_error("Wrong quality length: " + std::to_string(umi_qual.size()) +
			                         ", expected: " + std::to_string(m.qual_sum.size()));
		for (size_t i = 0; i < m.qual_sum.size(); ++i) m.qual_sum[i] += unsigned(umi_qual[i]);
		if (ins.second) cells[cid].total_umis++;

		Cell &c = cells[cid];
		c.total_reads++;
		if (mark & MARK_EXON) { chr_inc(c, CHR_EXON, chr); ++exon_reads; }
		if (mark & MARK_INTRON) { chr_inc(c, CHR_INTRON, chr); ++intron_reads; }
		if (mark & MARK_NOT_ANNOTATED) ++not_annotated_reads;
	}

	static size_t requested_in_gene(const umis_t &umis, const std::vector<uint8_t> &q, bool reads) {  // Gene.cpp:60-79
		size_t n = 0;
		for (auto const &u : umis) if (mark_matches(u.second.mark, q)) n += reads ? u.second.reads : 1;
		return n;
	}

	// CellsDataContainer.cpp:111-125, :250-276, :329-344; Cell.cpp:130-143
	size_t update_cell_sizes(size_t genes_threshold, int cell_threshold) {
		real_cells = 0;
		for (Cell &c : cells) {
			c.req_genes = c.req_umis = 0;
			for (auto const &g : c.genes) {
				size_t n = requested_in_gene(g.second, query, false);
				if (n == 0) continue;
				c.req_umis += n;
				c.req_genes++;
			}
			if (is_real(c)) real_cells++;
		}
		filtered.clear();
		for (size_t i = 0; i < cells.size(); ++i)
			if (is_real(cells[i]) && cells[i].req_genes >= genes_threshold) filtered.push_back(i);
		std::sort(filtered.begin(), filtered.end(), [this](size_t a, size_t b) {
			const Cell &x = cells[a], &y = cells[b];
			if (x.req_genes != y.req_genes) return x.req_genes < y.req_genes;
			if (x.req_umis != y.req_umis) return x.req_umis < y.req_umis;
			if (umis_number(x) != umis_number(y)) return umis_number(x) < umis_number(y);
			return x.barcode < y.barcode;
		});
		size_t n = filtered.size();
		if (cell_threshold > 0 && size_t(cell_threshold) < filtered.size())
			filtered.erase(filtered.begin(), filtered.end() - unsigned(cell_threshold));
		return n;
	}

	void set_initialized() {   // CellsDataContainer.cpp:163-175
		if (initialized) throw std::runtime_error("Container is already initialized");
		update_cell_sizes(0, -1);
		initialized = true;
	}

	// CellsDataContainer.cpp:90-104; Gene.cpp:26-36; UMI.cpp:15-19; Stats.cpp:29-43
	void merge_cells(size_t src_id, size_t tgt_id) {
		Cell &src = cells.at(src_id), &tgt = cells.at(tgt_id);
		for (auto const &g : src.genes) {
			umis_t &tu = tgt.genes[g.first];
			for (auto const &u : g.second) {
				auto ins = tu.insert(u);
				if (ins.second) continue;
				ins.first->second.reads += u.second.reads;
				ins.first->second.mark |= u.second.mark;
			}
		}
		tgt.total_reads += src.total_reads;
		tgt.total_umis += src.total_umis;    // quirk: adds even when UMIs coincide
		for (int k = 0; k < CHR_KINDS; ++k)
			for (auto const &kv : src.chr[k]) tgt.chr[k][kv.first] += kv.second;
		src.merged = true;
	}

	// MergeStrategyBase.cpp:100-147 : |{(gene,umi)} of a  intersect  {(gene,umi)} of b|
	static size_t umig_intersection(const Cell &a, const Cell &b) {
		size_t n = 0;
		auto ga = a.genes.begin(), gb = b.genes.begin();
		while (ga != a.genes.end() && gb != b.genes.end()) {
			if (ga->first < gb->first) { ++ga; continue; }
			if (ga->first > gb->first) { ++gb; continue; }
			auto ua = ga->second.begin(), ub = gb->second.begin();
			while (ua != ga->second.end() && ub != gb->second.end()) {
				if (ua->first < ub->first) { ++ua; continue; }
				if (ua->first > ub->first) { ++ub; continue; }
				++n; ++ua; ++ub;
			}
			++ga; ++gb;
		}
		return n;
	}

	// RealBarcodesMergeStrategy.cpp:63-109
	std::vector<size_t> real_neighbour_cells(size_t base) const {
		std::vector<ComboDist> dists = wl.real_neighbours(cells.at(base).barcode);
		std::vector<size_t> out;
		if (dists.empty()) return out;
		std::sort(dists.begin(), dists.end(), [](const ComboDist &x, const ComboDist &y) { return x.ed < y.ed; });
		unsigned max_dist = dists.front().ed;   // get_max_merge_dist(min) == min (:111-114)
		if (cfg.merge_kind == 3) max_dist = max_dist == 0 ? 2 : max_dist + 1;   // PoissonRealBarcodesMergeStrategy.cpp:21-24
		for (const ComboDist &cd : dists) {
			if (cd.ed > max_dist && !out.empty()) break;
			auto it = cell_by_cb.find(wl.barcode_of(cd.part_inds));
			if (it != cell_by_cb.end()) {
				const Cell &c = cells[it->second];
				if (c.genes.size() >= cfg.min_genes_before && umis_number(c) >= umis_number(cells[base]))
					out.push_back(it->second);
			}
			max_dist = std::max(max_dist, cd.ed);
		}
		return out;
	}

	// RealBarcodesMergeStrategy.cpp:22-61
	long real_merge_target(size_t base) const {
		std::vector<size_t> nb = real_neighbour_cells(base);
		if (nb.empty()) return -1;
		if (nb[0] == base) return long(base);
		double best_frac = 0;
		size_t best = nb[0];
		for (size_t n : nb) {
			size_t inter = umig_intersection(cells[base], cells[n]);
			double frac = 0.5 * inter * (1. / umis_number(cells[base]) + 1. / umis_number(cells[n]));
			if (best_frac < frac) { best_frac = frac; best = n; }
		}
		if (best_frac < cfg.min_merge_fraction) return -1;
		return long(best);
	}

	// SimpleMergeStrategy (Estimation/Merge/SimpleMergeStrategy.cpp).  The reference keys its inverted index with
	// Tools::PairHash, whose seed is an uninitialised local (UtilFunctions.h:36-41); the index is only ever probed with
	// emplace / at, never iterated, so any consistent hash gives the same results -- std::map is used here.  What DOES
	// shape the result is the iteration order of the two unordered containers keyed by cell id (identity hash), kept.
	std::map<std::pair<size_t, size_t>, std::unordered_set<size_t>> cell_ids_by_umig;
	void simple_init() {                                       // SimpleMergeStrategy::init (:88-102)
		cell_ids_by_umig.clear();
		for (size_t cell_id : filtered)
			for (auto const &g : cells[cell_id].genes)
				for (auto const &u : g.second)
					cell_ids_by_umig[std::make_pair(u.first, g.first)].emplace(cell_id);
	}
	std::unordered_map<size_t, size_t> cells_with_common_umigs(size_t base) const {   // SimpleMergeStrategy.cpp:16-40
		std::unordered_map<size_t, size_t> common;
		for (auto const &g : cells[base].genes)
			for (auto const &u : g.second)
				for (size_t other : cell_ids_by_umig.at(std::make_pair(u.first, g.first))) {
					if (other == base) continue;
					if (cells[other].genes.size() >= cells[base].genes.size()) common[other]++;
				}
		return common;
	}
	long simple_merge_target(size_t base) const {              // get_merge_target (:47-86)
		const double EPS = 0.00001;
		const std::unordered_map<size_t, size_t> common = cells_with_common_umigs(base);
		long top = -1, top_genes = -1;
		double top_frac = -1;
		for (auto const &c : common) {
			const size_t ind = c.first;
			const double frac = 0.5 * c.second * (1. / umis_number(cells[base]) + 1. / umis_number(cells[ind]));
			if (frac - top_frac > EPS || (std::abs(frac - top_frac) < EPS && long(cells[ind].genes.size()) > top_genes)) {
				const int ed = int(edit_distance(cells[base].barcode.c_str(), cells[ind].barcode.c_str()));
				if (ed >= cfg.max_cb_merge_ed) continue;
				top = long(ind); top_frac = frac; top_genes = long(cells[ind].genes.size());
			}
		}
		if (top_frac < cfg.min_merge_fraction) return long(base);
		return top;
	}

	// ---- PoissonTargetEstimator (Estimation/Merge/PoissonTargetEstimator.cpp) --------------------------------------
	// Rcpp::ppois(k - 1, lambda, lower = false) = P(X >= k), X ~ Poisson(lambda): summed here term by term from the
	// mode outwards in log space (R uses its own pgamma; pinned against scipy.stats.poisson.sf in the tests)
	static double poisson_upper_tail(long k, double lambda) {
		if (k <= 0) return 1.0;
		if (!(lambda > 0)) return 0.0;
		// P(X >= k) = 1 - sum_{j<k} pmf(j) when k is below the mean (stable: the sum is not close to 1 ... unless k >> lambda),
		// else the direct sum of the upper terms
		auto logpmf = [&](long j) { return -lambda + double(j) * std::log(lambda) - std::lgamma(double(j) + 1.0); };
		if (double(k) > lambda) {
			double sum = 0;
			for (long j = k;; ++j) {
				const double t = std::exp(logpmf(j));
				sum += t;
				if (t < sum * 1e-18 || t == 0) break;
			}
			return sum > 1 ? 1.0 : sum;
		}
		double sum = 0;
		for (long j = k - 1; j >= 0; --j) sum += std::exp(logpmf(j));
		return sum >= 1 ? 0.0 : 1.0 - sum;
	}
	std::vector<double> umi_probs;
	CollisionsAdjuster adjuster;
	std::map<std::pair<size_t, size_t>, double> est_cache;
	void poisson_init() {                                         // PoissonTargetEstimator::init (:46-60)
		std::unordered_map<std::string, size_t> dist;              // CellsDataContainer::umi_distribution (:182-197)
		for (size_t id : filtered)
			for (auto const &g : cells[id].genes)
				for (auto const &u : g.second) dist[umi_ix.values.at(u.first)]++;
		double sum = 0;
		for (auto const &it : dist) sum += double(it.second);
		umi_probs.clear();
		for (auto const &it : dist) umi_probs.push_back(double(it.second) / sum);
		adjuster = CollisionsAdjuster();
		adjuster.init(umi_probs, 0);
		est_cache.clear();
	}
	double estimate_genes_intersection_size(size_t g1, size_t g2) {   // (:96-127)
		if (g1 > g2) std::swap(g1, g2);
		g1 = adjuster.estimate(g1); g2 = adjuster.estimate(g2);
		auto it = est_cache.find(std::make_pair(g1, g2));
		if (it != est_cache.end()) return it->second;
		const size_t d = g2 - g1;
		double est = 0;
		for (size_t i = 0; i < umi_probs.size(); ++i) {
			const double mn = fpow(1 - umi_probs[i], long(g1));
			const double mx = mn * fpow(1 - umi_probs[i], long(d));
			est += (1 - mn) * (1 - mx);
		}
		est_cache.emplace(std::make_pair(g1, g2), est);
		return est;
	}
	double intersection_prob(size_t c1, size_t c2, double *expected_out = nullptr) {   // estimate_intersection_prob (:67-94)
		const size_t inter = umig_intersection(cells[c1], cells[c2]);
		if (expected_out) *expected_out = -1;
		if (inter == 0) return 1.0;
		double expected = 0;
		for (auto const &g1 : cells[c1].genes) {
			auto g2 = cells[c2].genes.find(g1.first);
			if (g2 == cells[c2].genes.end()) continue;
			expected += estimate_genes_intersection_size(g1.second.size(), g2->second.size());
		}
		if (expected_out) *expected_out = expected;
		return poisson_upper_tail(long(inter), expected);
	}
	long merge_all_target(size_t base) const {                    // MergeAllMergeStrategy.h:16-50
		int min_ed = std::numeric_limits<int>::max(), max_umi_num = 0;
		size_t target = std::numeric_limits<size_t>::max();
		for (size_t ind : filtered) {
			const int n_umis = umis_number(cells[ind]);
			if (n_umis <= umis_number(cells[base])) continue;
			const int ed = int(edit_distance(cells[base].barcode.c_str(), cells[ind].barcode.c_str(), false, unsigned(cfg.max_cb_merge_ed)));
			if (ed > cfg.max_cb_merge_ed) continue;
			if (min_ed > ed) { min_ed = ed; max_umi_num = n_umis; target = ind; }
			else if ((min_ed == ed) & (max_umi_num < n_umis)) { max_umi_num = n_umis; target = ind; }
		}
		return target != std::numeric_limits<size_t>::max() ? long(target) : long(base);
	}
	long poisson_simple_merge_target(size_t base) {               // PoissonSimpleMergeStrategy::get_merge_target (:15-43)
		std::vector<size_t> nb;
		for (auto const &c : cells_with_common_umigs(base)) {
			const unsigned ed = edit_distance(cells[base].barcode.c_str(), cells[c.first].barcode.c_str());
			if (ed > unsigned(cfg.max_cb_merge_ed)) continue;      // (:27; the plain Simple strategy uses >=)
			nb.push_back(c.first);
		}
		if (nb.empty()) return long(base);
		const long t = poisson_best_target(base, nb);
		return t != -1 ? t : long(base);
	}
	long poisson_merge_target(size_t base) {                      // get_merge_target (:22-29) + get_best_merge_target (:14-44)
		std::vector<size_t> nb = real_neighbour_cells(base);
		if (nb.empty()) return -1;
		return poisson_best_target(base, nb);
	}
	long poisson_best_target(size_t base, const std::vector<size_t> &nb) {   // PoissonTargetEstimator::get_best_merge_target (:14-44)
		const bool base_real = nb.at(0) == base;
		double max_prob = (base_real ? cfg.max_merge_prob : cfg.max_real_merge_prob) / double(nb.size());
		long best = -1;
		double min_prob = 2;
		for (size_t n : nb) {
			if (n == base) continue;
			const double prob = intersection_prob(base, n);
			if (prob < min_prob) { min_prob = prob; best = long(n); }
		}
		if (min_prob > max_prob) return base_real ? long(base) : -1;
		return best;
	}

	// MergeStrategyBase.cpp:11-57, :64-82 ; DummyMergeStrategy.h:12-17
	std::vector<size_t> run_cb_merge() {
		std::vector<size_t> reassign(cells.size());
		std::iota(reassign.begin(), reassign.end(), size_t(0));
		if (cfg.merge_kind == 0) return reassign;

		std::unordered_map<size_t, std::unordered_set<size_t>> reassigned_to;
		std::vector<long> targets(filtered.size());
		if (cfg.merge_kind == 2 || cfg.merge_kind == 4) simple_init();
		if (cfg.merge_kind == 3 || cfg.merge_kind == 4) poisson_init();
		for (size_t i = 0; i < filtered.size(); ++i)
			targets[i] = cfg.merge_kind == 5 ? merge_all_target(filtered[i])
			           : cfg.merge_kind == 4 ? poisson_simple_merge_target(filtered[i])
			           : cfg.merge_kind == 2 ? simple_merge_target(filtered[i])
			           : cfg.merge_kind == 3 ? poisson_merge_target(filtered[i]) : real_merge_target(filtered[i]);
		cell_ids_by_umig.clear();

		for (size_t i = 0; i < filtered.size(); ++i) {
			const size_t base = filtered[i];
			long tgt = targets[i];
			if (tgt < 0) { cells.at(base).excluded = true; continue; }
			if (size_t(tgt) != reassign.at(size_t(tgt))) tgt = long(reassign[size_t(tgt)]);
			if (size_t(tgt) == base) continue;
			merge_cells(base, size_t(tgt));
			// reassign(): MergeStrategyBase.cpp:64-82
			reassign[base] = size_t(tgt);
			reassigned_to[size_t(tgt)].insert(base);
			auto it = reassigned_to.find(base);
			if (it != reassigned_to.end()) {
				for (size_t moved : it->second) {
					reassign[moved] = size_t(tgt);
					reassigned_to[size_t(tgt)].insert(moved);
				}
				reassigned_to.find(base)->second.clear();
			}
		}
		return reassign;
	}

	// MergeUMIsStrategyAbstract.cpp:11-23 (glibc rand(), seeded 42 in the strategy ctor)
	static std::string fix_n_with_random(const std::string &umi) {
		static const char nt[] = "ACGT";
		std::string t(umi);
		for (char &c : t) if (c == 'N') c = nt[rand() % 4];
		return t;
	}

	// Gene.cpp:38-58
	void rekey_umi(umis_t &umis, const std::string &src, const std::string &tgt) {
		if (src == tgt) return;
		auto s = umis.find(umi_ix.get(src));
		if (s == umis.end()) throw std::runtime_error("Source UMI doesn't belong to the gene: " + src);
		auto t = umis.emplace(umi_ix.add(tgt), s->second);
		if (!t.second) { t.first->second.reads += s->second.reads; t.first->second.mark |= s->second.mark; }
		umis.erase(s);
	}
	// Cell.cpp:31-42 (TOTAL_UMIS decremented once per re-targeted UMI, quirk ii)
	void merge_umis(size_t cell_id, size_t gene, const std::unordered_map<std::string, std::string> &targets) {
		Cell &c = cells.at(cell_id);
		umis_t &umis = c.genes.at(gene);
		for (auto const &t : targets) {
			if (t.second == t.first) continue;
			rekey_umi(umis, t.first, t.second);
			c.total_umis--;
		}
	}

	// MergeUMIsStrategySimple.cpp:66-102
	std::unordered_map<std::string, std::string>
	n_umi_targets(const umis_t &all, const std::unordered_set<std::string> &bad) const {
		std::unordered_map<std::string, std::string> out;
		for (auto const &b : bad) {
			unsigned min_ed = std::numeric_limits<unsigned>::max();
			std::string best;
			size_t best_size = 0;
			for (auto const &cand : all) {
				const std::string &seq = umi_ix.values.at(cand.first);
				if (bad.find(seq) != bad.end()) continue;
				unsigned ed = hamming_distance(seq, b);
				if (ed < min_ed || (ed == min_ed && cand.second.reads > best_size)) {
					min_ed = ed; best = seq; best_size = cand.second.reads;
				}
			}
			if (best.empty() || min_ed > cfg.max_umi_merge_ed) out[b] = fix_n_with_random(b);
			else out[b] = best;
		}
		return out;
	}
	// MergeUMIsStrategySimple.cpp:21-59
	void run_umi_merge_simple() {
		for (size_t cid = 0; cid < cells.size(); ++cid) {
			if (!is_real(cells[cid])) continue;
			for (auto const &g : cells[cid].genes) {
				std::unordered_set<std::string> bad;
				for (auto const &u : g.second) {
					const std::string &seq = umi_ix.values.at(u.first);
					if (seq.find('N') != std::string::npos) bad.insert(seq);
				}
				if (bad.empty()) continue;
				auto targets = n_umi_targets(g.second, bad);
				merge_umis(cid, g.first, targets);
			}
		}
	}

	// MergeUMIsStrategyDirectional.cpp:55-116
	struct UmiWrap { std::string seq; size_t n_reads; };
	std::string directional_target(size_t src, const std::vector<UmiWrap> &v) const {
		const UmiWrap &s = v[src];
		const bool has_n = s.seq.find('N') != std::string::npos;
		std::string target;
		unsigned min_ed = std::numeric_limits<unsigned>::max();
		for (long d = long(v.size()) - 1; d > long(src); --d) {
			const UmiWrap &t = v[size_t(d)];
			if (double(s.n_reads) * cfg.umi_mult > double(t.n_reads)) break;
			unsigned ed = edit_distance(s.seq.c_str(), t.seq.c_str(), true, cfg.max_umi_merge_ed);
			if (ed > cfg.max_umi_merge_ed) continue;
			if (ed < min_ed) {
				target = t.seq;
				if ((!has_n && ed <= 1) || ed == 0) break;
				min_ed = ed;
			}
		}
		if (has_n && target.empty()) return fix_n_with_random(s.seq);
		return target;
	}
	std::unordered_map<std::string, std::string> directional_targets(std::vector<UmiWrap> &v) const {
		std::sort(v.begin(), v.end(), [](const UmiWrap &a, const UmiWrap &b) { return a.n_reads < b.n_reads; });
		std::unordered_map<std::string, std::string> out;
		for (size_t i = 0; i < v.size(); ++i) {
			std::string t = directional_target(i, v);
			if (!t.empty()) out[v[i].seq] = t;
		}
		for (long i = long(v.size()) - 1; i >= 0; --i) {
			auto it = out.find(v[size_t(i)].seq);
			if (it == out.end()) continue;
			auto it2 = out.find(it->second);
			if (it2 == out.end()) continue;
			out[v[size_t(i)].seq] = it2->second;
		}
		return out;
	}
	void run_umi_merge_directional() {   // MergeUMIsStrategyDirectional.cpp:18-53
		for (size_t cid = 0; cid < cells.size(); ++cid) {
			if (!is_real(cells[cid])) continue;
			for (auto const &g : cells[cid].genes) {
				std::vector<UmiWrap> v;
				for (auto const &u : g.second) v.push_back(UmiWrap{umi_ix.values.at(u.first), u.second.reads});
				auto targets = directional_targets(v);
				if (targets.empty()) continue;
				merge_umis(cid, g.first, targets);
			}
		}
	}

	void merge_and_filter() {   // CellsDataContainer.cpp:39-57
		if (!initialized) throw std::runtime_error("You must initialize container");
		merge_targets = run_cb_merge();
		if (cfg.umi_merge_kind == 0) run_umi_merge_simple(); else run_umi_merge_directional();
		update_cell_sizes(cfg.min_genes_after, cfg.max_cells);
	}
};

}  // namespace orc

// ================================================================================================
// C interface for the Python tests / bench cpu_baseline (ctypes).  All functions return 0 on success,
// -1 on a caught exception (message via orc_last_error()).
// ================================================================================================

static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH } catch (const std::exception &e) { g_err = e.what(); return -1; } return 0;

extern "C" {

struct orc_config {
	int merge_kind, barcodes_kind;
	const char *barcodes_file;
	int min_genes_before, min_genes_after;
	double min_merge_fraction;
	int max_cb_merge_ed;
	int umi_merge_kind, max_umi_merge_ed;
	double umi_mult;
	const char *match_levels;
	int max_cells;
	double max_merge_prob, max_real_merge_prob;
};

const char *orc_last_error() { return g_err.c_str(); }

void *orc_create(const orc_config *c) {
	try {
		orc::Config k;
		k.merge_kind = c->merge_kind; k.barcodes_kind = c->barcodes_kind;
		k.barcodes_file = c->barcodes_file ? c->barcodes_file : "";
		k.min_genes_before = size_t(c->min_genes_before); k.min_genes_after = size_t(c->min_genes_after);
		k.min_merge_fraction = c->min_merge_fraction; k.max_cb_merge_ed = c->max_cb_merge_ed;
		k.umi_merge_kind = c->umi_merge_kind; k.max_umi_merge_ed = unsigned(c->max_umi_merge_ed);
		k.umi_mult = c->umi_mult; k.match_levels = c->match_levels ? c->match_levels : "eEBA";
		k.max_cells = c->max_cells;
		k.max_merge_prob = c->max_merge_prob; k.max_real_merge_prob = c->max_real_merge_prob;
		return new orc::Container(k);
	} catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void orc_destroy(void *h) { delete static_cast<orc::Container *>(h); }

int orc_add_record(void *h, const char *cb, const char *umi, const char *umi_qual, const char *gene,
                   const char *chr, int mark) {
	ORC_TRY static_cast<orc::Container *>(h)->add_record(cb, umi, umi_qual, gene, chr, uint8_t(mark)); ORC_CATCH
}

// Bulk ingest of a packed stream (same record layout as include/dropest_amd.h): 2-bit codes with a
// leading sentinel bit, gene id (0xFFFFFFFF = no gene), aux = chr | mark<<16.  Strings are rebuilt
// here: gene "G%u", chromosome "chr%u".  N-containing barcodes/UMIs (bit 63 set) index `side`.
static std::string unpack2(uint64_t code, const char *const *side) {
	if (code >> 63) return side[code & 0x7FFFFFFFFFFFFFFFull];
	int top = 63 - __builtin_clzll(code);   // sentinel position = 2*len
	int len = top / 2;
	std::string s(size_t(len), 'A');
	for (int i = 0; i < len; ++i) s[size_t(i)] = "ACGT"[(code >> (2 * (len - 1 - i))) & 3];
	return s;
}
int orc_add_packed(void *h, const uint64_t *cb, const uint64_t *umi, const uint32_t *gene, const uint32_t *aux,
                   uint64_t n, const char *const *side) {
	ORC_TRY
	auto *c = static_cast<orc::Container *>(h);
	for (uint64_t i = 0; i < n; ++i) {
		std::string g = gene[i] == 0xFFFFFFFFu ? std::string() : "G" + std::to_string(gene[i]);
		c->add_record(unpack2(cb[i], side), unpack2(umi[i], side), "", g,
		              "chr" + std::to_string(aux[i] & 0xFFFFu), uint8_t((aux[i] >> 16) & 0xFF));
	}
	ORC_CATCH
}

// same with UMI qualities: qual = n x qlen bytes (phred+33 characters)
int orc_add_packed_q(void *h, const uint64_t *cb, const uint64_t *umi, const uint32_t *gene, const uint32_t *aux,
                     uint64_t n, const char *const *side, const uint8_t *qual, uint32_t qlen) {
	ORC_TRY
	auto *c = static_cast<orc::Container *>(h);
	for (uint64_t i = 0; i < n; ++i) {
		std::string g = gene[i] == 0xFFFFFFFFu ? std::string() : "G" + std::to_string(gene[i]);
		c->add_record(unpack2(cb[i], side), unpack2(umi[i], side), std::string(reinterpret_cast<const char *>(qual) + size_t(i) * qlen, qlen), g,
		              "chr" + std::to_string(aux[i] & 0xFFFFu), uint8_t((aux[i] >> 16) & 0xFF));
	}
	ORC_CATCH
}

// UMI::_sum_quality of every molecule, in the order of orc_molecules; qlen values each (molecules with another
// length -- impossible through orc_add_packed_q -- report -1)
int orc_molecule_qualities(void *h, uint32_t qlen, uint32_t *out) {
	auto *c = static_cast<orc::Container *>(h);
	uint64_t n = 0;
	for (size_t i = 0; i < c->cells.size(); ++i)
		for (auto const &g : c->cells[i].genes)
			for (auto const &u : g.second) {
				if (u.second.qual_sum.size() != qlen) return -1;
				for (uint32_t p = 0; p < qlen; ++p) out[n * qlen + p] = u.second.qual_sum[p];
				++n;
			}
	return 0;
}

// strings of several lengths: rows of `stride` bytes, lens[i] characters of read i count (UMI.cpp:21-34 per molecule)
int orc_add_packed_qvar(void *h, const uint64_t *cb, const uint64_t *umi, const uint32_t *gene, const uint32_t *aux,
                        uint64_t n, const char *const *side, const uint8_t *qual, uint32_t stride, const uint8_t *lens) {
	ORC_TRY
	auto *c = static_cast<orc::Container *>(h);
	for (uint64_t i = 0; i < n; ++i) {
		std::string g = gene[i] == 0xFFFFFFFFu ? std::string() : "G" + std::to_string(gene[i]);
		c->add_record(unpack2(cb[i], side), unpack2(umi[i], side), std::string(reinterpret_cast<const char *>(qual) + size_t(i) * stride, lens[i]), g,
		              "chr" + std::to_string(aux[i] & 0xFFFFu), uint8_t((aux[i] >> 16) & 0xFF));
	}
	ORC_CATCH
}
// ... and their sums: `stride` values per molecule (zeros beyond its own length) and the length itself
int orc_molecule_qualities_var(void *h, uint32_t stride, uint32_t *out, uint32_t *out_len) {
	auto *c = static_cast<orc::Container *>(h);
	uint64_t n = 0;
	for (size_t i = 0; i < c->cells.size(); ++i)
		for (auto const &g : c->cells[i].genes)
			for (auto const &u : g.second) {
				if (u.second.qual_sum.size() > stride) return -1;
				for (uint32_t p = 0; p < stride; ++p) out[n * stride + p] = p < u.second.qual_sum.size() ? u.second.qual_sum[p] : 0u;
				out_len[n] = uint32_t(u.second.qual_sum.size());
				++n;
			}
	return 0;
}

int orc_set_initialized(void *h) { ORC_TRY static_cast<orc::Container *>(h)->set_initialized(); ORC_CATCH }
int orc_merge_and_filter(void *h) { ORC_TRY static_cast<orc::Container *>(h)->merge_and_filter(); ORC_CATCH }
int orc_merge_umis_only(void *h) {   // Tests/TestEstimation.cpp:525 calls the UMI strategy alone
	ORC_TRY
	auto *c = static_cast<orc::Container *>(h);
	if (c->cfg.umi_merge_kind == 0) c->run_umi_merge_simple(); else c->run_umi_merge_directional();
	ORC_CATCH
}

uint64_t orc_n_cells(void *h) { return static_cast<orc::Container *>(h)->cells.size(); }
uint64_t orc_n_genes(void *h) { return static_cast<orc::Container *>(h)->gene_ix.values.size(); }
uint64_t orc_n_filtered(void *h) { return static_cast<orc::Container *>(h)->filtered.size(); }
uint64_t orc_n_real(void *h) { return static_cast<orc::Container *>(h)->real_cells; }
void orc_filtered(void *h, uint64_t *out) {
	auto *c = static_cast<orc::Container *>(h);
	for (size_t i = 0; i < c->filtered.size(); ++i) out[i] = c->filtered[i];
}
void orc_merge_targets(void *h, uint64_t *out) {
	auto *c = static_cast<orc::Container *>(h);
	for (size_t i = 0; i < c->merge_targets.size(); ++i) out[i] = c->merge_targets[i];
}
uint64_t orc_n_merge_targets(void *h) { return static_cast<orc::Container *>(h)->merge_targets.size(); }
long orc_cell_id_by_cb(void *h, const char *cb) {
	auto *c = static_cast<orc::Container *>(h);
	auto it = c->cell_by_cb.find(cb);
	return it == c->cell_by_cb.end() ? -1 : long(it->second);
}
const char *orc_cell_barcode(void *h, uint64_t id) { return static_cast<orc::Container *>(h)->cells.at(id).barcode.c_str(); }
const char *orc_gene_name(void *h, uint64_t id) { return static_cast<orc::Container *>(h)->gene_ix.values.at(id).c_str(); }
const char *orc_chr_name(void *h, uint64_t id) { return static_cast<orc::Container *>(h)->chr_ix.values.at(id).c_str(); }
uint64_t orc_n_chr(void *h) { return static_cast<orc::Container *>(h)->chr_ix.values.size(); }

// per-cell row: [merged, excluded, real, n_genes, req_genes, req_umis, total_reads, total_umis]
void orc_cell_rows(void *h, int64_t *out) {
	auto *c = static_cast<orc::Container *>(h);
	for (size_t i = 0; i < c->cells.size(); ++i) {
		const orc::Cell &x = c->cells[i];
		int64_t *r = out + 8 * i;
		r[0] = x.merged; r[1] = x.excluded; r[2] = c->is_real(x); r[3] = int64_t(x.genes.size());
		r[4] = int64_t(x.req_genes); r[5] = int64_t(x.req_umis); r[6] = x.total_reads; r[7] = x.total_umis;
	}
}
// global counters: [intergenic, exon, intron, not_annotated]
void orc_global_counters(void *h, uint64_t *out) {
	auto *c = static_cast<orc::Container *>(h);
	out[0] = c->intergenic_reads; out[1] = c->exon_reads; out[2] = c->intron_reads; out[3] = c->not_annotated_reads;
}

// Molecule table in (cell id, gene idx, umi idx) order.  Call with out pointers == NULL to count.
// `umi_buf` receives fixed-stride NUL-padded strings (stride bytes each).
uint64_t orc_molecules(void *h, uint64_t *cell, uint64_t *gene, char *umi_buf, int stride, uint64_t *reads,
                       uint8_t *mark) {
	auto *c = static_cast<orc::Container *>(h);
	uint64_t n = 0;
	for (size_t i = 0; i < c->cells.size(); ++i)
		for (auto const &g : c->cells[i].genes)
			for (auto const &u : g.second) {
				if (cell) {
					cell[n] = i; gene[n] = g.first; reads[n] = u.second.reads; mark[n] = u.second.mark;
					std::strncpy(umi_buf + size_t(stride) * n, c->umi_ix.values.at(u.first).c_str(), size_t(stride));
				}
				++n;
			}
	return n;
}

// Count matrices as triplets (gene idx, column, value); follows Cell.cpp:54-68 +
// ResultsPrinter.cpp:334-361 (filtered: columns = filtered cells ascending, requested UMIs only) and
// ResultsPrinter.cpp:363-396 (raw: columns = real cells in cell-id order, all UMIs).  Row ids here are
// the container's gene indices (the .rds row order is a pure relabelling).
// Triplets are emitted column-major, genes ascending.  out == NULL counts.
uint64_t orc_count_matrix(void *h, int filtered, int reads_output, uint64_t *gene, uint64_t *col, uint64_t *val) {
	auto *c = static_cast<orc::Container *>(h);
	uint64_t n = 0, column = 0;
	auto emit_cell = [&](const orc::Cell &x, bool requested_only) {
		for (auto const &g : x.genes) {
			size_t v;
			if (requested_only) v = orc::Container::requested_in_gene(g.second, c->query, reads_output != 0);
			else if (!reads_output) v = g.second.size();
			else { v = 0; for (auto const &u : g.second) v += u.second.reads; }
			if (requested_only && v == 0) continue;
			if (gene) { gene[n] = g.first; col[n] = column; val[n] = v; }
			++n;
		}
	};
	if (filtered) {
		for (size_t id : c->filtered) { emit_cell(c->cells[id], true); ++column; }
	} else {
		for (auto const &x : c->cells) { if (!c->is_real(x)) continue; emit_cell(x, false); ++column; }
	}
	return n;
}

// ResultsPrinter::get_count_matrix_filtered(container, query_marks) for an explicit mark query
// (save_intron_exon_matrices, ResultsPrinter.cpp:455-474): filtered cells, zero entries dropped
uint64_t orc_count_matrix_levels(void *h, const char *levels, int reads_output, uint64_t *gene, uint64_t *col, uint64_t *val) {
	auto *c = static_cast<orc::Container *>(h);
	const std::vector<uint8_t> query = orc::marks_by_code(levels);
	uint64_t n = 0, column = 0;
	for (size_t id : c->filtered) {
		for (auto const &g : c->cells[id].genes) {
			const size_t v = orc::Container::requested_in_gene(g.second, query, reads_output != 0);
			if (v == 0) continue;
			if (gene) { gene[n] = g.first; col[n] = column; val[n] = v; }
			++n;
		}
		++column;
	}
	return n;
}

// Per-chromosome stats of real cells (CellsDataContainer.cpp:291-307 content, order-free form):
// rows (cell id, kind, chr id, count) for every non-zero entry of every real cell.
uint64_t orc_chr_stats(void *h, uint64_t *cell, int32_t *kind, uint64_t *chr, int64_t *count) {
	auto *c = static_cast<orc::Container *>(h);
	uint64_t n = 0;
	for (size_t i = 0; i < c->cells.size(); ++i) {
		if (!c->is_real(c->cells[i])) continue;
		for (int k = 0; k < orc::CHR_KINDS; ++k) {
			std::map<size_t, int> ordered(c->cells[i].chr[k].begin(), c->cells[i].chr[k].end());
			for (auto const &kv : ordered) {
				if (cell) { cell[n] = i; kind[n] = k; chr[n] = kv.first; count[n] = kv.second; }
				++n;
			}
		}
	}
	return n;
}

// CellsDataContainer::umi_distribution (CellsDataContainer.cpp:182-197): molecules per UMI over the filtered
// cells; returned sorted by UMI string (the reference returns an unordered_map).  out == NULL counts.
uint64_t orc_umi_distribution(void *h, char *umi_buf, int stride, uint64_t *counts) {
	auto *c = static_cast<orc::Container *>(h);
	std::map<std::string, size_t> dist;
	for (size_t id : c->filtered)
		for (auto const &g : c->cells[id].genes)
			for (auto const &u : g.second) dist[c->umi_ix.values.at(u.first)]++;
	uint64_t n = 0;
	for (auto const &kv : dist) {
		if (umi_buf) { std::strncpy(umi_buf + size_t(stride) * n, kv.first.c_str(), size_t(stride)); counts[n] = kv.second; }
		++n;
	}
	return n;
}

// ---- fine-grained entry points used to pin the oracle on the reference's unit tests ----
// PoissonTargetEstimator (Tests/TestEstimationMergeProbs.cpp:93-140)
int orc_poisson_init(void *h) { ORC_TRY static_cast<orc::Container *>(h)->poisson_init(); ORC_CATCH }
uint64_t orc_poisson_distribution_size(void *h) { return static_cast<orc::Container *>(h)->umi_probs.size(); }
double orc_poisson_gene_intersection(void *h, uint64_t g1, uint64_t g2) {
	try { return static_cast<orc::Container *>(h)->estimate_genes_intersection_size(g1, g2); } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
double orc_poisson_intersection_prob(void *h, uint64_t c1, uint64_t c2) {
	try { return static_cast<orc::Container *>(h)->intersection_prob(c1, c2); } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
double orc_poisson_expected_intersection(void *h, uint64_t c1, uint64_t c2) {   // -1: empty intersection (:72-75)
	try { double e = -1; static_cast<orc::Container *>(h)->intersection_prob(c1, c2, &e); return e; } catch (const std::exception &e) { g_err = e.what(); return -2; }
}
long orc_poisson_merge_target(void *h, uint64_t cell) {
	try { return static_cast<orc::Container *>(h)->poisson_merge_target(cell); } catch (const std::exception &e) { g_err = e.what(); return -2; }
}
double orc_poisson_upper_tail(long k, double lambda) { return orc::Container::poisson_upper_tail(k, lambda); }

unsigned orc_edit_distance(const char *a, const char *b, int skip_n, unsigned max_ed) { return orc::edit_distance(a, b, skip_n != 0, max_ed); }
int orc_hamming_distance(const char *a, const char *b, int skip_n) {
	try { return int(orc::hamming_distance(a, b, skip_n != 0)); } catch (...) { return -1; }
}
int orc_parse_encoded_id(const char *id, char *cb, char *umi, int cap) {
	std::string c, u;
	if (!orc::parse_encoded_id(id, c, u)) return -1;
	std::strncpy(cb, c.c_str(), size_t(cap)); std::strncpy(umi, u.c_str(), size_t(cap));
	return 0;
}
int orc_reverse_complement(const char *s, char *out, int cap) {
	try { std::strncpy(out, orc::reverse_complement(s).c_str(), size_t(cap)); return 0; } catch (...) { return -1; }
}
long orc_merge_target(void *h, uint64_t cell) {
	try { return static_cast<orc::Container *>(h)->real_merge_target(cell); }
	catch (const std::exception &e) { g_err = e.what(); return -2; }
}
uint64_t orc_real_neighbours(void *h, uint64_t cell, uint64_t *out, uint64_t cap) {
	auto v = static_cast<orc::Container *>(h)->real_neighbour_cells(cell);
	for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
	return v.size();
}
uint64_t orc_umig_intersection(void *h, uint64_t a, uint64_t b) {
	auto *c = static_cast<orc::Container *>(h);
	return orc::Container::umig_intersection(c->cells.at(a), c->cells.at(b));
}
// whitelist inspection: number of parts / size of a part / a barcode of a part
uint64_t orc_wl_parts(void *h) { return static_cast<orc::Container *>(h)->wl.parts.size(); }
uint64_t orc_wl_part_size(void *h, uint64_t p) { return static_cast<orc::Container *>(h)->wl.parts.at(p).size(); }
const char *orc_wl_barcode(void *h, uint64_t p, uint64_t i) { return static_cast<orc::Container *>(h)->wl.parts.at(p).at(i).c_str(); }
// sorted per-part distances of a barcode (value and index arrays of the given part)
uint64_t orc_wl_distances(void *h, const char *cb, uint64_t part, int64_t *values, uint64_t *indices) {
	try {
		auto d = static_cast<orc::Container *>(h)->wl.distances(cb);
		for (size_t i = 0; i < d.at(part).size(); ++i) { values[i] = d[part][i].value; indices[i] = d[part][i].index; }
		return d[part].size();
	} catch (const std::exception &e) { g_err = e.what(); return 0; }
}
// standalone whitelist for parser tests (no container): returns handle to a Container with merge off
int orc_wl_load(void *h, int kind, const char *file) {
	ORC_TRY static_cast<orc::Container *>(h)->wl.load(orc::Whitelist::Kind(kind), file); ORC_CATCH
}
int orc_wl_set(void *h, int kind, const char *const *part0, int n0, const char *const *part1, int n1) {
	ORC_TRY
	auto &w = static_cast<orc::Container *>(h)->wl;
	w.kind = orc::Whitelist::Kind(kind);
	w.parts.assign(2, {});
	for (int i = 0; i < n0; ++i) w.parts[0].push_back(part0[i]);
	for (int i = 0; i < n1; ++i) w.parts[1].push_back(part1[i]);
	w.finish_init();
	ORC_CATCH
}
int orc_wl_split(void *h, const char *cb, char *out, int stride) {
	try {
		auto v = static_cast<orc::Container *>(h)->wl.split(cb);
		for (size_t i = 0; i < v.size(); ++i) std::strncpy(out + size_t(stride) * i, v[i].c_str(), size_t(stride));
		return int(v.size());
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}
// explicit UMI re-keying (CellsDataContainer::merge_umis, Tests/TestEstimation.cpp:468-488)
int orc_merge_umis_explicit(void *h, uint64_t cell, const char *gene, const char *const *src, const char *const *tgt, int n) {
	ORC_TRY
	auto *c = static_cast<orc::Container *>(h);
	std::unordered_map<std::string, std::string> m;
	for (int i = 0; i < n; ++i) m[src[i]] = tgt[i];
	c->merge_umis(cell, c->gene_ix.get(gene), m);
	ORC_CATCH
}
// CellsDataContainer::merge_cells / exclude_cell called directly (CellsDataContainer.cpp:90-109); the filtered list is
// refreshed like merge_and_filter does (update_cell_sizes keeps the last threshold)
int orc_merge_cells_explicit(void *h, uint64_t src, uint64_t tgt) {
	ORC_TRY
	auto *c = static_cast<orc::Container *>(h);
	c->merge_cells(src, tgt);
	ORC_CATCH
}
int orc_exclude_cell_explicit(void *h, uint64_t cell) {
	ORC_TRY
	static_cast<orc::Container *>(h)->cells.at(cell).excluded = true;
	ORC_CATCH
}
// MergeUMIsStrategySimple::fill_wrong_umis (MergeUMIsStrategySimple.cpp:104-112)
int orc_fill_wrong_umi(const char *umi, char *out, int cap) {
	std::strncpy(out, orc::Container::fix_n_with_random(umi).c_str(), size_t(cap));
	return 0;
}
// MergeUMIsStrategyDirectional::find_targets on an explicit (sequence, reads) list
int orc_directional_targets(void *h, const char *const *seqs, const uint64_t *reads, int n, char *src_out, char *tgt_out, int stride) {
	auto *c = static_cast<orc::Container *>(h);
	std::vector<orc::Container::UmiWrap> v;
	for (int i = 0; i < n; ++i) v.push_back({seqs[i], size_t(reads[i])});
	auto t = c->directional_targets(v);
	std::map<std::string, std::string> ordered(t.begin(), t.end());
	int k = 0;
	for (auto const &kv : ordered) {
		std::strncpy(src_out + size_t(stride) * size_t(k), kv.first.c_str(), size_t(stride));
		std::strncpy(tgt_out + size_t(stride) * size_t(k), kv.second.c_str(), size_t(stride));
		++k;
	}
	return k;
}
// Tools::CollisionsAdjuster table: adjusted_size[s] for s = 1..max_expr
int orc_collisions_table(const double *probs, uint64_t n, uint64_t max_expr, uint64_t *out) {
	ORC_TRY
	orc::CollisionsAdjuster a;
	a.init(std::vector<double>(probs, probs + n), max_expr);
	for (uint64_t s = 0; s < max_expr; ++s) out[s] = a.adjusted[s];
	ORC_CATCH
}

}  // extern "C"
