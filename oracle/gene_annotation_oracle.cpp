// gene_annotation_oracle.cpp -- CPU ORACLE for the reference's gene annotation (-g): Tools/GeneAnnotation/*
// and ReadParamsParser::get_gene_from_reference.
//
// *** TEST INFRASTRUCTURE ONLY *** (same rules as dropest_oracle.cpp: nothing under dropest_amd/ may use it).
//
// A restatement that keeps the reference's structure, so that the boundary behaviour of its containers is inherited:
//   Interval (Interval.cpp:8-47), IntervalsContainer<Label> (IntervalsContainer.h:24-239: per-label lists merged on
//   insertion, open/close events in a multimap, homogeneous intervals, lower_bound query), GtfRecord
//   (GtfRecord.cpp:7-54), RefGenesContainer (RefGenesContainer.cpp:23-268: GTF / BED parsing, transcripts, exons by
//   transcript, get_gene_info) and ReadParamsParser::get_gene_from_reference / find_exon (ReadParamsParser.cpp:92-176).
// The gzip layer (boost::iostreams in the reference) is zlib here.  Pinned on Tests/TestTools.cpp (testGtf,
// testGeneMerge, testInitGtf, testInterval, testGenesWithIntrons) by tests/test_oracle_reference_kat.py.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace ga {

typedef size_t coord_t;

struct Interval {                                        // Interval.cpp
	coord_t start, end;
	Interval(coord_t s, coord_t e) : start(s), end(e) {}
	bool is_intercept(const Interval &o) const { return start <= o.end && end > o.start; }
	void merge(const Interval &o) { start = std::min(start, o.start); end = std::max(end, o.end); }
};

template <class Label>
struct IntervalsContainer {                              // IntervalsContainer.h
	struct Base : Interval { Label label; Base(coord_t s, coord_t e, Label l) : Interval(s, e), label(l) {} };
	struct Query : Interval { std::set<Label> labels; Query(coord_t s, coord_t e, const std::set<Label> &l) : Interval(s, e), labels(l) {} };
	struct Event { const Label *label; bool open; };
	bool initialized = false, allow_intercepts;
	unsigned min_len;
	std::vector<Query> homogeneous;
	std::map<Label, std::list<Base>> base;
	explicit IntervalsContainer(bool allow = true, unsigned min_interval_length = 1) : allow_intercepts(allow), min_len(min_interval_length) {}

	void add_interval(coord_t s, coord_t e, Label label, bool force = false) {           // :151-190
		if (!force && initialized) throw std::runtime_error("IntervalsContainer is already initialized");
		Base query(s, e, label);
		auto &cur = base[label];
		auto it = cur.begin();
		while (it != cur.end() && !query.is_intercept(*it)) {
			if (it->start > query.end) { cur.insert(it, query); return; }
			++it;
		}
		if (it == cur.end()) { cur.push_back(query); return; }
		auto end_it = it;
		++end_it;
		while (end_it != cur.end() && query.is_intercept(*end_it)) { query.merge(*end_it); ++end_it; }
		it->merge(query);
		++it;
		cur.erase(it, end_it);
	}
	void set_initialized(bool clear = true) {                                            // :192-210, :102-142
		initialized = true;
		std::vector<Base> all;
		for (auto &l : base) for (auto &b : l.second) all.push_back(b);
		std::multimap<coord_t, Event> events;
		for (auto const &b : all) { events.emplace(b.start, Event{&b.label, true}); events.emplace(b.end, Event{&b.label, false}); }
		homogeneous.clear();
		coord_t start_pos = 0, end_pos;
		std::set<Label> cur;
		for (auto const &ev : events) {
			end_pos = ev.first;
			if (!cur.empty() && end_pos - start_pos >= min_len) {
				if (!allow_intercepts && cur.size() > 1)
					throw std::runtime_error("Intervals intersection at (" + std::to_string(start_pos) + ", " + std::to_string(end_pos) + ")");
				homogeneous.push_back(Query(start_pos, end_pos, cur));
			}
			if (ev.second.open) cur.insert(*ev.second.label); else cur.erase(*ev.second.label);
			start_pos = end_pos;
		}
		if (clear) base.clear();
	}
	std::set<Label> get_intervals(coord_t s, coord_t e) const {                          // :216-238
		if (!initialized) throw std::runtime_error("Interval must be initialized");
		auto it = std::lower_bound(homogeneous.begin(), homogeneous.end(), s, [](const Query &q, coord_t pos) { return q.end <= pos; });
		if (it == homogeneous.end() || it->start >= e) return std::set<Label>();
		std::set<Label> res;
		while (it != homogeneous.end() && it->start < e) { res.insert(it->labels.begin(), it->labels.end()); ++it; }
		return res;
	}
};

enum RecordType { NONE = 0, INTRON = 1, EXON = 2 };      // GtfRecord.h:20-26

struct GtfRecord : Interval {                            // GtfRecord.cpp
	std::string chr, gene_id_, gene_name_, transcript_;
	RecordType type = NONE;
	GtfRecord() : Interval(0, 0) {}
	GtfRecord(const std::string &c, const std::string &id, const std::string &name, coord_t s, coord_t e, RecordType t, const std::string &tr = "")
		: Interval(s, e), chr(c), gene_id_(id), gene_name_(name == id ? "" : name), transcript_(tr), type(t) {}
	bool is_valid() const { return !gene_id_.empty(); }
	const std::string &gene_name() const { return gene_name_.empty() ? gene_id_ : gene_name_; }
	const std::string &transcript_id() const { return transcript_.empty() ? gene_id_ : transcript_; }
};

struct QueryResult {                                     // RefGenesContainer.h / .cpp:253-266
	std::string gene_name; RecordType type;
	bool operator<(const QueryResult &o) const { return type == o.type ? gene_name < o.gene_name : type < o.type; }
};

struct RefGenes {                                        // RefGenesContainer.cpp
	bool use_introns_from_gtf = false, gtf_has_transcripts = true;
	std::string format;
	std::unordered_map<std::string, std::map<std::string, GtfRecord>> transcript_positions;
	std::unordered_map<std::string, IntervalsContainer<std::string>> transcript_intervals;
	std::unordered_map<std::string, std::unordered_map<std::string, IntervalsContainer<RecordType>>> exons_by_transcripts;
	std::unordered_map<std::string, std::string> genes_by_transcripts;

	static std::vector<std::string> split(const std::string &rec) { std::istringstream is(rec); std::string c; std::vector<std::string> v; while (is >> c) v.push_back(c); return v; }

	GtfRecord parse_gtf_record(const std::string &record) {                              // :110-177
		GtfRecord result;
		if (record.at(0) == '#') return result;
		std::vector<std::string> col(split(record));
		if (col.size() < 9) throw std::runtime_error("Can't parse record: \n" + record);
		if (col[0] == "." || col[3] == "." || col[4] == "." || col.size() == 9) return result;
		RecordType type;
		if (col[2] == "exon") type = EXON;
		else if (col[2] == "intron") { type = INTRON; use_introns_from_gtf = true; }
		else return result;
		std::string id, name, transcript;
		for (size_t a = 8; a < col.size() - 1; ++a) {
			const std::string &key = col[a], &value = col[a + 1];
			if (key == "gene_id") id = value.substr(1, value.length() - 3);
			if (key == "gene_name") name = value.substr(1, value.length() - 3);
			if (key == "transcript_id") transcript = value.substr(1, value.length() - 3);
		}
		if (transcript.empty()) gtf_has_transcripts = false;
		if (id.empty()) {
			if (name.empty()) throw std::runtime_error("GTF record doesn't contain either gene name or id:\n" + record);
			id = name;
		}
		const size_t s = strtoul(col[3].c_str(), nullptr, 10) - 1, e = strtoul(col[4].c_str(), nullptr, 10);
		return GtfRecord(col[0], id, name, s, e, type, transcript);
	}
	static GtfRecord parse_bed_record(const std::string &record) {                       // :213-229
		GtfRecord result;
		auto first = record.find_first_not_of("\t ");
		if (first == std::string::npos || record[first] == '#') return result;
		std::vector<std::string> col(split(record));
		if (col.size() < 4) throw std::runtime_error("Bed record is too short:\n" + record);
		return GtfRecord(col[0], col[3], "", strtoul(col[1].c_str(), nullptr, 10), strtoul(col[2].c_str(), nullptr, 10), EXON);
	}
	void save_transcript(GtfRecord record) {                                             // :93-108
		auto tr = transcript_positions[record.chr].insert(std::make_pair(record.transcript_id(), record));
		tr.first->second.merge(record);
		auto ex = exons_by_transcripts[record.chr].emplace(record.transcript_id(), IntervalsContainer<RecordType>(false));
		ex.first->second.add_interval(record.start, record.end, record.type);
		auto g = genes_by_transcripts.emplace(record.transcript_id(), record.gene_name());
		if (!g.second && g.first->second != record.gene_name())
			throw std::runtime_error("Different gene names (" + record.gene_name() + ", " + g.first->second + ") for the same transcript (" + record.transcript_id() + ")");
	}
	explicit RefGenes(const std::string &filename) {                                     // :23-91
		const std::string wrong = "Wrong genes file format: '" + filename + "'";
		if (filename.length() < 3) throw std::runtime_error(wrong);
		format = filename.substr(filename.length() - 3);
		if (format == ".gz") {
			if (filename.length() < 6) throw std::runtime_error(wrong);
			format = filename.substr(filename.length() - 6, 3);
		}
		if (format != "bed" && format != "gtf") throw std::runtime_error(wrong);
		gzFile f = gzopen(filename.c_str(), "rb");            // transparent for plain files
		if (!f) throw std::runtime_error("Can't open GTF file: '" + filename + "'");
		std::string line;
		char buf[1 << 16];
		auto handle = [&](const std::string &ln) {
			GtfRecord rec;
			try { rec = format == "gtf" ? parse_gtf_record(ln) : parse_bed_record(ln); }
			catch (std::runtime_error &) { return; }
			catch (std::out_of_range &) { throw; }           // record.at(0) on an empty line throws out of the loop in the reference too
			if (!rec.is_valid()) return;
			save_transcript(rec);
		};
		while (gzgets(f, buf, sizeof(buf))) {
			line += buf;
			if (!line.empty() && line.back() == '\n') { line.pop_back(); handle(line); line.clear(); }
		}
		if (!line.empty()) handle(line);
		gzclose(f);
		for (auto const &chr : transcript_positions) {
			auto &iv = transcript_intervals.emplace(chr.first, IntervalsContainer<std::string>(true)).first->second;
			for (auto const &tr : chr.second) {
				iv.add_interval(tr.second.start, tr.second.end, tr.first);
				exons_by_transcripts.at(chr.first).at(tr.first).set_initialized();
			}
			iv.set_initialized();
		}
	}
	struct ChrNotFound { std::string chr; };
	std::set<QueryResult> get_gene_info(const std::string &chr, coord_t s, coord_t e) const {   // :179-211
		if (e < s) return std::set<QueryResult>();
		auto it = transcript_intervals.find(chr);
		if (it == transcript_intervals.end()) throw ChrNotFound{chr};
		std::set<QueryResult> results;
		for (const std::string &tr : it->second.get_intervals(s, e)) {
			auto types = exons_by_transcripts.at(chr).at(tr).get_intervals(s, e);
			const std::string &gene = genes_by_transcripts.at(tr);
			if (types.empty() && !use_introns_from_gtf) { results.insert(QueryResult{gene, INTRON}); continue; }
			for (auto t : types) results.insert(QueryResult{gene, t});
		}
		return results;
	}
	static bool find_exon(const std::set<QueryResult> &qr, QueryResult &exon) {          // ReadParamsParser.cpp:153-172
		for (auto const &q : qr) {
			if (q.type != EXON) continue;
			if (exon.gene_name.empty()) { exon = q; continue; }
			if (exon.gene_name != q.gene_name) return false;
		}
		return true;
	}
	// ReadParamsParser::get_gene_from_reference (:92-151): position = 0-based alignment start, end_position = half-open
	// end of the alignment on the reference (BamAlignment::GetEndPosition()); returns the UMI::Mark bits
	int gene_from_reference(const std::string &chr, coord_t position, coord_t end_position, std::string &gene) const {
		enum { HAS_NOT_ANNOTATED = 1, HAS_EXONS = 2, HAS_INTRONS = 4 };
		auto bit = [](RecordType t) { if (t == EXON) return int(HAS_EXONS); if (t == INTRON) return int(HAS_INTRONS); throw std::runtime_error("Unexpected GtfRecord type"); };
		int mark = 0;
		gene.clear();
		auto s1 = get_gene_info(chr, position, position + 1);
		auto s2 = get_gene_info(chr, end_position - 1, end_position);
		if (s1.empty() && s2.empty()) return mark;
		if (s1.size() == 1 && s2.size() == 1) {
			if (s1.begin()->gene_name == s2.begin()->gene_name) { mark |= bit(s1.begin()->type) | bit(s2.begin()->type); gene = s1.begin()->gene_name; }
			return mark;
		}
		if (s1.size() <= 1 && s2.size() <= 1) {
			const QueryResult &ne = s1.empty() ? *s2.begin() : *s1.begin();
			gene = ne.gene_name;
			return mark | bit(ne.type) | HAS_NOT_ANNOTATED;
		}
		if (s1.empty() || s2.empty()) return mark;
		QueryResult e1{"", NONE}, e2{"", NONE};
		if (!find_exon(s1, e1)) return mark;
		if (!find_exon(s2, e2)) return mark;
		if (!e1.gene_name.empty() && !e2.gene_name.empty()) {
			if (e1.gene_name != e2.gene_name) return mark;
			gene = e1.gene_name;
			return mark | bit(e1.type) | bit(e2.type);
		}
		return mark;
	}
};

}  // namespace ga

static std::string g_ga_err;

extern "C" {

const char *orc_ga_last_error() { return g_ga_err.c_str(); }
void *orc_ga_load(const char *path) {
	try { return new ga::RefGenes(path); } catch (const std::exception &e) { g_ga_err = e.what(); return nullptr; }
}
void orc_ga_free(void *h) { delete static_cast<ga::RefGenes *>(h); }
uint64_t orc_ga_n_chromosomes(void *h) { return static_cast<ga::RefGenes *>(h)->transcript_intervals.size(); }
long orc_ga_n_homogeneous(void *h, const char *chr) {
	auto *g = static_cast<ga::RefGenes *>(h);
	auto it = g->transcript_intervals.find(chr);
	return it == g->transcript_intervals.end() ? -1 : long(it->second.homogeneous.size());
}
long orc_ga_homogeneous_labels(void *h, const char *chr, uint64_t i) {
	return long(static_cast<ga::RefGenes *>(h)->transcript_intervals.at(chr).homogeneous.at(i).labels.size());
}
// parse_gtf_record of one line: fills chr / gene id (cap bytes each) and the 0-based half-open extent; 0 = invalid record
int orc_ga_parse_gtf(void *h, const char *line, char *chr, char *gene_id, int cap, uint64_t *start, uint64_t *end) {
	try {
		ga::GtfRecord r = static_cast<ga::RefGenes *>(h)->parse_gtf_record(line);
		if (!r.is_valid()) return 0;
		std::strncpy(chr, r.chr.c_str(), size_t(cap)); std::strncpy(gene_id, r.gene_id_.c_str(), size_t(cap));
		*start = r.start; *end = r.end;
		return 1;
	} catch (const std::exception &e) { g_ga_err = e.what(); return -1; }
}
// get_gene_info: results in std::set order; names (stride bytes each) and types; returns the count, -1 = unknown chromosome
long orc_ga_query(void *h, const char *chr, uint64_t start, uint64_t end, char *names, int stride, int *types, int cap) {
	try {
		auto res = static_cast<ga::RefGenes *>(h)->get_gene_info(chr, start, end);
		long n = 0;
		for (auto const &q : res) {
			if (n < cap) { std::strncpy(names + size_t(stride) * size_t(n), q.gene_name.c_str(), size_t(stride)); types[n] = int(q.type); }
			++n;
		}
		return n;
	} catch (const ga::RefGenes::ChrNotFound &) { return -1; }
}
// get_gene_from_reference: returns the mark bits (>= 0), -1 = unknown chromosome
int orc_ga_gene_for_read(void *h, const char *chr, uint64_t position, uint64_t end_position, char *gene, int cap) {
	try {
		std::string g;
		const int mark = static_cast<ga::RefGenes *>(h)->gene_from_reference(chr, position, end_position, g);
		std::strncpy(gene, g.c_str(), size_t(cap));
		return mark;
	} catch (const ga::RefGenes::ChrNotFound &) { return -1; }
}

// IntervalsContainer<std::string> on its own (testGeneMerge, testInterval)
void *orc_iv_new() { return new ga::IntervalsContainer<std::string>(); }
void orc_iv_free(void *h) { delete static_cast<ga::IntervalsContainer<std::string> *>(h); }
int orc_iv_add(void *h, uint64_t s, uint64_t e, const char *label, int force) {
	try { static_cast<ga::IntervalsContainer<std::string> *>(h)->add_interval(s, e, label, force != 0); return 0; }
	catch (const std::exception &ex) { g_ga_err = ex.what(); return -1; }
}
void orc_iv_set_initialized(void *h, int clear) { static_cast<ga::IntervalsContainer<std::string> *>(h)->set_initialized(clear != 0); }
uint64_t orc_iv_n_homogeneous(void *h) { return static_cast<ga::IntervalsContainer<std::string> *>(h)->homogeneous.size(); }
void orc_iv_homogeneous(void *h, uint64_t i, uint64_t *s, uint64_t *e) {
	auto &q = static_cast<ga::IntervalsContainer<std::string> *>(h)->homogeneous.at(i);
	*s = q.start; *e = q.end;
}
long orc_iv_query(void *h, uint64_t s, uint64_t e, char *labels, int stride, int cap) {
	auto res = static_cast<ga::IntervalsContainer<std::string> *>(h)->get_intervals(s, e);
	long n = 0;
	for (auto const &l : res) { if (n < cap) std::strncpy(labels + size_t(stride) * size_t(n), l.c_str(), size_t(stride)); ++n; }
	return n;
}

}  // extern "C"
