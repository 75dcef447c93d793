// gene_annotation.cpp -- see gene_annotation.h.
#include "gene_annotation.h"

#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <set>
#include <sstream>

namespace Tools {
namespace GeneAnnotation {

namespace {

struct Record {
	std::string chr, gene_id, gene_name, transcript;
	size_t start = 0, end = 0;
	GtfRecord::RecordType type = GtfRecord::NONE;
	bool valid() const { return !gene_id.empty(); }
	const std::string &name() const { return gene_name.empty() ? gene_id : gene_name; }            // GtfRecord::gene_name
	const std::string &transcript_id() const { return transcript.empty() ? gene_id : transcript; }  // GtfRecord::transcript_id
};

std::vector<std::string> split(const std::string &s) {   // RefGenesContainer::split: whitespace-separated columns
	std::istringstream is(s);
	std::vector<std::string> out;
	std::string c;
	while (is >> c) out.push_back(c);
	return out;
}

// RefGenesContainer::parse_gtf_record (RefGenesContainer.cpp:110-177)
Record parse_gtf(const std::string &line, bool &use_introns, bool &has_transcripts) {
	Record r;
	if (line.at(0) == '#') return r;
	const std::vector<std::string> col = split(line);
	if (col.size() < 9) throw std::runtime_error("Can't parse record: \n" + line);
	if (col[0] == "." || col[3] == "." || col[4] == "." || col.size() == 9) return r;
	if (col[2] == "exon") r.type = GtfRecord::EXON;
	else if (col[2] == "intron") { r.type = GtfRecord::INTRON; use_introns = true; }
	else return r;
	std::string id, name, transcript;
	for (size_t a = 8; a + 1 < col.size(); ++a) {
		const std::string &key = col[a], &value = col[a + 1];          // "VALUE";  -> VALUE
		if (key == "gene_id") id = value.substr(1, value.length() - 3);
		if (key == "gene_name") name = value.substr(1, value.length() - 3);
		if (key == "transcript_id") transcript = value.substr(1, value.length() - 3);
	}
	if (transcript.empty()) has_transcripts = false;
	if (id.empty()) {
		if (name.empty()) throw std::runtime_error("GTF record doesn't contain either gene name or id:\n" + line);
		id = name;
	}
	r.chr = col[0]; r.gene_id = id; r.gene_name = name == id ? "" : name; r.transcript = transcript;
	r.start = strtoul(col[3].c_str(), nullptr, 10) - 1;               // GTF is 1-based, closed
	r.end = strtoul(col[4].c_str(), nullptr, 10);
	return r;
}

// RefGenesContainer::parse_bed_record (:213-229)
Record parse_bed(const std::string &line) {
	Record r;
	const size_t first = line.find_first_not_of("\t ");
	if (first == std::string::npos || line[first] == '#') return r;
	const std::vector<std::string> col = split(line);
	if (col.size() < 4) throw std::runtime_error("Bed record is too short:\n" + line);
	r.chr = col[0]; r.gene_id = col[3]; r.type = GtfRecord::EXON;
	r.start = strtoul(col[1].c_str(), nullptr, 10); r.end = strtoul(col[2].c_str(), nullptr, 10);
	return r;
}

// What the reference's IntervalsContainer computes, including where it is NOT the union of the intervals of a label:
// add_interval (IntervalsContainer.h:151-190) keeps one list per label and merges a new interval only with the run of
// list neighbours it meets first (touching counts on one side only), so lists can end up unsorted with overlapping
// entries; set_initialized (:102-142, :192-210) then sweeps open / close events with a std::set of labels, so the
// close of one of two overlapping same-label intervals drops the label although the other is still open.  refFlat-style
// BED files (all isoforms of a gene under one label, in file order) do hit this, so it is reproduced step by step:
// the result is the list of "homogeneous" pieces, which the container below flattens for binary search.
struct RawSpan { size_t start, end; };
template <class Label> struct Piece { size_t start, end; std::vector<Label> labels; };

template <class Label>
class Sweep {
	std::map<Label, std::list<RawSpan>> _base;
	static bool intercept(const RawSpan &a, const RawSpan &b) { return a.start <= b.end && a.end > b.start; }   // Interval::is_intercept
	static void merge(RawSpan &a, const RawSpan &b) { a.start = std::min(a.start, b.start); a.end = std::max(a.end, b.end); }
public:
	void add(size_t start, size_t end, const Label &label) {
		RawSpan query{start, end};
		std::list<RawSpan> &cur = _base[label];
		auto it = cur.begin();
		while (it != cur.end() && !intercept(query, *it)) {
			if (it->start > query.end) { cur.insert(it, query); return; }
			++it;
		}
		if (it == cur.end()) { cur.push_back(query); return; }
		auto end_it = it;
		++end_it;
		while (end_it != cur.end() && intercept(query, *end_it)) { merge(query, *end_it); ++end_it; }
		merge(*it, query);
		++it;
		cur.erase(it, end_it);
	}
	std::vector<Piece<Label>> pieces(bool allow_intercepts, unsigned min_len = 1) const {
		struct Event { size_t pos; const Label *label; bool open; };
		std::vector<Event> events;                      // insertion order = the multimap's order among equal positions
		for (auto const &l : _base)
			for (auto const &sp : l.second) { events.push_back(Event{sp.start, &l.first, true}); events.push_back(Event{sp.end, &l.first, false}); }
		std::stable_sort(events.begin(), events.end(), [](const Event &a, const Event &b) { return a.pos < b.pos; });
		std::vector<Piece<Label>> out;
		std::set<Label> cur;
		size_t start_pos = 0;
		for (const Event &ev : events) {
			const size_t end_pos = ev.pos;
			if (!cur.empty() && end_pos - start_pos >= min_len) {
				if (!allow_intercepts && cur.size() > 1)
					throw std::runtime_error("Intervals intersection at (" + std::to_string(start_pos) + ", " + std::to_string(end_pos) + ")");
				out.push_back(Piece<Label>{start_pos, end_pos, std::vector<Label>(cur.begin(), cur.end())});
			}
			if (ev.open) cur.insert(*ev.label); else cur.erase(*ev.label);
			start_pos = end_pos;
		}
		return out;
	}
};

template <class V>
bool intersects(const V &spans, size_t s, size_t e) {   // some span with end > s and start < e (IntervalsContainer::get_intervals, :216-238)
	auto it = std::lower_bound(spans.begin(), spans.end(), s, [](const auto &sp, size_t pos) { return sp.end <= pos; });
	return it != spans.end() && it->start < e;
}

}  // namespace

RefGenesContainer::RefGenesContainer(const std::string &genes_filename) : _is_empty(false) {
	const std::string wrong = "Wrong genes file format: '" + genes_filename + "'";
	if (genes_filename.length() < 3) throw std::runtime_error(wrong);
	std::string format = genes_filename.substr(genes_filename.length() - 3);
	if (format == ".gz") {
		if (genes_filename.length() < 6) throw std::runtime_error(wrong);
		format = genes_filename.substr(genes_filename.length() - 6, 3);
	}
	if (format != "bed" && format != "gtf") throw std::runtime_error(wrong);
	gzFile f = gzopen(genes_filename.c_str(), "rb");   // reads plain files as they are
	if (!f) throw std::runtime_error("Can't open GTF file: '" + genes_filename + "'");

	// transcripts of one chromosome while loading (ordered by id like the reference's std::map: the order of the
	// transcripts is not observable, the merged extents and the gene of each are)
	struct Loading { std::string gene; size_t start, end; Sweep<GtfRecord::RecordType> exons; };
	std::map<std::string, std::map<std::string, Loading>> by_chr;
	std::unordered_map<std::string, std::string> gene_of_transcript;      // across chromosomes (RefGenesContainer.cpp:103-107)
	auto handle = [&](const std::string &line) {
		Record rec;
		try { rec = format == "gtf" ? parse_gtf(line, _use_introns_from_gtf, _gtf_has_transcripts) : parse_bed(line); }
		catch (std::runtime_error &) { return; }                            // logged and skipped by the reference (:71-75)
		if (!rec.valid()) return;
		auto ins = by_chr[rec.chr].emplace(rec.transcript_id(), Loading{rec.name(), rec.start, rec.end, {}});
		Loading &t = ins.first->second;
		t.start = std::min(t.start, rec.start); t.end = std::max(t.end, rec.end);   // transcript_iter.first->second.merge(record) (:95-97)
		t.exons.add(rec.start, rec.end, rec.type);                                     // exon_iter.first->second.add_interval (:99-102)
		auto g = gene_of_transcript.emplace(rec.transcript_id(), rec.name());
		if (!g.second && g.first->second != rec.name())
			throw std::runtime_error("Different gene names (" + rec.name() + ", " + g.first->second + ") for the same transcript (" +
			                         rec.transcript_id() + ")");
	};
	try {
		std::string line;
		char buf[1 << 16];
		while (gzgets(f, buf, sizeof(buf))) {
			line += buf;
			if (!line.empty() && line.back() == '\n') { line.pop_back(); handle(line); line.clear(); }
		}
		if (!line.empty()) handle(line);
	} catch (...) { gzclose(f); throw; }
	gzclose(f);

	std::unordered_map<std::string, uint32_t> gene_index;
	for (auto &chr : by_chr) {
		Chromosome &c = _chromosomes[chr.first];
		for (auto &kv : chr.second) {
			Loading &l = kv.second;
			Transcript t;
			t.id = kv.first; t.start = l.start; t.end = l.end;
			auto gi = gene_index.emplace(l.gene, uint32_t(_genes.size()));
			if (gi.second) _genes.push_back(l.gene);
			t.gene = gi.first->second;
			// the exon container of the transcript (allow_intercepts = false: exon and intron records may not overlap)
			for (auto const &pc : l.exons.pieces(false))
				for (GtfRecord::RecordType ty : pc.labels) (ty == GtfRecord::EXON ? t.exons : t.introns).push_back(Span{pc.start, pc.end});
			c.transcripts.push_back(std::move(t));
		}
		// the transcript container of the chromosome: transcripts enter in id order (the reference walks a std::map)
		Sweep<uint32_t> extents;
		std::vector<uint32_t> by_id(c.transcripts.size());
		for (uint32_t ti = 0; ti < by_id.size(); ++ti) by_id[ti] = ti;      // by_chr's inner std::map already iterates in id order
		for (uint32_t ti : by_id) extents.add(c.transcripts[ti].start, c.transcripts[ti].end, ti);
		const auto pieces = extents.pieces(true);
		const size_t n_seg = pieces.size();
		std::vector<std::vector<uint32_t>> cover(n_seg);
		for (size_t sgm = 0; sgm < n_seg; ++sgm) { c.seg_start.push_back(pieces[sgm].start); c.seg_end.push_back(pieces[sgm].end); cover[sgm] = pieces[sgm].labels; }
		c.seg_begin.assign(n_seg + 1, 0);
		for (size_t s = 0; s < n_seg; ++s) { c.seg_begin[s + 1] = c.seg_begin[s] + uint32_t(cover[s].size()); c.seg_transcripts.insert(c.seg_transcripts.end(), cover[s].begin(), cover[s].end()); }
	}
}

void RefGenesContainer::collect(const Chromosome &c, pos_t start, pos_t end, query_results_t &out) const {
	// first piece whose end is beyond `start`, then every piece that begins before `end` (IntervalsContainer.h:216-238)
	size_t seg = size_t(std::lower_bound(c.seg_end.begin(), c.seg_end.end(), start, [](pos_t piece_end, pos_t pos) { return piece_end <= pos; }) - c.seg_end.begin());
	uint32_t seen_small[16];
	size_t n_seen = 0;
	std::vector<uint32_t> seen_big;
	for (; seg < c.seg_start.size() && c.seg_start[seg] < end; ++seg) {
		for (uint32_t k = c.seg_begin[seg]; k < c.seg_begin[seg + 1]; ++k) {
			const uint32_t ti = c.seg_transcripts[k];
			bool dup = false;
			for (size_t j = 0; j < n_seen && !dup; ++j) dup = seen_small[j] == ti;
			for (size_t j = 0; j < seen_big.size() && !dup; ++j) dup = seen_big[j] == ti;
			if (dup) continue;
			if (n_seen < 16) seen_small[n_seen++] = ti; else seen_big.push_back(ti);
			const Transcript &t = c.transcripts[ti];
			const bool ex = intersects(t.exons, start, end), in = intersects(t.introns, start, end);
			const std::string &gene = _genes[t.gene];
			if (!ex && !in) { if (!_use_introns_from_gtf) out.emplace(gene, GtfRecord::INTRON); continue; }   // RefGenesContainer.cpp:198-202
			if (ex) out.emplace(gene, GtfRecord::EXON);
			if (in) out.emplace(gene, GtfRecord::INTRON);
		}
	}
}

RefGenesContainer::query_results_t RefGenesContainer::get_gene_info(const std::string &chr_name, pos_t start_pos, pos_t end_pos) const {
	query_results_t results;
	if (end_pos < start_pos) return results;
	auto it = _chromosomes.find(chr_name);
	if (it == _chromosomes.end()) throw ChrNotFoundException(chr_name);
	collect(it->second, start_pos, end_pos, results);
	return results;
}

int RefGenesContainer::gene_of_alignment(const std::string &chr_name, pos_t position, pos_t end_position, std::string &gene) const {
	enum { HAS_NOT_ANNOTATED = 1, HAS_EXONS = 2, HAS_INTRONS = 4 };    // UMI::Mark (UMI.h:16-22), Mark::add(RecordType) (UMI.cpp:87-100)
	auto bit = [](GtfRecord::RecordType t) {
		if (t == GtfRecord::EXON) return int(HAS_EXONS);
		if (t == GtfRecord::INTRON) return int(HAS_INTRONS);
		throw std::runtime_error("Unexpected GtfRecord type: " + std::to_string(int(t)));
	};
	gene.clear();
	auto it = _chromosomes.find(chr_name);
	if (it == _chromosomes.end()) throw ChrNotFoundException(chr_name);
	query_results_t s1, s2;                                            // TODO of the reference: no CIGAR, two end points
	collect(it->second, position, position + 1, s1);
	if (end_position >= 1) collect(it->second, end_position - 1, end_position, s2);
	int mark = 0;
	if (s1.empty() && s2.empty()) return mark;
	if (s1.size() == 1 && s2.size() == 1) {
		if (s1.begin()->gene_name == s2.begin()->gene_name) { mark |= bit(s1.begin()->type) | bit(s2.begin()->type); gene = s1.begin()->gene_name; }
		return mark;
	}
	if (s1.size() <= 1 && s2.size() <= 1) {
		const QueryResult &non_empty = s1.empty() ? *s2.begin() : *s1.begin();
		gene = non_empty.gene_name;
		return mark | bit(non_empty.type) | HAS_NOT_ANNOTATED;
	}
	if (s1.empty() || s2.empty()) return mark;
	auto find_exon = [](const query_results_t &qr, QueryResult &exon) {   // ReadParamsParser::find_exon (:153-172)
		for (auto const &q : qr) {
			if (q.type != GtfRecord::EXON) continue;
			if (exon.gene_name.empty()) { exon = q; continue; }
			if (exon.gene_name != q.gene_name) return false;
		}
		return true;
	};
	QueryResult e1, e2;
	if (!find_exon(s1, e1) || !find_exon(s2, e2)) return mark;
	if (!e1.gene_name.empty() && !e2.gene_name.empty()) {
		if (e1.gene_name != e2.gene_name) return mark;
		gene = e1.gene_name;
		return mark | bit(e1.type) | bit(e2.type);
	}
	return mark;
}

RefGenesContainer::Flat RefGenesContainer::flatten() const {
	Flat f;
	f.gene_names = _genes;
	f.use_introns_from_gtf = _use_introns_from_gtf;
	auto u32 = [](pos_t v) { if (v > 0xFFFFFFFFull) throw std::runtime_error("annotation position beyond 32 bits"); return uint32_t(v); };
	f.chr_seg_begin.push_back(0); f.seg_tr_begin.push_back(0); f.tr_exon_begin.push_back(0); f.tr_intron_begin.push_back(0);
	for (auto const &kv : _chromosomes) {
		f.chr_names.push_back(kv.first);
		const Chromosome &c = kv.second;
		const uint32_t tr0 = uint32_t(f.tr_gene.size());
		for (const Transcript &t : c.transcripts) {
			f.tr_gene.push_back(t.gene);
			for (const Span &sp : t.exons) { f.exon_start.push_back(u32(sp.start)); f.exon_end.push_back(u32(sp.end)); }
			for (const Span &sp : t.introns) { f.intron_start.push_back(u32(sp.start)); f.intron_end.push_back(u32(sp.end)); }
			f.tr_exon_begin.push_back(uint32_t(f.exon_start.size())); f.tr_intron_begin.push_back(uint32_t(f.intron_start.size()));
		}
		for (size_t sgm = 0; sgm < c.seg_start.size(); ++sgm) {
			f.seg_start.push_back(u32(c.seg_start[sgm])); f.seg_end.push_back(u32(c.seg_end[sgm]));
			for (uint32_t k = c.seg_begin[sgm]; k < c.seg_begin[sgm + 1]; ++k) f.seg_tr.push_back(tr0 + c.seg_transcripts[k]);
			f.seg_tr_begin.push_back(uint32_t(f.seg_tr.size()));
		}
		f.chr_seg_begin.push_back(uint32_t(f.seg_start.size()));
	}
	return f;
}

}  // namespace GeneAnnotation
}  // namespace Tools

// ---- plain-C access -------------------------------------------------------------------------------------------
static thread_local std::string g_ga_error;
using Tools::GeneAnnotation::RefGenesContainer;

extern "C" {

const char *dropest_gene_annotation_error(void) { return g_ga_error.c_str(); }
void *dropest_gene_annotation_load(const char *path) {
	try { return new RefGenesContainer(path); } catch (const std::exception &e) { g_ga_error = e.what(); return nullptr; }
}
void dropest_gene_annotation_free(void *h) { delete static_cast<RefGenesContainer *>(h); }
long dropest_gene_annotation_query(void *h, const char *chr, uint64_t start, uint64_t end, char *names, int stride, int *types, int cap) {
	try {
		const auto res = static_cast<RefGenesContainer *>(h)->get_gene_info(chr, start, end);
		long n = 0;
		for (auto const &q : res) {
			if (n < cap) { std::strncpy(names + size_t(stride) * size_t(n), q.gene_name.c_str(), size_t(stride)); types[n] = int(q.type); }
			++n;
		}
		return n;
	} catch (const RefGenesContainer::ChrNotFoundException &) { return -1; }
}
namespace {
struct FlatCache { void *owner = nullptr; RefGenesContainer::Flat flat; };
thread_local FlatCache g_flat;
const RefGenesContainer::Flat &flat_of(void *h) {
	if (g_flat.owner != h) { g_flat.flat = static_cast<RefGenesContainer *>(h)->flatten(); g_flat.owner = h; }
	return g_flat.flat;
}
}  // namespace
void dropest_gene_annotation_flat_sizes(void *h, uint32_t sizes[8]) {
	const auto &f = flat_of(h);
	sizes[0] = uint32_t(f.chr_names.size()); sizes[1] = uint32_t(f.seg_start.size()); sizes[2] = uint32_t(f.tr_gene.size());
	sizes[3] = uint32_t(f.gene_names.size()); sizes[4] = uint32_t(f.seg_tr.size()); sizes[5] = uint32_t(f.exon_start.size());
	sizes[6] = uint32_t(f.intron_start.size()); sizes[7] = f.use_introns_from_gtf ? 1u : 0u;
}
void dropest_gene_annotation_flat_fill(void *h, uint32_t *const arrays[12]) {
	const auto &f = flat_of(h);
	const std::vector<uint32_t> *src[12] = {&f.chr_seg_begin, &f.seg_start, &f.seg_end, &f.seg_tr_begin, &f.seg_tr, &f.tr_gene, &f.tr_exon_begin,
	                                        &f.tr_intron_begin, &f.exon_start, &f.exon_end, &f.intron_start, &f.intron_end};
	for (int i = 0; i < 12; ++i) std::copy(src[i]->begin(), src[i]->end(), arrays[i]);
}
const char *dropest_gene_annotation_chr_name(void *h, uint32_t i) { return flat_of(h).chr_names.at(i).c_str(); }
const char *dropest_gene_annotation_gene_name(void *h, uint32_t i) { return flat_of(h).gene_names.at(i).c_str(); }
int dropest_gene_annotation_read(void *h, const char *chr, uint64_t position, uint64_t end_position, char *gene, int cap) {
	try {
		std::string g;
		const int mark = static_cast<RefGenesContainer *>(h)->gene_of_alignment(chr, position, end_position, g);
		std::strncpy(gene, g.c_str(), size_t(cap));
		return mark;
	} catch (const RefGenesContainer::ChrNotFoundException &) { return -1; }
}

}  // extern "C"
