// bam_ingest.cpp -- see bam_ingest.h.
#include "bam_ingest.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <future>
#include <stdexcept>
#include <thread>
#include <unordered_set>

namespace Estimation {
namespace BamProcessing {

namespace {

inline uint16_t le16(const uint8_t *p) { return uint16_t(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t *p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }

struct RawBlock { std::vector<uint8_t> cdata; uint32_t isize = 0, crc = 0; };

// one BGZF block (SAMv1 §4.1): gzip member, FEXTRA with subfield 'B','C' = total block size - 1
bool read_block(FILE *f, RawBlock &b, const std::string &path) {
	uint8_t h[12];
	const size_t got = fread(h, 1, 12, f);
	if (got == 0) return false;
	if (got != 12 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("Not a BGZF/BAM file: " + path);
	const uint16_t xlen = le16(h + 10);
	std::vector<uint8_t> extra(xlen);
	if (fread(extra.data(), 1, xlen, f) != xlen) throw std::runtime_error("Truncated BGZF header: " + path);
	int bsize = -1;
	for (size_t o = 0; o + 4 <= extra.size();) {
		const uint16_t slen = le16(extra.data() + o + 2);
		if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= extra.size()) bsize = le16(extra.data() + o + 4);
		o += 4u + slen;
	}
	if (bsize < 0) throw std::runtime_error("BGZF block without BC subfield: " + path);
	const long clen = long(bsize) - long(xlen) - 19;
	if (clen < 0) throw std::runtime_error("Corrupt BGZF block size: " + path);
	b.cdata.resize(size_t(clen));
	uint8_t tail[8];
	if (fread(b.cdata.data(), 1, size_t(clen), f) != size_t(clen) || fread(tail, 1, 8, f) != 8) throw std::runtime_error("Truncated BGZF block: " + path);
	b.crc = le32(tail); b.isize = le32(tail + 4);
	return true;
}

void inflate_block(const RawBlock &b, uint8_t *out) {
	if (b.isize == 0) return;
	z_stream zs;
	std::memset(&zs, 0, sizeof(zs));
	if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib: inflateInit2 failed");
	zs.next_in = const_cast<Bytef *>(b.cdata.data()); zs.avail_in = uInt(b.cdata.size());
	zs.next_out = out; zs.avail_out = b.isize;
	const int rc = inflate(&zs, Z_FINISH);
	inflateEnd(&zs);
	if (rc != Z_STREAM_END || zs.total_out != b.isize) throw std::runtime_error("Corrupt BGZF block (inflate)");
	if (crc32(crc32(0L, Z_NULL, 0), out, b.isize) != b.crc) throw std::runtime_error("Corrupt BGZF block (CRC)");
}

}  // namespace

struct BamReader::Impl {
	std::string path;
	FILE *f = nullptr;
	unsigned threads = 1;
	static constexpr size_t BATCH_BLOCKS = 512;        // <= 32 MB of BAM per batch
	std::vector<uint8_t> data;                         // decompressed window
	size_t pos = 0;
	bool file_done = false;
	std::future<std::vector<uint8_t>> ahead;           // next batch, being inflated while the caller parses this one
	std::vector<std::string> refs;
	std::string text;

	std::vector<uint8_t> load_batch() {
		std::vector<RawBlock> blocks;
		blocks.reserve(BATCH_BLOCKS);
		while (blocks.size() < BATCH_BLOCKS) {
			RawBlock b;
			if (!read_block(f, b, path)) { file_done = true; break; }
			blocks.push_back(std::move(b));
		}
		std::vector<size_t> off(blocks.size() + 1, 0);
		for (size_t i = 0; i < blocks.size(); ++i) off[i + 1] = off[i] + blocks[i].isize;
		std::vector<uint8_t> out(off.back());
		const unsigned nt = unsigned(std::min<size_t>(threads, std::max<size_t>(1, blocks.size() / 8)));
		std::vector<std::thread> pool;
		std::vector<std::string> errors(nt);
		for (unsigned t = 0; t < nt; ++t)
			pool.emplace_back([&, t] {
				try { for (size_t i = t; i < blocks.size(); i += nt) inflate_block(blocks[i], out.data() + off[i]); }
				catch (const std::exception &e) { errors[t] = e.what(); }
			});
		for (auto &th : pool) th.join();
		for (auto const &e : errors) if (!e.empty()) throw std::runtime_error(e + ": " + path);
		return out;
	}
	// makes at least `need` bytes available at data[pos..]; false if the stream ends first
	bool ensure(size_t need) {
		while (data.size() - pos < need) {
			std::vector<uint8_t> next;
			if (ahead.valid()) next = ahead.get();
			else if (!file_done) next = load_batch();
			if (next.empty() && file_done && !ahead.valid()) return false;
			if (!file_done) ahead = std::async(std::launch::async, [this] { return load_batch(); });
			if (pos) { data.erase(data.begin(), data.begin() + long(pos)); pos = 0; }
			data.insert(data.end(), next.begin(), next.end());
		}
		return true;
	}
};

BamReader::BamReader(const std::string &path, unsigned threads) : impl(new Impl()) {
	impl->path = path;
	impl->threads = threads ? threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	impl->f = fopen(path.c_str(), "rb");
	if (!impl->f) { delete impl; throw std::runtime_error("Can't open BAM file: " + path); }
	try {
		Impl &m = *impl;
		if (!m.ensure(12) || std::memcmp(m.data.data() + m.pos, "BAM\1", 4) != 0) throw std::runtime_error("Can't open BAM file: " + path);
		const uint32_t l_text = le32(m.data.data() + m.pos + 4);
		if (!m.ensure(12 + size_t(l_text))) throw std::runtime_error("Truncated BAM header: " + path);
		m.text.assign(reinterpret_cast<const char *>(m.data.data() + m.pos + 8), l_text);
		const uint32_t n_ref = le32(m.data.data() + m.pos + 8 + l_text);
		m.pos += 12 + size_t(l_text);
		for (uint32_t r = 0; r < n_ref; ++r) {
			if (!m.ensure(4)) throw std::runtime_error("Truncated BAM header: " + path);
			const uint32_t l_name = le32(m.data.data() + m.pos);
			if (!m.ensure(8 + size_t(l_name))) throw std::runtime_error("Truncated BAM header: " + path);
			m.refs.emplace_back(reinterpret_cast<const char *>(m.data.data() + m.pos + 4), l_name ? l_name - 1 : 0);
			m.pos += 8 + size_t(l_name);
		}
	} catch (...) {
		if (impl->ahead.valid()) impl->ahead.wait();
		fclose(impl->f); delete impl; throw;
	}
}

BamReader::~BamReader() {
	if (impl->ahead.valid()) { try { impl->ahead.get(); } catch (...) {} }
	if (impl->f) fclose(impl->f);
	delete impl;
}

const std::vector<std::string> &BamReader::reference_names() const { return impl->refs; }
const std::string &BamReader::header_text() const { return impl->text; }

bool BamReader::next(BamRecord &rec) {
	Impl &m = *impl;
	if (!m.ensure(4)) return false;
	const uint32_t block_size = le32(m.data.data() + m.pos);
	if (block_size < 32) throw std::runtime_error("Corrupt BAM record: " + m.path);
	if (!m.ensure(4 + size_t(block_size))) throw std::runtime_error("Truncated BAM record: " + m.path);
	const uint8_t *p = m.data.data() + m.pos + 4;
	rec.ref_id = int32_t(le32(p));
	const uint32_t l_read_name = p[8];
	const uint32_t n_cigar = le16(p + 12);
	rec.flag = le16(p + 14);
	const uint32_t l_seq = le32(p + 16);
	const size_t fixed = 32, name_end = fixed + l_read_name;
	const size_t aux = name_end + size_t(n_cigar) * 4 + (size_t(l_seq) + 1) / 2 + l_seq;
	if (aux > block_size) throw std::runtime_error("Corrupt BAM record: " + m.path);
	rec.name.assign(reinterpret_cast<const char *>(p + fixed), l_read_name ? l_read_name - 1 : 0);
	rec.tags = p + aux; rec.tags_size = block_size - aux;
	m.pos += 4 + size_t(block_size);
	return true;
}

bool BamRecord::get_string_tag(const std::string &tag, std::string &value, char *type_out) const {
	if (tag.size() != 2) return false;
	size_t o = 0;
	while (o + 3 <= tags_size) {
		const char t0 = char(tags[o]), t1 = char(tags[o + 1]), type = char(tags[o + 2]);
		o += 3;
		size_t len = 0;
		bool text = false;
		switch (type) {
			case 'A': case 'c': case 'C': len = 1; break;
			case 's': case 'S': len = 2; break;
			case 'i': case 'I': case 'f': len = 4; break;
			case 'Z': case 'H': { const void *e = std::memchr(tags + o, 0, tags_size - o); if (!e) return false; len = size_t(static_cast<const uint8_t *>(e) - (tags + o)) + 1; text = true; break; }
			case 'B': {
				if (o + 5 > tags_size) return false;
				const char sub = char(tags[o]);
				const uint32_t cnt = le32(tags + o + 1);
				const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
				len = 5 + size_t(cnt) * w;
				break;
			}
			default: return false;   // unknown type: cannot skip safely
		}
		if (o + len > tags_size) return false;
		if (t0 == tag[0] && t1 == tag[1]) {
			if (type_out) *type_out = type;
			if (text) { value.assign(reinterpret_cast<const char *>(tags + o), len - 1); return true; }
			if (type == 'A') { value.assign(1, char(tags[o])); return true; }
			return false;    // numeric tag: not a string (BamTools would read past it; no caller relies on that)
		}
		o += len;
	}
	return false;
}

BamController::BamController(const BamTags &tags, bool filled_bam, const std::string &read_param_filenames, const std::string &gtf_path,
                             bool gene_in_chromosome_name, int min_barcode_phred, unsigned threads)
	: _tags(tags), _filled_bam(filled_bam), _gene_in_chromosome_name(gene_in_chromosome_name), _min_barcode_phred(min_barcode_phred),
	  _threads(threads) {
	if (!gtf_path.empty()) throw std::runtime_error("gene annotation from a GTF (-g) is not built: the BAM must carry gene tags");
	if (!read_param_filenames.empty()) throw std::runtime_error("read-parameter files (-r) are not built");
	if (!_tags.read_type.empty() && _tags.intronic_read_value.empty())
		throw std::runtime_error("You have to specify tag values to be able to parse info about read types (see conf_desc.xml \"Estimation/BamTags/Type/\")");
}

void BamController::parse_bam_files(const std::vector<std::string> &bam_files, CellsDataContainer &container) {
	const int quality_offset = 33;                                       // Tools::ReadParameters::quality_offset
	for (auto const &bam_name : bam_files) {
		BamReader reader(bam_name, _threads);
		const auto &refs = reader.reference_names();
		BamRecord al;
		std::string cb, umi, cbq, umiq, gene, read_type;
		while (reader.next(al)) {
			if (!al.is_mapped() || !al.is_primary()) continue;            // BamController.cpp:87-88
			if (al.ref_id < 0 || size_t(al.ref_id) >= refs.size()) { ++_counters.cant_parse; continue; }   // :90-104
			const std::string &chr_name = refs[size_t(al.ref_id)];
			++_counters.total_reads;
			// get_read_params
			cbq.clear(); umiq.clear();
			bool pass_quality = true;
			if (_filled_bam) {                                            // FilledBamParamsParser.cpp:12-40
				if (!al.get_string_tag(_tags.cell_barcode, cb) || !al.get_string_tag(_tags.umi, umi)) { ++_counters.cant_parse; continue; }
				al.get_string_tag(_tags.cell_barcode_quality, cbq);
				al.get_string_tag(_tags.umi_quality, umiq);
				if (cb.empty() || umi.empty()) { ++_counters.cant_parse; continue; }   // ReadParameters ctor throws -> false
				if (_min_barcode_phred > quality_offset) {                 // ReadParameters::check_quality (:118-136)
					for (char q : cbq) pass_quality &= q >= char(_min_barcode_phred);
					for (char q : umiq) pass_quality &= q >= char(_min_barcode_phred);
				}
			} else {                                                      // ReadParamsParser.cpp:20-33: "id!CB#UMI"
				const size_t up = al.name.rfind('#');
				const size_t cp = up == std::string::npos ? std::string::npos : al.name.rfind('!', up);
				if (up == std::string::npos || cp == std::string::npos) { ++_counters.cant_parse; continue; }
				cb = al.name.substr(cp + 1, up - cp - 1); umi = al.name.substr(up + 1);
				if (cb.empty() || umi.empty()) { ++_counters.cant_parse; continue; }
				// parse_encoded_id builds ReadParameters(cb, umi, "", "") = min_phred_score 0: always passes (ReadParameters.cpp:42-56)
			}
			if (!pass_quality) { ++_counters.low_quality; continue; }
			// get_gene (ReadParamsParser.cpp:36-65) + parse_read_type (:67-90)
			UMI::Mark mark;
			gene.clear();
			if (_gene_in_chromosome_name) {
				gene = chr_name;
				if (!chr_name.empty()) mark.add(UMI::Mark::HAS_EXONS);
			} else if (!al.get_string_tag(_tags.gene, gene)) {
				gene.clear();
				mark.add(UMI::Mark::HAS_NOT_ANNOTATED);
			} else {
				char type = 0;
				if (_tags.read_type.empty() || !al.get_string_tag(_tags.read_type, read_type, &type)) mark.add(UMI::Mark::HAS_EXONS);
				else if (read_type == _tags.intronic_read_value) mark.add(UMI::Mark::HAS_INTRONS);
				else if (!_tags.intergenic_read_value.empty() && read_type == _tags.intergenic_read_value) mark.add(UMI::Mark::HAS_NOT_ANNOTATED);
				else mark.add(UMI::Mark::HAS_EXONS);
			}
			container.add_record(ReadInfo(Tools::ReadParameters(cb, umi, cbq, umiq), gene, chr_name, mark));
			++_counters.saved;
		}
	}
}

}  // namespace BamProcessing
}  // namespace Estimation
