// k_bamparse.h -- BAM records of an inflated window on the device: where they start, and the fields the Estimation path looks at.
//
// What it replaces (with k_inflate.h in front): BamReader::GetNextAlignment + BamController::process_alignment
// (Estimation/BamProcessing/BamController.cpp:85-172) up to the point where the dictionaries are asked:
//   FilledBamParamsParser::get_read_params  (-f: CB / UB tags, FilledBamParamsParser.cpp:12-40)
//   ReadParamsParser::get_read_params       (read name "id!CB#UMI", ReadParamsParser.cpp:20-33)
//   ReadParamsParser::get_gene / parse_read_type (gene tag + optional read-type tag, :36-90)
//   Tools::ReadParameters::check_quality    (Tools/ReadParameters.cpp:118-136)
// The BAM stream (SAMv1 §4.2) is one chain of records, each naming its own length, cut into BGZF blocks anywhere.  The chain is
// found in parallel by guessing: the window is cut into segments of 16 KB; for every segment a wave looks for the first offset at which
// four plausible records follow each other (bam_seg_guess); a lane per segment then walks its records from that offset
// (bam_seg_walk) and the host checks that every walk ends where the next guess stands -- a wrong guess is replaced by the
// predecessor's end and walked again, so the result is the true chain whatever the guesses were.
// One lane per record then reads the fixed fields and walks the tags once (bam_parse): 2-bit codes of barcode and UMI (0 = the host
// packs it: an N, more than 31 bases), FNV-1a of the gene name (the hash of host/facade.cpp: the gene dictionary stays on the host),
// UMI::Mark, the reader's status.  Byte work, latency-bound, massively parallel; no MFMA.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dropest {

constexpr uint32_t BAM_SEG = 16384;
constexpr uint64_t BAM_NONE = ~0ull;
enum : uint8_t { BAM_OK = 0, BAM_SKIP = 1, BAM_CANT_PARSE_NO_COUNT = 2, BAM_CANT_PARSE = 3, BAM_LOW_QUALITY = 4 };   // host/bam_ingest.cpp: parse_one

struct BamParseCfg {
	uint16_t tag[6];           // cell barcode, UMI, barcode quality, UMI quality, gene, read type: letters lo | hi << 8, 0 = not asked
	int32_t filled_bam;        // 1: barcode and UMI from tags, 0: from the read name "id!CB#UMI"
	int32_t min_phred;         // the quality filter is on when > 33 (Tools::ReadParameters::quality_offset)
	int32_t has_read_type;     // a read-type tag is configured
	int32_t n_refs;
	uint32_t intronic_len, intergenic_len;
	uint8_t intronic[24], intergenic[24];
};

// (global memory takes a word at any address on this target: one load instead of four byte loads and their shifts; little-endian like BAM)
__device__ inline uint32_t b_le16(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ inline uint32_t b_le32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// Does a record that makes sense start at offset o?  (block_size, refID, pos, l_read_name, n_cigar_op, l_seq, next_refID, next_pos, the
// NUL that ends the name.)  bs = its block_size.
__device__ inline bool bam_plausible(const uint8_t *d, uint64_t len, uint64_t o, int32_t n_refs, uint32_t &bs) {
	if (o + 36 > len) return false;
	const uint8_t *p = d + o;
	bs = b_le32(p);
	if (bs < 32u || bs > (1u << 26)) return false;
	const int32_t ref = int32_t(b_le32(p + 4)), pos = int32_t(b_le32(p + 8));
	if (ref < -1 || ref >= n_refs || pos < -1) return false;
	const uint32_t lrn = p[12], ncig = b_le16(p + 16), lseq = b_le32(p + 20);
	if (lrn < 1u || lseq > (1u << 26)) return false;
	const int32_t nref = int32_t(b_le32(p + 24)), npos = int32_t(b_le32(p + 28));
	if (nref < -1 || nref >= n_refs || npos < -1) return false;
	const uint64_t aux = 32ull + lrn + 4ull * ncig + (uint64_t(lseq) + 1) / 2 + lseq;
	if (aux > bs) return false;
	const uint64_t nul = o + 36 + lrn - 1;
	if (nul < len && d[nul] != 0) return false;
	if (lrn > 1u && o + 36 < len && (d[o + 36] < 33 || d[o + 36] > 126)) return false;
	return true;
}

// seg_start[k] (k >= 1) = the first offset in [k SEG, (k + 1) SEG) from which four plausible records follow one another (or the data
// ends first), BAM_NONE if there is none.  One wave per segment; seg_start[0] is the caller's.
__global__ __launch_bounds__(256) void bam_seg_guess_kernel(const uint8_t *__restrict__ d, uint64_t len, int32_t n_refs, uint32_t n_segs,
                                                            uint64_t *__restrict__ seg_start) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6) + 1;
	if (k >= n_segs) return;
	const uint64_t lo = uint64_t(k) * BAM_SEG, hi = lo + BAM_SEG < len ? lo + BAM_SEG : len;
	uint64_t found = BAM_NONE;
	for (uint64_t base = lo; base < hi && found == BAM_NONE; base += 64) {
		uint64_t o = base + lane;
		bool ok = o < hi;
		if (ok) {
			uint64_t at = o;
			for (int depth = 0; depth < 4 && ok; ++depth) {
				uint32_t bs;
				if (at + 36 > len) break;               // the data ends: nothing more to check
				ok = bam_plausible(d, len, at, n_refs, bs);
				at += 4ull + bs;
			}
		}
		const uint64_t m = __ballot(ok);
		if (m) found = base + uint64_t(__ffsll((long long)m) - 1);
	}
	if (lane == 0) seg_start[k] = found;
}

// A lane per listed segment walks the records that START in it, from seg_start: count[k] of them (their offsets to rec_off + base[k]
// when rec_off is given), seg_exit[k] = where the chain stands afterwards (the first start at or beyond the segment's end, or the start
// of the record the data cuts off).  bad[0] is raised by a record shorter than its fixed part or longer than 2^26 bytes.
__global__ __launch_bounds__(256) void bam_seg_walk_kernel(const uint8_t *__restrict__ d, uint64_t len, const uint32_t *__restrict__ list, uint32_t n_list,
                                                           const uint64_t *__restrict__ seg_start, uint32_t *__restrict__ count, uint64_t *__restrict__ seg_exit,
                                                           const uint32_t *__restrict__ base, uint64_t *__restrict__ rec_off, uint32_t *__restrict__ bad) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_list) return;
	const uint32_t k = list ? list[i] : i;
	uint64_t o = seg_start[k];
	const uint64_t end = uint64_t(k + 1) * BAM_SEG;
	uint32_t c = 0;
	if (o == BAM_NONE) { count[k] = 0; seg_exit[k] = BAM_NONE; return; }
	while (o < end && o + 4 <= len) {
		const uint32_t bs = b_le32(d + o);
		if (bs < 32u || bs > (1u << 26)) { atomicOr(bad, 1u); break; }   // (beyond any record, as bam_plausible has it: garbage must not become a tail carried from window to window)
		if (o + 4ull + bs > len) break;                // cut off by the end of the window: the tail of the next one
		if (rec_off) rec_off[base[k] + c] = o;
		++c;
		o += 4ull + bs;
	}
	count[k] = c;
	seg_exit[k] = o;
}

// What a record becomes (per record of the window; the accepted ones are made dense afterwards): the four columns of dropest_push_reads,
// the reader's status, and `need` = the host must see this record's strings (a gene name or chromosome the dictionaries do not hold yet, a
// barcode / UMI that does not pack into a 2-bit code).
struct BamRecordOut {
	unsigned long long *cb, *umi;
	uint32_t *gene, *aux;
	uint16_t *umiq_len;
	unsigned long long *qoff;     // where the record's UMI quality string stands in the window (~0: it has none)
	uint8_t *status, *need;
	// -g (genes from a GTF / BED annotation, ReadParamsParser::get_gene_from_reference :92-151): the record's chromosome in the annotation's
	// numbering and the two ends of its alignment go to annotate_reads (annotation_api.hip); bam_resolve_annotated finishes the columns
	int32_t *a_chr; uint32_t *a_pos, *a_end;
};

// The host's dictionaries as the kernels read them: gene-name hash (FNV-1a, host/facade.cpp hash_name) -> gene index in an open-addressing
// table (vals = index + 1, 0 = empty slot), reference id -> chromosome index (-1 = no read touched it yet).
struct BamDict {
	const unsigned long long *gkeys;
	const uint32_t *gvals;
	uint32_t gmask;
	const int32_t *chr_of_ref;
	const int32_t *ann_chr_of_ref;    // -g: reference id -> chromosome of the annotation (-1: it has none of that name); null without -g
	const int32_t *id_of_ann_gene;    // -g: gene of the annotation -> index in the host's gene dictionary, -1 = not in it yet
	// the names of the dictionary's genes by index (dropest_bam_decoder_set_gene_names; null: none given): a hash that is found is confirmed
	// byte by byte, so two names with one FNV-1a value cannot share an index -- the second one goes to the host like any unseen name
	const uint32_t *name_off;         // [n_names + 1]
	const uint8_t *name_pool;
	uint32_t n_names;
	unsigned long long hash_mask;     // all ones (tests: a few bits, so that names do collide)
};
__host__ __device__ inline uint32_t bam_dict_slot(unsigned long long h, uint32_t mask) { return uint32_t((h ^ (h >> 29)) * 0x9E3779B97F4A7C15ull >> 40) & mask; }

__device__ inline unsigned long long bam_pack_bases(const uint8_t *s, uint32_t n) {   // host/facade.cpp pack_bases; 0 = not packable
	if (n == 0u || n > 31u) return 0ull;
	unsigned long long c = 1;
	for (uint32_t i = 0; i < n; ++i) {
		const uint8_t ch = s[i];
		uint32_t b;
		if (ch == 'A') b = 0; else if (ch == 'C') b = 1; else if (ch == 'G') b = 2; else if (ch == 'T') b = 3; else return 0ull;
		c = (c << 2) | b;
	}
	return c;
}

__device__ inline bool bam_equal(const uint8_t *a, uint32_t n, const uint8_t *b, uint32_t m) {
	if (n != m) return false;
	for (uint32_t i = 0; i < n; ++i) if (a[i] != b[i]) return false;
	return true;
}

// One lane per record (host/bam_ingest.cpp: parse_one, the branches that need no dictionary).
// The 64 records of a wave follow one another in the stream (~17 KB of a 10x BAM): the wave copies them into LDS with coalesced 16-byte loads
// and every lane walks ITS record there -- the walk is ~150 loads of single bytes per record, and from memory each of them is 64 transactions
// per wave (the records lie 270 bytes apart), which is what bounded the kernel.  Records that do not fit the wave's 20 KB are walked in memory.
constexpr uint32_t BAM_PARSE_T = 128, BAM_PARSE_STAGE = 20480;
__global__ __launch_bounds__(BAM_PARSE_T) void bam_parse_kernel(const uint8_t *__restrict__ d, const uint64_t *__restrict__ rec_off, uint32_t n_rec, BamParseCfg cfg,
                                                                BamDict dict, BamRecordOut out, uint32_t *__restrict__ bad) {
	__shared__ __attribute__((aligned(16))) uint8_t stage[BAM_PARSE_T / 64][BAM_PARSE_STAGE];
	const uint32_t i = blockIdx.x * BAM_PARSE_T + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const bool valid = i < n_rec;
	const uint64_t my_off = valid ? rec_off[i] : 0ull;
	const uint64_t my_end = valid ? my_off + 4ull + b_le32(d + my_off) : 0ull;
	const uint64_t m = __ballot(valid);
	if (!m) return;
	const int last = 63 - __clzll((long long)m);
	const uint64_t r_begin = (uint64_t(uint32_t(__shfl(int(uint32_t(my_off)), 0))) | (uint64_t(uint32_t(__shfl(int(uint32_t(my_off >> 32)), 0))) << 32));
	const uint64_t r_end = (uint64_t(uint32_t(__shfl(int(uint32_t(my_end)), last))) | (uint64_t(uint32_t(__shfl(int(uint32_t(my_end >> 32)), last))) << 32));
	const uint8_t *at = d + my_off;
	if (r_end > r_begin && r_end - r_begin <= BAM_PARSE_STAGE) {
		const uint32_t bytes = uint32_t(r_end - r_begin);
		for (uint32_t o = lane * 16u; o < bytes; o += 64u * 16u) {          // (the last slice may read up to 15 bytes past the records: the window's buffer has that room)
			uint4 v;
			__builtin_memcpy(&v, d + r_begin + o, 16);
			*reinterpret_cast<uint4 *>(&stage[wave][o]) = v;
		}
		at = &stage[wave][uint32_t(my_off - r_begin)];
	}
	if (!valid) return;
	const uint32_t block_size = b_le32(at);
	const uint8_t *p = at + 4;
	const int32_t ref_id = int32_t(b_le32(p));
	const uint32_t l_read_name = p[8], n_cigar = b_le16(p + 12), flag = b_le16(p + 14), l_seq = b_le32(p + 16);
	out.umiq_len[i] = 0; out.need[i] = 0; out.qoff[i] = ~0ull;
	const uint64_t aux = 32ull + l_read_name + 4ull * n_cigar + (uint64_t(l_seq) + 1) / 2 + l_seq;
	if (aux > block_size) { out.status[i] = BAM_CANT_PARSE; atomicOr(bad, 1u); return; }   // the host reader throws "Corrupt BAM record" (host/bam_ingest.cpp parse_one): so does the window, from this flag
	if ((flag & 0x4u) || (flag & 0x100u)) { out.status[i] = BAM_SKIP; return; }                    // BamController.cpp:87-88
	if (ref_id < 0 || ref_id >= cfg.n_refs) { out.status[i] = BAM_CANT_PARSE_NO_COUNT; return; }  // :90-104
	// the tags asked for, in one walk (BamRecord::get_string_tags)
	const uint8_t *tags = p + aux;
	const uint32_t tags_size = block_size - uint32_t(aux);
	const uint8_t *val[6]; uint32_t vlen[6]; bool found[6], closed[6];
#pragma unroll
	for (int k = 0; k < 6; ++k) { found[k] = false; closed[k] = false; val[k] = nullptr; vlen[k] = 0; }
	uint32_t o = 0;
	while (o + 3u <= tags_size) {
		const uint32_t name = b_le16(tags + o);
		const uint8_t type = tags[o + 2];
		o += 3;
		uint32_t len = 0;
		bool text = false, stop = false;
		switch (type) {
			case 'A': case 'c': case 'C': len = 1; break;
			case 's': case 'S': len = 2; break;
			case 'i': case 'I': case 'f': len = 4; break;
			case 'Z': case 'H': {
				uint32_t e = o;
				while (e < tags_size && tags[e]) ++e;
				if (e >= tags_size) { stop = true; break; }
				len = e - o + 1; text = true; break;
			}
			case 'B': {
				if (o + 5u > tags_size) { stop = true; break; }
				const uint8_t sub = tags[o];
				const uint32_t cnt = b_le32(tags + o + 1);
				const uint32_t w = (sub == 'c' || sub == 'C') ? 1u : ((sub == 's' || sub == 'S') ? 2u : 4u);
				const uint64_t l64 = 5ull + uint64_t(cnt) * w;
				if (l64 > tags_size) { stop = true; break; }
				len = uint32_t(l64); break;
			}
			default: stop = true;
		}
		if (stop || o + len > tags_size) break;
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			if (cfg.tag[k] != name || !cfg.tag[k] || found[k] || closed[k]) continue;
			if (text) { val[k] = tags + o; vlen[k] = len - 1; found[k] = true; }
			else if (type == 'A') { val[k] = tags + o; vlen[k] = 1; found[k] = true; }
			else closed[k] = true;
		}
		o += len;
	}
	enum { T_CB, T_UMI, T_CBQ, T_UMIQ, T_GENE, T_TYPE };
	const uint8_t *cb = nullptr, *umi = nullptr;
	uint32_t cb_n = 0, umi_n = 0;
	bool pass = true;
	if (cfg.filled_bam) {                                                     // FilledBamParamsParser.cpp:12-40
		if (!found[T_CB] || !found[T_UMI]) { out.status[i] = BAM_CANT_PARSE; return; }
		cb = val[T_CB]; cb_n = vlen[T_CB]; umi = val[T_UMI]; umi_n = vlen[T_UMI];
		if (!cb_n || !umi_n) { out.status[i] = BAM_CANT_PARSE; return; }
		if (cfg.min_phred > 33) {                                             // ReadParameters::check_quality
			if (found[T_CBQ]) for (uint32_t j = 0; j < vlen[T_CBQ]; ++j) pass &= int32_t(int8_t(val[T_CBQ][j])) >= int32_t(int8_t(cfg.min_phred));
			if (found[T_UMIQ]) for (uint32_t j = 0; j < vlen[T_UMIQ]; ++j) pass &= int32_t(int8_t(val[T_UMIQ][j])) >= int32_t(int8_t(cfg.min_phred));
		}
		out.umiq_len[i] = uint16_t(found[T_UMIQ] ? (vlen[T_UMIQ] > 0xFFFFu ? 0xFFFFu : vlen[T_UMIQ]) : 0u);
		if (found[T_UMIQ]) out.qoff[i] = my_off + uint64_t(val[T_UMIQ] - at);
	} else {                                                                  // ReadParamsParser.cpp:20-33: "id!CB#UMI"
		const uint8_t *name = p + 32;
		const uint32_t nl = l_read_name ? l_read_name - 1 : 0;
		int32_t up = -1, cp = -1;
		for (int32_t j = int32_t(nl) - 1; j >= 0; --j) if (name[j] == '#') { up = j; break; }
		if (up >= 0) for (int32_t j = up; j >= 0; --j) if (name[j] == '!') { cp = j; break; }     // rfind('!', up): at or before `up`
		if (up < 0 || cp < 0) { out.status[i] = BAM_CANT_PARSE; return; }
		cb = name + cp + 1; cb_n = uint32_t(up - cp - 1); umi = name + up + 1; umi_n = nl - uint32_t(up) - 1;
		if (!cb_n || !umi_n) { out.status[i] = BAM_CANT_PARSE; return; }
	}
	if (!pass) { out.status[i] = BAM_LOW_QUALITY; return; }
	if (dict.ann_chr_of_ref) {                                                // -g: the gene comes from the alignment's place (bam_resolve_annotated)
		int64_t ref_len = 0;                                                  // BamAlignment::GetEndPosition: reference bases of M, D, N, =, X
		const uint8_t *cig = p + 32 + l_read_name;
		for (uint32_t k = 0; k < n_cigar; ++k) {
			const uint32_t op = b_le32(cig + 4ull * k), kind = op & 0xFu;
			if (kind == 0 || kind == 2 || kind == 3 || kind == 7 || kind == 8) ref_len += op >> 4;
		}
		const int32_t position = int32_t(b_le32(p + 4));
		out.a_chr[i] = dict.ann_chr_of_ref[ref_id];
		out.a_pos[i] = uint32_t(position); out.a_end[i] = uint32_t(int32_t(int64_t(position) + ref_len));
		out.cb[i] = bam_pack_bases(cb, cb_n); out.umi[i] = bam_pack_bases(umi, umi_n);
		out.aux[i] = uint32_t(ref_id);
		out.status[i] = BAM_OK;
		return;
	}
	uint32_t mark;                                                            // get_gene + parse_read_type (ReadParamsParser.cpp:36-90)
	unsigned long long gh = 0;
	if (!found[T_GENE]) mark = 1;                                             // HAS_NOT_ANNOTATED
	else {
		if (vlen[T_GENE]) {
			gh = 1469598103934665603ull;                                      // FNV-1a, as CellsDataContainer::hash_name
			for (uint32_t j = 0; j < vlen[T_GENE]; ++j) { gh ^= val[T_GENE][j]; gh *= 1099511628211ull; }
			gh &= dict.hash_mask;
		}
		if (!cfg.has_read_type || !found[T_TYPE]) mark = 2;                   // HAS_EXONS
		else if (bam_equal(val[T_TYPE], vlen[T_TYPE], cfg.intronic, cfg.intronic_len)) mark = 4;
		else if (cfg.intergenic_len && bam_equal(val[T_TYPE], vlen[T_TYPE], cfg.intergenic, cfg.intergenic_len)) mark = 1;
		else mark = 2;
	}
	// the columns of dropest_push_reads, as fast_window of host/bam_ingest.cpp fills them
	const bool has_gene = found[T_GENE] && vlen[T_GENE];
	bool need = false;
	const unsigned long long cbc = bam_pack_bases(cb, cb_n);
	need |= !cbc;
	unsigned long long uc = 1;
	uint32_t gid = 0xFFFFFFFFu;                                               // DROPEST_NO_GENE
	if (has_gene) {
		uc = bam_pack_bases(umi, umi_n);
		need |= !uc;
		uint32_t sl = bam_dict_slot(gh, dict.gmask);
		gid = 0;
		for (;;) {
			const uint32_t v = dict.gvals[sl];
			if (!v) { need = true; break; }
			if (dict.gkeys[sl] == gh) { gid = v - 1; break; }
			sl = (sl + 1) & dict.gmask;
		}
		if (!need && dict.name_off) {      // the name behind the hash, byte by byte ("exact only with probability" otherwise)
			bool same = gid < dict.n_names;
			if (same) {
				const uint32_t o = dict.name_off[gid], nlen = dict.name_off[gid + 1] - o;
				same = nlen == vlen[T_GENE];
				for (uint32_t j = 0; same && j < nlen; ++j) same = dict.name_pool[o + j] == val[T_GENE][j];
			}
			if (!same) { need = true; gid = 0; }
		}
	}
	uint32_t aux_w = mark << 16;
	if (!has_gene || (mark & 6u)) {                                           // the read reaches Stats::inc(chromosome)
		const int32_t chr = dict.chr_of_ref[ref_id];
		if (chr < 0) need = true; else aux_w |= uint32_t(chr);
	}
	out.cb[i] = cbc; out.umi[i] = uc; out.gene[i] = gid; out.aux[i] = aux_w;
	out.need[i] = need ? (has_gene ? 3 : 1) : (has_gene ? 2 : 0);             // bit 0: the host must see the record; bit 1: it carries a gene
	out.status[i] = BAM_OK;
}

// -g: what annotate_reads (annotation_api.hip) said about the accepted records -> the columns (host/bam_ingest.cpp parse_one, the -g branch)
__global__ __launch_bounds__(256) void bam_resolve_annotated_kernel(uint32_t n_rec, BamDict dict, BamRecordOut out, const uint32_t *__restrict__ ann_gene,
                                                                    const int32_t *__restrict__ ann_mark) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_rec || out.status[i] != BAM_OK) return;
	const int32_t m = ann_mark[i];
	if (m == -1) { out.status[i] = BAM_CANT_PARSE; return; }                  // RefGenesContainer::ChrNotFoundException (BamController.cpp:153-161)
	const int32_t ref_id = int32_t(out.aux[i]);
	bool need = m == -2;                                                      // more results at an end point than the kernel holds: the host decides
	const uint32_t g = ann_gene[i];
	const bool has_gene = !need && g != 0xFFFFFFFFu;
	const uint32_t mark = need ? 0u : uint32_t(m) & 7u;
	need |= !out.cb[i];
	uint32_t gid = 0xFFFFFFFFu;
	if (has_gene) {
		need |= !out.umi[i];
		const int32_t id = dict.id_of_ann_gene[g];
		if (id < 0) need = true; else gid = uint32_t(id);
	} else out.umi[i] = 1;
	uint32_t aux_w = mark << 16;
	if (!has_gene || (mark & 6u)) {
		const int32_t chr = dict.chr_of_ref[ref_id];
		if (chr < 0) need = true; else aux_w |= uint32_t(chr);
	}
	out.gene[i] = gid; out.aux[i] = aux_w;
	out.need[i] = need ? (has_gene ? 3 : 1) : (has_gene ? 2 : 0);
}

// ---- the accepted records, dense ------------------------------------------------------------------------------------------------
constexpr uint32_t BAM_FIN_PER = 16, BAM_FIN_TILE = 256 * BAM_FIN_PER;
// ql_max / ql_min_inv: over the accepted gene-bearing reads, the longest UMI quality string and 0xFFFFFFFF - the shortest (both grow from 0)
struct BamWindowCounts { uint32_t status[5]; uint32_t quality, any_gene, ql_max, ql_min_inv, pad; };

// accepted records and accepted records the host must see, per tile of 4 096 records; the window's counters
__global__ __launch_bounds__(256) void bam_fin_count_kernel(const uint8_t *__restrict__ status, const uint8_t *__restrict__ need, const uint16_t *__restrict__ uql,
                                                            uint32_t n, uint32_t *__restrict__ tile_ok, uint32_t *__restrict__ tile_need, BamWindowCounts *__restrict__ wc) {
	__shared__ uint32_t acc[10];
	if (threadIdx.x < 10) acc[threadIdx.x] = 0;
	__syncthreads();
	uint32_t c[5] = {0, 0, 0, 0, 0}, nn = 0, q = 0, g = 0, qmax = 0, qmin_inv = 0;
	const uint32_t first = blockIdx.x * BAM_FIN_TILE + threadIdx.x * BAM_FIN_PER;
	for (uint32_t j = 0; j < BAM_FIN_PER; ++j) {
		const uint32_t i = first + j;
		if (i >= n) break;
		const uint32_t st = status[i];
		c[st < 5u ? st : 3u]++;
		if (st == BAM_OK) {
			nn += need[i] & 1u; g |= (need[i] >> 1) & 1u; q |= uql[i] ? 1u : 0u;
			if (need[i] & 2u) { const uint32_t l = uql[i]; qmax = l > qmax ? l : qmax; qmin_inv = (0xFFFFFFFFu - l) > qmin_inv ? (0xFFFFFFFFu - l) : qmin_inv; }
		}
	}
#pragma unroll
	for (int k = 0; k < 5; ++k) if (c[k]) atomicAdd(&acc[k], c[k]);
	if (nn) atomicAdd(&acc[5], nn);
	if (q) atomicOr(&acc[6], 1u);
	if (g) atomicOr(&acc[7], 1u);
	if (g) { atomicMax(&acc[8], qmax); atomicMax(&acc[9], qmin_inv); }
	__syncthreads();
	if (threadIdx.x == 0) {
		tile_ok[blockIdx.x] = acc[0]; tile_need[blockIdx.x] = acc[5];
		for (int k = 0; k < 5; ++k) if (acc[k]) atomicAdd(&wc->status[k], acc[k]);
		if (acc[6]) atomicOr(&wc->quality, 1u);
		if (acc[7]) { atomicOr(&wc->any_gene, 1u); atomicMax(&wc->ql_max, acc[8]); atomicMax(&wc->ql_min_inv, acc[9]); }
	}
}

// exclusive scan of two arrays of tile counts by ONE workgroup (a window has a few thousand tiles); totals[0 / 1] = the sums
__global__ __launch_bounds__(1024) void bam_fin_scan_kernel(uint32_t *__restrict__ a, uint32_t *__restrict__ b, uint32_t n, uint32_t *__restrict__ totals) {
	__shared__ uint32_t sa[1024], sb[1024];
	__shared__ uint32_t carry[2];
	if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; }
	__syncthreads();
	for (uint32_t base = 0; base < n; base += 1024) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t va = i < n ? a[i] : 0u, vb = i < n ? b[i] : 0u;
		sa[threadIdx.x] = va; sb[threadIdx.x] = vb;
		__syncthreads();
		for (uint32_t dd = 1; dd < 1024; dd <<= 1) {
			const uint32_t xa = threadIdx.x >= dd ? sa[threadIdx.x - dd] : 0u, xb = threadIdx.x >= dd ? sb[threadIdx.x - dd] : 0u;
			__syncthreads();
			sa[threadIdx.x] += xa; sb[threadIdx.x] += xb;
			__syncthreads();
		}
		if (i < n) { a[i] = carry[0] + sa[threadIdx.x] - va; b[i] = carry[1] + sb[threadIdx.x] - vb; }
		__syncthreads();
		if (threadIdx.x == 1023) { carry[0] += sa[1023]; carry[1] += sb[1023]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) { totals[0] = carry[0]; totals[1] = carry[1]; }
}

struct BamDense { unsigned long long *cb, *umi; uint32_t *gene, *aux; uint32_t *need_rec, *need_pos, *need_size; unsigned long long *qoff; };

// the accepted records to their dense places (file order), and the list of those the host must see: (record, dense place, bytes)
__global__ __launch_bounds__(256) void bam_fin_scatter_kernel(const uint8_t *__restrict__ d, const uint64_t *__restrict__ rec_off, BamRecordOut in, uint32_t n,
                                                              const uint32_t *__restrict__ tile_ok, const uint32_t *__restrict__ tile_need, BamDense out) {
	__shared__ uint32_t s_ok[256], s_need[256];
	const uint32_t first = blockIdx.x * BAM_FIN_TILE + threadIdx.x * BAM_FIN_PER;
	uint32_t ok = 0, nd = 0;
	for (uint32_t j = 0; j < BAM_FIN_PER; ++j) {
		const uint32_t i = first + j;
		if (i >= n) break;
		if (in.status[i] == BAM_OK) { ++ok; nd += in.need[i] & 1u; }
	}
	s_ok[threadIdx.x] = ok; s_need[threadIdx.x] = nd;
	__syncthreads();
	for (uint32_t dd = 1; dd < 256; dd <<= 1) {
		const uint32_t xa = threadIdx.x >= dd ? s_ok[threadIdx.x - dd] : 0u, xb = threadIdx.x >= dd ? s_need[threadIdx.x - dd] : 0u;
		__syncthreads();
		s_ok[threadIdx.x] += xa; s_need[threadIdx.x] += xb;
		__syncthreads();
	}
	uint32_t at = tile_ok[blockIdx.x] + s_ok[threadIdx.x] - ok, nat = tile_need[blockIdx.x] + s_need[threadIdx.x] - nd;
	for (uint32_t j = 0; j < BAM_FIN_PER; ++j) {
		const uint32_t i = first + j;
		if (i >= n) break;
		if (in.status[i] != BAM_OK) continue;
		out.cb[at] = in.cb[i]; out.umi[at] = in.umi[i]; out.gene[at] = in.gene[i]; out.aux[at] = in.aux[i];
		out.qoff[at] = (in.need[i] & 2u) ? in.qoff[i] : ~0ull;       // (the quality string of a read without a gene is not looked at)
		if (in.need[i] & 1u) { out.need_rec[nat] = i; out.need_pos[nat] = at; out.need_size[nat] = 4u + b_le32(d + rec_off[i]); ++nat; }
		++at;
	}
}

// one row of ql bytes per accepted read: its UMI quality string (zeros for a read without a gene or without a string); the caller has seen that
// every gene-bearing read's string is ql long
__global__ __launch_bounds__(256) void bam_quality_rows_kernel(const uint8_t *__restrict__ d, const unsigned long long *__restrict__ qoff, uint32_t n, uint32_t ql,
                                                               uint8_t *__restrict__ rows) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k >= n) return;
	const unsigned long long o = qoff[k];
	uint8_t *row = rows + size_t(k) * ql;
	if (o == ~0ull) { for (uint32_t j = 0; j < ql; ++j) row[j] = 0; }
	else { const uint8_t *q = d + o; for (uint32_t j = 0; j < ql; ++j) row[j] = q[j]; }
}

// what the host resolved (new dictionary entries, strings with N) back into the dense columns
__global__ __launch_bounds__(256) void bam_patch_kernel(const uint32_t *__restrict__ pos, const unsigned long long *__restrict__ cb, const unsigned long long *__restrict__ umi,
                                                        const uint32_t *__restrict__ gene, const uint32_t *__restrict__ aux, uint32_t n, BamDense out) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k >= n) return;
	const uint32_t at = pos[k];
	out.cb[at] = cb[k]; out.umi[at] = umi[k]; out.gene[at] = gene[k]; out.aux[at] = aux[k];
}

// bytes of the listed records, one after the other (the host asks for the records whose strings it must see: a new gene name, an N)
__global__ __launch_bounds__(256) void bam_gather_records_kernel(const uint8_t *__restrict__ d, const uint64_t *__restrict__ rec_off, const uint32_t *__restrict__ idx,
                                                                 const uint64_t *__restrict__ dst_off, uint32_t n, uint8_t *__restrict__ dst) {
	const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (r >= n) return;
	const uint8_t *src = d + rec_off[idx[r]];
	const uint32_t bytes = 4u + b_le32(src);
	uint8_t *to = dst + dst_off[r];
	for (uint32_t j = lane; j < bytes; j += 64) to[j] = src[j];
}

}  // namespace dropest
