// bam_ingest.h -- BAM records -> CellsDataContainer::add_record, without BamTools.
//
// Mirrors the part of Estimation::BamProcessing that feeds the container:
//   BamController::parse_bam_file / process_alignment  (Estimation/BamProcessing/BamController.cpp:70-172)
//   FilledBamParamsParser::get_read_params (-f: CB / UB tags, FilledBamParamsParser.cpp:12-40)
//   ReadParamsParser::get_read_params (read name "id!CB#UMI", ReadParamsParser.cpp:20-33)
//   ReadParamsParser::get_gene / parse_read_type (gene tag + optional read-type tag, :36-90)
//   BamTags defaults (BamTags.cpp:7-24), Tools::ReadParameters quality check (Tools/ReadParameters.cpp:118-136)
//   ReadParamsParser::get_gene_from_reference (-g: gene annotation from a GTF / BED file, gene_annotation.h)
//   ReadMapParamsParser::get_read_params (-r: droptag's read-parameter files "name cb umi cb_quality umi_quality", gzip;
//     ReadMapParamsParser.cpp:22-48, :50-108; every name serves once)
// Not built: filtered BAM output (-F, -b).
//
// The container format: BGZF (gzip members with a 'BC' extra field, SAMv1 §4.1) holding the BAM stream (§4.2).
// Blocks are inflated by a pool of host threads, records are parsed in stream order by the caller's thread (the
// order of add_record calls defines cell, gene and UMI ids).  Host-only code; zlib.
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <string_view>
#include <vector>

#include "facade.h"
#include "gene_annotation.h"

namespace Estimation {
namespace BamProcessing {

struct BamTags {   // BamTags.cpp:7-24 (defaults of the XML config)
	std::string cell_barcode = "CB", cell_barcode_raw = "CR", umi = "UB", umi_raw = "UR", gene = "GX";
	std::string cell_barcode_quality = "CQ", umi_quality = "UQ";
	std::string read_type, intronic_read_value, intergenic_read_value, exonic_read_value;
};

// One decoded alignment (the fields the path looks at)
struct BamRecord {
	int32_t ref_id = -1;
	int32_t position = -1;           // 0-based leftmost coordinate
	int32_t end_position = -1;       // BamAlignment::GetEndPosition(): position + reference bases consumed by the CIGAR (M, D, N, =, X)
	uint16_t flag = 0;
	std::string name;
	std::string_view name_view;      // set by parse_record (points into the window)
	const uint8_t *tags = nullptr;   // aux data, valid until the next record is read
	size_t tags_size = 0;
	bool is_mapped() const { return !(flag & 0x4); }
	bool is_primary() const { return !(flag & 0x100); }
	// Z / A / H tags as text (BamAlignment::GetTag(tag, std::string&)); false if absent or numeric
	bool get_string_tag(const std::string &tag, std::string &value, char *type = nullptr) const;
	bool get_string_tag(const std::string &tag, std::string_view &value, char *type = nullptr) const;   // view into `tags`
	// several tags in ONE walk over the aux data (the walk is most of a record's parse time when six tags are asked for one by
	// one): wanted[k] = the two tag letters packed as lo | hi << 8, 0 = not asked; same answers as get_string_tag per tag
	void get_string_tags(const uint16_t *wanted, int n_wanted, std::string_view *values, bool *found) const;
};

class BamReader {
	struct Impl;
	Impl *impl;
public:
	explicit BamReader(const std::string &path, unsigned threads = 0);   // throws std::runtime_error("Can't open BAM file: ...")
	~BamReader();
	BamReader(const BamReader &) = delete;
	BamReader &operator=(const BamReader &) = delete;
	const std::vector<std::string> &reference_names() const;
	const std::string &header_text() const;
	double file_over_first_batch() const;                                // file bytes / bytes the first decompressed batch covered (>= 1)
	bool next(BamRecord &rec);                                           // false at end of file
	// A run of whole records (as many as the decompressed window holds), valid until the next call of next / next_window;
	// offsets[i] = start of record i (its block_size field) relative to data
	bool next_window(const uint8_t *&data, std::vector<uint32_t> &offsets);
	static void parse_record(const uint8_t *at, BamRecord &rec);         // name is NOT copied: see name_view
};

class BamController {
public:
	struct Counters {
		size_t total_reads = 0, cant_parse = 0, low_quality = 0, saved = 0;
		double wait_ms = 0, parse_ms = 0, add_ms = 0;   // waiting for decompressed data / parallel record parsing / add_record (serial)
	};
private:
	BamTags _tags;
	struct ReadParams { std::string cb, umi, umi_quality; bool pass_quality; };
	std::unordered_map<std::string, ReadParams> _read_params;     // -r: read name -> parameters (erased when served)
	bool _params_from_files = false;
	void load_read_params(const std::string &filenames);
	bool _filled_bam, _gene_in_chromosome_name;
	int _min_barcode_phred;
	unsigned _threads;
	bool _device_decode = false;
	Counters _counters;
	Tools::GeneAnnotation::RefGenesContainer _genes;   // -g: empty unless a GTF / BED file was given
public:
	// gtf_path: GTF / BED annotation (-g), "" = genes come from the BAM's gene tag; read_param_filenames must be empty (not built)
	BamController(const BamTags &tags, bool filled_bam, const std::string &read_param_filenames, const std::string &gtf_path,
	              bool gene_in_chromosome_name, int min_barcode_phred, unsigned threads = 0);
	// BamController::parse_bam_files with a BamProcessor: every accepted read reaches container.add_record in file order
	void parse_bam_files(const std::vector<std::string> &bam_files, CellsDataContainer &container);
	// BGZF inflate, record chain and tag walk on the container's GPU (include/dropest_bgzf.h) where the configuration allows it: tags or read
	// names as the source of barcode / UMI, the gene tag or a -g annotation (annotation_api.hip) as the source of the gene; not with -r, gene =
	// chromosome name, or a sharded container (the host reader then runs).
	// CRC-32 and ISIZE of every block are checked there too.  DROPEST_BAM_DEVICE=1 in the environment does the same.
	void set_device_decode(bool on) { _device_decode = on; }
	// the device path keeps one decoder per GPU (streams, 2-7 GB of device buffers, up to 512 MB of pinned staging) between the files of one
	// parse_bam_files call; they are given back on a helper thread when the call ends (DROPEST_BAM_KEEP_DECODERS=1: kept for the next call).
	// This waits for that thread and frees whatever is still cached.
	static void release_device_decoders();
	const Counters &counters() const { return _counters; }
};

}  // namespace BamProcessing
}  // namespace Estimation
