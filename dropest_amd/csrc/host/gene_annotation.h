// gene_annotation.h -- gene annotation of aligned reads from a GTF / BED file (-g), behind the reference's
// Tools::GeneAnnotation::RefGenesContainer interface (Tools/GeneAnnotation/RefGenesContainer.h:18-98) plus the
// read-level decision of ReadParamsParser::get_gene_from_reference (Estimation/BamProcessing/ReadParamsParser.cpp:92-176).
//
// The reference keeps, per chromosome, an IntervalsContainer of transcripts (per-label linked lists, a multimap of
// open / close events, "homogeneous intervals" of std::set<std::string>) and one more container per transcript for
// its exons.  Loading reproduces what those containers compute (including where that is not the union of a label's
// intervals, see gene_annotation.cpp); the result is flat and immutable, so that the BAM reader's worker threads can
// query it concurrently: per chromosome the sorted pieces with the transcripts of each (CSR), per transcript the sorted
// pieces of its exons (and introns); a query is two binary searches.
// Host-only code; zlib for .gz files.
#pragma once

#include <cstdint>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace Tools {
namespace GeneAnnotation {

struct GtfRecord {
	enum RecordType { NONE = 0, INTRON, EXON };                            // GtfRecord.h:20-26
};

class RefGenesContainer {
public:
	using pos_t = size_t;
	struct QueryResult {
		std::string gene_name;
		GtfRecord::RecordType type;
		QueryResult(const std::string &gene_name_ = "", GtfRecord::RecordType type_ = GtfRecord::NONE) : gene_name(gene_name_), type(type_) {}
		bool operator<(const QueryResult &other) const { return type == other.type ? gene_name < other.gene_name : type < other.type; }
	};
	using query_results_t = std::set<QueryResult>;
	class ChrNotFoundException : public std::runtime_error {
	public:
		const std::string chr_name;
		explicit ChrNotFoundException(const std::string &chr) : std::runtime_error("Can't find chromosome '" + chr + "'"), chr_name(chr) {}
	};

	RefGenesContainer() = default;                                          // is_empty()
	explicit RefGenesContainer(const std::string &genes_filename);          // .gtf / .bed, optionally .gz
	// labels of the transcripts / exons intersecting [start_pos, end_pos), 0-based (RefGenesContainer.cpp:179-211)
	query_results_t get_gene_info(const std::string &chr_name, pos_t start_pos, pos_t end_pos) const;
	bool is_empty() const { return _is_empty; }
	bool has_introns() const { return _gtf_has_transcripts || _use_introns_from_gtf; }

	// ReadParamsParser::get_gene_from_reference: gene name + UMI::Mark bits (1 not annotated, 2 exon, 4 intron) of an
	// alignment covering [position, end_position) on `chr_name`.  Throws ChrNotFoundException.
	int gene_of_alignment(const std::string &chr_name, pos_t position, pos_t end_position, std::string &gene) const;
	size_t chromosomes_number() const { return _chromosomes.size(); }

	// Flat tables for the device kernel (include/dropest_annotation.h); chromosomes in the order of chr_names, genes in the
	// order of gene_names.  Positions must fit 32 bits.
	struct Flat {
		std::vector<std::string> chr_names, gene_names;
		bool use_introns_from_gtf = false;
		std::vector<uint32_t> chr_seg_begin, seg_start, seg_end, seg_tr_begin, seg_tr, tr_gene, tr_exon_begin, tr_intron_begin,
		                      exon_start, exon_end, intron_start, intron_end;
	};
	Flat flatten() const;

private:
	struct Span { pos_t start, end; };
	struct Transcript {
		std::string id;
		uint32_t gene;                      // index into _genes
		pos_t start, end;                   // merged extent of its records
		std::vector<Span> exons, introns;   // sorted, disjoint unions
	};
	struct Chromosome {
		std::vector<Transcript> transcripts;
		std::vector<pos_t> seg_start, seg_end;   // the reference's "homogeneous" pieces, ascending, possibly with gaps
		std::vector<uint32_t> seg_begin;    // CSR into seg_transcripts
		std::vector<uint32_t> seg_transcripts;
	};
	bool _is_empty = true, _use_introns_from_gtf = false, _gtf_has_transcripts = true;
	std::vector<std::string> _genes;
	std::unordered_map<std::string, Chromosome> _chromosomes;

	void collect(const Chromosome &c, pos_t start, pos_t end, query_results_t &out) const;
};

}  // namespace GeneAnnotation
}  // namespace Tools

// Plain-C access (what a non-C++ host binds; also used by the CPU tests)
extern "C" {
void *dropest_gene_annotation_load(const char *path);                      // NULL on error (dropest_gene_annotation_error)
void dropest_gene_annotation_free(void *handle);
const char *dropest_gene_annotation_error(void);
// results in std::set order; names: stride bytes each; returns the count, -1 = unknown chromosome
long dropest_gene_annotation_query(void *handle, const char *chr, uint64_t start, uint64_t end, char *names, int stride, int *types, int cap);
// mark bits >= 0, -1 = unknown chromosome
int dropest_gene_annotation_read(void *handle, const char *chr, uint64_t position, uint64_t end_position, char *gene, int cap);
// flat tables (RefGenesContainer::Flat): sizes = {n_chr, n_seg, n_tr, n_genes, covering transcripts, exon spans, intron
// spans, use_introns_from_gtf}; `arrays` = the 12 tables in the order of dropest_flat_annotation's pointers
void dropest_gene_annotation_flat_sizes(void *handle, uint32_t sizes[8]);
void dropest_gene_annotation_flat_fill(void *handle, uint32_t *const arrays[12]);
const char *dropest_gene_annotation_chr_name(void *handle, uint32_t index);
const char *dropest_gene_annotation_gene_name(void *handle, uint32_t index);
}
