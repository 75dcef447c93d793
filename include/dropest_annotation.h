/* dropest_annotation.h -- gene annotation of aligned reads on the device: the read-level decision of
 * ReadParamsParser::get_gene_from_reference (Estimation/BamProcessing/ReadParamsParser.cpp:92-176) over the flat form of
 * Tools::GeneAnnotation::RefGenesContainer (Tools/GeneAnnotation/RefGenesContainer.cpp:179-211), one thread per read,
 * binary searches over sorted pieces.  The flat tables come from the host-side loader
 * (dropest_gene_annotation_flat, libdropest_facade); plain C, no torch types. */
#ifndef DROPEST_ANNOTATION_H
#define DROPEST_ANNOTATION_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* All positions 0-based, half-open.  Chromosome c owns the pieces chr_seg_begin[c] .. chr_seg_begin[c+1]; piece s is
 * covered by the transcripts seg_tr[seg_tr_begin[s] .. seg_tr_begin[s+1]]; transcript t belongs to gene tr_gene[t] and
 * has the sorted, disjoint exon spans [tr_exon_begin[t], tr_exon_begin[t+1]) and intron spans likewise. */
typedef struct {
	uint32_t n_chr, n_seg, n_tr, n_genes;
	int32_t use_introns_from_gtf;
	const uint32_t *chr_seg_begin;                       /* [n_chr + 1] */
	const uint32_t *seg_start, *seg_end, *seg_tr_begin;  /* [n_seg], [n_seg], [n_seg + 1] */
	const uint32_t *seg_tr;                              /* [seg_tr_begin[n_seg]] */
	const uint32_t *tr_gene, *tr_exon_begin, *tr_intron_begin;   /* [n_tr], [n_tr + 1], [n_tr + 1] */
	const uint32_t *exon_start, *exon_end, *intron_start, *intron_end;
} dropest_flat_annotation;

typedef struct dropest_annotation dropest_annotation;

/* uploads the tables (host pointers) to `device`; 0 = ok, else dropest_annotation_last_error() */
int dropest_annotation_create(int device, const dropest_flat_annotation *flat, dropest_annotation **out);
void dropest_annotation_destroy(dropest_annotation *a);
const char *dropest_annotation_last_error(void);

/* n reads given as host arrays: chromosome index into the annotation's chromosome list (-1 = unknown name),
 * alignment start and end (BamAlignment::Position / GetEndPosition).  Per read: gene index or 0xFFFFFFFF, and the
 * UMI::Mark bits (1 not annotated, 2 exon, 4 intron), -1 for an unknown chromosome
 * (RefGenesContainer::ChrNotFoundException), -2 when more than 16 (gene, type) results met at one end point (the
 * caller resolves those on the host). */
int dropest_annotation_query(dropest_annotation *a, uint64_t n, const int32_t *chr, const uint32_t *position,
                             const uint32_t *end_position, uint32_t *gene, int32_t *mark);

/* The same for DEVICE arrays of the annotation's GPU, asynchronous on `stream` (a hipStream_t; the device BAM path, dropest_bgzf.h). */
int dropest_annotation_query_device(dropest_annotation *a, void *stream, uint64_t n, const int32_t *d_chr, const uint32_t *d_position,
                                    const uint32_t *d_end_position, uint32_t *d_gene, int32_t *d_mark);
uint32_t dropest_annotation_genes(const dropest_annotation *a);
int dropest_annotation_device(const dropest_annotation *a);

#ifdef __cplusplus
}
#endif
#endif
