/* dropest_synth.h -- counter-based synthetic "10x-shaped" read stream (bench / parity input only).
 *
 * Not part of the reference's interface: the reference has no generator (SURVEY.md §8d defines this one).
 * Read i of the stream is a pure function of (seed, stream_id, i) -- Philox-4x32-10 keyed by
 * (seed, stream_id), counter = ordinal -- evaluated with integer arithmetic only, by the SAME source
 * function on the host and on the device, so the CPU oracle and the GPU see bit-identical streams and a
 * 10^9-read stream never has to cross PCIe.
 *
 * Model (per read): 92 % of reads belong to a real cell drawn from `cell_cdf`; `permille_neighbour`
 * reads carry a barcode at Hamming distance 1 from a real cell's (same molecule pool); `permille_ambient`
 * reads carry a uniformly random barcode.  Gene ~ `gene_cdf` (Zipf on the host side); molecule = uniform
 * index into a per-(cell, gene) pool sized expected_reads/reads_per_molecule; UMI = hash(cell, gene,
 * molecule) truncated to umi_len bases.  `permille_intergenic` reads have no gene.  Marks: exon /
 * intron / exon+not-annotated by the given per-mille split.  Chromosome = gene mod n_chr.
 */
#ifndef DROPEST_SYNTH_H
#define DROPEST_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	uint64_t seed;
	uint32_t stream_id;
	uint32_t n_cells;              /* real cells */
	const uint64_t *cell_cb;       /* [n_cells] packed barcode codes (host or device pointer, see calls) */
	const uint32_t *cell_cdf;      /* [n_cells] inclusive upper bounds on a u32 draw, last = 0xFFFFFFFF */
	uint32_t n_genes;
	const uint32_t *gene_cdf;      /* [n_genes] same convention */
	uint32_t cb_len, umi_len;      /* bases, <= 31 */
	uint32_t n_chr;
	uint32_t permille_neighbour, permille_ambient, permille_intergenic;
	uint32_t permille_intron, permille_exon_na;   /* remainder = plain exon */
	uint64_t n_effective;          /* gene-bearing real-cell reads of the whole stream (sizes the pools) */
	uint32_t reads_per_molecule;   /* mean duplicates per molecule (pool divisor) */
} dropest_synth_params;

/* Host generation of reads [first, first+n) into host arrays.  Table pointers are host pointers. */
int dropest_synth_generate_host(const dropest_synth_params *p, uint64_t first, uint64_t n, uint64_t *cb,
                                uint64_t *umi, uint32_t *gene, uint32_t *aux);
/* Device generation into device arrays on `device`.  Table pointers in *p are HOST pointers (copied). */
int dropest_synth_generate_device(const dropest_synth_params *p, int device, uint64_t first, uint64_t n,
                                  uint64_t *d_cb, uint64_t *d_umi, uint32_t *d_gene, uint32_t *d_aux);

/* plain device memory helpers for callers without a HIP binding (bench.py): 0 on success */
int dropest_dev_alloc(int device, uint64_t bytes, void **out);
int dropest_dev_free(int device, void *p);
int dropest_dev_copy_to_host(int device, void *dst, const void *d_src, uint64_t bytes);
int dropest_dev_copy_from_host(int device, void *d_dst, const void *src, uint64_t bytes);
int dropest_dev_count(void);
int dropest_dev_sync(int device);

#ifdef __cplusplus
}
#endif
#endif
