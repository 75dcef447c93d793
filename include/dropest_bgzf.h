/* dropest_bgzf.h -- BGZF blocks inflated on the device (the BAM reader that feeds the path).
 *
 * What it replaces: BamTools' BgzfStream (zlib inflate, one block at a time) under BamReader::GetNextAlignment, the loop of
 * Estimation/BamProcessing/BamController.cpp:85.  A BGZF block (SAMv1 §4.1) is an independent DEFLATE stream of at most 64 KB:
 * dropest_bgzf_scan walks the block headers on the host (18 + 8 bytes per block), dropest_bgzf_inflate_device decodes every block
 * with one 64-lane wave whose lanes take different chunks of the block's symbol stream (csrc/k_inflate_par.h; csrc/k_inflate.h, one chain of
 * symbols per block, behind DROPEST_INFLATE_PAR=0 -- both written from RFC 1951).  Plain C, no torch types.
 *
 * Checked on the device: every Huffman code, distance and length, the input and output bounds, ISIZE, and the block's CRC-32 (the wave
 * that inflated a block reads it back: 64 partial CRCs joined in GF(2)).  A block the device refuses (status != 0: damaged, or a CRC that
 * does not match) is left for the caller to inflate elsewhere; nothing outside its own output range is written. */
#ifndef DROPEST_BGZF_H
#define DROPEST_BGZF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Walks the BGZF blocks of data[0 .. len): for block k the DEFLATE payload is data[in_off[k] .. + in_len[k]), its ISIZE bytes belong at
 * out_off[k] (running sum) and number out_len[k]; crc32[k] is the stored CRC-32 (may be NULL).  Stops at `cap` blocks, at a block that
 * is not complete inside the buffer, or at the end; *bytes_used = offset of the first byte not consumed.  0 = ok, 1 = not a BGZF header. */
int dropest_bgzf_scan(const uint8_t *data, uint64_t len, uint64_t cap, uint64_t *in_off, uint32_t *in_len, uint64_t *out_off,
                      uint32_t *out_len, uint32_t *crc32, uint64_t *n_blocks, uint64_t *bytes_used, uint64_t *out_total);

/* Every pointer is DEVICE memory of `device`; d_in 8-byte aligned, in_total = bytes of d_in that may be read.  Asynchronous on `stream`
 * (a hipStream_t, NULL = the default stream).  d_status[k] = 0 when block k came out whole. */
int dropest_bgzf_inflate_device(int device, void *stream, const uint8_t *d_in, uint64_t in_total, const uint64_t *d_in_off,
                                const uint32_t *d_in_len, const uint64_t *d_out_off, const uint32_t *d_out_len, uint32_t n_blocks,
                                uint8_t *d_out, uint32_t *d_status, const uint32_t *d_crc32 /* stored CRC-32 per block, or NULL: not checked */);

/* Host buffer in, host buffer out (tests, scripts/bench_bgzf_inflate.py): scan + upload + kernel (`repeats` times; *kernel_ms = mean
 * of the runs by HIP events; repeats < 0: |repeats| runs WITHOUT the CRC-32 check) + download.  status[0 .. min(n_blocks, status_cap)) per block.  Returns 0, or 1 with
 * dropest_bgzf_last_error() set (no GPU, out_cap too small, not BGZF). */
int dropest_bgzf_inflate_buffer(int device, const uint8_t *data, uint64_t len, uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                                uint32_t *status, uint64_t status_cap, uint64_t *n_blocks, double *kernel_ms, int repeats);

/* ---- BAM records of a window of BGZF blocks, on the device (csrc/k_bamparse.h) -------------------------------------------------------
 * What BamController::process_alignment (BamController.cpp:132-172) decides without a dictionary: the reader's status of a record, the
 * 2-bit codes of its barcode and UMI, its UMI::Mark, and -- against a copy of the caller's dictionaries -- its gene and chromosome index:
 * the accepted records leave as the four dense columns of dropest_push_reads IN DEVICE MEMORY (dropest_push_reads_device takes them as
 * they are).  The dictionaries themselves (gene names, chromosomes, strings with N) stay with the caller, who asks for the bytes of the
 * few records that bring something new, resolves them in file order and patches their rows. */
typedef struct dropest_bam_decoder dropest_bam_decoder;

typedef struct dropest_bam_parse_cfg {
	uint16_t tag[6];          /* cell barcode, UMI, barcode quality, UMI quality, gene, read type (BamTags.cpp:7-24): letters lo | hi << 8, 0 = none */
	int32_t filled_bam;       /* -f: 1 = barcode / UMI from tags (FilledBamParamsParser), 0 = from the read name "id!CB#UMI" */
	int32_t min_phred;        /* min_barcode_phred as a character code (33 + score); the filter is on when > 33 */
	int32_t has_read_type;
	int32_t n_refs;
	uint32_t intronic_len, intergenic_len;
	uint8_t intronic[24], intergenic[24];
} dropest_bam_parse_cfg;

enum { DROPEST_BAM_OK = 0, DROPEST_BAM_SKIP = 1, DROPEST_BAM_CANT_PARSE_NO_COUNT = 2, DROPEST_BAM_CANT_PARSE = 3, DROPEST_BAM_LOW_QUALITY = 4 };

typedef struct dropest_bam_window {        /* host pointers: pinned memory of the decoder; d_*: DEVICE memory; all valid until its next window */
	uint64_t n_records;                    /* complete records that START in this window (the cut-off last one opens the next window) */
	uint64_t counts[5];                    /* records per DROPEST_BAM_* status */
	uint64_t n_accepted;                   /* = counts[DROPEST_BAM_OK]: rows of the dense columns below, in file order */
	const uint64_t *d_cb, *d_umi;          /* the columns of dropest_push_reads (dropest_amd.h), as BamController's bulk path fills them: */
	const uint32_t *d_gene, *d_aux;        /*   gene = index from the dictionary given to the decoder, aux = chromosome index | mark << 16 */
	uint32_t n_need;                       /* accepted records the caller must see as bytes (a gene or chromosome the dictionaries lack, an N): */
	const uint32_t *need_rec, *need_pos, *need_size;   /* record of the window, its row in the dense columns, its bytes (block_size field included) */
	uint32_t quality_seen, any_gene;       /* an accepted record carries a UMI quality tag / a gene name */
	uint64_t window_bytes, tail_bytes;     /* inflated bytes looked at; bytes of the cut-off last record carried over */
	uint32_t n_blocks, refused_blocks;     /* blocks the device left to `inflate_fallback` */
	uint32_t guesses_repaired, pad;        /* segments whose guessed first record was not on the chain (walked again from the true place) */
	double ms_inflate, ms_boundaries, ms_parse, ms_copy;
	uint32_t quality_len_min, quality_len_max;   /* shortest / longest UMI quality string over the accepted GENE-BEARING reads (0 = none) */
} dropest_bam_window;

/* raw DEFLATE of one block on the host (zlib or the like) for the blocks the device refuses; 0 = ok */
typedef int (*dropest_bgzf_host_inflate)(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, void *user);

/* Loads this path's kernels onto the current device (their first launch otherwise does, ~10 ms); dropest_ctx_create calls it. */
void dropest_bgzf_warm_up(void *stream);
int dropest_bam_decoder_create(int device, const dropest_bam_parse_cfg *cfg, dropest_bam_decoder **out);
/* The decoder for another file: its device buffers, streams and pinned memory stay (creating and freeing them is ~60 ms per file), the
 * dictionaries are emptied, the annotation and the record carried over are dropped. */
int dropest_bam_decoder_reset(dropest_bam_decoder *d, const dropest_bam_parse_cfg *cfg);
/* Blocks the device inflates at the same time (one wave each): a window of that many blocks takes about as long as a window of fewer. */
uint32_t dropest_bam_decoder_wave_slots(const dropest_bam_decoder *d);
/* Two pinned host buffers of the decoder for the caller's compressed bytes (which = 0 / 1, at least `bytes` long): a window handed over from
 * one of them crosses PCIe at the link's rate; any other host memory works too (pageable memory is staged by the runtime at a third of it). */
int dropest_bam_decoder_staging(dropest_bam_decoder *d, int which, uint64_t bytes, uint8_t **out);
/* The first `len` bytes of staging buffer `which` start for the device now, on a stream of their own; the window call that is given exactly that
 * buffer and length then waits for this copy instead of making one.  For a reader thread that calls it when its read is done: the copy of
 * window k + 1 runs under the kernels of window k.  The one decoder call that may run beside a window call; allocates nothing; a length the
 * buffers were not sized for is left to the window call (0 is returned all the same). */
int dropest_bam_decoder_upload(dropest_bam_decoder *d, int which, uint64_t len);
/* The same road in pieces, so that the pinned memory (0.15-0.4 ms per MB to allocate: 35-65 ms for two windows of 128 MB) does not grow with the
 * window: .._reserve sizes the device side of up-buffer / front `which` for windows of `bytes` compressed bytes; .._pieces makes n pinned pieces
 * of `bytes` each (out[k] = piece k); a reader fills a piece (.._piece_wait(piece) first: the copy that read it last is through), sends its
 * first `len` bytes to byte `dst_off` of up-buffer `which` (.._upload_piece: asynchronous, callable from several threads on different pieces),
 * and when a window's pieces are all on their way says what the buffer holds: .._upload_done(which, host, len, blocks) -- `host` are the same bytes in
 * host memory of any kind (the mapped file), from which the block table is made and which the window call is then given
 * (dropest_bam_decoder_window(d, host, len, ...)); they stay readable until that call has returned. */
int dropest_bam_decoder_reserve(dropest_bam_decoder *d, int which, uint64_t bytes, uint64_t inflated_bytes /* what such a window inflates to, if the caller knows (0: 14 x bytes) */);
int dropest_bam_decoder_pieces(dropest_bam_decoder *d, uint32_t n, uint64_t bytes, uint8_t **out);
int dropest_bam_decoder_piece_wait(dropest_bam_decoder *d, uint32_t piece);
int dropest_bam_decoder_upload_begin(dropest_bam_decoder *d, int which);   /* before a window's first piece: picks the stream its pieces travel on */
int dropest_bam_decoder_upload_piece(dropest_bam_decoder *d, int which, uint32_t piece, uint64_t dst_off, uint64_t len);
/* blocks != NULL: the window's block table as the caller found it while reading the file (block k: payload at in_off[k], in_len[k] bytes long, ISIZE
 * out_len[k], CRC-32 crc32[k]); `host` is then read only if the device refuses a block. */
typedef struct { uint64_t n; const uint64_t *in_off; const uint32_t *in_len, *out_len, *crc32; } dropest_bgzf_blocks;
int dropest_bam_decoder_upload_done(dropest_bam_decoder *d, int which, const uint8_t *host, uint64_t len, const dropest_bgzf_blocks *blocks);
/* The decoder's kernels run on `stream` (a hipStream_t of its device that the caller lends: one that exists and has run kernels saves the ~16 ms
 * a new stream and its first dispatch cost) until NULL gives it back -- before that stream is destroyed.  Not while a window is in flight. */
int dropest_bam_decoder_use_stream(dropest_bam_decoder *d, void *stream);
void dropest_bam_decoder_destroy(dropest_bam_decoder *d);
/* comp[0 .. len): whole BGZF blocks, following the previous window's.  first_skip: bytes of the first block to pass over (the BAM header;
 * first window only).  final != 0: the file ends here (a cut-off record is then an error). */
int dropest_bam_decoder_window(dropest_bam_decoder *d, const uint8_t *comp, uint64_t len, uint32_t first_skip, int final,
                               dropest_bgzf_host_inflate inflate_fallback, void *user, dropest_bam_window *out);
/* The same in two halves, for a caller that overlaps windows: _begin (copy in, inflate, the chain of records; needs no dictionary) of window
 * k + 1 may run on ANOTHER THREAD while _finish (fields against the dictionaries, dense columns) of window k and the caller's own work go on.
 * Windows are begun in file order, one at a time; a window is finished before the one after the next is begun (two sets of buffers). */
int dropest_bam_decoder_window_begin(dropest_bam_decoder *d, const uint8_t *comp, uint64_t len, uint32_t first_skip, int final,
                                     dropest_bgzf_host_inflate inflate_fallback, void *user, int *slot);
int dropest_bam_decoder_window_finish(dropest_bam_decoder *d, int slot, dropest_bam_window *out);
/* _begin in two parts, for ONE thread that keeps the device busy (round 6; what BamController::parse_bam_files does on the device path,
 * BamProcessing/BamController.cpp:70-116): _inflate enqueues the copies and the inflate kernel of a window and returns at once -- it needs
 * nothing from the window before, so window k + 1 is given to the device BEFORE the caller waits for window k; _chain then waits for the
 * inflate of `slot`, puts the record the window before cut off in front of the data and finds the chain of records (windows take _chain in
 * file order).  The order per window k:  _inflate(k + 1); _chain(k); _finish(k).  comp must stay untouched until _chain has returned. */
int dropest_bam_decoder_window_inflate(dropest_bam_decoder *d, const uint8_t *comp, uint64_t len, int *slot);
int dropest_bam_decoder_window_chain(dropest_bam_decoder *d, int slot, uint32_t first_skip, int final, dropest_bgzf_host_inflate inflate_fallback, void *user);
/* The dictionaries the records are looked up in (copied): gene_hash[k] (FNV-1a of the name) -> gene_id[k]; chr_of_ref[r] = chromosome
 * index of reference r, -1 = none yet.  Call again whenever they have grown. */
int dropest_bam_decoder_set_dictionaries(dropest_bam_decoder *d, const uint64_t *gene_hash, const uint32_t *gene_id, uint32_t n_genes,
                                         const int32_t *chr_of_ref, uint32_t n_refs);
/* The names of the dictionary's genes by index (name k = pool[off[k] .. off[k + 1]), n_names + 1 offsets): a gene found by its hash is then
 * confirmed byte by byte on the device, and a name that only shares the FNV-1a value of another comes back to the host like an unseen one
 * (CellsDataContainer::intern_gene compares the bytes: Estimation/StringIndexer.cpp:10-18 keys by the string itself).  Optional: without it
 * the hash alone decides.  Call after _set_dictionaries, whenever the dictionary has grown. */
int dropest_bam_decoder_set_gene_names(dropest_bam_decoder *d, const uint32_t *off, const uint8_t *pool, uint32_t n_names);
/* -g: genes from a GTF / BED annotation instead of the gene tag (ReadParamsParser::get_gene_from_reference, ReadParamsParser.cpp:92-151): the
 * decoder asks `a` (dropest_annotation.h, same GPU; stays the caller's) about the two ends of every accepted alignment.  ann_chr_of_ref[r] =
 * the annotation's chromosome for reference r, -1 = it has none of that name (such a record cannot be parsed, BamController.cpp:153-161).
 * _annotation_genes: id_of_ann_gene[g] = index of the annotation's gene g in the caller's gene dictionary, -1 = not in it yet (a record
 * that names it comes back through need_*); call again whenever the dictionary has grown. */
struct dropest_annotation;
int dropest_bam_decoder_set_annotation(dropest_bam_decoder *d, struct dropest_annotation *a, const int32_t *ann_chr_of_ref, uint32_t n_refs);
int dropest_bam_decoder_set_annotation_genes(dropest_bam_decoder *d, const int32_t *id_of_ann_gene, uint32_t n);
/* the bytes of records idx[0 .. n) of the LAST window (block_size field first), one after the other: dst_off[k] = where record idx[k] starts in dst */
int dropest_bam_decoder_fetch_records(dropest_bam_decoder *d, const uint32_t *idx, uint32_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *dst_off);
/* The dense columns of the LAST window copied to host arrays of n_accepted entries each (any of them may be NULL): for a caller that keeps its
 * reads on the host after all. */
int dropest_bam_decoder_columns_to_host(dropest_bam_decoder *d, uint64_t *cb, uint64_t *umi, uint32_t *gene, uint32_t *aux);
/* One row of ql bytes per accepted read of the LAST window, in the order of the dense columns, in pinned HOST memory: the read's UMI quality string
 * (zeros for a read without a gene).  For a window whose quality_len_min == quality_len_max == ql. */
int dropest_bam_decoder_quality_rows(dropest_bam_decoder *d, uint32_t ql, const uint8_t **rows);
/* rows pos[0 .. n) of the LAST window's dense columns take these values (what the caller resolved for the `need` records) */
int dropest_bam_decoder_patch(dropest_bam_decoder *d, const uint32_t *pos, const uint64_t *cb, const uint64_t *umi, const uint32_t *gene,
                              const uint32_t *aux, uint32_t n);

const char *dropest_bgzf_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
