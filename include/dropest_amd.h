/* dropest_amd.h -- C-ABI of the MI355X-native dropEst Estimation hot path.
 *
 * The reference (kharchenkolab/dropEst v0.8.6) has no plugin/FFI seam; the seam of this path is the
 * C++ class Estimation::CellsDataContainer (Estimation/CellsDataContainer.h:33-123).  Every entry point
 * below names the reference interface it replaces (file:line under the reference tree).  The C++ facade
 * `dropest_amd::CellsDataContainer` (dropest_amd/csrc/host/CellsDataContainer.h) re-creates the reference
 * class on top of exactly these calls; INTEGRATION.md shows the binding a dropEst maintainer would add.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no exceptions cross the boundary: every call returns a
 *     dropest_status and dropest_last_error() holds the message of the last failure on this thread
 *     (the reference throws std::runtime_error / std::out_of_range, dropest.cpp:322-336).
 *   - result getters follow "ask the size, then pass a caller-owned buffer of that size".
 *   - one context = one container = one GPU (the reference container is single-threaded; Stats keeps
 *     process-global statics, Estimation/Stats.cpp:5-7).  Contexts are not thread-safe.
 *
 * Packed read record (replaces Estimation::ReadInfo, Estimation/ReadInfo.h:9-24 and
 * Tools::ReadParameters, Tools/ReadParameters.h:9-50), structure-of-arrays, one entry per read in
 * stream (= BAM record) order:
 *   cb[i], umi[i] : uint64 "2-bit code": bases A=0 C=1 G=2 T=3, first base most significant, with a
 *                   sentinel 1 bit above the top base: code = (1 << 2*len) | bases, len <= 31.
 *                   Strings that do not fit (an 'N', other letters, len > 31) are ESCAPED:
 *                   code = DROPEST_ESCAPE | k, k = index into the side-string table registered with
 *                   dropest_set_side_strings().  The table is shared by barcodes and UMIs; entries used as
 *                   UMIs MUST be registered in the order in which those UMI strings first occur on
 *                   gene-bearing reads (= their StringIndexer order, Gene.cpp:19): the N-UMI merge inserts
 *                   them into a std::unordered_set in that order (MergeUMIsStrategySimple.cpp:31-41) and
 *                   the random fills depend on it.  The umi[] entry of a read WITHOUT a gene is ignored
 *                   (such reads never reach Gene::add_umi) and must not register a side string.
 *   gene[i]       : dense id of the gene NAME in first-seen order over gene-bearing reads
 *                   (= StringIndexer ids, Estimation/StringIndexer.cpp:10-18), DROPEST_NO_GENE when the
 *                   read has no gene (ReadInfo::gene empty, CellsDataContainer.cpp:73-78).
 *   aux[i]        : chromosome id in bits 0..15, UMI::Mark bits (UMI.h:16-22: 1 not-annotated, 2 exon, 4 intron)
 *                   in bits 16..23.  Chromosome ids are first-seen dense ids of the chromosome NAME over the
 *                   reads that reach Stats::inc(chr) (Stats.cpp:22-27,:81-90): reads without a gene and reads
 *                   whose mark has the exon or intron bit (CellsDataContainer.cpp:73-78,:312-321).  The
 *                   chromosome field of any other read is ignored.
 */
#ifndef DROPEST_AMD_H
#define DROPEST_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DROPEST_ESCAPE   0x8000000000000000ull
#define DROPEST_NO_GENE  0xFFFFFFFFu

typedef enum {
	DROPEST_OK = 0,
	DROPEST_ERR_INVALID = 1,        /* bad argument / call order (reference: std::runtime_error) */
	DROPEST_ERR_RANGE = 2,          /* index out of range (reference: std::out_of_range) */
	DROPEST_ERR_DEVICE = 3,         /* HIP runtime failure, no GPU, allocation failure */
	DROPEST_ERR_UNSUPPORTED = 4,    /* input shape outside this build's limits */
	DROPEST_ERR_IO = 5              /* whitelist file unreadable / malformed */
} dropest_status;

/* Which CB-merge strategy (MergeStrategyFactory::get_cb_strat, Estimation/Merge/MergeStrategyFactory.cpp:61-103) */
enum { DROPEST_MERGE_NONE = 0,          /* DummyMergeStrategy.h:12-17 (no -m) */
       DROPEST_MERGE_REAL_BARCODES = 1, /* RealBarcodesMergeStrategy.cpp (-m + barcodes_file) */
       DROPEST_MERGE_SIMPLE = 2,        /* SimpleMergeStrategy.cpp (-m without a barcodes file); uses max_cb_merge_edit_distance */
       DROPEST_MERGE_POISSON_REAL = 3,  /* PoissonRealBarcodesMergeStrategy.cpp + PoissonTargetEstimator.cpp (-M + barcodes_file):
                                           floating-point decisions, see DESIGN.md for the agreement bar */
       DROPEST_MERGE_POISSON_SIMPLE = 4,/* PoissonSimpleMergeStrategy.cpp (-M without a barcodes file) */
       DROPEST_MERGE_ALL = 5            /* MergeAllMergeStrategy.h:16-50 (Estimation.Merge.merge_type = "all") */ };
/* Whitelist file flavour (MergeStrategyFactory.cpp:23-59 barcodes_type) */
enum { DROPEST_BARCODES_INDROP = 0,     /* InDropBarcodesParser.cpp:15-48 */
       DROPEST_BARCODES_CONST = 1       /* ConstLengthBarcodesParser.cpp:23-68 */ };
/* UMI-merge strategy (MergeStrategyFactory::get_umi, :105-111) */
enum { DROPEST_UMI_MERGE_SIMPLE = 0,      /* MergeUMIsStrategySimple.cpp:21-102 (fix UMIs with N); random fills follow glibc's
                                             rand() after srand(42) (:15-19), from a generator the context owns */
       DROPEST_UMI_MERGE_DIRECTIONAL = 1  /* -u: MergeUMIsStrategyDirectional.cpp:18-116; the reference never seeds here, a fresh
                                             process draws from srand(1): the context's generator starts there */ };

/* Replaces the constructor arguments of CellsDataContainer (CellsDataContainer.h:82-85) together with
 * the Estimation.Merge.* keys read by MergeStrategyFactory (MergeStrategyFactory.cpp:26-58). */
typedef struct {
	int32_t device;                       /* HIP device ordinal */
	int32_t merge_kind;                   /* DROPEST_MERGE_* */
	int32_t barcodes_kind;                /* DROPEST_BARCODES_* */
	const char *barcodes_file;            /* whitelist path (Estimation.Merge.barcodes_file), may be NULL */
	int32_t min_genes_before_merge;       /* default 10 in the reference */
	int32_t min_genes_after_merge;        /* default 10; effective value = max(after, before) (MergeStrategyAbstract.cpp:8-11) */
	double  min_merge_fraction;           /* default 0.2 */
	int32_t max_cb_merge_edit_distance;   /* SimpleMergeStrategy: candidates need edit distance < this (SimpleMergeStrategy.cpp:68-69);
	                                         RealBarcodes ignores it (RealBarcodesMergeStrategy.cpp:111-114) */
	int32_t umi_merge_kind;               /* DROPEST_UMI_MERGE_* */
	int32_t max_umi_merge_edit_distance;  /* default 1 */
	const char *gene_match_levels;        /* -L code, default "eEBA" (UMI.cpp:112-154) */
	int32_t max_cells;                    /* -C, <= 0: unlimited (CellsDataContainer.cpp:269-273) */
	uint64_t cb_table_capacity;           /* 0 = auto; power of two >= 2 x distinct barcodes otherwise */
	double  umi_merge_multiplier;         /* Estimation.Merge.umi_merge_multiplier, default 2 (MergeStrategyFactory.cpp:58) */
	double  max_merge_prob;               /* Estimation.PreciseMerge.max_merge_prob, default 1e-4 (MergeStrategyFactory.cpp:54) */
	double  max_real_merge_prob;          /* Estimation.PreciseMerge.max_real_merge_prob, default 1e-7 (:55) */
} dropest_cfg;

typedef struct dropest_ctx dropest_ctx;

/* Fills *cfg with the reference defaults (MergeStrategyFactory.cpp:26-58; dropest.cpp:239-254). */
void dropest_cfg_defaults(dropest_cfg *cfg);

const char *dropest_last_error(void);

/* CellsDataContainer::CellsDataContainer (CellsDataContainer.cpp:20-37) + strategy construction
 * (MergeStrategyFactory.cpp:23-59, RealBarcodesMergeStrategy.cpp:12-20 loads the whitelist). */
dropest_status dropest_ctx_create(const dropest_cfg *cfg, dropest_ctx **out);
/* (The library never changes process-global state: no environment variables, no srand().  A caller that wants polled
 * completion signals sets HSA_ENABLE_INTERRUPT=0 itself before the ROCm runtime starts, as bench.py does -- DESIGN.md §5.) */
void dropest_ctx_destroy(dropest_ctx *ctx);

/* Side strings for escaped codes (barcodes / UMIs containing 'N' etc.); the table may only grow. */
dropest_status dropest_set_side_strings(dropest_ctx *ctx, const char *const *strings, uint64_t n);

/* CellsDataContainer::add_record (CellsDataContainer.cpp:59-88), batched.  Host pointers; the batch is copied into ONE
 * growing set of device arrays on a copy stream of its own -- straight from the caller's arrays when they are pinned
 * (hipHostMalloc / hipHostRegister), through two pinned staging buffers otherwise -- and the caller keeps ownership: the
 * arrays may be reused when the call returns.  Fails with DROPEST_ERR_INVALID after dropest_set_initialized
 * ("Container is already initialized", :61-62). */
dropest_status dropest_push_reads(dropest_ctx *ctx, const uint64_t *cb, const uint64_t *umi,
                                  const uint32_t *gene, const uint32_t *aux, uint64_t n);
/* Optional: room for n_total pushed reads up front (otherwise the device arrays grow geometrically while reads arrive). */
/* Several host ranges that follow one another in the stream, in one call (same meaning as dropest_push_reads on each of them in turn):
 * cb[k] .. aux[k] hold counts[k] reads.  A reader whose worker threads each produce a dense run of records hands them over without
 * concatenating them first. */
dropest_status dropest_push_reads_gather(dropest_ctx *ctx, uint64_t n_segments, const uint64_t *const *cb, const uint64_t *const *umi,
                                         const uint32_t *const *gene, const uint32_t *const *aux, const uint64_t *counts);
dropest_status dropest_reserve_reads(dropest_ctx *ctx, uint64_t n_total);
/* Same, for arrays already resident in this GPU's HBM (device pointers).  With adopt != 0 the context
 * uses the caller's buffers in place (no copy); they must stay alive and unmodified until destroy. */
dropest_status dropest_push_reads_device(dropest_ctx *ctx, const uint64_t *d_cb, const uint64_t *d_umi,
                                         const uint32_t *d_gene, const uint32_t *d_aux, uint64_t n, int adopt);

/* CellsDataContainer::set_initialized (CellsDataContainer.cpp:163-175): cell ids, UMI de-duplication,
 * per-cell sizes, real/filtered cells.  Runs the device pipeline. */
dropest_status dropest_set_initialized(dropest_ctx *ctx);
/* CellsDataContainer::merge_and_filter (CellsDataContainer.cpp:39-57): CB merge, UMI merge, final filter. */
dropest_status dropest_merge_and_filter(dropest_ctx *ctx);
/* Drops everything computed by set_initialized/merge_and_filter but keeps the pushed reads, so the same
 * resident stream can be processed again (bench steps).  No reference counterpart. */
dropest_status dropest_reset_results(dropest_ctx *ctx);

/* ---- accessors (CellsDataContainer.h:99-122) ---- */
typedef struct {                 /* one row per cell id (first-seen order, CellsDataContainer.cpp:64-69) */
	uint64_t barcode;            /* packed code of the barcode (Cell::barcode, Cell.cpp:100-103) */
	uint32_t first_read;         /* ordinal of the first read of this barcode */
	uint32_t n_genes;            /* Cell::size (Cell.cpp:120-123) */
	uint32_t requested_genes;    /* Cell::requested_genes_num (Cell.cpp:130-143) */
	uint32_t requested_umis;     /* Cell::requested_umis_num */
	int32_t  total_reads;        /* Stats::TOTAL_READS_PER_CB */
	int32_t  total_umis;         /* Stats::TOTAL_UMIS_PER_CB = Cell::umis_number (quirks of Stats.cpp:29-43, Cell.cpp:31-42 kept) */
	uint8_t  is_merged, is_excluded, is_real, pad;
} dropest_cell_row;

dropest_status dropest_total_cells(dropest_ctx *ctx, uint64_t *n);                 /* total_cells_number() */
dropest_status dropest_real_cells(dropest_ctx *ctx, uint64_t *n);                  /* real_cells_number() */
dropest_status dropest_cell_rows(dropest_ctx *ctx, uint64_t first, uint64_t count, dropest_cell_row *out); /* cell(i) */
dropest_status dropest_cell_id_by_cb(dropest_ctx *ctx, uint64_t barcode, int64_t *id); /* cell_id_by_cb(); -1 if absent */
dropest_status dropest_filtered_cells(dropest_ctx *ctx, uint64_t *n, uint64_t *ids);   /* filtered_cells(); ids may be NULL */
/* merge_targets(): only entries with target != self are returned (pairs source -> target, ascending
 * source id); every other cell is its own target (MergeStrategyBase.cpp:13-15). */
dropest_status dropest_merge_targets(dropest_ctx *ctx, uint64_t *n, uint64_t *src, uint64_t *tgt);
/* [0] intergenic_reads_num [1] has_exon_reads_num [2] has_intron_reads_num [3] has_not_annotated_reads_num */
dropest_status dropest_global_counters(dropest_ctx *ctx, uint64_t out[4]);

/* Molecule table of one cell: rows (gene id, umi code, read count, mark), ascending (gene, umi code).
 * Replaces Cell::genes() / Gene::umis() walks (Cell.h:19, Gene.h:19). */
dropest_status dropest_cell_molecules(dropest_ctx *ctx, uint64_t cell, uint64_t *n, uint32_t *gene,
                                      uint64_t *umi, uint32_t *reads, uint8_t *mark);

/* The container's public mutators (the reference's strategies and tests call them directly); they act on the
 * initialised state.
 *   exclude_cell   CellsDataContainer::exclude_cell (CellsDataContainer.cpp:106-109)
 *   merge_cells    CellsDataContainer::merge_cells (:90-104): union of the molecules (counts add, marks OR), Stats::merge,
 *                  the source becomes merged.  Supported between real-candidate cells (what strategies merge).
 *   merge_umis     CellsDataContainer::merge_umis (:209-213) -> Cell::merge_umis (Cell.cpp:31-42): the (source, target)
 *                  pairs are applied in the given order (the reference walks the caller's unordered_map); a pair with
 *                  source == target is skipped, a missing source is DROPEST_ERR_INVALID ("Source UMI doesn't belong to
 *                  the gene"); UMIs in the packed / escaped form of dropest_push_reads (side strings already set). */
dropest_status dropest_exclude_cell(dropest_ctx *ctx, uint64_t cell);
dropest_status dropest_merge_cells(dropest_ctx *ctx, uint64_t source_cell, uint64_t target_cell);
dropest_status dropest_merge_umis(dropest_ctx *ctx, uint64_t cell, uint32_t gene, uint64_t n, const uint64_t *source_umis,
                                  const uint64_t *target_umis);
/* CellsDataContainer::add_umi_to_cell (CellsDataContainer.h:90, CellsDataContainer.cpp:356-364) on the initialised container:
 * one more read of `umi_code` (packed as on the way in) for gene index `gene` in cell `cell` with mark bits `mark` --
 * Gene::add_umi: a new molecule (TOTAL_UMIS_PER_CB + 1) or read_count + 1 / mark OR of the existing one; like the reference's
 * member it touches no read counter and no chromosome statistic.  The gene index and the UMI must fit the key layout the
 * container was initialised with (a UMI of the container's length, a gene index below the next power of two);
 * DROPEST_ERR_UNSUPPORTED otherwise.  On a container with UMI qualities the read's quality string is added to the molecule's
 * sums (UMI::add_read, UMI.cpp:21-34); a length other than the container's fails with the reference's message. */
dropest_status dropest_add_umi_to_cell(dropest_ctx *ctx, uint64_t cell, uint32_t gene, uint64_t umi_code, uint32_t mark,
                                       const uint8_t *umi_quality, uint32_t quality_length);
/* CellsDataContainer::umi_indexer() (CellsDataContainer.h:118): the UMIs in index order -- the order of first appearance among
 * the gene-bearing reads (Gene::add_umi -> StringIndexer::add, Gene.cpp:19), followed by the UMIs that only a merge brought in
 * (random fills of N-UMIs: Gene.cpp:47), those in (cell id, gene index, UMI code) order of their groups (the reference appends
 * them in the iteration order of its merge; no reference test pins that order).  Produced on demand: one pass over the resident
 * UMI column.  umi_codes = NULL returns the count. */
dropest_status dropest_umi_first_seen(dropest_ctx *ctx, uint64_t *n, uint64_t *umi_codes);

/* UMI base qualities (ReadParameters::umi_quality, Tools/ReadParameters.h:9-50): one fixed-length string per pushed
 * read, in push order, quality_length bytes each (host memory; raw phred+33 characters as in the BAM tag).  Call after
 * the last push and before set_initialized.  The container then accumulates the per-position sums of every molecule
 * (UMI::add_read, UMI.cpp:21-34) and carries them through the merges the way the reference does: UMI::merge does
 * NOT add qualities (UMI.cpp:15-19), a molecule copied into another cell / re-keyed to a new UMI takes its sums along
 * (Gene.cpp:26-58).  The reference allows a different length per molecule and throws on a mismatch inside one
 * (UMI.cpp:26-28); one length per container covers what its BAM readers produce. */
dropest_status dropest_set_umi_qualities(dropest_ctx *ctx, const uint8_t *qualities, uint32_t quality_length, uint64_t n_reads);
/* The same for strings of several lengths: rows of row_bytes bytes (shorter strings padded with anything), lengths[i] <= row_bytes
 * characters of read i count.  The reference fixes a molecule's quality length when the molecule is created (Gene::add_umi, Gene.cpp:20:
 * UMI(read_info.params.umi_quality().length(), 0)) and throws "Wrong quality length: <got>, expected: <molecule's>" at the first later read
 * of the molecule with another length (UMI::add_read, UMI.cpp:26-28).  Here the check runs in set_initialized: DROPEST_ERR_INVALID with
 * exactly that text for the EARLIEST such read of the stream -- the read whose add_record would have thrown.  lengths = NULL: all row_bytes. */
dropest_status dropest_set_umi_qualities_var(dropest_ctx *ctx, const uint8_t *qualities, uint32_t row_bytes, const uint8_t *lengths, uint64_t n_reads);
dropest_status dropest_umi_quality_length(dropest_ctx *ctx, uint32_t *quality_length);   /* the row width (the longest string); 0 when none were given */
/* UMI::_sum_quality of the molecules of one cell, in the order of dropest_cell_molecules: n x quality_length sums
 * (UMI::mean_quality, UMI.cpp:46-55, is (sum - 33) / read_count in unsigned integer arithmetic).  n must equal the
 * cell's molecule count. */
dropest_status dropest_cell_molecule_qualities(dropest_ctx *ctx, uint64_t cell_id, uint64_t n, uint32_t *quality_sums);
/* UMI::_sum_quality.size() of the same molecules (one length per container unless dropest_set_umi_qualities_var gave several: a molecule
 * keeps the length of the read that created it; sums beyond it are 0); all 0 on a container without qualities. */
dropest_status dropest_cell_molecule_quality_lengths(dropest_ctx *ctx, uint64_t cell_id, uint64_t n, uint32_t *lengths);
/* Whole molecule table, ascending (cell id, gene id, umi code). */
dropest_status dropest_molecules(dropest_ctx *ctx, uint64_t *n, uint32_t *cell, uint32_t *gene, uint64_t *umi,
                                 uint32_t *reads, uint8_t *mark);

/* Count matrices as triplets; replaces ResultsPrinter::get_count_matrix_filtered
 * (Estimation/ResultsPrinter.cpp:334-361; columns = filtered cells ascending, values = requested UMIs,
 * or reads with reads_output) and ::get_count_matrix_raw (:363-396; columns = real cells in cell-id
 * order, values = all UMIs).  Rows are gene ids.  Triplets come column-major, genes ascending in a column. */
dropest_status dropest_count_matrix(dropest_ctx *ctx, int filtered, int reads_output, uint64_t *nnz,
                                    uint32_t *gene, uint32_t *col, uint32_t *val);
/* The same matrices in compressed-sparse-column form -- what ResultsPrinter::create_matrix finally builds
 * (Eigen triplets -> dgCMatrix, ResultsPrinter.cpp:433-442): colptr[ncols + 1], rowidx[nnz] (gene ids,
 * ascending inside a column), values[nnz].  Zero-copy: the three pointers refer to context-owned (pinned)
 * host memory that stays valid until the next call for the same `filtered` or until destroy. */
dropest_status dropest_count_matrix_csc(dropest_ctx *ctx, int filtered, int reads_output, uint64_t *ncols,
                                        uint64_t *nnz, const uint32_t **colptr, const uint32_t **rowidx,
                                        const uint32_t **values);
/* How the 32-bit slots of a large matrix (>= 2^18 entries) reach the host.  On (the default): the device emits the byte form
 * below, the copy is cut into chunks of whole columns and the library's host threads widen every chunk into rowidx / values as
 * soon as it has landed (csrc/matrix_decode.h) -- 2 bytes per entry on the PCIe link instead of 8, the decode hidden under the
 * copy; a matrix whose row list would not fit (very sparse columns) is emitted as 32-bit arrays after all.  Off: always 32-bit
 * arrays over the link.  Same results either way; DROPEST_MATRIX_DIRECT=1 in the environment switches it off for the process. */
dropest_status dropest_set_matrix_wire(dropest_ctx *ctx, int enabled);

/* Molecule keys of two words (Estimation/StringIndexer.cpp:10-18 hands out size_t indices: gene and UMI have no width limit there).
 * One context sorts cell | gene | UMI in ONE 64-bit word with the UMI's own 2-bit code as its field.  When the gene and UMI fields
 * alone reach 64 bits (a 24-base UMI beside 2^16 genes) the pass builds a dictionary of the stream's UMIs on the device
 * (csrc/k_umidict.h: the distinct clean UMIs of the gene-bearing reads, ascending) and the key carries a UMI's RANK in it -- never
 * more than 32 bits; ranks ascend with the codes, so every observable order stays what the plain layout gives.
 * mode 0 (default): only then.  1: also before a key wider than 64 bits would be refused with "sort key needs ..." (one context
 * then does what dropest_ctx_split does with several).  2: always (tests).  DROPEST_UMI_DICT in the environment overrides the mode.
 * Not in sharded / split runs (ranks are local to a context): there gene + UMI >= 64 bits stays DROPEST_ERR_UNSUPPORTED. */
dropest_status dropest_set_umi_dictionary(dropest_ctx *ctx, int mode);

/* ResultsPrinter::save_results (ResultsPrinter.cpp:23-79) always builds both matrices.  This call starts cm_raw on a
 * second stream -- emit kernel and the device-to-host copy -- and returns at once; what the caller does next (the
 * ordering of the filtered cells, cm) runs under that copy.  A following dropest_count_matrix_csc(filtered = 0, same
 * reads_output) only waits for it; any call that changes the container in between discards the prefetch. */
dropest_status dropest_prefetch_raw_matrix(dropest_ctx *ctx, int reads_output);

/* The NARROW form of the same CSC matrices: 16-bit row indices and 16-bit values -- 4 bytes per entry over PCIe instead of 8,
 * which is what bounds a pass once the kernels are done (the two matrices of C2 are 0.29 GB as 2 x u32).  ResultsPrinter turns
 * every entry into a double anyway (Eigen triplets -> dgCMatrix, ResultsPrinter.cpp:433-442), so its writers read this form
 * directly.  Lossless: a value beyond 65534 is stored as 0xFFFF and listed exactly in (overflow_pos[k] = entry index,
 * overflow_val[k]), k < n_overflow, ascending by position.  Available when every gene id is below 65536
 * (dropest_narrow_matrix_possible); DROPEST_ERR_UNSUPPORTED otherwise, and the 32-bit form always works.  Pointers refer
 * to context-owned pinned memory, valid until the next matrix call for the same `filtered`. */
dropest_status dropest_narrow_matrix_possible(dropest_ctx *ctx, int *possible);
dropest_status dropest_count_matrix_csc_narrow(dropest_ctx *ctx, int filtered, int reads_output, uint64_t *ncols, uint64_t *nnz,
                                               const uint32_t **colptr, const uint16_t **rowidx, const uint16_t **values,
                                               uint64_t *n_overflow, const uint32_t **overflow_pos, const uint32_t **overflow_val);
dropest_status dropest_prefetch_raw_matrix_narrow(dropest_ctx *ctx, int reads_output);
/* The same matrix in the BYTE form: two bytes per entry over PCIe instead of eight, for any gene id.  Entry k of column c
 * (colptr[c] <= k < colptr[c + 1]; rows of a column ascend) carries
 *   row_delta[k] = row[k] - row[k - 1]  (the column's first entry: row[k] + 1, i.e. counted from row -1); 255 = listed: the exact ROW stands in
 *                  (row_listed_pos, row_listed_row), and the deltas after it count from that row
 *   value[k]     = the count; 255 = listed in (value_listed_pos, value_listed_value)
 * The lists come in no particular order (the device appends to them as it goes).  A cell with a few thousand of 30 000 genes has row gaps of ~10 and counts of a few UMIs: well under
 * 1 % of the entries are listed.  dropest_matrix_bytes_widen decodes into the dgCMatrix slots i / x on host threads (what ResultsPrinter
 * does while it writes its doubles); tests/test_gpu_narrow.py checks it against dropest_count_matrix_csc bit for bit.  More than
 * max(2^20, nnz / 8) listed rows or 2^20 listed values: DROPEST_ERR_UNSUPPORTED (take the 32-bit form: dropest_count_matrix_csc never refuses).  Pointers refer to pinned host memory of the
 * context, valid until the next emit of the same matrix. */
typedef struct dropest_matrix_bytes {
	uint64_t ncols, nnz;
	const uint32_t *colptr;
	const uint8_t *row_delta, *value;
	uint64_t n_row_listed;
	const uint32_t *row_listed_pos, *row_listed_row;
	uint64_t n_value_listed;
	const uint32_t *value_listed_pos, *value_listed_value;
} dropest_matrix_bytes;
dropest_status dropest_count_matrix_csc_bytes(dropest_ctx *ctx, int filtered, int reads_output, dropest_matrix_bytes *out);
dropest_status dropest_prefetch_raw_matrix_bytes(dropest_ctx *ctx, int reads_output);
dropest_status dropest_matrix_bytes_widen(const dropest_matrix_bytes *m, uint32_t *rowidx, uint32_t *values);
/* A matrix that RIDES on another one's rows (round 6: what dropest_count_matrix_csc(filtered = 1) does inside the library when cm_raw is on
 * its way in the byte form).  cm's columns are a subset of cm_raw's and a column's entries are cm_raw's entries of that cell with equal or
 * smaller values (get_count_matrix_filtered / _raw, ResultsPrinter.cpp:334-396; Cell::requested_umis_per_gene, Cell.cpp:54-68), so cm crosses
 * PCIe as ONE byte per entry of cm_raw: value[k] = the value entry k of `base` has in the rider (0: not an entry of it; 255: listed -- in
 * (listed_pos, listed_value), pos = the entry's place in the RIDER's slots).  base_rows = the base's widened row slots (dropest_matrix_bytes_widen
 * of `base`); column c of the base lands at [out_begin[c], out_begin[c] + out_count[c]) of (rowidx, values), out_begin[c] = 0xFFFFFFFF: not a
 * column of the rider.  DROPEST_ERR_INVALID when a column keeps another number of entries than announced.  Plain host code. */
dropest_status dropest_matrix_rider_widen(const dropest_matrix_bytes *base, const uint32_t *base_rows, const uint8_t *value, const uint32_t *out_begin,
                                          const uint32_t *out_count, uint64_t rider_nnz, uint64_t n_listed, const uint32_t *listed_pos,
                                          const uint32_t *listed_value, uint32_t *rowidx, uint32_t *values);
/* Announces that cm_raw will be asked for in `form` (0: 32-bit, 1: 16-bit, 2: bytes; -1 takes the announcement back) with UMI counts
 * (reads_output = 0) or read counts: the container then starts the prefetch by itself as early as the matrix is final -- at the end of
 * set_initialized when merge_and_filter cannot change it (no CB merge, the Simple UMI merge, no UMI with N: merge_and_filter then only
 * re-filters the cells), at the end of merge_and_filter otherwise -- and the copy to the host runs under the host work in between.
 * A setting of the context: it survives dropest_reset_results / dropest_clear_reads.  Results are the same with and without. */
dropest_status dropest_set_raw_matrix_prefetch(dropest_ctx *ctx, int form, int reads_output);

/* ResultsPrinter::get_count_matrix_filtered(container, query_marks) (ResultsPrinter.cpp:333-361) for a mark query other
 * than the container's own -- what ResultsPrinter::save_intron_exon_matrices asks for (-V: "e", "i", "BA",
 * ResultsPrinter.cpp:455-474).  Columns = the filtered cells in their order, zero entries dropped; same CSC
 * conventions and lifetime as dropest_count_matrix_csc (valid until the next call of this function). */
dropest_status dropest_count_matrix_csc_levels(dropest_ctx *ctx, const char *gene_match_levels, int reads_output,
                                               uint64_t *ncols, uint64_t *nnz, const uint32_t **colptr,
                                               const uint32_t **rowidx, const uint32_t **values);
/* Per-chromosome read counts of real cells (CellsDataContainer::get_stat_by_real_cells,
 * CellsDataContainer.cpp:291-307): rows (cell id, kind 0 exon / 1 intron / 2 intergenic, chr id, count),
 * non-zero entries only, ascending (cell, kind, chr). */
dropest_status dropest_chr_stats(dropest_ctx *ctx, uint64_t *n, uint32_t *cell, uint32_t *kind, uint32_t *chr,
                                 int32_t *count);

/* CellsDataContainer::umi_distribution (CellsDataContainer.cpp:182-197): number of molecules per UMI over the
 * FILTERED cells, ascending UMI code (the reference returns an unordered_map; this is its content, ordered). */
dropest_status dropest_umi_distribution(dropest_ctx *ctx, uint64_t *n, uint64_t *umi, uint64_t *count);
/* Tools::CollisionsAdjuster (Tools/CollisionsAdjuster.cpp:12-49): adjusted_size[s-1] for s = 1..max_expression, from
 * the UMI probability vector (PoissonTargetEstimator::init, PoissonTargetEstimator.cpp:46-60).  Evaluated on the
 * device in double precision, one launch pair per s (the recurrence over s is sequential); the inner sum over the
 * UMIs is a FIXED-ORDER parallel reduction, so results are reproducible run to run but may differ from the
 * reference's left-to-right sum in the last bits of `new_umi_prob` (the table itself is integer, after lround).
 * The reference pins this component only to 1e-2 (Tests/TestEstimationMergeProbs.cpp:113-140). */
dropest_status dropest_collisions_adjusted_sizes(int device, const double *umi_probabilities, uint64_t n,
                                                 uint64_t max_expression, uint64_t *adjusted_sizes);

/* PoissonTargetEstimator::estimate_intersection_prob (PoissonTargetEstimator.cpp:67-94; the reference's tests call
 * it directly, Tests/TestEstimationMergeProbs.cpp:93-111) for two cells of the un-merged container: the UMI-gene
 * intersection size, its expected size under independent sampling from the filtered cells' UMI distribution, and
 * P(Poisson(expected) >= intersection).  expected = -1 and probability = 1 when the intersection is empty (:72-75). */
dropest_status dropest_poisson_intersection_prob(dropest_ctx *ctx, uint64_t cell1, uint64_t cell2, uint64_t *intersection_size,
                                                 double *expected_intersection_size, double *merge_probability);
/* RealBarcodesMergeStrategy::get_merge_target (RealBarcodesMergeStrategy.cpp:22-29) for one cell,
 * evaluated on the un-merged state; valid between set_initialized and merge_and_filter. */
dropest_status dropest_merge_target(dropest_ctx *ctx, uint64_t cell, int64_t *target);

/* ---- multi-GPU building blocks (the reference has no distributed runtime) ----------------------
 * Reads are sharded by barcode: owner(cb) = dropest_owner_of(cb, n_parts).  All pointers are DEVICE pointers on
 * `device`.  dropest_partition_by_owner groups n reads by owner, STABLY (reads of one owner keep their stream
 * order, which keeps first-seen order meaningful after the exchange); out_idx[i] is the position the i-th output
 * read had in the input; counts[p] (host) is the number of reads of owner p.  One all-to-all(v) of the five
 * output arrays (RCCL) is the only data-path collective.  d_scratch: caller-owned device memory of at least
 * dropest_partition_scratch_bytes(n) bytes (the call allocates nothing). */
uint32_t dropest_owner_of(uint64_t barcode, uint32_t n_parts);
dropest_status dropest_partition_scratch_bytes(uint64_t n, uint64_t *bytes);
dropest_status dropest_partition_by_owner(int device, const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene,
                                          const uint32_t *d_aux, uint64_t n, uint32_t n_parts, uint64_t *d_out_cb,
                                          uint64_t *d_out_umi, uint32_t *d_out_gene, uint32_t *d_out_aux, uint32_t *d_out_idx,
                                          uint64_t *counts, void *d_scratch, uint64_t scratch_bytes);
/* Rows of the real-candidate cells (n_genes >= min_genes_before_merge at set_initialized) with the host-tracked
 * merge state: ids[k] = cell id, rows[k] as dropest_cell_rows would return it.  Ascending cell id. */
dropest_status dropest_real_candidate_rows(dropest_ctx *ctx, uint64_t *n, uint64_t *ids, dropest_cell_row *rows);
/* Plain device-to-device copy on `device` (lets a caller stage results into buffers it owns, e.g. torch tensors). */
dropest_status dropest_dev_copy_device(int device, void *d_dst, const void *d_src, uint64_t bytes);
/* Forgets the pushed reads (and all results) so that a new batch can be pushed into the same context. */
dropest_status dropest_clear_reads(dropest_ctx *ctx);
/* The device arrays of the reads pushed so far, for a second context that adopts them in place (dropest_push_reads_device
 * with adopt = 1): how the C++ facade answers accessors and mutators BEFORE set_initialized, which the reference allows
 * (its containers exist from the first add_record; Tests/TestEstimation.cpp:468-488 merges UMIs before initialising).  Valid
 * until the next push. */
dropest_status dropest_resident_reads(dropest_ctx *ctx, const uint64_t **d_cb, const uint64_t **d_umi, const uint32_t **d_gene,
                                      const uint32_t **d_aux, uint64_t *n);
/* As dropest_count_matrix_csc, but rowidx / values stay in HBM (device pointers) for a gather over RCCL;
 * colptr is a host pointer. */
dropest_status dropest_count_matrix_device(dropest_ctx *ctx, int filtered, int reads_output, uint64_t *ncols,
                                           uint64_t *nnz, const uint32_t **colptr, const uint32_t **d_rowidx,
                                           const uint32_t **d_values);
/* Device pointer to first_read of every cell (ascending: cell ids are first-seen ranks). */
dropest_status dropest_cell_first_reads_device(dropest_ctx *ctx, uint64_t *n_cells, const uint32_t **d_first);
/* Copies n_cols column segments (src_start, dst_start, len: host arrays) of (rows, vals) into their places in
 * the destination matrix: the final column permutation of the gathered per-shard matrices. */
dropest_status dropest_assemble_columns(int device, uint64_t n_cols, const uint64_t *src_start, const uint64_t *dst_start,
                                        const uint64_t *len, const uint32_t *d_src_rows, const uint32_t *d_src_vals,
                                        uint32_t *d_dst_rows, uint32_t *d_dst_vals);
/* Same without waiting: the copy kernel is queued on the device's default stream and the call returns; `slot` (0 or 1)
 * names the descriptor buffer to use -- a slot may be reused only after dropest_dev_sync.  Lets the host prepare the next
 * matrix while this one crosses PCIe. */
dropest_status dropest_assemble_columns_async(int device, int slot, uint64_t n_cols, const uint64_t *src_start,
                                              const uint64_t *dst_start, const uint64_t *len, const uint32_t *d_src_rows,
                                              const uint32_t *d_src_vals, uint32_t *d_dst_rows, uint32_t *d_dst_vals);

/* Host memory shared by the ranks of one node (e.g. a /dev/shm mapping): registered once, then each rank's
 * dropest_assemble_columns writes ITS columns of the global matrix straight into it through *d_ptr, so the final
 * matrix reaches the host over all PCIe links at once instead of through one GPU. */
dropest_status dropest_host_register(int device, void *host, uint64_t bytes, void **d_ptr);
dropest_status dropest_host_unregister(int device, void *host);

/* ---- sharded runs, continued: the two places where shards must agree --------------------------------------
 * (1) The sort key's gene / UMI fields must have one layout on every shard so that molecule rows can move between
 *     shards: dropest_ingest runs the first half of set_initialized (barcode table, cell ids, key statistics);
 *     the caller all-reduces the summary (min of umi_clean_min, max of the rest; the gene -> chromosome tables by
 *     element-wise min / max, a difference = conflict) and hands it back before dropest_set_initialized.
 * (2) The whitelist CB merge (RealBarcodesMergeStrategy.cpp:22-114 driven by MergeStrategyBase.cpp:11-57): a
 *     barcode's target can live on another shard.  Phases, with the caller's collectives between them:
 *       search    this shard's real cells (bases) against the all-gathered real cells of every shard
 *       export    molecule rows of the bases that need an intersection size (device arrays owned by the context)
 *       intersect pairs whose candidate lives here, against all-gathered base rows
 *       decide    targets of this shard's bases from the intersection sizes of its pairs
 *       dropest_merge_apply   (no context) the sequential application over the global compare_cells order
 *       finish    flags / stats of the local cells, local re-keying, import of the rows merged into local cells
 *     dropest_merge_and_filter then skips its own CB merge and continues with the UMI merge and the filtering. */
typedef struct {
	uint64_t umi_clean_min, umi_clean_max;   /* over clean UMI codes (sentinel included); min = ~0 if none */
	uint64_t umi_escape_max_plus1;           /* 1 + largest escaped UMI index, 0 if none */
	uint32_t gene_max_plus1, chr_max_plus1;
	uint32_t gene_chr_conflict, reserved;    /* a gene was counted on two chromosomes */
} dropest_ingest_summary;
dropest_status dropest_ingest(dropest_ctx *ctx);
dropest_status dropest_ingest_summary_get(dropest_ctx *ctx, dropest_ingest_summary *out);
dropest_status dropest_ingest_summary_set(dropest_ctx *ctx, const dropest_ingest_summary *global);
/* device table gene id -> chromosome id (0xFFFFFFFF = gene never counted on a chromosome), owned by the context */
dropest_status dropest_gene_chr_table(dropest_ctx *ctx, uint32_t **d_table, uint64_t *n);

/* g_*[n_global]: packed barcode, gene count and TOTAL_UMIS stat of the real cells of all shards (identical arrays on
 * every shard); base_global / base_local[n_bases]: this shard's real cells as indices into g_* and as local ids. */
dropest_status dropest_shard_merge_search(dropest_ctx *ctx, uint64_t n_global, const uint64_t *g_barcode, const uint32_t *g_n_genes,
                                          const int32_t *g_total_umis, uint64_t n_bases, const uint32_t *base_global,
                                          const uint32_t *base_local, uint64_t *n_pairs);
/* the (base, candidate) pairs of the search, as indices into g_*; pairs of one base are adjacent, bases ascending */
dropest_status dropest_shard_merge_pairs(dropest_ctx *ctx, uint32_t *pair_base_global, uint32_t *pair_cand_global);
/* rows of the bases that have pairs: listed_global[n_listed], row_offset[n_listed + 1] (arrays may be NULL to query
 * n_listed); *d_low = gene|UMI key fields, d_cols = {reads, mark, exon reads, intron reads}, row_offset[n_listed] rows */
dropest_status dropest_shard_merge_export(dropest_ctx *ctx, uint64_t *n_listed, uint32_t *listed_global, uint64_t *row_offset,
                                          const uint64_t **d_low, const uint32_t *d_cols[4]);
/* inter[p] = |UMI-genes(base p) n UMI-genes(candidate p)| (MergeStrategyBase.cpp:100-147); the base's rows are
 * [base_begin[p], base_end[p]) of the device array d_base_low, the candidate is this shard's cell cand_local[p] */
dropest_status dropest_shard_merge_intersect(dropest_ctx *ctx, uint64_t n_pairs, const uint32_t *cand_local, const uint64_t *base_begin,
                                             const uint64_t *base_end, const uint64_t *d_base_low, uint32_t *inter);
/* inter[n_pairs] in the order of dropest_shard_merge_pairs; target_global[n_bases] = index into g_* or -1 (exclude) */
dropest_status dropest_shard_merge_decide(dropest_ctx *ctx, const uint32_t *inter, int64_t *target_global);
/* MergeStrategyBase::merge_inited second loop (:30-51) with reassign (:64-82) over n_cells cells: step i merges cell
 * order[i] into target[i] (or excludes it when -1); total_reads / total_umis are updated as Stats::merge does;
 * final_target[c] = the cell holding c's molecules afterwards. */
dropest_status dropest_merge_apply(uint64_t n_cells, uint64_t n_order, const uint32_t *order, const int64_t *target, int32_t *total_reads,
                                   int32_t *total_umis, uint32_t *final_target, uint8_t *excluded);
/* local_id[n_local] with their new flags and stats; (move_src -> move_tgt)[n_moves]: merges between two local cells;
 * n_import device rows (target local cell, gene|UMI fields, {reads, mark, exon, intron}) merged in from other shards */
dropest_status dropest_shard_merge_finish(dropest_ctx *ctx, uint64_t n_local, const uint32_t *local_id, const uint8_t *excluded,
                                          const uint8_t *merged_away, const int32_t *total_reads, const int32_t *total_umis,
                                          uint64_t n_moves, const uint32_t *move_src, const uint32_t *move_tgt, uint64_t n_import,
                                          const uint32_t *d_cell, const uint64_t *d_low, const uint32_t *const d_cols[4]);

/* ---- instrumentation (no reference counterpart; Tools::trace_time stage stamps, Tools/Logs.cpp:63-71) ---- */
typedef struct {
	const char *name;      /* kernel family */
	uint32_t launches;
	double   ms;           /* sum of HIP-event durations on the context's stream */
	double   bytes;        /* algorithmic bytes moved by those launches */
} dropest_kernel_stat;
dropest_status dropest_kernel_stats(dropest_ctx *ctx, uint32_t *n, dropest_kernel_stat *out);
/* Sort-record layout chosen for the pushed reads: [0] cell bits [1] gene bits [2] UMI bits
 * [3] mark bits folded under the key [4] value bytes per record (0, 1 or 4) [5] passes of the main sort over the records
 * (LSD radix passes, or 3 = two partitions + the LDS-resident finishing sort) [6] which sort ran: 0 LSD radix sort,
 * 1 splitter sort (k_ssort.h). */
dropest_status dropest_sort_layout(dropest_ctx *ctx, uint32_t out[7]);
/* Digit windows (shift, bits) of the LSD radix sort for keys whose varying bits are `varying_mask` (host logic only, no
 * device needed; at most 8 passes): 8-bit windows, the top ones widened to 9 bits when that saves a pass. */
dropest_status dropest_radix_plan(uint64_t varying_mask, uint32_t *n_passes, int32_t shifts[8], int32_t bits[8]);
/* The first n values glibc's rand() returns after srand(seed), from the restated generator every context owns for the
 * random fills of N-UMIs (MergeUMIsStrategyAbstract.cpp:11-23; host logic only, no device needed). */
dropest_status dropest_rand_sequence(uint32_t seed, uint64_t n, int32_t *out);
/* Sizes of the container's tables after set_initialized: [0] reads [1] cells (distinct barcodes) [2] molecules
 * [3] (cell, gene) rows -- the terms of the compulsory-traffic figure of the pipeline roofline. */
dropest_status dropest_table_sizes(dropest_ctx *ctx, uint64_t out[4]);
dropest_status dropest_set_profiling(dropest_ctx *ctx, int enabled);   /* HIP events per launch; off by default */
/* Restrict the events to the launches whose stat name starts with `name_prefix` -- several prefixes may be given,
 * separated by '|' -- (NULL / "" = all launches and the host stages).  Two events per launch cost ~0.5 ms per C2 pass when every kernel is timed; bench.py times only the
 * dominant kernel inside its timed region and collects the full table in a separate pass. */
dropest_status dropest_set_profiling_filter(dropest_ctx *ctx, const char *name_prefix);
/* Debug aids of the device allocator (csrc/util.h; not on the product path).  dropest_debug_poison_scratch overwrites every
 * live device buffer of the process that is not an input (reads, whitelists, qualities stay) and every pinned staging buffer
 * with a pseudo-random pattern of `seed` -- between two passes of one context this turns "a buffer kept across passes still
 * holds the previous pass" into "it holds garbage"; a pass must give the same results either way (tests/test_gpu_reuse.py).
 * Only meaningful while no pass is in flight.  dropest_debug_trim_pool frees the blocks DROPEST_DEBUG_POOL=1 recycles. */
dropest_status dropest_debug_poison_scratch(uint64_t seed, uint64_t *n_blocks);
dropest_status dropest_debug_trim_pool(void);
/* The debug allocator's environment switches (csrc/util.h) are read when the library first allocates; a process that changes them
 * afterwards (the test suite does) calls this to have them read again.  DROPEST_DEBUG_REGISTRY=1 alone turns the registry on, which
 * dropest_debug_poison_scratch needs: without any switch allocations go straight to hipMalloc and are not tracked. */
dropest_status dropest_debug_refresh(void);
/* Allocations are numbered; with DROPEST_ALLOC_TRACE=1 the call site of each is kept: `next ordinal` brackets a pass, and
 * DROPEST_POISON_ZERO=a:b zero-fills the allocations numbered [a, b) -- how scripts/hunt_stale.py bisects a dependence. */
dropest_status dropest_debug_alloc_ordinal(uint64_t *next_ordinal);
dropest_status dropest_debug_alloc_site(uint64_t ordinal, char *out, uint64_t out_bytes);
/* The HIP stream all kernels of this context are launched on (hipStream_t). */
void *dropest_stream(dropest_ctx *ctx);

/* ---- the sharded runner: one dropest_shard = one shard = one context on one GPU (csrc/shard_run.h) -----------------
 * The reference has one CellsDataContainer fed by one thread (CellsDataContainer.h:33-123, dropest.cpp:245-252).  Here the
 * stream is cut into contiguous ordinal ranges, one per shard; a pass re-distributes the reads by owner(cb) =
 * dropest_owner_of(cb, world) with ONE all-to-all(v) over RCCL / xGMI, runs the single-GPU pipeline on every shard, merges
 * barcodes across shards (-m with a whitelist), resolves N-UMIs against the one global rand() sequence, and lets every
 * shard write ITS columns of the two count matrices into host memory shared by the node's shards.  Results equal ONE
 * container over the whole stream.
 *   dropest_shard_create        one process per GPU: `id` = the 128 bytes of dropest_shard_unique_id (called on rank 0,
 *                               distributed by the launcher -- torch.distributed, MPI, a file); builds the RCCL communicator
 *   dropest_shard_group_create  all shards in THIS process (the C++ facade owning N GPUs; several shards may share a device):
 *                               device-to-device / xGMI peer copies and a host barrier; drive each member from its own
 *                               host thread (dropest_shard_step) or all of them with dropest_shard_group_step
 *   dropest_shard_step          collective: every shard of the run calls it once per pass
 *   dropest_shard_matrix        (shard 0) the GLOBAL matrix in CSC form: colptr[ncols + 1], rowidx / values in the shared host
 *                               buffer, col_barcodes[ncols] = packed barcode of every column; valid until the next step
 * -u runs sharded too (the shards' UMI first-occurrence tables are reduced to one rank table, random fills come from agreed
 * offsets of the one rand() sequence), and so does -M with a whitelist (PoissonRealBarcodesMergeStrategy: the UMI histograms of the shards
 * are added, every shard builds the same estimator tables; UMI fields of at most 26 bits).  The merges without a whitelist (-m / -M without
 * barcodes, merge-all) run sharded too: the (UMI-gene -> cells) index of SimpleMergeStrategy::init is sharded by hash(UMI-gene), partial
 * counts of common UMI-genes travel to the owner of the base, ties are replayed over global cell indices (csrc/shard_merge_free.h). */
typedef struct dropest_shard dropest_shard;
dropest_status dropest_shard_unique_id(uint8_t id[128]);
dropest_status dropest_shard_create(const dropest_cfg *cfg, int32_t rank, int32_t world, const uint8_t id[128], dropest_shard **out);
dropest_status dropest_shard_group_create(const dropest_cfg *cfg, int32_t n, const int32_t *devices, dropest_shard **out /* [n] */);
void dropest_shard_destroy(dropest_shard *shard);
dropest_ctx *dropest_shard_ctx(dropest_shard *shard);   /* the shard's context: accessors, kernel statistics */
/* this shard's contiguous range of the stream, resident in ITS GPU's HBM (used in place: keep it alive); first_ordinal =
 * stream ordinal of its first read (ordinals define first-seen cell ids and N-UMI tie breaks across shards) */
dropest_status dropest_shard_set_reads_device(dropest_shard *shard, const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene,
                                              const uint32_t *d_aux, uint64_t n, uint64_t first_ordinal);
/* The same from host memory, batch by batch (pinned arrays are copied from in place, pageable ones staged; the copies run
 * on their own stream under the caller's next batch): every batch continues the shard's range (first_ordinal = end of the
 * previous batch).  The ranges of the shards must ascend with the rank and must not overlap -- received blocks, concatenated
 * in source-rank order, are then in stream order; dropest_shard_step checks it.  Ranges may be of any length (a feeder
 * that does not know the stream's length fills shard 0 up to a quota, then shard 1, ...: the exchange spreads the work). */
dropest_status dropest_shard_push_reads(dropest_shard *shard, const uint64_t *cb, const uint64_t *umi, const uint32_t *gene,
                                        const uint32_t *aux, uint64_t n, uint64_t first_ordinal);
/* UMI quality strings of the shard's resident reads (see dropest_set_umi_qualities): quality_length bytes per read, in the order the reads
 * were pushed / set; call on EVERY shard of the run (a shard without reads: n_reads = 0), same length everywhere, before
 * dropest_shard_step.  The strings travel with their reads in the exchange; the sums are accumulated where the cell lives and follow
 * the UMI merges there; in a barcode merge across shards the sums rows of the molecules that change shards travel with them, and a molecule
 * several merged cells had keeps the sums of the first of them in merge order, wherever they lived (Gene::merge, Gene.cpp:26-36). */
dropest_status dropest_shard_set_umi_qualities(dropest_shard *shard, const uint8_t *qualities, uint32_t quality_length, uint64_t n_reads);
/* The same with strings of several lengths (see dropest_set_umi_qualities_var): rows of row_bytes, lengths[i] <= row_bytes the length of
 * read i's string; the lengths travel with the reads (one byte each).  Every shard of the run must use the same call and row width. */
dropest_status dropest_shard_set_umi_qualities_var(dropest_shard *shard, const uint8_t *qualities, uint32_t row_bytes, const uint8_t *lengths,
                                                   uint64_t n_reads);
dropest_status dropest_shard_step(dropest_shard *shard);
dropest_status dropest_shard_group_step(dropest_shard *const *shards, int32_t n);   /* one host thread per shard */
/* The step ENDS with rowidx / values filled (shard options "byte_matrix" and "slots_matrix", both on by default): the 32-bit dgCMatrix
 * slots i / x of ResultsPrinter::create_matrix (Estimation/ResultsPrinter.cpp:433-442) in the node-shared host buffer -- every shard's
 * columns cross ITS PCIe link as bytes and are widened into the shared slots by that shard's host threads while the next chunk of
 * columns is on the link (csrc/matrix_decode.h), as dropest_count_matrix_csc does for one context.  With "slots_matrix" off the step ends
 * at the byte form and this call widens on first use. */
dropest_status dropest_shard_matrix(dropest_shard *shard, int filtered, uint64_t *ncols, uint64_t *nnz, const uint64_t **colptr,
                                    const uint32_t **rowidx, const uint32_t **values, const uint64_t **col_barcodes);
/* What the last step left in the shared buffer: 0 = 32-bit arrays, 1 = the 16-bit form, 2 = the byte form only, 3 = the byte form AND
 * the 32-bit slots widened from it inside the step (the default). */
dropest_status dropest_shard_matrix_form(dropest_shard *shard, int filtered, int32_t *form);
/* The same matrix in the narrow form (see dropest_count_matrix_csc_narrow): what the step writes when every gene id fits 16 bits
 * the shard option "narrow_matrix" is on (the default) and "byte_matrix" is OFF -- each shard then puts half the bytes on its PCIe link; with it
 * dropest_shard_matrix widens on the host on first use.  overflow_pos = GLOBAL entry indices, ascending. */
dropest_status dropest_shard_matrix_narrow(dropest_shard *shard, int filtered, uint64_t *ncols, uint64_t *nnz, const uint64_t **colptr,
                                           const uint16_t **rowidx, const uint16_t **values, const uint64_t **col_barcodes,
                                           uint64_t *n_overflow, const uint64_t **overflow_pos, const uint32_t **overflow_val);
/* The same matrix in the BYTE form (see dropest_matrix_bytes above; colptr is 32-bit here as there: a byte-form matrix has fewer than
 * 2^32 entries): what the step writes by default (shard option "byte_matrix", on; it wins over "narrow_matrix") -- every shard puts two
 * bytes per entry on its PCIe link, straight into the node-shared buffer, and appends what does not fit a byte to its own segment of
 * that buffer; the lists returned here are the segments of all shards, one after the other.  dropest_matrix_bytes_widen decodes it;
 * dropest_shard_matrix does that on first use.  A shard with more than min(2^20, nnz) listed entries of a kind fails the step with
 * DROPEST_ERR_UNSUPPORTED (switch byte_matrix off). */
dropest_status dropest_shard_matrix_bytes(dropest_shard *shard, int filtered, dropest_matrix_bytes *out, const uint64_t **col_barcodes);
/* (source, target) barcodes of the cells the CB merge folded, ascending source: CellsDataContainer::merge_targets by barcode */
dropest_status dropest_shard_merged_barcodes(dropest_shard *shard, uint64_t *n, uint64_t *source, uint64_t *target);
/* The column order of a global matrix from the all-gathered table of the real cells, as every shard computes it (host logic
 * only, no device): filtered = 1: cells with req_genes >= min_genes_after_merge in CellsDataContainer::compare_cells order
 * (CellsDataContainer.cpp:329-344, barcodes compared as strings), the last max_cells of them when max_cells > 0
 * (:269-273); filtered = 0: all rows by first_global = cell-id order of ONE container.  order[n_cols] = row indices. */
dropest_status dropest_plan_columns(uint64_t n, const uint64_t *barcode, const uint64_t *first_global, const uint32_t *n_genes,
                                    const uint32_t *req_genes, const uint32_t *req_umis, const int32_t *total_umis, int filtered,
                                    uint32_t min_genes_after_merge, int32_t max_cells, const char *const *side_strings, uint64_t n_side,
                                    uint64_t *n_cols, uint32_t *order);
/* wall time per phase of the steps so far (name, steps, ms; bytes = what this shard put on the links in "all_to_all") */
dropest_status dropest_shard_phase_stats(dropest_shard *shard, uint32_t *n, dropest_kernel_stat *out);
/* "trace" (synchronise the device at every phase boundary: diagnostic), "force_exchange" (partition + all-to-all even with
 * one shard: measures what a second shard would add), "reset_phase_stats", "byte_matrix" / "narrow_matrix" (the form the step writes
 * the matrices in: bytes by default, else 16-bit when every gene id fits, else 32-bit), "packed_exchange" */
dropest_status dropest_shard_set_option(dropest_shard *shard, const char *key, int64_t value);

/* Sort keys wider than 64 bits.  The reference's containers have no such limit (StringIndexer.cpp:10-18: ids are size_t); one
 * context packs cell id | gene | UMI into one 64-bit sort key and dropest_set_initialized fails with DROPEST_ERR_UNSUPPORTED
 * ("sort key needs ... bits") when the three fields do not fit.  dropest_key_width reports the field widths of that plan.  Only
 * the cell field depends on how much of the stream a context sees: dropest_ctx_split runs the SAME reads as `parts` (2..64)
 * shards on the context's own device -- split by barcode owner like a multi-GPU run, every shard numbering 1/parts of the
 * barcodes -- and writes the shard handles to out[parts]; drive them with dropest_shard_group_step and read the result with
 * dropest_shard_matrix / dropest_shard_merged_barcodes on out[0].  The shards borrow the context's device-resident reads: the
 * context must outlive them and must not be used for anything else meanwhile (its own tables are released).  Every merge and
 * the UMI qualities of the context (copied to the shards) are carried; the gene + UMI fields alone must leave room for the cells
 * of a shard. */
dropest_status dropest_key_width(dropest_ctx *ctx, uint32_t *cell_bits, uint32_t *gene_bits, uint32_t *umi_bits);
dropest_status dropest_ctx_split(dropest_ctx *ctx, int32_t parts, dropest_shard **out);

#ifdef __cplusplus
}
#endif
#endif /* DROPEST_AMD_H */
