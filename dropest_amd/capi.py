"""ctypes binding of the C-ABI in include/dropest_amd.h (the product path; fails loudly without the
HIP library or without a GPU -- there is no CPU implementation behind it)."""
import ctypes as C
import os

import numpy as np

from .build import LIB, build

ESCAPE = 1 << 63
NO_GENE = 0xFFFFFFFF

MERGE_NONE, MERGE_REAL_BARCODES, MERGE_SIMPLE, MERGE_POISSON_REAL, MERGE_POISSON_SIMPLE, MERGE_ALL = 0, 1, 2, 3, 4, 5
BARCODES_INDROP, BARCODES_CONST = 0, 1
UMI_MERGE_SIMPLE, UMI_MERGE_DIRECTIONAL = 0, 1

_STATUS = {1: "INVALID", 2: "RANGE", 3: "DEVICE", 4: "UNSUPPORTED", 5: "IO"}


class MatrixBytes(C.Structure):
    """dropest_matrix_bytes (include/dropest_amd.h)."""
    _fields_ = [("ncols", C.c_uint64), ("nnz", C.c_uint64), ("colptr", C.c_void_p), ("row_delta", C.c_void_p), ("value", C.c_void_p),
                ("n_row_listed", C.c_uint64), ("row_listed_pos", C.c_void_p), ("row_listed_row", C.c_void_p),
                ("n_value_listed", C.c_uint64), ("value_listed_pos", C.c_void_p), ("value_listed_value", C.c_void_p)]


class DropestError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("dropest_amd [%s]: %s" % (_STATUS.get(status, status), msg))
        self.status = status


class Cfg(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("merge_kind", C.c_int32), ("barcodes_kind", C.c_int32),
        ("barcodes_file", C.c_char_p), ("min_genes_before_merge", C.c_int32), ("min_genes_after_merge", C.c_int32),
        ("min_merge_fraction", C.c_double), ("max_cb_merge_edit_distance", C.c_int32),
        ("umi_merge_kind", C.c_int32), ("max_umi_merge_edit_distance", C.c_int32),
        ("gene_match_levels", C.c_char_p), ("max_cells", C.c_int32), ("cb_table_capacity", C.c_uint64),
        ("umi_merge_multiplier", C.c_double), ("max_merge_prob", C.c_double), ("max_real_merge_prob", C.c_double),
    ]


class CellRow(C.Structure):
    _fields_ = [
        ("barcode", C.c_uint64), ("first_read", C.c_uint32), ("n_genes", C.c_uint32),
        ("requested_genes", C.c_uint32), ("requested_umis", C.c_uint32), ("total_reads", C.c_int32),
        ("total_umis", C.c_int32), ("is_merged", C.c_uint8), ("is_excluded", C.c_uint8), ("is_real", C.c_uint8),
        ("pad", C.c_uint8),
    ]


CELL_ROW_DTYPE = np.dtype([
    ("barcode", "<u8"), ("first_read", "<u4"), ("n_genes", "<u4"), ("requested_genes", "<u4"),
    ("requested_umis", "<u4"), ("total_reads", "<i4"), ("total_umis", "<i4"), ("is_merged", "u1"),
    ("is_excluded", "u1"), ("is_real", "u1"), ("pad", "u1")], align=True)
assert CELL_ROW_DTYPE.itemsize == C.sizeof(CellRow)


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_uint32), ("ms", C.c_double), ("bytes", C.c_double)]


class IngestSummary(C.Structure):
    _fields_ = [("umi_clean_min", C.c_uint64), ("umi_clean_max", C.c_uint64), ("umi_escape_max_plus1", C.c_uint64),
                ("gene_max_plus1", C.c_uint32), ("chr_max_plus1", C.c_uint32), ("gene_chr_conflict", C.c_uint32),
                ("reserved", C.c_uint32)]


class SynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("stream_id", C.c_uint32), ("n_cells", C.c_uint32), ("cell_cb", C.c_void_p),
        ("cell_cdf", C.c_void_p), ("n_genes", C.c_uint32), ("gene_cdf", C.c_void_p), ("cb_len", C.c_uint32),
        ("umi_len", C.c_uint32), ("n_chr", C.c_uint32), ("permille_neighbour", C.c_uint32),
        ("permille_ambient", C.c_uint32), ("permille_intergenic", C.c_uint32), ("permille_intron", C.c_uint32),
        ("permille_exon_na", C.c_uint32), ("n_effective", C.c_uint64), ("reads_per_molecule", C.c_uint32),
    ]


_lib = None


def lib():
    """Loads (building first if the sources are newer) dropest_amd/lib/libdropest_amd.so."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    L = C.CDLL(LIB)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    P = C.POINTER
    sig = {
        "dropest_last_error": (C.c_char_p, []),
        "dropest_cfg_defaults": (None, [P(Cfg)]),
        "dropest_ctx_create": (C.c_int, [P(Cfg), P(vp)]),
        "dropest_ctx_destroy": (None, [vp]),
        "dropest_set_side_strings": (C.c_int, [vp, P(C.c_char_p), C.c_uint64]),
        "dropest_push_reads": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64]),
        "dropest_push_reads_device": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, C.c_int]),
        "dropest_set_initialized": (C.c_int, [vp]),
        "dropest_merge_and_filter": (C.c_int, [vp]),
        "dropest_reset_results": (C.c_int, [vp]),
        "dropest_total_cells": (C.c_int, [vp, u64p]),
        "dropest_real_cells": (C.c_int, [vp, u64p]),
        "dropest_cell_rows": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp]),
        "dropest_cell_id_by_cb": (C.c_int, [vp, C.c_uint64, P(C.c_int64)]),
        "dropest_filtered_cells": (C.c_int, [vp, u64p, vp]),
        "dropest_merge_targets": (C.c_int, [vp, u64p, vp, vp]),
        "dropest_global_counters": (C.c_int, [vp, vp]),
        "dropest_cell_molecules": (C.c_int, [vp, C.c_uint64, u64p, vp, vp, vp, vp]),
        "dropest_molecules": (C.c_int, [vp, u64p, vp, vp, vp, vp, vp]),
        "dropest_count_matrix": (C.c_int, [vp, C.c_int, C.c_int, u64p, vp, vp, vp]),
        "dropest_count_matrix_csc": (C.c_int, [vp, C.c_int, C.c_int, u64p, u64p, P(vp), P(vp), P(vp)]),
        "dropest_chr_stats": (C.c_int, [vp, u64p, vp, vp, vp, vp]),
        "dropest_count_matrix_csc_levels": (C.c_int, [vp, C.c_char_p, C.c_int, u64p, u64p, P(vp), P(vp), P(vp)]),
        "dropest_merge_target": (C.c_int, [vp, C.c_uint64, P(C.c_int64)]),
        "dropest_exclude_cell": (C.c_int, [vp, C.c_uint64]), "dropest_merge_cells": (C.c_int, [vp, C.c_uint64, C.c_uint64]),
        "dropest_merge_umis": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint64, vp, vp]),
        "dropest_add_umi_to_cell": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, vp, C.c_uint32]),
        "dropest_umi_first_seen": (C.c_int, [vp, u64p, vp]),
        "dropest_set_umi_qualities": (C.c_int, [vp, vp, C.c_uint32, C.c_uint64]),
        "dropest_set_umi_qualities_var": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64]),
        "dropest_cell_molecule_quality_lengths": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp]),
        "dropest_umi_quality_length": (C.c_int, [vp, P(C.c_uint32)]),
        "dropest_cell_molecule_qualities": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp]),
        "dropest_poisson_intersection_prob": (C.c_int, [vp, C.c_uint64, C.c_uint64, u64p, P(C.c_double), P(C.c_double)]),
        "dropest_umi_distribution": (C.c_int, [vp, u64p, vp, vp]),
        "dropest_collisions_adjusted_sizes": (C.c_int, [C.c_int, vp, C.c_uint64, C.c_uint64, vp]),
        "dropest_owner_of": (C.c_uint32, [C.c_uint64, C.c_uint32]),
        "dropest_partition_by_owner": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp, vp, vp, vp, vp,
                                                 C.c_uint64]),
        "dropest_partition_scratch_bytes": (C.c_int, [C.c_uint64, u64p]),
        "dropest_clear_reads": (C.c_int, [vp]),
        "dropest_resident_reads": (C.c_int, [vp, P(vp), P(vp), P(vp), P(vp), u64p]),
        "dropest_count_matrix_device": (C.c_int, [vp, C.c_int, C.c_int, u64p, u64p, P(vp), P(vp), P(vp)]),
        "dropest_cell_first_reads_device": (C.c_int, [vp, u64p, P(vp)]),
        "dropest_assemble_columns": (C.c_int, [C.c_int, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]),
        "dropest_assemble_columns_async": (C.c_int, [C.c_int, C.c_int, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]),
        "dropest_real_candidate_rows": (C.c_int, [vp, u64p, vp, vp]),
        "dropest_dev_copy_device": (C.c_int, [C.c_int, vp, vp, C.c_uint64]),
        "dropest_kernel_stats": (C.c_int, [vp, P(C.c_uint32), vp]),
        "dropest_set_profiling": (C.c_int, [vp, C.c_int]),
        "dropest_set_profiling_filter": (C.c_int, [vp, C.c_char_p]),
        "dropest_debug_poison_scratch": (C.c_int, [C.c_uint64, u64p]),
        "dropest_debug_trim_pool": (C.c_int, []),
        "dropest_debug_alloc_ordinal": (C.c_int, [u64p]),
        "dropest_debug_alloc_site": (C.c_int, [C.c_uint64, C.c_char_p, C.c_uint64]),
        "dropest_prefetch_raw_matrix": (C.c_int, [vp, C.c_int]),
        "dropest_prefetch_raw_matrix_narrow": (C.c_int, [vp, C.c_int]),
        "dropest_narrow_matrix_possible": (C.c_int, [vp, P(C.c_int)]),
        "dropest_prefetch_raw_matrix_bytes": (C.c_int, [vp, C.c_int]),
        "dropest_count_matrix_csc_bytes": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "dropest_matrix_bytes_widen": (C.c_int, [vp, vp, vp]),
        "dropest_matrix_rider_widen": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp, vp]),
        "dropest_set_raw_matrix_prefetch": (C.c_int, [vp, C.c_int, C.c_int]),
        "dropest_set_matrix_wire": (C.c_int, [vp, C.c_int]),
        "dropest_set_umi_dictionary": (C.c_int, [vp, C.c_int]),
        "dropest_debug_refresh": (C.c_int, []),
        "dropest_push_reads_gather": (C.c_int, [vp, C.c_uint64, vp, vp, vp, vp, vp]),
        "dropest_shard_set_umi_qualities": (C.c_int, [vp, vp, C.c_uint32, C.c_uint64]),
        "dropest_shard_set_umi_qualities_var": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64]),
        "dropest_count_matrix_csc_narrow": (C.c_int, [vp, C.c_int, C.c_int, u64p, u64p, P(vp), P(vp), P(vp), u64p, P(vp), P(vp)]),
        "dropest_radix_plan": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "dropest_sort_layout": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
        "dropest_stream": (vp, [vp]),
        "dropest_host_register": (C.c_int, [C.c_int, vp, C.c_uint64, P(vp)]),
        "dropest_host_unregister": (C.c_int, [C.c_int, vp]),
        "dropest_ingest": (C.c_int, [vp]),
        "dropest_ingest_summary_get": (C.c_int, [vp, P(IngestSummary)]),
        "dropest_ingest_summary_set": (C.c_int, [vp, P(IngestSummary)]),
        "dropest_gene_chr_table": (C.c_int, [vp, P(vp), u64p]),
        "dropest_shard_merge_search": (C.c_int, [vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, u64p]),
        "dropest_shard_merge_pairs": (C.c_int, [vp, vp, vp]),
        "dropest_shard_merge_export": (C.c_int, [vp, u64p, vp, vp, P(vp), vp]),
        "dropest_shard_merge_intersect": (C.c_int, [vp, C.c_uint64, vp, vp, vp, vp, vp]),
        "dropest_shard_merge_decide": (C.c_int, [vp, vp, vp]),
        "dropest_merge_apply": (C.c_int, [C.c_uint64, C.c_uint64, vp, vp, vp, vp, vp, vp]),
        "dropest_shard_merge_finish": (C.c_int, [vp, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp]),
        "dropest_synth_generate_host": (C.c_int, [P(SynthParams), C.c_uint64, C.c_uint64, vp, vp, vp, vp]),
        "dropest_synth_generate_device": (C.c_int, [P(SynthParams), C.c_int, C.c_uint64, C.c_uint64, vp, vp, vp, vp]),
        "dropest_dev_alloc": (C.c_int, [C.c_int, C.c_uint64, P(vp)]),
        "dropest_dev_free": (C.c_int, [C.c_int, vp]),
        "dropest_dev_copy_to_host": (C.c_int, [C.c_int, vp, vp, C.c_uint64]),
        "dropest_dev_copy_from_host": (C.c_int, [C.c_int, vp, vp, C.c_uint64]),
        "dropest_dev_count": (C.c_int, []),
        "dropest_dev_sync": (C.c_int, [C.c_int]),
        "dropest_rand_sequence": (C.c_int, [C.c_uint32, C.c_uint64, vp]),
        "dropest_table_sizes": (C.c_int, [vp, vp]),
        "dropest_shard_unique_id": (C.c_int, [vp]),
        "dropest_shard_create": (C.c_int, [P(Cfg), C.c_int32, C.c_int32, vp, P(vp)]),
        "dropest_shard_group_create": (C.c_int, [P(Cfg), C.c_int32, vp, vp]),
        "dropest_shard_destroy": (None, [vp]),
        "dropest_shard_ctx": (vp, [vp]),
        "dropest_shard_set_reads_device": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64]),
        "dropest_shard_step": (C.c_int, [vp]),
        "dropest_shard_push_reads": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64]),
        "dropest_reserve_reads": (C.c_int, [vp, C.c_uint64]),
        "dropest_shard_group_step": (C.c_int, [vp, C.c_int32]),
        "dropest_shard_matrix": (C.c_int, [vp, C.c_int, u64p, u64p, P(vp), P(vp), P(vp), P(vp)]),
        "dropest_shard_matrix_form": (C.c_int, [vp, C.c_int, P(C.c_int32)]),
        "dropest_shard_matrix_narrow": (C.c_int, [vp, C.c_int, u64p, u64p, P(vp), P(vp), P(vp), P(vp), u64p, P(vp), P(vp)]),
        "dropest_shard_matrix_bytes": (C.c_int, [vp, C.c_int, P(MatrixBytes), P(vp)]),
        "dropest_shard_merged_barcodes": (C.c_int, [vp, u64p, vp, vp]),
        "dropest_shard_phase_stats": (C.c_int, [vp, P(C.c_uint32), vp]),
        "dropest_shard_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
        "dropest_key_width": (C.c_int, [vp, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]),
        "dropest_ctx_split": (C.c_int, [vp, C.c_int32, vp]),
        "dropest_plan_columns": (C.c_int, [C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_int, C.c_uint32, C.c_int32, vp, C.c_uint64, u64p, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "dropest_last_error", "dropest_cfg_defaults", "dropest_ctx_create", "dropest_ctx_destroy",
    "dropest_set_side_strings", "dropest_push_reads", "dropest_push_reads_device", "dropest_set_initialized",
    "dropest_merge_and_filter", "dropest_reset_results", "dropest_total_cells", "dropest_real_cells",
    "dropest_cell_rows", "dropest_cell_id_by_cb", "dropest_filtered_cells", "dropest_merge_targets",
    "dropest_global_counters", "dropest_cell_molecules", "dropest_molecules", "dropest_count_matrix",
    "dropest_count_matrix_csc", "dropest_count_matrix_csc_levels", "dropest_owner_of", "dropest_partition_by_owner", "dropest_partition_scratch_bytes", "dropest_clear_reads",
    "dropest_count_matrix_device", "dropest_cell_first_reads_device", "dropest_assemble_columns", "dropest_assemble_columns_async",
    "dropest_real_candidate_rows", "dropest_dev_copy_device", "dropest_umi_distribution",
    "dropest_collisions_adjusted_sizes", "dropest_poisson_intersection_prob",
    "dropest_set_umi_qualities", "dropest_umi_quality_length", "dropest_cell_molecule_qualities",
    "dropest_set_umi_qualities_var", "dropest_cell_molecule_quality_lengths",
    "dropest_exclude_cell", "dropest_merge_cells", "dropest_merge_umis",
    "dropest_chr_stats", "dropest_merge_target", "dropest_kernel_stats", "dropest_set_profiling", "dropest_set_profiling_filter", "dropest_prefetch_raw_matrix", "dropest_radix_plan", "dropest_stream",
    "dropest_sort_layout", "dropest_host_register", "dropest_host_unregister", "dropest_ingest", "dropest_ingest_summary_get", "dropest_ingest_summary_set",
    "dropest_gene_chr_table", "dropest_shard_merge_search", "dropest_shard_merge_pairs", "dropest_shard_merge_export",
    "dropest_shard_merge_intersect", "dropest_shard_merge_decide", "dropest_merge_apply", "dropest_shard_merge_finish",
    "dropest_synth_generate_host", "dropest_synth_generate_device", "dropest_dev_alloc", "dropest_dev_free",
    "dropest_dev_copy_to_host", "dropest_dev_copy_from_host", "dropest_dev_count", "dropest_dev_sync",
    "dropest_rand_sequence", "dropest_table_sizes",
    "dropest_shard_unique_id", "dropest_shard_create", "dropest_shard_group_create", "dropest_shard_destroy", "dropest_shard_ctx",
    "dropest_shard_set_reads_device", "dropest_shard_push_reads", "dropest_reserve_reads", "dropest_shard_step", "dropest_shard_group_step", "dropest_shard_matrix", "dropest_shard_matrix_form",
    "dropest_shard_merged_barcodes", "dropest_shard_phase_stats", "dropest_shard_set_option", "dropest_plan_columns",
    "dropest_key_width", "dropest_ctx_split", "dropest_shard_matrix_narrow", "dropest_shard_matrix_bytes", "dropest_add_umi_to_cell", "dropest_umi_first_seen", "dropest_resident_reads", "dropest_prefetch_raw_matrix_narrow", "dropest_narrow_matrix_possible", "dropest_count_matrix_csc_narrow",
    "dropest_prefetch_raw_matrix_bytes", "dropest_count_matrix_csc_bytes", "dropest_matrix_bytes_widen", "dropest_matrix_rider_widen", "dropest_set_raw_matrix_prefetch", "dropest_set_matrix_wire", "dropest_set_umi_dictionary", "dropest_debug_refresh", "dropest_push_reads_gather", "dropest_shard_set_umi_qualities", "dropest_shard_set_umi_qualities_var",
    "dropest_debug_poison_scratch", "dropest_debug_trim_pool", "dropest_debug_alloc_ordinal", "dropest_debug_alloc_site",
]


# ---- 2-bit codes (include/dropest_amd.h) ----
_B2C = {"A": 0, "C": 1, "G": 2, "T": 3}


def pack_seq(s):
    """Returns the packed code of `s`, or None when it needs an escape (N, other letters, > 31 bases)."""
    if not s or len(s) > 31:
        return None
    c = 1
    for ch in s:
        b = _B2C.get(ch)
        if b is None:
            return None
        c = (c << 2) | b
    return c


def unpack_code(code, side=()):
    code = int(code)
    if code & ESCAPE:
        return side[code & (ESCAPE - 1)]
    n = (code.bit_length() - 1) // 2
    return "".join("ACGT"[(code >> (2 * (n - 1 - i))) & 3] for i in range(n))


def make_cfg(device=0, merge_kind=MERGE_NONE, barcodes_kind=BARCODES_INDROP, barcodes_file=None,
             min_genes_before_merge=10, min_genes_after_merge=10, min_merge_fraction=0.2,
             max_cb_merge_edit_distance=2, max_umi_merge_edit_distance=1, gene_match_levels="eEBA",
             max_cells=-1, cb_table_capacity=0, umi_merge_kind=UMI_MERGE_SIMPLE, umi_merge_multiplier=2.0,
             max_merge_prob=1e-4, max_real_merge_prob=1e-7):
    """dropest_cfg from keyword arguments; returns (cfg, the byte strings it points to -- keep them alive)."""
    cfg = Cfg()
    lib().dropest_cfg_defaults(C.byref(cfg))
    cfg.device = device; cfg.merge_kind = merge_kind; cfg.barcodes_kind = barcodes_kind
    bf = barcodes_file.encode() if barcodes_file else None
    ml = gene_match_levels.encode()
    cfg.barcodes_file = bf; cfg.gene_match_levels = ml
    cfg.min_genes_before_merge = min_genes_before_merge; cfg.min_genes_after_merge = min_genes_after_merge
    cfg.min_merge_fraction = min_merge_fraction; cfg.max_cb_merge_edit_distance = max_cb_merge_edit_distance
    cfg.max_umi_merge_edit_distance = max_umi_merge_edit_distance; cfg.max_cells = max_cells
    cfg.cb_table_capacity = cb_table_capacity
    cfg.umi_merge_kind = umi_merge_kind; cfg.umi_merge_multiplier = umi_merge_multiplier
    cfg.max_merge_prob = max_merge_prob; cfg.max_real_merge_prob = max_real_merge_prob
    return cfg, (bf, ml)


class Context:
    """One container on one GPU.  Method names follow Estimation::CellsDataContainer."""

    def __init__(self, device=0, _borrowed=None, **kw):
        self.L = lib()
        self.device = device
        self.side = []
        self._keep = []
        self.h = None
        self._owned = _borrowed is None
        if _borrowed is not None:          # the context of a shard (dropest_shard_ctx): accessors only, never destroyed here
            self.h = C.c_void_p(_borrowed)
            return
        cfg, self._cfg_keep = make_cfg(device=device, **kw)
        h = C.c_void_p()
        self._chk(self.L.dropest_ctx_create(C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if self.h and self._owned:
            self.L.dropest_ctx_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise DropestError(rc, self.L.dropest_last_error().decode())

    # ---- ingest ----
    def set_side_strings(self, strings):
        self.side = list(strings)
        arr = (C.c_char_p * max(1, len(self.side)))(*[s.encode() for s in self.side])
        self._chk(self.L.dropest_set_side_strings(self.h, arr, len(self.side)))

    def push_reads(self, cb, umi, gene, aux):
        cb = np.ascontiguousarray(cb, np.uint64); umi = np.ascontiguousarray(umi, np.uint64)
        gene = np.ascontiguousarray(gene, np.uint32); aux = np.ascontiguousarray(aux, np.uint32)
        assert len(cb) == len(umi) == len(gene) == len(aux)
        self._chk(self.L.dropest_push_reads(self.h, cb.ctypes.data, umi.ctypes.data, gene.ctypes.data,
                                            aux.ctypes.data, len(cb)))

    def push_reads_device(self, d_cb, d_umi, d_gene, d_aux, n, adopt=True):
        self._chk(self.L.dropest_push_reads_device(self.h, d_cb, d_umi, d_gene, d_aux, n, int(adopt)))

    def set_initialized(self):
        self._chk(self.L.dropest_set_initialized(self.h))

    def key_width(self):
        """(cell, gene, UMI) field widths of the last key-layout plan, also of one that did not fit 64 bits."""
        c, g, u = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.dropest_key_width(self.h, C.byref(c), C.byref(g), C.byref(u)))
        return c.value, g.value, u.value

    def merge_and_filter(self):
        self._chk(self.L.dropest_merge_and_filter(self.h))

    def reset_results(self):
        self._chk(self.L.dropest_reset_results(self.h))

    # ---- accessors ----
    def total_cells_number(self):
        n = C.c_uint64()
        self._chk(self.L.dropest_total_cells(self.h, C.byref(n)))
        return n.value

    def real_cells_number(self):
        n = C.c_uint64()
        self._chk(self.L.dropest_real_cells(self.h, C.byref(n)))
        return n.value

    def cell_rows(self, first=0, count=None):
        if count is None:
            count = self.total_cells_number() - first
        out = np.zeros(count, CELL_ROW_DTYPE)
        self._chk(self.L.dropest_cell_rows(self.h, first, count, out.ctypes.data))
        return out

    def cell_id_by_cb(self, barcode_code):
        i = C.c_int64()
        self._chk(self.L.dropest_cell_id_by_cb(self.h, barcode_code, C.byref(i)))
        return i.value

    def filtered_cells(self):
        n = C.c_uint64()
        self._chk(self.L.dropest_filtered_cells(self.h, C.byref(n), None))
        out = np.zeros(n.value, np.uint64)
        if n.value:
            self._chk(self.L.dropest_filtered_cells(self.h, C.byref(n), out.ctypes.data))
        return out

    def merge_target_pairs(self):
        """(source cell ids, target cell ids) of the cells the CB merge folded: the non-identity entries of merge_targets()."""
        n = C.c_uint64()
        self._chk(self.L.dropest_merge_targets(self.h, C.byref(n), None, None))
        src = np.zeros(n.value, np.uint64); tgt = np.zeros(n.value, np.uint64)
        if n.value:
            self._chk(self.L.dropest_merge_targets(self.h, C.byref(n), src.ctypes.data, tgt.ctypes.data))
        return src, tgt

    def merge_targets(self):
        """Full merge_targets() vector of the reference (identity where nothing was merged)."""
        src, tgt = self.merge_target_pairs()
        full = np.arange(self.total_cells_number(), dtype=np.uint64)
        full[src.astype(np.int64)] = tgt
        return full

    def global_counters(self):
        out = np.zeros(4, np.uint64)
        self._chk(self.L.dropest_global_counters(self.h, out.ctypes.data))
        return out

    def molecules(self):
        n = C.c_uint64()
        self._chk(self.L.dropest_molecules(self.h, C.byref(n), None, None, None, None, None))
        cell = np.zeros(n.value, np.uint32); gene = np.zeros(n.value, np.uint32); umi = np.zeros(n.value, np.uint64)
        reads = np.zeros(n.value, np.uint32); mark = np.zeros(n.value, np.uint8)
        if n.value:
            self._chk(self.L.dropest_molecules(self.h, C.byref(n), cell.ctypes.data, gene.ctypes.data, umi.ctypes.data,
                                               reads.ctypes.data, mark.ctypes.data))
        return cell, gene, umi, reads, mark

    def cell_molecules(self, cell):
        n = C.c_uint64()
        self._chk(self.L.dropest_cell_molecules(self.h, cell, C.byref(n), None, None, None, None))
        gene = np.zeros(n.value, np.uint32); umi = np.zeros(n.value, np.uint64)
        reads = np.zeros(n.value, np.uint32); mark = np.zeros(n.value, np.uint8)
        if n.value:
            self._chk(self.L.dropest_cell_molecules(self.h, cell, C.byref(n), gene.ctypes.data, umi.ctypes.data,
                                                    reads.ctypes.data, mark.ctypes.data))
        return gene, umi, reads, mark

    def count_matrix(self, filtered=True, reads_output=False):
        n = C.c_uint64()
        self._chk(self.L.dropest_count_matrix(self.h, int(filtered), int(reads_output), C.byref(n), None, None, None))
        g = np.zeros(n.value, np.uint32); c = np.zeros(n.value, np.uint32); v = np.zeros(n.value, np.uint32)
        if n.value:
            self._chk(self.L.dropest_count_matrix(self.h, int(filtered), int(reads_output), C.byref(n), g.ctypes.data,
                                                  c.ctypes.data, v.ctypes.data))
        return g, c, v

    def count_matrix_csc(self, filtered=True, reads_output=False):
        """(colptr, rowidx, values) as zero-copy numpy views of context-owned pinned memory (valid until the
        next call for the same matrix)."""
        ncols, nnz = C.c_uint64(), C.c_uint64()
        pc, pr, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.dropest_count_matrix_csc(self.h, int(filtered), int(reads_output), C.byref(ncols), C.byref(nnz),
                                                  C.byref(pc), C.byref(pr), C.byref(pv)))
        def view(ptr, n):
            if n == 0 or not ptr.value:
                return np.zeros(0, np.uint32)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n,))
        return view(pc, ncols.value + 1), view(pr, nnz.value), view(pv, nnz.value)

    def prefetch_raw_matrix(self, reads_output=False, narrow=False, form=None):
        """Start cm_raw (emit + copy to the host) on a second stream; count_matrix_csc(filtered=False) then only waits.
        form: 0 32-bit, 1 16-bit (= narrow), 2 bytes."""
        if form == 2:
            self._chk(self.L.dropest_prefetch_raw_matrix_bytes(self.h, int(reads_output)))
        elif narrow or form == 1:
            self._chk(self.L.dropest_prefetch_raw_matrix_narrow(self.h, int(reads_output)))
        else:
            self._chk(self.L.dropest_prefetch_raw_matrix(self.h, int(reads_output)))

    def push_reads_gather(self, segments):
        """segments: list of (cb, umi, gene, aux) host arrays that follow one another in the stream."""
        segs = [(np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64), np.ascontiguousarray(c, np.uint32), np.ascontiguousarray(d, np.uint32)) for a, b, c, d in segments]
        n = len(segs)
        P = C.c_void_p * max(1, n)
        cols = [P(*[s[k].ctypes.data for s in segs]) for k in range(4)]
        counts = (C.c_uint64 * max(1, n))(*[len(s[0]) for s in segs])
        self._chk(self.L.dropest_push_reads_gather(self.h, n, cols[0], cols[1], cols[2], cols[3], counts))

    def set_matrix_wire(self, on=True):
        """dropest_set_matrix_wire: large 32-bit matrices cross PCIe as bytes and are widened by host threads under the copy (default on)."""
        self._chk(self.L.dropest_set_matrix_wire(self.h, int(bool(on))))

    def set_umi_dictionary(self, mode=0):
        """dropest_set_umi_dictionary: 0 = UMI ranks in the key only when gene + UMI fields reach 64 bits, 1 = also before a key
        wider than 64 bits is refused, 2 = always."""
        self._chk(self.L.dropest_set_umi_dictionary(self.h, int(mode)))

    def set_raw_matrix_prefetch(self, form=2, reads_output=False):
        """Announce the form cm_raw will be asked for (0 / 1 / 2; -1: off): the container starts its prefetch by itself."""
        self._chk(self.L.dropest_set_raw_matrix_prefetch(self.h, int(form), int(reads_output)))

    def narrow_matrix_possible(self):
        v = C.c_int()
        self._chk(self.L.dropest_narrow_matrix_possible(self.h, C.byref(v)))
        return bool(v.value)

    def count_matrix_csc_narrow(self, filtered=True, reads_output=False):
        """(colptr u32, rowidx u16, values u16, overflow_pos u32, overflow_val u32): zero-copy views of context-owned pinned
        memory; values[overflow_pos[k]] == 0xFFFF stands for overflow_val[k]."""
        ncols, nnz, novf = C.c_uint64(), C.c_uint64(), C.c_uint64()
        pc, pr, pv, po, pw = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.dropest_count_matrix_csc_narrow(self.h, int(filtered), int(reads_output), C.byref(ncols), C.byref(nnz),
                                                         C.byref(pc), C.byref(pr), C.byref(pv), C.byref(novf), C.byref(po), C.byref(pw)))
        def view(ptr, n, ct, dt):
            if n == 0 or not ptr.value:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))
        return (view(pc, ncols.value + 1, C.c_uint32, np.uint32), view(pr, nnz.value, C.c_uint16, np.uint16), view(pv, nnz.value, C.c_uint16, np.uint16),
                view(po, novf.value, C.c_uint32, np.uint32), view(pw, novf.value, C.c_uint32, np.uint32))

    def count_matrix_csc_bytes(self, filtered=True, reads_output=False):
        """The byte form (include/dropest_amd.h: dropest_matrix_bytes) as a MatrixBytes structure; its arrays are context-owned pinned
        memory.  widen_bytes() decodes it into (colptr, rowidx, values) of 32 bits on the library's host threads."""
        m = MatrixBytes()
        self._chk(self.L.dropest_count_matrix_csc_bytes(self.h, int(filtered), int(reads_output), C.byref(m)))
        return m

    def widen_bytes(self, m, out=None):
        colptr = np.ctypeslib.as_array(C.cast(m.colptr, C.POINTER(C.c_uint32)), shape=(m.ncols + 1,)) if m.ncols else np.zeros(1, np.uint32)
        rows, vals = out if out is not None else (np.zeros(m.nnz, np.uint32), np.zeros(m.nnz, np.uint32))
        assert len(rows) >= m.nnz and len(vals) >= m.nnz
        self._chk(self.L.dropest_matrix_bytes_widen(C.byref(m), rows.ctypes.data, vals.ctypes.data))
        return colptr, rows, vals

    @staticmethod
    def widen(narrow):
        """The 32-bit (colptr, rowidx, values) of a narrow matrix."""
        colptr, r16, v16, opos, oval = narrow
        vals = v16.astype(np.uint32)
        vals[opos] = oval
        return colptr, r16.astype(np.uint32), vals

    def count_matrix_levels(self, levels, reads_output=False):
        """Filtered count matrix under another mark query, as triplets (gene, column, value) in column-major order."""
        ncols, nnz = C.c_uint64(), C.c_uint64()
        pc, pr, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.dropest_count_matrix_csc_levels(self.h, levels.encode(), int(reads_output), C.byref(ncols), C.byref(nnz),
                                                         C.byref(pc), C.byref(pr), C.byref(pv)))
        n = nnz.value
        if n == 0:
            z = np.zeros(0, np.uint32)
            return z, z, z
        colptr = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint32)), shape=(ncols.value + 1,))
        rows = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_uint32)), shape=(n,)).copy()
        vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_uint32)), shape=(n,)).copy()
        cols = np.repeat(np.arange(ncols.value, dtype=np.uint32), np.diff(colptr.astype(np.int64)))
        return rows, cols, vals

    def count_matrix_device(self, filtered=True, reads_output=False):
        """(colptr numpy copy, d_rowidx, d_values, nnz): row indices / values stay in HBM (raw device pointers)."""
        ncols, nnz = C.c_uint64(), C.c_uint64()
        pc, pr, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.dropest_count_matrix_device(self.h, int(filtered), int(reads_output), C.byref(ncols), C.byref(nnz),
                                                     C.byref(pc), C.byref(pr), C.byref(pv)))
        colptr = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint32)), shape=(ncols.value + 1,)).copy()
        return colptr, pr.value, pv.value, nnz.value

    def cell_first_reads_device(self):
        n, p = C.c_uint64(), C.c_void_p()
        self._chk(self.L.dropest_cell_first_reads_device(self.h, C.byref(n), C.byref(p)))
        return n.value, p.value

    def real_candidate_rows(self):
        n = C.c_uint64()
        self._chk(self.L.dropest_real_candidate_rows(self.h, C.byref(n), None, None))
        ids = np.zeros(n.value, np.uint64); rows = np.zeros(n.value, CELL_ROW_DTYPE)
        if n.value:
            self._chk(self.L.dropest_real_candidate_rows(self.h, C.byref(n), ids.ctypes.data, rows.ctypes.data))
        return ids, rows

    # ---- sharded runs: phases with the caller's collectives between them (include/dropest_amd.h) ----
    def ingest(self):
        self._chk(self.L.dropest_ingest(self.h))

    def ingest_summary(self):
        s = IngestSummary()
        self._chk(self.L.dropest_ingest_summary_get(self.h, C.byref(s)))
        return s

    def set_ingest_summary(self, s):
        self._chk(self.L.dropest_ingest_summary_set(self.h, C.byref(s)))

    def gene_chr_table(self):
        p = C.c_void_p(); n = C.c_uint64()
        self._chk(self.L.dropest_gene_chr_table(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def shard_merge_search(self, g_barcode, g_n_genes, g_total_umis, base_global, base_local):
        gb = np.ascontiguousarray(g_barcode, np.uint64); gg = np.ascontiguousarray(g_n_genes, np.uint32)
        gu = np.ascontiguousarray(g_total_umis, np.int32)
        bg = np.ascontiguousarray(base_global, np.uint32); bl = np.ascontiguousarray(base_local, np.uint32)
        n = C.c_uint64()
        self._chk(self.L.dropest_shard_merge_search(self.h, len(gb), gb.ctypes.data, gg.ctypes.data, gu.ctypes.data, len(bg),
                                                    bg.ctypes.data, bl.ctypes.data, C.byref(n)))
        pb = np.zeros(n.value, np.uint32); pc = np.zeros(n.value, np.uint32)
        if n.value:
            self._chk(self.L.dropest_shard_merge_pairs(self.h, pb.ctypes.data, pc.ctypes.data))
        return pb, pc

    def shard_merge_export(self):
        """-> listed_global, row_offset, device pointer of the key fields, 4 device pointers {reads, mark, exon, intron}"""
        n = C.c_uint64()
        self._chk(self.L.dropest_shard_merge_export(self.h, C.byref(n), None, None, None, None))
        listed = np.zeros(n.value, np.uint32); off = np.zeros(n.value + 1, np.uint64)
        low = C.c_void_p(); cols = (C.c_void_p * 4)()
        self._chk(self.L.dropest_shard_merge_export(self.h, C.byref(n), listed.ctypes.data, off.ctypes.data, C.byref(low), cols))
        return listed, off, low.value, [cols[k] for k in range(4)]

    def shard_merge_intersect(self, cand_local, base_begin, base_end, d_base_low):
        c = np.ascontiguousarray(cand_local, np.uint32); b = np.ascontiguousarray(base_begin, np.uint64)
        e = np.ascontiguousarray(base_end, np.uint64)
        out = np.zeros(len(c), np.uint32)
        self._chk(self.L.dropest_shard_merge_intersect(self.h, len(c), c.ctypes.data, b.ctypes.data, e.ctypes.data, d_base_low,
                                                       out.ctypes.data))
        return out

    def shard_merge_decide(self, inter, n_bases):
        i = np.ascontiguousarray(inter, np.uint32)
        out = np.full(n_bases, -1, np.int64)
        self._chk(self.L.dropest_shard_merge_decide(self.h, i.ctypes.data, out.ctypes.data))
        return out

    def shard_merge_finish(self, local_id, excluded, merged_away, total_reads, total_umis, move_src, move_tgt, n_import,
                           d_cell, d_low, d_cols):
        li = np.ascontiguousarray(local_id, np.uint32); ex = np.ascontiguousarray(excluded, np.uint8)
        mg = np.ascontiguousarray(merged_away, np.uint8); tr = np.ascontiguousarray(total_reads, np.int32)
        tu = np.ascontiguousarray(total_umis, np.int32)
        ms = np.ascontiguousarray(move_src, np.uint32); mt = np.ascontiguousarray(move_tgt, np.uint32)
        cols = (C.c_void_p * 4)(*[C.c_void_p(x) for x in d_cols])
        self._chk(self.L.dropest_shard_merge_finish(self.h, len(li), li.ctypes.data, ex.ctypes.data, mg.ctypes.data, tr.ctypes.data,
                                                    tu.ctypes.data, len(ms), ms.ctypes.data, mt.ctypes.data, int(n_import),
                                                    d_cell, d_low, cols))

    def clear_reads(self):
        self._chk(self.L.dropest_clear_reads(self.h))

    def chr_stats(self):
        n = C.c_uint64()
        self._chk(self.L.dropest_chr_stats(self.h, C.byref(n), None, None, None, None))
        cell = np.zeros(n.value, np.uint32); kind = np.zeros(n.value, np.uint32); chr_ = np.zeros(n.value, np.uint32)
        cnt = np.zeros(n.value, np.int32)
        if n.value:
            self._chk(self.L.dropest_chr_stats(self.h, C.byref(n), cell.ctypes.data, kind.ctypes.data, chr_.ctypes.data,
                                               cnt.ctypes.data))
        return cell, kind, chr_, cnt

    def umi_distribution(self):
        n = C.c_uint64()
        self._chk(self.L.dropest_umi_distribution(self.h, C.byref(n), None, None))
        umi = np.zeros(n.value, np.uint64); cnt = np.zeros(n.value, np.uint64)
        if n.value:
            self._chk(self.L.dropest_umi_distribution(self.h, C.byref(n), umi.ctypes.data, cnt.ctypes.data))
        return umi, cnt

    def merge_target(self, cell):
        t = C.c_int64()
        self._chk(self.L.dropest_merge_target(self.h, cell, C.byref(t)))
        return t.value

    def exclude_cell(self, cell):
        self._chk(self.L.dropest_exclude_cell(self.h, cell))

    def merge_cells(self, src, tgt):
        self._chk(self.L.dropest_merge_cells(self.h, src, tgt))

    def merge_umis(self, cell, gene, pairs):
        """pairs: [(source code, target code)] applied in order (Cell::merge_umis walks the caller's map)."""
        s = np.array([p[0] for p in pairs], np.uint64); t = np.array([p[1] for p in pairs], np.uint64)
        self._chk(self.L.dropest_merge_umis(self.h, cell, gene, len(pairs), s.ctypes.data, t.ctypes.data))

    def add_umi_to_cell(self, cell, gene, umi_code, mark, quality=b""):
        q = np.frombuffer(bytes(quality), np.uint8)
        self._chk(self.L.dropest_add_umi_to_cell(self.h, int(cell), int(gene), int(umi_code), int(mark), q.ctypes.data if len(q) else None, len(q)))

    def umi_first_seen(self):
        """UMI codes in umi_indexer() order."""
        n = C.c_uint64()
        self._chk(self.L.dropest_umi_first_seen(self.h, C.byref(n), None))
        out = np.zeros(n.value, np.uint64)
        if n.value:
            self._chk(self.L.dropest_umi_first_seen(self.h, C.byref(n), out.ctypes.data))
        return out

    def set_umi_qualities(self, qual, lengths=None):
        """qual: uint8 array [n_reads, quality_length] (phred+33 characters), read order; lengths: per read, when the strings differ in length."""
        qual = np.ascontiguousarray(qual, np.uint8)
        assert qual.ndim == 2
        if lengths is None:
            self._chk(self.L.dropest_set_umi_qualities(self.h, qual.ctypes.data, qual.shape[1], qual.shape[0]))
        else:
            lengths = np.ascontiguousarray(lengths, np.uint8)
            assert lengths.shape == (qual.shape[0],)
            self._chk(self.L.dropest_set_umi_qualities_var(self.h, qual.ctypes.data, qual.shape[1], lengths.ctypes.data, qual.shape[0]))

    def umi_quality_length(self):
        q = C.c_uint32()
        self._chk(self.L.dropest_umi_quality_length(self.h, C.byref(q)))
        return q.value

    def cell_molecule_quality_lengths(self, cell, n):
        out = np.zeros(n, np.uint32)
        self._chk(self.L.dropest_cell_molecule_quality_lengths(self.h, cell, n, out.ctypes.data))
        return out

    def cell_molecule_qualities(self, cell, n):
        """Quality sums [n, quality_length] of the cell's molecules, in the order of cell_molecules()."""
        out = np.zeros((n, self.umi_quality_length()), np.uint32)
        self._chk(self.L.dropest_cell_molecule_qualities(self.h, cell, n, out.ctypes.data))
        return out

    def poisson_intersection_prob(self, cell1, cell2):
        """(intersection size, expected intersection size, merge probability) -- PoissonTargetEstimator::estimate_intersection_prob."""
        n, e, p = C.c_uint64(), C.c_double(), C.c_double()
        self._chk(self.L.dropest_poisson_intersection_prob(self.h, cell1, cell2, C.byref(n), C.byref(e), C.byref(p)))
        return n.value, e.value, p.value

    def sort_layout(self):
        out = (C.c_uint32 * 7)()
        self._chk(self.L.dropest_sort_layout(self.h, out))
        d = dict(zip(("cell_bits", "gene_bits", "umi_bits", "mark_bits_in_key", "value_bytes", "passes", "sort"), map(int, out)))
        d["sort"] = ("lsd", "splitter")[d["sort"]]
        return d

    def table_sizes(self):
        out = (C.c_uint64 * 4)()
        self._chk(self.L.dropest_table_sizes(self.h, out))
        return dict(zip(("reads", "cells", "molecules", "cell_gene_rows"), map(int, out)))

    def set_profiling(self, on=True, only=None):
        """HIP events around the launches; only="rs_scatter": just the launches whose stat name starts with that."""
        self._chk(self.L.dropest_set_profiling_filter(self.h, (only or "").encode()))
        self._chk(self.L.dropest_set_profiling(self.h, int(on)))

    def kernel_stats(self):
        n = C.c_uint32()
        self._chk(self.L.dropest_kernel_stats(self.h, C.byref(n), None))
        arr = (KernelStat * max(1, n.value))()
        self._chk(self.L.dropest_kernel_stats(self.h, C.byref(n), arr))
        return {arr[i].name.decode(): dict(launches=arr[i].launches, ms=arr[i].ms, bytes=arr[i].bytes)
                for i in range(n.value)}


def merge_apply(order, target, total_reads, total_umis):
    """MergeStrategyBase::merge_inited second loop over flat arrays (no context): returns (final_target, excluded,
    total_reads, total_umis) after applying step i = (order[i] -> target[i] or -1)."""
    o = np.ascontiguousarray(order, np.uint32); t = np.ascontiguousarray(target, np.int64)
    r = np.array(total_reads, np.int32); u = np.array(total_umis, np.int32)
    n = len(r)
    final = np.zeros(n, np.uint32); excl = np.zeros(n, np.uint8)
    L = lib()
    rc = L.dropest_merge_apply(n, len(o), o.ctypes.data, t.ctypes.data, r.ctypes.data, u.ctypes.data, final.ctypes.data,
                               excl.ctypes.data)
    if rc != 0:
        raise DropestError(rc, L.dropest_last_error().decode())
    return final, excl, r, u


def plan_columns(barcode, first_global, n_genes, req_genes, req_umis, total_umis, filtered, min_after, max_cells=-1, side=()):
    """Row indices of the global table in the column order of cm (filtered) / cm_raw (host logic, no GPU)."""
    cols = [np.ascontiguousarray(a, dt) for a, dt in ((barcode, np.uint64), (first_global, np.uint64), (n_genes, np.uint32),
                                                       (req_genes, np.uint32), (req_umis, np.uint32), (total_umis, np.int32))]
    n = len(cols[0])
    order = np.zeros(max(n, 1), np.uint32)
    k = C.c_uint64()
    arr = (C.c_char_p * max(1, len(side)))(*[x.encode() for x in side])
    rc = lib().dropest_plan_columns(n, *[c.ctypes.data for c in cols], int(filtered), int(min_after), int(max_cells), arr, len(side),
                                    C.byref(k), order.ctypes.data)
    if rc != 0:
        raise DropestError(rc, lib().dropest_last_error().decode())
    return order[:k.value].copy()


def radix_plan(varying_mask):
    """[(shift, bits), ...] of the radix sort for keys whose varying bits are `varying_mask` (host logic, no GPU)."""
    n = C.c_uint32()
    sh = (C.c_int32 * 8)(); bt = (C.c_int32 * 8)()
    rc = lib().dropest_radix_plan(C.c_uint64(varying_mask), C.byref(n), sh, bt)
    if rc != 0:
        raise DropestError(rc, lib().dropest_last_error().decode())
    return [(int(sh[i]), int(bt[i])) for i in range(n.value)]


def collisions_adjusted_sizes(probs, max_expression, device=0):
    """Tools::CollisionsAdjuster table on the device."""
    p = np.ascontiguousarray(probs, np.float64)
    out = np.zeros(max_expression, np.uint64)
    rc = lib().dropest_collisions_adjusted_sizes(device, p.ctypes.data, len(p), max_expression, out.ctypes.data)
    if rc != 0:
        raise DropestError(rc, lib().dropest_last_error().decode())
    return out


class DeviceArrays:
    """Four device arrays (cb, umi, gene, aux) of n reads, allocated through the library."""

    def __init__(self, device, n):
        self.L = lib()
        self.device, self.n = device, n
        self.ptrs = []
        for width in (8, 8, 4, 4):
            p = C.c_void_p()
            if self.L.dropest_dev_alloc(device, n * width, C.byref(p)) != 0:
                self.free()
                raise DropestError(3, "device allocation of %d bytes failed" % (n * width))
            self.ptrs.append(p)

    def free(self):
        for p in self.ptrs:
            self.L.dropest_dev_free(self.device, p)
        self.ptrs = []

    @classmethod
    def from_host(cls, device, cb, umi, gene, aux):
        """Uploads four host arrays (used by tests to give every shard its range of an arbitrary stream)."""
        self = cls(device, len(cb))
        for a, p, dt in zip((cb, umi, gene, aux), self.ptrs, (np.uint64, np.uint64, np.uint32, np.uint32)):
            a = np.ascontiguousarray(a, dt)
            if len(a) and self.L.dropest_dev_copy_from_host(device, p, a.ctypes.data, a.nbytes) != 0:
                raise DropestError(3, "host-to-device copy failed")
        return self

    def to_host(self):
        out = [np.zeros(self.n, np.uint64), np.zeros(self.n, np.uint64), np.zeros(self.n, np.uint32),
               np.zeros(self.n, np.uint32)]
        for a, p, w in zip(out, self.ptrs, (8, 8, 4, 4)):
            if self.L.dropest_dev_copy_to_host(self.device, a.ctypes.data, p, self.n * w) != 0:
                raise DropestError(3, "device-to-host copy failed")
        return out
