"""ctypes binding of oracle/liboracle.so (the CPU restatement in dropest_oracle.cpp).

TEST INFRASTRUCTURE ONLY -- see dropest_oracle.cpp's header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build_oracle(force=False):
    """Compile liboracle.so with g++ (a few seconds).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("dropest_oracle.cpp", "gene_annotation_oracle.cpp", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


class _Cfg(C.Structure):
    _fields_ = [
        ("merge_kind", C.c_int), ("barcodes_kind", C.c_int), ("barcodes_file", C.c_char_p),
        ("min_genes_before", C.c_int), ("min_genes_after", C.c_int), ("min_merge_fraction", C.c_double),
        ("max_cb_merge_ed", C.c_int), ("umi_merge_kind", C.c_int), ("max_umi_merge_ed", C.c_int),
        ("umi_mult", C.c_double), ("match_levels", C.c_char_p), ("max_cells", C.c_int),
        ("max_merge_prob", C.c_double), ("max_real_merge_prob", C.c_double),
    ]


def oracle_lib():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    L = C.CDLL(_LIB_PATH)
    vp, u64, i64 = C.c_void_p, C.c_uint64, C.c_int64
    P = C.POINTER
    sig = {
        "orc_last_error": (C.c_char_p, []),
        "orc_create": (vp, [P(_Cfg)]),
        "orc_destroy": (None, [vp]),
        "orc_add_record": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
        "orc_add_packed": (C.c_int, [vp, vp, vp, vp, vp, u64, P(C.c_char_p)]),
        "orc_set_initialized": (C.c_int, [vp]),
        "orc_merge_and_filter": (C.c_int, [vp]),
        "orc_merge_umis_only": (C.c_int, [vp]),
        "orc_n_cells": (u64, [vp]), "orc_n_genes": (u64, [vp]), "orc_n_filtered": (u64, [vp]),
        "orc_n_real": (u64, [vp]), "orc_n_merge_targets": (u64, [vp]), "orc_n_chr": (u64, [vp]),
        "orc_filtered": (None, [vp, vp]), "orc_merge_targets": (None, [vp, vp]),
        "orc_cell_id_by_cb": (C.c_long, [vp, C.c_char_p]),
        "orc_cell_barcode": (C.c_char_p, [vp, u64]), "orc_gene_name": (C.c_char_p, [vp, u64]),
        "orc_chr_name": (C.c_char_p, [vp, u64]),
        "orc_cell_rows": (None, [vp, vp]), "orc_global_counters": (None, [vp, vp]),
        "orc_molecules": (u64, [vp, vp, vp, vp, C.c_int, vp, vp]),
        "orc_count_matrix": (u64, [vp, C.c_int, C.c_int, vp, vp, vp]),
        "orc_chr_stats": (u64, [vp, vp, vp, vp, vp]),
        "orc_umi_distribution": (u64, [vp, C.c_char_p, C.c_int, vp]),
        "orc_edit_distance": (C.c_uint, [C.c_char_p, C.c_char_p, C.c_int, C.c_uint]),
        "orc_hamming_distance": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
        "orc_parse_encoded_id": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
        "orc_reverse_complement": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
        "orc_merge_target": (C.c_long, [vp, u64]),
        "orc_real_neighbours": (u64, [vp, u64, vp, u64]),
        "orc_umig_intersection": (u64, [vp, u64, u64]),
        "orc_wl_parts": (u64, [vp]), "orc_wl_part_size": (u64, [vp, u64]),
        "orc_wl_barcode": (C.c_char_p, [vp, u64, u64]),
        "orc_wl_distances": (u64, [vp, C.c_char_p, u64, vp, vp]),
        "orc_wl_load": (C.c_int, [vp, C.c_int, C.c_char_p]),
        "orc_wl_set": (C.c_int, [vp, C.c_int, P(C.c_char_p), C.c_int, P(C.c_char_p), C.c_int]),
        "orc_wl_split": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_int]),
        "orc_merge_umis_explicit": (C.c_int, [vp, u64, C.c_char_p, P(C.c_char_p), P(C.c_char_p), C.c_int]),
        "orc_fill_wrong_umi": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
        "orc_directional_targets": (C.c_int, [vp, P(C.c_char_p), vp, C.c_int, C.c_char_p, C.c_char_p, C.c_int]),
        "orc_collisions_table": (C.c_int, [vp, u64, u64, vp]),
        "orc_merge_cells_explicit": (C.c_int, [vp, u64, u64]), "orc_exclude_cell_explicit": (C.c_int, [vp, u64]),
        "orc_count_matrix_levels": (u64, [vp, C.c_char_p, C.c_int, vp, vp, vp]),
        "orc_add_packed_q": (C.c_int, [vp, vp, vp, vp, vp, u64, C.POINTER(C.c_char_p), vp, C.c_uint32]),
        "orc_add_packed_qvar": (C.c_int, [vp, vp, vp, vp, vp, u64, C.POINTER(C.c_char_p), vp, C.c_uint32, vp]),
        "orc_molecule_qualities_var": (C.c_int, [vp, C.c_uint32, vp, vp]),
        "orc_molecule_qualities": (C.c_int, [vp, C.c_uint32, vp]),
        "orc_poisson_init": (C.c_int, [vp]), "orc_poisson_distribution_size": (u64, [vp]),
        "orc_poisson_gene_intersection": (C.c_double, [vp, u64, u64]),
        "orc_poisson_intersection_prob": (C.c_double, [vp, u64, u64]),
        "orc_poisson_expected_intersection": (C.c_double, [vp, u64, u64]),
        "orc_poisson_merge_target": (C.c_long, [vp, u64]),
        "orc_poisson_upper_tail": (C.c_double, [C.c_long, C.c_double]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


class OracleConfig:
    def __init__(self, merge_kind=0, barcodes_kind=0, barcodes_file="", min_genes_before=10, min_genes_after=10,
                 min_merge_fraction=0.2, max_cb_merge_ed=0, umi_merge_kind=0, max_umi_merge_ed=1, umi_mult=2.0,
                 match_levels="eEBA", max_cells=-1, max_merge_prob=1e-4, max_real_merge_prob=1e-7):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def _strs(buf, n, stride):
    return [buf[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode() for i in range(n)]


class Oracle:
    """Thin object wrapper: mirrors CellsDataContainer's call sequence."""

    def __init__(self, cfg=None, **kw):
        cfg = cfg or OracleConfig(**kw)
        self.cfg = cfg
        self.L = oracle_lib()
        c = _Cfg(cfg.merge_kind, cfg.barcodes_kind, cfg.barcodes_file.encode(), cfg.min_genes_before,
                 cfg.min_genes_after, cfg.min_merge_fraction, cfg.max_cb_merge_ed, cfg.umi_merge_kind,
                 cfg.max_umi_merge_ed, cfg.umi_mult, cfg.match_levels.encode(), cfg.max_cells, cfg.max_merge_prob,
                 cfg.max_real_merge_prob)
        self.h = self.L.orc_create(C.byref(c))
        if not self.h:
            raise RuntimeError(self.L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.orc_last_error().decode())

    # ---- ingest ----
    def add_record(self, cb, umi, gene, chr_name="", mark=2, umi_qual=""):
        self._chk(self.L.orc_add_record(self.h, cb.encode(), umi.encode(), umi_qual.encode(), gene.encode(),
                                        chr_name.encode(), mark))

    def add_packed(self, cb, umi, gene, aux, side=()):
        cb = np.ascontiguousarray(cb, np.uint64); umi = np.ascontiguousarray(umi, np.uint64)
        gene = np.ascontiguousarray(gene, np.uint32); aux = np.ascontiguousarray(aux, np.uint32)
        arr = (C.c_char_p * max(1, len(side)))(*[s.encode() for s in side])
        self._chk(self.L.orc_add_packed(self.h, cb.ctypes.data, umi.ctypes.data, gene.ctypes.data,
                                        aux.ctypes.data, len(cb), arr))

    def add_packed_q(self, cb, umi, gene, aux, qual, side=()):
        """Like add_packed, with UMI qualities: uint8 [n, quality_length]."""
        cb = np.ascontiguousarray(cb, np.uint64); umi = np.ascontiguousarray(umi, np.uint64)
        gene = np.ascontiguousarray(gene, np.uint32); aux = np.ascontiguousarray(aux, np.uint32)
        qual = np.ascontiguousarray(qual, np.uint8)
        arr = (C.c_char_p * max(1, len(side)))(*[s.encode() for s in side])
        self._chk(self.L.orc_add_packed_q(self.h, cb.ctypes.data, umi.ctypes.data, gene.ctypes.data, aux.ctypes.data, len(cb), arr,
                                          qual.ctypes.data, qual.shape[1]))

    def add_packed_qvar(self, cb, umi, gene, aux, qual, lens, side=()):
        """Like add_packed_q for strings of several lengths: qual uint8 [n, stride], lens uint8 [n]."""
        cb = np.ascontiguousarray(cb, np.uint64); umi = np.ascontiguousarray(umi, np.uint64)
        gene = np.ascontiguousarray(gene, np.uint32); aux = np.ascontiguousarray(aux, np.uint32)
        qual = np.ascontiguousarray(qual, np.uint8); lens = np.ascontiguousarray(lens, np.uint8)
        arr = (C.c_char_p * max(1, len(side)))(*[s.encode() for s in side])
        self._chk(self.L.orc_add_packed_qvar(self.h, cb.ctypes.data, umi.ctypes.data, gene.ctypes.data, aux.ctypes.data, len(cb), arr,
                                             qual.ctypes.data, qual.shape[1], lens.ctypes.data))

    def molecule_qualities_var(self, n_molecules, stride):
        """(sums [n, stride], lengths [n]) of every molecule, in the order of molecules()."""
        out = np.zeros((n_molecules, stride), np.uint32); lens = np.zeros(n_molecules, np.uint32)
        if self.L.orc_molecule_qualities_var(self.h, stride, out.ctypes.data, lens.ctypes.data) != 0:
            raise RuntimeError("a molecule's quality is longer than the stride")
        return out, lens

    def molecule_qualities(self, n_molecules, qlen):
        """UMI::_sum_quality of every molecule, in the order of molecules()."""
        out = np.zeros((n_molecules, qlen), np.uint32)
        if self.L.orc_molecule_qualities(self.h, qlen, out.ctypes.data) != 0:
            raise RuntimeError("a molecule has another quality length")
        return out

    def set_initialized(self):
        self._chk(self.L.orc_set_initialized(self.h))

    def merge_and_filter(self):
        self._chk(self.L.orc_merge_and_filter(self.h))

    def merge_umis_only(self):
        self._chk(self.L.orc_merge_umis_only(self.h))

    # ---- results ----
    @property
    def n_cells(self):
        return int(self.L.orc_n_cells(self.h))

    @property
    def n_genes(self):
        return int(self.L.orc_n_genes(self.h))

    @property
    def n_real(self):
        return int(self.L.orc_n_real(self.h))

    def filtered_cells(self):
        out = np.zeros(int(self.L.orc_n_filtered(self.h)), np.uint64)
        self.L.orc_filtered(self.h, out.ctypes.data)
        return out

    def merge_targets(self):
        out = np.zeros(int(self.L.orc_n_merge_targets(self.h)), np.uint64)
        self.L.orc_merge_targets(self.h, out.ctypes.data)
        return out

    def cell_id_by_cb(self, cb):
        return int(self.L.orc_cell_id_by_cb(self.h, cb.encode()))

    def cell_barcode(self, i):
        return self.L.orc_cell_barcode(self.h, i).decode()

    def gene_name(self, i):
        return self.L.orc_gene_name(self.h, i).decode()

    def chr_names(self):
        return [self.L.orc_chr_name(self.h, i).decode() for i in range(int(self.L.orc_n_chr(self.h)))]

    def cell_rows(self):
        """int64 [n_cells, 8]: merged, excluded, real, n_genes, req_genes, req_umis, total_reads, total_umis"""
        out = np.zeros((self.n_cells, 8), np.int64)
        self.L.orc_cell_rows(self.h, out.ctypes.data)
        return out

    def global_counters(self):
        out = np.zeros(4, np.uint64)
        self.L.orc_global_counters(self.h, out.ctypes.data)
        return out

    def molecules(self, stride=40):
        n = int(self.L.orc_molecules(self.h, None, None, None, stride, None, None))
        cell = np.zeros(n, np.uint64); gene = np.zeros(n, np.uint64); reads = np.zeros(n, np.uint64)
        mark = np.zeros(n, np.uint8); buf = C.create_string_buffer(max(1, n * stride))
        self.L.orc_molecules(self.h, cell.ctypes.data, gene.ctypes.data, buf, stride, reads.ctypes.data,
                             mark.ctypes.data)
        return cell, gene, _strs(buf.raw, n, stride), reads, mark

    def count_matrix(self, filtered=True, reads_output=False):
        n = int(self.L.orc_count_matrix(self.h, int(filtered), int(reads_output), None, None, None))
        g = np.zeros(n, np.uint64); c = np.zeros(n, np.uint64); v = np.zeros(n, np.uint64)
        self.L.orc_count_matrix(self.h, int(filtered), int(reads_output), g.ctypes.data, c.ctypes.data, v.ctypes.data)
        return g, c, v

    def merge_cells(self, src, tgt):
        self._chk(self.L.orc_merge_cells_explicit(self.h, src, tgt))

    def exclude_cell(self, cell):
        self._chk(self.L.orc_exclude_cell_explicit(self.h, cell))

    def count_matrix_levels(self, levels, reads_output=False):
        """get_count_matrix_filtered(container, query) for an explicit -L style code ("e", "i", "BA", ...)."""
        n = int(self.L.orc_count_matrix_levels(self.h, levels.encode(), int(reads_output), None, None, None))
        g = np.zeros(n, np.uint64); c = np.zeros(n, np.uint64); v = np.zeros(n, np.uint64)
        self.L.orc_count_matrix_levels(self.h, levels.encode(), int(reads_output), g.ctypes.data, c.ctypes.data, v.ctypes.data)
        return g, c, v

    def chr_stats(self):
        n = int(self.L.orc_chr_stats(self.h, None, None, None, None))
        cell = np.zeros(n, np.uint64); kind = np.zeros(n, np.int32); chr_ = np.zeros(n, np.uint64)
        cnt = np.zeros(n, np.int64)
        self.L.orc_chr_stats(self.h, cell.ctypes.data, kind.ctypes.data, chr_.ctypes.data, cnt.ctypes.data)
        return cell, kind, chr_, cnt

    def umi_distribution(self, stride=40):
        n = int(self.L.orc_umi_distribution(self.h, None, stride, None))
        buf = C.create_string_buffer(max(1, n * stride)); counts = np.zeros(n, np.uint64)
        self.L.orc_umi_distribution(self.h, buf, stride, counts.ctypes.data)
        return _strs(buf.raw, n, stride), counts

    # ---- fine-grained (pinning) ----
    def merge_target(self, cell):
        r = int(self.L.orc_merge_target(self.h, cell))
        if r == -2:
            raise RuntimeError(self.L.orc_last_error().decode())
        return r

    # PoissonTargetEstimator pieces (Tests/TestEstimationMergeProbs.cpp)
    def poisson_init(self):
        self._chk(self.L.orc_poisson_init(self.h))
        return int(self.L.orc_poisson_distribution_size(self.h))

    def poisson_gene_intersection(self, g1, g2):
        return float(self.L.orc_poisson_gene_intersection(self.h, g1, g2))

    def poisson_intersection_prob(self, c1, c2):
        return float(self.L.orc_poisson_intersection_prob(self.h, c1, c2))

    def poisson_expected_intersection(self, c1, c2):
        return float(self.L.orc_poisson_expected_intersection(self.h, c1, c2))

    def poisson_merge_target(self, cell):
        r = int(self.L.orc_poisson_merge_target(self.h, cell))
        if r == -2:
            raise RuntimeError(self.L.orc_last_error().decode())
        return r

    def real_neighbours(self, cell):
        out = np.zeros(4096, np.uint64)
        n = int(self.L.orc_real_neighbours(self.h, cell, out.ctypes.data, 4096))
        return out[:n]

    def umig_intersection(self, a, b):
        return int(self.L.orc_umig_intersection(self.h, a, b))

    def wl_part(self, p):
        return [self.L.orc_wl_barcode(self.h, p, i).decode() for i in range(int(self.L.orc_wl_part_size(self.h, p)))]

    def wl_parts(self):
        return int(self.L.orc_wl_parts(self.h))

    def wl_load(self, kind, path):
        self._chk(self.L.orc_wl_load(self.h, kind, path.encode()))

    def wl_set(self, kind, part0, part1):
        a = (C.c_char_p * len(part0))(*[s.encode() for s in part0])
        b = (C.c_char_p * len(part1))(*[s.encode() for s in part1])
        self._chk(self.L.orc_wl_set(self.h, kind, a, len(part0), b, len(part1)))

    def wl_split(self, cb, stride=64):
        buf = C.create_string_buffer(stride * 8)
        n = self.L.orc_wl_split(self.h, cb.encode(), buf, stride)
        if n < 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return _strs(buf.raw, n, stride)

    def wl_distances(self, cb, part):
        n = int(self.L.orc_wl_part_size(self.h, part))
        vals = np.zeros(n, np.int64); idx = np.zeros(n, np.uint64)
        self.L.orc_wl_distances(self.h, cb.encode(), part, vals.ctypes.data, idx.ctypes.data)
        return vals, idx

    def merge_umis_explicit(self, cell, gene, targets):
        src = (C.c_char_p * len(targets))(*[k.encode() for k in targets])
        tgt = (C.c_char_p * len(targets))(*[v.encode() for v in targets.values()])
        self._chk(self.L.orc_merge_umis_explicit(self.h, cell, gene.encode(), src, tgt, len(targets)))

    def directional_targets(self, umis):
        seqs = (C.c_char_p * len(umis))(*[s.encode() for s, _ in umis])
        reads = np.array([r for _, r in umis], np.uint64)
        stride = 64
        a = C.create_string_buffer(stride * len(umis)); b = C.create_string_buffer(stride * len(umis))
        n = self.L.orc_directional_targets(self.h, seqs, reads.ctypes.data, len(umis), a, b, stride)
        return dict(zip(_strs(a.raw, n, stride), _strs(b.raw, n, stride)))

    def molecule_dict(self):
        """{(cell barcode, gene name): {umi: (reads, mark)}} -- small cases only."""
        cell, gene, umi, reads, mark = self.molecules()
        out = {}
        for c, g, u, r, m in zip(cell, gene, umi, reads, mark):
            out.setdefault((self.cell_barcode(int(c)), self.gene_name(int(g))), {})[u] = (int(r), int(m))
        return out


def edit_distance(a, b, skip_n=True, max_ed=10000):
    return int(oracle_lib().orc_edit_distance(a.encode(), b.encode(), int(skip_n), max_ed))


def hamming_distance(a, b, skip_n=True):
    return int(oracle_lib().orc_hamming_distance(a.encode(), b.encode(), int(skip_n)))


def parse_encoded_id(s):
    cb = C.create_string_buffer(256); umi = C.create_string_buffer(256)
    if oracle_lib().orc_parse_encoded_id(s.encode(), cb, umi, 255) != 0:
        raise RuntimeError("unable to parse: " + s)
    return cb.value.decode(), umi.value.decode()


def reverse_complement(s):
    out = C.create_string_buffer(len(s) + 1)
    if oracle_lib().orc_reverse_complement(s.encode(), out, len(s) + 1) != 0:
        raise RuntimeError("bad base")
    return out.value.decode()


def fill_wrong_umi(umi):
    out = C.create_string_buffer(len(umi) + 1)
    oracle_lib().orc_fill_wrong_umi(umi.encode(), out, len(umi) + 1)
    return out.value.decode()


def poisson_upper_tail(k, lam):
    """P(X >= k), X ~ Poisson(lam) = Rcpp::ppois(k - 1, lam, lower = false)"""
    return float(oracle_lib().orc_poisson_upper_tail(k, lam))


def collisions_table(probs, max_expr):
    p = np.ascontiguousarray(probs, np.float64)
    out = np.zeros(max_expr, np.uint64)
    if oracle_lib().orc_collisions_table(p.ctypes.data, len(p), max_expr, out.ctypes.data) != 0:
        raise RuntimeError(oracle_lib().orc_last_error().decode())
    return out


# ---- gene annotation (-g): Tools/GeneAnnotation/*, ReadParamsParser::get_gene_from_reference -----------------
class GeneAnnotationOracle:
    """RefGenesContainer restated (oracle/gene_annotation_oracle.cpp)."""
    TYPE = {0: "NONE", 1: "INTRON", 2: "EXON"}

    def __init__(self, path):
        self.L = oracle_lib()
        L = self.L
        L.orc_ga_load.restype = C.c_void_p; L.orc_ga_load.argtypes = [C.c_char_p]
        L.orc_ga_last_error.restype = C.c_char_p
        L.orc_ga_free.argtypes = [C.c_void_p]
        L.orc_ga_n_chromosomes.restype = C.c_uint64; L.orc_ga_n_chromosomes.argtypes = [C.c_void_p]
        L.orc_ga_n_homogeneous.restype = C.c_long; L.orc_ga_n_homogeneous.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_ga_homogeneous_labels.restype = C.c_long; L.orc_ga_homogeneous_labels.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.orc_ga_parse_gtf.restype = C.c_int
        L.orc_ga_parse_gtf.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_ga_query.restype = C.c_long
        L.orc_ga_query.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.orc_ga_gene_for_read.restype = C.c_int
        L.orc_ga_gene_for_read.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int]
        self.h = L.orc_ga_load(path.encode())
        if not self.h:
            raise RuntimeError(L.orc_ga_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_ga_free(self.h)
            self.h = None

    def n_chromosomes(self):
        return int(self.L.orc_ga_n_chromosomes(self.h))

    def n_homogeneous(self, chr_):
        return int(self.L.orc_ga_n_homogeneous(self.h, chr_.encode()))

    def homogeneous_labels(self, chr_, i):
        return int(self.L.orc_ga_homogeneous_labels(self.h, chr_.encode(), i))

    def parse_gtf(self, line):
        chr_ = C.create_string_buffer(64); gid = C.create_string_buffer(64)
        s = C.c_uint64(); e = C.c_uint64()
        rc = self.L.orc_ga_parse_gtf(self.h, line.encode(), chr_, gid, 64, C.byref(s), C.byref(e))
        if rc < 0:
            raise RuntimeError(self.L.orc_ga_last_error().decode())
        return None if rc == 0 else (chr_.value.decode(), gid.value.decode(), int(s.value), int(e.value))

    def query(self, chr_, start, end):
        """get_gene_info -> [(gene name, type name)] in std::set order; None for an unknown chromosome"""
        names = C.create_string_buffer(64 * 64); types = (C.c_int * 64)()
        n = int(self.L.orc_ga_query(self.h, chr_.encode(), start, end, names, 64, types, 64))
        if n < 0:
            return None
        return [(names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode(), self.TYPE[types[i]]) for i in range(n)]

    def gene_for_read(self, chr_, position, end_position):
        """get_gene_from_reference -> (gene, mark bits); None for an unknown chromosome"""
        gene = C.create_string_buffer(128)
        m = int(self.L.orc_ga_gene_for_read(self.h, chr_.encode(), position, end_position, gene, 128))
        return None if m < 0 else (gene.value.decode(), m)


class IntervalsOracle:
    """IntervalsContainer<std::string> restated."""

    def __init__(self):
        self.L = oracle_lib()
        L = self.L
        L.orc_iv_new.restype = C.c_void_p
        L.orc_iv_free.argtypes = [C.c_void_p]
        L.orc_iv_add.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int]
        L.orc_iv_set_initialized.argtypes = [C.c_void_p, C.c_int]
        L.orc_iv_n_homogeneous.restype = C.c_uint64; L.orc_iv_n_homogeneous.argtypes = [C.c_void_p]
        L.orc_iv_homogeneous.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_iv_query.restype = C.c_long; L.orc_iv_query.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_int]
        self.h = L.orc_iv_new()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_iv_free(self.h)
            self.h = None

    def add(self, s, e, label, force=False):
        if self.L.orc_iv_add(self.h, s, e, label.encode(), int(force)) != 0:
            raise RuntimeError(self.L.orc_ga_last_error().decode())

    def set_initialized(self, clear=True):
        self.L.orc_iv_set_initialized(self.h, int(clear))

    def homogeneous(self):
        out = []
        for i in range(int(self.L.orc_iv_n_homogeneous(self.h))):
            s = C.c_uint64(); e = C.c_uint64()
            self.L.orc_iv_homogeneous(self.h, i, C.byref(s), C.byref(e))
            out.append((int(s.value), int(e.value)))
        return out

    def query(self, s, e):
        buf = C.create_string_buffer(64 * 32)
        n = int(self.L.orc_iv_query(self.h, s, e, buf, 64, 32))
        return [buf.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode() for i in range(n)]
