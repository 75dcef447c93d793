"""Gene annotation of reads ON THE DEVICE (include/dropest_annotation.h, dropest_amd/csrc/annotation_api.hip): the flat
tables of the host loader uploaded to the GPU, one thread per read, against the host implementation
(RefGenesContainer::gene_of_alignment) and the oracle on the reference's GTF and on random annotations."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from dropest_amd import capi
from oracle import binding as ob

from test_gene_annotation import GTF, Product

pytestmark = pytest.mark.gpu


class DeviceAnnotation:
    def __init__(self, product):
        F, L = product.L, capi.lib()
        F.dropest_gene_annotation_flat_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        F.dropest_gene_annotation_flat_fill.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        F.dropest_gene_annotation_chr_name.restype = C.c_char_p; F.dropest_gene_annotation_chr_name.argtypes = [C.c_void_p, C.c_uint32]
        F.dropest_gene_annotation_gene_name.restype = C.c_char_p; F.dropest_gene_annotation_gene_name.argtypes = [C.c_void_p, C.c_uint32]
        sizes = (C.c_uint32 * 8)()
        F.dropest_gene_annotation_flat_sizes(product.h, sizes)
        n_chr, n_seg, n_tr, n_genes, n_cover, n_exon, n_intron, use_introns = [int(x) for x in sizes]
        lens = [n_chr + 1, n_seg, n_seg, n_seg + 1, n_cover, n_tr, n_tr + 1, n_tr + 1, n_exon, n_exon, n_intron, n_intron]
        self.arrays = [np.zeros(max(1, k), np.uint32) for k in lens]
        ptrs = (C.c_void_p * 12)(*[a.ctypes.data for a in self.arrays])
        F.dropest_gene_annotation_flat_fill(product.h, ptrs)
        self.chr_index = {F.dropest_gene_annotation_chr_name(product.h, i).decode(): i for i in range(n_chr)}
        self.genes = [F.dropest_gene_annotation_gene_name(product.h, i).decode() for i in range(n_genes)]

        class Flat(C.Structure):
            _fields_ = [("n_chr", C.c_uint32), ("n_seg", C.c_uint32), ("n_tr", C.c_uint32), ("n_genes", C.c_uint32),
                        ("use_introns_from_gtf", C.c_int32)] + [(nm, C.c_void_p) for nm in (
                            "chr_seg_begin", "seg_start", "seg_end", "seg_tr_begin", "seg_tr", "tr_gene", "tr_exon_begin", "tr_intron_begin",
                            "exon_start", "exon_end", "intron_start", "intron_end")]
        flat = Flat(n_chr, n_seg, n_tr, n_genes, use_introns, *[a.ctypes.data for a in self.arrays])
        L.dropest_annotation_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.dropest_annotation_query.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 5
        L.dropest_annotation_destroy.argtypes = [C.c_void_p]
        L.dropest_annotation_last_error.restype = C.c_char_p
        self.L, self.h = L, C.c_void_p()
        assert L.dropest_annotation_create(0, C.byref(flat), C.byref(self.h)) == 0, L.dropest_annotation_last_error()

    def query(self, chrs, pos, end):
        n = len(chrs)
        ci = np.array([self.chr_index.get(c, -1) for c in chrs], np.int32)
        pos = np.ascontiguousarray(pos, np.uint32); end = np.ascontiguousarray(end, np.uint32)
        gene = np.zeros(n, np.uint32); mark = np.zeros(n, np.int32)
        assert self.L.dropest_annotation_query(self.h, n, ci.ctypes.data, pos.ctypes.data, end.ctypes.data, gene.ctypes.data, mark.ctypes.data) == 0
        return [None if m == -1 else ("" if g == 0xFFFFFFFF else self.genes[int(g)], int(m)) for g, m in zip(gene, mark)]

    def close(self):
        self.L.dropest_annotation_destroy(self.h)


def _check(path, chrs, lo, hi, n, seed):
    p = Product(path)
    o = ob.GeneAnnotationOracle(path)
    d = DeviceAnnotation(p)
    rng = np.random.default_rng(seed)
    cs = [chrs[int(i)] for i in rng.integers(0, len(chrs), n)]
    pos = rng.integers(lo, hi, n)
    end = pos + rng.integers(1, 400, n)
    got = d.query(cs, pos, end)
    hits = 0
    for i in range(n):
        want = p.gene_for_read(cs[i], int(pos[i]), int(end[i]))
        assert got[i] == want, (cs[i], int(pos[i]), int(end[i]), got[i], want)
        if i % 7 == 0:
            assert want == o.gene_for_read(cs[i], int(pos[i]), int(end[i]))
        hits += bool(want and want[0])
    d.close()
    return hits


def test_device_annotation_on_the_reference_gtf():
    assert _check(GTF, ["chr1", "chr2", "chr3", "chrM", "chrNope"], 0, 60_000, 20_000, 1) > 1000


@pytest.mark.parametrize("seed,with_introns", [(1, False), (2, True)])
def test_device_annotation_on_random_annotations(tmp_path, seed, with_introns):
    rng = np.random.default_rng(seed)
    lines = []
    for chr_ in ("chr1", "chr2", "chrX"):
        for g in range(40):
            gs = int(rng.integers(0, 90_000))
            for t in range(int(rng.integers(1, 4))):
                pos = gs + int(rng.integers(0, 300))
                tid = "T%s_%d_%d" % (chr_, g, t) if (with_introns or rng.random() < 0.85) else ""
                prev_end = None
                for x in range(int(rng.integers(1, 7))):
                    ln = int(rng.choice([1, 5, 50, 200, 800]))
                    attrs = 'gene_id "G%s_%d"; gene_name "N%s_%d";' % (chr_, g, chr_, g) + (' transcript_id "%s";' % tid if tid else "") + ' tss_id "x";'
                    if with_introns and prev_end is not None and pos > prev_end + 1:
                        lines.append("%s\tsrc\tintron\t%d\t%d\t.\t+\t.\t%s" % (chr_, prev_end + 1, pos, attrs))
                    lines.append("%s\tsrc\texon\t%d\t%d\t.\t+\t.\t%s" % (chr_, pos + 1, pos + ln, attrs))
                    prev_end = pos + ln
                    pos += ln + int(rng.choice([0, 0, 1, 30, 400]))
    path = str(tmp_path / "ann.gtf.gz")
    with gzip.open(path, "wt") as f:
        f.write("\n".join(lines) + "\n")
    assert _check(path, ["chr1", "chr2", "chrX", "chrNope"], 0, 95_000, 30_000, seed) > 3000
