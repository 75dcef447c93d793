"""Gene annotation from GTF / BED (-g): the product's flat, binary-search implementation
(dropest_amd/csrc/host/gene_annotation.cpp) against the oracle's restatement of the reference's IntervalsContainer
design (oracle/gene_annotation_oracle.cpp, pinned on Tests/TestTools.cpp in test_oracle_reference_kat.py).
Host-only code: runs without a GPU."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from dropest_amd.build import FACADE_LIB, build_facade
from oracle import binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
GTF = os.path.join(HERE, "golden", "gtf_test.gtf.gz")
TYPE = {0: "NONE", 1: "INTRON", 2: "EXON"}


class Product:
    def __init__(self, path):
        build_facade()
        L = self.L = C.CDLL(FACADE_LIB)
        L.dropest_gene_annotation_load.restype = C.c_void_p; L.dropest_gene_annotation_load.argtypes = [C.c_char_p]
        L.dropest_gene_annotation_error.restype = C.c_char_p
        L.dropest_gene_annotation_free.argtypes = [C.c_void_p]
        L.dropest_gene_annotation_query.restype = C.c_long
        L.dropest_gene_annotation_query.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.dropest_gene_annotation_read.restype = C.c_int
        L.dropest_gene_annotation_read.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int]
        self.h = L.dropest_gene_annotation_load(path.encode())
        if not self.h:
            raise RuntimeError(L.dropest_gene_annotation_error().decode())

    def query(self, chr_, s, e):
        names = C.create_string_buffer(64 * 64); types = (C.c_int * 64)()
        n = int(self.L.dropest_gene_annotation_query(self.h, chr_.encode(), s, e, names, 64, types, 64))
        if n < 0:
            return None
        return [(names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode(), TYPE[types[i]]) for i in range(n)]

    def gene_for_read(self, chr_, pos, end):
        gene = C.create_string_buffer(128)
        m = int(self.L.dropest_gene_annotation_read(self.h, chr_.encode(), pos, end, gene, 128))
        return None if m < 0 else (gene.value.decode(), m)


def _compare(path, chrs, lo, hi, n, seed):
    p, o = Product(path), ob.GeneAnnotationOracle(path)
    rng = np.random.default_rng(seed)
    hits = 0
    for _ in range(n):
        c = chrs[int(rng.integers(0, len(chrs)))]
        s = int(rng.integers(lo, hi))
        ln = int(rng.choice([1, 1, 1, 2, 10, 100, 1000]))
        assert p.query(c, s, s + ln) == o.query(c, s, s + ln), (c, s, ln)
        e = s + int(rng.integers(1, 400))
        got, want = p.gene_for_read(c, s, e), o.gene_for_read(c, s, e)
        assert got == want, (c, s, e, got, want)
        hits += bool(want and want[0])
    return hits


def test_reference_known_answers_through_the_product():
    """testGenesWithIntrons (Tests/TestTools.cpp:266-287) on the reference's data/gtf/gtf_test.gtf.gz."""
    p = Product(GTF)
    assert p.query("chr1", 20000, 20010) == [("WASH7P", "INTRON")]
    assert p.query("chr1", 24750, 24760) == [("WASH7P", "EXON")]
    assert p.query("chr1", 10, 20) == [] and p.query("chrNope", 10, 20) is None
    assert p.gene_for_read("chr1", 24750, 24800) == ("WASH7P", 2) and p.gene_for_read("chrNope", 1, 50) is None


def test_product_equals_oracle_on_the_reference_gtf():
    assert _compare(GTF, ["chr1", "chr2", "chr3", "chrM", "chrNope"], 0, 60_000, 4000, 1) > 200


@pytest.mark.parametrize("seed,with_introns,bed", [(1, False, False), (2, True, False), (3, False, True)])
def test_product_equals_oracle_on_random_annotations(tmp_path, seed, with_introns, bed):
    """Overlapping transcripts of several genes, touching and nested exons, single-base exons, records without a
    transcript id, comment and malformed lines; explicit intron records; the BED flavour."""
    rng = np.random.default_rng(seed)
    lines = ["# comment", "chr1\tsrc\tgene\t1\t1000\t.\t+\t.\tgene_id \"G0\";", "too short"]
    for chr_ in ("chr1", "chr2", "chrX"):
        for g in range(40):
            gs = int(rng.integers(0, 90_000))
            for t in range(int(rng.integers(1, 4))):
                pos = gs + int(rng.integers(0, 300))
                # (records without a transcript id fall into one transcript per gene: with explicit introns that would make
                #  exon and intron records overlap, which the reference rejects -- covered by test_bad_annotation_files)
                tid = "T%s_%d_%d" % (chr_, g, t) if (with_introns or rng.random() < 0.85) else ""
                prev_end = None
                for x in range(int(rng.integers(1, 7))):
                    ln = int(rng.choice([1, 5, 50, 200, 800]))
                    if bed:
                        lines.append("%s\t%d\t%d\tG%s_%d" % (chr_, pos, pos + ln, chr_, g))
                    else:
                        attrs = 'gene_id "G%s_%d"; gene_name "N%s_%d";' % (chr_, g, chr_, g) + (' transcript_id "%s";' % tid if tid else "") + ' tss_id "x";'
                        if with_introns and prev_end is not None and pos > prev_end + 1:
                            lines.append("%s\tsrc\tintron\t%d\t%d\t.\t+\t.\t%s" % (chr_, prev_end + 1, pos, attrs))
                        lines.append("%s\tsrc\texon\t%d\t%d\t.\t+\t.\t%s" % (chr_, pos + 1, pos + ln, attrs))
                    prev_end = pos + ln
                    pos += ln + int(rng.choice([0, 0, 1, 30, 400]))
    path = str(tmp_path / ("ann.bed.gz" if bed else "ann.gtf.gz"))
    with gzip.open(path, "wt") as f:
        f.write("\n".join(lines) + "\n")
    assert _compare(path, ["chr1", "chr2", "chrX", "chrNope"], 0, 95_000, 5000, seed) > 300


def test_bad_annotation_files(tmp_path):
    build_facade()
    overlap = ('chr1\ts\texon\t1\t100\t.\t+\t.\tgene_id "A"; transcript_id "T";\n'
               'chr1\ts\tintron\t50\t150\t.\t+\t.\tgene_id "A"; transcript_id "T";\n')
    for name, text in (("x.txt", "chr1\t1\t2\tG\n"), ("overlap.gtf", overlap), ("dup.gtf", 'chr1\ts\texon\t1\t9\t.\t+\t.\tgene_id "A"; transcript_id "T";\n'
                                                                'chr1\ts\texon\t20\t29\t.\t+\t.\tgene_id "B"; transcript_id "T";\n')):
        path = str(tmp_path / name)
        open(path, "w").write(text)
        with pytest.raises(RuntimeError):
            Product(path)
        with pytest.raises(RuntimeError):
            ob.GeneAnnotationOracle(path)
