"""Synthetic 10x-shaped read streams (SURVEY.md §8d / BASELINE.md §4): table construction in numpy, the
per-read function in C (dropest_amd/csrc/synth.h) evaluated on the host or on the device."""
import ctypes as C
import os

import numpy as np

from . import capi

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "barcodes")
_RC = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def reverse_complement(s):
    return "".join(_RC[c] for c in reversed(s))


def load_whitelist(path):
    """Whitelist parts as the reference loads them (one part per line, reverse-complemented,
    Estimation/Merge/BarcodesParsing/BarcodesParser.cpp:117-144)."""
    parts = []
    with open(path) as f:
        for line in f:
            toks = line.split()
            if toks:
                parts.append([reverse_complement(t) for t in toks])
    return parts


def _cdf_u32(weights):
    w = np.asarray(weights, np.float64)
    cum = np.cumsum(w / w.sum())
    cdf = np.minimum(np.floor(cum * 4294967296.0), 4294967295.0).astype(np.uint64)
    # strictly increasing so that every entry has non-zero width
    for i in range(1, len(cdf)):
        if cdf[i] <= cdf[i - 1]:
            cdf[i] = cdf[i - 1] + 1
    cdf[-1] = 0xFFFFFFFF
    assert np.all(np.diff(cdf.astype(np.int64)) > 0) or len(cdf) == 1
    return cdf.astype(np.uint32)


class SynthStream:
    """A reproducible stream: read i is a pure function of (seed, stream_id, i)."""

    def __init__(self, n_reads, n_cells, n_genes=30000, cb_len=16, umi_len=10, whitelist="10x_aug_2016_split",
                 seed=20260928, stream_id=2, n_chr=25, permille_neighbour=50, permille_ambient=30,
                 permille_intergenic=70, permille_intron=60, permille_exon_na=40, reads_per_molecule=4,
                 sigma=0.6, zipf_s=1.1):
        self.n_reads, self.n_cells, self.n_genes = int(n_reads), int(n_cells), int(n_genes)
        self.cb_len, self.umi_len, self.n_chr = cb_len, umi_len, n_chr
        self.whitelist_path = os.path.join(DATA, whitelist) if whitelist else None
        rng = np.random.default_rng(seed ^ (stream_id << 20))
        if self.whitelist_path:
            parts = load_whitelist(self.whitelist_path)
            sizes = [len(p) for p in parts]
            total = int(np.prod(sizes))
            pick = rng.choice(total, size=self.n_cells, replace=False)
            cbs = []
            for k in pick:
                s, k = "", int(k)
                for p, sz in zip(reversed(parts), reversed(sizes)):
                    s = p[k % sz] + s
                    k //= sz
                cbs.append(s)
            assert all(len(s) == cb_len for s in cbs), "whitelist barcode length != cb_len"
            self.cell_barcodes = cbs
            self.cell_cb = np.array([capi.pack_seq(s) for s in cbs], np.uint64)
        else:
            codes = rng.choice(1 << (2 * cb_len), size=self.n_cells, replace=False).astype(np.uint64)
            self.cell_cb = codes | np.uint64(1 << (2 * cb_len))
            self.cell_barcodes = [capi.unpack_code(c) for c in self.cell_cb]
        self.cell_cdf = _cdf_u32(rng.lognormal(0.0, sigma, self.n_cells))
        self.gene_cdf = _cdf_u32(1.0 / np.arange(1, self.n_genes + 1, dtype=np.float64) ** zipf_s)
        p = capi.SynthParams()
        p.seed, p.stream_id = seed, stream_id
        p.n_cells, p.n_genes = self.n_cells, self.n_genes
        p.cell_cb, p.cell_cdf, p.gene_cdf = self.cell_cb.ctypes.data, self.cell_cdf.ctypes.data, self.gene_cdf.ctypes.data
        p.cb_len, p.umi_len, p.n_chr = cb_len, umi_len, n_chr
        p.permille_neighbour, p.permille_ambient, p.permille_intergenic = permille_neighbour, permille_ambient, permille_intergenic
        p.permille_intron, p.permille_exon_na = permille_intron, permille_exon_na
        p.n_effective = int(self.n_reads * (1 - permille_ambient / 1000.0) * (1 - permille_intergenic / 1000.0))
        p.reads_per_molecule = reads_per_molecule
        self.params = p

    def generate_host(self, first=0, n=None):
        n = self.n_reads - first if n is None else n
        cb = np.zeros(n, np.uint64); umi = np.zeros(n, np.uint64); gene = np.zeros(n, np.uint32); aux = np.zeros(n, np.uint32)
        rc = capi.lib().dropest_synth_generate_host(C.byref(self.params), first, n, cb.ctypes.data, umi.ctypes.data,
                                                    gene.ctypes.data, aux.ctypes.data)
        if rc != 0:
            raise RuntimeError("dropest_synth_generate_host failed (%d)" % rc)
        return cb, umi, gene, aux

    def generate_device(self, device=0, first=0, n=None):
        n = self.n_reads - first if n is None else n
        arrs = capi.DeviceArrays(device, n)
        rc = capi.lib().dropest_synth_generate_device(C.byref(self.params), device, first, n, *arrs.ptrs)
        if rc != 0:
            arrs.free()
            raise RuntimeError("dropest_synth_generate_device failed (%d)" % rc)
        return arrs


def inject_n(umi, gene, rate, seed, umi_len):
    """Replaces one base of a fraction `rate` of the UMIs of gene-bearing reads by N: returns (codes with
    escapes, side strings).  Side strings are registered in first-seen order over gene-bearing reads, which is
    what the C-ABI requires (the UMI of a read without a gene is ignored and must not be registered)."""
    rng = np.random.default_rng(seed)
    umi = umi.copy()
    hit = np.nonzero((rng.random(len(umi)) < rate) & (gene != capi.NO_GENE))[0]
    side, index = [], {}
    for i in hit:
        s = list(capi.unpack_code(umi[i]))
        s[int(rng.integers(0, umi_len))] = "N"
        s = "".join(s)
        k = index.get(s)
        if k is None:
            k = index[s] = len(side)
            side.append(s)
        umi[i] = capi.ESCAPE | k
    return umi, side
