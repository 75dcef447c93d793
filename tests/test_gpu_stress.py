"""Randomised parity sweep: many small, adversarially shaped streams (heavy duplication, one giant run, all-distinct
barcodes, tile-boundary sizes, variable lengths, Ns, sparse gene ids) through the HIP path and the oracle."""
import os

import numpy as np
import pytest

from dropest_amd import capi
from oracle import Oracle

import parity

pytestmark = pytest.mark.gpu

# soak runs: DROPEST_STRESS_SEED_OFFSET=<n> shifts every random stream of this file (default 0 = the committed cases)
import os as _os
SEED_OFFSET = int(_os.environ.get("DROPEST_STRESS_SEED_OFFSET", "0"))
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dropest_amd", "data", "barcodes")


def random_stream(rng, n, n_cb, n_gene, n_umi, cb_len=(12, 12), umi_len=(6, 6), n_rate=0.0, p_nogene=0.1, n_chr=5,
                  cb_n_rate=0.0):
    def seqs(count, lo, hi):
        out = []
        for _ in range(count):
            L = int(rng.integers(lo, hi + 1))
            out.append("".join(rng.choice(list("ACGT"), L)))
        return out
    cbs, umis = seqs(n_cb, *cb_len), seqs(n_umi, *umi_len)
    side, index = [], {}

    def code(s):
        c = capi.pack_seq(s)
        if c is not None:
            return c
        k = index.get(s)
        if k is None:
            k = index[s] = len(side)
            side.append(s)
        return capi.ESCAPE | k

    def with_n(s, rate):
        if rate and rng.random() < rate:
            i = int(rng.integers(0, len(s)))
            return s[:i] + "N" + s[i + 1:]
        return s
    cb = np.zeros(n, np.uint64); umi = np.zeros(n, np.uint64); gene = np.zeros(n, np.uint32); aux = np.zeros(n, np.uint32)
    # zipf-ish choice so that some barcodes / UMIs are very hot
    w_cb = 1.0 / np.arange(1, n_cb + 1) ** rng.uniform(0.0, 1.5); w_cb /= w_cb.sum()
    w_umi = 1.0 / np.arange(1, n_umi + 1) ** rng.uniform(0.0, 1.0); w_umi /= w_umi.sum()
    ci = rng.choice(n_cb, n, p=w_cb); ui = rng.choice(n_umi, n, p=w_umi)
    gi = rng.integers(0, n_gene, n)
    for r in range(n):
        has_gene = rng.random() >= p_nogene
        cb[r] = code(with_n(cbs[ci[r]], cb_n_rate))
        # the UMI of a gene-less read is ignored by the path and must not register a side string
        umi[r] = code(with_n(umis[ui[r]], n_rate)) if has_gene else capi.pack_seq(umis[ui[r]])
        gene[r] = gi[r] if has_gene else capi.NO_GENE
        aux[r] = int(rng.integers(0, n_chr)) | (int(rng.choice([1, 2, 3, 4, 5, 6, 7])) << 16)
    return parity.canonical_stream(cb, umi, gene, aux) + (side,)


def run_case(rng, **kw):
    min_before = int(kw.pop("min_before", rng.integers(0, 4)))
    min_after = int(kw.pop("min_after", rng.integers(0, 6)))
    levels = kw.pop("levels", str(rng.choice(["eEBA", "e", "eE", "iIBA", "A"])))
    cb, umi, gene, aux, side = random_stream(rng, **kw)
    o = parity.oracle_run(Oracle, dict(min_genes_before=min_before, min_genes_after=min_after, match_levels=levels),
                          cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(min_genes_before_merge=min_before, min_genes_after_merge=min_after, gene_match_levels=levels),
                       cb, umi, gene, aux, side, chunks=int(rng.integers(1, 4)))
    parity.compare(o, c, side)


@pytest.mark.parametrize("seed", range(12))
def test_random_small_streams(seed):
    rng = np.random.default_rng(1000 + seed + SEED_OFFSET)
    run_case(rng, n=int(rng.integers(1, 6000)), n_cb=int(rng.integers(1, 60)), n_gene=int(rng.integers(1, 40)),
             n_umi=int(rng.integers(1, 80)))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 16385])
def test_tile_boundary_sizes(n):
    run_case(np.random.default_rng(n), n=n, n_cb=7, n_gene=5, n_umi=9, min_before=0, min_after=0)


def test_one_giant_molecule_and_all_distinct_barcodes():
    rng = np.random.default_rng(5)
    run_case(rng, n=20_000, n_cb=1, n_gene=1, n_umi=1, p_nogene=0.0, min_before=0, min_after=0)     # a single run of 20k reads
    run_case(rng, n=20_000, n_cb=1, n_gene=1, n_umi=1, p_nogene=1.0, min_before=0, min_after=0)     # only gene-less reads
    run_case(rng, n=12_000, n_cb=12_000, n_gene=3, n_umi=4, cb_len=(16, 16), min_before=0, min_after=0)   # table growth path


@pytest.mark.parametrize("seed", range(6))
def test_sampled_table_sizing_and_hot_list_on_small_streams(seed, monkeypatch):
    """The barcode table sized from every 64th read and the LDS table of the hot barcodes normally start at 2^22 reads;
    DROPEST_CB_SAMPLE_MIN brings both (with the growth retry when the sample underestimates) to streams the oracle runs in seconds:
    skewed barcode frequencies, all-distinct barcodes, Ns in barcodes."""
    monkeypatch.setenv("DROPEST_CB_SAMPLE_MIN", "1000")
    rng = np.random.default_rng(8800 + seed + SEED_OFFSET)
    run_case(rng, n=int(rng.integers(20_000, 60_000)), n_cb=int(rng.integers(200, 6000)), n_gene=int(rng.integers(5, 400)),
             n_umi=int(rng.integers(20, 400)), cb_len=(12, 12) if seed % 2 else (9, 14), cb_n_rate=0.02 if seed == 3 else 0.0)
    if seed == 0:
        run_case(rng, n=30_000, n_cb=30_000, n_gene=3, n_umi=4, cb_len=(16, 16), min_before=0, min_after=0)   # nothing repeats: no hot list, growth


@pytest.mark.parametrize("twist", ["longer_umi", "escaped_umi", "gene_on_two_chromosomes", "higher_gene_id", "none",
                                   "chromosome_ids_past_254", "chromosome_ids_past_254_gene_on_two", "rare_gene_on_two_chromosomes"])
def test_key_layout_planned_from_a_sample_is_checked(twist, monkeypatch):
    """Large single-context passes plan the key layout from every 256th read and gather the exact statistics with the key pass
    (cb_insert then reads the barcodes only).  What the sample cannot see -- one longer UMI, the only UMI with an N, a gene that
    also occurs on a second chromosome, a gene id one bit wider -- must lead to a second key pass, never to a wrong key.  The key pass
    answers "same chromosome as before?" from a byte table in LDS (k_misc.h: GCL); chromosome ids that do not fit a byte, genes the sample
    never saw and genes past the table take the exact path."""
    monkeypatch.setenv("DROPEST_CB_SAMPLE_MIN", "1000")
    rng = np.random.default_rng(4242)
    cb, umi, gene, aux, side = random_stream(rng, n=30_000, n_cb=300, n_gene=100, n_umi=200, cb_len=(12, 12), umi_len=(6, 6), p_nogene=0.05)
    side = list(side)
    has = gene != capi.NO_GENE                                # the chromosome a function of the gene (random_stream draws it per read)
    aux = np.where(has, (aux & np.uint32(0xFFFF0000)) | (gene % 5).astype(np.uint32), aux).astype(np.uint32)
    at = 12_345                                               # not a multiple of 256 / 2048: neither sample sees this read
    if gene[at] == capi.NO_GENE:
        at += 1
    assert at % 256 and gene[at] != capi.NO_GENE
    if twist == "longer_umi":
        umi[at] = capi.pack_seq("ACGTACGT")
    elif twist == "escaped_umi":
        side.append("ACNTAC"); umi[at] = capi.ESCAPE | (len(side) - 1)
    elif twist == "gene_on_two_chromosomes":
        aux[at] = (aux[at] & np.uint32(0xFFFF0000)) | np.uint32(((int(aux[at]) & 0xFFFF) + 1) % 5) | np.uint32(2 << 16)
    elif twist == "higher_gene_id":
        gene[at] = 100 + 200                                  # ids stay first-seen dense enough for the test: canonical_stream below
    elif twist.startswith("chromosome_ids_past_254"):
        aux = np.where(has, aux + np.uint32(300), aux).astype(np.uint32)       # 300 .. 304: no byte of the LDS table can hold them
        if twist.endswith("gene_on_two"):
            aux[at] = (aux[at] & np.uint32(0xFFFF0000)) | np.uint32(300 + ((int(aux[at]) & 0xFFFF) - 300 + 1) % 5) | np.uint32(2 << 16)
    elif twist == "rare_gene_on_two_chromosomes":
        for k, pos in enumerate((at, at + 4001)):             # a gene with two reads only, on two chromosomes, neither read in a sample
            assert pos % 256 and pos % 2048
            gene[pos] = 100; aux[pos] = np.uint32((2 << 16) | (1 + k))
    cb, umi, gene, aux = parity.canonical_stream(cb, umi, gene, aux)
    o = parity.oracle_run(Oracle, dict(min_genes_before=1, min_genes_after=2), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(min_genes_before_merge=1, min_genes_after_merge=2), cb, umi, gene, aux, side, profile=True)
    parity.compare(o, c, side)
    redone = c.kernel_stats().get("count:key_plan_redone", {"launches": 0})["launches"]
    assert (redone >= 1) == (twist in ("longer_umi", "escaped_umi", "gene_on_two_chromosomes", "chromosome_ids_past_254_gene_on_two",
                                       "rare_gene_on_two_chromosomes")), (twist, redone)


def test_variable_lengths_and_ns():
    rng = np.random.default_rng(9)
    run_case(rng, n=5000, n_cb=30, n_gene=12, n_umi=40, cb_len=(8, 19), umi_len=(6, 6), n_rate=0.05, min_before=0)
    run_case(rng, n=5000, n_cb=30, n_gene=12, n_umi=25, cb_len=(10, 10), umi_len=(4, 9), min_before=1)       # mixed UMI lengths keep the sentinel
    run_case(rng, n=4000, n_cb=25, n_gene=6, n_umi=12, umi_len=(5, 5), n_rate=0.3, cb_n_rate=0.1, min_before=0)   # many Ns, also in barcodes


def test_key_wider_than_64_bits_is_refused_loudly():
    """cell + gene + UMI = 6 + 20 + 40 bits: one context refuses it and names dropest_ctx_split (gene + UMI fields that ALONE reach 64 bits take
    the UMI dictionary instead: tests/test_gpu_umi_dict.py)."""
    P = capi.pack_seq
    n = 40
    cb = np.array([P("ACGT" * 7 + "".join("ACGT"[(i >> (2 * j)) & 3] for j in range(3))) for i in range(n)], np.uint64)      # 31 bases, all distinct
    umi = np.array([P("TTGCA" * 4)] * n, np.uint64)                                              # 20 bases -> 40 bits
    gene = np.array([1_000_000] * n, np.uint32)                                                  # 20 bits
    c = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    c.push_reads(cb, umi, gene, np.full(n, 2 << 16, np.uint32))
    with pytest.raises(capi.DropestError) as e:
        c.set_initialized()
    assert e.value.status == 4 and "bits" in str(e.value) and "dropest_ctx_split" in str(e.value)


@pytest.mark.parametrize("poisson", [False, True])
@pytest.mark.parametrize("n_parts", [5, 6, 8])
def test_whitelists_of_more_than_four_parts(n_parts, poisson, tmp_path):
    """The reference's const-length whitelist files have one part per line and no limit on the lines
    (ConstLengthBarcodesParser.cpp:50-68); beyond the four parts the device kernel is built for the neighbour search runs on
    the host (merge_host.h: search_merge_candidates_host), literally as the reference runs it."""
    for seed in range(3):
        test_random_whitelist_merges(100 * n_parts + seed, poisson, tmp_path, force_parts=n_parts)


def random_whitelist_case(seed, poisson, tmp_path, force_parts=None):
    """-> (cb, umi, gene, aux, side, oracle kwargs, C-ABI kwargs) of one random whitelist merge (see test_random_whitelist_merges)."""
    rng = np.random.default_rng(7000 + seed + SEED_OFFSET)
    rc = {"A": "T", "C": "G", "G": "C", "T": "A"}
    def rnd(L):
        return "".join(rng.choice(list("ACGT"), L))
    const_kind = bool(rng.integers(0, 2)) or force_parts is not None
    l1 = int(rng.integers(3, 7)); l2 = int(rng.integers(5, 9))
    if force_parts:
        l1, l2 = 3, 3
    p1 = sorted({rnd(l1 if const_kind else int(rng.integers(l1, l1 + 2))) for _ in range(int(rng.integers(3, 9)))})
    p2 = sorted({rnd(l2) for _ in range(int(rng.integers(3, 10)))})
    lines = [p1, p2]
    if force_parts or (const_kind and seed % 3 != 1):   # const-length files may have any number of lines: one, three or four parts too
        n_parts = force_parts or int(rng.choice([1, 3, 4]))
        lines = ([p1] if n_parts == 1 else
                 [p1, p2] + [sorted({rnd(int(rng.integers(2, 5)) if k == 0 else 3) for _ in range(int(rng.integers(2, 6)))}) for k in range(n_parts - 2)])
        for k in range(2, len(lines)):     # one length per line
            L = len(lines[k][0]); lines[k] = sorted({x[:L].ljust(L, "A") for x in lines[k]})
    wl = tmp_path / "wl"
    # the file stores reverse complements (the loader reverses them back, BarcodesParser.cpp:140)
    wl.write_text("".join(" ".join("".join(rc[c] for c in reversed(s)) for s in part) + "\n" for part in lines))
    real = [""]
    for part in lines:
        real = [a + b for a in real for b in part]
    rng.shuffle(real)
    real = real[:max(2, len(real) // 2)]
    if force_parts:
        real = real[:400]

    def mutate(s):
        k = rng.integers(0, 5)
        i = int(rng.integers(0, len(s)))
        if k == 0:
            return s[:i] + str(rng.choice(list("ACGT"))) + s[i + 1:]
        if k == 1 and not const_kind:
            return s[:i] + str(rng.choice(list("ACGT"))) + s[i:]
        if k == 2 and not const_kind and len(s) > l2 + 2:
            return s[:i] + s[i + 1:]
        if k == 3:
            return s[:i] + "N" + s[i + 1:]
        return s
    pool = list(real) + [mutate(str(rng.choice(real))) for _ in range(40)] + [mutate(mutate(str(rng.choice(real)))) for _ in range(15)]
    genes = ["g%d" % i for i in range(int(rng.integers(2, 12)))]
    # -M: enough distinct UMIs that no gene's size comes near their number (the collisions adjustment diverges there)
    umis = [rnd(6) for _ in range(int(rng.integers(300, 1000)))] if poisson else [rnd(5) for _ in range(int(rng.integers(3, 25)))]
    n = int(rng.integers(200, 3000))
    w = np.concatenate([np.full(len(real), 8.0), np.ones(len(pool) - len(real))]); w /= w.sum()
    side, index, gids = [], {}, {}

    def code(s):
        c = capi.pack_seq(s)
        if c is not None:
            return c
        if s not in index:
            index[s] = len(side); side.append(s)
        return capi.ESCAPE | index[s]
    cb = np.array([code(pool[i]) for i in rng.choice(len(pool), n, p=w)], np.uint64)
    umi = np.array([code(str(u)) for u in rng.choice(umis, n)], np.uint64)
    gene = np.array([gids.setdefault(str(g), len(gids)) for g in rng.choice(genes, n)], np.uint32)
    aux = np.full(n, 2 << 16, np.uint32)
    frac = 0.0 if rng.integers(0, 2) else 0.2
    min_before = int(rng.integers(0, 3))
    kind = capi.BARCODES_CONST if const_kind else capi.BARCODES_INDROP
    if poisson:   # PoissonRealBarcodesMergeStrategy: wider neighbour levels, real bases can merge, probability thresholds
        p_merge, p_real = float(rng.choice([1e-4, 0.05, 0.9])), float(rng.choice([1e-7, 0.05, 0.9]))
        okw = dict(merge_kind=3, barcodes_kind=kind, barcodes_file=str(wl), min_genes_before=min_before, min_genes_after=min_before,
                   max_merge_prob=p_merge, max_real_merge_prob=p_real)
        gkw = dict(merge_kind=capi.MERGE_POISSON_REAL, barcodes_kind=kind, barcodes_file=str(wl), min_genes_before_merge=min_before,
                   min_genes_after_merge=min_before, max_merge_prob=p_merge, max_real_merge_prob=p_real)
    else:
        okw = dict(merge_kind=1, barcodes_kind=kind, barcodes_file=str(wl), min_genes_before=min_before, min_genes_after=min_before,
                   min_merge_fraction=frac)
        gkw = dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=kind, barcodes_file=str(wl), min_genes_before_merge=min_before,
                   min_genes_after_merge=min_before, min_merge_fraction=frac)
    return cb, umi, gene, aux, side, okw, gkw


@pytest.mark.parametrize("poisson", [False, True])
@pytest.mark.parametrize("seed", range(10))
def test_random_whitelist_merges(seed, poisson, tmp_path, force_parts=None):
    """Random small whitelists (inDrop-style two lines, variable first-part length allowed) and barcodes that are exact,
    mutated (substitution / insertion / deletion -> different length) or carry an N: stresses the neighbour search,
    the tie replay (min_merge_fraction 0 half of the time) and the sequential merge application."""
    cb, umi, gene, aux, side, okw, gkw = random_whitelist_case(seed, poisson, tmp_path, force_parts)
    o = parity.oracle_run(Oracle, okw, cb, umi, gene, aux, side)
    c = parity.gpu_run(gkw, cb, umi, gene, aux, side)
    parity.compare(o, c, side)


@pytest.mark.parametrize("seed", range(10))
def test_random_directional_umi_merge(seed):
    """-u on adversarial streams: very short UMIs (everything is a neighbour of everything), several UMI lengths in
    one run (host replay of the banded edit distance), Ns (random fills), hot genes with more than 16 UMIs."""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    rng = np.random.default_rng(7000 + seed + SEED_OFFSET)
    var_len = seed % 3 == 0
    cb, umi, gene, aux, side = random_stream(
        rng, n=int(rng.integers(200, 9000)), n_cb=int(rng.integers(1, 12)), n_gene=int(rng.integers(1, 12)),
        n_umi=int(rng.integers(2, 120)), umi_len=(3, 5) if var_len else (4, 4), n_rate=0.05 if seed % 2 else 0.0)
    max_ed, mult = int(rng.integers(1, 4)), float(rng.choice([1.0, 1.5, 2.0, 3.0]))
    libc.srand(1)
    o = parity.oracle_run(Oracle, dict(min_genes_before=0, min_genes_after=0, umi_merge_kind=1, max_umi_merge_ed=max_ed,
                                       umi_mult=mult), cb, umi, gene, aux, side)
    libc.srand(1)
    c = parity.gpu_run(dict(min_genes_before_merge=0, min_genes_after_merge=0, umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL,
                            max_umi_merge_edit_distance=max_ed, umi_merge_multiplier=mult), cb, umi, gene, aux, side)
    parity.compare(o, c, side)


@pytest.mark.parametrize("seed", range(10))
def test_random_merge_all(seed):
    """merge_type = all: chains of merges towards ever larger cells; barcodes of one length go through the device
    kernel, mixed lengths and barcodes with N through the host's banded edit distance."""
    rng = np.random.default_rng(9700 + seed + SEED_OFFSET)
    cb, umi, gene, aux, side = random_stream(
        rng, n=int(rng.integers(300, 8000)), n_cb=int(rng.integers(2, 60)), n_gene=int(rng.integers(1, 15)),
        n_umi=int(rng.integers(2, 60)), cb_len=(6, 8) if seed % 3 == 0 else (8, 8), cb_n_rate=0.02 if seed % 4 == 1 else 0.0)
    max_ed, mb = int(rng.integers(0, 9)), int(rng.integers(0, 3))
    o = parity.oracle_run(Oracle, dict(merge_kind=5, max_cb_merge_ed=max_ed, min_genes_before=mb, min_genes_after=mb), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_ALL, max_cb_merge_edit_distance=max_ed, min_genes_before_merge=mb, min_genes_after_merge=mb),
                       cb, umi, gene, aux, side)
    parity.compare(o, c, side)


@pytest.mark.parametrize("seed", range(10))
def test_random_poisson_simple_merge(seed):
    """-M without a whitelist: few genes / many exact probability ties (resolved by the unordered_map order replay),
    barcodes with N, every edit-distance threshold and loose to strict probability thresholds."""
    rng = np.random.default_rng(9500 + seed + SEED_OFFSET)
    cb, umi, gene, aux, side = random_stream(
        rng, n=int(rng.integers(300, 8000)), n_cb=int(rng.integers(2, 40)), n_gene=int(rng.integers(1, 15)),
        n_umi=int(rng.integers(300, 900)), cb_len=(6, 8) if seed % 3 == 0 else (8, 8), cb_n_rate=0.02 if seed % 2 else 0.0)
    max_ed, p_real = int(rng.integers(0, 8)), float(rng.choice([1e-7, 1e-3, 0.3, 0.9]))
    mb = int(rng.integers(0, 3))
    o = parity.oracle_run(Oracle, dict(merge_kind=4, max_cb_merge_ed=max_ed, max_real_merge_prob=p_real, min_genes_before=mb,
                                       min_genes_after=mb), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_POISSON_SIMPLE, max_cb_merge_edit_distance=max_ed, max_real_merge_prob=p_real,
                            min_genes_before_merge=mb, min_genes_after_merge=mb), cb, umi, gene, aux, side)
    parity.compare(o, c, side)


@pytest.mark.parametrize("seed", range(10))
def test_random_simple_merge(seed):
    """-m without a whitelist on adversarial streams (few UMIs and genes: many exact ties, barcodes with N, variable
    barcode lengths, every edit-distance threshold)."""
    rng = np.random.default_rng(9000 + seed + SEED_OFFSET)
    cb, umi, gene, aux, side = random_stream(
        rng, n=int(rng.integers(300, 8000)), n_cb=int(rng.integers(2, 40)), n_gene=int(rng.integers(1, 15)),
        n_umi=int(rng.integers(2, 60)), cb_len=(6, 8) if seed % 3 == 0 else (8, 8), cb_n_rate=0.02 if seed % 2 else 0.0)
    max_ed, frac = int(rng.integers(0, 10)), float(rng.choice([0.0, 0.05, 0.2, 0.5]))
    mb = int(rng.integers(0, 3))
    o = parity.oracle_run(Oracle, dict(merge_kind=2, max_cb_merge_ed=max_ed, min_merge_fraction=frac, min_genes_before=mb,
                                       min_genes_after=mb), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=max_ed, min_merge_fraction=frac,
                            min_genes_before_merge=mb, min_genes_after_merge=mb), cb, umi, gene, aux, side)
    parity.compare(o, c, side)
