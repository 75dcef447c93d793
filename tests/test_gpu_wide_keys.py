"""Sort keys wider than 64 bits (cell id + gene + UMI fields): one context refuses them (its molecule key is one 64-bit word);
`dropest_ctx_split` runs the same reads as several shards on the context's own device -- the cell field is the only one that
shrinks when the stream is split by barcode -- and the result must equal the reference's (the oracle has no key at all:
StringIndexer.cpp:10-18 hands out size_t ids).  Plain, with N-UMIs, and with the whitelist CB merge."""
import os

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.multi import ShardGroup, cfg_kwargs
from dropest_amd.synth import SynthStream, inject_n
from oracle import Oracle

import parity
import test_gpu_stress as ts

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")


def split_run(arrays, cfg_kw, side=(), chunks=2):
    """One context first (must refuse), then the split; returns the result of shard 0 and the number of parts used."""
    c = capi.Context(**cfg_kw)
    if side:
        c.set_side_strings(side)
    n = len(arrays[0])
    b = np.linspace(0, n, chunks + 1).astype(np.int64)
    for lo, hi in zip(b[:-1], b[1:]):
        c.push_reads(*[a[lo:hi] for a in arrays])
    with pytest.raises(capi.DropestError) as e:
        c.set_initialized()
    assert e.value.status == 4 and "sort key needs" in str(e.value)
    cell, gene, umi = c.key_width()
    assert cell + gene + umi > 64
    parts = 1 << (cell + gene + umi - 64)
    while True:
        assert parts <= 64
        g = ShardGroup.split(c, parts)
        try:
            g.step()
            break
        except capi.DropestError as err:         # uneven owners: one more bit
            assert err.status == 4 and "sort key needs" in str(err)
            g.close()
            parts *= 2
    g.step()                                      # and once more over the same reads (buffer reuse)
    s0 = g.shards[0]
    out = {"cm": [x.copy() for x in s0.matrix(True)], "raw": [x.copy() for x in s0.matrix(False)], "merged": s0.merged_barcodes()}
    g.close()
    return out, parts


def check_against_oracle(got, o, side=()):
    orows = o.cell_rows()
    for filt, name in ((True, "cm"), (False, "raw")):
        og, ocol, ov = o.count_matrix(filtered=filt)
        p, i, x, bc = got[name]
        col = np.repeat(np.arange(len(p) - 1, dtype=np.uint64), np.diff(p.astype(np.int64)))
        assert len(i) == len(og), name
        assert np.array_equal(i.astype(np.uint64), og) and np.array_equal(col, ocol) and np.array_equal(x.astype(np.uint64), ov), name
        cells = list(o.filtered_cells()) if filt else [k for k in range(o.n_cells) if orows[k, 2]]
        assert [capi.unpack_code(b, side) for b in bc] == [o.cell_barcode(int(k)) for k in cells], name
    mt = list(o.merge_targets())
    want = {o.cell_barcode(k): o.cell_barcode(t) for k, t in enumerate(mt) if t != k}
    have = {capi.unpack_code(a, side): capi.unpack_code(b, side) for a, b in zip(*got["merged"])}
    assert have == want
    return want


@pytest.mark.parametrize("n_rate", [0.0, 0.02])
def test_wide_key_random_stream(n_rate):
    rng = np.random.default_rng(4100 + int(n_rate * 1000))
    # 20-base UMIs (40 bits, 41 with escapes) + 20 000 gene ids (15 bits) + 1 500 barcodes (11 bits)
    cb, umi, gene, aux, side = ts.random_stream(rng, n=60_000, n_cb=1500, n_gene=20_000, n_umi=3000, cb_len=(14, 14), umi_len=(20, 20),
                                                n_rate=n_rate, p_nogene=0.05)
    kw = dict(min_genes_before_merge=2, min_genes_after_merge=4)
    got, parts = split_run((cb, umi, gene, aux), kw, side)
    assert parts >= 4
    o = parity.oracle_run(Oracle, dict(min_genes_before=2, min_genes_after=4), cb, umi, gene, aux, side)
    check_against_oracle(got, o, side)
    assert len(got["cm"][3]) > 100


def test_wide_key_whitelist_merge():
    s = SynthStream(n_reads=200_000, n_cells=40, n_genes=2000, umi_len=21, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 2e-3, 5, 21)
    wl = os.path.join(DATA, "10x_aug_2016_split")
    kw = cfg_kwargs({"min_before": 3, "min_after": 20, "merge": {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": wl}})
    got, parts = split_run((cb, umi, gene, aux), kw, side)
    o = parity.oracle_run(Oracle, dict(merge_kind=1, barcodes_kind=capi.BARCODES_CONST, barcodes_file=wl, min_genes_before=3,
                                       min_genes_after=20), cb, umi, gene, aux, side)
    want = check_against_oracle(got, o, side)
    assert len(want) > 20 and parts >= 2


def test_wide_key_merge_all_with_umi_qualities():
    """A split run carries what a sharded run carries (round 4): a merge without a whitelist (merge_type = all; the simple merges' UMI-gene
    index needs UMI-gene + cell + shard within 63 bits and says so) and the UMI quality strings of the context, copied to the shards with
    their reads.  Matrices, merge targets and the quality sums of every molecule against the oracle."""
    s = SynthStream(n_reads=120_000, whitelist="10x_aug_2016_split", n_cells=30, n_genes=1500, umi_len=21, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    qual = np.random.default_rng(5).integers(33, 75, size=(len(cb), 7), dtype=np.uint8)
    okw = dict(merge_kind=5, max_cb_merge_ed=2, min_genes_before=3, min_genes_after=10)
    kw = dict(merge_kind=capi.MERGE_ALL, max_cb_merge_edit_distance=2, min_genes_before_merge=3, min_genes_after_merge=10)
    o = Oracle(**okw)
    o.add_packed_q(cb, umi, gene, aux, qual)
    o.set_initialized(); o.merge_and_filter()
    c = capi.Context(**kw)
    c.push_reads(cb, umi, gene, aux)
    c.set_umi_qualities(qual)
    with pytest.raises(capi.DropestError) as e:
        c.set_initialized()
    assert "sort key needs" in str(e.value)
    cell, gene_b, umi_b = c.key_width()
    g = ShardGroup.split(c, 1 << max(1, cell + gene_b + umi_b - 64 + 1))
    g.step(); g.step()
    s0 = g.shards[0]
    got = {"cm": [x.copy() for x in s0.matrix(True)], "raw": [x.copy() for x in s0.matrix(False)], "merged": s0.merged_barcodes()}
    want = check_against_oracle(got, o)
    assert len(want) > 20
    oc, og, ou, orr, om = o.molecules()
    oq = o.molecule_qualities(len(oc), 7)
    merged = o.cell_rows()[:, 0] != 0
    want_q = {(o.cell_barcode(int(oc[i])), int(og[i]), ou[i]): (int(orr[i]), tuple(int(x) for x in oq[i])) for i in range(len(oc)) if not merged[int(oc[i])]}
    got_q = {}
    for sh in g.shards:
        rows = sh.ctx.cell_rows()
        for cell_id in np.flatnonzero(rows["is_real"].astype(bool) & ~rows["is_merged"].astype(bool)):
            gg, uu, rr, mm = sh.ctx.cell_molecules(int(cell_id))
            q = sh.ctx.cell_molecule_qualities(int(cell_id), len(gg))
            for j in range(len(gg)):
                got_q[(capi.unpack_code(int(rows["barcode"][cell_id])), int(gg[j]), capi.unpack_code(uu[j]))] = (int(rr[j]), tuple(int(x) for x in q[j]))
    real_keys = {k for k in want_q if k in got_q}
    assert len(got_q) > 5000 and len(real_keys) == len(got_q)
    bad = [k for k in real_keys if got_q[k] != want_q[k]]
    assert not bad, (len(bad), bad[0], got_q[bad[0]], want_q[bad[0]])
    g.close()


def test_gene_and_umi_alone_too_wide_take_the_umi_dictionary():
    """Gene + UMI fields that alone fill the key: one context keys the UMIs by their rank in a dictionary (round 5, tests/test_gpu_umi_dict.py);
    the shards of a split run gather ONE dictionary over all of them (round 6, shard_run.h: global_umi_dictionary) -- they used to refuse."""
    P = capi.pack_seq
    n = 5
    cb = np.array([P("ACGT" * 7 + "AC" + "ACGT"[i % 4]) for i in range(n)], np.uint64)
    umi = np.array([P("TTGCA" * 6)] * n, np.uint64)            # 60 bits
    gene = np.array([1_000_000] * n, np.uint32)                # 20 bits
    c = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    c.push_reads(cb, umi, gene, np.full(n, 2 << 16, np.uint32))
    c.set_initialized(); c.merge_and_filter()
    assert c.key_width()[2] <= 9 and c.total_cells_number() == 4
    mc, mg, mu, mr, mm = c.molecules()
    assert sorted(mr.tolist()) == [1, 1, 1, 2] and set(capi.unpack_code(x) for x in mu) == {"TTGCA" * 6} and set(mg.tolist()) == {1_000_000}
    c2 = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    c2.push_reads(cb, umi, gene, np.full(n, 2 << 16, np.uint32))
    g = ShardGroup.split(c2, 2)
    g.step()
    s0 = g.shards[0]
    assert s0.phase_stats().get("umi_dictionary", {"steps": 0})["steps"] == 1
    p, i, x, b = s0.matrix(False)
    assert len(p) - 1 == 4 and sorted(x.tolist()) == [1, 1, 1, 1]          # four cells, one molecule each (the barcode seen twice: two reads, one molecule)
    g.close()


@pytest.mark.skipif(not os.environ.get("DROPEST_WIDE_FULL"), reason="2^25 barcodes: ~3 minutes, most of it the oracle; DROPEST_WIDE_FULL=1 runs it "
                    "(last run: profiles/r02b_wide_key_2_25.log)")
def test_wide_key_at_2_25_barcodes():
    """40 000 genes (16 bits), 12-base UMIs (24 bits), >= 2^25 distinct barcodes (26 bits) = 66 bits."""
    n_cb = (1 << 25) + 1000
    rng = np.random.default_rng(99)
    base = rng.choice(1 << 32, size=n_cb, replace=False).astype(np.uint64) | np.uint64(1 << 32)      # 16-base barcodes
    hot = base[:2000]
    n = n_cb + 6_000_000
    cb = np.concatenate([base, hot[rng.integers(0, len(hot), n - n_cb)]])
    rng.shuffle(cb)
    umi = rng.integers(0, 1 << 24, n).astype(np.uint64) | np.uint64(1 << 24)
    gene = rng.integers(0, 40_000, n).astype(np.uint32)
    aux = (rng.integers(0, 20, n) | (2 << 16)).astype(np.uint32)
    cb, umi, gene, aux = parity.canonical_stream(cb, umi, gene, aux)
    kw = dict(min_genes_before_merge=20, min_genes_after_merge=50)
    got, parts = split_run((cb, umi, gene, aux), kw)
    assert parts >= 4
    o = parity.oracle_run(Oracle, dict(min_genes_before=20, min_genes_after=50), cb, umi, gene, aux)
    check_against_oracle(got, o)
    assert len(got["cm"][3]) > 1000
