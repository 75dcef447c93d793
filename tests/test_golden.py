"""Committed golden fixtures (tests/golden/digests.json, made by tests/golden_digests.py with the CPU oracle):
the oracle must keep reproducing them (CPU), the HIP path must match them (GPU)."""
import ctypes
import json
import os

import numpy as np
import pytest

import golden_digests as gd

GOLD = json.load(open(gd.PATH))


@pytest.mark.parametrize("case", sorted(gd.CASES))
def test_oracle_reproduces_golden_digests(case):
    assert gd.digests(gd.oracle_view(case)) == GOLD[case]


def hip_view(case):
    from dropest_amd import capi
    import parity
    cb, umi, gene, aux, side, cfg = gd.stream(case)
    kw = dict(min_genes_before_merge=cfg["min_before"], min_genes_after_merge=cfg["min_after"])
    if cfg["merge"] == "real":
        kw.update(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=os.path.join(gd.DATA, cfg["whitelist"]))
    elif cfg["merge"] == "simple":
        kw.update(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=cfg["max_ed"])
    elif cfg["merge"] == "poisson_real":
        kw.update(merge_kind=capi.MERGE_POISSON_REAL, barcodes_kind=capi.BARCODES_CONST, barcodes_file=os.path.join(gd.DATA, cfg["whitelist"]))
    elif cfg["merge"] == "poisson_simple":
        kw.update(merge_kind=capi.MERGE_POISSON_SIMPLE, max_cb_merge_edit_distance=cfg["max_ed"])
    elif cfg["merge"] == "all":
        kw.update(merge_kind=capi.MERGE_ALL, max_cb_merge_edit_distance=cfg["max_ed"])
    if cfg.get("umi") == "directional":
        kw.update(umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL)
        ctypes.CDLL("libc.so.6").srand(1)
    qlen = cfg.get("quality")
    if qlen:
        c = capi.Context(**kw)
        if side:
            c.set_side_strings(side)
        c.push_reads(cb, umi, gene, aux)
        c.set_umi_qualities(gd.qualities(len(cb), qlen))
        c.set_initialized(); c.merge_and_filter()
    else:
        c = parity.gpu_run(kw, cb, umi, gene, aux, side, chunks=2)
    rows = c.cell_rows()
    rows8 = np.stack([rows[k].astype(np.int64) for k in ("is_merged", "is_excluded", "is_real", "n_genes", "requested_genes",
                                                         "requested_umis", "total_reads", "total_umis")], axis=1)
    mc, mg, mu, mr, mm = c.molecules()
    mols = [(int(a), int(b), capi.unpack_code(u, side), int(r), int(m)) for a, b, u, r, m in zip(mc, mg, mu, mr, mm)]
    quality = None
    if qlen:
        quality = {}
        for cell in sorted({m[0] for m in mols}):
            g, u, r, m = c.cell_molecules(cell)
            q = c.cell_molecule_qualities(cell, len(g))
            for j in range(len(g)):
                quality[(cell, int(g[j]), capi.unpack_code(u[j], side))] = [int(x) for x in q[j]]
    return gd.canonical([capi.unpack_code(x, side) for x in rows["barcode"]], rows8, c.filtered_cells(), c.merge_targets(),
                        c.global_counters(), c.count_matrix(filtered=True), c.count_matrix(filtered=False), c.chr_stats(), mols, quality)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(gd.CASES))
def test_hip_path_matches_golden_digests(case):
    got = gd.digests(hip_view(case))
    assert got == GOLD[case], {k: (got[k], GOLD[case][k]) for k in got if got[k] != GOLD[case][k]}
