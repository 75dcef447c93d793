"""Shared helpers of the parity tests: run the same packed stream through the CPU oracle and through the
HIP path (C-ABI) and compare every observable bit-exactly."""
import numpy as np

from dropest_amd import capi


def first_seen_ids(values, skip=None):
    """Dense ids in first-seen order (what StringIndexer / the facade's dictionaries produce)."""
    values = np.asarray(values)
    out = np.full(len(values), skip if skip is not None else 0, dtype=np.uint32)
    mask = np.ones(len(values), bool) if skip is None else values != skip
    uniq, first = np.unique(values[mask], return_index=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(uniq), np.uint32)
    rank[order] = np.arange(len(uniq), dtype=np.uint32)
    out[mask] = rank[np.searchsorted(uniq, values[mask])]
    return out


def canonical_stream(cb, umi, gene, aux):
    """Re-labels gene and chromosome ids to first-seen order, as the C-ABI requires."""
    gene = first_seen_ids(gene, skip=capi.NO_GENE)
    # chromosome ids: first-seen over the reads that reach Stats::inc(chr), i.e. gene-less reads and reads whose
    # mark has the exon or intron bit (CellsDataContainer.cpp:73-78, :312-321); other reads never touch the
    # chromosome dictionary and carry chr 0 (ignored by the path)
    raw_chr = aux & 0xFFFF
    mark = (aux >> 16) & 0xFF
    touches = (gene == capi.NO_GENE) | ((mark & 6) != 0)
    sentinel = np.uint32(0xFFFFFFFF)
    chr_ids = first_seen_ids(np.where(touches, raw_chr, sentinel).astype(np.uint32), skip=sentinel)
    chr_ids = np.where(touches, chr_ids, 0).astype(np.uint32)
    aux = (aux & np.uint32(0xFFFF0000)) | chr_ids
    return cb, umi, gene, aux.astype(np.uint32)


def oracle_run(oracle_cls, cfg_kw, cb, umi, gene, aux, side=()):
    o = oracle_cls(**cfg_kw)
    o.add_packed(cb, umi, gene, aux, side)
    o.set_initialized()
    o.merge_and_filter()
    return o


def gpu_run(ctx_kw, cb, umi, gene, aux, side=(), chunks=1, profile=False):
    c = capi.Context(**ctx_kw)
    if profile:
        c.set_profiling(True)
    if side:
        c.set_side_strings(side)
    n = len(cb)
    bounds = np.linspace(0, n, chunks + 1).astype(np.int64)
    for a, b in zip(bounds[:-1], bounds[1:]):
        c.push_reads(cb[a:b], umi[a:b], gene[a:b], aux[a:b])
    c.set_initialized()
    c.merge_and_filter()
    return c


def compare(o, c, side=(), check_molecules=True, reads_output=False):
    """Asserts that every observable of the container agrees between oracle `o` and HIP context `c`."""
    n_cells = o.n_cells
    assert c.total_cells_number() == n_cells
    rows = c.cell_rows()
    orows = o.cell_rows()   # merged, excluded, real, n_genes, req_genes, req_umis, total_reads, total_umis
    # barcodes in cell-id order
    bc = [capi.unpack_code(x, side) for x in rows["barcode"]]
    obc = [o.cell_barcode(i) for i in range(n_cells)]
    assert bc == obc
    for col, name in ((0, "is_merged"), (1, "is_excluded"), (2, "is_real"), (3, "n_genes"), (4, "requested_genes"),
                      (5, "requested_umis"), (6, "total_reads"), (7, "total_umis")):
        got, want = rows[name].astype(np.int64), orows[:, col]
        if name in ("n_genes", "requested_genes", "requested_umis"):
            # the reference keeps the stale gene maps of MERGED source cells; sizes are compared on the others
            keep = orows[:, 0] == 0
            got, want = got[keep], want[keep]
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, "%s differs at %s: got %s want %s" % (name, bad[:5], got[bad[:5]], want[bad[:5]])
    assert c.real_cells_number() == o.n_real
    assert list(c.filtered_cells()) == list(o.filtered_cells())
    assert list(c.merge_targets()) == list(o.merge_targets())
    assert list(c.global_counters()) == list(o.global_counters())

    for filt in (True, False):
        g, col, v = c.count_matrix(filtered=filt, reads_output=reads_output)
        og, ocol, ov = o.count_matrix(filtered=filt, reads_output=reads_output)
        assert len(g) == len(og), "nnz differs (filtered=%s): %d vs %d" % (filt, len(g), len(og))
        assert np.array_equal(g.astype(np.uint64), og) and np.array_equal(col.astype(np.uint64), ocol)
        assert np.array_equal(v.astype(np.uint64), ov)

    cell, kind, chr_, cnt = c.chr_stats()
    ocell, okind, ochr, ocnt = o.chr_stats()
    assert np.array_equal(cell.astype(np.uint64), ocell) and np.array_equal(kind.astype(np.int32), okind)
    assert np.array_equal(chr_.astype(np.uint64), ochr) and np.array_equal(cnt.astype(np.int64), ocnt)

    if check_molecules:
        mc, mg, mu, mr, mm = c.molecules()
        oc, og, ou, orr, om = o.molecules()
        # the oracle keeps merged source cells' stale molecules; drop them (the reference never reads them again)
        keep = orows[oc.astype(np.int64), 0] == 0
        oc, og, orr, om = oc[keep], og[keep], orr[keep], om[keep]
        ou = [u for u, k in zip(ou, keep) if k]
        got = sorted(zip(mc.tolist(), mg.tolist(), [capi.unpack_code(x, side) for x in mu], mr.tolist(), mm.tolist()))
        want = sorted(zip(oc.tolist(), og.tolist(), ou, orr.tolist(), om.tolist()))
        assert len(got) == len(want), "molecule count %d vs %d" % (len(got), len(want))
        assert got == want
