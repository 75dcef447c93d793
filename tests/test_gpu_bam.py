"""BAM ingest without BamTools (dropest_amd/csrc/host/bam_ingest.cpp): BAM -> BamController -> CellsDataContainer ->
ResultsPrinter, compared with the CPU oracle fed with the same reads through add_record (strings)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.build import build_facade
from dropest_amd.synth import SynthStream
from oracle import Oracle

import bam_writer as bw
import rds_reader as rr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tests", "cpp", "bam_to_counts")


def _reads(n_reads, seed_cells=20):
    s = SynthStream(n_reads=n_reads, n_cells=seed_cells, n_genes=300, umi_len=8, permille_intergenic=80, permille_intron=100)
    cb, umi, gene, aux = s.generate_host()
    out = []
    for i in range(n_reads):
        g = None if gene[i] == capi.NO_GENE else "ENSG%05d" % gene[i]
        mark = int(aux[i] >> 16) & 7
        out.append((capi.unpack_code(cb[i]), capi.unpack_code(umi[i]), g, "chr%d" % (int(aux[i]) & 0xFFFF), mark))
    return out


def _oracle(reads, min_before, min_after, wl=None):
    kw = dict(merge_kind=1, barcodes_kind=1, barcodes_file=wl) if wl else {}
    o = Oracle(min_genes_before=min_before, min_genes_after=min_after, **kw)
    for cb, umi, g, chr_, mark in reads:
        o.add_record(cb, umi, g or "", chr_, mark)
    o.set_initialized(); o.merge_and_filter()
    gi, ci, v = o.count_matrix(filtered=True)
    cols = [o.cell_barcode(int(k)) for k in o.filtered_cells()]
    return {(o.gene_name(int(g)), cols[int(c)]): int(x) for g, c, x in zip(gi, ci, v)}, cols


def _run(tmp_path, mode, bams, min_before, min_after, wl="-", threads=3, env=None):
    build_facade()
    os.makedirs(str(tmp_path), exist_ok=True)
    out = str(tmp_path / "res")
    env = dict(env or {})
    if env.get("DROPEST_BAM_DEVICE"):
        env["DROPEST_BAM_TRACE"] = "1"
    res = subprocess.run([TOOL, out, mode, str(min_before), str(min_after), wl, str(threads)] + bams, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, **env))
    assert res.returncode == 0, res.stdout + res.stderr
    if env.get("DROPEST_BAM_DEVICE"):      # the device path really ran (it hands a file it cannot do to the host reader without a word otherwise)
        assert res.stderr.count("[bam] device path:") >= 2 * len(bams), res.stderr
    import re
    rep = re.findall(r"\((\d+) guesses repaired\)", res.stderr)
    stats = json.loads(res.stdout.strip().splitlines()[-1])
    stats["_guesses_repaired"] = sum(int(x) for x in rep)
    d = rr.read_rds(out + ".rds")
    cm, genes, cells = rr.dgcmatrix_to_dense(d["cm"])
    got = {(genes[r], cells[c]): int(cm[r, c]) for r, c in zip(*np.nonzero(cm))}
    return got, cells, stats, d


def test_filled_bam_tags(tmp_path):
    """-f: CB / UB / GX tags + a read-type tag, with every kind of record the controller must skip or count."""
    reads = _reads(40_000)
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    recs, kept = [], []
    rng = np.random.default_rng(1)
    for i, (cb, umi, g, chr_, mark) in enumerate(reads):
        rid = int(chr_[3:])
        # the read-type tag decides the mark when a gene tag is present (ReadParamsParser.cpp:67-90)
        if g is None:
            tags, m = [("CB", "Z", cb), ("UB", "Z", umi)], 1                      # no gene tag: gene "", NOT_ANNOTATED
        else:
            code, m = (("N", 4) if mark & 4 else ("I", 1) if mark == 1 else ("E", 2))
            tags = [("xf", "i", 25), ("CB", "Z", cb), ("fx", "B", [1, -2, 3]), ("UB", "Z", umi), ("GX", "Z", g), ("RE", "A", code)]
        kind = rng.integers(0, 40)
        if kind == 0:
            recs.append(bw.record(rid, i, "r%d" % i, flag=4, tags=tags)); continue          # unmapped
        if kind == 1:
            recs.append(bw.record(rid, i, "r%d" % i, flag=0x100, tags=tags)); continue      # secondary
        if kind == 2:
            recs.append(bw.record(rid, i, "r%d" % i, tags=[t for t in tags if t[0] != "CB"])); continue   # can't parse
        if kind == 3:
            recs.append(bw.record(-1, i, "r%d" % i, tags=tags)); continue                   # unknown chromosome id
        recs.append(bw.record(rid, i, "r%d" % i, flag=16 if i % 2 else 0, tags=tags))
        kept.append((cb, umi, g, chr_, m))
    half = len(recs) // 2
    b1, b2 = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    bw.write_bam(b1, refs, recs[:half], block=40_000)          # records straddle block boundaries
    bw.write_bam(b2, refs, recs[half:], block=0xFF00)
    got, cells, stats, d = _run(tmp_path, "filled", [b1, b2], 5, 10)
    want, cols = _oracle(kept, 5, 10)
    assert cells == cols and got == want and len(want) > 500
    assert stats["saved"] == len(kept) and stats["low_quality"] == 0
    n_skipped = sum(1 for r in recs) - len(kept)
    assert 0 < stats["cant_parse"] < n_skipped                 # missing CB tag + unknown chromosome; unmapped / secondary are not counted
    chr_frames = d["reads_per_chr_per_cells"]
    assert len(chr_frames["Intron"].value) > 0 and len(chr_frames["Intergenic"].value) > 0
    # the record-by-record path (add_record per read on the caller's thread) and the bulk path (workers write packed records, the
    # caller resolves only what is new to the dictionaries) give the same container; several worker counts
    # ... and the device path (DROPEST_BAM_DEVICE: blocks inflated, records found and tags walked on the GPU, csrc/k_inflate.h + k_bamparse.h),
    # with one window per megabyte and with windows of whole files
    for env, threads in (({"DROPEST_BAM_RECORD_BY_RECORD": "1"}, 3), ({}, 1), ({}, 7), ({"DROPEST_BAM_DEVICE": "1"}, 3),
                         ({"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_DEVICE_WINDOW_MB": "1"}, 6)):
        got2, cells2, stats2, _ = _run(tmp_path / ("again%d%d" % (threads, len(env))), "filled", [b1, b2], 5, 10, threads=threads, env=env)
        assert cells2 == cells and got2 == got
        assert {k: stats2[k] for k in ("total_reads", "cant_parse", "low_quality", "saved")} == {k: stats[k] for k in ("total_reads", "cant_parse", "low_quality", "saved")}


@pytest.mark.parametrize("block", [3, 61, 997, 4093])
def test_record_boundaries_found_by_the_loader(tmp_path, block):
    """The reader inflates 512 BGZF blocks per batch and its workers hand the position of the next record from block to block
    (bam_ingest.cpp: loader_walks).  Files of thousands of tiny blocks put every case in front of it: records and their 4-byte length
    fields cut by block and by batch boundaries, blocks shorter than a length field, batches without a complete record's start.
    Same container as when the caller's thread walks the records (DROPEST_BAM_CALLER_WALKS), same as the oracle."""
    reads = _reads(12_000 if block < 100 else 30_000)
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    recs, kept = [], []
    for i, (cb, umi, g, chr_, mark) in enumerate(reads):
        tags = [("CB", "Z", cb), ("UB", "Z", umi)] + ([("GX", "Z", g)] if g else [])
        recs.append(bw.record(int(chr_[3:]), i, "read%d" % i, seq="ACGT" * (1 + i % 9), tags=tags))   # records of different lengths
        kept.append((cb, umi, g, chr_, 1 if g is None else 2))
    bam = str(tmp_path / "tiny_blocks.bam")
    bw.write_bam(bam, refs, recs, block=block)
    assert os.path.getsize(bam) // max(1, (block + 26)) > 1100 or block > 997          # well over two batches of 512 blocks (the largest block size: 3+)
    got, cells, stats, _ = _run(tmp_path, "filled", [bam], 3, 5, threads=5)
    want, cols = _oracle(kept, 3, 5)
    assert stats["saved"] == len(kept) and cells == cols and got == want and len(want) > 200
    got2, cells2, stats2, _ = _run(tmp_path / "caller", "filled", [bam], 3, 5, threads=5, env={"DROPEST_BAM_CALLER_WALKS": "1"})
    assert cells2 == cells and got2 == got and stats2["saved"] == stats["saved"]
    got3, cells3, _, _ = _run(tmp_path / "zlib", "filled", [bam], 3, 5, threads=2, env={"DROPEST_BAM_ZLIB": "1"})
    assert cells3 == cells and got3 == got
    # the device finds the chain of records by guessing per 16 KB segment and checking the guesses (k_bamparse.h): same container
    got4, cells4, stats4, _ = _run(tmp_path / "device", "filled", [bam], 3, 5, threads=4, env={"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_TRACE": "1"})
    assert cells4 == cells and got4 == got and stats4["saved"] == stats["saved"]
    # ... whatever the guesses were: every third one spoiled by a byte, every seventh withheld (a test switch of the library) -- the host's check
    # walks those segments again from the true place
    got5, cells5, stats5, _ = _run(tmp_path / "spoiled", "filled", [bam], 3, 5, threads=4, env={"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_TEST_SPOIL_GUESSES": "1"})
    assert cells5 == cells and got5 == got and stats5["saved"] == stats["saved"]
    if block > 100:
        assert stats5["_guesses_repaired"] > 0


def test_gene_names_that_share_a_hash_do_not_share_an_index(tmp_path):
    """The device finds a record's gene by the FNV-1a value of its name in a copy of the host's dictionary (k_bamparse.h) and, since round 6,
    confirms the hit against the name's bytes (dropest_bam_decoder_set_gene_names).  With the hash cut to 3 bits (a test switch of both sides)
    nearly every name collides with another: the container must be the host reader's all the same -- and without the names on the device it is not."""
    reads = _reads(20_000)
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    recs = [bw.record(int(chr_[3:]), i, "r%d" % i, tags=[("CB", "Z", cb), ("UB", "Z", umi)] + ([("GX", "Z", g)] if g else [])) for i, (cb, umi, g, chr_, mark) in enumerate(reads)]
    bam = str(tmp_path / "genes.bam")
    bw.write_bam(bam, refs, recs, block=20_000)
    host = _run(tmp_path / "host", "filled", [bam], 3, 5, threads=4)
    assert len({g for g, _ in host[0]}) > 50
    for bits in ("3", "1"):
        dev = _run(tmp_path / ("dev" + bits), "filled", [bam], 3, 5, threads=4, env={"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_DEVICE_WINDOW_MB": "1", "DROPEST_BAM_TEST_GENE_HASH_BITS": bits})
        assert dev[1] == host[1] and dev[0] == host[0] and dev[2]["saved"] == host[2]["saved"]
        host_weak = _run(tmp_path / ("hostweak" + bits), "filled", [bam], 3, 5, threads=4, env={"DROPEST_BAM_TEST_GENE_HASH_BITS": bits})
        assert host_weak[1] == host[1] and host_weak[0] == host[0]


def test_read_name_encoding_and_whitelist_merge(tmp_path):
    """Without -f the barcodes come from the read name "id!CB#UMI" (ReadParamsParser.cpp:20-33); with -m + whitelist."""
    s = SynthStream(n_reads=40_000, n_cells=20, n_genes=300, umi_len=8, permille_neighbour=150)
    cb, umi, gene, aux = s.generate_host()
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    recs, kept = [], []
    for i in range(len(cb)):
        c, u = capi.unpack_code(cb[i]), capi.unpack_code(umi[i])
        g = None if gene[i] == capi.NO_GENE else "G%d" % gene[i]
        chr_ = "chr%d" % (int(aux[i]) & 0xFFFF)
        tags = [("GX", "Z", g)] if g else []
        recs.append(bw.record(int(aux[i]) & 0xFFFF, i, "@read%d!%s#%s" % (i, c, u), tags=tags))
        kept.append((c, u, g, chr_, 2 if g else 1))
    recs.insert(100, bw.record(0, 0, "no_separator_here", tags=[("GX", "Z", "G1")]))
    bam = str(tmp_path / "n.bam")
    bw.write_bam(bam, refs, recs)
    wl = os.path.join(ROOT, "dropest_amd", "data", "barcodes", "10x_aug_2016_split")
    got, cells, stats, d = _run(tmp_path, "name", [bam], 3, 10, wl=wl, threads=1)
    want, cols = _oracle(kept, 3, 10, wl=wl)
    assert cells == cols and got == want
    assert stats["cant_parse"] == 1 and stats["saved"] == len(kept)
    assert len(d["merge_targets"].value) > 10


def _plain(v):
    """An RObject tree (rds_reader) as plain python, for equality: (kind, value, attributes)."""
    if not isinstance(v, rr.RObject):
        if isinstance(v, np.ndarray):
            return [None if (isinstance(x, float) and x != x) else x for x in v.tolist()]
        if isinstance(v, (list, tuple)):
            return [_plain(x) for x in v]
        return v
    return (v.kind, _plain(v.value), {k: _plain(x) for k, x in sorted(v.attributes.items())})


@pytest.mark.parametrize("wl,qual", [(False, False), (True, False), (False, True), (True, True)])
def test_sharded_container_writes_the_same_rds(tmp_path, wl, qual):
    """ResultsPrinter::save_results of a container sharded over three shards (one GPU) = the .rds of one container: both
    matrices with the reference's row order, per-chromosome frames, mean reads per UMI, saturation info, aligned / requested
    counts per cell, reads_per_umi_per_cell; merge_targets as a set (the order of that list is the reference's hash order).
    With UQ tags (qual) the mean qualities of reads_per_umi_per_cell must agree too: every shard gets the strings of its own
    range of reads, and under the whitelist merge (wl) the quality sums follow the molecules across shards."""
    qrng = np.random.default_rng(31)
    s = SynthStream(n_reads=60_000, n_cells=25, n_genes=300, umi_len=8, permille_neighbour=150, permille_intron=100)
    cb, umi, gene, aux = s.generate_host()
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    recs = []
    for i in range(len(cb)):
        c, u = capi.unpack_code(cb[i]), capi.unpack_code(umi[i])
        if i % 97 == 0:
            u = u[:3] + "N" + u[4:]
        g = None if gene[i] == capi.NO_GENE else "G%d" % gene[i]
        mark = int(aux[i] >> 16) & 7
        tags = [("CB", "Z", c), ("UB", "Z", u)]
        if qual:
            tags.append(("UQ", "Z", "".join(chr(int(x)) for x in qrng.integers(35, 74, 8))))
        if g:
            tags += [("GX", "Z", g), ("RE", "A", "N" if mark & 4 else "E")]
        recs.append(bw.record(int(aux[i]) & 0xFFFF, i, "r%d" % i, tags=tags))
    bam = str(tmp_path / "s.bam")
    bw.write_bam(bam, refs, recs)
    wlf = os.path.join(ROOT, "dropest_amd", "data", "barcodes", "10x_aug_2016_split") if wl else "-"
    env = {"DROPEST_RPUPC": "1"}
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    _, cells1, st1, d1 = _run(tmp_path / "a", "filled", [bam], 3, 10, wl=wlf, threads=2, env=env)
    _, cells3, st3, d3 = _run(tmp_path / "b", "filled", [bam], 3, 10, wl=wlf, threads=2,
                              env=dict(env, DROPEST_DEVICES="0,0,0"))
    assert cells1 == cells3 and len(cells1) >= 20 and st1["real_cells"] == st3["real_cells"]
    assert d1.names == d3.names and "reads_per_umi_per_cell" in d1.names
    for key in d1.names:
        if key == "merge_targets":
            a, b = ({n: _plain(x) for n, x in zip(d.names or [], d.value)} for d in (d1[key], d3[key]))
            assert a == b and (len(a) > 10 or not wl)
        else:
            assert _plain(d1[key]) == _plain(d3[key]), key
    per_gene = d3["reads_per_umi_per_cell"]["reads_per_umi"].value
    with_quality = sum(1 for g in per_gene for e in g.value if len(e.value[1].value) == 8)
    assert (with_quality > 1000) if qual else (with_quality == 0)


def test_bad_files(tmp_path):
    build_facade()
    p = str(tmp_path / "x.bam")
    open(p, "wb").write(b"not a bam")
    for bam in (p, str(tmp_path / "missing.bam")):
        res = subprocess.run([TOOL, str(tmp_path / "o"), "filled", "1", "1", "-", "1", bam], capture_output=True, text=True)
        assert res.returncode == 1 and "BAM" in res.stderr
        res = subprocess.run([TOOL, str(tmp_path / "o"), "filled", "1", "1", "-", "1", bam], capture_output=True, text=True, env=dict(os.environ, DROPEST_BAM_DEVICE="1"))
        assert res.returncode == 1 and "BAM" in res.stderr
    # damage inside a file: a record whose length field says 7 bytes, a block with a flipped payload byte, a file cut in the middle of a block and in
    # the middle of a record -- the host reader and the device path both refuse them (no result is written)
    reads = _reads(6000)
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    recs = [bw.record(int(c[3:]), i, "r%d" % i, tags=[("CB", "Z", cb), ("UB", "Z", umi)] + ([("GX", "Z", g)] if g else [])) for i, (cb, umi, g, c, m) in enumerate(reads)]
    good = str(tmp_path / "good.bam")
    bw.write_bam(good, refs, recs, block=20_000)
    blob = open(good, "rb").read()
    short = list(recs); short[3000] = struct.pack("<I", 7) + short[3000][4:]
    bad_len = str(tmp_path / "bad_len.bam")
    bw.write_bam(bad_len, refs, short, block=20_000)
    flipped = bytearray(blob); flipped[len(blob) // 2] ^= 0x40
    cut_block = blob[: len(blob) // 2]
    whole_blocks = 0
    at = 0
    while at + 18 <= len(blob) * 2 // 3:
        at += struct.unpack("<H", blob[at + 16:at + 18])[0] + 1
    cut_record = blob[:at]                                       # whole blocks, but the stream ends inside a record (and without the EOF block)
    for name, data in (("flipped", bytes(flipped)), ("cut_block", cut_block), ("cut_record", cut_record)):
        open(str(tmp_path / (name + ".bam")), "wb").write(data)
    for name in ("bad_len", "flipped", "cut_block", "cut_record"):
        for env in ({}, {"DROPEST_BAM_DEVICE": "1"}):
            res = subprocess.run([TOOL, str(tmp_path / "o2"), "filled", "1", "1", "-", "2", str(tmp_path / (name + ".bam"))], capture_output=True, text=True,
                                 env=dict(os.environ, **env))
            assert res.returncode == 1 and ("BAM" in res.stderr or "BGZF" in res.stderr), (name, env, res.stderr)


def test_genes_from_a_gtf_annotation(tmp_path):
    """-g: no gene tags in the BAM, genes and marks come from the alignment's two end points in a GTF
    (ReadParamsParser::get_gene_from_reference); spliced alignments (N in the CIGAR) reach the next exon, soft clips
    and insertions do not move the end.  The oracle gets the reads annotated by its own restatement of the reference."""
    import gzip
    from oracle import binding as ob
    rng = np.random.default_rng(6)
    lines = []
    for chr_ in ("chr1", "chr2"):
        pos = 1000
        for g in range(60):
            for x in range(int(rng.integers(1, 5))):
                ln = int(rng.integers(60, 400))
                lines.append('%s\tsrc\texon\t%d\t%d\t.\t+\t.\tgene_id "G%s_%d"; transcript_id "T%s_%d";' % (chr_, pos + 1, pos + ln, chr_, g, chr_, g))
                pos += ln + int(rng.integers(50, 600))
            pos += int(rng.integers(0, 2000))
    gtf = str(tmp_path / "ann.gtf.gz")
    with gzip.open(gtf, "wt") as f:
        f.write("\n".join(lines) + "\n")
    ann = ob.GeneAnnotationOracle(gtf)
    s = SynthStream(n_reads=30_000, n_cells=15, n_genes=10, umi_len=8)
    cb, umi, _, _ = s.generate_host()
    refs = [("chr1", 1_000_000), ("chr2", 1_000_000), ("chrUn", 1000)]
    recs, kept, n_unknown = [], [], 0
    for i in range(len(cb)):
        c, u = capi.unpack_code(cb[i]), capi.unpack_code(umi[i])
        rid = int(rng.integers(0, 3)) if i % 50 == 0 else int(rng.integers(0, 2))
        p = int(rng.integers(0, 120_000))
        cigar = [[(40, "M")], [(5, "S"), (35, "M")], [(15, "M"), (int(rng.integers(50, 900)), "N"), (25, "M")],
                 [(20, "M"), (3, "I"), (10, "M"), (2, "D"), (7, "M")]][int(rng.integers(0, 4))]
        end = p + sum(ln for ln, op in cigar if op in "MDN=X")
        recs.append(bw.record(rid, p, "r%d" % i, tags=[("CB", "Z", c), ("UB", "Z", u)], cigar=cigar))
        got = ann.gene_for_read(refs[rid][0], p, end)
        if got is None:
            n_unknown += 1
            continue
        kept.append((c, u, got[0] or None, refs[rid][0], got[1]))
    bam = str(tmp_path / "g.bam")
    bw.write_bam(bam, refs, recs, block=30_000)
    os.environ["DROPEST_GTF"] = gtf
    try:
        got, cells, stats, d = _run(tmp_path, "filled", [bam], 2, 3)
        # the same with the BAM inflated, walked and annotated on the device (k_bamparse.h -> annotation_api.hip -> bam_resolve_annotated)
        got_d, cells_d, stats_d, _ = _run(tmp_path / "device", "filled", [bam], 2, 3, env={"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_DEVICE_WINDOW_MB": "1", "DROPEST_BAM_TRACE": "1"})
    finally:
        del os.environ["DROPEST_GTF"]
    want, cols = _oracle(kept, 2, 3)
    assert cells == cols and got == want and len(want) > 100
    assert stats["cant_parse"] == n_unknown > 0 and stats["saved"] == len(kept)
    assert cells_d == cells and got_d == got
    assert {k: stats_d[k] for k in ("total_reads", "cant_parse", "low_quality", "saved")} == {k: stats[k] for k in ("total_reads", "cant_parse", "low_quality", "saved")}
    assert sum(1 for k in kept if k[4] & 1) > 100 and sum(1 for k in kept if k[2] is None) > 1000    # half-annotated and intergenic reads


def test_umi_quality_tags_reach_reads_per_umi_per_cell(tmp_path):
    """UQ tags -> UMI::add_read sums -> UMI::mean_quality in reads_per_umi_per_cell (ResultsPrinter.cpp:261-314), against the oracle."""
    reads = _reads(30_000, seed_cells=12)
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    rng = np.random.default_rng(9)
    recs, o = [], Oracle(min_genes_before=5, min_genes_after=10)
    for i, (cb, umi, g, chr_, mark) in enumerate(reads):
        q = "".join(chr(int(x)) for x in rng.integers(35, 74, 8))
        if g is None:
            tags, m = [("CB", "Z", cb), ("UB", "Z", umi), ("UQ", "Z", q)], 1
        else:
            code, m = (("N", 4) if mark & 4 else ("I", 1) if mark == 1 else ("E", 2))
            tags = [("CB", "Z", cb), ("UB", "Z", umi), ("UQ", "Z", q), ("GX", "Z", g), ("RE", "A", code)]
        recs.append(bw.record(int(chr_[3:]), i, "r%d" % i, tags=tags))
        o.add_record(cb, umi, g or "", chr_, m, umi_qual=q)
    o.set_initialized(); o.merge_and_filter()
    b = str(tmp_path / "q.bam")
    bw.write_bam(b, refs, recs)
    got, cells, stats, d = _run(tmp_path, "filled", [b], 5, 10, env={"DROPEST_RPUPC": "1"})
    assert stats["saved"] == len(reads)
    oc, og, ou, orr, om = o.molecules()
    oq = o.molecule_qualities(len(oc), 8)
    want = {}
    filtered = {int(x) for x in o.filtered_cells()}
    for i in range(len(oc)):
        if int(oc[i]) in filtered and om[i] in (2, 3, 6, 7):                      # requested UMIs (default -L eEBA: marks with an exon bit)
            want[(o.cell_barcode(int(oc[i])), o.gene_name(int(og[i])), ou[i])] = (int(orr[i]), [float((int(s) - 33) // int(orr[i])) for s in oq[i]])
    rp = d["reads_per_umi_per_cell"]
    cells_l, genes_l = rp["cells"].value, rp["genes"].value
    seen = {}
    for ci, gi, per_gene in zip(rp["cell_indexes"].value, rp["gene_indexes"].value, rp["reads_per_umi"].value):
        for name, entry in zip(per_gene.names, per_gene.value):
            seen[(cells_l[int(ci)], genes_l[int(gi)], name)] = (int(entry.value[0].value[0]), [float(x) for x in entry.value[1].value])
    assert len(seen) > 1000 and seen == want


def test_read_parameter_files(tmp_path):
    """-r: droptag's read-parameter files (ReadMapParamsParser.cpp): barcodes, UMIs and qualities come from gzip text
    rows keyed by the read name; every name serves once (a second primary alignment of the same name cannot be parsed),
    unknown names cannot be parsed, rows below min_barcode_quality are low-quality reads, malformed rows are skipped."""
    import gzip
    reads = _reads(30_000, seed_cells=12)
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    rng = np.random.default_rng(17)
    recs, rows, kept = [], [], []
    n_missing = n_low = n_dup = 0
    for i, (cb, umi, g, chr_, mark) in enumerate(reads):
        if g is None:
            tags, m = [], 1
        else:
            code, m = (("N", 4) if mark & 4 else ("I", 1) if mark == 1 else ("E", 2))
            tags = [("GX", "Z", g), ("RE", "A", code)]
        name = "read%d" % i
        recs.append(bw.record(int(chr_[3:]), i, name, tags=tags))
        kind = int(rng.integers(0, 30))
        if kind == 0:
            n_missing += 1; continue                                              # no row for this read
        low = kind == 1
        qcb = "".join(chr(int(x)) for x in rng.integers(53, 74, len(cb)))
        qumi = "".join(chr(int(x)) for x in rng.integers(53, 74, len(umi)))
        if low:
            qumi = qumi[:2] + chr(33 + 5) + qumi[3:]; n_low += 1
        rows.append("%s%s %s %s %s %s" % ("@" if i % 2 else "", name, cb, umi, qcb, qumi))
        if kind == 2:                                                             # the same name aligned twice as primary
            recs.append(bw.record(int(chr_[3:]), i, name, tags=tags)); n_dup += 1
        if kind == 3:
            rows.append("%s %s %s %s %s" % (name, "AAAA", "CCCC", "IIII", "IIII"))  # a repeated name: the first row stays
        if not low:
            kept.append((cb, umi, g, chr_, m, qumi))
    rows.insert(5, "broken row without enough fields")
    rows.insert(9, "emptycb  ACGT II II")                                          # empty barcode: ReadParameters throws, row skipped
    half = len(rows) // 2
    f1, f2 = str(tmp_path / "p1.gz"), str(tmp_path / "p2.gz")
    for path, part in ((f1, rows[:half]), (f2, rows[half:])):
        with gzip.open(path, "wt") as f:
            f.write("\n".join(part) + "\n")
    b = str(tmp_path / "r.bam")
    bw.write_bam(b, refs, recs)
    got, cells, stats, d = _run(tmp_path, "params:%s %s" % (f1, f2), [b], 5, 10, env={"DROPEST_MIN_PHRED": "10", "DROPEST_RPUPC": "1"})
    o = Oracle(min_genes_before=5, min_genes_after=10)
    for cb, umi, g, chr_, m, q in kept:
        o.add_record(cb, umi, g or "", chr_, m, umi_qual=q)
    o.set_initialized(); o.merge_and_filter()
    gi, ci, v = o.count_matrix(filtered=True)
    cols = [o.cell_barcode(int(k)) for k in o.filtered_cells()]
    want = {(o.gene_name(int(g)), cols[int(c)]): int(x) for g, c, x in zip(gi, ci, v)}
    assert cells == cols and got == want and len(want) > 300
    assert stats["saved"] == len(kept) and stats["low_quality"] == n_low and stats["cant_parse"] == n_missing + n_dup
    assert stats["total_reads"] == len(recs)
    # the qualities of the rows reached the molecules
    rp = d["reads_per_umi_per_cell"]
    assert len(rp["reads_per_umi"].value[0].value[0].value[1].value) == 8


def test_device_path_read_names_long_records_and_strings_with_n(tmp_path):
    """The device path on what its guesses find hard: read-name mode ("id!CB#UMI"), records longer than a 16 KB segment (no record starts in
    some segments), names that look like nothing in particular, barcodes / UMIs with N (packed by the host), a gene met first in the LAST
    window, UMI quality tags in one file (that window takes the record-by-record path).  Same .rds as the host reader, same counters."""
    reads = _reads(30_000)
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    rng = np.random.default_rng(9)
    recs = []
    for i, (cb, umi, g, chr_, mark) in enumerate(reads):
        if i % 997 == 0:
            umi = umi[:3] + "N" + umi[4:]
        if i % 1499 == 0:
            cb = "N" + cb[1:]
        g2 = "LATE_GENE" if i > 29_900 and g else g
        tags = ([("GX", "Z", g2)] if g2 else []) + [("NH", "i", 1)]
        seq = "ACGT" * (6000 if i % 4001 == 7 else 1 + i % 13)                     # a few records of 36 KB
        recs.append(bw.record(int(chr_[3:]), i, "x%d!%s#%s" % (i, cb, umi), seq=seq, tags=tags))
    bam = str(tmp_path / "names.bam")
    bw.write_bam(bam, refs, recs, block=30_000)
    host = _run(tmp_path / "host", "name", [bam], 3, 5, threads=4)
    dev = _run(tmp_path / "dev", "name", [bam], 3, 5, threads=4, env={"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_DEVICE_WINDOW_MB": "1"})
    assert dev[1] == host[1] and dev[0] == host[0] and len(host[0]) > 500
    assert {k: dev[2][k] for k in ("total_reads", "cant_parse", "low_quality", "saved")} == {k: host[2][k] for k in ("total_reads", "cant_parse", "low_quality", "saved")}
    assert any(g == "LATE_GENE" for g, _ in host[0])
    # round 6: a window's blocks are inflated at a fixed distance from the start of its buffer, before the record the window before cut off is put
    # in front of them; a cut-off record longer than that distance (here: 64 bytes, so nearly every window's) moves the window behind it.  With
    # DROPEST_BAM_PIPELINE window k + 1 is given to the device before window k's chain is known: the same container
    # (the compressed bytes go to the device in pinned pieces with the block table read from the file; DROPEST_BAM_WHOLE_WINDOW_STAGING: whole windows
    # through two pinned buffers as in rounds 4-5; DROPEST_BAM_NO_UPLOAD_AHEAD: copied by the window call; one reader thread instead of four)
    for env in ({"DROPEST_BAM_TEST_TAIL_RESERVE": "64"}, {"DROPEST_BAM_PIPELINE": "1"}, {"DROPEST_BAM_PIPELINE": "1", "DROPEST_BAM_TEST_TAIL_RESERVE": "64"},
                {"DROPEST_BAM_PIPELINE": "1", "DROPEST_BAM_TEST_TAIL_RESERVE": "0", "DROPEST_BAM_TEST_SPOIL_GUESSES": "1"},
                {"DROPEST_BAM_WHOLE_WINDOW_STAGING": "1"}, {"DROPEST_BAM_NO_UPLOAD_AHEAD": "1"}, {"DROPEST_BAM_READERS": "1"}, {"DROPEST_INFLATE_PAR": "0"},
                # the block table of a window from four stretches, each begun at a block found by its magic bytes (files of many small blocks)
                {"DROPEST_BAM_TEST_STRETCH": "100000,8"}, {"DROPEST_BAM_TEST_STRETCH": "100000,8", "DROPEST_BAM_PIECE_MB": "1", "DROPEST_BAM_READERS": "3"}):
        again = _run(tmp_path / ("dev_" + "_".join(sorted(env))), "name", [bam], 3, 5, threads=4, env=dict(env, DROPEST_BAM_DEVICE="1", DROPEST_BAM_DEVICE_WINDOW_MB="1"))
        assert again[1] == host[1] and again[0] == host[0] and again[2]["saved"] == host[2]["saved"]
    # UMI quality tags: the windows that carry them go record by record; the per-molecule quality sums land in the .rds
    recs_q = []
    for i, (cb, umi, g, chr_, mark) in enumerate(reads[:8000]):
        tags = [("CB", "Z", cb), ("UB", "Z", umi), ("UQ", "Z", "".join(chr(33 + int(x)) for x in rng.integers(2, 40, len(umi))))] + ([("GX", "Z", g)] if g else [])
        recs_q.append(bw.record(int(chr_[3:]), i, "q%d" % i, tags=tags))
    bam_q = str(tmp_path / "qual.bam")
    bw.write_bam(bam_q, refs, recs_q, block=20_000)
    host_q = _run(tmp_path / "host_q", "filled", [bam_q], 2, 3, threads=3, env={"DROPEST_RPUPC": "1"})
    dev_q = _run(tmp_path / "dev_q", "filled", [bam_q], 2, 3, threads=3, env={"DROPEST_BAM_DEVICE": "1", "DROPEST_RPUPC": "1"})
    assert dev_q[1] == host_q[1] and dev_q[0] == host_q[0] and dev_q[2]["saved"] == host_q[2]["saved"] == 8000
    def molecules(d):
        rp = d["reads_per_umi_per_cell"]
        cells_l, genes_l = rp["cells"].value, rp["genes"].value
        seen = {}
        for ci, gi, per_gene in zip(rp["cell_indexes"].value, rp["gene_indexes"].value, rp["reads_per_umi"].value):
            for name, entry in zip(per_gene.names, per_gene.value):
                seen[(cells_l[int(ci)], genes_l[int(gi)], name)] = (int(entry.value[0].value[0]), [float(x) for x in entry.value[1].value])
        return seen
    a, b = molecules(host_q[3]), molecules(dev_q[3])
    assert len(a) > 1000 and a == b


def test_quality_strings_in_bulk_and_where_bulk_must_not_be_taken(tmp_path):
    """UQ tags of one length go through the bulk paths (one quality row per read beside the packed columns: host reader and device path); a file
    whose later reads have no UQ tag, or one of another length, leaves those windows to the record-by-record path -- three readers, one answer:
    the container of DROPEST_BAM_RECORD_BY_RECORD=1, matrices and the per-molecule quality means of reads_per_umi_per_cell."""
    rng = np.random.default_rng(17)
    reads = _reads(24_000, seed_cells=16)
    refs = [("chr%d" % i, 1_000_000) for i in range(25)]
    recs = []
    for i, (cb, umi, g, chr_, mark) in enumerate(reads):
        third = i * 3 // len(reads)
        cb = ("AAAA", "CCCC", "GGGG")[third] + cb[4:]      # every third has cells of its own: its molecules are new ones (no length to clash with)
        tags = [("CB", "Z", cb), ("UB", "Z", umi)] + ([("GX", "Z", g)] if g else [])
        if third == 0:
            tags.append(("UQ", "Z", "".join(chr(33 + int(x)) for x in rng.integers(2, 40, len(umi)))))
        elif third == 2:
            tags.append(("UQ", "Z", "".join(chr(33 + int(x)) for x in rng.integers(2, 40, len(umi) - 2))))     # shorter strings
        recs.append(bw.record(int(chr_[3:]), i, "q%d" % i, tags=tags))
    bam = str(tmp_path / "mixed.bam")
    bw.write_bam(bam, refs, recs, block=25_000)

    def molecules(d):
        rp = d["reads_per_umi_per_cell"]
        cells_l, genes_l = rp["cells"].value, rp["genes"].value
        seen = {}
        for ci, gi, per_gene in zip(rp["cell_indexes"].value, rp["gene_indexes"].value, rp["reads_per_umi"].value):
            for name, entry in zip(per_gene.names, per_gene.value):
                seen[(cells_l[int(ci)], genes_l[int(gi)], name)] = (int(entry.value[0].value[0]), [float(x) for x in entry.value[1].value])
        return seen
    runs = {}
    for name, env in (("one_by_one", {"DROPEST_BAM_RECORD_BY_RECORD": "1"}), ("bulk", {}), ("device", {"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_DEVICE_WINDOW_MB": "1"})):
        got, cells, stats, d = _run(tmp_path / name, "filled", [bam], 2, 3, threads=4, env=dict(env, DROPEST_RPUPC="1"))
        runs[name] = (got, cells, {k: stats[k] for k in ("total_reads", "cant_parse", "low_quality", "saved")}, molecules(d))
    assert runs["bulk"] == runs["one_by_one"] and runs["device"] == runs["one_by_one"]
    m = runs["bulk"][3]
    assert len(m) > 3000 and sum(1 for v in m.values() if len(v[1]) == 8) > 500 and sum(1 for v in m.values() if len(v[1]) == 6) > 500 and sum(1 for v in m.values() if not v[1]) > 500
    # the same reads in BGZF blocks of 40 bytes: the device path takes the file in several windows, and the window where tagged and untagged
    # reads meet goes record by record AFTER windows that went to the container as device columns -- the look at the molecules made so far
    # (quality length of an existing molecule) then reads columns that were pushed from the device
    bam3 = str(tmp_path / "small_blocks.bam")
    bw.write_bam(bam3, refs, recs[:16_000], block=40)
    small = {}
    for name, env in (("one_by_one", {"DROPEST_BAM_RECORD_BY_RECORD": "1"}), ("device", {"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_DEVICE_WINDOW_MB": "1"})):
        got, cells, stats, d = _run(tmp_path / ("small_" + name), "filled", [bam3], 2, 3, threads=4, env=dict(env, DROPEST_RPUPC="1"))
        small[name] = (got, cells, {k: stats[k] for k in ("total_reads", "cant_parse", "low_quality", "saved")}, molecules(d))
    assert small["device"] == small["one_by_one"] and small["device"][2]["total_reads"] == 16_000
    # ... and a read without a quality string that meets a molecule created with one is UMI::add_read's exception (UMI.cpp:26-28) from all three
    clash = list(recs[:8000]) + [bw.record(int(c[3:]), 8000 + i, "x%d" % i, tags=[("CB", "Z", "AAAA" + cb[4:]), ("UB", "Z", umi)] + ([("GX", "Z", g)] if g else []))
                                for i, (cb, umi, g, c, m) in enumerate(reads[:8000])]
    bam2 = str(tmp_path / "clash.bam")
    bw.write_bam(bam2, refs, clash, block=25_000)
    for env in ({"DROPEST_BAM_RECORD_BY_RECORD": "1"}, {}, {"DROPEST_BAM_DEVICE": "1"}):
        res = subprocess.run([TOOL, str(tmp_path / "clash_out"), "filled", "2", "3", "-", "4", bam2], capture_output=True, text=True, env=dict(os.environ, **env))
        assert res.returncode == 1 and "Wrong quality length: 0, expected: 8" in res.stderr, (env, res.stderr)
