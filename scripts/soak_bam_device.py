"""Soak (not part of the suites): random BAM files through tests/cpp/bam_to_counts by the host reader and by the device path
(DROPEST_BAM_DEVICE=1: csrc/k_inflate.h + k_bamparse.h) -- same counters, same cells, same count matrix.  What is drawn per file: BGZF block
size (3 bytes .. 64 KB: records and their length fields cut anywhere), zlib level, tags in random order with numeric and array tags in
between, missing barcode / UMI / gene tags, unmapped / secondary records, unknown reference ids, N in barcodes and UMIs, read-name mode,
records of up to 70 KB, names of 1 .. 200 characters, the window size of the device path; every third file takes its genes from a
random GTF (-g) with spliced alignments.
Run on a GPU box: PYTHONPATH=. python scripts/soak_bam_device.py   (SOAK_CASES, SOAK_SEED)"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                   # noqa: E402
import bam_writer as bw                              # noqa: E402
import rds_reader as rr                              # noqa: E402
from dropest_amd.build import build_facade           # noqa: E402

build_facade()
TOOL = os.path.join(ROOT, "tests", "cpp", "bam_to_counts")
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "31")))
B = list("ACGT")


def run(out, mode, bam, env):
    res = subprocess.run([TOOL, out, mode, "2", "3", "-", str(int(rng.integers(1, 9))), bam], capture_output=True, text=True, timeout=600, env=dict(os.environ, DROPEST_RPUPC="1", **env))
    if res.returncode:
        return {"error": res.stderr.strip().splitlines()[-1] if res.stderr.strip() else "rc %d" % res.returncode}
    st = json.loads(res.stdout.strip().splitlines()[-1])
    d = rr.read_rds(out + ".rds")
    cm, genes, cells = rr.dgcmatrix_to_dense(d["cm"])
    mol = {}
    try:
        rp = d["reads_per_umi_per_cell"]
        cells_l, genes_l = rp["cells"].value, rp["genes"].value
        for ci, gi, per_gene in zip(rp["cell_indexes"].value, rp["gene_indexes"].value, rp["reads_per_umi"].value):
            for name, entry in zip(per_gene.names, per_gene.value):
                mol[(cells_l[int(ci)], genes_l[int(gi)], name)] = (int(entry.value[0].value[0]), tuple(float(x) for x in entry.value[1].value))
    except Exception as e:      # (no such entry: compare what there is)
        mol = {"unreadable": str(e)}
    return {"stats": {k: st[k] for k in ("total_reads", "cant_parse", "low_quality", "saved", "cells", "real_cells")}, "cells": cells, "molecules": mol,
            "cm": {(genes[r], cells[c]): int(cm[r, c]) for r, c in zip(*np.nonzero(cm))}}


for case in range(int(os.environ.get("SOAK_CASES", "16"))):
    t0 = time.time()
    mode = "name" if case % 4 == 3 else "filled"
    n = int(rng.integers(2_000, 60_000))
    n_cb, n_umi, n_gene, n_ref = int(rng.integers(5, 300)), int(rng.integers(20, 3000)), int(rng.integers(3, 800)), int(rng.integers(1, 30))
    cbs = ["".join(rng.choice(B, int(rng.integers(8, 17)))) for _ in range(n_cb)]
    umis = ["".join(rng.choice(B, 8)) for _ in range(n_umi)]
    p_n, p_odd, long_every = float(rng.choice([0, 0.002, 0.02])), float(rng.choice([0.0, 0.03, 0.15])), int(rng.choice([0, 0, 997, 211]))
    # UMI quality tags: none / on every read / on the second half of the file (other cells: new molecules) / on nearly every read with a rare other length
    uq_mode = int(rng.integers(0, 4)) if mode == "filled" else 0
    recs = []
    for i in range(n):
        cb, umi = cbs[int(rng.integers(0, n_cb))], umis[int(rng.integers(0, n_umi))]
        if uq_mode == 2:
            cb = ("A" if i > n // 2 else "C") + cb[1:]
        if rng.random() < p_n:
            j = int(rng.integers(0, len(umi))); umi = umi[:j] + "N" + umi[j + 1:]
        if rng.random() < p_n / 2:
            cb = "N" + cb[1:]
        gene = None if rng.random() < 0.1 else "G%d_%s" % (int(rng.integers(0, n_gene)), "x" * int(rng.integers(0, 12)))
        tags = []
        if mode == "filled":
            tags += [("CB", "Z", cb), ("UB", "Z", umi)]
            if uq_mode == 1 or (uq_mode == 2 and i > n // 2) or (uq_mode == 3 and rng.random() < 0.999):
                tags.append(("UQ", "Z", "".join(chr(33 + int(x)) for x in rng.integers(2, 41, 8 if uq_mode != 3 or rng.random() < 0.9995 else 6))))
        if gene is not None:
            tags.append(("GX", "Z", gene))
            if rng.random() < 0.5:
                tags.append(("RE", "A", str(rng.choice(["N", "I", "E", "Q"]))))
        extra = [("NH", "i", int(rng.integers(0, 9))), ("xs", "C", 7), ("fl", "f", 0.5), ("ar", "B", [1, -2, 3, 4]), ("zz", "Z", "some text"), ("em", "Z", "")]
        tags += [extra[k] for k in rng.permutation(len(extra))[: int(rng.integers(0, len(extra) + 1))]]
        tags = [tags[k] for k in rng.permutation(len(tags))]
        flag, ref = 0, int(rng.integers(0, n_ref))
        r = rng.random()
        if r < p_odd / 4:
            flag = 4
        elif r < p_odd / 2:
            flag = 0x100
        elif r < 3 * p_odd / 4:
            ref = -1
        elif r < p_odd and mode == "filled":
            tags = [t for t in tags if t[0] != str(rng.choice(["CB", "UB"]))]
        seq = "ACGT" * (int(rng.integers(8000, 17000)) if long_every and i % long_every == 5 else int(rng.integers(1, 40)))
        name = ("r%d" % i) + "q" * int(rng.integers(0, 3) ** 5 % 190)
        if mode == "name":
            name = "%s!%s#%s" % (name, cb, umi) if rng.random() > p_odd / 4 else name
        recs.append(bw.record(ref, int(rng.integers(0, 60_000 if case % 3 == 1 else 1 << 28)), name, flag=flag, seq=seq, tags=tags,
                              cigar=None if rng.random() < 0.6 else [(len(seq) // 2, "M"), (int(rng.integers(1, 2000)), "N"), (len(seq) - len(seq) // 2, "M")]))
    tmp = tempfile.mkdtemp()
    bam = os.path.join(tmp, "t.bam")
    genv = {}
    if case % 3 == 1:      # -g: genes from a random GTF over the first references (the others: "chromosome not found" = cannot be parsed)
        import gzip
        lines = []
        for k in range(max(1, n_ref - 2)):
            pos = int(rng.integers(0, 3000))
            for g in range(int(rng.integers(5, 80))):
                for x in range(int(rng.integers(1, 5))):
                    ln = int(rng.integers(40, 500))
                    lines.append('chr%d\tsrc\texon\t%d\t%d\t.\t+\t.\tgene_id "G%d_%d"; transcript_id "T%d_%d_%d";' % (k, pos + 1, pos + ln, k, g, k, g, x % 2))
                    pos += ln + int(rng.integers(20, 700))
                pos += int(rng.integers(0, 3000)) - 400 * int(rng.random() < 0.3)      # some genes overlap the next one
                pos = max(pos, 0)
        gtf = os.path.join(tmp, "a.gtf.gz")
        with gzip.open(gtf, "wt") as f:
            f.write("\n".join(lines) + "\n")
        genv = {"DROPEST_GTF": gtf}
    block = int(rng.choice([3, 17, 250, 4000, 30_000, 0xFF00]))
    if block < 100 and n > 15_000:
        block = 4000
    bw.write_bam(bam, [("chr%d" % k, 1 << 28) for k in range(n_ref)], recs, block=block)
    host = run(os.path.join(tmp, "host"), mode, bam, dict(genv))
    if uq_mode and n < 40_000:      # the bulk path with quality rows against add_record per read
        one = run(os.path.join(tmp, "one"), mode, bam, dict(genv, DROPEST_BAM_RECORD_BY_RECORD="1"))
        if one != host:
            print(case, "record-by-record and bulk DIFFER", one.get("stats", one.get("error")), host.get("stats", host.get("error")), flush=True)
            sys.exit(1)
    dev = run(os.path.join(tmp, "dev"), mode, bam, dict(genv, DROPEST_BAM_DEVICE="1", DROPEST_BAM_DEVICE_WINDOW_MB=str(int(rng.choice([1, 1, 4, 32])))))
    same = host == dev
    print(case, mode + (" -g" if genv else ""), "reads", n, "block", block, "p_n", p_n, "p_odd", p_odd, "long", long_every, "uq", uq_mode, "->", host.get("stats", host.get("error")),
          "SAME" if same else "DIFFERENT %s" % (dev.get("stats", dev.get("error")),), "%.1fs" % (time.time() - t0), flush=True)
    if not same:
        sys.exit(1)
    import shutil; shutil.rmtree(tmp, ignore_errors=True)
print("all the same")
