"""Runs the reference-style C++ test driver (tests/cpp/test_facade.cpp) against the facade on the GPU."""
import os
import subprocess

import pytest

from dropest_amd.build import build_facade

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_reference_cases_through_the_cpp_facade(tmp_path):
    _, exe = build_facade()
    res = subprocess.run([exe, os.path.join(ROOT, "dropest_amd", "data", "barcodes"), str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all reference test cases passed" in res.stdout
    assert os.path.exists(tmp_path / "cell.counts.mtx") and os.path.exists(tmp_path / "cell.counts.genes.tsv")


def test_results_rds_written_by_the_facade(tmp_path):
    """ResultsPrinter::save_results writes the R list `d` natively; parsed back and checked against the fixture's
    known answers (Tests/TestEstimation.cpp:237-280) and against the MatrixMarket triple of the same run."""
    import numpy as np
    import rds_reader as rr
    _, exe = build_facade()
    res = subprocess.run([exe, os.path.join(ROOT, "dropest_amd", "data", "barcodes"), str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    d = rr.read_rds(str(tmp_path / "cell.counts.rds"))
    assert d.names == ["cm", "cm_raw", "reads_per_chr_per_cells", "mean_reads_per_umi", "saturation_info", "merge_targets",
                       "aligned_reads_per_cell", "aligned_umis_per_cell", "requested_umis_per_cb", "requested_reads_per_cb"]
    cm, genes, cells = rr.dgcmatrix_to_dense(d["cm"])
    assert cells == ["AAATTAGGTCCC", "AAATTAGGTCCA"]                       # filtered cells, ascending compare_cells order
    val = lambda g, c: int(cm[genes.index(g), cells.index(c)])           # noqa: E731
    assert val("Gene1", "AAATTAGGTCCA") == 2 and val("Gene3", "AAATTAGGTCCA") == 2 and val("Gene10", "AAATTAGGTCCC") == 1
    assert int((cm != 0).sum()) == 7
    # the .mtx triple of the same run describes the same matrix
    lines = open(tmp_path / "cell.counts.mtx").read().split("\n")[2:]
    trip = np.array([[int(float(x)) for x in ln.split()] for ln in lines if ln])
    dense = np.zeros_like(cm); dense[trip[:, 0] - 1, trip[:, 1] - 1] = trip[:, 2]
    assert np.array_equal(dense, cm)
    assert open(tmp_path / "cell.counts.genes.tsv").read().split() == genes
    raw, rgenes, rcells = rr.dgcmatrix_to_dense(d["cm_raw"])
    assert rcells == ["AAATTAGGTCCA", "AAATTAGGTCCC"] and raw.sum() >= cm.sum()
    assert d["merge_targets"].names == ["AAATTAGGTCCG", "AAATTAGGTCGG", "CCCTTAGGTCCA", "CAATTAGGTCCG"]
    assert [t.value[0] for t in d["merge_targets"].value] == ["AAATTAGGTCCC", "AAATTAGGTCCA", "AAATTAGGTCCA", "AAATTAGGTCCA"]
    umis = dict(zip(d["aligned_umis_per_cell"].names, d["aligned_umis_per_cell"].value.tolist()))
    assert umis["AAATTAGGTCCC"] == 4                                       # Stats::merge quirk, as asserted by the C++ driver
    sat = d["saturation_info"]
    assert sat.names == ["reads", "cbs", "umis"] and len(sat["reads"].value) == len(sat["cbs"].value) == len(sat["umis"].value)
    assert int(sat["reads"].value.sum()) == int(d["requested_reads_per_cb"].value.sum())
    chr_frames = d["reads_per_chr_per_cells"]
    assert chr_frames.names == ["Exon", "Intron", "Intergenic"]
    assert list(chr_frames["Exon"].attributes["class"].value) == ["data.frame"]
    assert int(sum(col.value.sum() for col in chr_frames["Exon"].value)) == 16          # 17 exon reads, one of them in the excluded cell AAAAAAAAAAAA
    assert d["mean_reads_per_umi"].names == rcells and np.all(d["mean_reads_per_umi"].value >= 1.0)
    vel = rr.read_rds(str(tmp_path / "cell.counts.matrices.rds"))                     # save_intron_exon_matrices (-V)
    assert vel.names == ["exon", "intron", "spanning"]
    ex, exg, exc = rr.dgcmatrix_to_dense(vel["exon"])
    assert exc == cells and np.array_equal(ex, cm) and exg == genes                     # all reads of the fixture are exonic
    assert rr.dgcmatrix_to_dense(vel["intron"])[0].sum() == 0 and rr.dgcmatrix_to_dense(vel["spanning"])[0].sum() == 0
    full = rr.read_rds(str(tmp_path / "cell.counts.full.rds"))
    rp = full["reads_per_umi_per_cell"]
    assert rp.names == ["cells", "genes", "cell_indexes", "gene_indexes", "reads_per_umi"]
    assert rp["cells"].value == cells and len(rp["reads_per_umi"].value) == len(rp["cell_indexes"].value) == 7
    first = rp["reads_per_umi"].value[0]
    assert first.kind == "list" and first.names and first.value[0].value[0].kind == "integer"
    # the reference's fixture passes the UMI itself as its quality string (Tests/TestEstimation.cpp:27-31): every molecule's
    # sums are k x the UMI's characters, k = reads added by add_read to the molecule whose sums survived the merges
    # (UMI::merge adds no qualities), and UMI::mean_quality is (sum - 33) / read_count in unsigned integer arithmetic
    n_checked = 0
    for per_gene in rp["reads_per_umi"].value:
        for umi, entry in zip(per_gene.names, per_gene.value):
            reads, mean = int(entry.value[0].value[0]), [int(x) for x in entry.value[1].value]
            assert len(mean) == len(umi)
            assert any(mean == [(k * ord(ch) - 33) // reads for ch in umi] for k in range(1, reads + 1)), (umi, reads, mean)
            n_checked += 1
    assert n_checked >= 7
