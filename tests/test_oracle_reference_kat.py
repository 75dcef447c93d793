"""Pins the CPU oracle (oracle/dropest_oracle.cpp) on the reference's own known-answer tests.

Every test below replays one Boost.Test case of the reference that touches the hot path and asserts
the same values (reference file:line in each docstring; files under /root/reference/Tests).  The inputs
(reads, whitelists) are the reference fixtures' data, re-typed here as data; no reference source is
executed.  CPU only.
"""
import os

import numpy as np
import pytest

from oracle import Oracle
from oracle import binding as ob

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dropest_amd", "data", "barcodes")
TEST_EST = os.path.join(DATA, "test_est")


def fixture_container():
    """Tests/TestEstimation.cpp:33-80 (struct Fixture): RealBarcodesMergeStrategy(parser, 0, 0, 7, 0),
    MergeUMIsStrategySimple(1), default marks; 17 reads over 7 CBs; read_info() uses umi as quality (:27-31)."""
    o = Oracle(merge_kind=1, barcodes_kind=0, barcodes_file=TEST_EST, min_genes_before=0, min_genes_after=0,
               max_cb_merge_ed=7, min_merge_fraction=0.0, umi_merge_kind=0, max_umi_merge_ed=1)
    reads = [
        ("AAATTAGGTCCA", "AAACCT", "Gene1"), ("AAATTAGGTCCA", "CCCCCT", "Gene2"),
        ("AAATTAGGTCCA", "ACCCCT", "Gene3"), ("AAATTAGGTCCA", "ACCCCT", "Gene4"),
        ("AAATTAGGTCCC", "CAACCT", "Gene1"), ("AAATTAGGTCCC", "CAACCT", "Gene10"),
        ("AAATTAGGTCCC", "CAACCT", "Gene20"),
        ("AAATTAGGTCCG", "CAACCT", "Gene1"),
        ("AAATTAGGTCGG", "AAACCT", "Gene1"), ("AAATTAGGTCGG", "CCCCCT", "Gene2"),
        ("CCCTTAGGTCCA", "CCATTC", "Gene3"), ("CCCTTAGGTCCA", "CCCCCT", "Gene2"),
        ("CCCTTAGGTCCA", "ACCCCT", "Gene3"),
        ("CAATTAGGTCCG", "CAACCT", "Gene1"), ("CAATTAGGTCCG", "AAACCT", "Gene1"),
        ("CAATTAGGTCCG", "CCCCCT", "Gene2"),
        ("AAAAAAAAAAAA", "CCCCCT", "Gene2"),
    ]
    for cb, umi, gene in reads:
        o.add_record(cb, umi, gene, "", 2, umi_qual=umi)
    o.set_initialized()
    return o


def test_barcodes_file():
    """TestEstimation.cpp:98-121 testBarcodesFile: reverse-complemented whitelist parts + missing file throws."""
    o = fixture_container()
    assert o.wl_part(0) == ["AAT", "GAA", "AAA"]
    assert o.wl_part(1) == ["TTAGGTCCA", "TTAGGGGCC", "TTAGGTCCC"]
    with pytest.raises(RuntimeError):
        Oracle().wl_load(0, "/barcodes.wrong")


def test_umigs_intersection():
    """TestEstimation.cpp:160-178 testUmigsIntersection = 2, 1, 0."""
    o = fixture_container()
    cid = o.cell_id_by_cb
    assert o.umig_intersection(cid("AAATTAGGTCCA"), cid("CCCTTAGGTCCA")) == 2
    assert o.umig_intersection(cid("AAATTAGGTCCC"), cid("AAATTAGGTCCG")) == 1
    assert o.umig_intersection(cid("AAATTAGGTCCA"), cid("AAATTAGGTCCC")) == 0


def test_fill_distances():
    """TestEstimation.cpp:180-206 testFillDistances: parts {AAT,AAA,CCT}x2, barcode ACTACT."""
    o = Oracle()
    o.wl_set(0, ["AAT", "AAA", "CCT"], ["AAT", "AAA", "CCT"])
    for part in (0, 1):
        vals, idx = o.wl_distances("ACTACT", part)
        assert len(vals) == 3
        assert list(vals) == [1, 1, 2]
        assert idx[2] == 1


def test_real_neighbours_cbs():
    """TestEstimation.cpp:208-225 testRealNeighboursCbs."""
    o = fixture_container()
    ids = o.real_neighbours(o.cell_id_by_cb("CAATTAGGTCCG"))
    assert [o.cell_barcode(int(i)) for i in ids] == ["AAATTAGGTCCA", "AAATTAGGTCCC"]
    ids = o.real_neighbours(o.cell_id_by_cb("AAATTAGGTCCC"))
    assert len(ids) == 1 and o.cell_barcode(int(ids[0])) == "AAATTAGGTCCC"


def test_real_neighbours_targets():
    """TestEstimation.cpp:227-235 testRealNeighbours: targets of cells 0..5 = 0,1,1,0,0,0."""
    o = fixture_container()
    assert [o.merge_target(i) for i in range(6)] == [0, 1, 1, 0, 0, 0]


def test_merge_by_real_barcodes():
    """TestEstimation.cpp:237-280 testMergeByRealBarcodes (+ SURVEY probe: merge_targets 0 1 1 0 0 0 6)."""
    o = fixture_container()
    o.merge_and_filter()
    assert o.n_cells == 7
    f = o.filtered_cells()
    assert len(f) == 2
    rows = o.cell_rows()
    assert rows[int(f[0]), 3] == 3 and rows[int(f[1]), 3] == 4          # cell sizes (#genes)
    mol = o.molecule_dict()
    c0, c1 = o.cell_barcode(int(f[0])), o.cell_barcode(int(f[1]))
    assert len(mol[(c0, "Gene1")]) == 1 and mol[(c0, "Gene1")]["CAACCT"][0] == 2
    assert len(mol[(c1, "Gene1")]) == 2 and mol[(c1, "Gene1")]["AAACCT"][0] == 3
    assert len(mol[(c1, "Gene2")]) == 1 and mol[(c1, "Gene2")]["CCCCCT"][0] == 4
    assert len(mol[(c1, "Gene3")]) == 2
    assert mol[(c1, "Gene3")]["ACCCCT"][0] == 2 and mol[(c1, "Gene3")]["CCATTC"][0] == 1
    assert list(rows[:, 0]) == [0, 0, 1, 1, 1, 1, 0]                    # is_merged
    assert int(rows[:, 1].sum()) == 1                                    # exactly one excluded
    assert list(o.merge_targets()) == [0, 1, 1, 0, 0, 0, 6]


def test_split_barcode_and_const_length_parser():
    """TestEstimation.cpp:282-320 testSplitBarcode + testConstLengthBarcodeParser."""
    o = Oracle()
    o.wl_load(1, os.path.join(DATA, "indrop_v3"))
    assert o.wl_split("TAATGAGCACTAATGA") == ["TAATGAGC", "ACTAATGA"]
    assert o.wl_parts() == 2
    p0, p1 = o.wl_part(0), o.wl_part(1)
    assert len(p0) == 384 and len(p1) == 384 and len(p0[0]) == 8 and len(p1[0]) == 8
    t = Oracle()
    t.wl_load(1, os.path.join(DATA, "10x_aug_2016_split"))
    q0, q1 = t.wl_part(0), t.wl_part(1)
    assert t.wl_parts() == 2 and len(q0) == 480 and len(q1) == 1536 and len(q0[0]) == 7 and len(q1[0]) == 9
    v0, _ = t.wl_distances("GGTGCGTAGCTAAACA", 0)
    v1, _ = t.wl_distances("GGTGCGTAGCTAAACA", 1)
    assert v0[0] == 0 and v1[0] == 0


def test_umi_exclusion():
    """TestEstimation.cpp:369-397 testUmiExclusion: marks OR per UMI; query 'e' uses mark equality."""
    o = Oracle(merge_kind=1, barcodes_file=TEST_EST, min_genes_before=0, min_genes_after=0,
               min_merge_fraction=0.0, match_levels="e")
    cb = "AAATTAGGTCCA"
    for umi, gene in [("AAACCT", "Gene1"), ("CCCCCT", "Gene2"), ("ACCCCT", "Gene3"), ("ACCCCT", "Gene4")]:
        o.add_record(cb, umi, gene, "", 2, umi_qual=umi)
    assert o.molecule_dict()[(cb, "Gene4")]["ACCCCT"][0] == 1
    o.add_record(cb, "TTTTTT", "Gene3", "chr1", 1, umi_qual="TTTTTT")
    o.add_record(cb, "ACCCCT", "Gene4", "chr1", 1, umi_qual="ACCCCT")
    mol = o.molecule_dict()
    assert mol[(cb, "Gene3")]["TTTTTT"][1] & 1 and mol[(cb, "Gene4")]["ACCCCT"][1] & 1
    o.set_initialized()
    o.merge_and_filter()
    mol = o.molecule_dict()
    # requested (mark == EXON exactly): Gene3 keeps only ACCCCT, Gene4 has nothing requested
    req3 = {u: r for u, (r, m) in mol[(cb, "Gene3")].items() if m == 2}
    req4 = {u: r for u, (r, m) in mol[(cb, "Gene4")].items() if m == 2}
    assert req3 == {"ACCCCT": 1} and req4 == {}
    assert mol[(cb, "Gene4")]["ACCCCT"][0] == 2
    g, c, v = o.count_matrix(filtered=True)
    names = {o.gene_name(int(x)) for x in g}
    assert "Gene4" not in names and "Gene3" in names


def test_gene_match_level_exclusion():
    """TestEstimation.cpp:399-466 (core-container part): same UMI seen with EXON then EXON|NOT_ANNOTATED."""
    cb, g = "TGAGTTCTGTTACTGCATC", "FAM138A"
    o = Oracle(merge_kind=1, barcodes_file=TEST_EST, min_genes_before=0, min_genes_after=0,
               min_merge_fraction=0.0, match_levels="e")
    o.add_record(cb, "ATGGGC", g, "chrX", 2)
    assert o.molecule_dict()[(cb, g)]["ATGGGC"][0] == 1
    o.add_record(cb, "ATGGGC", g, "chrX", 3)
    o.add_record(cb, "ATGGGC", g, "chrX", 2)
    o.add_record(cb, "ATTTTC", g, "chrX", 3)
    mol = o.molecule_dict()
    assert mol[(cb, g)]["ATGGGC"][1] & 1 and mol[(cb, g)]["ATTTTC"][1] & 1
    o.set_initialized(); o.merge_and_filter()
    gi, _, _ = o.count_matrix(filtered=True)
    assert len(gi) == 0                       # nothing requested under "e"
    o2 = Oracle(merge_kind=1, barcodes_file=TEST_EST, min_genes_before=0, min_genes_after=0,
                min_merge_fraction=0.0, match_levels="eE")
    o2.add_record(cb, "ATGGGC", g, "chrX", 2); o2.add_record(cb, "ATGGGC", g, "chrX", 3)
    o2.add_record(cb, "ATTTTC", g, "chrX", 2)
    assert o2.molecule_dict()[(cb, g)]["ATTTTC"][1] == 2
    o2.add_record(cb, "ATTTTC", g, "chrX", 3)
    o2.set_initialized(); o2.merge_and_filter()
    assert len(o2.molecule_dict()[(cb, g)]) == 2


def test_umi_merge_explicit():
    """TestEstimation.cpp:468-488 testUMIMerge."""
    o = Oracle(merge_kind=1, barcodes_file=TEST_EST, min_genes_before=0, min_genes_after=0, min_merge_fraction=0.0)
    cb = "AAATTAGGTCCA"
    for umi in ["AAACCT", "CCCCCT", "AAATTN", "ACCCCT"]:
        o.add_record(cb, umi, "Gene1", "", 2, umi_qual=umi)
    o.merge_umis_explicit(0, "Gene1", {"AAACCT": "CCCCCT", "AAATTN": "GGGGGG", "ACCCCT": "ACCCCT"})
    m = o.molecule_dict()[(cb, "Gene1")]
    assert len(m) == 3 and m["CCCCCT"][0] == 2 and m["GGGGGG"][0] == 1 and m["ACCCCT"][0] == 1


def test_fill_wrong_umis():
    """TestEstimation.cpp:490-503 testFillWrongUmis."""
    for umi in ["AAANTTT", "AAANCTT", "NNNNNNN"]:
        t = ob.fill_wrong_umi(umi)
        assert t != umi and ob.hamming_distance(umi, t) == 0 and "N" not in t


def test_umi_merge_strategy_simple():
    """TestEstimation.cpp:505-540 testUMIMergeStrategySimple."""
    o = Oracle(merge_kind=1, barcodes_file=TEST_EST, min_genes_before=0, min_genes_after=0, min_merge_fraction=0.0)
    cb = "AAATTAGGTCCA"
    for umi in ["AAACCT", "AAACCT", "AAACCG", "AAACCN", "CCCCCT", "ACCCCT"]:
        o.add_record(cb, umi, "Gene1", "", 2, umi_qual=umi)
    for umi in ["TTTTTT", "TTTNNG", "TTGNNG", "ACCCCT", "NNNNNN"]:
        o.add_record(cb, umi, "Gene2", "", 2, umi_qual=umi)
    o.set_initialized()
    o.merge_umis_only()
    mol = o.molecule_dict()
    g1, g2 = mol[(cb, "Gene1")], mol[(cb, "Gene2")]
    assert len(g1) == 4 and len(g2) == 3
    assert g1["AAACCT"][0] == 3 and g1["AAACCG"][0] == 1 and g1["CCCCCT"][0] == 1 and g1["ACCCCT"][0] == 1
    assert "TTTTTT" in g2 and "ACCCCT" in g2
    assert all("N" not in u for u in g2)


def test_umi_merge_strategy_directional():
    """TestEstimation.cpp:588-608 testUMIMergeStrategyDirectional."""
    o = Oracle(umi_merge_kind=1, max_umi_merge_ed=1, umi_mult=2.0)
    t = o.directional_targets([("AAA", 2), ("AAC", 5), ("AAT", 6), ("AGT", 20), ("CCC", 10), ("TCC", 20)])
    assert t == {"AAA": "AGT", "AAT": "AGT", "CCC": "TCC"}


def test_edit_distance():
    """Tests/TestTools.cpp:47-54 testEditDistance."""
    assert ob.edit_distance("ATTTTC", "ATTTGC") == 1
    assert ob.edit_distance("ATTTTCC", "ATTTGNC") == 1
    assert ob.edit_distance("ATTTTCC", "ATTTGNC", False) == 2
    assert ob.edit_distance("ATTTTCC", "ATTTGTC") == 2
    assert ob.edit_distance("ATTTTCC", "ATTTTCC") == 0


def test_read_params():
    """Tests/TestTools.cpp:56-87 testReadParams (id!CB#UMI encoding)."""
    assert ob.parse_encoded_id("@111!ATTTGC#ATATC") == ("ATTTGC", "ATATC")
    assert ob.parse_encoded_id("111!ATTTG#ATAT") == ("ATTTG", "ATAT")
    assert ob.parse_encoded_id("!ATTTGC#ATATC") == ("ATTTGC", "ATATC")
    assert ob.parse_encoded_id("trash!ATTTG#ATAT") == ("ATTTG", "ATAT")
    assert ob.parse_encoded_id("1111!AAATTTTATA#TTGG") == ("AAATTTTATA", "TTGG")
    with pytest.raises(RuntimeError):
        ob.parse_encoded_id("ATTTG#ATAT")


def test_survey_probes():
    """SURVEY.md §7 'hard parts' observations made with the compiled reference while surveying:
    (i) Stats::merge adds TOTAL_UMIS even for coinciding UMIs; (ii) random fill after srand(42) draws
    G,A,C,C and decrements TOTAL_UMIS for the re-keyed UMI."""
    o = fixture_container()
    o.merge_and_filter()
    rows = o.cell_rows()
    # cell 1 (AAATTAGGTCCC: 3 UMIs) absorbed cell 2 (1 UMI, same UMI-gene CAACCT/Gene1) -> stat 4, 3 distinct
    assert rows[1, 7] == 4
    assert sum(len(v) for (cb, _), v in o.molecule_dict().items() if cb == "AAATTAGGTCCC") == 3
    p = Oracle(min_genes_before=0, min_genes_after=0)
    p.add_record("AAAA", "ACGTCC", "G"); p.add_record("AAAA", "NNNNAA", "G")
    p.set_initialized(); p.merge_and_filter()
    # all-N prefix has wildcard distance 2 to ACGTCC (> 1) -> random fill GACC + AA
    assert set(p.molecule_dict()[("AAAA", "G")]) == {"ACGTCC", "GACCAA"}
    assert p.cell_rows()[0, 7] == 1


# ---------------------------------------------------------------------------------------------------
# -M: PoissonTargetEstimator / PoissonRealBarcodesMergeStrategy (Tests/TestEstimationMergeProbs.cpp:30-140)
# ---------------------------------------------------------------------------------------------------
def _poisson_fixture():
    o = Oracle(merge_kind=3, barcodes_kind=0, barcodes_file=os.path.join(DATA, "test_est"), min_genes_before=0, min_genes_after=0,
               max_merge_prob=1e-4, max_real_merge_prob=1e-7)
    reads = [("AAATTAGGTCCA", "AAACCT", "Gene1"), ("AAATTAGGTCCA", "CCCCCT", "Gene2"), ("AAATTAGGTCCA", "ACCCCT", "Gene3"),
             ("AAATTAGGTCCC", "CAACCT", "Gene1"), ("AAATTAGGTCCG", "CAACCT", "Gene1"),
             ("AAATTAGGTCGG", "AAACCT", "Gene1"), ("AAATTAGGTCGG", "CCCCCT", "Gene2"),
             ("CCCTTAGGTCCA", "CCATTC", "Gene3"), ("CCCTTAGGTCCA", "CCCCCT", "Gene2"), ("CCCTTAGGTCCA", "ACCCCT", "Gene3"),
             ("CAATTAGGTCCG", "CAACCT", "Gene1"), ("CAATTAGGTCCG", "AAACCT", "Gene1"), ("CAATTAGGTCCG", "CCCCCT", "Gene2"),
             ("CAATTAGGTCCG", "TTTTTT", "Gene2"), ("CAATTAGGTCCG", "TTCTTT", "Gene2"),
             ("CCCCCCCCCCCC", "CAACCT", "Gene1"), ("CCCCCCCCCCCC", "AAACCT", "Gene1"), ("CCCCCCCCCCCC", "CCCCCT", "Gene2"),
             ("CCCCCCCCCCCC", "TTTTTT", "Gene2"), ("CCCCCCCCCCCC", "TTCTTT", "Gene2"), ("TAATTAGGTCCA", "AAAAAA", "Gene4")]
    for cb, umi, g in reads:
        o.add_record(cb, umi, g)
    o.set_initialized()
    return o


def test_poisson_merge_init_and_intersection_size_estimation():
    """testPoissonMergeInit (:93-111): 8 distinct UMIs, cells 5 and 6 have 2 genes.
    testIntersectionSizeEstimation (:113-125) CANNOT pin the restatement: its expected values ('obtained with R':
    0.7264, 1.4484, 2.1380, 2.7923, 3.4346 for (1..5, 5)) are not produced by PoissonTargetEstimator.cpp:96-127 as
    written for ANY table of adjusted sizes (best integer fit is off by 0.16; they are close to g1 * sum p (1-(1-p)^8.4),
    i.e. an earlier estimator).  What is checked instead: the restatement equals an independent numpy evaluation of the
    formula in the code, sum_i (1 - (1-p_i)^a1) (1 - (1-p_i)^a2) with a = CollisionsAdjuster table."""
    o = _poisson_fixture()
    assert o.poisson_init() == 8
    rows = o.cell_rows()
    assert rows[5, 3] == 2 and rows[6, 3] == 2
    counts = np.array([1, 4, 2, 4, 1, 5, 2, 2], float)        # umi_distribution of the fixture (any order: a sum)
    p = counts / counts.sum()
    adj = ob.collisions_table(p, 8)
    for a, b in ((1, 5), (2, 5), (3, 5), (4, 5), (5, 5), (5, 3)):
        a1, a2 = sorted((int(adj[a - 1]), int(adj[b - 1])))
        want = float(((1 - (1 - p) ** a1) * (1 - (1 - p) ** a2)).sum())
        assert abs(o.poisson_gene_intersection(a, b) - want) < 1e-12
    assert o.poisson_gene_intersection(5, 3) == o.poisson_gene_intersection(3, 5)


def test_poisson_merge_probs_and_rejection():
    """testPoissonMergeProbs (:127-134) and testPoissonMergeRejections (:136-140).  Three of the four probabilities are
    within the reference's tolerances; the fourth ((5, 6): 0.05 +- 0.01 expected, 0.105 from the code's formula) goes
    with the intersection-size values above and is recorded, not asserted."""
    o = _poisson_fixture()
    o.poisson_init()
    assert o.poisson_intersection_prob(0, 1) == 1
    assert abs(o.poisson_intersection_prob(1, 2) - 0.16) <= 0.05
    assert abs(o.poisson_intersection_prob(3, 4) - 0.15) <= 0.05
    assert 0.04 <= o.poisson_intersection_prob(5, 6) <= 0.12
    assert o.poisson_merge_target(7) == -1


# The reference's own expectations that the restatement of the reference's CODE does not meet, kept with their literal
# values and tolerances as strict expected failures: if a future reading of PoissonTargetEstimator.cpp /
# CollisionsAdjuster.cpp ever reproduces them, these turn into XPASS(strict) = a test failure that forces the oracle to be
# re-examined.  Until then a15 (CollisionsAdjuster), a11 and the -M half of f3 are "parity unpinned" (DESIGN.md §4).
_REF_INTERSECTION_SIZES = [((1, 5), 0.7264), ((2, 5), 1.4484), ((3, 5), 2.1380), ((4, 5), 2.7923), ((5, 5), 3.4346), ((5, 3), 2.1380)]


@pytest.mark.parametrize("sizes,expected", _REF_INTERSECTION_SIZES)
@pytest.mark.xfail(strict=True, reason="Tests/TestEstimationMergeProbs.cpp:113-125 ('values were obtained with R'): not produced by "
                                       "PoissonTargetEstimator.cpp:96-127 + CollisionsAdjuster.cpp:21-39 as written (independent "
                                       "re-derivation: 0.775 / 1.944 / 2.385 / 3.332 / 3.916)")
def test_reference_intersection_size_expectations_literal(sizes, expected):
    """BOOST_CHECK_LE(abs(estimate_genes_intersection_size(a, b) - expected), 1e-2), literally."""
    o = _poisson_fixture()
    o.poisson_init()
    assert abs(o.poisson_gene_intersection(*sizes) - expected) <= 1e-2


@pytest.mark.xfail(strict=True, reason="Tests/TestEstimationMergeProbs.cpp:133: 0.05 +- 0.01 expected, the code's formula gives 0.105")
def test_reference_merge_prob_5_6_literal():
    """BOOST_CHECK_LE(abs(estimate_intersection_prob(container, 5, 6).merge_probability - 0.05), 0.01), literally."""
    o = _poisson_fixture()
    o.poisson_init()
    assert abs(o.poisson_intersection_prob(5, 6) - 0.05) <= 0.01


def test_poisson_upper_tail_against_scipy():
    """Rcpp::ppois(k - 1, lambda, lower = false) is R's; the restatement sums the pmf and is pinned on scipy here."""
    from scipy.stats import poisson
    rng = np.random.default_rng(4)
    for _ in range(400):
        lam = float(10 ** rng.uniform(-4, 3.5))
        k = int(rng.integers(0, 60)) if rng.random() < 0.5 else int(max(0, rng.normal(lam, 3 * np.sqrt(lam) + 2)))
        got, want = ob.poisson_upper_tail(k, lam), float(poisson.sf(k - 1, lam))
        assert abs(got - want) <= 1e-12 * max(want, 1e-300) + 1e-15 or abs(got - want) / max(want, 1e-300) < 1e-9, (k, lam, got, want)


# ---------------------------------------------------------------------------------------------------
# -g: Tools/GeneAnnotation (Tests/TestTools.cpp)
# ---------------------------------------------------------------------------------------------------
GTF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gtf_test.gtf.gz")   # the reference's data/gtf/gtf_test.gtf.gz


def test_gtf_record_parsing():
    """testGtf (Tests/TestTools.cpp:31-44)."""
    g = ob.GeneAnnotationOracle(GTF)
    line = ('chr1\tunknown\texon\t878633  878757  .       +       2       gene_id "SAMD11"; gene_name "SAMD11"; p_id "P11277"; '
            'transcript_id "NM_152486"; tss_id "TSS28354";')
    assert g.parse_gtf(line) == ("chr1", "SAMD11", 878632, 878757)


def test_intervals_container_merging():
    """testGeneMerge (:89-126)."""
    iv = ob.IntervalsOracle()
    for s, e in ((0, 100), (200, 300), (400, 500)):
        iv.add(s, e, "")
    iv.set_initialized(False)
    h = iv.homogeneous()
    assert len(h) == 3 and h[-1] == (400, 500)
    iv.add(90, 110, "", True); iv.set_initialized(False)
    assert iv.homogeneous()[0][1] == 110
    iv.add(150, 190, "", True); iv.set_initialized(False)
    assert len(iv.homogeneous()) == 4
    iv.add(110, 151, "", True); iv.set_initialized(False)
    h = iv.homogeneous()
    assert len(h) == 3 and h[0][1] == 190
    iv.add(190, 401, "", True); iv.set_initialized(False)
    assert iv.homogeneous() == [(0, 500)]


def test_intervals_container_queries():
    """testInterval (:244-264)."""
    iv = ob.IntervalsOracle()
    iv.add(10, 20, "i1"); iv.add(10, 20, "i1"); iv.add(15, 30, "i1"); iv.add(15, 20, "i2")
    iv.set_initialized()
    assert iv.query(0, 11) == ["i1"] and iv.query(0, 5) == [] and iv.query(25, 30) == ["i1"] and iv.query(17, 20) == ["i1", "i2"]


def test_gtf_container_init_and_intron_queries():
    """testInitGtf (:128-140) and testGenesWithIntrons (:266-287)."""
    g = ob.GeneAnnotationOracle(GTF)
    assert g.n_chromosomes() == 3
    assert g.n_homogeneous("chr1") == 8 and g.homogeneous_labels("chr1", 7) == 1 and g.n_homogeneous("chr2") == 5
    assert g.query("chr1", 20000, 20010) == [("WASH7P", "INTRON")]
    assert g.query("chr1", 24750, 24760) == [("WASH7P", "EXON")]
    assert g.query("chr1", 10, 20) == []
    assert g.query("chrNope", 10, 20) is None


def test_const_length_whitelists_of_three_parts():
    """ConstLengthBarcodesParser reads one part per line, any number of lines (ConstLengthBarcodesParser.cpp:50-68); the
    reference ships two three-part files (data/barcodes/split_seq for configs/split_seq.xml, 10x_v2_0_split).  The parts
    are reverse-complemented one by one and split_barcode cuts the barcode in file order (:33-48)."""
    o = Oracle()
    o.wl_load(1, os.path.join(DATA, "split_seq"))
    assert o.wl_parts() == 3 and [len(o.wl_part(p)) for p in range(3)] == [96, 96, 96]
    raw = open(os.path.join(DATA, "split_seq")).read().split("\n")[1].split()[0]
    assert o.wl_part(1)[0] == "".join({"A": "T", "C": "G", "G": "C", "T": "A"}[c] for c in reversed(raw))
    cb = o.wl_part(0)[5] + o.wl_part(1)[7] + o.wl_part(2)[11]
    assert o.wl_split(cb) == [o.wl_part(0)[5], o.wl_part(1)[7], o.wl_part(2)[11]]
    for p, i in ((0, 5), (1, 7), (2, 11)):
        vals, idx = o.wl_distances(cb, p)
        assert vals[0] == 0 and idx[0] == i and np.all(np.diff(vals) >= 0)      # sorted by distance, the exact entry first
    with pytest.raises(RuntimeError):
        o.wl_split(cb[:-1])                                                         # wrong total length
    o.wl_load(1, os.path.join(DATA, "10x_v2_0_split"))
    assert [len(o.wl_part(p)) for p in range(3)] == [384, 1248, 3840]
