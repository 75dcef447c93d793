"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the headers declare,
fails loudly without a GPU, and the product package never touches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.synth import SynthStream, load_whitelist, reverse_complement

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dropest_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = _declared("dropest_amd.h") + _declared("dropest_synth.h")
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    assert sorted(names) == sorted(capi.EXPORTED_SYMBOLS)
    for n in _declared("dropest_annotation.h"):          # the device gene annotation (bound by tests/test_gpu_annotation.py)
        assert hasattr(L, n), "missing export: " + n
    for n in _declared("dropest_bgzf.h"):                # BGZF blocks inflated on the device (bound by tests/test_gpu_bgzf.py)
        assert hasattr(L, n), "missing export: " + n


def test_cfg_defaults_match_reference_defaults():
    """MergeStrategyFactory.cpp:26-58 defaults."""
    cfg = capi.Cfg()
    capi.lib().dropest_cfg_defaults(C.byref(cfg))
    assert cfg.min_genes_before_merge == 10 and cfg.min_genes_after_merge == 10
    assert cfg.min_merge_fraction == 0.2 and cfg.max_umi_merge_edit_distance == 1
    assert cfg.gene_match_levels == b"eEBA" and cfg.max_cells == -1


@pytest.mark.skipif(capi.lib().dropest_dev_count() > 0, reason="a GPU is present")
def test_fails_loudly_without_gpu():
    with pytest.raises(capi.DropestError) as e:
        capi.Context()
    assert e.value.status == 3 and "no CPU implementation" in str(e.value)


def test_product_package_never_references_the_oracle():
    pkg = os.path.join(ROOT, "dropest_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % os.path.join(base, f)


def test_pack_roundtrip_and_order():
    for s in ["A", "ACGT", "TTTTTTTTTTTTTTTT", "GATTACAGATTACAGATTACAGATTACAGAT"]:
        assert capi.unpack_code(capi.pack_seq(s)) == s
    assert capi.pack_seq("ACGN") is None and capi.pack_seq("") is None and capi.pack_seq("A" * 32) is None
    seqs = sorted(["ACGT", "AAAA", "TTTT", "CAGT", "ACGA"])
    assert sorted(seqs, key=capi.pack_seq) == seqs      # equal-length codes sort like the strings


def test_host_generator_is_deterministic_and_shaped():
    s = SynthStream(n_reads=200_000, n_cells=50, n_genes=2000)
    a = s.generate_host(0, 50_000)
    b = s.generate_host(0, 50_000)
    c = s.generate_host(10_000, 1000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for x, y in zip(a, c):
        assert np.array_equal(x[10_000:11_000], y)       # read i depends on i only
    cb, umi, gene, aux = a
    frac_ig = float((gene == capi.NO_GENE).mean())
    assert 0.05 < frac_ig < 0.09
    assert set(np.unique(aux >> 16)) <= {2, 3, 4}
    real = np.isin(cb, s.cell_cb)
    assert 0.90 < real.mean() < 0.94
    assert all(len(capi.unpack_code(x)) == 16 for x in cb[:100])
    assert all(len(capi.unpack_code(x)) == 10 for x in umi[:100])


def test_whitelist_loader_matches_reference_fixture():
    """Tests/TestEstimation.cpp:98-121: reverse-complemented parts of data/barcodes/test_est."""
    parts = load_whitelist(os.path.join(ROOT, "dropest_amd", "data", "barcodes", "test_est"))
    assert parts[0] == ["AAT", "GAA", "AAA"] and parts[1] == ["TTAGGTCCA", "TTAGGGGCC", "TTAGGTCCC"]
    assert reverse_complement("AACG") == "CGTT"


def test_radix_plan_covers_the_varying_bits_with_the_fewest_passes():
    """plan_radix_passes (dropest_amd.hip): windows ascend without overlap, every varying bit is inside one, digits are 8
    bits wide except the top ones, which widen to 9 when that saves a whole pass (C2: 57 bits -> 7, C4: 53 -> 6)."""
    import random
    from dropest_amd import capi
    def check(mask):
        plan = capi.radix_plan(mask)
        covered = 0
        at = -1
        for shift, bits in plan:
            assert bits in (8, 9) and at < shift < 64
            assert shift + bits <= 64 or bits == 8          # a plain top window may hang over bit 63 (it reads zeros there)
            at = shift + bits - 1
            covered |= ((1 << bits) - 1) << shift
        assert (mask & ~covered & ((1 << 64) - 1)) == 0
        if mask:
            lo = (mask & -mask).bit_length() - 1
            width = mask.bit_length() - lo
            plain = sum(1 for s in range(lo, 64, 8) if (mask >> s) & 0xFF)
            assert len(plan) == min(plain, -(-width // 9)) or len(plan) == plain
            assert len(plan) <= plain
        return plan
    assert check(0) == []
    assert [b for _, b in check(((1 << 57) - 1) << 3)] == [8] * 6 + [9]                 # C2 key under 3 mark bits
    assert [b for _, b in check(((1 << 53) - 1) << 3)] == [8] + [9] * 5                 # C4
    assert [b for _, b in check((1 << 64) - 1)] == [8] * 8                              # C3 at 1e9 reads
    assert check(0xFF << 32) == [(32, 8)]
    assert len(check((0xFF << 48) | 0xFF)) == 2                                         # constant windows are skipped
    rng = random.Random(7)
    for _ in range(2000):
        lo, width = rng.randrange(0, 60), rng.randrange(1, 65)
        width = min(width, 64 - lo)
        m = ((1 << width) - 1) << lo
        if rng.random() < 0.5:
            m &= rng.getrandbits(64) | (1 << lo) | (1 << (lo + width - 1))
        check(m)


def test_restated_glibc_rand_equals_libc():
    """The context's own generator (context.h: GlibcRand) against libc's srand / rand: MergeUMIsStrategySimple seeds 42
    (MergeUMIsStrategySimple.cpp:15-19), a fresh process starts from 1; srand(0) is srand(1)."""
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    for seed in (42, 1, 0, 7, 20260928, 0xFFFFFFFF):
        got = np.zeros(5000, np.int32)
        assert capi.lib().dropest_rand_sequence(seed, len(got), got.ctypes.data) == 0
        libc.srand(C.c_uint(seed))
        want = np.array([libc.rand() for _ in range(len(got))], np.int32)
        assert np.array_equal(got, want), seed
