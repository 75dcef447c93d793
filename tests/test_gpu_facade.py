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
