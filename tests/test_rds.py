"""The native .rds writer (dropest_amd/csrc/host/rds_writer.cpp; ResultsPrinter::save_rds, ResultsPrinter.cpp:442-452).
The reference pins nothing about the .rds ("parity unpinned"), so the check is semantic: the file is parsed
back with a reader that is itself pinned on files written by R (tests/golden/*.rds from the reference's data/)."""
import os
import subprocess

import numpy as np

import rds_reader as rr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_reader_on_files_written_by_r():
    v = rr.read_rds(os.path.join(GOLDEN, "mit_genes_human.rds"))
    assert v.kind == "character" and len(v.value) == 23 and v.value[0] == "uc004coq.4"
    df = rr.read_rds(os.path.join(GOLDEN, "SRR1784310_curve.rds"))
    assert df.kind == "list" and df.is_object and list(df.attributes["class"].value) == ["data.frame"]
    assert df.names == ["sample.size", "umigs_count"]
    assert df["sample.size"].kind == "double" and df["sample.size"].value[0] == 467174.0
    assert len(df["sample.size"].value) == len(df["umigs_count"].value) > 100
    assert np.all(np.diff(df["umigs_count"].value) > 0)


def _check_long_vectors(base):
    """The file of long vectors (tests/cpp/test_rds_writer.cpp) written by one thread and by eight: several gzip members each, the same
    serialisation, every value in place."""
    one, many = open(base + ".one.rds", "rb").read(), open(base + ".many.rds", "rb").read()
    assert one == many                                            # the pieces do not depend on who deflates them
    import zlib
    members, rest = 0, many
    while rest:                                                   # concatenated gzip members (RFC 1952 2.2), as gzfile() reads them
        z = zlib.decompressobj(31)
        z.decompress(rest)
        assert z.eof
        rest = z.unused_data
        members += 1
    assert members > 5
    d = rr.read_rds(base + ".many.rds")
    assert d.names == ["cm", "ints", "reals", "strs", "packed", "tail"]
    pk = d["packed"].value
    assert len(pk) == 200003
    for j in (0, 1, 31, 32, 33, 4097, 100000, 200002):
        assert pk[j] == "".join("ACGT"[(j >> (b % 17)) & 3] for b in range(j % 32)), j
    n = 700001
    colptr = np.arange(1001, dtype=np.uint64) * n // 1000
    k = np.arange(n, dtype=np.uint64)
    m = d["cm"]
    assert m["p"].value.tolist() == colptr.tolist() and m["i"].kind == "integer" and m["x"].kind == "double"
    assert np.array_equal(m["i"].value, (k - np.repeat(colptr[:-1], np.diff(colptr).astype(np.int64))).astype(np.int64))
    assert np.array_equal(m["x"].value, ((((k * 2654435761) & 0xFFFFFFFF) >> 12) + 1).astype(np.float64))
    assert m["Dim"].value.tolist() == [701, 1000] and m["Dimnames"].value[1].value[999] == "C999"
    ki = np.arange(1000003, dtype=np.uint64)
    assert np.array_equal(d["ints"].value, (((ki * 7919) & 0xFFFFFFFF).astype(np.uint32).astype(np.int32).astype(np.int64) - 1000000).astype(np.int32))
    assert np.array_equal(d["reals"].value, np.arange(500009, dtype=np.float64) * 0.37 - 11.0)
    strs = d["strs"].value
    assert len(strs) == 300001 and all(strs[j] == "ABCD"[j % 4] * (j % 37) + str(j) for j in (0, 1, 36, 37, 4095, 4096, 150000, 300000))
    assert d["tail"].value.tolist() == [1, 2, 3]


def test_writer_round_trip(tmp_path):
    exe, out = str(tmp_path / "w"), str(tmp_path / "t.rds")
    src = [os.path.join(ROOT, "tests", "cpp", "test_rds_writer.cpp"), os.path.join(ROOT, "dropest_amd", "csrc", "host", "rds_writer.cpp")]
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall"] + src + ["-o", exe, "-lz", "-pthread"])
    assert subprocess.run([exe, out, str(tmp_path / "big")], capture_output=True, text=True).stdout.strip() == "ok"
    _check_long_vectors(str(tmp_path / "big"))
    raw = open(out, "rb").read()
    assert raw[:2] == b"\x1f\x8b"                                  # gzip container, like saveRDS
    data = rr.decompress(raw)
    assert data[:2] == b"X\n" and data[2:6] == b"\x00\x00\x00\x02"  # XDR, version 2
    d = rr.read_rds(out)
    assert d.names == ["cm", "frame", "named_int", "named_real", "chars", "nested", "targets", "empty"]
    dense, rows, cols = rr.dgcmatrix_to_dense(d["cm"])
    assert rows == ["g0", "g1", "g2", "g3"] and cols == ["AAAC", "AAAG", "AAAT"]
    assert dense.tolist() == [[7, 0, 0], [0, 0, 2], [0, 0, 300000], [1, 0, 5]]
    m = d["cm"]
    assert m.is_object and m["x"].kind == "double" and m["i"].kind == "integer" and m["factors"].kind == "list"
    assert list(m["class"].attributes["package"].value) == ["Matrix"]
    f = d["frame"]
    assert list(f.attributes["class"].value) == ["data.frame"] and f.names == ["chr1", "chrX"]
    assert list(f.attributes["row.names"].value) == ["AAAC", "AAAT"] and f["chrX"].value.tolist() == [30, 40]
    assert d["named_int"].names == ["a", "b", "c"] and d["named_int"].value.tolist() == [5, -7, 2147483647]
    assert d["named_real"].value[:2].tolist() == [1.5, -0.25] and np.isnan(d["named_real"].value[2])
    assert d["chars"].value == ["ACGT", "", "géne"]
    assert d["nested"].value[0].value[0].value.tolist() == [3] and len(d["nested"].value[0].value[1].value) == 0
    assert d["nested"].value[1].kind == "NULL"
    assert d["targets"].names == ["AAAG"] and d["targets"].value[0].value == ["AAAC"]
    assert d["empty"].kind == "list" and len(d["empty"].value) == 0
