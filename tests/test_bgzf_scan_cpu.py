"""dropest_bgzf_scan (include/dropest_bgzf.h) is host code: the walk over the BGZF block headers of a written BAM, checked here without a
GPU -- payload offsets and lengths, ISIZE, the stored CRC-32, the stop at an incomplete block, the refusal of something that is not BGZF.
(The inflate itself has no CPU implementation: tests/test_gpu_bgzf.py.)"""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from dropest_amd import capi

import bam_writer as bw

P = C.POINTER


def scan(blob, cap=None):
    L = capi.lib()
    L.dropest_bgzf_scan.restype = C.c_int
    L.dropest_bgzf_scan.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)]
    L.dropest_bgzf_last_error.restype = C.c_char_p
    cap = cap if cap is not None else len(blob) // 26 + 1
    src = np.frombuffer(blob, np.uint8)
    in_off, out_off = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
    in_len, out_len, crc = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    n, used, total = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = L.dropest_bgzf_scan(src.ctypes.data, len(blob), cap, in_off.ctypes.data, in_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, crc.ctypes.data,
                             C.byref(n), C.byref(used), C.byref(total))
    if rc:
        raise RuntimeError(L.dropest_bgzf_last_error().decode())
    k = n.value
    return in_off[:k], in_len[:k], out_off[:k], out_len[:k], crc[:k], used.value, total.value


def test_scan_of_a_written_bam(tmp_path):
    rng = np.random.default_rng(2)
    recs = [bw.record(int(rng.integers(0, 3)), i, "r%d" % i, seq="ACGT" * 12, tags=[("CB", "Z", "ACGTACGTAC"), ("UB", "Z", "ACGTAA")]) for i in range(5000)]
    path = str(tmp_path / "s.bam")
    bw.write_bam(path, [("chr%d" % i, 1000) for i in range(3)], recs, block=9000)
    blob = open(path, "rb").read()
    in_off, in_len, out_off, out_len, crc, used, total = scan(blob)
    assert used == len(blob) and len(in_off) > 50 and out_len[-1] == 0            # the EOF marker block
    pieces = [zlib.decompress(blob[int(o):int(o) + int(n)], -15) for o, n in zip(in_off, in_len)]
    assert [len(p) for p in pieces] == out_len.tolist() and total == sum(len(p) for p in pieces)
    assert [zlib.crc32(p) & 0xFFFFFFFF for p in pieces] == crc.tolist()
    assert out_off.tolist() == np.concatenate([[0], np.cumsum(out_len[:-1], dtype=np.uint64)]).tolist()
    # a buffer that ends inside a block: the whole blocks before it, and where they end
    cut = int(in_off[7]) + 5
    a = scan(blob[:cut])
    assert len(a[0]) == 7 and a[5] == int(in_off[7]) - 18
    assert len(scan(blob, cap=3)[0]) == 3
    # an extra subfield in front of BC (the header is then longer than 18 bytes)
    first = blob[:int(in_off[0]) + int(in_len[0]) + 8]
    bsize = len(first) + 8
    odd = first[:10] + struct.pack("<H", 14) + b"XY" + struct.pack("<H", 4) + b"abcd" + b"BC" + struct.pack("<HH", 2, bsize - 1) + first[18:]
    b = scan(odd)
    assert len(b[0]) == 1 and int(b[0][0]) == 26 and int(b[1][0]) == int(in_len[0]) and b[5] == len(odd)
    with pytest.raises(RuntimeError):
        scan(b"definitely not a BGZF block header..........")
