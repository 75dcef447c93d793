"""The BGZF block decoder of the BAM reader (dropest_amd/csrc/host/fast_inflate.h) against zlib: raw DEFLATE streams of every block type
(stored, fixed, dynamic Huffman), every level, data from incompressible to one repeated byte, BAM-like records; truncated and corrupted
streams must be refused, never decoded into something else (the reader then falls back to zlib and checks the block's CRC-32 anyway)."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from dropest_amd.build import FACADE_LIB, build_facade


@pytest.fixture(scope="module")
def lib():
    build_facade()
    L = C.CDLL(FACADE_LIB)
    L.dropest_test_fast_inflate.restype = C.c_int
    L.dropest_test_fast_inflate.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.dropest_test_crc32.restype = C.c_uint32
    L.dropest_test_crc32.argtypes = [C.c_char_p, C.c_uint64]
    return L


def inflate(L, comp, n):
    out = np.full(n + 16, 0xAB, np.uint8)                       # (the decoder may not write past n: checked below)
    ok = L.dropest_test_fast_inflate(comp, len(comp), out.ctypes.data, n)
    assert (out[n:] == 0xAB).all()
    return bool(ok), out[:n].tobytes()


def deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for i in range(0, len(data), flush_every):                  # several blocks in one stream (full flushes put stored-block markers in)
        out += c.compress(data[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH if (i // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
    return out + c.flush()


def samples():
    rng = np.random.default_rng(7)
    yield "empty", b""
    yield "one byte", b"x"
    yield "random", rng.integers(0, 256, 65280, dtype=np.uint8).tobytes()
    yield "zeros", bytes(65280)
    yield "two symbols", rng.integers(0, 2, 60000, dtype=np.uint8).tobytes()
    yield "skewed", np.minimum(rng.geometric(0.05, 65000), 255).astype(np.uint8).tobytes()       # long Huffman codes for the rare bytes
    yield "short period", (b"ACGTTGCA" * 8000)[:65000]
    yield "period 1..9", b"".join(bytes([65 + k]) * (k + 1) for k in range(9)) * 1500
    rec = []
    for i in range(700):                                                                              # BAM-like: binary header, name, tags
        rec.append(rng.integers(0, 256, 36, dtype=np.uint8).tobytes() + b"read.%07d\x00" % i + b"CBZ" + bytes(rng.choice(list(b"ACGT"), 16)) +
                   b"\x00UBZ" + bytes(rng.choice(list(b"ACGT"), 10)) + b"\x00GXZENSG%011d\x00" % int(rng.integers(0, 30000)))
    yield "bam-like", b"".join(rec)[:65280]
    yield "text", (b"the quick brown fox jumps over the lazy dog; " * 2000)[:64000]


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_every_level_and_kind_of_data(lib, level):
    for name, data in samples():
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
            for flush_every in (0, 5000):
                comp = deflate(data, level, strategy, flush_every)
                ok, got = inflate(lib, comp, len(data))
                assert ok and got == data, (name, level, strategy, flush_every)
                assert lib.dropest_test_crc32(data, len(data)) == zlib.crc32(data), name


def test_wrong_sizes_and_damaged_streams_are_refused(lib):
    rng = np.random.default_rng(3)
    for name, data in samples():
        if len(data) < 100:
            continue
        comp = deflate(data, 6)
        assert not inflate(lib, comp, len(data) - 1)[0], name                 # more output than asked for
        assert not inflate(lib, comp, len(data) + 1)[0], name                 # less
        assert not inflate(lib, comp[:len(comp) // 2], len(data))[0], name    # truncated input
        for _ in range(50):                                                     # flipped bits: refused, or decoded to something the CRC catches -- never a crash
            bad = bytearray(comp)
            k = int(rng.integers(0, len(bad)))
            bad[k] ^= 1 << int(rng.integers(0, 8))
            ok, got = inflate(lib, bytes(bad), len(data))
            if ok and got != data:
                assert zlib.crc32(got) != zlib.crc32(data)
    for n in (0, 1, 2, 3, 7, 64):                                               # garbage of every small length
        for _ in range(200):
            junk = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            inflate(lib, junk, 100)


def test_crc32_of_odd_lengths(lib):
    rng = np.random.default_rng(5)
    for n in list(range(0, 40)) + [255, 256, 257, 65279, 65280]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert lib.dropest_test_crc32(d, n) == zlib.crc32(d)
