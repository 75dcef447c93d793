"""BGZF blocks inflated (and their CRC-32 checked) on the device (include/dropest_bgzf.h, csrc/k_inflate.h: one wave per block) against zlib: every block type
(stored, fixed code, dynamic code), every level, data that is random, repetitive (matches longer than their distance), BAM-like and empty;
several DEFLATE blocks inside one BGZF block; damaged streams are refused block by block and nothing else is touched."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

from dropest_amd import capi

import bam_writer as bw

pytestmark = pytest.mark.gpu
P = C.POINTER


def bgzf_block(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    if flush_every:      # several DEFLATE blocks inside the one BGZF block
        cdata = b"".join(comp.compress(data[o:o + flush_every]) + comp.flush(zlib.Z_FULL_FLUSH) for o in range(0, len(data), flush_every)) + comp.flush()
    else:
        cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
            + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def inflate(blob, repeats=1):
    L = C.CDLL(os.environ["DROPEST_BGZF_LIB"]) if os.environ.get("DROPEST_BGZF_LIB") else capi.lib()      # (a variant build of csrc/bgzf_api.hip: kernel experiments)
    L.dropest_bgzf_inflate_buffer.restype = C.c_int
    L.dropest_bgzf_inflate_buffer.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, P(C.c_uint64), C.c_void_p, C.c_uint64, P(C.c_uint64),
                                              P(C.c_double), C.c_int]
    L.dropest_bgzf_last_error.restype = C.c_char_p
    src = np.frombuffer(blob, np.uint8)
    cap = max(1, len(blob) // 26 + 1)
    out = np.zeros(min(cap * 65536, max(len(blob) * 64, 1 << 22)), np.uint8)      # (pages are touched only where the copy lands)
    status = np.full(cap, 77, np.uint32)
    n_out, n_blocks, ms = C.c_uint64(), C.c_uint64(), C.c_double()
    rc = L.dropest_bgzf_inflate_buffer(0, src.ctypes.data, len(blob), out.ctypes.data, len(out), C.byref(n_out), status.ctypes.data, cap, C.byref(n_blocks),
                                       C.byref(ms), repeats)
    if rc:
        raise RuntimeError(L.dropest_bgzf_last_error().decode())
    return out[:n_out.value].tobytes(), status[:n_blocks.value].copy(), ms.value


def kinds(rng):
    bam_like = b"".join(bw.record(int(rng.integers(0, 25)), int(rng.integers(0, 1 << 28)), "A00000:1:HXXXX:1:1101:%d:%d" % (i, i * 7), seq="ACGT" * 24 + "AC",
                                  tags=[("CB", "Z", "".join(rng.choice(list("ACGT"), 16)) + "-1"), ("UB", "Z", "".join(rng.choice(list("ACGT"), 10))),
                                        ("GX", "Z", "ENSG%011d" % int(rng.integers(0, 3000))), ("NH", "i", 1)]) for i in range(220))
    return {
        "random": rng.integers(0, 256, 60_000, dtype=np.uint8).tobytes(),
        "few_symbols": rng.choice(np.frombuffer(b"ACGT\n", np.uint8), 65_000).tobytes(),
        "runs": b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 700)) for _ in range(180))[:65_000],
        "period_3_and_7": (b"abc" * 9000 + b"0123456" * 4000)[:65_000],
        "text": (b"the quick brown fox jumps over the lazy dog; " * 1500)[:64_000],
        "bam_like": bam_like[:65_000],
        "one_byte": b"x",
        "empty": b"",
        "skewed": rng.choice(256, 65_000, p=np.r_[np.full(8, 0.11), np.full(248, 0.12 / 248)]).astype(np.uint8).tobytes(),   # long codes for the rare bytes
    }


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_every_level_and_kind_of_data(level):
    rng = np.random.default_rng(level)
    data = kinds(rng)
    blob = b"".join(bgzf_block(d if level or len(d) < 65_000 else d[:65_000], level) for d in data.values()) + bgzf_block(b"")
    out, status, _ = inflate(blob)
    assert len(status) == len(data) + 1 and not status.any(), status
    assert out == b"".join(data.values())


def test_fixed_codes_huffman_only_rle_and_several_deflate_blocks():
    rng = np.random.default_rng(11)
    data = kinds(rng)
    blocks, want = [], []
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        for name, d in data.items():
            d = d[:40_000] if strategy == zlib.Z_FIXED and name in ("random", "skewed") else d     # (fixed codes expand random bytes)
            blocks.append(bgzf_block(d, 6, strategy)); want.append(d)
    for name, d in data.items():
        blocks.append(bgzf_block(d, 6, flush_every=5000)); want.append(d)       # ~13 DEFLATE blocks, stored empty blocks in between
    out, status, _ = inflate(b"".join(blocks))
    assert not status.any(), np.flatnonzero(status)
    assert out == b"".join(want)


def test_a_written_bam_file(tmp_path):
    rng = np.random.default_rng(5)
    recs = [bw.record(int(rng.integers(0, 5)), i * 3, "read%d" % i, seq="ACGT" * 20, tags=[("CB", "Z", "ACGTACGTACGTAC"), ("UB", "Z", "ACGTAC"), ("GX", "Z", "G%d" % (i % 50))])
            for i in range(20_000)]
    path = str(tmp_path / "t.bam")
    bw.write_bam(path, [("chr%d" % i, 1000) for i in range(5)], recs, block=0xFF00)
    blob = open(path, "rb").read()
    out, status, _ = inflate(blob)
    assert not status.any() and len(status) > 40
    import gzip
    assert out == gzip.decompress(blob)


def test_damaged_blocks_are_refused_one_by_one():
    rng = np.random.default_rng(3)
    good = [rng.choice(np.frombuffer(b"ACGTN", np.uint8), 30_000).tobytes() for _ in range(6)]
    blocks = [bytearray(bgzf_block(d)) for d in good]
    blocks[1][40] ^= 0x5A                                     # inside the DEFLATE payload
    blocks[3][-4:] = struct.pack("<I", 29_999)                # ISIZE one short
    blocks[4] = bytearray(blocks[4][:18]) + bytearray(b"\x07") + blocks[4][19:]     # block type 3
    out, status, _ = inflate(b"".join(bytes(b) for b in blocks))
    assert status[0] == 0 and status[2] == 0 and status[5] == 0
    assert status[1] != 0 and status[3] != 0 and status[4] != 0      # (a flipped payload byte that still decodes is caught by the block's CRC-32)
    blocks[1] = bytearray(bgzf_block(good[1])); blocks[1][-8] ^= 1   # a good payload under a wrong stored CRC
    _, status2, _ = inflate(b"".join(bytes(b) for b in blocks[:3]))
    assert list(status2) == [0, 10, 0]
    off = np.cumsum([0] + [30_000, 30_000, 30_000, 29_999, 30_000, 30_000])
    for k in (0, 2, 5):
        assert out[off[k]:off[k + 1]] == good[k]
    with pytest.raises(RuntimeError):
        inflate(b"not a bgzf file at all, not even close........")


def test_literals_waiting_before_a_stored_block_stay_inside_the_blocks_own_range():
    """A non-final literal-only DEFLATE block leaves up to 63 literals waiting for their store; the stored block behind it flushes them.  With an
    ISIZE smaller than the stream (a damaged or crafted block) that flush must be refused like every other store beyond the block's own output
    range (it ran before the capacity test and wrote into the neighbour's bytes)."""
    rng = np.random.default_rng(17)
    good = [rng.choice(np.frombuffer(b"ACGTN", np.uint8), 20_000).tobytes() for _ in range(2)]
    lits = rng.integers(0, 256, 40, dtype=np.uint8).tobytes()
    tail = rng.integers(0, 256, 300, dtype=np.uint8).tobytes()
    comp = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_HUFFMAN_ONLY)
    cdata = comp.compress(lits) + comp.flush(zlib.Z_FULL_FLUSH)          # a literal-only block, then an empty STORED block (00 00 ff ff)
    comp0 = zlib.compressobj(0, zlib.DEFLATED, -15)
    cdata += comp0.compress(tail) + comp0.flush()                        # ... and a stored block with bytes, final
    assert zlib.decompress(cdata, -15) == lits + tail

    def block(isize, crc):
        return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, len(cdata) + 25)
                + cdata + struct.pack("<II", crc, isize))
    whole = block(len(lits) + len(tail), zlib.crc32(lits + tail) & 0xFFFFFFFF)
    out, status, _ = inflate(bgzf_block(good[0]) + whole + bgzf_block(good[1]))
    assert list(status) == [0, 0, 0] and out == good[0] + lits + tail + good[1]
    for isize in (10, 39, 41):                                           # shorter than the waiting literals / than literals + stored bytes
        for _ in range(5):                                               # (which wave stores first is not decided: a few tries)
            out, status, _ = inflate(bgzf_block(good[0]) + block(isize, 0) + bgzf_block(good[1]))
            assert status[0] == 0 and status[2] == 0 and status[1] != 0
            assert out[:20_000] == good[0] and out[20_000 + isize:] == good[1]


def test_fuzzed_payloads_end_with_a_verdict_and_touch_nothing_else():
    """600 blocks with one to three random bytes of their DEFLATE payload changed, between untouched blocks: every block gets a status, an
    untouched block comes out whole, a block that reports 0 after all (the change hit bits that do not matter, or a stored block's bytes
    AND its CRC agree again: impossible here) equals zlib's output -- and the kernel comes back (every loop of the decoder is bounded by
    the output or the input it has)."""
    rng = np.random.default_rng(99)
    data = kinds(rng)
    pool = [d for d in data.values() if len(d) > 1000]
    blocks, want, touched = [], [], []
    for k in range(900):
        d = pool[k % len(pool)][: int(rng.integers(1000, 65_000))]
        level = int(rng.choice([0, 1, 6, 9]))
        b = bytearray(bgzf_block(d, level, zlib.Z_FIXED if k % 11 == 5 else zlib.Z_DEFAULT_STRATEGY))
        hit = k % 3 != 0
        if hit:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(18, len(b) - 8))] = int(rng.integers(0, 256))
        blocks.append(bytes(b)); want.append(d); touched.append(hit)
    out, status, _ = inflate(b"".join(blocks))
    assert len(status) == 900
    off = np.concatenate([[0], np.cumsum([len(d) for d in want])])
    n_refused = 0
    for k in range(900):
        if not touched[k]:
            assert status[k] == 0 and out[off[k]:off[k + 1]] == want[k], k
        elif status[k] == 0:
            assert out[off[k]:off[k + 1]] == want[k], k         # the CRC-32 agreed: the bytes are the original ones
        else:
            n_refused += 1
    assert n_refused > 500
