"""Minimal BAM writer (SAMv1 §4: BGZF container + BAM records), TEST INFRASTRUCTURE for the native BAM reader
(dropest_amd/csrc/host/bam_ingest.cpp).  Records may straddle BGZF blocks (the byte stream is cut at a fixed size)."""
import struct
import zlib

_SEQ = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def _bgzf_block(data, level=6):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
            + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _tag(tag, typ, value):
    t = tag.encode() + typ.encode()
    if typ == "Z":
        return t + value.encode() + b"\x00"
    if typ == "A":
        return t + value.encode()[:1]
    if typ == "i":
        return t + struct.pack("<i", value)
    if typ == "C":
        return t + struct.pack("<B", value)
    if typ == "f":
        return t + struct.pack("<f", value)
    if typ == "B":                     # value = list of int16
        return t + b"s" + struct.pack("<I", len(value)) + struct.pack("<%dh" % len(value), *value)
    raise ValueError(typ)


_CIGAR = {c: i for i, c in enumerate("MIDNSHP=X")}


def record(ref_id, pos, name, flag=0, mapq=255, seq="ACGT" * 10, tags=(), cigar=None):
    """cigar: list of (length, op) -- default one M over the whole read"""
    n = len(seq)
    ops = cigar or [(n, "M")]
    cigar = b"".join(struct.pack("<I", (ln << 4) | _CIGAR[op]) for ln, op in ops)
    packed = bytearray((n + 1) // 2)
    for i, c in enumerate(seq):
        packed[i // 2] |= _SEQ[c] << (4 if i % 2 == 0 else 0)
    body = struct.pack("<iiBBHHHIiii", ref_id, pos, len(name) + 1, mapq, 4680, len(ops), flag, n, -1, -1, 0)
    body += name.encode() + b"\x00" + cigar + bytes(packed) + b"\xff" * n + b"".join(_tag(*t) for t in tags)
    return struct.pack("<I", len(body)) + body


def write_bam(path, refs, records, block=0xFF00, header_text="@HD\tVN:1.6\tSO:unsorted\n", repeat=1):
    text = header_text + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, ln) for n, ln in refs)
    raw = b"BAM\x01" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(refs))
    for n, ln in refs:
        raw += struct.pack("<I", len(n) + 1) + n.encode() + b"\x00" + struct.pack("<I", ln)
    if repeat > 1:      # the records `repeat` times over (benchmarks): the header in blocks of its own, the records' blocks written again and again
        body = b"".join(records)
        blocks = [_bgzf_block(body[o:o + block]) for o in range(0, len(body), block)]
        with open(path, "wb") as f:
            for o in range(0, len(raw), block):
                f.write(_bgzf_block(raw[o:o + block]))
            for _ in range(repeat):
                for b in blocks:
                    f.write(b)
            f.write(_bgzf_block(b""))
        return
    raw += b"".join(records)
    with open(path, "wb") as f:
        for o in range(0, len(raw), block):
            f.write(_bgzf_block(raw[o:o + block]))
        f.write(_bgzf_block(b""))       # EOF marker block
