"""Minimal reader of R's serialisation format (readRDS), TEST INFRASTRUCTURE: checks the native .rds writer
(dropest_amd/csrc/host/rds_writer.cpp).  It is itself pinned on files written by R (tests/golden/*.rds, taken from the
reference's data/ directory), so writer and reader are not merely consistent with each other.

Supports XDR format, versions 2 and 3; gzip / bzip2 / xz containers; the node types of plain data: NULL, symbols,
pairlists, logical / integer / double / character vectors, lists, S4 objects, references, attributes."""
import bz2
import gzip
import lzma
import struct

import numpy as np

NILVALUE, REFSXP, SYMSXP, LISTSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP, S4SXP = 254, 255, 1, 2, 9, 10, 13, 14, 16, 19, 25
NA_INT = -2147483648


class RObject:
    def __init__(self, kind, value=None, attributes=None, is_object=False):
        self.kind, self.value, self.attributes, self.is_object = kind, value, attributes or {}, is_object

    @property
    def names(self):
        n = self.attributes.get("names")
        return None if n is None else list(n.value)

    def __getitem__(self, key):
        if isinstance(key, str):
            if self.kind == "S4":
                return self.attributes[key]
            return self.value[self.names.index(key)]
        return self.value[key]

    def __repr__(self):
        return "RObject(%s, %d attrs)" % (self.kind, len(self.attributes))


class _Reader:
    def __init__(self, data):
        self.d, self.pos, self.refs = data, 0, []

    def i32(self):
        v = struct.unpack_from(">i", self.d, self.pos)[0]
        self.pos += 4
        return v

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        is_obj, has_attr, has_tag = bool(flags & 0x100), bool(flags & 0x200), bool(flags & 0x400)
        if t == NILVALUE:
            return RObject("NULL")
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == SYMSXP:
            name = self.item()
            sym = RObject("symbol", name.value)
            self.refs.append(sym)
            return sym
        if t == LISTSXP:
            # pairlist: [attributes] [tag] car cdr; flattened into an ordered dict tag -> value
            out = {}
            while True:
                if has_attr:
                    self.item()
                tag = self.item().value if has_tag else None
                out[tag if tag is not None else len(out)] = self.item()
                flags = self.i32()
                t2 = flags & 0xFF
                if t2 == NILVALUE:
                    break
                if t2 != LISTSXP:
                    raise ValueError("unexpected cdr type %d" % t2)
                has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
            return RObject("pairlist", out)
        if t == CHARSXP:
            n = self.i32()
            if n == -1:
                return RObject("char", None)
            s = self.d[self.pos:self.pos + n].decode("utf-8", "replace")
            self.pos += n
            return RObject("char", s)
        if t in (LGLSXP, INTSXP):
            n = self.i32()
            v = np.frombuffer(self.d, ">i4", n, self.pos).astype(np.int64)
            self.pos += 4 * n
            obj = RObject("logical" if t == LGLSXP else "integer", v, is_object=is_obj)
        elif t == REALSXP:
            n = self.i32()
            v = np.frombuffer(self.d, ">f8", n, self.pos).astype(np.float64)
            self.pos += 8 * n
            obj = RObject("double", v, is_object=is_obj)
        elif t == STRSXP:
            n = self.i32()
            obj = RObject("character", [self.item().value for _ in range(n)], is_object=is_obj)
        elif t == VECSXP:
            n = self.i32()
            obj = RObject("list", [self.item() for _ in range(n)], is_object=is_obj)
        elif t == S4SXP:
            obj = RObject("S4", None, is_object=is_obj)
        else:
            raise ValueError("unsupported SEXPTYPE %d at byte %d" % (t, self.pos))
        if has_attr:
            attrs = self.item()
            obj.attributes = dict(attrs.value) if attrs.kind == "pairlist" else {}
        return obj


def decompress(raw):
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    if raw[:3] == b"BZh":
        return bz2.decompress(raw)
    if raw[:6] == b"\xfd7zXZ\x00":
        return lzma.decompress(raw)
    return raw


def read_rds(path):
    data = decompress(open(path, "rb").read())
    if data[:2] != b"X\n":
        raise ValueError("not an XDR serialisation")
    r = _Reader(data)
    r.pos = 2
    version = r.i32()
    r.i32(); r.i32()                      # writer version, minimal reader version
    if version == 3:
        n = r.i32()
        r.pos += n                        # native encoding name
    elif version != 2:
        raise ValueError("unsupported serialisation version %d" % version)
    obj = r.item()
    if r.pos != len(data):
        raise ValueError("trailing bytes: %d of %d consumed" % (r.pos, len(data)))
    return obj


def dgcmatrix_to_dense(m):
    """dgCMatrix RObject -> (dense int64 array, row names, col names)"""
    assert m.kind == "S4" and list(m["class"].value) == ["dgCMatrix"]
    p, i, x = m["p"].value, m["i"].value, m["x"].value
    nrow, ncol = [int(v) for v in m["Dim"].value]
    out = np.zeros((nrow, ncol), np.int64)
    for c in range(ncol):
        out[i[p[c]:p[c + 1]], c] = x[p[c]:p[c + 1]]
    dn = m["Dimnames"].value
    return out, list(dn[0].value), list(dn[1].value)
