"""The device BAM decoder through its C-ABI alone (include/dropest_bgzf.h: dropest_bam_decoder_*), without the facade: one window over a written
BAM -- the reader's status of every record (BamController.cpp:87-107, FilledBamParamsParser.cpp:12-40), the dense columns of the accepted ones,
the `need` list (everything while the dictionaries are empty; only strings with N once they are given), the bytes fetch_records returns, patch."""
import ctypes as C
import gzip
import struct

import numpy as np
import pytest

from dropest_amd import capi

import bam_writer as bw

pytestmark = pytest.mark.gpu
P = C.POINTER


class Cfg(C.Structure):
    _fields_ = [("tag", C.c_uint16 * 6), ("filled_bam", C.c_int32), ("min_phred", C.c_int32), ("has_read_type", C.c_int32), ("n_refs", C.c_int32),
                ("intronic_len", C.c_uint32), ("intergenic_len", C.c_uint32), ("intronic", C.c_uint8 * 24), ("intergenic", C.c_uint8 * 24)]


class Window(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("counts", C.c_uint64 * 5), ("n_accepted", C.c_uint64), ("d_cb", C.c_void_p), ("d_umi", C.c_void_p), ("d_gene", C.c_void_p),
                ("d_aux", C.c_void_p), ("n_need", C.c_uint32), ("need_rec", P(C.c_uint32)), ("need_pos", P(C.c_uint32)), ("need_size", P(C.c_uint32)),
                ("quality_seen", C.c_uint32), ("any_gene", C.c_uint32), ("window_bytes", C.c_uint64), ("tail_bytes", C.c_uint64), ("n_blocks", C.c_uint32),
                ("refused_blocks", C.c_uint32), ("guesses_repaired", C.c_uint32), ("pad", C.c_uint32), ("ms", C.c_double * 4), ("quality_len_min", C.c_uint32), ("quality_len_max", C.c_uint32)]


def fnv1a(s):
    h = 1469598103934665603
    for c in s.encode():
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def tag16(t):
    return ord(t[0]) | (ord(t[1]) << 8)


def columns(L, dec, n):
    cb, umi = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    gene, aux = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    L.dropest_bam_decoder_columns_to_host.argtypes = [C.c_void_p] * 5
    assert L.dropest_bam_decoder_columns_to_host(dec, cb.ctypes.data, umi.ctypes.data, gene.ctypes.data, aux.ctypes.data) == 0, L.dropest_bgzf_last_error()
    return cb, umi, gene, aux


def test_one_window_through_the_c_abi(tmp_path):
    L = capi.lib()
    L.dropest_bgzf_last_error.restype = C.c_char_p
    rng = np.random.default_rng(12)
    refs = [("chr%d" % i, 1 << 20) for i in range(4)]
    genes = ["GENE%d" % i for i in range(40)]
    recs, want = [], []          # want: (status, cb, umi, gene or None, mark, ref) per record, by the rules of the host reader
    for i in range(5000):
        cb = "".join(rng.choice(list("ACGT"), 12)); umi = "".join(rng.choice(list("ACGT"), 8))
        if i % 211 == 0:
            umi = umi[:2] + "N" + umi[3:]
        g = None if i % 9 == 0 else genes[int(rng.integers(0, len(genes)))]
        rt = [None, "N", "I", "E"][i % 4] if g else None
        uq = "".join(chr(33 + int(x)) for x in rng.integers(2, 40, 8))
        tags = [("CB", "Z", cb), ("UB", "Z", umi), ("UQ", "Z", uq)] + ([("GX", "Z", g)] if g else []) + ([("RE", "A", rt)] if rt else []) + [("NH", "i", 1)]
        flag, ref, status = 0, int(rng.integers(0, 4)), 0
        if i % 97 == 1:
            flag, status = 4, 1
        elif i % 97 == 2:
            flag, status = 0x100, 1
        elif i % 97 == 3:
            ref, status = -1, 2
        elif i % 97 == 4:
            tags, status = [t for t in tags if t[0] != "UB"], 3
        mark = 1 if g is None else (4 if rt == "N" else 1 if rt == "I" else 2)
        recs.append(bw.record(ref, i, "r%d" % i, flag=flag, tags=tags))
        want.append((status, cb, umi, g, mark, ref, uq))
    path = str(tmp_path / "c.bam")
    bw.write_bam(path, refs, recs, block=12_345)
    blob = open(path, "rb").read()
    raw = gzip.decompress(blob)
    l_text = struct.unpack_from("<I", raw, 4)[0]
    h = 12 + l_text
    for _ in refs:
        h += 8 + struct.unpack_from("<I", raw, h)[0]
    assert raw[h:h + len(recs[0])] == recs[0]
    at, cum, c0, u0 = 0, 0, None, None        # the block that holds byte h of the inflated stream
    while at < len(blob):
        bsize = struct.unpack_from("<H", blob, at + 16)[0] + 1
        isize = struct.unpack_from("<I", blob, at + bsize - 4)[0]
        if cum <= h < cum + isize:
            c0, u0 = at, h - cum
            break
        cum += isize; at += bsize
    cfg = Cfg()
    for k, t in enumerate(("CB", "UB", "CQ", "UQ", "GX", "RE")):
        cfg.tag[k] = tag16(t)
    cfg.filled_bam, cfg.min_phred, cfg.has_read_type, cfg.n_refs = 1, 0, 1, len(refs)
    cfg.intronic_len, cfg.intergenic_len = 1, 1
    cfg.intronic[0], cfg.intergenic[0] = ord("N"), ord("I")
    dec = C.c_void_p()
    assert L.dropest_bam_decoder_create(0, C.byref(cfg), C.byref(dec)) == 0, L.dropest_bgzf_last_error()
    comp = np.frombuffer(blob[c0:], np.uint8)
    INFLATE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p)
    L.dropest_bam_decoder_window.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, P(Window)]
    ok_idx = [i for i, w in enumerate(want) if w[0] == 0]

    def window():
        w = Window()
        assert L.dropest_bam_decoder_window(dec, comp.ctypes.data, len(comp), u0, 1, None, None, C.byref(w)) == 0, L.dropest_bgzf_last_error()
        return w
    # 1. empty dictionaries: every accepted record brings something new
    w = window()
    assert w.n_records == len(recs) and w.tail_bytes == 0 and w.refused_blocks == 0
    assert list(w.counts) == [sum(1 for x in want if x[0] == s) for s in range(5)]
    assert w.n_accepted == len(ok_idx) == w.n_need and w.any_gene == 1 and w.quality_seen == 1 and w.quality_len_min == w.quality_len_max == 8
    need_rec = np.ctypeslib.as_array(w.need_rec, (w.n_need,)).copy(); need_pos = np.ctypeslib.as_array(w.need_pos, (w.n_need,)).copy()
    need_size = np.ctypeslib.as_array(w.need_size, (w.n_need,)).copy()
    assert need_rec.tolist() == ok_idx and need_pos.tolist() == list(range(len(ok_idx))) and need_size.tolist() == [len(recs[i]) for i in ok_idx]
    buf = np.zeros(int(need_size.sum()) + 16, np.uint8); off = np.zeros(len(ok_idx), np.uint64)
    L.dropest_bam_decoder_fetch_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    assert L.dropest_bam_decoder_fetch_records(dec, need_rec.ctypes.data, len(need_rec), buf.ctypes.data, len(buf), off.ctypes.data) == 0
    assert buf[: int(need_size.sum())].tobytes() == b"".join(recs[i] for i in ok_idx)
    cbc = columns(L, dec, int(w.n_accepted))[0]
    assert cbc.tolist() == [capi.pack_seq(want[i][1]) for i in ok_idx]
    # 2. the dictionaries given: only the UMIs with N are left to the caller; gene and chromosome indices come from the tables
    gene_id = {g: 100 + k for k, g in enumerate(genes)}
    hashes = np.array([fnv1a(g) for g in genes], np.uint64); ids = np.array([gene_id[g] for g in genes], np.uint32)
    chr_of_ref = np.array([7, 5, 3, 1], np.int32)
    L.dropest_bam_decoder_set_dictionaries.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    assert L.dropest_bam_decoder_set_dictionaries(dec, hashes.ctypes.data, ids.ctypes.data, len(genes), chr_of_ref.ctypes.data, len(refs)) == 0
    L.dropest_bam_decoder_reset.argtypes = [C.c_void_p, C.c_void_p]
    w = window()
    with_n = [k for k, i in enumerate(ok_idx) if "N" in want[i][2] and want[i][3]]
    assert np.ctypeslib.as_array(w.need_pos, (w.n_need,)).tolist() == with_n and len(with_n) > 5
    _, umi, gene, aux = columns(L, dec, int(w.n_accepted))
    for k, i in enumerate(ok_idx):
        st, cb, u, g, mark, ref, _ = want[i]
        assert gene[k] == (gene_id[g] if g else 0xFFFFFFFF)
        assert umi[k] == ((capi.pack_seq(u) or 0) if g else 1)
        touches = g is None or (mark & 6)
        assert aux[k] == (mark << 16) | (int(chr_of_ref[ref]) if touches else 0), (k, i, aux[k], mark, ref)
    # the UMI quality strings of the accepted reads, one row each (zeros where the read has no gene)
    rows_p = C.c_void_p()
    L.dropest_bam_decoder_quality_rows.argtypes = [C.c_void_p, C.c_uint32, P(C.c_void_p)]
    assert L.dropest_bam_decoder_quality_rows(dec, 8, C.byref(rows_p)) == 0
    rows = np.ctypeslib.as_array(C.cast(rows_p, P(C.c_uint8)), (int(w.n_accepted), 8))
    for k, i in enumerate(ok_idx):
        assert rows[k].tobytes() == (want[i][6].encode() if want[i][3] else b"\0" * 8), (k, i)
    # 3. patch: the caller's values for the rows it resolved
    pos = np.array(with_n, np.uint32)
    pc, pu = np.full(len(pos), 11, np.uint64), np.full(len(pos), 22, np.uint64); pg, pa = np.full(len(pos), 33, np.uint32), np.full(len(pos), 44, np.uint32)
    L.dropest_bam_decoder_patch.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint32]
    assert L.dropest_bam_decoder_patch(dec, pos.ctypes.data, pc.ctypes.data, pu.ctypes.data, pg.ctypes.data, pa.ctypes.data, len(pos)) == 0
    _, umi2, _, aux2 = columns(L, dec, int(w.n_accepted))
    assert (umi2[pos] == 22).all() and (aux2[pos] == 44).all() and (np.delete(umi2, pos) == np.delete(umi, pos)).all()
    # a reset decoder starts over: empty dictionaries again
    assert L.dropest_bam_decoder_reset(dec, C.byref(cfg)) == 0
    w = window()
    assert w.n_need == len(ok_idx)
    # the same bytes from the decoder's pinned staging buffer, sent ahead of the window call (dropest_bam_decoder_upload: what a reader thread does
    # while the window before is in the kernels): the same window; a second call with that buffer (nothing sent ahead) copies by itself
    L.dropest_bam_decoder_staging.argtypes = [C.c_void_p, C.c_int, C.c_uint64, P(C.c_void_p)]
    L.dropest_bam_decoder_upload.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    stage = C.c_void_p()
    assert L.dropest_bam_decoder_staging(dec, 1, len(comp) + 4096, C.byref(stage)) == 0, L.dropest_bgzf_last_error()
    C.memmove(stage, comp.ctypes.data, len(comp))
    for ahead in (True, False, True):
        assert L.dropest_bam_decoder_reset(dec, C.byref(cfg)) == 0
        if ahead:
            assert L.dropest_bam_decoder_upload(dec, 1, len(comp)) == 0
        w2 = Window()
        assert L.dropest_bam_decoder_window(dec, stage, len(comp), u0, 1, None, None, C.byref(w2)) == 0, L.dropest_bgzf_last_error()
        assert (w2.n_records, w2.n_accepted, w2.n_need, list(w2.counts)) == (w.n_records, w.n_accepted, w.n_need, list(w.counts))
        assert columns(L, dec, int(w2.n_accepted))[0].tolist() == cbc.tolist()
    # bytes sent ahead that are not whole blocks: the window call says so (not the upload)
    assert L.dropest_bam_decoder_reset(dec, C.byref(cfg)) == 0
    assert L.dropest_bam_decoder_upload(dec, 1, len(comp) - 5) == 0
    w2 = Window()
    assert L.dropest_bam_decoder_window(dec, stage, len(comp) - 5, u0, 1, None, None, C.byref(w2)) != 0
    assert L.dropest_bam_decoder_upload(dec, 1, 1 << 40) == 0 and L.dropest_bam_decoder_upload(dec, 2, 10) != 0      # too long: left to the window call; no such buffer
    # the window call in its two halves (first: copy, inflate, chain, on a stream of its own; second: fields and dense columns)
    L.dropest_bam_decoder_window_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, P(C.c_int)]
    L.dropest_bam_decoder_window_finish.argtypes = [C.c_void_p, C.c_int, P(Window)]
    for rep in range(2):
        assert L.dropest_bam_decoder_reset(dec, C.byref(cfg)) == 0
        slot, w3 = C.c_int(-1), Window()
        assert L.dropest_bam_decoder_window_begin(dec, comp.ctypes.data, len(comp), u0, 1, None, None, C.byref(slot)) == 0, L.dropest_bgzf_last_error()
        assert slot.value in (0, 1)
        assert L.dropest_bam_decoder_window_finish(dec, slot.value, C.byref(w3)) == 0, L.dropest_bgzf_last_error()
        assert (w3.n_records, w3.n_accepted, w3.n_need, list(w3.counts)) == (w.n_records, w.n_accepted, w.n_need, list(w.counts))
        assert columns(L, dec, int(w3.n_accepted))[0].tolist() == cbc.tolist()
        assert L.dropest_bam_decoder_window_finish(dec, slot.value, C.byref(w3)) != 0 and b"not begun" in L.dropest_bgzf_last_error()
    # the compressed bytes in pinned PIECES (what BamController's readers do: the pinned memory does not grow with the window): pieces of 4 KB so that
    # this small file takes several, sent in a scrambled order from two slots; the block table the caller's, the decoder's own, and a wrong one
    # (refused: the decoder makes its own from the host bytes); a window whose pieces were never declared is copied by the window call itself
    class Blocks(C.Structure):
        _fields_ = [("n", C.c_uint64), ("in_off", C.c_void_p), ("in_len", C.c_void_p), ("out_len", C.c_void_p), ("crc32", C.c_void_p)]
    L.dropest_bam_decoder_reserve.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]
    L.dropest_bam_decoder_pieces.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, P(C.c_void_p)]
    L.dropest_bam_decoder_piece_wait.argtypes = [C.c_void_p, C.c_uint32]
    L.dropest_bam_decoder_upload_begin.argtypes = [C.c_void_p, C.c_int]
    L.dropest_bam_decoder_upload_piece.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_uint64]
    L.dropest_bam_decoder_upload_done.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, P(Blocks)]
    PIECE = 4096
    ptrs = (C.c_void_p * 3)()
    assert L.dropest_bam_decoder_reserve(dec, 0, len(comp) + 4096, 0) == 0 and L.dropest_bam_decoder_reserve(dec, 0, len(comp) + 4096, 40 * len(comp)) == 0
    assert L.dropest_bam_decoder_pieces(dec, 3, PIECE, ptrs) == 0, L.dropest_bgzf_last_error()
    assert L.dropest_bam_decoder_upload_piece(dec, 0, 0, 0, 16) != 0                       # no .._upload_begin yet
    # the table: header (12 + XLEN bytes), payload, CRC-32, ISIZE per block
    in_off, in_len, out_len, crcs, at = [], [], [], [], 0
    raw = comp.tobytes()
    while at < len(raw):
        xlen = int.from_bytes(raw[at + 10:at + 12], "little"); bsize = int.from_bytes(raw[at + 16:at + 18], "little") + 1
        in_off.append(at + 12 + xlen); in_len.append(bsize - 12 - xlen - 8)
        crcs.append(int.from_bytes(raw[at + bsize - 8:at + bsize - 4], "little")); out_len.append(int.from_bytes(raw[at + bsize - 4:at + bsize], "little"))
        at += bsize
    a_off, a_len, a_out, a_crc = np.array(in_off, np.uint64), np.array(in_len, np.uint32), np.array(out_len, np.uint32), np.array(crcs, np.uint32)
    for table in ("callers", "none", "wrong", "undeclared"):
        assert L.dropest_bam_decoder_reset(dec, C.byref(cfg)) == 0
        assert L.dropest_bam_decoder_upload_begin(dec, 0) == 0
        order = list(range(0, len(comp), PIECE))
        order = order[1::2] + order[0::2]
        for k, from_ in enumerate(order):
            slot = k % 3
            assert L.dropest_bam_decoder_piece_wait(dec, slot) == 0
            n = min(PIECE, len(comp) - from_)
            C.memmove(ptrs[slot], comp.ctypes.data + from_, n)
            assert L.dropest_bam_decoder_upload_piece(dec, 0, slot, from_, n) == 0
        assert L.dropest_bam_decoder_upload_piece(dec, 0, 0, 1 << 40, PIECE) != 0 and L.dropest_bam_decoder_upload_piece(dec, 0, 7, 0, 16) != 0      # beyond what .._reserve made room for; no such piece
        if table == "callers":
            b = Blocks(len(a_off), a_off.ctypes.data, a_len.ctypes.data, a_out.ctypes.data, a_crc.ctypes.data)
            assert L.dropest_bam_decoder_upload_done(dec, 0, comp.ctypes.data, len(comp), C.byref(b)) == 0
        elif table == "wrong":
            bad = a_len.copy(); bad[0] += 3
            b = Blocks(len(a_off), a_off.ctypes.data, bad.ctypes.data, a_out.ctypes.data, a_crc.ctypes.data)
            assert L.dropest_bam_decoder_upload_done(dec, 0, comp.ctypes.data, len(comp), C.byref(b)) == 0
        elif table == "none":
            assert L.dropest_bam_decoder_upload_done(dec, 0, comp.ctypes.data, len(comp), None) == 0
        w4 = Window()
        assert L.dropest_bam_decoder_window(dec, comp.ctypes.data, len(comp), u0, 1, None, None, C.byref(w4)) == 0, L.dropest_bgzf_last_error()
        assert (w4.n_records, w4.n_accepted, w4.n_need, list(w4.counts), w4.refused_blocks) == (w.n_records, w.n_accepted, w.n_need, list(w.counts), 0), table
        assert columns(L, dec, int(w4.n_accepted))[0].tolist() == cbc.tolist(), table
    # the kernels on a stream the caller lends, and back on the decoder's own
    L.dropest_bam_decoder_use_stream.argtypes = [C.c_void_p, C.c_void_p]
    ctx = capi.Context(device=0)                                        # (what BamController lends: the container's context's stream)
    lent = C.c_void_p(capi.lib().dropest_stream(ctx.h))
    assert lent.value
    for stream in (lent, None, lent):
        assert L.dropest_bam_decoder_use_stream(dec, stream) == 0, L.dropest_bgzf_last_error()
        assert L.dropest_bam_decoder_reset(dec, C.byref(cfg)) == 0
        w5 = window()
        assert (w5.n_records, w5.n_accepted, w5.n_need, list(w5.counts)) == (w.n_records, w.n_accepted, w.n_need, list(w.counts))
        assert columns(L, dec, int(w5.n_accepted))[0].tolist() == cbc.tolist()
    assert L.dropest_bam_decoder_use_stream(dec, None) == 0
    ctx.close()
    L.dropest_bam_decoder_destroy.argtypes = [C.c_void_p]
    L.dropest_bam_decoder_destroy(dec)
