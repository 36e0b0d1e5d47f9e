"""Calls the C ABI with arguments it must refuse: every such call returns FALSE (bool_t style,
src/pdr.h:40-58), says why in mrpdr_last_error() and launches nothing."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mrnet_amd import capi  # noqa: E402
from mrnet_amd import wave as W  # noqa: E402


def refused(rc, *words):
    msg = capi.last_error()
    assert rc == 0, "the call was accepted"
    for w in words:
        assert w in msg, (w, msg)


def test_codec_calls_refuse_bad_arguments(codec):
    lib, h = capi.load(), codec.handle
    fmt, var = W.Format("%d"), W.Format("%d %ad")
    n = 8
    dev = codec.device
    vals = torch.zeros(n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int64, device=dev)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    arena = torch.zeros(1024, dtype=torch.uint8, device=dev)
    ok = torch.zeros(n, dtype=torch.int32, device=dev)
    col = (capi.Column * 2)()
    col[0] = capi.Column(vals.data_ptr(), None, None, n, None, None, 0)
    col[1] = capi.Column(vals.data_ptr(), cnt.data_ptr(), offs.data_ptr(), n, None, None, 0)
    packed = capi.Wire(arena.data_ptr(), None, None, 4, 1024)
    listed = capi.Wire(arena.data_ptr(), offs.data_ptr(), cnt.data_ptr(), 0, 1024)
    st = torch.cuda.current_stream().cuda_stream

    refused(lib.mrpdr_wave_decode(None, fmt.handle, C.byref(packed), n, 0, col, ok.data_ptr(), st), "NULL ctx")
    refused(lib.mrpdr_wave_decode(h, None, C.byref(packed), n, 0, col, ok.data_ptr(), st), "NULL format")
    refused(lib.mrpdr_wave_decode(h, fmt.handle, None, n, 0, col, ok.data_ptr(), st), "NULL wire")
    refused(lib.mrpdr_wave_decode(h, fmt.handle, C.byref(packed), n, 7, col, ok.data_ptr(), st), "byte order")
    refused(lib.mrpdr_wave_decode(h, fmt.handle, C.byref(packed), codec.max_packets + 1, 0, col, ok.data_ptr(), st),
            "exceeds")
    refused(lib.mrpdr_wave_decode(h, fmt.handle, C.byref(packed), n, 0, None, ok.data_ptr(), st), "NULL columns")
    short = capi.Wire(arena.data_ptr(), None, None, 3, 1024)
    refused(lib.mrpdr_wave_decode(h, fmt.handle, C.byref(short), n, 0, col, ok.data_ptr(), st), "stride")
    small = capi.Wire(arena.data_ptr(), None, None, 4, 16)
    refused(lib.mrpdr_wave_decode(h, fmt.handle, C.byref(small), n, 0, col, ok.data_ptr(), st), "arena too small")
    # variable-length formats need offsets and lens
    refused(lib.mrpdr_wave_decode(h, var.handle, C.byref(packed), n, 0, col, ok.data_ptr(), st), "lens")
    nolens = capi.Wire(arena.data_ptr(), offs.data_ptr(), None, 0, 1024)
    refused(lib.mrpdr_wave_encode(h, var.handle, col, n, C.byref(nolens), st), "lens")
    bad = (capi.Column * 2)()
    bad[0] = col[0]
    bad[1] = capi.Column(vals.data_ptr(), None, None, n, None, None, 0)
    refused(lib.mrpdr_wave_encode(h, var.handle, bad, n, C.byref(listed), st), "counts and offs")
    refused(lib.mrpdr_wave_decode_mixed(h, var.handle, C.byref(listed), n, None, col, ok.data_ptr(), st), "byteorders")
    total = C.c_uint64(0)
    refused(lib.mrpdr_wave_sizeof(h, var.handle, col, n, 3, 0, cnt.data_ptr(), offs.data_ptr(), C.byref(total), st),
            "align")
    assert capi.load().mrpdr_fmt_compile(b"%d %q") is None and "%q" in capi.last_error() or "q" in capi.last_error()
    torch.cuda.synchronize()
    assert codec.status() == 0


def test_filter_calls_refuse_bad_arguments(codec):
    lib, h = capi.load(), codec.handle
    dev = codec.device
    n = 16
    v = torch.zeros(n, dtype=torch.float64, device=dev)
    res = torch.zeros(48, dtype=torch.uint8, device=dev)
    ok = torch.zeros(3, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f64, i32 = int(W.DataType.DOUBLE_T), int(W.DataType.INT32_T)
    refused(lib.mrpdr_wave_filter(h, 99, f64, W.SHAPE_TREE, v.data_ptr(), 8, None, 0, n, 0, res.data_ptr(), ok.data_ptr(), st),
            "operator")
    refused(lib.mrpdr_wave_filter(h, W.TFILTER_SUM, int(W.DataType.STRING_T), W.SHAPE_TREE, v.data_ptr(), 8, None, 0, n, 0,
                                  res.data_ptr(), ok.data_ptr(), st), "scalar")
    refused(lib.mrpdr_wave_filter(h, W.TFILTER_SUM, f64, 5, v.data_ptr(), 8, None, 0, n, 0, res.data_ptr(), ok.data_ptr(), st),
            "shape")
    refused(lib.mrpdr_wave_filter(h, W.TFILTER_AVG, f64, W.SHAPE_TREE, v.data_ptr(), 8, None, 0, n, 0, res.data_ptr(),
                                  ok.data_ptr(), st), "counts")
    refused(lib.mrpdr_wave_filter(h, W.FILTER_OR, f64, W.SHAPE_TREE, v.data_ptr(), 8, None, 0, n, 0, res.data_ptr(),
                                  ok.data_ptr(), st), "integer")
    refused(lib.mrpdr_wave_filter(h, W.TFILTER_SUM, i32, W.SHAPE_TREE, v.data_ptr(), 8, None, 0, n, 0, None, ok.data_ptr(), st),
            "NULL result")
    refused(lib.mrpdr_wave_filter_stats(h, f64, v.data_ptr(), 8, None, 4, n, 0, res.data_ptr(), ok.data_ptr(), st), "NULL")
    # array filters
    cnt = torch.ones(n, dtype=torch.int64, device=dev)
    offs = torch.arange(n + 1, dtype=torch.int64, device=dev)
    ranks = capi.Column(torch.zeros(n, dtype=torch.int32, device=dev).data_ptr(), cnt.data_ptr(), offs.data_ptr(), n, None, None, 0)
    data = capi.Column(v.data_ptr(), cnt.data_ptr(), offs.data_ptr(), n, None, None, 0)
    out_len = torch.zeros(1, dtype=torch.int64, device=dev)
    out_rank = torch.zeros(1, dtype=torch.int32, device=dev)
    refused(lib.mrpdr_wave_perfdata(h, W.TFILTER_SUM, i32, C.byref(ranks), C.byref(data), n, 0, res.data_ptr(), 1,
                                    out_len.data_ptr(), out_rank.data_ptr(), None, 0, st), "type")
    refused(lib.mrpdr_wave_perfdata(h, W.TFILTER_SUM, f64, C.byref(ranks), C.byref(data), n, 4, res.data_ptr(), 1,
                                    out_len.data_ptr(), out_rank.data_ptr(), None, 0, st), "scratch")
    o = [capi.Column(res.data_ptr(), out_len.data_ptr(), offs.data_ptr(), 4, None, None, 0) for _ in range(3)]
    refused(lib.mrpdr_wave_int_eq_class(h, C.byref(data), C.byref(data), C.byref(data), n, C.byref(o[0]), C.byref(o[1]),
                                        C.byref(o[2]), None, 0, st), "scratch")
    ms = C.c_double(0)
    refused(lib.mrpdr_ctx_perf_plain_copy(h, res.data_ptr() + 1, v.data_ptr(), 16, 1, C.byref(ms), st), "multiples of 16")
    torch.cuda.synchronize()
    assert codec.status() == 0


def test_untrusted_sizes_cannot_reach_memory(codec):
    """Offsets and lengths that come off the wire (a Message's size vector) are data, not
    arguments: a header outside its buffer fails as a packet, a size vector whose entries would
    wrap the offsets is refused by the length check, a stride times n that wraps is refused."""
    from wavegen import O
    lib, h = capi.load(), codec.handle
    fmt = W.Format("%d")
    n = 6
    hdr = O.header_encode(9, 100, 1, "%d")
    L = hdr.size
    blob = np.concatenate([hdr] * n)
    offs = (np.arange(n, dtype=np.uint64) * np.uint64(L))
    lens = np.full(n, L, np.uint64)
    offs[1] = np.uint64(n * L)                 # starts at the end of the buffer
    offs[2] = np.uint64(2**64 - 8)             # offset + length wraps
    lens[3] = np.uint64(2**63)                 # longer than the buffer
    offs[4] = np.uint64(n * L - L + 1)         # last byte outside
    out = codec.decode_headers(fmt, n, codec.to_device(blob), offsets=codec.to_device(offs), lens=codec.to_device(lens))
    torch.cuda.synchronize()
    assert W.Codec.to_numpy(out["ok"]).tolist() == [1, 0, 0, 0, 0, 1]
    # a message whose size vector announces absurd packets
    vals = np.arange(n, dtype=np.int32)
    msg = codec.message_build(fmt, codec.columns_from_numpy(fmt, [vals]), n)
    good = W.Codec.to_numpy(msg).copy()
    for at, val in ((5 + 16 * 2, 2**64 - 1), (5 + 16 * 3 + 8, 2**63 + 5), (5, 2**62)):
        bad = good.copy()
        bad[at:at + 8] = np.frombuffer(int(val).to_bytes(8, "little"), np.uint8)
        with pytest.raises(W.PdrError):
            codec.message_parse(fmt, codec.to_device(bad))
    bad = good.copy()                          # every size huge: the sum must not wrap into something small
    for i in range(2 * n):
        bad[5 + 8 * i:13 + 8 * i] = 0xFF
    with pytest.raises(W.PdrError):
        codec.message_parse(fmt, codec.to_device(bad))
    nn, hd, wire = codec.message_parse(fmt, msg)
    assert nn == n and bool(hd["ok"].all())
    # the same without the host-side length check (the receiver knows the count from the socket):
    # nothing is refused up front, the packets that do not lie inside the buffer fail one by one
    nn, hd, wire = codec.message_parse(fmt, msg, n=n, check_length=False)
    dcols, dok = codec.decode(fmt, wire)
    assert bool(hd["ok"].all()) and bool(dok.all())
    assert np.array_equal(W.Codec.to_numpy(dcols[0].values).view(np.int32)[:n], vals)
    for at, val in ((5 + 16 * 2, 2**64 - 1), (5 + 16 * 3 + 8, 2**63 + 5), (5, 2**62)):
        bad = good.copy()
        bad[at:at + 8] = np.frombuffer(int(val).to_bytes(8, "little"), np.uint8)
        nn, hd, wire = codec.message_parse(fmt, codec.to_device(bad), n=n, check_length=False)
        dcols, dok = codec.decode(fmt, wire)
        torch.cuda.synchronize()
        first_bad = (at - 5) // 16
        hok, dk = W.Codec.to_numpy(hd["ok"]), W.Codec.to_numpy(dok)
        assert hok[:first_bad].all() and dk[:first_bad].all()          # the packets in front are intact
        assert not (hok[first_bad:].all() and dk[first_bad:].all())
    # n * stride wraps around 2^64
    arena = torch.zeros(64, dtype=torch.uint8, device=codec.device)
    col = (capi.Column * 1)()
    v = torch.zeros(n, dtype=torch.int32, device=codec.device)
    col[0] = capi.Column(v.data_ptr(), None, None, n, None, None, 0)
    wrap = capi.Wire(arena.data_ptr(), None, None, (1 << 63) + 8, 64)
    ok = torch.zeros(n, dtype=torch.int32, device=codec.device)
    refused(lib.mrpdr_wave_decode(h, fmt.handle, C.byref(wrap), 2, 0, col, ok.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream), "arena too small")
    torch.cuda.synchronize()
    assert codec.status() == 0


def test_column_without_room_is_reported(codec):
    """A variable field decoded into a column of capacity 0 while the wave holds elements: nothing
    can be written, and status bit 0 says so (the long-payload path, which plans chunks from the
    capacity, and the short-packet tiles alike)."""
    from wavegen import O, make_wave
    for fmt_s, count in (("%d %ad", 3), ("%ad", 5000)):
        n = 40
        fmt = W.Format(fmt_s)
        ocols = make_wave(fmt_s, n, 3, max_count=count, counts={len(fmt) - 1: np.full(n, count, np.uint64)})
        arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
        wire = W.WireWave(codec.to_device(arena), n, codec.to_device(offs), codec.to_device(lens), 0)
        cols = codec.alloc_columns(fmt, n)            # capacities 0: no values buffers
        ok = codec.empty(n, torch.int32)
        codec.decode_into(fmt, wire, cols, ok)
        torch.cuda.synchronize()
        assert int(W.Codec.to_numpy(cols[-1].offs).view(np.uint64)[n]) == n * count
        assert codec.status() & 1, fmt_s
        assert codec.status() == 0                    # reading cleared it
