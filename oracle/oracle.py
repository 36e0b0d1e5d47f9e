"""ctypes front end of the CPU oracle (oracle/libpdr_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports it.

A wave is described by numpy columns, one dict per format token:
    scalar        {"type": t, "values": T[n]}
    array/string  {"type": t, "counts": u64[n], "offs": u64[n+1], "values": packed}
                  (a string counts strlen+1 bytes, NUL included)
    string array  adds "sub_counts": u64[S], "sub_offs": u64[S+1]; "values" = bytes
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpdr_oracle.so")

# include/mrnet/DataElement.h:27-45
(UNKNOWN_T, CHAR_T, UCHAR_T, CHAR_ARRAY_T, UCHAR_ARRAY_T, STRING_T, STRING_ARRAY_T,
 INT16_T, UINT16_T, INT16_ARRAY_T, UINT16_ARRAY_T, INT32_T, UINT32_T, INT32_ARRAY_T,
 UINT32_ARRAY_T, INT64_T, UINT64_T, INT64_ARRAY_T, UINT64_ARRAY_T, FLOAT_T, DOUBLE_T,
 FLOAT_ARRAY_T, DOUBLE_ARRAY_T, CHAR_LRG_ARRAY_T, UCHAR_LRG_ARRAY_T, STRING_LRG_ARRAY_T,
 INT16_LRG_ARRAY_T, UINT16_LRG_ARRAY_T, INT32_LRG_ARRAY_T, UINT32_LRG_ARRAY_T,
 INT64_LRG_ARRAY_T, UINT64_LRG_ARRAY_T, FLOAT_LRG_ARRAY_T, DOUBLE_LRG_ARRAY_T) = range(34)

SUM, MIN, MAX, AVG, OR, AND = 0, 1, 2, 3, 4, 5
LITTLE, BIG = 0, 1

_ELEM_DTYPE = {
    CHAR_T: np.int8, UCHAR_T: np.uint8, INT16_T: np.int16, UINT16_T: np.uint16,
    INT32_T: np.int32, UINT32_T: np.uint32, INT64_T: np.int64, UINT64_T: np.uint64,
    FLOAT_T: np.float32, DOUBLE_T: np.float64,
    CHAR_ARRAY_T: np.int8, UCHAR_ARRAY_T: np.uint8, INT16_ARRAY_T: np.int16,
    UINT16_ARRAY_T: np.uint16, INT32_ARRAY_T: np.int32, UINT32_ARRAY_T: np.uint32,
    INT64_ARRAY_T: np.int64, UINT64_ARRAY_T: np.uint64, FLOAT_ARRAY_T: np.float32,
    DOUBLE_ARRAY_T: np.float64,
    CHAR_LRG_ARRAY_T: np.int8, UCHAR_LRG_ARRAY_T: np.uint8, INT16_LRG_ARRAY_T: np.int16,
    UINT16_LRG_ARRAY_T: np.uint16, INT32_LRG_ARRAY_T: np.int32, UINT32_LRG_ARRAY_T: np.uint32,
    INT64_LRG_ARRAY_T: np.int64, UINT64_LRG_ARRAY_T: np.uint64, FLOAT_LRG_ARRAY_T: np.float32,
    DOUBLE_LRG_ARRAY_T: np.float64,
    STRING_T: np.uint8, STRING_ARRAY_T: np.uint8, STRING_LRG_ARRAY_T: np.uint8,
}


class _Column(C.Structure):
    _fields_ = [("type", C.c_int), ("values", C.c_void_p), ("counts", C.c_void_p),
                ("offs", C.c_void_p), ("sub_counts", C.c_void_p), ("sub_offs", C.c_void_p)]


_lib = None


def build(force=False):
    """Compile the oracle with its Makefile (gcc); building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pdr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u64p, i32p, vp = C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_void_p
        L.orc_fmt_types.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int]
        L.orc_fmt_types.restype = C.c_int
        L.orc_type_width.argtypes = [C.c_int]; L.orc_type_width.restype = C.c_int
        L.orc_type_kind.argtypes = [C.c_int]; L.orc_type_kind.restype = C.c_int
        L.orc_local_bo.restype = C.c_int
        L.orc_wave_sizeof.argtypes = [C.c_char_p, C.POINTER(_Column), C.c_uint64, vp]
        L.orc_wave_sizeof.restype = C.c_int32
        L.orc_wave_encode.argtypes = [C.c_char_p, C.POINTER(_Column), C.c_uint64, vp, vp, vp]
        L.orc_wave_encode.restype = C.c_int32
        L.orc_wave_decode.argtypes = [C.c_char_p, vp, vp, vp, C.c_uint64, C.c_int,
                                      C.POINTER(_Column), vp]
        L.orc_wave_decode.restype = C.c_uint64
        L.orc_wave_decode_orders.argtypes = [C.c_char_p, vp, vp, vp, C.c_uint64, vp, C.POINTER(_Column), vp]
        L.orc_wave_decode_orders.restype = C.c_uint64
        L.orc_header_encode.argtypes = [C.c_uint32, C.c_int32, C.c_uint32, C.c_char_p, vp,
                                        C.c_uint64, vp, C.c_uint64]
        L.orc_header_encode.restype = C.c_uint64
        L.orc_header_decode.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32), i32p,
                                        C.POINTER(C.c_uint32), C.c_char_p, C.c_uint64, vp,
                                        C.c_uint64, u64p, C.POINTER(C.c_int)]
        L.orc_header_decode.restype = C.c_int32
        L.orc_filter_flat.argtypes = [C.c_int, C.c_int, vp, vp, C.c_uint64, vp, i32p]
        L.orc_filter_flat.restype = C.c_int32
        L.orc_filter_tree.argtypes = [C.c_int, C.c_int, vp, vp, C.c_uint64, vp, C.c_int, vp, i32p]
        L.orc_filter_tree.restype = C.c_int32
        L.orc_message_build.argtypes = [C.c_uint32, vp, vp, vp, vp, vp, vp, vp, C.c_uint64]
        L.orc_message_build.restype = C.c_uint64
        L.orc_message_parse.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32), vp, vp, vp, vp, C.c_uint32]
        L.orc_message_parse.restype = C.c_int32
        L.orc_filter_perfdata.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_uint64,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
        L.orc_filter_perfdata.restype = C.c_int32
        L.orc_filter_int_eq_class.argtypes = [vp] * 6 + [C.c_uint64, vp, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_filter_int_eq_class.restype = C.c_int32
        L.orc_byte_swap.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
        L.orc_byte_swap.restype = None
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data


def fmt_types(fmt):
    buf = (C.c_int * 64)()
    n = lib().orc_fmt_types(fmt.encode(), buf, 64)
    if n < 0:
        raise ValueError("unknown token in format %r" % fmt)
    return [buf[i] for i in range(n)]


def type_kind(t):
    return lib().orc_type_kind(t)


def type_width(t):
    return lib().orc_type_width(t)


def elem_dtype(t):
    return np.dtype(_ELEM_DTYPE[t])


def local_byteorder():
    return lib().orc_local_bo()


def _c_columns(cols, keep):
    arr = (_Column * max(1, len(cols)))()
    for i, c in enumerate(cols):
        arr[i].type = c["type"]
        for key in ("values", "counts", "offs", "sub_counts", "sub_offs"):
            a = c.get(key)
            if a is not None:
                a = np.ascontiguousarray(a)
                if key != "values":
                    assert a.dtype == np.uint64, key
                c[key] = a
                keep.append(a)
                setattr(arr[i], key, a.ctypes.data)
    return arr


def wave_sizeof(fmt, cols, n):
    keep = []
    lens = np.zeros(n, dtype=np.uint64)
    ok = lib().orc_wave_sizeof(fmt.encode(), _c_columns(cols, keep), n, _ptr(lens))
    if not ok:
        raise RuntimeError("orc_wave_sizeof failed")
    return lens


def scan_offsets(lens, align=1, phase=0):
    """The arena layout rule: packet i owns a slot of round_up(phase + len, align) bytes and
    starts `phase` bytes into it; offsets[n] is the arena size."""
    a = np.uint64(align)
    slots = (lens.astype(np.uint64) + np.uint64(phase) + (a - np.uint64(1))) // a * a
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(slots, out=offs[1:])
    offs[:-1] += np.uint64(phase)
    return offs


def wave_encode(fmt, cols, n, align=1, phase=0):
    """Returns (arena u8[], offsets u64[n+1], lens u64[n]); padding bytes are zero."""
    lens = wave_sizeof(fmt, cols, n)
    offs = scan_offsets(lens, align, phase)
    arena = np.zeros(int(offs[-1]), dtype=np.uint8)
    keep = []
    lens2 = np.zeros(n, dtype=np.uint64)
    ok = lib().orc_wave_encode(fmt.encode(), _c_columns(cols, keep), n, _ptr(offs),
                               _ptr(arena), _ptr(lens2))
    if not ok:
        raise RuntimeError("orc_wave_encode failed")
    assert np.array_equal(lens, lens2)
    return arena, offs, lens


def wave_decode(fmt, arena, offsets, lens, sender_bo=LITTLE):
    """Returns (cols, ok i32[n]).  Columns are trimmed to the decoded totals.
    sender_bo: LITTLE / BIG for the wave, or a uint8 array with every packet's own order."""
    types = fmt_types(fmt)
    n = len(lens)
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    cap = int(arena.size)
    cols = []
    for t in types:
        k = type_kind(t)
        c = {"type": t}
        dt = elem_dtype(t)
        if k == 0:
            c["values"] = np.zeros(n, dtype=dt)
        else:
            c["counts"] = np.zeros(n, dtype=np.uint64)
            c["offs"] = np.zeros(n + 1, dtype=np.uint64)
            c["values"] = np.zeros(cap // dt.itemsize + 1, dtype=dt)
            if k == 3:
                c["sub_counts"] = np.zeros(cap + 1, dtype=np.uint64)
                c["sub_offs"] = np.zeros(cap + 2, dtype=np.uint64)
        cols.append(c)
    ok = np.zeros(n, dtype=np.int32)
    keep = []
    if isinstance(sender_bo, np.ndarray):           # one byte order per packet
        orders = np.ascontiguousarray(sender_bo, dtype=np.uint8)
        assert orders.size == n
        lib().orc_wave_decode_orders(fmt.encode(), _ptr(arena), _ptr(offsets), _ptr(lens), n, _ptr(orders),
                                     _c_columns(cols, keep), _ptr(ok))
    else:
        lib().orc_wave_decode(fmt.encode(), _ptr(arena), _ptr(offsets), _ptr(lens), n, sender_bo,
                              _c_columns(cols, keep), _ptr(ok))
    for c in cols:
        k = type_kind(c["type"])
        if k in (1, 2):
            c["values"] = c["values"][: int(c["offs"][n])].copy()
        elif k == 3:
            s = int(c["offs"][n])
            c["sub_counts"] = c["sub_counts"][:s].copy()
            c["sub_offs"] = c["sub_offs"][: s + 1].copy()
            c["values"] = c["values"][: int(c["sub_offs"][s])].copy()
    return cols, ok


def header_encode(stream_id, tag, src_rank, fmt, dest=()):
    dest = np.asarray(dest, dtype=np.uint32)
    out = np.zeros(64 + len(fmt) + 4 * len(dest), dtype=np.uint8)
    n = lib().orc_header_encode(stream_id, tag, src_rank, fmt.encode(), _ptr(dest), len(dest),
                                _ptr(out), out.size)
    if n == 0:
        raise RuntimeError("orc_header_encode failed")
    return out[:n].copy()


def header_decode(hdr):
    hdr = np.ascontiguousarray(hdr, dtype=np.uint8)
    sid, tag, src = C.c_uint32(), C.c_int32(), C.c_uint32()
    fmt = C.create_string_buffer(4096)
    dest = np.zeros(max(1, hdr.size // 4), dtype=np.uint32)
    nd, bo = C.c_uint64(), C.c_int()
    ok = lib().orc_header_decode(_ptr(hdr), hdr.size, C.byref(sid), C.byref(tag), C.byref(src),
                                 fmt, 4096, _ptr(dest), dest.size, C.byref(nd), C.byref(bo))
    if not ok:
        return None
    return {"stream_id": sid.value, "tag": tag.value, "src_rank": src.value,
            "fmt": fmt.value.decode("latin-1"), "dest": dest[: nd.value].copy(), "byteorder": bo.value}


def byte_swap(a, elsize):
    """src/byte_order.c:8-17 over a whole buffer."""
    a = np.ascontiguousarray(a).view(np.uint8)
    out = np.empty_like(a)
    assert a.size % elsize == 0
    lib().orc_byte_swap(_ptr(out), _ptr(a), a.size // elsize, elsize)
    return out


def filter_flat(op, t, vals, cnt=None):
    vals = np.ascontiguousarray(vals, dtype=elem_dtype(t))
    if cnt is not None:
        cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    out = np.zeros(1, dtype=elem_dtype(t))
    oc = C.c_int32(0)
    ok = lib().orc_filter_flat(op, t, _ptr(vals), _ptr(cnt), vals.size, _ptr(out), C.byref(oc))
    if not ok:
        return None
    return out[0], oc.value


def bench_small_wave(reps, n=4, value=1):
    """Seconds per wave of the n-packet '%d' decode + Sum + encode loop, timed inside the C library."""
    fn = lib().orc_bench_small_wave
    fn.restype = C.c_double
    fn.argtypes = [C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_int32)]
    s = C.c_int32(0)
    t = fn(reps, n, value, C.byref(s))
    if t < 0 or s.value != (n * value + 2**31) % 2**32 - 2**31:      # the int32 sum wraps
        raise RuntimeError("orc_bench_small_wave failed")
    return t


def bench_chain(vals, reps=50):
    """Seconds per tight left fold (the chain of additions alone) of a float32 / float64 column on one
    host core; checked against filter_flat's bits."""
    vals = np.ascontiguousarray(vals)
    assert vals.dtype in (np.float32, np.float64)
    fn = lib().orc_bench_chain
    fn.restype = C.c_double
    fn.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    out = np.zeros(1, vals.dtype)
    t = fn(int(vals.dtype == np.float64), _ptr(vals), vals.size, reps, _ptr(out))
    ref = filter_flat(SUM, DOUBLE_T if vals.dtype == np.float64 else FLOAT_T, vals)
    if out.tobytes() != np.asarray(ref[0]).tobytes():
        raise RuntimeError("orc_bench_chain: the tight fold differs from the oracle's fold")
    return t


def filter_tree(op, t, vals, cnt, fanin):
    vals = np.ascontiguousarray(vals, dtype=elem_dtype(t))
    if cnt is not None:
        cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    fan = np.ascontiguousarray(fanin, dtype=np.uint32)
    out = np.zeros(1, dtype=elem_dtype(t))
    oc = C.c_int32(0)
    ok = lib().orc_filter_tree(op, t, _ptr(vals), _ptr(cnt), vals.size, _ptr(fan), fan.size,
                               _ptr(out), C.byref(oc))
    if not ok:
        return None
    return out[0], oc.value


def message_build(hdr_arena, hdr_offs, hdr_lens, buf_arena, buf_offs, buf_lens):
    """src/Message.C:201-335: the bytes one Message::send puts on the link.  The spare byte
    behind the size vector (never written by the reference) is zero here."""
    n = len(hdr_lens)
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (hdr_arena, buf_arena)]
    u64 = [np.ascontiguousarray(a, dtype=np.uint64) for a in (hdr_offs, hdr_lens, buf_offs, buf_lens)]
    cap = 6 + 16 * n + int(u64[1].sum()) + int(u64[3].sum())
    out = np.zeros(cap, dtype=np.uint8)
    total = lib().orc_message_build(n, _ptr(arrs[0]), _ptr(u64[0]), _ptr(u64[1]), _ptr(arrs[1]),
                                    _ptr(u64[2]), _ptr(u64[3]), _ptr(out), cap)
    if total == 0 and cap != 0:
        raise RuntimeError("orc_message_build failed")
    return out[:total]


def message_parse(msg):
    """src/Message.C:48-199: (n, hdr_offs, hdr_lens, buf_offs, buf_lens) or None."""
    msg = np.ascontiguousarray(msg, dtype=np.uint8)
    cap = max(1, msg.size // 16)
    n = C.c_uint32(0)
    arrs = [np.zeros(cap, dtype=np.uint64) for _ in range(4)]
    ok = lib().orc_message_parse(_ptr(msg), msg.size, C.byref(n), _ptr(arrs[0]), _ptr(arrs[1]),
                                 _ptr(arrs[2]), _ptr(arrs[3]), cap)
    if not ok:
        return None
    return (n.value,) + tuple(a[: n.value].copy() for a in arrs)


def filter_perfdata(op, t, ranks, data):
    """tfilter_PerfData (src/FilterDefinitions.C:904-1239), one invocation.  ranks / data are
    (counts, values) pairs; returns (out_rank, out_data) -- the output packet is
    ([out_rank], [len(out_data)], out_data)."""
    rc, rv = (np.ascontiguousarray(ranks[0], dtype=np.uint64), np.ascontiguousarray(ranks[1], dtype=np.int32))
    dc, dv = (np.ascontiguousarray(data[0], dtype=np.uint64), np.ascontiguousarray(data[1], dtype=elem_dtype(t)))
    ro = np.zeros(rc.size + 1, np.uint64); np.cumsum(rc, out=ro[1:])
    do = np.zeros(dc.size + 1, np.uint64); np.cumsum(dc, out=do[1:])
    cap = int(dc.max()) if dc.size else 0
    out = np.zeros(max(cap, 1), dtype=elem_dtype(t))
    olen, orank = C.c_uint64(0), C.c_int32(0)
    ok = lib().orc_filter_perfdata(op, t, _ptr(rv), _ptr(rc), _ptr(ro), _ptr(dv), _ptr(dc), _ptr(do), dc.size,
                                   _ptr(out), cap, C.byref(olen), C.byref(orank))
    if not ok:
        return None
    return orank.value, out[: olen.value].copy()


def filter_perfdata_tree(op, t, ranks, data, fanins):
    """The same filter applied level by level: level L folds groups of fanins[L] consecutive
    packets of the level below (0 = all of them); a group without data yields an empty packet."""
    rc, rv = np.asarray(ranks[0], np.uint64), np.asarray(ranks[1], np.int32)
    dc, dv = np.asarray(data[0], np.uint64), np.asarray(data[1], elem_dtype(t))
    for fan in fanins:
        m = dc.size
        f = int(fan) if fan else max(m, 1)
        ro = np.zeros(m + 1, np.int64); np.cumsum(rc, out=ro[1:])
        do = np.zeros(m + 1, np.int64); np.cumsum(dc, out=do[1:])
        nrc, nrv, ndc, ndv = [], [], [], []
        for g in range(0, m, f):
            h = min(m, g + f)
            r = filter_perfdata(op, t, (rc[g:h], rv[ro[g]:ro[h]]), (dc[g:h], dv[do[g]:do[h]]))
            orank, odata = r
            if odata.size:
                nrc.append(1); nrv.append(orank); ndc.append(odata.size); ndv.append(odata)
            else:
                nrc.append(0); ndc.append(0)
        rc, rv = np.asarray(nrc, np.uint64), np.asarray(nrv, np.int32)
        dc = np.asarray(ndc, np.uint64)
        dv = np.concatenate(ndv) if ndv else np.zeros(0, elem_dtype(t))
        if dc.size <= 1:
            break
    if dc.size != 1:
        return None
    return (int(rv[0]) if rv.size else 0), dv


def filter_int_eq_class(vals, memcnts, mems):
    """tfilter_IntEqClass (src/FilterDefinitions.C:812-902).  vals / memcnts / mems are
    (counts, values) pairs of the three '%aud' fields; returns (values, memcnts, mems)."""
    vc, vv = np.ascontiguousarray(vals[0], np.uint64), np.ascontiguousarray(vals[1], np.uint32)
    mv = np.ascontiguousarray(memcnts[1], np.uint32)
    ec, ev = np.ascontiguousarray(mems[0], np.uint64), np.ascontiguousarray(mems[1], np.uint32)
    vo = np.zeros(vc.size + 1, np.uint64); np.cumsum(vc, out=vo[1:])
    eo = np.zeros(ec.size + 1, np.uint64); np.cumsum(ec, out=eo[1:])
    ov, om, oe = np.zeros(max(vv.size, 1), np.uint32), np.zeros(max(vv.size, 1), np.uint32), np.zeros(max(ev.size, 1), np.uint32)
    nc, nm = C.c_uint64(0), C.c_uint64(0)
    lib().orc_filter_int_eq_class(_ptr(vv), _ptr(vc), _ptr(vo), _ptr(mv), _ptr(ev), _ptr(eo), vc.size,
                                  _ptr(ov), _ptr(om), _ptr(oe), C.byref(nc), C.byref(nm))
    return ov[: nc.value].copy(), om[: nc.value].copy(), oe[: nm.value].copy()
