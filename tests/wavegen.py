"""Seeded synthetic waves for the parity tests (numpy columns in the oracle's form)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402  (test infrastructure)


def splitmix64(seed, n):
    """n 64-bit values of the splitmix64 sequence started at `seed` (vectorised)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_values(dtype, n, seed, finite=True):
    """n values of dtype from the seed; floats are finite bit patterns unless finite=False."""
    dtype = np.dtype(dtype)
    bits = splitmix64(seed, n)
    if dtype.kind == "f":
        if dtype.itemsize == 4:
            b = (bits >> np.uint64(32)).astype(np.uint32)
            if finite:
                exp = (b >> np.uint32(23)) & np.uint32(0xFF)
                b = np.where(exp == 0xFF, b & np.uint32(0xBF7FFFFF), b)
            return b.view(np.float32)
        b = bits.copy()
        if finite:
            exp = (b >> np.uint64(52)) & np.uint64(0x7FF)
            b = np.where(exp == 0x7FF, b & np.uint64(0xBFEFFFFFFFFFFFFF), b)
        return b.view(np.float64)
    return bits.astype(np.dtype("u%d" % dtype.itemsize)).view(dtype) if dtype.itemsize < 8 \
        else bits.view(dtype)


def make_wave(fmt, n, seed, max_count=40, counts=None, printable=True):
    """Random columns for `fmt`.  counts: optional {field_index: array of counts}."""
    types = O.fmt_types(fmt)
    cols = []
    for f, t in enumerate(types):
        k = O.type_kind(t)
        dt = O.elem_dtype(t)
        s = seed * 1000 + f * 7 + 1
        c = {"type": t}
        if k == 0:
            c["values"] = random_values(dt, n, s)
        elif k == 1:
            cnt = (counts[f] if counts and f in counts
                   else splitmix64(s + 1, n) % np.uint64(max_count + 1)).astype(np.uint64)
            offs = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(cnt, out=offs[1:])
            c["counts"], c["offs"] = cnt, offs
            c["values"] = random_values(dt, int(offs[-1]), s + 2)
        elif k == 2:
            ln = (counts[f] if counts and f in counts
                  else splitmix64(s + 1, n) % np.uint64(max_count + 1)).astype(np.uint64)
            cnt = ln + np.uint64(1)
            offs = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(cnt, out=offs[1:])
            body = (splitmix64(s + 2, int(offs[-1])) % np.uint64(94) + np.uint64(33)).astype(np.uint8)
            body[(offs[1:] - np.uint64(1)).astype(np.int64)] = 0       # the NUL of every string
            c["counts"], c["offs"], c["values"] = cnt, offs, body
        else:
            ns = (splitmix64(s + 1, n) % np.uint64(4)).astype(np.uint64)
            offs = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(ns, out=offs[1:])
            S = int(offs[-1])
            sl = (splitmix64(s + 3, S) % np.uint64(9) + np.uint64(1)).astype(np.uint64)
            so = np.zeros(S + 1, dtype=np.uint64)
            np.cumsum(sl, out=so[1:])
            body = (splitmix64(s + 2, int(so[-1])) % np.uint64(94) + np.uint64(33)).astype(np.uint8)
            if S:
                body[(so[1:] - np.uint64(1)).astype(np.int64)] = 0
            c.update(counts=ns, offs=offs, sub_counts=sl, sub_offs=so, values=body)
        cols.append(c)
    return cols


def to_sender_order(fmt, arena, offsets, lens):
    """The same packets as a big-endian sender would have written them: every scalar,
    array element and length prefix byte-reversed with the reference's byte_swap; raw
    chars and string bytes untouched (src/pdr_mem.c:394-412)."""
    types = O.fmt_types(fmt)
    out = arena.copy()
    kinds = [O.type_kind(t) for t in types]
    widths = [O.type_width(t) for t in types]

    def rev(a, b, w):
        if w > 1 and b > a:
            out[a:b] = O.byte_swap(arena[a:b], w)

    n = len(lens)
    if n > 1 and all(k == 0 for k in kinds):
        # scalar-only format in a constant-stride arena: swap whole columns at once
        strides = np.diff(offsets[: n + 1].astype(np.int64))
        if np.all(strides == strides[0]) and np.all(lens == sum(widths)):
            stride = int(strides[0])
            rows_in = arena[: n * stride].reshape(n, stride)
            rows_out = out[: n * stride].reshape(n, stride)
            at = 0
            for w in widths:
                rows_out[:, at:at + w] = rows_in[:, at:at + w][:, ::-1]
                at += w
            return out
    for i in range(len(lens)):
        p = int(offsets[i])
        for k, w in zip(kinds, widths):
            if k == 0:
                rev(p, p + w, w); p += w
            elif k == 1:
                cnt = int(arena[p:p + 8].view(np.uint64)[0])
                rev(p, p + 8, 8); p += 8
                rev(p, p + cnt * w, w); p += cnt * w
            elif k == 2:
                cnt = int(arena[p:p + 4].view(np.uint32)[0])
                rev(p, p + 4, 4); p += 4 + cnt
            else:
                ns = int(arena[p:p + 8].view(np.uint64)[0])
                rev(p, p + 8, 8); p += 8
                for _ in range(ns):
                    cnt = int(arena[p:p + 4].view(np.uint32)[0])
                    rev(p, p + 4, 4); p += 4 + cnt
        assert p == int(offsets[i]) + int(lens[i])
    return out


def columns_equal(a, b):
    """Compare two oracle-form column lists bit for bit."""
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        for key in ("values", "counts", "offs", "sub_counts", "sub_offs"):
            u, v = x.get(key), y.get(key)
            if (u is None) != (v is None):
                return False
            if u is not None:
                u = np.ascontiguousarray(u).view(np.uint8)
                v = np.ascontiguousarray(v).view(np.uint8)
                if u.shape != v.shape or not np.array_equal(u, v):
                    return False
    return True
