"""Randomly drawn format strings (every token kind, 1-12 tokens), wave sizes, arena layouts and
sender byte orders: device encode / decode against the oracle, bit for bit.  Seeds are fixed."""
import numpy as np
import pytest
import torch

from wavegen import O, columns_equal, make_wave, splitmix64, to_sender_order
from test_gpu_codec import dev_cols, host_cols, wire_from_numpy

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402

import os

# MRPDR_SOAK=k multiplies the number of drawn cases (a longer run of the same tests, by hand)
SOAK = max(1, int(os.environ.get("MRPDR_SOAK", "1")))

SCALARS = ["c", "uc", "hd", "uhd", "d", "ud", "ld", "uld", "f", "lf"]
VARS = ["s", "as", "As"] + ["a" + t for t in SCALARS] + ["A" + t for t in SCALARS]
SEPS = [" ", "  ", "\t", " \n"]


def draw(seed):
    r = splitmix64(seed, 64)
    ntok = int(r[0] % np.uint64(12)) + 1
    p_var = [0.0, 0.3, 0.6][int(r[1] % np.uint64(3))]
    toks = []
    for k in range(ntok):
        if (int(r[2 + k]) % 1000) / 1000.0 < p_var:
            toks.append(VARS[int(r[20 + k] % np.uint64(len(VARS)))])
        else:
            toks.append(SCALARS[int(r[20 + k] % np.uint64(len(SCALARS)))])
    fmt = "%" + toks[0]
    for k in range(1, ntok):
        fmt += SEPS[int(r[40 + k] % np.uint64(len(SEPS)))] + "%" + toks[k]
    n = [0, 1, 2, 63, 64, 65, 300, 1025, 4000][int(r[60] % np.uint64(9))]
    max_count = [0, 1, 5, 40, 700][int(r[61] % np.uint64(5))]
    align = [1, 2, 8, 16, 128][int(r[62] % np.uint64(5))]
    phase = int(r[63] % np.uint64(align)) if align > 1 else [0, 29][int(r[63] % np.uint64(2))]
    return fmt, n, max_count, align, phase


def blank_like(cols):
    """Columns of the same capacities, filled with a pattern."""
    def blank(t):
        return None if t is None else torch.full_like(t, 0x5A)
    return [W.Column(blank(c.values), blank(c.counts), blank(c.offs), c.capacity, blank(c.sub_counts),
                     blank(c.sub_offs), c.sub_capacity) for c in cols]


@pytest.mark.parametrize("seed", range(120 * SOAK))
def test_random_format_round_trip(codec, seed):
    fmt_s, n, max_count, align, phase = draw(1000 + seed)
    if n * max_count > 400000:
        max_count = 40
    ocols = make_wave(fmt_s, n, seed, max_count=max_count)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n, align=align, phase=phase)
    fmt = W.Format(fmt_s)
    cols = dev_cols(codec, fmt, ocols)
    dl, do, total = codec.sizeof(fmt, cols, n, align, phase)
    if fmt.is_fixed:
        # the packed closed form lays packets out by the same rule
        assert np.array_equal(W.Codec.to_numpy(do).view(np.uint64), offs)
    else:
        assert np.array_equal(W.Codec.to_numpy(dl).view(np.uint64), lens), fmt_s
        assert np.array_equal(W.Codec.to_numpy(do).view(np.uint64), offs), fmt_s
    assert total == int(offs[n])
    wire = W.WireWave(torch.zeros(max(total, 1), dtype=torch.uint8, device=codec.device), n, do,
                      dl if not fmt.is_fixed else codec.to_device(lens) if n else dl, 0)
    codec.encode_into(fmt, cols, wire)
    torch.cuda.synchronize()
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[: arena.size], arena), (fmt_s, n, align, phase)
    bo = W.BIGENDIAN if seed % 2 else W.LITTLEENDIAN
    src = to_sender_order(fmt_s, arena, offs, lens) if bo == W.BIGENDIAN else arena
    dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, src, offs, lens), bo)
    assert bool(ok.all()) if n else True
    assert columns_equal(host_cols(fmt, dcols, n), ocols), (fmt_s, n, bo)
    assert codec.status() == 0
    # the one-call form into columns the caller sized (no totals read back in between)
    cols2 = blank_like(dcols)
    ok2 = torch.zeros_like(ok)
    codec.decode_into(fmt, wire_from_numpy(codec, n, src, offs, lens), cols2, ok2, bo)
    assert torch.equal(ok2, ok)
    assert columns_equal(host_cols(fmt, cols2, n), ocols), (fmt_s, n, bo, "decode_into")
    assert codec.status() == 0


@pytest.mark.parametrize("seed", range(40 * SOAK))
def test_random_truncation(codec, seed):
    """Every packet cut at a random length (a third of them left whole): per-packet ok flags and
    the columns of what did decode, against the oracle."""
    fmt_s, n, max_count, _, _ = draw(5000 + seed)
    n = max(n, 65)
    if n * max_count > 200000:
        max_count = 40
    ocols = make_wave(fmt_s, n, seed, max_count=max_count)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    r = splitmix64(77 + seed, n)
    cut = np.where(r % np.uint64(3) == 0, lens, r % (lens + np.uint64(1))).astype(np.uint64)
    bo = W.BIGENDIAN if seed % 2 else W.LITTLEENDIAN
    src = to_sender_order(fmt_s, arena, offs, lens) if bo == W.BIGENDIAN else arena
    ref_cols, ref_ok = O.wave_decode(fmt_s, src, offs, cut, bo)
    fmt = W.Format(fmt_s)
    if fmt.is_fixed:
        wire = wire_from_numpy(codec, n, src, None, cut, stride=fmt.fixed_bytes)
    else:
        wire = wire_from_numpy(codec, n, src, offs, cut)
    dcols, ok = codec.decode(fmt, wire, bo)
    assert np.array_equal(W.Codec.to_numpy(ok), ref_ok), fmt_s
    assert columns_equal(host_cols(fmt, dcols, n), ref_cols), fmt_s
    cols2 = blank_like(dcols)
    ok2 = torch.zeros_like(ok)
    codec.decode_into(fmt, wire, cols2, ok2, bo)
    assert torch.equal(ok2, ok) and columns_equal(host_cols(fmt, cols2, n), ref_cols), (fmt_s, "decode_into")


def mixed_arena(fmt_s, arena, offs, lens, orders):
    """Every packet as its own sender wrote it: orders[i] != 0 = big-endian."""
    be = to_sender_order(fmt_s, arena, offs, lens)
    out = arena.copy()
    for i in np.nonzero(orders)[0]:
        a, b = int(offs[i]), int(offs[i]) + int(lens[i])
        out[a:b] = be[a:b]
    return out


@pytest.mark.parametrize("seed", range(60 * SOAK))
def test_random_mixed_byte_orders(codec, seed):
    """A wave collected from senders of both byte orders (src/Packet.C:96,118: every packet is
    decoded with the order its message announced), whole and truncated packets."""
    fmt_s, n, max_count, align, phase = draw(9000 + seed)
    n = max(n, 2)
    if n * max_count > 200000:
        max_count = 40
    ocols = make_wave(fmt_s, n, seed, max_count=max_count)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n, align=align, phase=phase)
    orders = (splitmix64(31 + seed, n) % np.uint64(3) == 0).astype(np.uint8)     # a third are foreign
    if seed % 5 == 0:
        orders[:] = 1
    src = mixed_arena(fmt_s, arena, offs, lens, orders)
    cut = lens
    if seed % 2:
        r = splitmix64(78 + seed, n)
        cut = np.where(r % np.uint64(3) == 0, lens, r % (lens + np.uint64(1))).astype(np.uint64)
    ref_cols, ref_ok = O.wave_decode(fmt_s, src, offs, cut, orders)
    if cut is lens:
        assert columns_equal(ref_cols, ocols) and bool(ref_ok.all())
    fmt = W.Format(fmt_s)
    dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, src, offs, cut), codec.to_device(orders))
    assert np.array_equal(W.Codec.to_numpy(ok), ref_ok), fmt_s
    assert columns_equal(host_cols(fmt, dcols, n), ref_cols), (fmt_s, n)
    assert codec.status() == 0


def test_mixed_orders_large_arrays(codec):
    """Arrays long enough for the chunk kernel, then the in-place reversal of the foreign packets."""
    n, m = 96, 40000
    fmt_s = "%d %alf %ahd"
    counts = {1: np.full(n, m, dtype=np.uint64), 2: (splitmix64(4, n) % np.uint64(3000)).astype(np.uint64)}
    ocols = make_wave(fmt_s, n, 12, counts=counts)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    orders = (np.arange(n) % 2).astype(np.uint8)
    src = mixed_arena(fmt_s, arena, offs, lens, orders)
    fmt = W.Format(fmt_s)
    dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, src, offs, lens), codec.to_device(orders))
    assert bool(ok.all())
    assert columns_equal(host_cols(fmt, dcols, n), ocols)


@pytest.mark.parametrize("seed", range(40 * SOAK))
def test_random_garbage(codec, seed):
    """Packets of random bytes (length prefixes of any size, zero string lengths, counts beyond the
    packet): the per-packet verdicts and whatever decoded before the first bad field must equal the
    oracle's, and nothing may be read or written outside the buffers."""
    fmt_s, n, _, _, _ = draw(7000 + seed)
    n = max(n, 64)
    r = splitmix64(900 + seed, n)
    lens = (r % np.uint64(120)).astype(np.uint64)
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens + np.uint64(3), out=offs[1:])
    arena = (splitmix64(901 + seed, int(offs[n]) + 8) & np.uint64(0xFF)).astype(np.uint8)
    if seed % 3 == 0:
        # mostly small numbers, so that some prefixes are plausible and decoding gets further
        arena[(splitmix64(902 + seed, arena.size) % np.uint64(4)) != 0] = 0
    bo = W.BIGENDIAN if seed % 2 else W.LITTLEENDIAN
    ref_cols, ref_ok = O.wave_decode(fmt_s, arena, offs, lens, bo)
    fmt = W.Format(fmt_s)
    if fmt.is_fixed:
        return                                       # scalar-only formats have no prefixes to corrupt
    dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, arena, offs, lens), bo)
    assert np.array_equal(W.Codec.to_numpy(ok), ref_ok), fmt_s
    assert columns_equal(host_cols(fmt, dcols, n), ref_cols), fmt_s
    assert codec.status() == 0


def test_random_garbage_headers(codec):
    """Headers of random bytes, and good headers with one byte damaged: ok[i] must be set exactly
    when the oracle's header decode succeeds with the expected format string."""
    fmt_s = "%d %s"
    fmt = W.Format(fmt_s)
    good = O.header_encode(77, 101, 5, fmt_s)
    n = 600
    hl = good.size + 9
    rows = np.zeros((n, hl), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint64)
    r = splitmix64(4242, n * 4).reshape(n, 4)
    for i in range(n):
        kind = int(r[i, 0] % np.uint64(4))
        if kind == 0:                                # random bytes, random length
            l = int(r[i, 1] % np.uint64(hl + 1))
            rows[i, :l] = (splitmix64(int(r[i, 2] % np.uint64(1 << 30)), l) & np.uint64(0xFF)).astype(np.uint8)
        elif kind == 1:                              # good header, one byte damaged
            l = good.size
            rows[i, :l] = good
            rows[i, int(r[i, 1] % np.uint64(l))] ^= np.uint8(1 + int(r[i, 2] % np.uint64(255)))
        elif kind == 2:                              # good header, cut short or padded
            l = int(r[i, 1] % np.uint64(hl + 1))
            rows[i, :good.size] = good
        else:
            l = good.size
            rows[i, :l] = good
        lens[i] = l
    want = np.zeros(n, dtype=np.int32)
    for i in range(n):
        d = O.header_decode(rows[i, :int(lens[i])]) if lens[i] else None
        want[i] = int(d is not None and d["fmt"] == fmt_s)
    offs = (np.arange(n, dtype=np.uint64) * np.uint64(hl))
    out = codec.decode_headers(fmt, n, codec.to_device(rows.reshape(-1)), offsets=codec.to_device(offs),
                               lens=codec.to_device(lens), max_dest=4)
    got = W.Codec.to_numpy(out["ok"])
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:5], [(rows[i, :int(lens[i])].tobytes().hex(), int(want[i])) for i in bad[:3]])
    assert want.sum() > n // 8


@pytest.mark.parametrize("fmt_s,n,max_count,field", [
    ("%d %s %auc", 500, 30, 2),           # short packets: tile kernels
    ("%af", 6, 100000, 0),                # long arrays: chunk kernel
    ("%ahd %ad", 3000, 900, 1),           # mid-size payloads, two fields
    ("%d %as", 200, 5, 1),                # string arrays: bytes
    ("%auc %auc %auc %auc %auc %auc %auc %auc %auc", 300, 20, 8),   # more variable fields than a tile takes
])
def test_decode_into_columns_too_small(codec, fmt_s, n, max_count, field):
    """The one-call decode into columns that cannot hold the wave: nothing is written to the
    column that is too small, the status word says so, and reading it clears it."""
    counts = {0: np.full(n, max_count, np.uint64)} if fmt_s == "%af" else None
    ocols = make_wave(fmt_s, n, 3, max_count=max_count, counts=counts)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    fmt = W.Format(fmt_s)
    wire = wire_from_numpy(codec, n, arena, offs, lens)
    dcols, ok = codec.decode(fmt, wire, W.LITTLEENDIAN)
    assert codec.status() == 0
    cols2 = blank_like(dcols)
    cols2[field].capacity = int(dcols[field].capacity) - 1
    guard = cols2[field].values.clone()
    ok2 = torch.zeros_like(ok)
    codec.decode_into(fmt, wire, cols2, ok2, W.LITTLEENDIAN)
    assert codec.status() & 1
    assert torch.equal(cols2[field].values, guard)
    assert codec.status() == 0                      # read and cleared


@pytest.mark.parametrize("seed", range(30 * SOAK))
def test_random_headers(codec, seed):
    """Drawn format string, wave size, per-packet stream ids / tags / source ranks and a destination
    list: the encoded headers against the oracle's byte for byte, then decoded again -- also after
    rewriting every other header as a big-endian sender would have."""
    fmt_s, n, _, _, _ = draw(11000 + seed)
    n = max(n, 1)
    r = splitmix64(12000 + seed, 8)
    ndest = [0, 0, 1, 3, 17][int(r[0] % np.uint64(5))]
    dest = (splitmix64(seed + 1, ndest) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    sids = (splitmix64(seed + 2, n) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    tags = (splitmix64(seed + 3, n) & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
    srcs = (splitmix64(seed + 4, n) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    fmt = W.Format(fmt_s)
    hdrs, stride = codec.encode_headers(fmt, n, stream_ids=codec.to_device(sids), tags=codec.to_device(tags),
                                        src_ranks=codec.to_device(srcs),
                                        dest_ranks=codec.to_device(dest) if ndest else None)
    torch.cuda.synchronize()
    h = W.Codec.to_numpy(hdrs)[:n * stride].reshape(n, stride).copy()
    for i in sorted(set([0, n // 2, n - 1] + [int(x % np.uint64(n)) for x in splitmix64(seed + 5, 6)])):
        want = O.header_encode(int(sids[i]), int(tags[i]), int(srcs[i]), fmt_s, dest=dest.tolist())
        assert want.size == stride and h[i].tobytes() == want.tobytes(), (fmt_s, i)
    flen = len(fmt_s) + 1
    for i in range(1, n, 2):                        # every other header from a big-endian sender
        for a in (0, 4, 8, 12):
            h[i, a:a + 4] = h[i, a:a + 4][::-1].copy()
        e = 16 + flen
        h[i, e:e + 8] = h[i, e:e + 8][::-1].copy()
        for k in range(ndest):
            h[i, e + 8 + 4 * k:e + 12 + 4 * k] = h[i, e + 8 + 4 * k:e + 12 + 4 * k][::-1].copy()
        h[i, stride - 1] = 1
    out = codec.decode_headers(fmt, n, codec.to_device(h.reshape(-1)), stride=stride, max_dest=max(ndest, 1))
    assert bool(out["ok"].all())
    assert np.array_equal(W.Codec.to_numpy(out["stream_id"]).view(np.uint32), sids)
    assert np.array_equal(W.Codec.to_numpy(out["tag"]), tags)
    assert np.array_equal(W.Codec.to_numpy(out["src_rank"]).view(np.uint32), srcs)
    assert np.array_equal(W.Codec.to_numpy(out["byteorder"]), (np.arange(n) % 2).astype(np.uint8))
    assert bool((out["ndest"] == ndest).all())
    if ndest:
        assert np.array_equal(W.Codec.to_numpy(out["dest_ranks"]).view(np.uint32), np.tile(dest, (n, 1)))


@pytest.mark.parametrize("fmt_s", ["%d", "%lf %d", "%c %uhd %f %uld", "%hd %hd %hd"])
@pytest.mark.parametrize("n", [1, 255, 256, 257, 5000])
def test_packed_scalar_wave_of_both_byte_orders(codec, fmt_s, n):
    """Scalar-only packets side by side (the layout the LDS-staged row kernels take), every packet
    in its own sender's byte order."""
    ocols = make_wave(fmt_s, n, n + len(fmt_s))
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    orders = (splitmix64(n, n) % np.uint64(2)).astype(np.uint8)
    src = mixed_arena(fmt_s, arena, offs, lens, orders)
    fmt = W.Format(fmt_s)
    dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, src, None, None, stride=fmt.fixed_bytes), codec.to_device(orders))
    assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)
