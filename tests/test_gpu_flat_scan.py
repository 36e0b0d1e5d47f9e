"""One reference filter call over float / double waves (tfilter_Sum, tfilter_Avg: every addition
rounds, src/FilterDefinitions.C:148, 1955-1960) computed by the binade-at-a-time scan (one workgroup
from 8,192 up to 16,383 floats, chunk summaries over the whole device from 16,384 floats / 8,192 doubles on) against the
plain chain of additions on the device (MRPDR_SHAPE_FLAT_SERIAL) and against the oracle's loop on the
CPU: the same bits on data chosen to break the scan -- ties at every step, sums hovering around a
power of two, cancellation, forty binades of dynamic range, zeros, subnormals, infinities, NaNs and
raw bit patterns.  Unpinned by reference artefacts beyond the captured sums (the reference's tests
only sum constants); the oracle restates the loop."""
import os

import numpy as np
import pytest
import torch

from wavegen import O, random_values, splitmix64

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402

SOAK = int(os.environ.get("MRPDR_SOAK", "1"))


@pytest.fixture(scope="module")
def codec():
    return W.Codec(max_packets=1 << 20)


def uniform(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def datasets(dt, n, seed):
    dt = np.dtype(dt)
    mant = 24 if dt.itemsize == 4 else 53
    u = uniform(seed, n)
    sign = np.where(splitmix64(seed + 1, n) & np.uint64(1), -1.0, 1.0)
    out = {}
    out["positive"] = u.astype(dt)
    out["mixed_sign"] = ((u - 0.5) * 8.0).astype(dt)
    out["filter_times"] = (1e-3 + 1e-4 * (uniform(seed + 2, n) - 0.5)).astype(dt)
    out["constant"] = np.full(n, 123.45, dt)
    # ties: a large head, then odd multiples of half an ulp of the running sum for a long while
    t = (2.0 * np.floor(u * 8.0) + 1.0) * sign
    t[0] = 2.0 ** mant
    out["ties"] = t.astype(dt)
    t2 = (2.0 * np.floor(u * 8.0) + 1.0) * 0.5
    t2[0] = 2.0 ** (mant - 1)
    out["ties_half"] = t2.astype(dt)
    # a sum that hovers around a power of two: +-1 steps around 2^k, in and out of the binade
    h = np.where(sign > 0, 1.0, -1.0) * (1.0 + np.floor(u * 3.0))
    h[0] = 1024.0
    out["hovering"] = h.astype(dt)
    # forty binades of range, both signs
    ex = (splitmix64(seed + 3, n) % np.uint64(41)).astype(np.float64) - 20.0
    out["wide_range"] = (sign * (1.0 + u) * np.exp2(ex)).astype(dt)
    # exact cancellation: pairs a, -a with a slow drift
    c = np.empty(n)
    c[0::2] = (1.0 + u[0::2]) * 1e3
    c[1::2] = -c[0::2][: len(c[1::2])] + 1e-3
    out["cancelling"] = c.astype(dt)
    # zeros, signed zeros and subnormals among small normals
    tiny = np.finfo(dt).tiny
    z = (u * 4.0 * tiny * sign).astype(dt)
    z[splitmix64(seed + 4, n) % np.uint64(3) == 0] = 0.0
    z[splitmix64(seed + 5, n) % np.uint64(7) == 0] = -0.0
    out["zeros_subnormals"] = z
    out["all_zero"] = np.zeros(n, dt)
    # infinities and NaNs dropped into ordinary data (the sum stays non-finite from there on)
    f = ((u - 0.3) * 100.0).astype(dt)
    where = splitmix64(seed + 6, n) % np.uint64(max(n // 3, 1)) == 0
    f[where] = np.array([np.inf, -np.inf, np.nan], dt)[(splitmix64(seed + 7, n) % np.uint64(3)).astype(np.int64)][where]
    f[: n // 2] = ((u - 0.3) * 100.0).astype(dt)[: n // 2]            # finite first half
    out["non_finite"] = f
    out["bit_patterns"] = random_values(dt, n, seed + 8, finite=False)
    out["finite_bit_patterns"] = random_values(dt, n, seed + 9)
    # excursions: from a sum near 1.5, a run of +3.99 far out of the sum's binade and back again, the
    # run's length chosen so that the partial sums, counted in units of the starting sum's last
    # place, come within a hair of what a signed integer of the element's width holds (float: 64
    # packets of 3.99 are 2^31 - 0.6 * 2^23 units; double: 513 are 2^63 - 1.1 * 2^52).  At every
    # chunk start of the device-wide form and in between.
    k = 64 if dt.itemsize == 4 else 513
    x = np.zeros(n)
    x[0] = 1.5
    for at in range(1024, n - 2 * k, 1024 + 2 * k + 37 if dt.itemsize == 4 else 2048):
        x[at:at + k] = 3.99
        x[at + k:at + 2 * k] = -3.99
    out["excursion"] = x.astype(dt)
    y = -x
    y[0] = -1.99                                                    # (the other side of zero, a sum near its binade's end)
    out["excursion_negative"] = y.astype(dt)
    # overflow to infinity and what follows
    big = (np.finfo(dt).max / 4.0 * (0.5 + u)).astype(dt)
    out["overflow"] = big
    return out


def same_bits(a, b):
    return np.asarray(a).tobytes() == np.asarray(b).tobytes()


def same_value(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b))     # NaN payloads differ between machines


@pytest.mark.parametrize("t,dt", [(O.FLOAT_T, np.float32), (O.DOUBLE_T, np.float64)])
@pytest.mark.parametrize("n", [2048, 8191, 8192, 8193, 12000, 16383, 16384, 16385, 65536, 300001, 1 << 20])
def test_scan_equals_the_chain(codec, t, dt, n):
    for rep in range(SOAK):
        for name, vals in datasets(dt, n, 1000 * n + 17 * t + 7919 * rep).items():
            dv = codec.to_device(vals)
            got = W.result_value(*codec.filter_columns(O.SUM, t, dv, None, W.SHAPE_FLAT), np.dtype(dt), O.SUM)
            chain = W.result_value(*codec.filter_columns(O.SUM, t, dv, None, W.SHAPE_FLAT_SERIAL), np.dtype(dt), O.SUM)
            # (a NaN result is compared as a NaN: which operand's payload an addition of two NaNs
            # hands on is the instruction's business, and the compiler is free to commute a sum)
            assert same_value(got[0], chain[0]), (name, n, "scan vs chain on the device", got[0], chain[0])
            with np.errstate(all="ignore"):
                ref = O.filter_flat(O.SUM, t, vals)
            assert same_value(got[0], ref[0]), (name, n, "scan vs oracle", got[0], ref[0])


@pytest.mark.parametrize("t,dt", [(O.FLOAT_T, np.float32), (O.DOUBLE_T, np.float64)])
@pytest.mark.parametrize("n", [9000, 65537])
def test_scan_avg_equals_the_chain(codec, t, dt, n):
    """AVG: value * count rounded in the element's type, then the same chain, then one division."""
    cnt = (splitmix64(n + 99, n) % np.uint64(5) + np.uint64(1)).astype(np.int32)
    dc = codec.to_device(cnt)
    for name, vals in datasets(dt, n, 31 * n + t).items():
        dv = codec.to_device(vals)
        got = W.result_value(*codec.filter_columns(O.AVG, t, dv, dc, W.SHAPE_FLAT), np.dtype(dt), O.AVG)
        chain = W.result_value(*codec.filter_columns(O.AVG, t, dv, dc, W.SHAPE_FLAT_SERIAL), np.dtype(dt), O.AVG)
        assert same_value(got[0], chain[0]) and got[1] == chain[1], (name, n, got, chain)
        with np.errstate(all="ignore"):
            ref = O.filter_flat(O.AVG, t, vals, cnt)
        assert same_value(got[0], ref[0]) and got[1] == ref[1], (name, n, got, ref)


def test_scan_on_the_wire_with_both_byte_orders(codec):
    """The scan reads the packets where they lie: a packed '%f %d' wave from a big-endian sender."""
    n = 70000
    vals = datasets(np.float32, n, 5)["mixed_sign"]
    cnt = (splitmix64(3, n) % np.uint64(4) + np.uint64(1)).astype(np.int32)
    rows = np.zeros((n, 8), np.uint8)
    rows[:, :4] = vals.view(np.uint8).reshape(n, 4)[:, ::-1]
    rows[:, 4:] = cnt.view(np.uint8).reshape(n, 4)[:, ::-1]
    wire = W.WireWave(codec.to_device(rows.reshape(-1)), n, None, None, 8)
    got = W.result_value(*codec.filter_wire(O.AVG, W.Format("%f %d"), wire, W.BIGENDIAN), np.dtype(np.float32), O.AVG)
    ref = O.filter_flat(O.AVG, O.FLOAT_T, vals, cnt)
    assert same_bits(got[0], ref[0]) and got[1] == ref[1]


@pytest.mark.parametrize("t,dt", [(O.FLOAT_T, np.float32), (O.DOUBLE_T, np.float64)])
@pytest.mark.parametrize("n", [12000, 70001])
def test_scan_over_packets_of_both_byte_orders(codec, t, dt, n):
    """A wave collected from senders of both byte orders, every packet read with its own
    (src/Packet.C:118): '%f %d' / '%lf %d' rows, tfilter_Avg and tfilter_Sum of one reference call."""
    dt = np.dtype(dt)
    w = dt.itemsize
    for kind in ("positive", "mixed_sign", "excursion"):
        vals = datasets(dt, n, 77 + n)[kind]
        cnt = (splitmix64(5 + n, n) % np.uint64(4) + np.uint64(1)).astype(np.int32)
        orders = (splitmix64(9 + n, n) % np.uint64(3) == 0).astype(np.uint8)
        rows = np.zeros((n, w + 4), np.uint8)
        vb, cb = vals.view(np.uint8).reshape(n, w), cnt.view(np.uint8).reshape(n, 4)
        rows[:, :w] = np.where(orders[:, None] != 0, vb[:, ::-1], vb)
        rows[:, w:] = np.where(orders[:, None] != 0, cb[:, ::-1], cb)
        dev, dord = codec.to_device(rows.reshape(-1)), codec.to_device(orders)
        for op in (O.AVG, O.SUM):
            res, ok = codec.filter_raw(op, t, dev.data_ptr(), w + 4, (dev.data_ptr() + w) if op == O.AVG else None, w + 4, n,
                                       dord, W.SHAPE_FLAT)
            torch.cuda.synchronize()
            got = W.result_value(res, ok, dt, op)
            with np.errstate(all="ignore"):
                ref = O.filter_flat(op, t, vals, cnt if op == O.AVG else None)
            assert same_value(got[0], ref[0]), (kind, n, op, got, ref)
            if op == O.AVG:
                assert got[1] == ref[1]


@pytest.mark.parametrize("t,dt,n", [(O.FLOAT_T, np.float32, 600001), (O.DOUBLE_T, np.float64, 300001)])
def test_one_workgroup_scan_for_callers_without_room(codec, t, dt, n):
    """The one-workgroup form of the scan (k_filter_flat_scan) is what callers without room for the
    device-wide form's summaries get: mrpdr_wave_filter_noalloc beyond what its built-in array holds
    (536,000 floats, 273,000 doubles).  Same data kinds, against the oracle's chain and against the
    ctx entry point (which takes the device-wide form)."""
    from mrnet_amd import capi
    lib = capi.load()
    st = torch.cuda.current_stream().cuda_stream
    for name, vals in datasets(dt, n, 5 * n + t).items():
        dv = codec.to_device(vals)
        res = torch.zeros(16, dtype=torch.uint8, device=codec.device)
        ok = torch.zeros(1, dtype=torch.int32, device=codec.device)
        assert lib.mrpdr_wave_filter_noalloc(O.SUM, int(t), W.SHAPE_FLAT, dv.data_ptr(), np.dtype(dt).itemsize, None, 0,
                                             n, 0, res.data_ptr(), ok.data_ptr(), st) == 1, capi.last_error()
        torch.cuda.synchronize()
        got = W.result_value(res, ok, np.dtype(dt), O.SUM)
        wide = W.result_value(*codec.filter_columns(O.SUM, t, dv, None, W.SHAPE_FLAT), np.dtype(dt), O.SUM)
        with np.errstate(all="ignore"):
            ref = O.filter_flat(O.SUM, t, vals)
        assert same_value(got[0], ref[0]), (name, n, "one workgroup vs oracle", got[0], ref[0])
        assert same_value(got[0], wide[0]), (name, n, "one workgroup vs device-wide", got[0], wide[0])
