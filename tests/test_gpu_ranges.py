"""One reference filter call over a wave that is held in ranges (mrpdr_wave_filter_partial /
mrpdr_wave_filter_combine): the packets of a wave are split into 2 and 8 consecutive ranges, as the
GPUs of a node hold them, every range is reduced to its 16-byte node, and the combined result is
compared with the oracle's ONE flat fold over the WHOLE wave (orc_filter_flat, the loop of
src/FilterDefinitions.C:101-149, 517-561 restated) -- never with the oracle applied per range.

Bar: bits, for every type and operator, the floating sums included (their chain of additions is
continued from range to range, so there is nothing to tolerate)."""
import numpy as np
import pytest
import torch

from wavegen import O, random_values, splitmix64
from test_gpu_filters import TYPES, OPS, counts_for, values_for

pytestmark = pytest.mark.gpu

from mrnet_amd import dist as D  # noqa: E402
from mrnet_amd import wave as W  # noqa: E402


def _rows(vals, cnt):
    """The packed wire rows of a '%T' / '%T %d' wave (little-endian sender)."""
    w = vals.dtype.itemsize
    stride = w + (4 if cnt is not None else 0)
    rows = np.zeros((vals.size, stride), np.uint8)
    rows[:, :w] = vals.view(np.uint8).reshape(-1, w)
    if cnt is not None:
        rows[:, w:] = cnt.view(np.uint8).reshape(-1, 4)
    return rows, stride


def ranged(codec, op, t, vals, cnt, edges, swap=False, always_front=False, mixed=False):
    """Nodes of the ranges [edges[r], edges[r+1]) as G ranks would produce them, then the combine.
    mixed: every third packet comes from a sender of the other byte order (one order per packet)."""
    rows, stride = _rows(vals, cnt)
    w = vals.dtype.itemsize
    orders = None
    if swap or mixed:
        sel = slice(None) if swap else slice(1, None, 3)
        rows[sel, :w] = rows[sel, :w][:, ::-1]
        if cnt is not None:
            rows[sel, w:] = rows[sel, w:][:, ::-1]
        if mixed:
            oh = np.zeros(max(1, vals.size), np.uint8)
            oh[1::3] = 1
            orders = codec.to_device(oh)
    dev = codec.to_device(rows.reshape(-1)) if rows.size else codec.empty(16, torch.uint8)
    base = dev.data_ptr()
    chained = codec.is_chained(op, t)
    nodes, front = [], None
    for lo, hi in zip(edges[:-1], edges[1:]):
        node = codec.filter_partial_raw(op, t, base + lo * stride, stride, base + lo * stride + w, stride, hi - lo,
                                        orders[lo:] if mixed else int(swap), front if (chained or always_front) else None)
        nodes.append(node)
        front = node
    if chained or always_front:
        res, ok = codec.filter_combine(op, t, nodes[-1] if always_front and not chained else torch.stack(nodes))
    else:
        res, ok = codec.filter_combine(op, t, torch.stack(nodes))
    torch.cuda.synchronize()
    return res, ok, nodes


def check(codec, op, t, dt, vals, cnt, edges, what, **kw):
    res, ok, _ = ranged(codec, op, t, vals, cnt, edges, **kw)
    got = W.result_value(res, ok, np.dtype(dt), op)
    ref = O.filter_flat(op, t, vals, cnt)
    if ref is None:
        assert got is None, what
        return
    assert got is not None, what
    if np.dtype(dt).kind == "f" and np.isnan(ref[0]):
        assert np.isnan(got[0]), what           # (which NaN payload an addition hands on is not defined)
    else:
        assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), (what, got, ref)
    if op == O.AVG:
        assert got[1] == ref[1], what


def even_edges(n, g):
    return [D.shard_range(n, r, g)[0] for r in range(g)] + [n]


@pytest.mark.parametrize("tok,t,dt", TYPES)
@pytest.mark.parametrize("opname,op", OPS)
def test_ranges_equal_one_flat_call(codec, tok, t, dt, opname, op):
    for n in (1, 2, 7, 64, 1000, 4097, 70001):
        vals = values_for(dt, n, 31 * n + t, op)
        cnt = counts_for(n, n + 5) if op == O.AVG else None
        for g in (1, 2, 3, 8):
            check(codec, op, t, dt, vals, cnt, even_edges(n, g), (tok, opname, n, g))
        # ragged: empty ranges at both ends and in the middle
        cut = sorted(int(x) for x in splitmix64(n, 5) % np.uint64(n + 1))
        check(codec, op, t, dt, vals, cnt, [0, 0] + cut + [n, n], (tok, opname, n, "ragged"))
        # a sender of the other byte order; senders of both (mrpdr_wave_filter_partial_mixed)
        check(codec, op, t, dt, vals, cnt, even_edges(n, 2), (tok, opname, n, "swapped"), swap=True)
        check(codec, op, t, dt, vals, cnt, even_edges(n, 3), (tok, opname, n, "mixed"), mixed=True)


@pytest.mark.parametrize("fmt,t,dt,op", [
    ("%d %d", O.INT32_T, np.int32, O.AVG), ("%hd %d", O.INT16_T, np.int16, O.AVG),
    ("%uld %d", O.UINT64_T, np.uint64, O.AVG), ("%f", O.FLOAT_T, np.float32, O.SUM),
    ("%lf %d", O.DOUBLE_T, np.float64, O.AVG), ("%f %d", O.FLOAT_T, np.float32, O.AVG),
    ("%lf", O.DOUBLE_T, np.float64, O.SUM), ("%d", O.INT32_T, np.int32, O.SUM),
    ("%f", O.FLOAT_T, np.float32, O.MIN), ("%lf", O.DOUBLE_T, np.float64, O.MAX)])
def test_a_million_packets_in_2_and_8_ranges(codec, fmt, t, dt, op):
    """The full-size wave of the sharded benchmark configuration, values in [0, 1000) as a monitoring
    stream carries them (integer averages then differ visibly between one division and one per range)."""
    n = 1 << 20
    u = splitmix64(4242 + t, n)
    if np.dtype(dt).kind == "f":
        vals = ((u % np.uint64(1_000_000)).astype(np.float64) / 1e3).astype(dt)
    else:
        vals = (u % np.uint64(1000)).astype(dt)
    cnt = (splitmix64(77, n) % np.uint64(4) + np.uint64(1)).astype(np.int32) if op == O.AVG else None
    for g in (2, 8):
        check(codec, op, t, dt, vals, cnt, even_edges(n, g), (fmt, g))


def test_front_is_honoured_for_order_free_operators(codec):
    """front may be passed for operators that do not need it: the node then covers front + range,
    and the last node alone combines to the same result."""
    n = 9000
    for tok, t, dt in TYPES:
        for opname, op in OPS:
            vals = values_for(dt, n, 5 + t, op)
            cnt = counts_for(n, 3) if op == O.AVG else None
            check(codec, op, t, dt, vals, cnt, even_edges(n, 4), (tok, opname, "front"), always_front=True)


def test_nan_restarts_across_ranges(codec):
    """(r > x) ? r : x keeps the NEW element whenever the comparison is false, so a NaN restarts the
    fold: NaNs at range ends, at range starts and as whole ranges."""
    for dt, t in ((np.float32, O.FLOAT_T), (np.float64, O.DOUBLE_T)):
        n = 4096
        for seed in range(12):
            vals = values_for(dt, n, 900 + seed, O.MIN).copy()
            edges = even_edges(n, 8)
            pos = splitmix64(seed, 6) % np.uint64(n)
            vals[pos.astype(np.int64)] = np.nan
            if seed % 3 == 0:
                vals[edges[3] - 1] = np.nan                    # last packet of a range
            if seed % 3 == 1:
                vals[edges[5]] = np.nan                        # first packet of a range
            if seed % 4 == 2:
                vals[edges[2]:edges[3]] = np.nan               # a whole range
            if seed == 11:
                vals[-1] = np.nan                              # the NaN that survives
            for op in (O.MIN, O.MAX):
                check(codec, op, t, dt, vals, None, edges, ("nan", seed, op))


def test_avg_divides_once(codec):
    """tfilter_Avg multiplies, sums and divides ONCE (src/FilterDefinitions.C:517-561): ranges whose
    own averages would truncate differently, counts that sum to zero in one range, in all ranges."""
    vals = np.array([7, 8, 10, 1, 1, 1, 999, 3], np.int32)
    cnt = np.array([1, 2, 4, 1, 1, 1, 1, 2], np.int32)
    check(codec, O.AVG, O.INT32_T, np.int32, vals, cnt, [0, 3, 6, 8], "small")
    zero_range = cnt.copy()
    zero_range[3:6] = 0
    check(codec, O.AVG, O.INT32_T, np.int32, vals, zero_range, [0, 3, 6, 8], "one range without weight")
    check(codec, O.AVG, O.INT32_T, np.int32, vals, np.zeros(8, np.int32), [0, 3, 6, 8], "no weight at all")
    check(codec, O.AVG, O.DOUBLE_T, np.float64, vals.astype(np.float64), np.zeros(8, np.int32), [0, 3, 6, 8], "no weight, double")
    # an empty wave: no ranges with packets
    check(codec, O.SUM, O.INT32_T, np.int32, np.zeros(0, np.int32), None, [0, 0, 0], "empty")
    check(codec, O.SUM, O.FLOAT_T, np.float32, np.zeros(0, np.float32), None, [0, 0, 0], "empty, chained")


def test_float_chain_kinds_in_ranges(codec):
    """The float chain's hard cases (sums that hover, cancel, change binade at a range boundary) through
    every kernel the partial call can take: short ranges (plain chain), 8,192+ (one-workgroup scan),
    16,384+ (device-wide)."""
    rng = np.random.default_rng(7)
    for dt, t in ((np.float32, O.FLOAT_T), (np.float64, O.DOUBLE_T)):
        for n, g in ((3000, 3), (40000, 3), (200000, 4), (300000, 2)):
            kinds = {
                "growing": np.abs(rng.standard_normal(n)) + 0.5,
                "zero_mean": rng.standard_normal(n),
                "cancel": np.where(np.arange(n) % 2 == 0, 1.0, -1.0) * (1.0 + rng.integers(0, 3, n) * 2.0 ** -20),
                "wide": rng.standard_normal(n) * 10.0 ** rng.integers(-20, 20, n),
                "ties": rng.integers(-3, 4, n) * 0.5 + 2.0 ** 20,
            }
            for name, v in kinds.items():
                vals = v.astype(dt)
                for op in (O.SUM, O.AVG):
                    cnt = (rng.integers(1, 4, n)).astype(np.int32) if op == O.AVG else None
                    check(codec, op, t, dt, vals, cnt, even_edges(n, g), (name, n, g, op))


def test_node_layout(codec):
    """struct mrpdr_node as the header documents it: integer AVG carries the undivided, wrapping sum of
    value * count in T and the summed count; float MIN / MAX the NaN mark."""
    vals = np.array([30000, 30000, 30000], np.int16)
    cnt = np.array([2, 3, 1], np.int32)
    _, _, nodes = ranged(codec, O.AVG, O.INT16_T, vals, cnt, [0, 3])
    raw = nodes[0].cpu().numpy()
    prod = (vals.astype(np.int32) * cnt).astype(np.int16)          # (T)(value * count), truncating
    want = np.array(prod.astype(np.int32).sum()).astype(np.int16)  # added up in T, wrapping
    assert raw[:2].view(np.int16)[0] == want
    assert raw[8:12].view(np.int32)[0] == 6 and raw[12:16].view(np.int32)[0] == 1
    f = np.array([1.0, np.nan, 3.0], np.float32)
    _, _, nodes = ranged(codec, O.MIN, O.FLOAT_T, f, None, [0, 3])
    raw = nodes[0].cpu().numpy()
    assert raw[:4].view(np.float32)[0] == 3.0 and raw[8:12].view(np.int32)[0] == 1 and raw[12:16].view(np.int32)[0] == 1
    _, _, nodes = ranged(codec, O.MIN, O.FLOAT_T, f[:1], None, [0, 0])
    assert not nodes[0].cpu().numpy().any()                    # no packet: an absent node, all zero
