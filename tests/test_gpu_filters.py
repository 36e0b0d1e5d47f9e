"""tfilter_Sum / Min / Max / Avg on the GPU against the oracle, through the C ABI.

Bar: bit-exact for every type when the GPU and the oracle apply the same shape
(flat = one reference filter call over the wave; tree = the declared level-by-level
shape).  Against the FLAT reference result the tree shape is additionally held to
1e-6 relative for double; for float the tree differs from a 64K-term sequential sum by
rounding (about 1e-5 relative), which is why the shape is declared and bit-checked."""
import numpy as np
import pytest
import torch

from wavegen import O, random_values, splitmix64

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402

TYPES = [("c", O.CHAR_T, np.int8), ("uc", O.UCHAR_T, np.uint8), ("hd", O.INT16_T, np.int16),
         ("uhd", O.UINT16_T, np.uint16), ("d", O.INT32_T, np.int32), ("ud", O.UINT32_T, np.uint32),
         ("ld", O.INT64_T, np.int64), ("uld", O.UINT64_T, np.uint64), ("f", O.FLOAT_T, np.float32),
         ("lf", O.DOUBLE_T, np.float64)]
OPS = [("sum", O.SUM), ("min", O.MIN), ("max", O.MAX), ("avg", O.AVG)]
SIZES = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4097, 65536]


def values_for(dt, n, seed, op):
    dt = np.dtype(dt)
    if dt.kind == "f":
        # magnitudes that make a float sum order-sensitive but finite
        u = splitmix64(seed, n)
        mant = (u % np.uint64(2_000_001)).astype(np.float64) / 1e3 - 1000.0
        scale = np.array([1e-3, 1.0, 1e3, 1e6])[(u >> np.uint64(40)) % np.uint64(4)]
        return (mant * scale).astype(dt)
    v = random_values(dt, n, seed)
    if op == O.AVG and dt.kind == "i":
        v = np.where(v == np.iinfo(dt).min, dt.type(1), v)      # keeps INT_MIN / -1 out
    return v


def counts_for(n, seed):
    return (splitmix64(seed + 99, n) % np.uint64(5) + np.uint64(1)).astype(np.int32)


def run_gpu(codec, op, t, dt, vals, cnt, shape):
    dv = codec.to_device(vals)
    dc = codec.to_device(cnt) if cnt is not None else None
    res, ok = codec.filter_columns(op, t, dv, dc, shape)
    torch.cuda.synchronize()
    return W.result_value(res, ok, np.dtype(dt), op)


@pytest.mark.parametrize("tok,t,dt", TYPES)
@pytest.mark.parametrize("opname,op", OPS)
def test_filter_parity_flat_and_tree(codec, tok, t, dt, opname, op):
    for n in SIZES:
        vals = values_for(dt, n, 17 * n + t, op)
        cnt = counts_for(n, n) if op == O.AVG else None
        # flat: exactly one reference filter call over the wave
        if True:
            ref = O.filter_flat(op, t, vals, cnt)
            got = run_gpu(codec, op, t, dt, vals, cnt, W.SHAPE_FLAT)
            assert got is not None and ref is not None
            assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), (tok, opname, n, "flat")
            if op == O.AVG:
                assert got[1] == ref[1]
        # tree: the declared shape, bit for bit
        ref = O.filter_tree(op, t, vals, cnt, W.TREE_FANINS)
        got = run_gpu(codec, op, t, dt, vals, cnt, W.SHAPE_TREE)
        assert got is not None and ref is not None
        assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), (tok, opname, n, "tree")
        if op == O.AVG:
            assert got[1] == ref[1]
        # integer SUM and NaN-free MIN/MAX do not depend on the shape at all
        if op in (O.MIN, O.MAX) or (op == O.SUM and np.dtype(dt).kind != "f"):
            flat = O.filter_flat(op, t, vals, cnt)
            assert np.asarray(got[0]).tobytes() == np.asarray(flat[0]).tobytes()
        # double: the tree is within 1e-6 relative of the flat reference fold
        if np.dtype(dt) == np.float64 and op in (O.SUM, O.AVG):
            flat = O.filter_flat(op, t, vals, cnt)
            scale = float(np.sum(np.abs(vals.astype(np.float64)) * (cnt if cnt is not None else 1)))
            scale = scale / (float(np.sum(cnt)) if cnt is not None else 1.0)
            assert abs(float(got[0]) - float(flat[0])) <= 1e-6 * max(scale, 1e-300)


def test_native_filters_expectations(codec, golden):
    """tests/test_NativeFilters_FE.C: TFILTER_SUM over num_backends copies of VAL."""
    g = golden["native_filters_test"]
    for tok, t, dt in TYPES:
        for nb in g["num_backends"]:
            vals = np.full(nb, g["values"][tok], dtype=dt)
            comp = np.zeros(1, dtype=dt)
            with np.errstate(over="ignore"):
                for _ in range(nb):
                    comp[0] = comp[0] + np.dtype(dt).type(g["values"][tok])
            got = run_gpu(codec, O.SUM, t, dt, vals, None, W.SHAPE_FLAT)
            assert np.asarray(got[0]).tobytes() == comp[0].tobytes(), (tok, nb)


def test_filter_known_answers(codec, golden):
    v = np.array(golden["filters"][0]["values"], np.float32)
    for case, op in zip(golden["filters"][:3], (O.SUM, O.MIN, O.MAX)):
        got = run_gpu(codec, op, O.FLOAT_T, np.float32, v, None, W.SHAPE_FLAT)
        assert "%f" % got[0] == case["printf_f"]
    a = golden["filters"][3]
    got = run_gpu(codec, O.AVG, O.INT32_T, np.int32, np.array(a["values"], np.int32),
                  np.array(a["counts"], np.int32), W.SHAPE_FLAT)
    assert (int(got[0]), got[1]) == (a["value"], a["count"])


def test_nan_and_signed_zero_rules(codec):
    nan = np.float32("nan")
    cases = [np.array([1, nan, 2], np.float32), np.array([1, 2, nan], np.float32),
             np.array([nan, 5, 7], np.float32), np.array([0.0, -0.0], np.float32),
             np.array([-0.0, 0.0], np.float32), np.array([-0.0, -0.0], np.float32)]
    for v in cases:
        for op in (O.MIN, O.MAX, O.SUM):
            if op == O.SUM and np.isnan(v).any():
                continue
            ref = O.filter_flat(op, O.FLOAT_T, v)
            got = run_gpu(codec, op, O.FLOAT_T, np.float32, v, None, W.SHAPE_FLAT)
            assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), (v, op)
            ref = O.filter_tree(op, O.FLOAT_T, v, None, W.TREE_FANINS)
            got = run_gpu(codec, op, O.FLOAT_T, np.float32, v, None, W.SHAPE_TREE)
            assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), (v, op)


@pytest.mark.parametrize("dt,t", [(np.float32, O.FLOAT_T), (np.float64, O.DOUBLE_T)])
def test_flat_min_max_nan_restarts_across_levels(codec, dt, t):
    """One reference call keeps the new element whenever the comparison is false, so a NaN
    replaces the result and is replaced in turn (src/FilterDefinitions.C:2042-2136).  The flat
    MIN / MAX of float types is computed level by level with that rule carried along: NaNs (with
    their payload bits) and signed zeros at lane, wavefront, workgroup and launch boundaries."""
    bits = np.dtype(dt).itemsize * 8
    udt = np.uint32 if bits == 32 else np.uint64
    nan_a = np.array([0x7FC00001 if bits == 32 else 0x7FF8000000000001], udt).view(dt)[0]
    nan_b = np.array([0xFFC12345 if bits == 32 else 0xFFF8000000012345], udt).view(dt)[0]
    for n in (5, 64, 257, 1024, 1025, 4096, 4100, 70000):
        base = values_for(dt, n, 31 * n + 7, O.MIN)
        spots = sorted({0, 1, 3, 4, 63, 64, 255, 256, 1023, 1024, 1025, 2047, 4095, 4096, n // 2, n - 2, n - 1})
        spots = [p for p in spots if 0 <= p < n]
        variants = [[p] for p in spots] + [[spots[0], spots[-1]], spots, [n - 1], []]
        for k, nan_at in enumerate(variants):
            v = base.copy()
            for q, p in enumerate(nan_at):
                v[p] = nan_a if q % 2 == 0 else nan_b
            if k % 3 == 0:          # ties between +0 and -0 after the last NaN: the later one wins
                tail = (nan_at[-1] + 1) if nan_at else 0
                if tail + 2 <= n:
                    v[tail:] = np.where(np.arange(n - tail) % 2 == 0, dt(0.0), dt(-0.0))
            for op in (O.MIN, O.MAX):
                ref = O.filter_flat(op, t, v)
                got = run_gpu(codec, op, t, dt, v, None, W.SHAPE_FLAT)
                assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), (n, nan_at[:4], op)


def test_empty_wave_and_zero_counts(codec):
    res, ok = codec.filter_raw(O.SUM, O.INT32_T, None, 4, None, 4, 0)
    assert int(ok.item()) == 0
    got = run_gpu(codec, O.AVG, O.INT32_T, np.int32, np.array([5, 6], np.int32), np.array([0, 0], np.int32),
                  W.SHAPE_TREE)
    assert got is None


@pytest.mark.parametrize("shape", [W.SHAPE_FLAT, W.SHAPE_TREE])
def test_fused_decode_filter_on_the_wire(codec, shape):
    """65,536 '%d' packets decoded and summed (the IntegerAddition wave); the '%lf %d' Avg shape."""
    n = 65536
    vals = random_values(np.int32, n, 2)
    cols = [{"type": O.INT32_T, "values": vals}]
    arena, offs, lens = O.wave_encode("%d", cols, n)
    fmt = W.Format("%d")
    for bo in (W.LITTLEENDIAN, W.BIGENDIAN):
        src = arena if bo == 0 else arena.reshape(n, 4)[:, ::-1].copy().reshape(-1)
        wire = W.WireWave(codec.to_device(src), n, None, None, 4)
        for op in (O.SUM, O.MIN, O.MAX):
            res, ok = codec.filter_wire(op, fmt, wire, bo, shape)
            got = W.result_value(res, ok, np.dtype(np.int32), op)
            ref = O.filter_flat(op, O.INT32_T, vals)
            assert int(got[0]) == int(ref[0])
    n = 20000
    v = values_for(np.float64, n, 5, O.AVG)
    c = counts_for(n, 5)
    cols = [{"type": O.DOUBLE_T, "values": v}, {"type": O.INT32_T, "values": c}]
    arena, offs, lens = O.wave_encode("%lf %d", cols, n)
    fmt = W.Format("%lf %d")
    be = arena.reshape(n, 12).copy()
    be[:, :8] = arena.reshape(n, 12)[:, 7::-1]
    be[:, 8:] = arena.reshape(n, 12)[:, :7:-1]
    for bo, src in ((0, arena), (1, be.reshape(-1))):
        wire = W.WireWave(codec.to_device(src), n, None, None, 12)
        res, ok = codec.filter_wire(O.AVG, fmt, wire, bo, shape)
        got = W.result_value(res, ok, np.dtype(np.float64), O.AVG)
        ref = (O.filter_flat(O.AVG, O.DOUBLE_T, v, c) if shape == W.SHAPE_FLAT
               else O.filter_tree(O.AVG, O.DOUBLE_T, v, c, W.TREE_FANINS))
        assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes() and got[1] == ref[1]
        # the 16 result bytes are the data of the output packet '%lf %d'
        out_cols = [{"type": O.DOUBLE_T, "values": np.array([ref[0]])},
                    {"type": O.INT32_T, "values": np.array([ref[1]], np.int32)}]
        out_arena, _, _ = O.wave_encode("%lf %d", out_cols, 1)
        assert res.cpu().numpy()[:12].tobytes() == out_arena.tobytes()


@pytest.mark.parametrize("tok,t,dt", TYPES)
def test_decode_and_filter_in_one_call(codec, tok, t, dt):
    """mrpdr_wave_decode_filter = decode_pdr_data of every packet + the filter's fold of (*pkt)[0]
    (src/FilterDefinitions.C:148): columns and ok as the decode alone writes them, the result as the
    filter over the decoded column gives it -- against the oracle, every operator, both shapes, both
    sender byte orders, packed rows with and without padding, formats with further fields."""
    dt = np.dtype(dt)
    for fmt_s, n in (("%" + tok, 4), ("%" + tok, 1025), ("%" + tok + " %d", 5000), ("%" + tok + " %d %uc %lf", 2049),
                     ("%" + tok + " %d", 65536)):
        fmt = W.Format(fmt_s)
        vals = values_for(dt, n, 3 * n + t, O.AVG)
        host = [vals]
        if len(fmt) > 1:
            host.append(counts_for(n, n + 1))
        if len(fmt) > 2:
            host += [random_values(np.uint8, n, 7), random_values(np.float64, n, 8)]
        ocols = [{"type": int(ty), "values": v} for ty, v in zip(fmt.types, host)]
        arena, _, _ = O.wave_encode(fmt_s, ocols, n)
        fixed = fmt.fixed_bytes
        for bo, stride in ((W.LITTLEENDIAN, fixed), (W.BIGENDIAN, fixed), (W.BIGENDIAN, (fixed + 15) // 16 * 16),
                           (W.LITTLEENDIAN, 48)):            # 48 > 32: the two-step path inside the call
            rows = np.full((n, stride), 0xEE, np.uint8)
            src = arena.reshape(n, fixed)
            at = 0
            for w in fmt.widths:
                rows[:, at:at + w] = src[:, at:at + w][:, ::-1] if bo == W.BIGENDIAN else src[:, at:at + w]
                at += w
            wire = W.WireWave(codec.to_device(rows.reshape(-1)), n, None, None, stride)
            for op in (O.SUM, O.MIN, O.MAX, O.AVG):
                if op == O.AVG and len(fmt) < 2:
                    continue
                for shape in (W.SHAPE_FLAT, W.SHAPE_TREE):
                    cols = codec.alloc_columns(fmt, n)
                    ok = codec.empty(n, torch.int32)
                    res, rok = codec.decode_filter_into(fmt, wire, cols, ok, op, shape, bo)
                    torch.cuda.synchronize()
                    assert bool(ok.all())
                    for f, v in enumerate(host):
                        assert W.Codec.to_numpy(cols[f].values).view(v.dtype).tobytes() == v.tobytes(), (fmt_s, f)
                    cnt = host[1] if op == O.AVG else None
                    ref = (O.filter_flat(op, t, vals, cnt) if shape == W.SHAPE_FLAT
                           else O.filter_tree(op, t, vals, cnt, W.TREE_FANINS))
                    got = W.result_value(res, rok, dt, op)
                    assert got is not None and np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), \
                        (fmt_s, n, bo, stride, op, shape)
                    if op == O.AVG:
                        assert got[1] == ref[1]


def test_decode_and_filter_refusals_and_short_packets(codec):
    fmt = W.Format("%d %d")
    n = 3000
    v, c = random_values(np.int32, n, 1), counts_for(n, 2)
    rows = np.concatenate([v.view(np.uint8).reshape(n, 4), c.view(np.uint8).reshape(n, 4)], axis=1)
    wire = W.WireWave(codec.to_device(rows.reshape(-1)), n, None, None, 8)
    cols, ok = codec.alloc_columns(fmt, n), codec.empty(n, torch.int32)
    with pytest.raises(W.PdrError):
        codec.decode_filter_into(W.Format("%d %as"), wire, cols, ok, O.SUM)          # not scalars only
    with pytest.raises(W.PdrError):
        codec.decode_filter_into(W.Format("%d %f"), wire, cols, ok, O.AVG)           # count is not '%d'
    with pytest.raises(W.PdrError):
        codec.decode_filter_into(W.Format("%f"), wire, cols, ok, W.FILTER_OR)        # bitwise on float
    with pytest.raises(W.PdrError):
        codec.decode_filter_into(fmt, wire, cols, ok, 17)
    # packets shorter than the format: ok = 0, the missing fields read as zero, in the columns
    # and in the fold alike (as the decode followed by the filter over the columns)
    lens = np.full(n, 8, np.uint64)
    lens[5::7] = 4
    lens[11::13] = 3
    short = W.WireWave(wire.arena, n, None, codec.to_device(lens), 8)
    res, rok = codec.decode_filter_into(fmt, short, cols, ok, O.AVG, W.SHAPE_FLAT)
    torch.cuda.synchronize()
    dv = np.where(lens >= 4, v, 0).astype(np.int32)
    dc = np.where(lens >= 8, c, 0).astype(np.int32)
    assert np.array_equal(W.Codec.to_numpy(ok), (lens >= 8).astype(np.int32))
    assert np.array_equal(W.Codec.to_numpy(cols[0].values), dv) and np.array_equal(W.Codec.to_numpy(cols[1].values), dc)
    ref = O.filter_flat(O.AVG, O.INT32_T, dv, dc)
    got = W.result_value(res, rok, np.dtype(np.int32), O.AVG)
    assert (int(got[0]), got[1]) == (int(ref[0]), ref[1])
    # an empty wave: nothing decoded, no output packet
    res, rok = codec.decode_filter_into(fmt, W.WireWave(wire.arena, 0, None, None, 8), cols, ok, O.SUM)
    assert int(rok.item()) == 0


@pytest.mark.parametrize("opname,op", OPS)
def test_sharded_wave_combine(codec, opname, op):
    """A 1M-packet '%lf %d' wave in 8 packet ranges (one per GPU of a node), in the declared-tree
    reading (shape=SHAPE_TREE): per-range tree result on the GPU, then the ranks as the last tree
    level -- against the oracle's tree of the same shape; held to 1e-6 relative of ONE flat fold over
    the whole wave, which is what the default (SHAPE_FLAT, tests/test_gpu_ranges.py) returns exactly."""
    from mrnet_amd import dist as D
    n, world = 1 << 20, 8
    v = values_for(np.float64, n, 55, op)
    c = counts_for(n, 55)
    cols = [{"type": O.DOUBLE_T, "values": v}, {"type": O.INT32_T, "values": c}]
    arena, offs, lens = O.wave_encode("%lf %d", cols, n)
    fmt = W.Format("%lf %d")
    rows, oks, ref_v, ref_c = [], [], [], []
    for r in range(world):
        lo, hi = D.shard_range(n, r, world)
        wire = W.WireWave(codec.to_device(arena[lo * 12:hi * 12]), hi - lo, None, None, 12)
        res, ok = codec.filter_raw(op, O.DOUBLE_T, wire.arena.data_ptr(), 12, wire.arena.data_ptr() + 8, 12,
                                   hi - lo, 0, W.SHAPE_TREE)
        rows.append(res.clone()); oks.append(ok.clone())
        ref = O.filter_tree(op, O.DOUBLE_T, v[lo:hi], c[lo:hi] if op == O.AVG else None, W.TREE_FANINS)
        ref_v.append(ref[0]); ref_c.append(ref[1] if op == O.AVG else 1)
    res, ok = D.combine_results(codec, op, W.DataType.DOUBLE_T, 8, torch.stack(rows), torch.cat(oks))
    got = W.result_value(res, ok, np.dtype(np.float64), op)
    want = O.filter_flat(op, O.DOUBLE_T, np.array(ref_v), np.array(ref_c, np.int32) if op == O.AVG else None)
    assert np.asarray(got[0]).tobytes() == np.asarray(want[0]).tobytes()
    if op == O.AVG:
        assert got[1] == want[1] == int(c.sum())
    one_call = O.filter_flat(op, O.DOUBLE_T, v, c if op == O.AVG else None)
    scale = float(np.sum(np.abs(v) * (c if op == O.AVG else 1))) / (float(c.sum()) if op == O.AVG else 1.0)
    assert abs(float(got[0]) - float(one_call[0])) <= 1e-6 * max(scale, abs(float(one_call[0])))
    # single-process group: the sharded entry point degenerates to the local tree + one more level
    wire = W.WireWave(codec.to_device(arena), n, None, None, 12)
    if op == O.AVG:
        res1, ok1 = D.filter_wire_sharded(codec, op, fmt, wire, shape=W.SHAPE_TREE)
        whole = O.filter_tree(op, O.DOUBLE_T, v, c, W.TREE_FANINS)
        last = O.filter_flat(op, O.DOUBLE_T, np.array([whole[0]]), np.array([whole[1]], np.int32))
        assert np.asarray(W.result_value(res1, ok1, np.dtype(np.float64), op)[0]).tobytes() == np.asarray(last[0]).tobytes()
        # and in the default reading: the one call's bits
        res2, ok2 = D.filter_wire_sharded(codec, op, fmt, wire)
        assert np.asarray(W.result_value(res2, ok2, np.dtype(np.float64), op)[0]).tobytes() == np.asarray(one_call[0]).tobytes()


def test_bitwise_or_and_integer_percentiles(codec):
    """The reducers of Examples/FaultRecovery/FaultRecoveryFilter.C: BitsetOr over '%uld', and
    IntegerPercentiles over '%uld %ud %ud' (bits | max | min) with its persistent state.
    Unpinned by reference artefacts; the oracle restates the loops."""
    from mrnet_amd import packet as P
    n = 5000
    bits = splitmix64(71, n) & splitmix64(72, n) & splitmix64(73, n)
    wave = P.PacketWave(codec, 9, 100, "%uld", [bits], n)
    for shape in (W.SHAPE_FLAT, W.SHAPE_TREE):
        out = P.BitsetOr(wave, shape)
        want = O.filter_flat(O.OR, O.UINT64_T, bits, None)[0]
        assert int(W.Codec.to_numpy(out.wire.arena)[:8].view(np.uint64)[0]) == int(want) == int(np.bitwise_or.reduce(bits))
    for t, dt in ((O.INT16_T, np.int16), (O.UCHAR_T, np.uint8), (O.INT32_T, np.int32)):
        v = random_values(dt, 777, 5)
        res, ok = codec.filter_columns(W.FILTER_OR, t, codec.to_device(v), None, W.SHAPE_TREE)
        assert W.Codec.to_numpy(res)[: np.dtype(dt).itemsize].tobytes() == np.asarray(O.filter_flat(O.OR, t, v, None)[0]).tobytes()
        res, ok = codec.filter_columns(W.FILTER_AND, t, codec.to_device(v | dt(0x11)), None, W.SHAPE_TREE)
        assert W.Codec.to_numpy(res)[: np.dtype(dt).itemsize].tobytes() == \
            np.asarray(O.filter_flat(O.AND, t, v | dt(0x11), None)[0]).tobytes()
    with pytest.raises(W.PdrError):
        codec.filter_columns(W.FILTER_OR, O.DOUBLE_T, codec.to_device(np.zeros(4)), None)
    # IntegerPercentiles: two waves through the same state
    state = None
    smin, smax, sbits = P.FR_RANGE_MAX, 0, 0
    for seed in (1, 2):
        m = 3000
        b = (splitmix64(seed, m) >> np.uint64(40)).astype(np.uint64)
        mx = (splitmix64(seed + 10, m) % np.uint64(1000)).astype(np.uint32)
        mn = (mx // np.uint32(2)).astype(np.uint32)
        w = P.PacketWave(codec, 9, P.PROT_WAVE, "%uld %ud %ud", [b, mx, mn], m)
        out, state = P.IntegerPercentiles(w, state)
        sbits |= int(np.bitwise_or.reduce(b)) & ((1 << P.FR_BINS) - 1)
        smax, smin = max(smax, int(mx.max())), min(smin, int(mn.min()))
        raw = W.Codec.to_numpy(out.wire.arena)
        assert (int(raw[:8].view(np.uint64)[0]), int(raw[8:12].view(np.uint32)[0]), int(raw[12:16].view(np.uint32)[0])) \
            == (sbits, smax, smin)
        assert out.get_Tag() == P.PROT_WAVE


@pytest.mark.parametrize("tok,t,dt", TYPES)
def test_fused_min_max_avg_equals_the_three_filters(codec, tok, t, dt):
    """mrpdr_wave_filter_stats reads the wave once; each of its results is the bytes of the
    separate tree-shaped filter (which the other tests hold against the oracle)."""
    for n in (0, 1, 5, 1023, 1025, 70000):
        vals = values_for(dt, n, 3 * n + t, W.TFILTER_AVG) if n else np.zeros(0, dt)
        cnt = counts_for(n, n) if n else np.zeros(0, np.int32)
        w = np.dtype(dt).itemsize
        rows = np.zeros((n, w + 4), np.uint8)
        if n:
            rows[:, :w] = vals.view(np.uint8).reshape(n, w)
            rows[:, w:] = cnt.view(np.uint8).reshape(n, 4)
        d = codec.to_device(rows.reshape(-1)) if n else codec.empty(16, torch.uint8)
        base = d.data_ptr()
        res, oks = codec.filter_stats_raw(t, base, w + 4, base + w, w + 4, n)
        for k, op in enumerate((W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG)):
            one, ok1 = codec.filter_raw(op, t, base, w + 4, base + w, w + 4, n, 0, W.SHAPE_TREE)
            assert int(oks[k].item()) == int(ok1.item()), (tok, n, op)
            assert torch.equal(res[16 * k:16 * k + 16], one), (tok, n, op)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 1025, 65536])
def test_result_written_to_pinned_host_memory(codec, n):
    """`result` / `result_ok` may be pinned host memory: the last launch writes the output packet's
    bytes there itself, and a node that forwards one wave at a time polls result_ok instead of
    synchronising the stream.  Same bytes as with a device buffer, and as the oracle's fold."""
    vals = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=codec.device)
    res_h, ok_h = torch.zeros(16, dtype=torch.uint8).pin_memory(), torch.zeros(1, dtype=torch.int32).pin_memory()
    flag = ok_h.numpy()
    for op in (W.TFILTER_SUM, W.TFILTER_MIN, W.TFILTER_MAX):
        res_d, ok_d = codec.filter_raw(op, W.DataType.INT32_T, vals.data_ptr(), 4, None, 0, n)
        flag[0] = RESULT_PENDING
        codec.filter_raw(op, W.DataType.INT32_T, vals.data_ptr(), 4, None, 0, n, result=res_h, result_ok=ok_h)
        assert _poll(flag) == 1, "the result never arrived in host memory"
        # the flag is stored last, behind a system-scope fence: whoever sees it sees the result bytes
        seen = bytes(res_h.numpy())
        torch.cuda.synchronize()
        assert seen == bytes(res_h.numpy()) == bytes(res_d.cpu().numpy())
        want = O.filter_flat({W.TFILTER_SUM: O.SUM, W.TFILTER_MIN: O.MIN, W.TFILTER_MAX: O.MAX}[op], O.INT32_T,
                             vals.cpu().numpy())
        assert int(res_h.numpy()[:4].view(np.int32)[0]) == int(want[0])


RESULT_PENDING = -1        # MRPDR_RESULT_PENDING of include/mrpdr.h


def _poll(flag, limit=50_000_000):
    spins = 0
    while flag[0] == RESULT_PENDING and spins < limit:
        spins += 1
    return int(flag[0])


def test_polling_ends_when_there_is_no_result(codec):
    """An absent result (an empty wave; an AVG whose counts sum to zero) writes result_ok = 0: a final
    value, distinct from the pending mark a polling host presets, so its loop ends -- for every kernel
    that can write a result (tree, one tile, the float chain in its three forms, decode + filter)."""
    res_h, ok_h = torch.zeros(16, dtype=torch.uint8).pin_memory(), torch.zeros(1, dtype=torch.int32).pin_memory()
    flag = ok_h.numpy()
    cases = []
    for n in (0, 7, 5000, 20000):
        for t, dt in ((W.DataType.INT32_T, torch.int32), (W.DataType.FLOAT_T, torch.float32), (W.DataType.DOUBLE_T, torch.float64)):
            cases.append((n, t, dt, W.TFILTER_AVG))
    cases += [(0, W.DataType.INT32_T, torch.int32, W.TFILTER_SUM), (0, W.DataType.FLOAT_T, torch.float32, W.TFILTER_SUM),
              (0, W.DataType.DOUBLE_T, torch.float64, W.TFILTER_MIN)]
    for n, t, dt, op in cases:
        vals = torch.ones(max(n, 1), dtype=dt, device=codec.device)
        cnts = torch.zeros(max(n, 1), dtype=torch.int32, device=codec.device)         # no weight at all
        for shape in (W.SHAPE_FLAT, W.SHAPE_TREE, W.SHAPE_FLAT_SERIAL):
            flag[0] = RESULT_PENDING
            res_h.fill_(0xEE)
            codec.filter_raw(op, t, vals.data_ptr(), vals.element_size(), cnts.data_ptr(), 4, n, 0, shape,
                             result=res_h, result_ok=ok_h)
            assert _poll(flag) == 0, (n, t, op, shape)
            assert not res_h.numpy().any(), "an absent result is sixteen zero bytes"
    # decode + filter of an empty wave and of a weightless one
    fmt = W.Format("%d %d")
    for n in (0, 3000):
        rows = torch.zeros(max(n, 1) * 8, dtype=torch.uint8, device=codec.device)
        wire = W.WireWave(rows, n, None, None, 8)
        cols, ok = codec.alloc_columns(fmt, max(n, 1)), codec.empty(max(n, 1), torch.int32)
        flag[0] = RESULT_PENDING
        codec.decode_filter_into(fmt, wire, cols, ok, W.TFILTER_AVG, W.SHAPE_FLAT, result=res_h, result_ok=ok_h)
        assert _poll(flag) == 0, n
    torch.cuda.synchronize()


import os  # noqa: E402

SOAK = max(1, int(os.environ.get("MRPDR_SOAK", "1")))


@pytest.mark.parametrize("seed", range(80 * SOAK))
def test_random_filters(codec, seed):
    """Drawn wave size, element type, operator, shape, field offset / packet stride and sender byte
    order: the fold straight on the wire bytes against the oracle, bit for bit.  MIN / MAX see raw
    bit patterns (NaNs and infinities included: selection copies bits), SUM / AVG finite values."""
    r = splitmix64(31000 + seed, 16)
    n = [1, 2, 3, 4, 5, 255, 1023, 1025, 4097, 20000, 70001][int(r[0] % np.uint64(11))]
    if int(r[1] % np.uint64(4)) == 0:
        n = int(r[2] % np.uint64(70000)) + 1
    tok, t, dt = TYPES[int(r[3] % np.uint64(len(TYPES)))]
    dt = np.dtype(dt)
    ops = [O.SUM, O.MIN, O.MAX, O.AVG] + ([O.OR, O.AND] if dt.kind != "f" else [])
    op = ops[int(r[4] % np.uint64(len(ops)))]
    shape = W.SHAPE_FLAT if (n <= 5000 and int(r[5] % np.uint64(2))) else W.SHAPE_TREE
    swap = int(r[6] % np.uint64(2))
    pad_front, pad_back = int(r[7] % np.uint64(9)), int(r[8] % np.uint64(7))
    if op in (O.MIN, O.MAX) and dt.kind == "f":
        vals = random_values(dt, n, 500 + seed, finite=False)
    else:
        vals = values_for(dt, n, 500 + seed, op)
    cnt = None
    if op == O.AVG:
        cnt = (splitmix64(seed + 7, n) % np.uint64(6)).astype(np.int32)          # zeros allowed
    w = dt.itemsize
    stride = pad_front + w + (4 if op == O.AVG else 0) + pad_back
    rows = (splitmix64(seed + 9, n * stride) & np.uint64(0xFF)).astype(np.uint8).reshape(n, stride)
    vb = vals.view(np.uint8).reshape(n, w)
    rows[:, pad_front:pad_front + w] = vb[:, ::-1] if swap else vb
    if op == O.AVG:
        cb = cnt.view(np.uint8).reshape(n, 4)
        rows[:, pad_front + w:pad_front + w + 4] = cb[:, ::-1] if swap else cb
    dev = codec.to_device(rows.reshape(-1))
    base = dev.data_ptr() + pad_front
    res, ok = codec.filter_raw(op, t, base, stride, (base + w) if op == O.AVG else None, stride, n, swap, shape)
    torch.cuda.synchronize()
    got = W.result_value(res, ok, dt, op if op in (O.SUM, O.MIN, O.MAX, O.AVG) else O.SUM)
    ref = O.filter_flat(op, t, vals, cnt) if shape == W.SHAPE_FLAT else O.filter_tree(op, t, vals, cnt, W.TREE_FANINS)
    assert (got is None) == (ref is None), (tok, op, n, shape)
    if ref is not None:
        assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), (tok, op, n, shape, swap)
        if op == O.AVG:
            assert got[1] == ref[1]


@pytest.mark.parametrize("seed", range(30 * SOAK))
def test_random_filters_mixed_byte_orders(codec, seed):
    """The fold on the wire bytes of a wave whose packets come from senders of both byte orders
    (every packet read with its own, src/Packet.C:118)."""
    r = splitmix64(33000 + seed, 8)
    n = int(r[0] % np.uint64([9, 3000, 70000][int(r[1] % np.uint64(3))])) + 1
    tok, t, dt = TYPES[int(r[2] % np.uint64(len(TYPES)))]
    dt = np.dtype(dt)
    op = [O.SUM, O.MIN, O.MAX, O.AVG][int(r[3] % np.uint64(4))]
    vals = values_for(dt, n, 700 + seed, op)
    cnt = counts_for(n, seed) if op == O.AVG else None
    orders = (splitmix64(seed + 5, n) % np.uint64(2)).astype(np.uint8)
    w = dt.itemsize
    stride = w + (4 if op == O.AVG else 0)
    rows = np.zeros((n, stride), np.uint8)
    vb = vals.view(np.uint8).reshape(n, w)
    rows[:, :w] = np.where(orders[:, None] != 0, vb[:, ::-1], vb)
    if op == O.AVG:
        cb = cnt.view(np.uint8).reshape(n, 4)
        rows[:, w:] = np.where(orders[:, None] != 0, cb[:, ::-1], cb)
    dev = codec.to_device(rows.reshape(-1))
    res, ok = codec.filter_raw(op, t, dev.data_ptr(), stride, (dev.data_ptr() + w) if op == O.AVG else None, stride, n,
                               codec.to_device(orders), W.SHAPE_TREE)
    torch.cuda.synchronize()
    got = W.result_value(res, ok, dt, op)
    ref = O.filter_tree(op, t, vals, cnt, W.TREE_FANINS)
    assert np.asarray(got[0]).tobytes() == np.asarray(ref[0]).tobytes(), (tok, op, n)
    if op == O.AVG:
        assert got[1] == ref[1]
