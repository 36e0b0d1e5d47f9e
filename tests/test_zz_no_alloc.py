"""mrpdr_wave_filter_noalloc -- the entry point that allocates nothing -- against
mrpdr_wave_filter, through the C ABI: same 16 result bytes and the same ok flag, on two calls
in a row, the second with a smaller input (the built-in device array keeps the nodes of the
first call: nothing of them may leak into the second)."""
import numpy as np
import pytest
import torch

from wavegen import O, random_values, splitmix64

pytestmark = pytest.mark.gpu

from mrnet_amd import capi  # noqa: E402
from mrnet_amd import wave as W  # noqa: E402

CASES = [  # (operator, DataType, numpy dtype, shape)
    (W.TFILTER_SUM, W.DataType.INT32_T, np.int32, W.SHAPE_FLAT),
    (W.TFILTER_SUM, W.DataType.FLOAT_T, np.float32, W.SHAPE_TREE),
    (W.TFILTER_SUM, W.DataType.FLOAT_T, np.float32, W.SHAPE_FLAT),
    (W.TFILTER_MIN, W.DataType.DOUBLE_T, np.float64, W.SHAPE_FLAT),
    (W.TFILTER_MAX, W.DataType.INT16_T, np.int16, W.SHAPE_TREE),
    (W.TFILTER_AVG, W.DataType.INT64_T, np.int64, W.SHAPE_FLAT),
    (W.TFILTER_AVG, W.DataType.DOUBLE_T, np.float64, W.SHAPE_TREE),
    (W.TFILTER_AVG, W.DataType.DOUBLE_T, np.float64, W.SHAPE_FLAT),      # (200,003 packets: chunk summaries in the built-in array; 300,001 doubles are beyond its room: the one-workgroup scan)
    (W.FILTER_OR, W.DataType.UINT64_T, np.uint64, W.SHAPE_TREE),
]


def both(codec, op, t, dt, shape, n, seed, swap):
    lib = capi.load()
    dt = np.dtype(dt)
    if dt.kind == "f":
        vals = ((splitmix64(seed, n) % np.uint64(2_000_001)).astype(np.float64) / 1e3 - 1000.0).astype(dt)
    else:
        vals = random_values(dt, n, seed)
        if dt.kind == "i":
            vals = np.where(vals == np.iinfo(dt).min, dt.type(1), vals)
    cnts = (splitmix64(seed + 1, n) % np.uint64(5) + np.uint64(1)).astype(np.int32)
    if swap:
        vals, cnts = vals.byteswap(), cnts.byteswap()
    dv, dc = codec.to_device(vals), codec.to_device(cnts)
    out = []
    st = torch.cuda.current_stream().cuda_stream
    for noalloc in (False, True):
        res = torch.full((16,), 0x5A, dtype=torch.uint8, device=codec.device)
        ok = torch.full((1,), -7, dtype=torch.int32, device=codec.device)
        args = (op, int(t), shape, dv.data_ptr(), dt.itemsize, dc.data_ptr() if op == W.TFILTER_AVG else None, 4,
                n, int(swap), res.data_ptr(), ok.data_ptr(), st)
        r = lib.mrpdr_wave_filter_noalloc(*args) if noalloc else lib.mrpdr_wave_filter(codec.handle, *args)
        assert r == 1, capi.last_error()
        torch.cuda.synchronize()
        out.append((res.cpu().numpy().tobytes(), int(ok.item())))
    return out, vals, cnts


@pytest.mark.parametrize("op,t,dt,shape", CASES)
def test_noalloc_filter_equals_ctx_filter_twice_in_a_row(codec, op, t, dt, shape):
    # first a wave of many workgroup nodes, then smaller ones, then a tiny one
    for n, seed, swap in ((300_001, 11, 0), (200_003, 14, 1), (5_000, 12, 1), (3, 13, 0)):
        (a, b), vals, cnts = both(codec, op, t, dt, shape, n, seed, swap)
        assert a == b, (op, int(t), shape, n)
        assert a[1] == 1
    # and the oracle agrees on the last one (the existing entry point is checked against it elsewhere)
    if op in (W.TFILTER_SUM, W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG):
        ref = O.filter_flat(op, int(t), vals, cnts if op == W.TFILTER_AVG else None)
        assert a[0][:np.dtype(dt).itemsize] == np.asarray(ref[0]).tobytes()


def test_noalloc_filter_limits(codec):
    lib = capi.load()
    res = torch.zeros(16, dtype=torch.uint8, device=codec.device)
    ok = torch.zeros(1, dtype=torch.int32, device=codec.device)
    v = torch.zeros(8, dtype=torch.int32, device=codec.device)
    st = torch.cuda.current_stream().cuda_stream
    # beyond the nodes the built-in array holds: refused before anything is launched
    assert lib.mrpdr_wave_filter_noalloc(W.TFILTER_SUM, int(W.DataType.INT32_T), W.SHAPE_TREE, v.data_ptr(), 0, None, 0,
                                         (1 << 24) + 1, 0, res.data_ptr(), ok.data_ptr(), st) == 0
    assert "2^24" in capi.last_error() or "16777216" in capi.last_error()
    # an empty wave: no packet, result_ok 0
    assert lib.mrpdr_wave_filter_noalloc(W.TFILTER_SUM, int(W.DataType.INT32_T), W.SHAPE_TREE, None, 4, None, 0,
                                         0, 0, res.data_ptr(), ok.data_ptr(), st) == 1
    torch.cuda.synchronize()
    assert int(ok.item()) == 0
    # the largest admissible n (stride 0: every packet reads the same value)
    one = torch.ones(1, dtype=torch.int32, device=codec.device)
    assert lib.mrpdr_wave_filter_noalloc(W.TFILTER_SUM, int(W.DataType.INT32_T), W.SHAPE_FLAT, one.data_ptr(), 0, None, 0,
                                         1 << 24, 0, res.data_ptr(), ok.data_ptr(), st) == 1
    torch.cuda.synchronize()
    assert int(ok.item()) == 1 and int(res[:4].cpu().numpy().view(np.int32)[0]) == 1 << 24


@pytest.mark.parametrize("op,t,dt,shape,n", [
    (W.TFILTER_SUM, W.DataType.INT32_T, np.int32, W.SHAPE_FLAT, 70_000),
    (W.TFILTER_AVG, W.DataType.INT64_T, np.int64, W.SHAPE_FLAT, 70_000),
    (W.TFILTER_SUM, W.DataType.FLOAT_T, np.float32, W.SHAPE_FLAT, 40_000),        # the device-wide float chain: three launches
    (W.TFILTER_SUM, W.DataType.FLOAT_T, np.float32, W.SHAPE_FLAT, 9_000),         # the one-workgroup scan
    (W.TFILTER_MIN, W.DataType.DOUBLE_T, np.float64, W.SHAPE_TREE, 5_000),
])
def test_noalloc_filter_under_graph_capture(codec, op, t, dt, shape, n):
    """The property in the name: after one warm call (which resolves the built-in array's address) the
    entry point is captured into a HIP graph -- a hipMalloc, hipFree or synchronisation inside the
    captured region would fail the capture -- and the graph is replayed on new data; every replay
    equals mrpdr_wave_filter (with a ctx) on the same data."""
    lib = capi.load()
    dt = np.dtype(dt)
    dev = codec.device
    vals = torch.zeros(n, dtype={np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64,
                                 np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[dt], device=dev)
    cnts = torch.ones(n, dtype=torch.int32, device=dev)
    res = torch.zeros(16, dtype=torch.uint8, device=dev)
    ok = torch.zeros(1, dtype=torch.int32, device=dev)

    def call(stream):
        r = lib.mrpdr_wave_filter_noalloc(op, int(t), shape, vals.data_ptr(), dt.itemsize,
                                          cnts.data_ptr() if op == W.TFILTER_AVG else None, 4, n, 0,
                                          res.data_ptr(), ok.data_ptr(), stream)
        assert r == 1, capi.last_error()

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        call(side.cuda_stream)                         # the warm call, outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for seed in (21, 22, 23):
        if dt.kind == "f":
            host = ((splitmix64(seed, n) % np.uint64(2_000_001)).astype(np.float64) / 1e3 - 900.0).astype(dt)
        else:
            host = (splitmix64(seed, n) % np.uint64(100000)).astype(dt)
        hc = (splitmix64(seed + 5, n) % np.uint64(5) + np.uint64(1)).astype(np.int32)
        vals.copy_(torch.from_numpy(host).to(dev))
        cnts.copy_(torch.from_numpy(hc).to(dev))
        res.fill_(0x5A)
        ok.fill_(-7)
        g.replay()
        torch.cuda.synchronize()
        want, wok = codec.filter_raw(op, t, vals.data_ptr(), dt.itemsize, cnts.data_ptr() if op == W.TFILTER_AVG else None,
                                     4, n, 0, shape)
        torch.cuda.synchronize()
        assert int(ok.item()) == int(wok.item()) == 1
        assert res.cpu().numpy().tobytes() == want.cpu().numpy().tobytes(), (op, int(t), n, seed)
        if shape == W.SHAPE_FLAT:
            ref = O.filter_flat(op, int(t), host, hc if op == W.TFILTER_AVG else None)
            assert res.cpu().numpy()[:dt.itemsize].tobytes() == np.asarray(ref[0]).tobytes()
