"""The entry points that take preallocated buffers enqueue kernels and nothing else: they can be
captured into a HIP graph and replayed on new data, and two contexts can work on two streams at once."""
import numpy as np
import pytest
import torch

from wavegen import O, make_wave, random_values
from test_gpu_codec import dev_cols, host_cols, wire_from_numpy
from wavegen import columns_equal

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402


def test_decode_filter_encode_replayed_from_a_graph(codec):
    n = 20000
    fmt = W.Format("%d")
    dev = codec.device
    src = torch.zeros(4 * n, dtype=torch.uint8, device=dev)
    wire = W.WireWave(src, n, None, None, 4)
    out_wire = W.WireWave(torch.zeros(4 * n, dtype=torch.uint8, device=dev), n, None, None, 4)
    cols = codec.alloc_columns(fmt, n)
    ok = torch.zeros(n, dtype=torch.int32, device=dev)
    res = torch.zeros(16, dtype=torch.uint8, device=dev)
    rok = torch.zeros(1, dtype=torch.int32, device=dev)

    def step():
        codec.filter_raw(W.TFILTER_SUM, W.DataType.INT32_T, src.data_ptr(), 4, None, 4, n, 1, W.SHAPE_TREE, res, rok)
        codec.decode_into(fmt, wire, cols, ok, W.BIGENDIAN)
        codec.encode_into(fmt, cols, out_wire)

    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        step()                                        # warm-up outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
    torch.cuda.synchronize()
    for seed in (1, 2, 3):
        vals = random_values(np.int32, n, seed)
        src.copy_(torch.from_numpy(vals.view(np.uint8).reshape(n, 4)[:, ::-1].copy().reshape(-1)).to(dev))
        g.replay()
        torch.cuda.synchronize()
        assert int(res[:4].view(torch.int32).item()) == int(vals.sum(dtype=np.int32)) and int(rok.item()) == 1
        assert np.array_equal(W.Codec.to_numpy(cols[0].values).view(np.int32), vals) and bool(ok.all())
        assert np.array_equal(W.Codec.to_numpy(out_wire.arena).view(np.int32), vals)
    assert codec.status() == 0


def test_two_contexts_on_two_streams(codec):
    """One ctx per stream (the contract): interleaved calls of two contexts do not disturb each other."""
    other = W.Codec(device=codec.device.index, max_packets=1 << 16)
    fmt_s = "%d %s %ad"
    fmt = W.Format(fmt_s)
    streams = [torch.cuda.Stream(device=codec.device) for _ in range(2)]
    jobs = []
    for k, (cd, st) in enumerate(zip((codec, other), streams)):
        n = 3000 + 500 * k
        ocols = make_wave(fmt_s, n, 60 + k, max_count=50)
        arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
        jobs.append((cd, st, n, ocols, arena, offs, lens))
    results = [None, None]
    for rep in range(3):
        for k, (cd, st, n, ocols, arena, offs, lens) in enumerate(jobs):
            with torch.cuda.stream(st):
                wire = cd.encode(fmt, dev_cols(cd, fmt, ocols), n)
                dcols, ok = cd.decode(fmt, wire_from_numpy(cd, n, arena, offs, lens), W.LITTLEENDIAN)
                results[k] = (wire, dcols, ok)
    torch.cuda.synchronize()
    for k, (cd, st, n, ocols, arena, offs, lens) in enumerate(jobs):
        wire, dcols, ok = results[k]
        assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
        assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)
        assert cd.status() == 0


def test_bound_calls_keep_their_buffers_alive(codec):
    """A call bound once (Codec.bind_*) freezes raw device pointers; it must hold on to the tensors
    they point into, so that a caller who drops its own references cannot free them under it."""
    import gc
    n = 4096
    fmt = W.Format("%d")
    vals = random_values(np.int32, n, 77)
    wire = W.WireWave(codec.to_device(vals.view(np.uint8)), n, None, None, 4)
    cols = codec.alloc_columns(fmt, n)
    ok = codec.empty(n, torch.int32)
    res, rok = codec.empty(16, torch.uint8), codec.empty(1, torch.int32)
    call = codec.bind_decode_filter(fmt, wire, cols, ok, W.TFILTER_SUM, W.SHAPE_FLAT, res, rok)
    col_ptr, res_ptr = cols[0].values.data_ptr(), res.data_ptr()
    owned = {t.data_ptr() for t in call.owners if isinstance(t, torch.Tensor)}
    assert {wire.arena.data_ptr(), col_ptr, ok.data_ptr(), res_ptr, rok.data_ptr()} <= owned
    del wire, cols, ok, rok
    gc.collect()
    torch.cuda.empty_cache()
    junk = [torch.full((n,), 0x55, dtype=torch.int32, device=codec.device) for _ in range(8)]    # would reuse freed blocks
    call()
    torch.cuda.synchronize()
    assert int(res[:4].view(torch.int32).item()) == int(vals.sum(dtype=np.int32))
    assert all(int(j[0].item()) == 0x55 for j in junk)
