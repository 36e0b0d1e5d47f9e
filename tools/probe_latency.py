"""Diagnostic: what a node that forwards one wave at a time waits for -- the filter result read through
a stream synchronisation, or polled from pinned host memory the kernel wrote it to."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
codec = W.Codec()
dev = codec.device
fmt = W.Format("%d")
for n in (4, 1024, 65536):
    ws = W.WireWave(torch.ones(n, dtype=torch.int32, device=dev).view(torch.uint8), n, None, None, 4)
    cs, oks = codec.alloc_columns(fmt, n), torch.empty(n, dtype=torch.int32, device=dev)
    res_d, rok_d = torch.empty(16, dtype=torch.uint8, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
    res_h, rok_h = torch.zeros(16, dtype=torch.uint8).pin_memory(), torch.zeros(1, dtype=torch.int32).pin_memory()
    call_d = codec.bind_decode_filter(fmt, ws, cs, oks, W.TFILTER_SUM, W.SHAPE_FLAT, res_d, rok_d)
    call_h = codec.bind_decode_filter(fmt, ws, cs, oks, W.TFILTER_SUM, W.SHAPE_FLAT, res_h, rok_h)
    flag = rok_h.numpy()
    val = res_h.numpy().view("int32")
    reps = 2000
    for _ in range(20):
        call_d(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call_d(); torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / reps
    for _ in range(20):
        flag[0] = 0; call_h()
        while flag[0] == 0: pass
    t0 = time.perf_counter()
    for _ in range(reps):
        flag[0] = 0
        call_h()
        while flag[0] == 0:
            pass
    t_poll = (time.perf_counter() - t0) / reps
    assert int(val[0]) == n
    print("n=%6d  synchronize %.2f us   polled from pinned host memory %.2f us" % (n, t_sync * 1e6, t_poll * 1e6))
