"""Diagnostic: what the result's system-scope fence costs a small filter call (C1 and C2 node work)."""
import sys, time
sys.path.insert(0, ".")
import torch
from mrnet_amd import wave as W
codec = W.Codec(max_packets=1 << 17)
dev = codec.device
fmt = W.Format("%d")
res = torch.empty(16, dtype=torch.uint8, device=dev); rok = torch.empty(1, dtype=torch.int32, device=dev)
for n in (4, 65536):
    w = W.WireWave(torch.ones(n, dtype=torch.int32, device=dev).view(torch.uint8), n, None, None, 4)
    cols, ok = codec.alloc_columns(fmt, n), torch.empty(n, dtype=torch.int32, device=dev)
    call = codec.bind_decode_filter(fmt, w, cols, ok, W.TFILTER_SUM, W.SHAPE_FLAT, res, rok)
    for _ in range(20): call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(2000): call()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 2000 * 1e6)
    print("n=%d  %.2f us per wave (streamed)" % (n, best))
