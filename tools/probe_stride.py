"""Diagnostic: the device-wide exact fold over a column of 131,072 doubles, contiguous against one of
eight interleaved columns (64 bytes from packet to packet, the layout of performance-data arrays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
n, k = 131072, 8
codec = W.Codec(max_packets=n)
dev = codec.device
res = torch.empty(16, dtype=torch.uint8, device=dev)
rok = torch.empty(1, dtype=torch.int32, device=dev)
m = torch.randn(n, k, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 1e-4 + 1e-3
col = m[:, 3].contiguous()
for name, ptr, stride in (("contiguous", col.data_ptr(), 8), ("one of eight", m.data_ptr() + 24, 64)):
    f = codec.bind_filter(W.TFILTER_SUM, W.DataType.DOUBLE_T, ptr, stride, None, 0, n, 0, W.SHAPE_FLAT, res, rok)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    print("%-14s %7.1f us   %s" % (name, (time.perf_counter() - t0) / 50 * 1e6, bytes(res.cpu().numpy())[:8].hex()), flush=True)
