"""Diagnostic: the flat float Sum over zero-mean draws (the bench's five seeds and five more)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
n = 65536
codec = W.Codec(max_packets=n)
dev = codec.device
res = torch.empty(16, dtype=torch.uint8, device=dev)
rok = torch.empty(1, dtype=torch.int32, device=dev)
out = []
for dt, ty in ((torch.float32, W.DataType.FLOAT_T), (torch.float64, W.DataType.DOUBLE_T)):
    row = []
    for sd in range(10):
        v = torch.randn(n, dtype=dt, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + sd))
        f = codec.bind_filter(W.TFILTER_SUM, ty, v.data_ptr(), v.element_size(), None, 0, n, 0, W.SHAPE_FLAT, res, rok)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            f()
        torch.cuda.synchronize()
        row.append(round((time.perf_counter() - t0) / 30 * 1e6, 1))
    print(str(dt), row, "mean %.1f" % (sum(row) / len(row)))
