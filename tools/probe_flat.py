"""Diagnostic: one reference call of tfilter_Sum / tfilter_Avg over float and double waves -- the scan
(MRPDR_SHAPE_FLAT) beside the plain chain (MRPDR_SHAPE_FLAT_SERIAL) on several kinds of data."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from mrnet_amd import wave as W
from test_gpu_flat_scan import datasets
from wavegen import O
codec = W.Codec(max_packets=1 << 20)
res = torch.empty(16, dtype=torch.uint8, device=codec.device)
rok = torch.empty(1, dtype=torch.int32, device=codec.device)


def t_call(f, reps=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for t, dt in ((O.FLOAT_T, np.float32), (O.DOUBLE_T, np.float64)):
    sets = datasets(dt, n, 1)
    for sd in (1, 2, 3):
        sets["randn%d" % sd] = np.random.default_rng(sd).standard_normal(n).astype(dt)
    for name, vals in sets.items():
        dv = codec.to_device(vals)
        w = vals.dtype.itemsize
        a = codec.bind_filter(W.TFILTER_SUM, t, dv.data_ptr(), w, None, 0, n, 0, W.SHAPE_FLAT, res, rok)
        b = codec.bind_filter(W.TFILTER_SUM, t, dv.data_ptr(), w, None, 0, n, 0, W.SHAPE_FLAT_SERIAL, res, rok)
        print("%-8s %-20s n=%d  scan %7.1f us   chain %7.1f us" % (np.dtype(dt).name, name, n, t_call(a), t_call(b)))
