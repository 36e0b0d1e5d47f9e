"""Diagnostic: one kind of data through the flat float Sum, for counter passes (argv: kind, n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from mrnet_amd import wave as W
from test_gpu_flat_scan import datasets
from wavegen import O
kind = sys.argv[1] if len(sys.argv) > 1 else "constant"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
codec = W.Codec(max_packets=1 << 20)
vals = datasets(np.float32, n, 1)[kind]
dv = codec.to_device(vals)
for _ in range(5):
    res, ok = codec.filter_columns(O.SUM, O.FLOAT_T, dv, None, W.SHAPE_FLAT)
torch.cuda.synchronize()
print(kind, n, W.result_value(res, ok, np.dtype(np.float32), O.SUM))
