"""Diagnostic: one kind of the parity tests' data through the exact fold, 30 times (for rocprofv3 --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mrnet_amd import wave as W
from test_gpu_flat_scan import datasets
kind, n = sys.argv[1], 65536
codec = W.Codec(max_packets=n)
res = torch.empty(16, dtype=torch.uint8, device=codec.device); rok = torch.empty(1, dtype=torch.int32, device=codec.device)
for npdt, ty in ((np.float32, W.DataType.FLOAT_T),):
    v = torch.from_numpy(datasets(npdt, n, 7)[kind].copy()).to(codec.device)
    f = codec.bind_filter(W.TFILTER_SUM, ty, v.data_ptr(), v.element_size(), None, 0, n, 0, W.SHAPE_FLAT, res, rok)
    for _ in range(30):
        f()
    torch.cuda.synchronize()
