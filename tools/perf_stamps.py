"""Diagnostic: where a workgroup of k_perf_nodes spends its life (the library built with
-DMRPDR_TILE_STAMPS as tools/libmrpdr_stamps.so: lane 0 notes the 100 MHz wall clock at the phase
boundaries).  One 131,072-packet PerfData shard through the tree given on the command line."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MRPDR_LIB"] = os.path.join(ROOT, "tools", "libmrpdr_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mrnet_amd import wave as W

fan = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,32,32,0").split(",")]
n, k = 131072, 8
codec = W.Codec(max_packets=1 << 18)
dev = codec.device
lib = C.CDLL(os.environ["MRPDR_LIB"])
lib.mrpdr_debug_tile_stamps.argtypes = [C.c_void_p]
ranks = W.Column(torch.arange(n, dtype=torch.int32, device=dev), torch.ones(n, dtype=torch.int64, device=dev),
                 torch.arange(n + 1, dtype=torch.int64, device=dev), n)
vals = torch.randn(n * k, dtype=torch.float64, device=dev) * 1e-4 + 1e-3
data = W.Column(vals, torch.full((n,), k, dtype=torch.int64, device=dev), torch.arange(n + 1, dtype=torch.int64, device=dev) * k, n * k)
outs = (codec.empty(k, torch.int64), codec.empty(1, torch.int64), codec.empty(1, torch.int32))
scr = codec.empty(1 << 23, torch.uint8)
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)


def run(f):
    codec.filter_perfdata_levels(W.TFILTER_AVG, W.DataType.DOUBLE_T, ranks, data, n, f, out_cap=k, outputs=outs, scratch=scr)


for _ in range(5):
    run(fan)
torch.cuda.synchronize()
# the first launch alone: a plan of one fused pair whose stamps are not overwritten by a second launch
lib.mrpdr_debug_tile_stamps(stamps.data_ptr())
run(fan[:2] + [0] if len(fan) > 2 else fan)
torch.cuda.synchronize()
lib.mrpdr_debug_tile_stamps(None)
s = stamps.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] != 0]
names = ["start -> records in LDS", "-> ranks in LDS, data asked for", "-> folded", "-> level above folded"]
print("workgroups stamped:", len(s), "(some are overwritten by the later, smaller launch)")
for i, nm in enumerate(names):
    d = (s[:, i + 1] - s[:, i]) * 10.0       # ns
    d = d[(s[:, i + 1] != 0) & (d >= 0) & (d < 1e6)]
    if len(d):
        print("%-40s median %.2f us  p90 %.2f us" % (nm, np.median(d) / 1e3, np.percentile(d, 90) / 1e3))
first, last = s[:, 0].min(), s[:, 1:5].max()
print("first start -> last stamp: %.2f us; spread of the starts: %.2f us" % ((last - first) / 100.0, (s[:, 0].max() - first) / 100.0))
