"""PerfData shard (131,072 x '%ad %ad %alf'[8]) through the declared trees, one call per wave with
preallocated outputs: for rocprofv3 --kernel-trace --stats (per-kernel time) and for wall time."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from mrnet_amd import wave as W  # noqa: E402

codec = W.Codec(max_packets=1 << 18)
dev = codec.device
n, k = int(os.environ.get('PROBE_N', '131072')), int(os.environ.get('PROBE_K', '8'))
if os.environ.get("PROBE_ONE_BLOCK"):
    # every column carved out of ONE allocation (are the separate tensors' pages what a level waits for?)
    big = torch.zeros(n * (4 + 8 + 8 + 8 + 8) + n * k * 8 + 4096, dtype=torch.uint8, device=dev)
    at = [0]

    def carve(count, dtype):
        size = count * torch.empty(0, dtype=dtype).element_size()
        v = big[at[0]:at[0] + size].view(dtype)
        at[0] += (size + 255) // 256 * 256
        return v
    rv, rcn, rof = carve(n, torch.int32), carve(n, torch.int64), carve(n + 1, torch.int64)
    dcn, dof, dv = carve(n, torch.int64), carve(n + 1, torch.int64), carve(n * k, torch.float64)
    rv.copy_(torch.arange(n, dtype=torch.int32, device=dev)); rcn.fill_(1); rof.copy_(torch.arange(n + 1, device=dev))
    dcn.fill_(k); dof.copy_(torch.arange(n + 1, device=dev) * k); dv.copy_(torch.randn(n * k, dtype=torch.float64, device=dev) * 1e-4 + 1e-3)
    ranks, data = W.Column(rv, rcn, rof, n), W.Column(dv, dcn, dof, n * k)
else:
    ranks = W.Column(torch.arange(n, dtype=torch.int32, device=dev), torch.ones(n, dtype=torch.int64, device=dev),
                     torch.arange(n + 1, dtype=torch.int64, device=dev), n)
    vals = torch.randn(n * k, dtype=torch.float64, device=dev) * 1e-4 + 1e-3
    data = W.Column(vals, torch.full((n,), k, dtype=torch.int64, device=dev),
                    torch.arange(n + 1, dtype=torch.int64, device=dev) * k, n * k)
outs = (codec.empty(k, torch.int64), codec.empty(1, torch.int64), codec.empty(1, torch.int32))
scr = codec.empty(1 << 23, torch.uint8)
shapes = [[int(x) for x in a.split(',')] for a in sys.argv[1:]] or [[32, 32, 32, 0], [32, 64, 0], [16, 64, 0], [64, 64, 0], [32, 32, 0], [128, 32, 0], [256, 0], [64, 0]]
for fan in shapes:
    def run():
        codec.filter_perfdata_levels(W.TFILTER_AVG, W.DataType.DOUBLE_T, ranks, data, n, fan, out_cap=k, outputs=outs, scratch=scr)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        run()
    torch.cuda.synchronize()
    print(fan, "%.1f us per wave" % ((time.perf_counter() - t0) / 100 * 1e6))
