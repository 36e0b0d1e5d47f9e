"""C2 (65,536 x '%d': decode + Sum in one pass, then re-encode -- three kernels) issued eagerly and
replayed from a HIP graph, for rocprofv3 --kernel-trace: the trace shows where the replay's extra
microseconds lie (the kernels' own durations, or the gaps between them).  Phase markers: a torch fill
kernel of 7, 8 or 9 elements is launched in front of the eager loop, the replay loop and at the end."""
import sys
import time

import torch

sys.path.insert(0, ".")
from mrnet_amd import wave as W  # noqa: E402

codec = W.Codec(max_packets=1 << 17)
dev = codec.device
n = 65536
fmt = W.Format("%d")
vals = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=dev)
wire = W.WireWave(vals.view(torch.uint8).clone(), n, None, None, 4)
res = torch.empty(16, dtype=torch.uint8, device=dev)
rok = torch.empty(1, dtype=torch.int32, device=dev)
cols = codec.alloc_columns(fmt, n)
ok = torch.empty(n, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
reps = 200
with torch.cuda.stream(side):
    dec_sum = codec.bind_decode_filter(fmt, wire, cols, ok, W.TFILTER_SUM, W.SHAPE_FLAT, res, rok)
    enc = codec.bind_encode(fmt, cols, wire)

    def step():
        dec_sum()
        enc()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step()
    torch.cuda.synchronize()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    marker = torch.empty(16, dtype=torch.int32, device=dev)
    marker[:7].fill_(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    marker[:8].fill_(2)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    marker[:9].fill_(3)
    torch.cuda.synchronize()
print("eager %.2f us per wave, replay %.2f us per wave (host clock, %d waves each)" % ((t1 - t0) / reps * 1e6, (t3 - t2) / reps * 1e6, reps))
