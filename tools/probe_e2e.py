"""Diagnostic: host-to-host pipelining of the headline wave with different range / stream counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench
codec = W.Codec(max_packets=1 << 18)
for slices, nbuf in ((4, 2), (8, 2), (8, 3), (16, 3), (16, 4), (32, 4), (64, 4)):
    g, s = bench.end_to_end_pipelined(codec, 4096, 65536, slices=slices, nbuf=nbuf, reps=3)
    print("slices %2d buffers %d: %.1f GB/s  %.2f ms" % (slices, nbuf, g, s * 1e3), flush=True)
# copies alone
n = 4096 * (8 + 4 * 65536)
h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(n, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
import time
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
print("H2D alone %.2f ms" % t(lambda: d.copy_(h, non_blocking=True)))
print("D2H alone %.2f ms" % t(lambda: h2.copy_(d2, non_blocking=True)))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
print("H2D + D2H on two streams %.2f ms" % t(both))
