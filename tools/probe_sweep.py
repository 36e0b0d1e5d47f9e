"""Diagnostic: the '%af' round trip (decode a foreign-order wave, encode it again) over array
lengths from 4 to 65,536 floats at about 256 MiB of payload per wave."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench
codec = W.Codec(max_packets=1 << 22)
for m in (4, 16, 64, 256, 1024, 4096, 16384, 65536):
    n = min(1 << 21, (256 << 20) // (4 * m))
    st = bench.ArrayWaveStep(codec, n, m, 3)
    for _ in range(3):
        st.step()
    t = bench.time_call(st.step, 20)
    nbytes = 2 * n * (8 + 4 * m) + 2 * n * 4 * m
    print("m=%6d n=%8d step %9.1f us  %7.1f GB/s" % (m, n, t * 1e6, nbytes / t / 1e9), flush=True)
    del st
    torch.cuda.empty_cache()
