"""Payload kernel in isolation: back-to-back launches of the decode and encode payload
stage on the C3 wave (diagnostic; prints ms per launch from the library's HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench

codec = W.Codec(max_packets=1 << 18)
st = bench.ArrayWaveStep(codec, 4096, 65536, 3)
st.step(); torch.cuda.synchronize()
codec.perf_enable(True)
for name, fn in (("decode x20", lambda: [st.decode_payload() for _ in range(20)]),
                 ("encode x20", lambda: [codec.encode_into(st.fmt, st.cols, st.dst) for _ in range(20)]),
                 ("alternating x20", lambda: [(st.decode_payload(), codec.encode_into(st.fmt, st.cols, st.dst)) for _ in range(20)]),
                 ("full step x20", lambda: [st.step() for _ in range(20)])):
    fn(); codec.perf_collect()
    fn(); p = codec.perf_collect()
    print("%-18s decode %.4f ms (%d)  encode %.4f ms (%d)" % (
        name, p["decode"][0] / max(1, p["decode"][1]), p["decode"][1],
        p["encode"][0] / max(1, p["encode"][1]), p["encode"][1]))
