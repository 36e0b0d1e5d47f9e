"""Diagnostic: where the time of the headline step goes (each part timed alone, back to back)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench
codec = W.Codec(max_packets=1 << 18)
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 65536)
st = bench.ArrayWaveStep(codec, n, m, 3)
for _ in range(3):
    st.step()
parts = {
    "decode_begin": st.decode, "decode_finish": st.decode_payload,
    "sizeof": lambda: codec.sizeof(st.fmt, st.cols, st.n, st.align, st.phase, want_total=False, lens=st.out_lens, offsets=st.out_offs),
    "encode": lambda: codec.encode_into(st.fmt, st.cols, st.dst),
    "step": st.step,
}
for k, fn in parts.items():
    print("%-14s %8.1f us" % (k, bench.time_call(fn, 100) * 1e6))
