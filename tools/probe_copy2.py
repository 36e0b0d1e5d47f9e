"""Diagnostic: is the encode payload kernel's time a property of the kernel or of the buffers?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench

codec = W.Codec(max_packets=1 << 18)
st = bench.ArrayWaveStep(codec, 4096, 65536, 3)
st.step(); torch.cuda.synchronize()
codec.perf_enable(True)

def run(name, fn, reps=20):
    for _ in range(3): fn()
    codec.perf_collect()
    for _ in range(reps): fn()
    p = codec.perf_collect()
    print("%-46s decode %.4f ms (%d)  encode %.4f ms (%d)" % (
        name, p["decode"][0] / max(1, p["decode"][1]), p["decode"][1],
        p["encode"][0] / max(1, p["encode"][1]), p["encode"][1]), flush=True)

run("decode BE src->cols", st.decode_payload)
run("encode cols->dst", lambda: codec.encode_into(st.fmt, st.cols, st.dst))
# decode the little-endian wire we just produced (no swap), same kernel direction as decode
le = W.WireWave(st.dst.arena, st.n, st.dst.offsets, st.dst.lens, 0)
run("decode LE dst->cols", lambda: codec.decode_finish(st.fmt, le, st.cols, W.LITTLEENDIAN))
# encode into the buffer that held the big-endian source
alt = W.WireWave(st.src.arena, st.n, st.dst.offsets, st.dst.lens, 0)
run("encode cols->src buffer", lambda: codec.encode_into(st.fmt, st.cols, alt))
big = torch.empty(st.dst.arena.numel() + (8 << 20), dtype=torch.uint8, device="cuda")
for off in (0, 8, 4096, 65536, 1 << 20, (1 << 20) + 8, 3 << 20):
    w = W.WireWave(big[off:off + st.dst.arena.numel()], st.n, st.dst.offsets, st.dst.lens, 0)
    run("encode cols->fresh buffer + %d" % off, lambda: codec.encode_into(st.fmt, st.cols, w))
cols2 = [W.Column(torch.empty_like(st.cols[0].values), st.cols[0].counts, st.cols[0].offs, capacity=st.cols[0].capacity)]
run("decode BE src->fresh cols", lambda: codec.decode_finish(st.fmt, st.src, cols2, W.BIGENDIAN))
