"""The 1M-packet mixed wave (config 4 x 16), encode and one-call decode, for rocprofv3:
    rocprofv3 --kernel-trace --stats -d gpurun_out/p -- python3 tools/probe_c4x16.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mrnet_amd import wave as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
codec = W.Codec(max_packets=max(n, 1 << 18))
print(bench.mixed_wave_extra(codec, n=n, key="c4_x16" if n == 1 << 20 else "c4"))
