"""Diagnostic: the mixed-format wave at 65,536 and 1,048,576 packets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench
big = W.Codec(max_packets=1 << 20)
print(bench.mixed_wave_extra(big, n=1 << 16, key="c4"))
print(bench.mixed_wave_extra(big, n=1 << 20, key="c4_x16"))
