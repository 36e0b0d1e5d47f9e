"""Diagnostic: a mixed-format wave of shorter packets (byte arrays of at most 360), 1,048,576 packets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench
big = W.Codec(max_packets=1 << 20)
print(bench.mixed_wave_extra(big, n=1 << 20, key="short", amax=360))
