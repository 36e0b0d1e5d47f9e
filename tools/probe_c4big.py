"""Diagnostic: the 1,048,576-packet mixed-format wave only (for counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench
big = W.Codec(max_packets=1 << 20)
align = int(sys.argv[1]) if len(sys.argv) > 1 else 16
print(bench.mixed_wave_extra(big, n=1 << 20, key="c4_x16", align=align))
