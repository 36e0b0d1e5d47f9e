"""Diagnostic: the mixed variable-length wave (config 4) alone, for rocprofv3 --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench
codec = W.Codec(max_packets=1 << 18)
print(bench.mixed_wave_extra(codec))
