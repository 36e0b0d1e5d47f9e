"""Diagnostic: the performance-data wave (131,072 x '%ad %ad %alf'[8], Avg, two levels) alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
import bench
codec = W.Codec(max_packets=1 << 18)
print(bench.perfdata_extra(codec))
