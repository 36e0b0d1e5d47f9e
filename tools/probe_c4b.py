"""Diagnostic: config-4 decode/encode timing only (no result checks), for kernel experiments."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mrnet_amd import wave as W
import bench
codec = W.Codec(max_packets=1 << 18)
n = 65536
fmt = W.Format("%d %uld %lf %s %auc")
rng = np.random.default_rng(4)
slen = np.minimum(rng.geometric(1.0 / 12.0, n), 255).astype(np.uint64) + 1
alen = rng.integers(0, 513, n).astype(np.uint64)
sbody = rng.integers(33, 127, int(slen.sum()), dtype=np.uint8)
cols = codec.columns_from_numpy(fmt, [rng.integers(-2**31, 2**31, n).astype(np.int32), rng.integers(0, 2**63, n).astype(np.uint64),
    rng.uniform(-1e6, 1e6, n), (slen, sbody), (alen, rng.integers(0, 256, int(alen.sum()), dtype=np.uint8))])
ALIGN = int(os.environ.get('C4_ALIGN', '16'))
lens, offs, total = codec.sizeof(fmt, cols, n, ALIGN, 0)
wire = W.WireWave(torch.zeros(total, dtype=torch.uint8, device=codec.device), n, offs, lens, 0)
dcols = codec.alloc_columns(fmt, n, capacities=[0, 0, 0, int(slen.sum()), int(alen.sum())])
ok = torch.empty(n, dtype=torch.int32, device=codec.device)
codec.encode_into(fmt, cols, wire)
codec.decode_begin(fmt, wire, dcols, ok, want_totals=False)
t_enc = bench.time_call(lambda: codec.encode_into(fmt, cols, wire), 50)
t_beg = bench.time_call(lambda: codec.decode_begin(fmt, wire, dcols, ok, want_totals=False), 50)
t_fin = bench.time_call(lambda: codec.decode_finish(fmt, wire, dcols), 50)
print(os.environ.get("MRPDR_LIB", "default"), "align", ALIGN, "encode %.1f us  decode_begin %.1f us  decode_finish %.1f us" % (t_enc * 1e6, t_beg * 1e6, t_fin * 1e6))
