"""Diagnostic: where a tile of the short-packet kernels spends its life.

Needs tools/libmrpdr_stamps.so (the library built with -DMRPDR_TILE_STAMPS: lane 0 of every workgroup
notes the 100 MHz wall clock at the phase boundaries).  Runs the 1,048,576-packet mixed wave once
through encode and once through the one-call decode and prints the time between the boundaries.
"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MRPDR_LIB"] = os.path.join(ROOT, "tools", "libmrpdr_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mrnet_amd import wave as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
codec = W.Codec(max_packets=max(n, 1 << 18))
lib = C.CDLL(os.environ["MRPDR_LIB"])
lib.mrpdr_debug_tile_stamps.argtypes = [C.c_void_p]
fmt = W.Format("%d %uld %lf %s %auc")
rng = np.random.default_rng(4)
slen = np.minimum(rng.geometric(1.0 / 12.0, n), 255).astype(np.uint64) + 1
alen = rng.integers(0, 513, n).astype(np.uint64)
sbody = rng.integers(33, 127, int(slen.sum()), dtype=np.uint8)
sbody[np.cumsum(slen).astype(np.int64) - 1] = 0
cols = codec.columns_from_numpy(fmt, [
    rng.integers(-2**31, 2**31, n).astype(np.int32), rng.integers(0, 2**63, n).astype(np.uint64),
    rng.uniform(-1e6, 1e6, n), (slen, sbody), (alen, rng.integers(0, 256, int(alen.sum()), dtype=np.uint8))])
lens, offs, total = codec.sizeof(fmt, cols, n, 16, 0)
wire = W.WireWave(torch.zeros(total, dtype=torch.uint8, device=codec.device), n, offs, lens, 0)
dcols = codec.alloc_columns(fmt, n, capacities=[0, 0, 0, int(slen.sum()), int(alen.sum())])
ok = torch.empty(n, dtype=torch.int32, device=codec.device)
ntiles = (n + 63) // 64
stamps = torch.zeros(ntiles * 8, dtype=torch.int64, device=codec.device)
for _ in range(3):
    codec.encode_into(fmt, cols, wire)
    codec.decode_into(fmt, wire, dcols, ok)
torch.cuda.synchronize()


def report(name, labels):
    s = stamps.cpu().numpy().reshape(ntiles, 8).astype(np.float64) * 0.01      # microseconds
    t0 = s[:, 0].min()
    last = len(labels)
    print("%s: %d tiles, kernel span %.1f us" % (name, ntiles, s[:, last].max() - t0))
    life = s[:, last] - s[:, 0]
    print("  tile life: mean %.2f  median %.2f  p90 %.2f us" % (life.mean(), np.median(life), np.percentile(life, 90)))
    for k, lab in enumerate(labels):
        d = s[:, k + 1] - s[:, k]
        print("  %-58s mean %6.2f  median %6.2f  p90 %6.2f us" % (lab, d.mean(), np.median(d), np.percentile(d, 90)))
    starts = np.sort(s[:, 0] - t0)
    print("  tile starts: p10 %.1f  p50 %.1f  p90 %.1f  last %.1f us" % (starts[int(0.1 * ntiles)], starts[ntiles // 2],
                                                                    starts[int(0.9 * ntiles)], starts[-1]))


lib.mrpdr_debug_tile_stamps(stamps.data_ptr())
stamps.zero_()
torch.cuda.synchronize()
codec.encode_into(fmt, cols, wire)
torch.cuda.synchronize()
report("k_tile_encode", ["boundaries (loads, two barriers)", "first field's loads issued, scalars and prefixes written",
                         "first field: wait for the loads, into the image", "barrier", "other fields: loads, wait, into the image",
                         "barrier", "image -> arena stores issued"])
stamps.zero_()
torch.cuda.synchronize()
codec.decode_into(fmt, wire, dcols, ok)
torch.cuda.synchronize()
report("k_tile_decode", ["boundaries (loads, two barriers)", "image loads issued, waited for, stored to LDS (lane 0)",
                         "field positions", "barrier (all wavefronts' pieces in the image)", "scalars",
                         "payloads: image -> column stores issued"])
lib.mrpdr_debug_tile_stamps(None)
assert bool(ok.all())
