#!/usr/bin/env python3
"""tfilter_PerfData, one invocation over 131,072 double packets: arrays of one length against drawn
lengths 0 .. 8 (the dense-copy path) and against the same ragged wave with a bound too small for the
column paths' scratch (the serial chain)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrnet_amd import wave as W  # noqa: E402


def main():
    codec = W.Codec()
    n, k = 131072, 8
    fmt = W.Format("%ad %ad %alf")
    rng = np.random.default_rng(3)
    for name, lens in (("regular", np.full(n, k, np.uint64)), ("ragged", rng.integers(0, k + 1, n).astype(np.uint64))):
        vals = rng.normal(1e-3, 1e-4, int(lens.sum()))
        ones = np.ones(n, np.uint64)
        cols = codec.columns_from_numpy(fmt, [(ones, np.arange(n, dtype=np.int32)), (ones, lens.astype(np.int32)), (lens, vals)])
        for op, opn in ((W.TFILTER_AVG, "avg"), (W.TFILTER_SUM, "sum"), (W.TFILTER_MIN, "min")):
            fn = lambda: codec.filter_perfdata(op, W.DataType.DOUBLE_T, cols[0], cols[2], n, group=0, out_cap=k)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            print("%-8s %-4s %8.1f us per wave" % (name, opn, (time.perf_counter() - t) / 20 * 1e6), flush=True)


if __name__ == "__main__":
    main()
