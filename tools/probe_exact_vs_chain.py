"""Diagnostic: the flat float / double Sum of 65,536 packets on growing, zero-mean and hostile data,
the device-wide exact fold against the plain chain (MRPDR_SHAPE_FLAT_SERIAL), with a check that both
leave the same bits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrnet_amd import wave as W
n = int(os.environ.get("PROBE_N", "65536"))
codec = W.Codec(max_packets=n)
dev = codec.device
res = torch.empty(16, dtype=torch.uint8, device=dev)
rok = torch.empty(1, dtype=torch.int32, device=dev)


def timed(v, ty, shape):
    f = codec.bind_filter(W.TFILTER_SUM, ty, v.data_ptr(), v.element_size(), None, 0, n, 0, shape, res, rok)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        f()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / 30 * 1e6, 1), bytes(res.cpu().numpy())


for dt, ty in ((torch.float32, W.DataType.FLOAT_T), (torch.float64, W.DataType.DOUBLE_T)):
    g = torch.Generator(device=dev).manual_seed(1)
    kinds = {"growing": torch.rand(n, dtype=dt, device=dev, generator=g) + 0.5}
    for sd in range(5):
        kinds["zero_mean_%d" % sd] = torch.randn(n, dtype=dt, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + sd))
    e = torch.randint(-20, 20, (n,), device=dev, generator=g).to(dt)
    kinds["wide_range"] = torch.randn(n, dtype=dt, device=dev, generator=g) * torch.exp2(e)
    x = torch.randn(n // 2, dtype=dt, device=dev, generator=g)
    kinds["cancelling"] = torch.stack([x, -x], 1).reshape(-1).contiguous()
    for name, v in kinds.items():
        a, ra = timed(v, ty, W.SHAPE_FLAT)
        b, rb = timed(v, ty, W.SHAPE_FLAT_SERIAL)
        print("%-8s %-12s exact fold %7.1f us   plain chain %7.1f us   %+5.1f %%   %s" % (str(dt)[6:], name, a, b, 100.0 * (a - b) / b,
                                                                                  "same bits" if ra == rb else "DIFFERENT"), flush=True)

# every kind of data the parity tests draw (tests/test_gpu_flat_scan.py), the same comparison
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np                                                     # noqa: E402
from test_gpu_flat_scan import datasets                                # noqa: E402
worst = {}
for dt, ty, npdt in ((torch.float32, W.DataType.FLOAT_T, np.float32), (torch.float64, W.DataType.DOUBLE_T, np.float64)):
    for name, host in datasets(npdt, n, 7).items():
        v = torch.from_numpy(host.copy()).to(dev)
        a, ra = timed(v, ty, W.SHAPE_FLAT)
        b, rb = timed(v, ty, W.SHAPE_FLAT_SERIAL)
        same = ra == rb or (np.isnan(np.frombuffer(ra[:v.element_size()], npdt)[0]) and np.isnan(np.frombuffer(rb[:v.element_size()], npdt)[0]))
        pct = 100.0 * (a - b) / b
        worst[str(dt)[6:]] = max(worst.get(str(dt)[6:], -100.0), pct)
        print("%-8s %-20s exact fold %7.1f us   plain chain %7.1f us   %+5.1f %%   %s" % (str(dt)[6:], name, a, b, pct,
                                                                                          "same bits" if same else "DIFFERENT"), flush=True)
print("worst against the chain:", {k: "%+.1f %%" % v for k, v in worst.items()})
