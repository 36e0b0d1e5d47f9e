#!/usr/bin/env python3
"""Headline benchmark: device-resident PDR decode+encode GB/s per wave.

Workload (the large-array configuration the GB/s metric is quoted on): per GPU a wave of
4,096 packets, each '%af' with 65,536 floats, written by a BIG-endian sender.  One step =
  decode : walk + count scan + byte-reversing payload kernel  (wire -> float column)
  sizeof : per-packet lengths + offset scan                   (the pdr_sizeof pass)
  encode : prefixes + payload kernel                          (float column -> little-endian wire)
with every buffer resident in HBM before the clock starts and no host synchronisation
inside a step.  N GPUs each take their own packet range of an N x 4,096-packet wave
(weak scaling, no data-path collective).

Bytes counted are ALGORITHMIC (minimum traffic): decode = data bytes read + decoded bytes
written; encode = decoded bytes read + data bytes written + 8(n+1) offset bytes.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mrnet_amd import wave as W  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak
SWEEP_SIZES = (4, 64, 1024, 16384, 65536)      # packets per wave of the node-wave sweep (GPU and CPU legs)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--packets", type=int, default=4096, help="packets per GPU")
    p.add_argument("--elems", type=int, default=65536, help="floats per packet")
    p.add_argument("--cpu-packets", type=int, default=4096, help="packets of the CPU baseline sample")
    p.add_argument("--cpu-reps", type=int, default=4)
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-extras", action="store_true")
    p.add_argument("--backend", default="nccl", help="nccl (= RCCL) for the real run; gloo to rehearse "
                                                     "several ranks on one GPU")
    p.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    p.add_argument("--exchange-timeout", type=int, default=240,
                   help="N > 1: seconds the sharded-filter leg behind the timed region may take before every "
                        "rank gives up on it (the headline line is printed either way)")
    p.add_argument("--perf-every", type=int, default=5,
                   help="bracket the payload kernels of every k-th step with HIP events (1 = every step)")
    p.add_argument("--overlap", action="store_true",
                   help="run the encode side's sizing, prefixes and plan on a second stream under the decode "
                        "payload kernel (measured: no gain -- the payload kernel fills the machine and the "
                        "event hand-offs cost what the hidden kernels save; default: one stream)")
    return p.parse_args()


def make_be_wave(n, m, seed, device):
    """n packets '%af'[m] as a big-endian sender wrote them, built on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    stride = 8 + 4 * m
    wire = torch.empty((n, stride), dtype=torch.uint8, device=device)
    wire[:, :8] = torch.tensor(list(int(m).to_bytes(8, "big")), dtype=torch.uint8, device=device)
    step = max(1, min(n, (1 << 28) // (4 * m)))
    for a in range(0, n, step):          # in slices: the generator's scratch stays small
        b = min(n, a + step)
        v = torch.randint(-2**31, 2**31 - 1, (b - a, m), dtype=torch.int32, device=device, generator=g)
        wire[a:b, 8:] = v.view(torch.uint8).view(b - a, m, 4).flip(-1).reshape(b - a, 4 * m)
        del v
    offs = torch.arange(n + 1, dtype=torch.int64, device=device) * stride
    lens = torch.full((n,), stride, dtype=torch.int64, device=device)
    return W.WireWave(wire.view(-1), n, offs, lens, 0)


class ArrayWaveStep:
    """Preallocated decode -> sizeof -> encode of one '%af' wave."""

    def __init__(self, codec, n, m, seed, overlap=False):
        self.codec, self.n, self.m = codec, n, m
        dev = codec.device
        # the encode side has its own context (workspace) so that its preparation can run on a
        # second stream while the decode payload kernel streams
        self.overlap = overlap
        self.enc = W.Codec(device=dev.index, max_packets=codec.max_packets) if overlap else codec
        self.side = torch.cuda.Stream(device=dev, priority=-1) if overlap else None
        self.ev_counts, self.ev_prepared = torch.cuda.Event(), torch.cuda.Event()
        self.fmt = W.Format("%af")
        self.src = make_be_wave(n, m, seed, dev)
        self.cols = [W.Column(torch.empty(n * m, dtype=torch.int32, device=dev),
                              torch.empty(n, dtype=torch.int64, device=dev),
                              torch.empty(n + 1, dtype=torch.int64, device=dev), capacity=n * m)]
        self.ok = torch.empty(n, dtype=torch.int32, device=dev)
        self.out_lens = torch.empty(n, dtype=torch.int64, device=dev)
        self.out_offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
        # Output arena layout (ours to choose, the packet bytes do not change): every packet
        # in a 128-byte aligned slot, shifted so that its float payload starts on a cache line.
        self.align = 128
        self.phase = codec.payload_phase(self.fmt, self.align)
        self.slot = (self.phase + 8 + 4 * m + self.align - 1) // self.align * self.align
        self.dst = W.WireWave(torch.zeros(n * self.slot, dtype=torch.uint8, device=dev), n,
                              self.out_offs, self.out_lens, 0)
        data = n * (8 + 4 * m)
        payload = n * 4 * m
        self.bytes_decode = data + payload
        self.bytes_encode = payload + data + 8 * (n + 1)
        self.bytes_step = self.bytes_decode + self.bytes_encode
        self.bytes_copy_kernel = 2 * payload          # one launch of the payload kernel

    def decode(self):
        self.codec.decode_begin(self.fmt, self.src, self.cols, self.ok, W.BIGENDIAN, want_totals=False)

    def decode_payload(self):
        self.codec.decode_finish(self.fmt, self.src, self.cols, W.BIGENDIAN)

    def encode_prepare(self):
        self.enc.sizeof(self.fmt, self.cols, self.n, self.align, self.phase, want_total=False,
                        lens=self.out_lens, offsets=self.out_offs)
        self.enc.encode_begin(self.fmt, self.cols, self.dst)

    def encode(self):
        self.encode_prepare()
        self.enc.encode_finish(self.fmt, self.cols, self.dst)

    def step(self):
        if not self.overlap:
            self.decode()
            self.decode_payload()
            self.encode()
            return
        main = torch.cuda.current_stream()
        self.decode()                                   # counts, offsets, scalars
        self.ev_counts.record(main)
        with torch.cuda.stream(self.side):              # needs only what decode() left
            self.side.wait_event(self.ev_counts)
            self.encode_prepare()
            self.ev_prepared.record(self.side)
        self.decode_payload()                           # meanwhile: the decode payload stream
        main.wait_event(self.ev_prepared)
        self.enc.encode_finish(self.fmt, self.cols, self.dst)

    def verify(self):
        """decode(BE) is the byte reversal of the wire; encode gives the LE wire of the same values."""
        n, m = self.n, self.m
        stride = 8 + 4 * m
        src = self.src.arena.view(n, stride)
        k = min(n, 64)
        want = src[:k, 8:].reshape(k, m, 4).flip(-1).reshape(k, 4 * m).contiguous().view(torch.int32)
        got = self.cols[0].values.view(n, m)[:k]
        assert torch.equal(got, want), "decoded values differ from the byte-reversed wire"
        assert torch.equal(self.out_offs[:n], torch.arange(n, device=src.device) * self.slot + self.phase)
        assert bool((self.out_lens == stride).all()) and int(self.out_offs[n]) == n * self.slot
        dst = self.dst.arena.view(n, self.slot)[:, self.phase:self.phase + stride]
        assert torch.equal(dst[:k, 8:].contiguous().view(torch.int32), want), "encoded payload differs"
        assert bool((dst[:, :8].contiguous().view(torch.int64) == m).all()), "encoded counts differ"
        assert bool(self.ok.all()) and self.codec.status() == 0


def cpu_baseline(n, m, reps):
    """The oracle (CPU restatement of the reference's per-element path) on a bounded sample
    of the same workload: one thread (the reported baseline), then the same packets split over
    the host cores this process may use, then a vectorised CPU line (numpy byteswap + copies)
    so that the comparison is not only against a per-element loop."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 2**32, size=n * m, dtype=np.uint32)
    counts = np.full(n, m, np.uint64)
    offs_e = np.arange(n + 1, dtype=np.uint64) * np.uint64(m)
    cols = [{"type": O.FLOAT_ARRAY_T, "counts": counts, "offs": offs_e, "values": vals.view(np.float32)}]
    arena, offs, lens = O.wave_encode("%af", cols, n)
    # the big-endian sender's bytes of the same packets
    be = arena.reshape(n, 8 + 4 * m).copy()
    be[:, :8] = be[:, 7::-1]
    be[:, 8:] = be[:, 8:].reshape(n, m, 4)[:, :, ::-1].reshape(n, 4 * m)
    be = be.reshape(-1)
    data, payload = n * (8 + 4 * m), n * 4 * m
    nbytes = (data + payload) + (payload + data + 8 * (n + 1))
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        dec, ok = O.wave_decode("%af", be, offs, lens, O.BIG)
        t1 = time.perf_counter()
        out, _, _ = O.wave_encode("%af", dec, n)
        t2 = time.perf_counter()
        dt = t2 - t0
        if best is None or dt < best[0]:
            best = (dt, t1 - t0, t2 - t1)
    assert ok.all() and np.array_equal(out, arena)

    # all cores: packets statically partitioned over threads (the oracle's C code runs without the GIL)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)), n))
    stride = 8 + 4 * m
    bounds = [n * t // cores for t in range(cores + 1)]

    def part(t):
        a, b = bounds[t], bounds[t + 1]
        k = b - a
        o = np.arange(k + 1, dtype=np.uint64) * np.uint64(stride)
        d, okp = O.wave_decode("%af", be[a * stride:b * stride], o, lens[a:b], O.BIG)
        e, _, _ = O.wave_encode("%af", d, k)
        return bool(okp.all()) and np.array_equal(e, arena[a * stride:b * stride])
    with ThreadPoolExecutor(cores) as pool:
        list(pool.map(part, range(cores)))                   # warm-up
        t0 = time.perf_counter()
        good = list(pool.map(part, range(cores)))
        t_all = time.perf_counter() - t0
    assert all(good)

    # vectorised CPU: what memcpy / bswap-vectorised code does with the same bytes (one thread)
    rows = be.reshape(n, stride)
    t0 = time.perf_counter()
    cnt = rows[:, :8].copy().view(">u8").astype("<u8")
    vals = rows[:, 8:].copy().view(">u4").astype("<u4")      # decode: reverse every element
    outv = np.empty((n, stride), np.uint8)
    outv[:, :8] = cnt.view(np.uint8)
    outv[:, 8:] = vals.view(np.uint8)                         # encode: native order
    t_vec = time.perf_counter() - t0
    assert np.array_equal(outv.reshape(-1), arena)
    # config 1 beside it: the 4-packet '%d' wave (value 1, as microbench_BE sends it) through the
    # oracle's per-packet path -- decode four packets, tfilter_Sum, encode the one output packet
    one = [{"type": O.INT32_T, "values": np.ones(4, np.int32)}]
    a1, o1, l1 = O.wave_encode("%d", one, 4)
    reps1 = 2000
    t0 = time.perf_counter()
    for _ in range(reps1):
        d1, ok1 = O.wave_decode("%d", a1, o1, l1, O.LITTLE)
        s1 = O.filter_flat(O.SUM, O.INT32_T, d1[0]["values"])
        e1, _, _ = O.wave_encode("%d", [{"type": O.INT32_T, "values": np.array([s1[0]], np.int32)}], 1)
    c1_ctypes_us = (time.perf_counter() - t0) / reps1 * 1e6
    assert int(e1.view(np.int32)[0]) == 4
    # the figure above is mostly Python and foreign-function calls; the same loop inside the C library
    c1_us = O.bench_small_wave(200000) * 1e6
    # and for growing waves of the same kind (a tree node with n children: decode n x '%d', Sum, encode one)
    sweep = {str(n): round(O.bench_small_wave(max(10, 200000 // n), n) * 1e6, 2) for n in SWEEP_SIZES}
    return {"value": round(nbytes / best[0] / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "c1_us_per_wave": round(c1_us, 2), "c1_us_per_wave_via_ctypes": round(c1_ctypes_us, 2),
            "node_wave_us_by_packets": sweep,
            "sample": "%d packets x '%%af'[%d] (%.0f MiB of data), decode big-endian + sizeof + encode, "
                      "best of %d, decode %.2f s encode %.2f s" % (n, m, data / 2**20, reps, best[1], best[2]),
            "all_cores": {"value": round(nbytes / t_all / 1e9, 3), "unit": "GB/s", "cores": cores,
                          "note": "same oracle, packets split statically over threads"},
            "vectorised": {"value": round(nbytes / t_vec / 1e9, 3), "unit": "GB/s", "cores": 1,
                           "note": "numpy byteswap + block copies instead of the per-element path"}}


def small_wave_extras(codec):
    """Launch-bound configs, reported as time per wave (not the headline)."""
    out = {}
    dev = codec.device
    # config 2: 65,536 x '%d' -- fused decode + Sum straight on the wire, then full decode + encode
    n = 65536
    fmt = W.Format("%d")
    vals = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=dev)
    wire = W.WireWave(vals.view(torch.uint8).clone(), n, None, None, 4)
    res = torch.empty(16, dtype=torch.uint8, device=dev)
    rok = torch.empty(1, dtype=torch.int32, device=dev)
    cols = codec.alloc_columns(fmt, n)
    ok = torch.empty(n, dtype=torch.int32, device=dev)

    def c2_unfused():
        codec.filter_raw(W.TFILTER_SUM, W.DataType.INT32_T, wire.arena.data_ptr(), 4, None, 4, n, 0,
                         W.SHAPE_TREE, res, rok)
        codec.decode_into(fmt, wire, cols, ok)
        codec.encode_into(fmt, cols, wire)
    out["c2_us_per_wave_unfused"] = round(time_call(c2_unfused, 200) * 1e6, 2)
    # the same work as a node does it wave after wave: decode + Sum in one pass over the rows
    # (two launches), the re-encode (one launch), the calls bound to their buffers once
    def bind_c2():          # bound to the stream that is current when this runs
        dec_sum = codec.bind_decode_filter(fmt, wire, cols, ok, W.TFILTER_SUM, W.SHAPE_FLAT, res, rok)
        enc = codec.bind_encode(fmt, cols, wire)

        def c2():
            dec_sum()
            enc()
        return c2
    out["c2_us_per_wave"] = round(time_call(bind_c2(), 200) * 1e6, 2)
    tg = time_graph(bind_c2, 200)
    out["c2_us_per_wave_hipgraph"] = None if tg is None else round(tg * 1e6, 2)
    assert int(res[:4].view(torch.int32).item()) == int(vals.sum(dtype=torch.int32).item())
    assert torch.equal(cols[0].values.view(torch.int32), vals) and torch.equal(wire.arena.view(torch.int32), vals)
    # config 1: the 4-packet '%d' wave of the reference's microbench (tests/microbench_FE.C:61,145,207
    # over local-1x4.top): decode + Sum; the 16 result bytes are the encoded output packet
    w1 = W.WireWave(torch.ones(4, dtype=torch.int32, device=dev).view(torch.uint8), 4, None, None, 4)
    cols1, ok1 = codec.alloc_columns(fmt, 4), torch.empty(4, dtype=torch.int32, device=dev)
    c1 = codec.bind_decode_filter(fmt, w1, cols1, ok1, W.TFILTER_SUM, W.SHAPE_FLAT, res, rok)
    out["c1_us_per_wave"] = round(time_call(c1, 500) * 1e6, 2)

    def c1_latency():
        c1()
        torch.cuda.synchronize()
    out["c1_us_per_wave_host_waits"] = round(time_call(c1_latency, 500) * 1e6, 2)
    assert int(res[:4].view(torch.int32).item()) == 4
    # the same node work for growing waves (n children): where the device starts to pay.  Two figures
    # each: calls streamed back to back, and with the host waiting for every result (what a node
    # that forwards one wave at a time sees).  The CPU leg reports the oracle's figure for the same n.
    sweep = {}
    for ns in SWEEP_SIZES:
        ws = W.WireWave(torch.ones(ns, dtype=torch.int32, device=dev).view(torch.uint8), ns, None, None, 4)
        cs, oks = codec.alloc_columns(fmt, ns), torch.empty(ns, dtype=torch.int32, device=dev)
        call = codec.bind_decode_filter(fmt, ws, cs, oks, W.TFILTER_SUM, W.SHAPE_FLAT, res, rok)

        def call_wait():
            call()
            torch.cuda.synchronize()
        # third figure: the result written by the kernel straight into pinned host memory, the host
        # polling result_ok there instead of synchronising the stream
        res_h, rok_h = torch.zeros(16, dtype=torch.uint8).pin_memory(), torch.zeros(1, dtype=torch.int32).pin_memory()
        call_h = codec.bind_decode_filter(fmt, ws, cs, oks, W.TFILTER_SUM, W.SHAPE_FLAT, res_h, rok_h)
        flag = rok_h.numpy()

        def call_poll():
            flag[0] = -1                     # MRPDR_RESULT_PENDING: 0 and 1 are both final values
            call_h()
            spins = 0
            while flag[0] == -1:
                spins += 1
                if spins > 200_000_000:
                    raise RuntimeError("the polled result never arrived")
        sweep[str(ns)] = {"streamed": round(time_call(call, 300) * 1e6, 2),
                          "host_waits": round(time_call(call_wait, 300) * 1e6, 2),
                          "host_polls_pinned_result": round(time_call(call_poll, 300) * 1e6, 2)}
        assert int(res[:4].view(torch.int32).item()) == ns and int(res_h.numpy()[:4].view(np.int32)[0]) == ns
    out["node_wave_us_by_packets"] = sweep
    # what ONE reference filter call (the flat shape, the mirrors' default) costs where the order of
    # the additions matters -- a serial chain -- beside the declared tree; 65,536 packets each
    gx = torch.Generator(device=dev).manual_seed(20)
    vf = torch.randn(n, dtype=torch.float32, device=dev, generator=gx)
    rows64 = torch.empty((n, 12), dtype=torch.uint8, device=dev)
    rows64[:, :8] = (torch.randn(n, dtype=torch.float64, device=dev, generator=gx) * 1e-4 + 1e-3).view(torch.uint8).view(n, 8)
    rows64[:, 8:] = torch.ones(n, dtype=torch.int32, device=dev).view(torch.uint8).view(n, 4)
    p64 = rows64.data_ptr()
    for name, shape in (("flat", W.SHAPE_FLAT), ("tree", W.SHAPE_TREE)):
        f_sum = codec.bind_filter(W.TFILTER_SUM, W.DataType.FLOAT_T, vf.data_ptr(), 4, None, 0, n, 0, shape, res, rok)
        f_avg = codec.bind_filter(W.TFILTER_AVG, W.DataType.DOUBLE_T, p64, 12, p64 + 8, 12, n, 0, shape, res, rok)
        i_avg = codec.bind_filter(W.TFILTER_AVG, W.DataType.INT32_T, p64, 12, p64 + 8, 12, n, 0, shape, res, rok)
        out["%s_sum_f32_64k_us" % name] = round(time_call(f_sum, 50) * 1e6, 1)
        if name == "flat":
            # a zero-mean sum's figure depends on the draw (how long the sum stays near zero, where it
            # changes its binade every few packets): five draws, the median under the plain key
            draws = []
            for sd in range(5):
                vz = torch.randn(n, dtype=torch.float32, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + sd))
                f_z = codec.bind_filter(W.TFILTER_SUM, W.DataType.FLOAT_T, vz.data_ptr(), 4, None, 0, n, 0, shape, res, rok)
                draws.append(round(time_call(f_z, 30) * 1e6, 1))
            draws.sort()
            out["flat_sum_f32_64k_us"] = draws[2]
            out["flat_sum_f32_64k_zero_mean_draws_us"] = draws
            # the flat float results are the reference's chain of additions; large waves take it a binade
            # at a time with a scan.  What that buys depends on the data: a sum that grows (timings,
            # counts) stays in a binade for long stretches, a zero-mean one hovers around powers of two
            vpos = torch.rand(n, dtype=torch.float32, device=dev, generator=gx)
            f_pos = codec.bind_filter(W.TFILTER_SUM, W.DataType.FLOAT_T, vpos.data_ptr(), 4, None, 0, n, 0, shape, res, rok)
            f_pos_chain = codec.bind_filter(W.TFILTER_SUM, W.DataType.FLOAT_T, vpos.data_ptr(), 4, None, 0, n, 0,
                                            W.SHAPE_FLAT_SERIAL, res, rok)
            f_avg_chain = codec.bind_filter(W.TFILTER_AVG, W.DataType.DOUBLE_T, p64, 12, p64 + 8, 12, n, 0,
                                            W.SHAPE_FLAT_SERIAL, res, rok)
            out["flat_sum_f32_64k_positive_us"] = round(time_call(f_pos, 50) * 1e6, 1)
            out["flat_sum_f32_64k_positive_plain_chain_us"] = round(time_call(f_pos_chain, 50) * 1e6, 1)
            out["flat_avg_f64_64k_plain_chain_us"] = round(time_call(f_avg_chain, 50) * 1e6, 1)
        out["%s_avg_f64_64k_us" % name] = round(time_call(f_avg, 50) * 1e6, 1)
        out["%s_avg_i32_64k_us" % name] = round(time_call(i_avg, 50) * 1e6, 1)
    # config 5 shape: '%lf %d', fused Min / Max / Avg on the wire (one GPU's 131,072-packet shard)
    n5 = 131072
    fmt5 = W.Format("%lf %d")
    v = (torch.randn(n5, dtype=torch.float64, device=dev) * 1e-4 + 1e-3)
    rows = torch.empty((n5, 12), dtype=torch.uint8, device=dev)
    rows[:, :8] = v.view(torch.uint8).view(n5, 8)
    rows[:, 8:] = torch.ones(n5, dtype=torch.int32, device=dev).view(torch.uint8).view(n5, 4)
    w5 = W.WireWave(rows.view(-1), n5, None, None, 12)
    r3 = [torch.empty(16, dtype=torch.uint8, device=dev) for _ in range(3)]

    def c5():
        for r, op in zip(r3, (W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG)):
            codec.filter_raw(op, W.DataType.DOUBLE_T, w5.arena.data_ptr(), 12, w5.arena.data_ptr() + 8, 12,
                             n5, 0, W.SHAPE_TREE, r, rok)
    out["c5_shard_us_per_wave_three_filters"] = round(time_call(c5, 200) * 1e6, 2)
    # the same three filters as ONE reference call each over the shard (MRPDR_SHAPE_FLAT, what the mirrors
    # default to): Min and Max level by level with the NaN rule carried along, Avg by the exact chain
    rf = [torch.empty(16, dtype=torch.uint8, device=dev) for _ in range(3)]

    def c5_one_call():
        for r, op in zip(rf, (W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG)):
            codec.filter_raw(op, W.DataType.DOUBLE_T, w5.arena.data_ptr(), 12, w5.arena.data_ptr() + 8, 12,
                             n5, 0, W.SHAPE_FLAT, r, rok)
    out["c5_shard_us_per_wave_one_reference_call_each"] = round(time_call(c5_one_call, 100) * 1e6, 2)
    r48 = torch.empty(48, dtype=torch.uint8, device=dev)
    ok3 = torch.empty(3, dtype=torch.int32, device=dev)

    def c5_fused():
        codec.filter_stats_raw(W.DataType.DOUBLE_T, w5.arena.data_ptr(), 12, w5.arena.data_ptr() + 8, 12, n5, 0,
                               r48, ok3)
    out["c5_shard_us_per_wave"] = round(time_call(c5_fused, 200) * 1e6, 2)
    for k in range(3):
        assert torch.equal(r48[16 * k:16 * k + 16], r3[k]), "fused Min/Max/Avg differs from the separate filters"
    out.update(perfdata_extra(codec))
    out.update(mixed_wave_extra(codec))
    big = W.Codec(device=dev.index, max_packets=1 << 20)
    out.update(mixed_wave_extra(big, n=1 << 20, key="c4_x16"))
    del big
    out.update(batched_small_waves_extra(dev))
    out.update(message_extra(codec))
    out.update(host_to_host_extra(codec))
    assert float(r3[0][:8].view(torch.float64).item()) == float(v.min().item())
    assert float(r3[1][:8].view(torch.float64).item()) == float(v.max().item())
    return out


def batched_small_waves_extra(dev):
    """Config 2 again, 64 waves per launch: 64 x 65,536 '%d' packets decoded from a big-endian
    sender and re-encoded by one call each -- the only way a 256 KiB wave can be measured
    against the memory system instead of against launch latency."""
    waves, n1 = 64, 65536
    n = waves * n1
    codec = W.Codec(device=dev.index, max_packets=n)
    fmt = W.Format("%d")
    vals = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=dev)
    wire = W.WireWave(vals.view(torch.uint8).clone(), n, None, None, 4)
    out_wire = W.WireWave(torch.empty(4 * n, dtype=torch.uint8, device=dev), n, None, None, 4)
    cols = codec.alloc_columns(fmt, n)
    ok = torch.empty(n, dtype=torch.int32, device=dev)

    def step():
        codec.decode_into(fmt, wire, cols, ok, W.BIGENDIAN)
        codec.encode_into(fmt, cols, out_wire)
    t = time_call(step, 50)
    want = vals.view(torch.uint8).view(n, 4).flip(-1).contiguous().view(torch.int32).view(-1)
    assert torch.equal(cols[0].values.view(torch.int32), want) and bool(ok.all())
    assert torch.equal(out_wire.arena.view(torch.int32), want)
    nbytes = 4 * 4 * n + 4 * n                        # wire + column, both ways, + ok flags
    return {"c2_x64_us_per_launch_pair": round(t * 1e6, 1), "c2_x64_gbps": round(nbytes / t / 1e9, 1)}


def perfdata_extra(codec):
    """Config 5, array variant: one GPU's shard of 131,072 '%ad %ad %alf' packets with 8 doubles
    each (the shape Stream::collect_PerfData emits), element-wise Avg
      - as a declared tree 16 x 32 x 32 x 8 (nodes of 16 packets, of 32 nodes, of 32 nodes, one node over
        the 8 results: two launches, each folding two levels -- the nodes a workgroup holds are the
        children of one node of the level above);
      - as the two-level tree of the earlier rounds (256 x 512), for comparison;
      - as ONE invocation over all packets (what the mirrors default to): doubles are a serial chain of
        131,072 roundings per element; int64 data of the same shape does not depend on the grouping and
        is folded as a tree of undivided nodes (the same bits as the serial loop)."""
    n, k = 131072, 8
    dev = codec.device
    ranks = W.Column(torch.arange(n, dtype=torch.int32, device=dev), torch.ones(n, dtype=torch.int64, device=dev),
                     torch.arange(n + 1, dtype=torch.int64, device=dev), n)
    vals = torch.randn(n * k, dtype=torch.float64, device=dev) * 1e-4 + 1e-3
    data = W.Column(vals, torch.full((n,), k, dtype=torch.int64, device=dev),
                    torch.arange(n + 1, dtype=torch.int64, device=dev) * k, n * k)
    ivals = (vals * 1e9).to(torch.int64)
    idata = W.Column(ivals, data.counts, data.offs, n * k)
    res = {}

    outs3 = (codec.empty(k, torch.int64), codec.empty(1, torch.int64), codec.empty(1, torch.int32))
    scr3 = codec.empty(1 << 22, torch.uint8)

    def tree3():            # (a node that filters wave after wave keeps its output packet and scratch)
        res["o"] = codec.filter_perfdata_levels(W.TFILTER_AVG, W.DataType.DOUBLE_T, ranks, data, n, [16, 32, 32, 0], out_cap=k,
                                                outputs=outs3, scratch=scr3)

    def tree2():
        res["o2"] = codec.filter_perfdata(W.TFILTER_AVG, W.DataType.DOUBLE_T, ranks, data, n, group=256, out_cap=k)

    def flat():
        res["f"] = codec.filter_perfdata(W.TFILTER_AVG, W.DataType.DOUBLE_T, ranks, data, n, group=0, out_cap=k)

    def flat_i64():
        res["i"] = codec.filter_perfdata(W.TFILTER_AVG, W.DataType.INT64_T, ranks, idata, n, group=0, out_cap=k)
    t3, t2 = time_call(tree3, 50), time_call(tree2, 50)
    tf, ti = time_call(flat, 5), time_call(flat_i64, 50)
    want = vals.view(n, k).mean(dim=0)
    for key in ("o", "o2", "f"):
        out, out_len, out_rank = res[key]
        assert int(out_len.item()) == k and int(out_rank.item()) == -n
        assert torch.allclose(out[:k].view(torch.float64), want, rtol=1e-9)
    out, out_len, out_rank = res["i"]
    # (the int64 sums of this data do not wrap: the plain sum over all packets, divided once, truncating toward zero)
    sums = ivals.view(n, k).sum(dim=0)
    assert int(out_len.item()) == k and torch.equal(out[:k], torch.div(sums, n, rounding_mode="trunc"))
    return {"c5b_perfdata_avg_us_per_wave": round(t3 * 1e6, 1), "c5b_perfdata_tree": "16 x 32 x 32 x 8",
            "c5b_perfdata_avg_two_level_256_us_per_wave": round(t2 * 1e6, 1),
            "c5b_perfdata_avg_flat_us_per_wave": round(tf * 1e6, 1),
            "c5b_perfdata_avg_i64_flat_us_per_wave": round(ti * 1e6, 1)}


def mixed_wave_extra(codec, n=65536, key="c4", align=16, amax=512):
    """Config 4: 65,536 packets '%d %uld %lf %s %auc' with variable-length fields (string length
    geometric, mean 12, max 255; byte array length uniform in [0, 512]).  With n = 1,048,576 (key
    c4_x16) the same wave sixteen times over: what the format costs per byte once launches no longer
    dominate."""
    dev = codec.device
    fmt = W.Format("%d %uld %lf %s %auc")
    rng = np.random.default_rng(4)
    slen = np.minimum(rng.geometric(1.0 / 12.0, n), 255).astype(np.uint64) + 1      # strlen+1
    alen = rng.integers(0, amax + 1, n).astype(np.uint64)
    sbody = rng.integers(33, 127, int(slen.sum()), dtype=np.uint8)
    sbody[np.cumsum(slen).astype(np.int64) - 1] = 0
    cols = codec.columns_from_numpy(fmt, [
        rng.integers(-2**31, 2**31, n).astype(np.int32), rng.integers(0, 2**63, n).astype(np.uint64),
        rng.uniform(-1e6, 1e6, n), (slen, sbody), (alen, rng.integers(0, 256, int(alen.sum()), dtype=np.uint8))])
    lens, offs, total = codec.sizeof(fmt, cols, n, align, 0)
    wire = W.WireWave(torch.zeros(total, dtype=torch.uint8, device=dev), n, offs, lens, 0)
    dcols = codec.alloc_columns(fmt, n, capacities=[0, 0, 0, int(slen.sum()), int(alen.sum())])
    ok = torch.empty(n, dtype=torch.int32, device=dev)

    def enc():
        codec.sizeof(fmt, cols, n, align, 0, want_total=False, lens=lens, offsets=offs)
        codec.encode_into(fmt, cols, wire)

    def dec():
        codec.decode_into(fmt, wire, dcols, ok)
    t_enc, t_dec = time_call(enc, 50), time_call(dec, 50)
    assert bool(ok.all()) and codec.status() == 0
    for i in (3, 4):
        assert torch.equal(dcols[i].values.view(torch.uint8), cols[i].values.view(torch.uint8))
    for i in (0, 1, 2):
        assert torch.equal(dcols[i].values.view(torch.uint8), cols[i].values.view(torch.uint8))
    data = int(lens.sum().item())
    decoded = 20 * n + int(slen.sum()) + int(alen.sum())
    nbytes = 2 * (data + decoded) + 8 * (n + 1)
    res = {key + "_encode_us": round(t_enc * 1e6, 1), key + "_decode_us": round(t_dec * 1e6, 1),
           key + "_gbps": round(nbytes / (t_enc + t_dec) / 1e9, 1), key + "_wire_mib": round(data / 2**20, 1)}
    return res


def message_extra(codec):
    """The unit MRNet puts on a link (src/Message.C): config 2 and config 4 waves built as one
    Message each (preamble + headers + data, encoded in place) and taken apart again (size vector,
    headers, data).  Both include the host's read of the packet count / the payload size, as
    Message::send / recv need them."""
    out = {}
    dev = codec.device
    rng = np.random.default_rng(6)
    n = 65536
    waves = {"c2": (W.Format("%d"), [rng.integers(-2**31, 2**31, n).astype(np.int32)])}
    slen = np.minimum(rng.geometric(1.0 / 12.0, n), 255).astype(np.uint64) + 1
    alen = rng.integers(0, 513, n).astype(np.uint64)
    sbody = rng.integers(33, 127, int(slen.sum()), dtype=np.uint8)
    sbody[np.cumsum(slen).astype(np.int64) - 1] = 0
    waves["c4"] = (W.Format("%d %uld %lf %s %auc"), [
        rng.integers(-2**31, 2**31, n).astype(np.int32), rng.integers(0, 2**63, n).astype(np.uint64),
        rng.uniform(-1e6, 1e6, n), (slen, sbody), (alen, rng.integers(0, 256, int(alen.sum()), dtype=np.uint8))])
    for name, (fmt, host_cols) in waves.items():
        cols = codec.columns_from_numpy(fmt, host_cols)
        msg = codec.message_build(fmt, cols, n)

        def build():
            codec.message_build(fmt, cols, n)

        def parse():
            _, hdr, wire = codec.message_parse(fmt, msg)
            codec.decode(fmt, wire)

        # the same for a node in its steady state: it read the count and the length off the link
        # itself, and its buffers (message, offsets, header fields, columns) exist from earlier waves
        ws = codec.message_workspace(n)
        caps = [None if c.capacity is None else int(c.capacity) for c in cols]
        dcols, dok = codec.alloc_columns(fmt, n, caps), codec.empty(n, torch.int32)
        lens, offsets = codec.empty(n, torch.int64), codec.empty(n + 1, torch.int64)
        out_msg = torch.empty_like(msg)

        def build_steady():
            codec.sizeof(fmt, cols, n, 1, fmt.header_bytes, want_total=False, lens=lens, offsets=offsets)
            codec.message_build_into(fmt, cols, n, lens, offsets, out_msg)

        def parse_steady():
            _, hdr, wire = codec.message_parse(fmt, msg, n=n, check_length=False, workspace=ws)
            codec.decode_into(fmt, wire, dcols, dok)
        tb, tp = time_call(build, 30), time_call(parse, 30)
        tbs, tps = time_call(build_steady, 30), time_call(parse_steady, 30)
        assert torch.equal(out_msg, msg) and bool(dok.all()) and bool(ws["ok"].all())
        assert torch.equal(dcols[-1].values.view(torch.uint8)[:cols[-1].values.numel() * cols[-1].values.element_size()],
                           cols[-1].values.view(torch.uint8))
        _, hdr, wire = codec.message_parse(fmt, msg)
        dcols, ok = codec.decode(fmt, wire)
        assert bool(ok.all()) and bool(hdr["ok"].all())
        assert torch.equal(dcols[0].values.view(torch.uint8), cols[0].values.view(torch.uint8))
        out["f1_%s_message_build_us" % name] = round(tb * 1e6, 1)
        out["f1_%s_message_parse_decode_us" % name] = round(tp * 1e6, 1)
        out["f1_%s_message_build_steady_us" % name] = round(tbs * 1e6, 1)
        out["f1_%s_message_parse_decode_steady_us" % name] = round(tps * 1e6, 1)
        out["f1_%s_message_mib" % name] = round(msg.numel() / 2**20, 1)
    return out


def host_to_host_extra(codec):
    """The filter path and the mixed-format codec from host memory to host memory: what a tree node
    pays per wave when its receive buffers are host memory (src/Message.C:48-199 reads the link into
    host buffers, src/Filter.C:42-117 then runs the filter).  Wall time of the host thread from the
    first copy's enqueue to the result being readable on the host; inputs and outputs pinned.
      C2        65,536 x '%d' rows (256 KiB) up, decode + Sum (mrpdr_wave_decode_filter), the 16 result
                bytes written by the kernel into pinned memory and polled there
      C5 shard  131,072 x '%lf %d' rows (1.5 MiB) up, fused Min / Max / Avg, 48 result bytes polled
      C4        65,536 mixed packets: wire up + decode to device columns; columns -> sizeof + encode +
                wire down (18.8 MiB each way)
    The oracle's time for the same waves is in cpu_baseline (host_to_host_cpu)."""
    out = {}
    dev = codec.device

    def poll(flags):
        spins = 0
        while (flags == -1).any():
            spins += 1
            if spins > 100_000_000:
                raise RuntimeError("the polled result never arrived")

    # ---- C2 ----
    n = 65536
    fmt = W.Format("%d")
    host_rows = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32).pin_memory()
    dev_rows = torch.empty(n, dtype=torch.int32, device=dev)
    wire = W.WireWave(dev_rows.view(torch.uint8), n, None, None, 4)
    cols, ok = codec.alloc_columns(fmt, n), torch.empty(n, dtype=torch.int32, device=dev)
    res_h = torch.zeros(16, dtype=torch.uint8).pin_memory()
    rok_h = torch.zeros(1, dtype=torch.int32).pin_memory()
    call = codec.bind_decode_filter(fmt, wire, cols, ok, W.TFILTER_SUM, W.SHAPE_FLAT, res_h, rok_h)
    flag = rok_h.numpy()

    def c2():
        flag[0] = -1
        dev_rows.copy_(host_rows, non_blocking=True)
        call()
        poll(flag)
    out["c2_host_to_host_us"] = round(time_call(c2, 200) * 1e6, 1)
    assert int(res_h.numpy()[:4].view(np.int32)[0]) == int(host_rows.sum(dtype=torch.int32).item())
    filt = codec.bind_filter(W.TFILTER_SUM, W.DataType.INT32_T, dev_rows.data_ptr(), 4, None, 0, n, 0, W.SHAPE_FLAT,
                             res_h, rok_h, owners=(dev_rows,))

    def c2_filter_only():                       # the fold alone, straight on the uploaded rows (no columns written)
        flag[0] = -1
        dev_rows.copy_(host_rows, non_blocking=True)
        filt()
        poll(flag)
    out["c2_host_to_host_filter_only_us"] = round(time_call(c2_filter_only, 200) * 1e6, 1)

    # ---- C5 shard ----
    n5 = 131072
    v = torch.randn(n5, dtype=torch.float64) * 1e-4 + 1e-3
    rows = torch.empty((n5, 12), dtype=torch.uint8)
    rows[:, :8] = v.view(torch.uint8).view(n5, 8)
    rows[:, 8:] = torch.ones(n5, dtype=torch.int32).view(torch.uint8).view(n5, 4)
    host5 = rows.view(-1).pin_memory()
    dev5 = torch.empty(n5 * 12, dtype=torch.uint8, device=dev)
    r48 = torch.zeros(48, dtype=torch.uint8).pin_memory()
    ok3 = torch.zeros(3, dtype=torch.int32).pin_memory()
    flags3 = ok3.numpy()
    p5 = dev5.data_ptr()

    def c5():
        flags3[:] = -1
        dev5.copy_(host5, non_blocking=True)
        codec.filter_stats_raw(W.DataType.DOUBLE_T, p5, 12, p5 + 8, 12, n5, 0, r48, ok3)
        poll(flags3)
    out["c5_shard_host_to_host_us"] = round(time_call(c5, 200) * 1e6, 1)
    assert float(r48.numpy()[:8].view(np.float64)[0]) == float(v.min().item())

    # ---- C4 ----
    n4 = 65536
    fmt4 = W.Format("%d %uld %lf %s %auc")
    rng = np.random.default_rng(4)
    slen = np.minimum(rng.geometric(1.0 / 12.0, n4), 255).astype(np.uint64) + 1
    alen = rng.integers(0, 513, n4).astype(np.uint64)
    sbody = rng.integers(33, 127, int(slen.sum()), dtype=np.uint8)
    sbody[np.cumsum(slen).astype(np.int64) - 1] = 0
    cols4 = codec.columns_from_numpy(fmt4, [
        rng.integers(-2**31, 2**31, n4).astype(np.int32), rng.integers(0, 2**63, n4).astype(np.uint64),
        rng.uniform(-1e6, 1e6, n4), (slen, sbody), (alen, rng.integers(0, 256, int(alen.sum()), dtype=np.uint8))])
    lens, offs, total = codec.sizeof(fmt4, cols4, n4, 16, 0)
    wire4 = W.WireWave(torch.zeros(total, dtype=torch.uint8, device=dev), n4, offs, lens, 0)
    codec.encode_into(fmt4, cols4, wire4)
    host_wire = torch.empty(total, dtype=torch.uint8).pin_memory()
    host_wire.copy_(wire4.arena)
    host_offs, host_lens = offs.cpu().pin_memory(), lens.cpu().pin_memory()
    d_offs, d_lens = torch.empty_like(offs), torch.empty_like(lens)
    in_wire = W.WireWave(torch.empty(total, dtype=torch.uint8, device=dev), n4, d_offs, d_lens, 0)
    dcols = codec.alloc_columns(fmt4, n4, capacities=[0, 0, 0, int(slen.sum()), int(alen.sum())])
    ok4 = torch.empty(n4, dtype=torch.int32, device=dev)

    def c4_in():            # the size vector's offsets / lengths travel with the wire, as a Message's do
        in_wire.arena.copy_(host_wire, non_blocking=True)
        d_offs.copy_(host_offs, non_blocking=True)
        d_lens.copy_(host_lens, non_blocking=True)
        codec.decode_into(fmt4, in_wire, dcols, ok4)
        torch.cuda.synchronize()
    out["c4_host_to_columns_us"] = round(time_call(c4_in, 30) * 1e6, 1)
    assert bool(ok4.all()) and torch.equal(dcols[4].values.view(torch.uint8), cols4[4].values.view(torch.uint8))
    host_out = torch.empty(total, dtype=torch.uint8).pin_memory()
    out_lens_h, out_offs_h = torch.empty_like(host_lens), torch.empty_like(host_offs)

    def c4_out():
        codec.sizeof(fmt4, cols4, n4, 16, 0, want_total=False, lens=lens, offsets=offs)
        codec.encode_into(fmt4, cols4, wire4)
        host_out.copy_(wire4.arena, non_blocking=True)
        out_lens_h.copy_(lens, non_blocking=True)
        out_offs_h.copy_(offs, non_blocking=True)
        torch.cuda.synchronize()
    out["c4_columns_to_host_us"] = round(time_call(c4_out, 30) * 1e6, 1)
    assert torch.equal(host_out, host_wire)
    out["c4_wire_mib"] = round(total / 2**20, 1)
    return out


def host_to_host_cpu():
    """The oracle's time (one core) for the waves of host_to_host_extra: what the node's existing CPU
    path spends on the same work, host buffers in, host buffers out."""
    from oracle import oracle as O
    out = {}
    rng = np.random.default_rng(4)
    # C2: decode 65,536 x '%d', tfilter_Sum, one output packet (timed inside the C library)
    out["c2_us"] = round(O.bench_small_wave(10, 65536) * 1e6, 1)
    # C5 shard: decode 131,072 x '%lf %d', then Min, Max and Avg over the decoded values
    n5 = 131072
    c5 = [{"type": O.DOUBLE_T, "values": rng.normal(1e-3, 1e-4, n5)}, {"type": O.INT32_T, "values": np.ones(n5, np.int32)}]
    a5, o5, l5 = O.wave_encode("%lf %d", c5, n5)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        d5, ok5 = O.wave_decode("%lf %d", a5, o5, l5, O.LITTLE)
        for op in (O.MIN, O.MAX):
            O.filter_flat(op, O.DOUBLE_T, d5[0]["values"])
        O.filter_flat(O.AVG, O.DOUBLE_T, d5[0]["values"], d5[1]["values"])
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["c5_shard_us"] = round(best * 1e6, 1)
    # C4: 65,536 mixed packets decoded to columns, and encoded from columns
    n4 = 65536
    slen = np.minimum(rng.geometric(1.0 / 12.0, n4), 255).astype(np.uint64) + 1
    alen = rng.integers(0, 513, n4).astype(np.uint64)
    sbody = rng.integers(33, 127, int(slen.sum()), dtype=np.uint8)
    sbody[np.cumsum(slen).astype(np.int64) - 1] = 0

    def offs_of(c):
        o = np.zeros(c.size + 1, np.uint64)
        np.cumsum(c, out=o[1:])
        return o
    c4 = [{"type": O.INT32_T, "values": rng.integers(-2**31, 2**31, n4).astype(np.int32)},
          {"type": O.UINT64_T, "values": rng.integers(0, 2**63, n4).astype(np.uint64)},
          {"type": O.DOUBLE_T, "values": rng.uniform(-1e6, 1e6, n4)},
          {"type": O.STRING_T, "counts": slen, "offs": offs_of(slen), "values": sbody},
          {"type": O.UCHAR_ARRAY_T, "counts": alen, "offs": offs_of(alen), "values": rng.integers(0, 256, int(alen.sum()), dtype=np.uint8)}]
    t0 = time.perf_counter()
    a4, o4, l4 = O.wave_encode("%d %uld %lf %s %auc", c4, n4)
    out["c4_columns_to_host_us"] = round((time.perf_counter() - t0) * 1e6, 1)
    t0 = time.perf_counter()
    d4, ok4 = O.wave_decode("%d %uld %lf %s %auc", a4, o4, l4, O.LITTLE)
    out["c4_host_to_columns_us"] = round((time.perf_counter() - t0) * 1e6, 1)
    assert ok4.all()
    # the fold alone: the 65,536-term float chain of tfilter_Sum on one host core (the decoded column is
    # already there), beside flat_sum_f32_64k_us
    vf = rng.standard_normal(65536).astype(np.float32)
    best = None
    for _ in range(20):
        t0 = time.perf_counter()
        O.filter_flat(O.SUM, O.FLOAT_T, vf)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["fold_only_us"] = round(best * 1e6, 1)
    vd = rng.normal(1e-3, 1e-4, 65536)
    best = None
    for _ in range(20):
        t0 = time.perf_counter()
        O.filter_flat(O.AVG, O.DOUBLE_T, vd, np.ones(65536, np.int32))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["fold_only_avg_f64_us"] = round(best * 1e6, 1)
    # and the chain with everything per-packet taken away (a tight loop of dependent additions):
    # the fastest a host core produces the same bits
    out["fold_only_tight_loop_us"] = round(O.bench_chain(vf) * 1e6, 1)
    out["fold_only_tight_loop_f64_us"] = round(O.bench_chain(vd) * 1e6, 1)
    return out


def time_call(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def time_graph(make_fn, reps):
    """The calls of make_fn() (made on the capturing stream) captured once in a HIP graph and
    replayed: what the wave costs on the GPU when the host does not pay a library call per
    launch.  None when capture is refused."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn = make_fn()
            fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                fn()
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    except Exception as e:                                  # noqa: BLE001
        print("note: graph capture failed: %s" % e, file=sys.stderr)
        return None


def end_to_end(step_obj, reps=5):
    """Same step, but starting and ending in (pinned) host memory: H2D + kernels + D2H."""
    s = step_obj
    host_in = torch.empty(s.src.arena.numel(), dtype=torch.uint8).pin_memory()
    host_in.copy_(s.src.arena)
    host_out = torch.empty(s.dst.arena.numel(), dtype=torch.uint8).pin_memory()   # incl. <1e-3 slot padding
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        s.src.arena.copy_(host_in, non_blocking=True)
        s.step()
        host_out.copy_(s.dst.arena, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return s.bytes_step / best / 1e9, best


def end_to_end_pipelined(codec, n, m, slices=16, nbuf=3, reps=5):
    """The same host-to-host step as a three-stage pipeline over packet ranges: one stream uploads
    range k+1 while a second runs the kernels of range k and a third downloads range k-1 (PCIe is
    full duplex and the copy engines are separate from the CUs).  Packets are independent, so a
    range is just a smaller wave; `nbuf` device buffers rotate through the stages."""
    dev = codec.device
    ps = n // slices
    fmt = W.Format("%af")
    stride = 8 + 4 * m
    full = make_be_wave(n, m, 3, dev)
    host_in = torch.empty(n * stride, dtype=torch.uint8).pin_memory()
    host_in.copy_(full.arena)
    del full
    align, phase = 128, codec.payload_phase(fmt, 128)
    slot = (phase + stride + align - 1) // align * align
    host_out = torch.empty(n * slot, dtype=torch.uint8).pin_memory()
    s_in, s_run, s_out = (torch.cuda.Stream(device=dev) for _ in range(3))
    bufs = []
    for k in range(nbuf):
        bufs.append({
            "src": W.WireWave(torch.empty(ps * stride, dtype=torch.uint8, device=dev), ps,
                              torch.arange(ps + 1, dtype=torch.int64, device=dev) * stride,
                              torch.full((ps,), stride, dtype=torch.int64, device=dev), 0),
            "cols": [W.Column(torch.empty(ps * m, dtype=torch.int32, device=dev),
                              torch.empty(ps, dtype=torch.int64, device=dev),
                              torch.empty(ps + 1, dtype=torch.int64, device=dev), capacity=ps * m)],
            "ok": torch.empty(ps, dtype=torch.int32, device=dev),
            "lens": torch.empty(ps, dtype=torch.int64, device=dev),
            "offs": torch.empty(ps + 1, dtype=torch.int64, device=dev),
            "dst": torch.zeros(ps * slot, dtype=torch.uint8, device=dev),
            "uploaded": torch.cuda.Event(), "computed": torch.cuda.Event(), "downloaded": torch.cuda.Event(),
        })

    def run():
        for s in range(slices):
            b = bufs[s % nbuf]
            with torch.cuda.stream(s_in):
                if s >= nbuf:
                    s_in.wait_event(b["computed"])          # the kernels of the range that used this buffer
                b["src"].arena.copy_(host_in[s * ps * stride:(s + 1) * ps * stride], non_blocking=True)
                b["uploaded"].record(s_in)
            with torch.cuda.stream(s_run):
                s_run.wait_event(b["uploaded"])
                if s >= nbuf:
                    s_run.wait_event(b["downloaded"])       # its output buffer has left
                codec.decode_into(fmt, b["src"], b["cols"], b["ok"], W.BIGENDIAN)
                codec.sizeof(fmt, b["cols"], ps, align, phase, want_total=False, lens=b["lens"], offsets=b["offs"])
                codec.encode_into(fmt, b["cols"], W.WireWave(b["dst"], ps, b["offs"], b["lens"], 0))
                b["computed"].record(s_run)
            with torch.cuda.stream(s_out):
                s_out.wait_event(b["computed"])
                host_out[s * ps * slot:(s + 1) * ps * slot].copy_(b["dst"], non_blocking=True)
                b["downloaded"].record(s_out)
        torch.cuda.synchronize()

    run()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # the little-endian re-encoding of what came in big-endian, range by range
    pick = slice(None, None, max(1, n // 64))
    got = host_out.view(n, slot)[:, phase + 8:phase + stride][pick]
    want = host_in.view(n, stride)[:, 8:][pick].reshape(-1, m, 4).flip(-1).reshape(-1, 4 * m)
    assert torch.equal(got, want), "pipelined end-to-end output differs"
    data, payload = n * stride, n * 4 * m
    return ((data + payload) + (payload + data + 8 * (n + 1))) / best / 1e9, best


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this same
    command (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would), hand
    rank 0's output through and return the worst exit code.  Runs before this process has touched
    the GPU in any way, and the ranks are children, never a replacement of this process."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, live, failed_at = 0, list(procs), None
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc, failed_at = code, time.monotonic()
        # a rank that is gone leaves the others in a barrier: they get a little time to fail by
        # themselves (and say why), then they are ended
        if failed_at is not None and time.monotonic() - failed_at > 20.0:
            for q in live:
                q.terminate()
            failed_at = time.monotonic() + 1e9
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    if a.one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the backends' own connection messages (gloo prints one per rank) go to stderr: stdout carries
        # the result line and nothing else
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(a.backend)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    if a.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)

    codec = W.Codec(device=local, max_packets=max(a.packets, 1 << 18))
    st = ArrayWaveStep(codec, a.packets, a.elems, seed=3 + rank, overlap=a.overlap)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        st.step()
    barrier()
    st.verify()
    barrier()

    # timed region: exactly K steps.  The library brackets every payload-kernel launch
    # with HIP events on the stream it launches on (mrpdr_ctx_perf_*), which gives the
    # per-launch duration of the dominant kernel from inside this same region.
    # every 5th step's payload kernels are bracketed (an event pair perturbs the stream by a few
    # microseconds; bracketing every launch costs 3 % of the step)
    codec.perf_enable(a.perf_every)
    if st.enc is not codec:
        st.enc.perf_enable(a.perf_every)
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        st.step()
    barrier()
    dt = time.perf_counter() - t0
    perf = codec.perf_collect()
    codec.perf_enable(False)
    if st.enc is not codec:
        perf["encode"] = st.enc.perf_collect()["encode"]
        st.enc.perf_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dec_ms = perf["decode"][0] / max(1, perf["decode"][1])
    enc_ms = perf["encode"][0] / max(1, perf["encode"][1])

    total_bytes = st.bytes_step * world * a.steps
    value = total_bytes / dt / 1e9

    out = None
    if rank == 0:
        # what a plain 128-bit read+write stream sustains on this device, same byte count
        plain_ms = codec.plain_copy_ms(st.dst.arena[:a.packets * a.elems * 4], st.cols[0].values, 20)
        plain = st.bytes_copy_kernel / (plain_ms * 1e-3) / 1e9
        achieved = st.bytes_copy_kernel / (dec_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters: not collected by this run (counters need
        # rocprofv3 passes of their own); the figure of the committed counter run of this same
        # command is carried along and labelled as such
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and (a.packets, a.elems) == (4096, 65536):
            try:
                traffic = json.load(open(tpath)).get("k_copy_decode_swap4_hbm_bytes_per_launch")
                traffic_source = "profiles/traffic.json (rocprofv3 --pmc passes of this command, not this run)"
            except Exception:
                traffic = None
        out = {
            "metric": "device-resident PDR decode+encode GB/s per wave",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C3: %d packets x '%%af'[%d] per GPU from a big-endian sender; "
                                   "decode -> sizeof -> encode little-endian, device-resident"
                                   % (a.packets, a.elems),
                       "packets_per_gpu": a.packets, "elems_per_packet": a.elems,
                       "streams": 2 if a.overlap else 1,
                       "algorithmic_bytes_per_step_per_gpu": st.bytes_step,
                       "pct_of_hbm_peak": round(100.0 * value / world / HBM_PEAK_GBPS, 2)},
            "roofline": {"bound": "hbm", "kernel": "k_copy<4,swap,decode>",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "bytes_per_launch": st.bytes_copy_kernel,
                         "avg_launch_ms": round(dec_ms, 4),
                         "encode_kernel": {"kernel": "k_copy<4,noswap,encode>", "avg_launch_ms": round(enc_ms, 4),
                                           "achieved": round(st.bytes_copy_kernel / (enc_ms * 1e-3) / 1e9, 1)},
                         "launches_timed": int(perf["decode"][1]),
                         "plain_copy": {"achieved": round(plain, 1), "avg_launch_ms": round(plain_ms, 4),
                                        "frac_of_peak": round(plain / HBM_PEAK_GBPS, 4),
                                        "kernel_frac_of_plain_copy": round(achieved / plain, 4)}},
        }
    # extras and the CPU leg run after the timed region, rank 0 at N=1 only
    if rank == 0 and world == 1:
        if not a.no_extras:
            extra = small_wave_extras(codec)
            gbps, sec = end_to_end(st)
            extra["e2e_gbps_pinned_h2d_kernels_d2h"] = round(gbps, 2)
            extra["e2e_ms_per_wave"] = round(sec * 1e3, 3)
            del gbps, sec
            if (a.packets, a.elems) == (4096, 65536):
                gp, sp = end_to_end_pipelined(codec, a.packets, a.elems)
                extra["e2e_pipelined_gbps"] = round(gp, 2)
                extra["e2e_pipelined_ms_per_wave"] = round(sp * 1e3, 3)
            if (a.packets, a.elems) == (4096, 65536):
                # the same gibibyte as a 64K-packet wave: 65,536 packets x '%af'[4096]
                st64 = ArrayWaveStep(codec, 65536, 4096, seed=11)
                for _ in range(3):
                    st64.step()
                torch.cuda.synchronize()
                st64.verify()
                sec64 = time_call(st64.step, 20)
                extra["c3_64k_packets_gbps"] = round(st64.bytes_step / sec64 / 1e9, 1)
                extra["c3_64k_packets_ms_per_wave"] = round(sec64 * 1e3, 4)
                del st64
            out["extra"] = extra
        if not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(min(a.cpu_packets, a.packets), a.elems, a.cpu_reps)
            h2h = host_to_host_cpu()
            out["cpu_baseline"]["fold_only_us"] = h2h.pop("fold_only_us")
            out["cpu_baseline"]["fold_only_avg_f64_us"] = h2h.pop("fold_only_avg_f64_us")
            out["cpu_baseline"]["fold_only_tight_loop_us"] = h2h.pop("fold_only_tight_loop_us")
            out["cpu_baseline"]["fold_only_tight_loop_f64_us"] = h2h.pop("fold_only_tight_loop_f64_us")
            out["cpu_baseline"]["host_to_host_us"] = h2h
    if world > 1:
        # (a failure here -- the same on every rank, the inputs are seeded -- must not cost the
        # headline line that is already measured: it is reported in the line instead; and a rank that
        # waits in a collective for one that failed alone must not cost it either: after
        # --exchange-timeout seconds every rank gives up, rank 0 prints the line as it stands)
        printed = threading.Lock()

        def give_up():
            if not printed.acquire(blocking=False):
                return
            print("rank %d: the sharded filter leg did not finish within %d s" % (rank, a.exchange_timeout), file=sys.stderr)
            if rank == 0:
                out.pop("extra", None)
                out["extra_error"] = "the sharded filter leg did not finish within %d s" % a.exchange_timeout
                out.setdefault("cpu_baseline", None)
                print(json.dumps(out), flush=True)
            os._exit(0)
        watchdog = threading.Timer(a.exchange_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            # the one exchange step of the path: a '%lf %d' wave of world x 131,072
            # packets, fused Avg per shard, results all-gathered (RCCL) and folded in rank order
            from mrnet_amd import dist as D
            n5 = 131072
            fmt5 = W.Format("%lf %d")
            g5 = torch.Generator(device="cuda").manual_seed(5 + rank)
            v = torch.randn(n5, dtype=torch.float64, device="cuda", generator=g5) * 1e-4 + 1e-3
            rows = torch.empty((n5, 12), dtype=torch.uint8, device="cuda")
            rows[:, :8] = v.view(torch.uint8).view(n5, 8)
            rows[:, 8:] = torch.ones(n5, dtype=torch.int32, device="cuda").view(torch.uint8).view(n5, 4)
            w5 = W.WireWave(rows.view(-1), n5, None, None, 12)
            def timed(fn, reps=20):
                for _ in range(3):
                    r = fn()
                barrier()
                t = time.perf_counter()
                for _ in range(reps):
                    r = fn()
                barrier()
                return (time.perf_counter() - t) / reps, r
            # ONE reference filter call over the whole wave (the default): nodes exchanged, the chain of
            # a double Avg relayed in rank order -- and the declared-tree reading (ranks = one more level)
            t5, (res5, ok5) = timed(lambda: D.filter_wire_sharded(codec, W.TFILTER_AVG, fmt5, w5))
            t5t, (res5t, ok5t) = timed(lambda: D.filter_wire_sharded(codec, W.TFILTER_AVG, fmt5, w5, shape=W.SHAPE_TREE))
            cnt5 = int(res5[8:12].view(torch.int32).item())
            assert int(ok5.item()) == 1 and cnt5 == n5 * world, "sharded Avg lost packets"
            assert int(res5t[8:12].view(torch.int32).item()) == cnt5
            # the configuration as it is named: Min, Max and Avg, one exchange for what can be gathered
            t5s, (st5, oks5) = timed(lambda: D.stats_wire_sharded(codec, fmt5, w5))
            t5st, (st5t, oks5t) = timed(lambda: D.stats_wire_sharded(codec, fmt5, w5, shape=W.SHAPE_TREE))
            assert bool(oks5.all()) and torch.equal(st5[32:48], res5), "sharded Min/Max/Avg differs from the sharded Avg"
            assert bool(oks5t.all()) and torch.equal(st5t[32:48], res5t) and torch.equal(st5t[:32], st5[:32])
            # the same for the performance-data shape: 131,072 x '%ad %ad %alf'[8] per rank, Avg, per-rank
            # output packets gathered (80 bytes each) and folded once more
            k5 = 8
            fmtp = W.Format("%ad %ad %alf")
            onesp = np.ones(n5, np.uint64)
            colsp = codec.columns_from_numpy(fmtp, [
                (onesp, (np.arange(n5, dtype=np.int32) + rank * n5)), (onesp, np.full(n5, k5, np.int32)),
                (np.full(n5, k5, np.uint64), np.random.default_rng(50 + rank).normal(1e-3, 1e-4, n5 * k5))])
            t5b, (pd5, pl5, pr5) = timed(lambda: D.perfdata_sharded(codec, W.TFILTER_AVG, W.DataType.DOUBLE_T, colsp[0], colsp[2],
                                                                    n5, k5, fanin=256, shape=W.SHAPE_TREE))
            assert int(pl5.item()) == k5 and int(pr5.item()) == -n5 * world, "sharded PerfData lost ranks"
            # one invocation over the whole wave: int64 states gathered (order-free), double states relayed
            icol = W.Column((colsp[2].values * 1e9).to(torch.int64), colsp[2].counts, colsp[2].offs, n5 * k5)
            t5bi, (pdi, pli, pri) = timed(lambda: D.perfdata_sharded(codec, W.TFILTER_AVG, W.DataType.INT64_T, colsp[0], icol, n5, k5))
            assert int(pli.item()) == k5 and int(pri.item()) == -n5 * world
            t5bf, (pdf, plf, prf) = timed(lambda: D.perfdata_sharded(codec, W.TFILTER_AVG, W.DataType.DOUBLE_T, colsp[0], colsp[2], n5, k5), reps=3)
            assert int(plf.item()) == k5 and int(prf.item()) == -n5 * world
            if rank == 0:
                out["extra"] = {"c5_sharded_avg_us_per_wave": round(t5 * 1e6, 1), "c5_packets": n5 * world,
                                "c5_sharded_avg_tree_us_per_wave": round(t5t * 1e6, 1),
                                "c5_sharded_min_max_avg_us_per_wave": round(t5s * 1e6, 1),
                                "c5_sharded_min_max_avg_tree_us_per_wave": round(t5st * 1e6, 1),
                                "c5_avg": float(res5[:8].view(torch.float64).item()),
                                "c5b_sharded_perfdata_avg_tree_us_per_wave": round(t5b * 1e6, 1),
                                "c5b_sharded_perfdata_avg_i64_one_call_us_per_wave": round(t5bi * 1e6, 1),
                                "c5b_sharded_perfdata_avg_one_call_us_per_wave": round(t5bf * 1e6, 1)}
        except Exception as e:                          # noqa: BLE001
            print("rank %d: sharded filter leg failed: %r" % (rank, e), file=sys.stderr)
            if rank == 0:
                out["extra_error"] = repr(e)
        watchdog.cancel()
        if not printed.acquire(blocking=False):         # the watchdog is printing: it ends the process
            time.sleep(60)
    if rank == 0:
        out.setdefault("cpu_baseline", None)
        print(json.dumps(out), flush=True)
    if world > 1:
        # (the line is out: a rank that is gone must not keep the others in the last barrier)
        bye = threading.Timer(60, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()


if __name__ == "__main__":
    main()
