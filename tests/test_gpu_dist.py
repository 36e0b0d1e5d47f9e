"""Two ranks (gloo rendezvous on 127.0.0.1) sharing one GPU: the N > 1 path with the device code in
it.  Every rank reduces its packet range on the GPU to a node, the nodes are exchanged (gathered, or
relayed in rank order for the floating sums) and combined; the outcome must equal ONE flat fold of
the oracle over the WHOLE wave, bit for bit (the reference hands a node's whole wave to one filter
call, src/Filter.C:42-117).  The declared-tree reading (shape=SHAPE_TREE, the ranks as a tree level)
is checked against the oracle's tree.  And RCCL itself: a one-rank `nccl` group in a fresh process
runs the same calls with the collectives on device memory."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wavegen import O, random_values, splitmix64
from test_dist_gloo import _free_port, _perf_wave

pytestmark = pytest.mark.gpu

from mrnet_amd import dist as D  # noqa: E402
from mrnet_amd import wave as W  # noqa: E402

N, K = 20000, 8


def _scalar_wave(n):
    vals = random_values(np.float64, n, 5) * 1e-300
    vals = np.where(np.isfinite(vals), vals, 1.0)
    cnts = (np.arange(n) % 3 + 1).astype(np.int32)
    return vals, cnts


CASES = [  # (format, oracle type, numpy dtype, operator)
    ("%d %d", O.INT32_T, np.int32, O.AVG), ("%hd %d", O.INT16_T, np.int16, O.AVG), ("%uld %d", O.UINT64_T, np.uint64, O.AVG),
    ("%f", O.FLOAT_T, np.float32, O.SUM), ("%lf %d", O.DOUBLE_T, np.float64, O.AVG), ("%f", O.FLOAT_T, np.float32, O.MIN),
    ("%lf", O.DOUBLE_T, np.float64, O.MAX), ("%d", O.INT32_T, np.int32, O.SUM)]
NC = 1 << 17


def _case_wave(dt, op, n):
    u = splitmix64(99, n)
    if np.dtype(dt).kind == "f":
        vals = ((u % np.uint64(1_000_000)).astype(np.float64) / 1e3 - 20.0).astype(dt)
        if op in (O.MIN, O.MAX):
            vals[n // 2 - 1] = np.nan            # the last packet of rank 0's range
            vals[n // 3] = np.nan
    else:
        vals = (u % np.uint64(1000)).astype(dt)
    cnt = (splitmix64(7, n) % np.uint64(4) + np.uint64(1)).astype(np.int32) if op == O.AVG else None
    return vals, cnt


def _case_rows(vals, cnt, lo, hi):
    w = vals.dtype.itemsize
    rows = np.zeros((hi - lo, w + (4 if cnt is not None else 0)), np.uint8)
    rows[:, :w] = vals[lo:hi].view(np.uint8).reshape(-1, w)
    if cnt is not None:
        rows[:, w:] = cnt[lo:hi].view(np.uint8).reshape(-1, 4)
    return rows


def _exact_cases(codec, rank, world):
    """Every case of CASES through filter_wire_sharded (one reference call over the whole wave)."""
    out = []
    for fmt_s, t, dt, op in CASES:
        vals, cnt = _case_wave(dt, op, NC)
        lo, hi = D.shard_range(NC, rank, world)
        rows = _case_rows(vals, cnt, lo, hi)
        wire = W.WireWave(codec.to_device(rows.reshape(-1)), hi - lo, None, None, rows.shape[1])
        res, ok = D.filter_wire_sharded(codec, op, W.Format(fmt_s), wire)
        out.append((res.cpu().numpy().copy(), int(ok.item())))
    return out


def _check_exact_cases(got):
    for (fmt_s, t, dt, op), (res, ok) in zip(CASES, got):
        vals, cnt = _case_wave(dt, op, NC)
        want = O.filter_flat(op, t, vals, cnt)             # ONE fold over the whole wave
        assert ok == 1, fmt_s
        w = np.dtype(dt).itemsize
        assert res[:w].tobytes() == np.asarray(want[0]).tobytes(), (fmt_s, op, res[:w].view(dt), want)
        if op == O.AVG:
            assert int(res[w:w + 4].view(np.int32)[0]) == want[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        codec = W.Codec(device=0, max_packets=1 << 17)
        exact = _exact_cases(codec, rank, world)
        lo, hi = D.shard_range(N, rank, world)
        vals, cnts = _scalar_wave(N)
        rows = np.zeros((hi - lo, 12), np.uint8)
        rows[:, :8] = vals[lo:hi].view(np.uint8).reshape(-1, 8)
        rows[:, 8:] = cnts[lo:hi].view(np.uint8).reshape(-1, 4)
        wire = W.WireWave(codec.to_device(rows.reshape(-1)), hi - lo, None, None, 12)
        res, ok = D.filter_wire_sharded(codec, W.TFILTER_AVG, W.Format("%lf %d"), wire, shape=W.SHAPE_TREE)
        # Min, Max and Avg with one exchange: the same bytes as three sharded filters, in both readings
        for shape in (W.SHAPE_TREE, W.SHAPE_FLAT):
            st3, ok3 = D.stats_wire_sharded(codec, W.Format("%lf %d"), wire, shape=shape)
            for k, op in enumerate((W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG)):
                r1, o1 = D.filter_wire_sharded(codec, op, W.Format("%lf %d") if op == W.TFILTER_AVG else W.Format("%lf"),
                                               wire if op == W.TFILTER_AVG else
                                               W.WireWave(codec.to_device(np.ascontiguousarray(rows[:, :8]).reshape(-1)),
                                                          hi - lo, None, None, 8), shape=shape)
                assert int(o1.item()) == int(ok3[k].item()) == 1
                assert torch.equal(r1, st3[16 * k:16 * k + 16]), "fused sharded stats differ from the sharded filter %d" % op
        flat_avg = st3[32:48].cpu().numpy().copy()
        rc, rv, dc, dv = _perf_wave(N, K)
        fmt = W.Format("%ad %ad %alf")
        cols = codec.columns_from_numpy(fmt, [(rc[lo:hi], rv[lo:hi]), (rc[lo:hi], np.full(hi - lo, K, np.int32)),
                                              (dc[lo:hi], dv[lo * K:hi * K])])
        pd, plen, prank = D.perfdata_sharded(codec, W.TFILTER_AVG, W.DataType.DOUBLE_T, cols[0], cols[2], hi - lo, K,
                                             shape=W.SHAPE_TREE)
        # one invocation over the whole wave: the double states relayed, the integer states gathered
        pdf, plf, prf = D.perfdata_sharded(codec, W.TFILTER_AVG, W.DataType.DOUBLE_T, cols[0], cols[2], hi - lo, K)
        icols = codec.columns_from_numpy(W.Format("%ad %ad %ald"), [
            (rc[lo:hi], rv[lo:hi]), (rc[lo:hi], np.full(hi - lo, K, np.int32)),
            (dc[lo:hi], (dv[lo * K:hi * K] * 1000).astype(np.int64))])
        pdi, pli, pri = D.perfdata_sharded(codec, W.TFILTER_AVG, W.DataType.INT64_T, icols[0], icols[2], hi - lo, K)
        torch.cuda.synchronize()
        flat_perf = (pdf.cpu().numpy().copy(), int(plf.item()), int(prf.item()), pdi.cpu().numpy().copy(), int(pli.item()),
                     int(pri.item()))
        q.put((rank, res.cpu().numpy().copy(), int(ok.item()), pd.cpu().numpy().copy(), int(plen.item()),
               int(prank.item()), codec.status(), exact, flat_avg, flat_perf))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals, cnts = _scalar_wave(N)
    parts = [O.filter_tree(O.AVG, O.DOUBLE_T, vals[lo:hi], cnts[lo:hi], W.TREE_FANINS)
             for lo, hi in (D.shard_range(N, r, world) for r in range(world))]
    want = O.filter_flat(O.AVG, O.DOUBLE_T, np.array([p[0] for p in parts]), np.array([p[1] for p in parts], np.int32))
    one_call = O.filter_flat(O.AVG, O.DOUBLE_T, vals, cnts)
    rc, rv, dc, dv = _perf_wave(N, K)
    tree = O.filter_perfdata_tree(O.AVG, O.DOUBLE_T, (rc, rv), (dc, dv), [N // 2, 0])
    one_perf = O.filter_perfdata(O.AVG, O.DOUBLE_T, (rc, rv), (dc, dv))
    one_perf_i = O.filter_perfdata(O.AVG, O.INT64_T, (rc, rv), (dc, (dv * 1000).astype(np.int64)))
    for rank, res, ok, pd, plen, prank, status, exact, flat_avg, flat_perf in got:
        pdf, plf, prf, pdi, pli, pri = flat_perf
        assert plf == pli == K and prf == pri == one_perf[0] == -N
        assert pdf.view(np.float64)[:K].tobytes() == one_perf[1].tobytes()
        assert pdi.view(np.int64)[:K].tobytes() == one_perf_i[1].tobytes()
        assert ok == 1 and status == 0
        # the declared tree with the ranks as its last level
        assert res[:8].tobytes() == np.float64(want[0]).tobytes()
        assert int(res[8:12].view(np.int32)[0]) == int(want[1]) == int(cnts.sum())
        # one reference call over the whole wave
        assert flat_avg[:8].tobytes() == np.float64(one_call[0]).tobytes()
        _check_exact_cases(exact)
        assert plen == K and prank == tree[0] == -N
        assert pd.view(np.float64)[:K].tobytes() == tree[1].tobytes()


def _rccl_one_rank(port, q):
    """A process group of ONE rank over RCCL: the collectives of the sharded filter (all_gather,
    broadcast) run on device memory through librccl."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        codec = W.Codec(device=0, max_packets=1 << 17)
        exact = _exact_cases(codec, 0, 1)
        vals, cnts = _scalar_wave(N)
        rows = _case_rows(vals, cnts, 0, N)
        wire = W.WireWave(codec.to_device(rows.reshape(-1)), N, None, None, 12)
        st3, ok3 = D.stats_wire_sharded(codec, W.Format("%lf %d"), wire)
        single, sok = codec.filter_wire(W.TFILTER_AVG, W.Format("%lf %d"), wire)
        torch.cuda.synchronize()
        loaded = any("librccl" in line for line in open("/proc/self/maps"))
        q.put((exact, st3.cpu().numpy().copy(), ok3.cpu().numpy().copy(), single.cpu().numpy().copy(), loaded,
               dist.get_backend()))
    finally:
        dist.destroy_process_group()


def test_one_rank_over_rccl():
    """RCCL executed: a one-rank nccl group in a fresh child process pushes a shard through
    filter_wire_sharded and stats_wire_sharded; librccl is mapped, the results equal the oracle's flat
    fold and the single-GPU call."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank, args=(_free_port(), q))
    p.start()
    exact, st3, ok3, single, loaded, backend = q.get(timeout=200)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert backend == "nccl" and loaded, "librccl was not loaded"
    _check_exact_cases(exact)
    assert ok3.tolist() == [1, 1, 1] and st3[32:48].tobytes() == single.tobytes()


def _rccl_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        codec = W.Codec(device=rank, max_packets=1 << 17)
        exact = _exact_cases(codec, rank, world)                    # gathered nodes and the relayed chains
        lo, hi = D.shard_range(N, rank, world)
        vals, cnts = _scalar_wave(N)
        rows = np.zeros((hi - lo, 12), np.uint8)
        rows[:, :8] = vals[lo:hi].view(np.uint8).reshape(-1, 8)
        rows[:, 8:] = cnts[lo:hi].view(np.uint8).reshape(-1, 4)
        wire = W.WireWave(codec.to_device(rows.reshape(-1)), hi - lo, None, None, 12)
        res, ok = D.filter_wire_sharded(codec, W.TFILTER_AVG, W.Format("%lf %d"), wire)    # send / recv / broadcast over RCCL
        torch.cuda.synchronize()
        q.put((rank, res.cpu().numpy().copy(), int(ok.item()), exact))
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_rccl():
    """The same exchange with the real backend: one rank per GPU, the 16-byte nodes gathered or
    relayed by RCCL on the device.  Needs two GPUs (the single-GPU box skips it; the two-ranks-one-GPU test
    above covers the path with gloo)."""
    world = 2
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=200) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals, cnts = _scalar_wave(N)
    want = O.filter_flat(O.AVG, O.DOUBLE_T, vals, cnts)        # one reference call over the whole wave
    for rank, res, ok, exact in got:
        assert ok == 1 and res[:8].tobytes() == np.float64(want[0]).tobytes()
        assert int(res[8:12].view(np.int32)[0]) == int(want[1])
        _check_exact_cases(exact)
