"""Two ranks (gloo rendezvous on 127.0.0.1) sharing one GPU: the N > 1 path with the device code in
it.  Every rank reduces its packet range on the GPU, the results are exchanged and folded in rank
order; the outcome must equal the oracle's declared tree over the whole wave, bit for bit."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wavegen import O, random_values
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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        codec = W.Codec(device=0, max_packets=1 << 16)
        lo, hi = D.shard_range(N, rank, world)
        vals, cnts = _scalar_wave(N)
        rows = np.zeros((hi - lo, 12), np.uint8)
        rows[:, :8] = vals[lo:hi].view(np.uint8).reshape(-1, 8)
        rows[:, 8:] = cnts[lo:hi].view(np.uint8).reshape(-1, 4)
        wire = W.WireWave(codec.to_device(rows.reshape(-1)), hi - lo, None, None, 12)
        res, ok = D.filter_wire_sharded(codec, W.TFILTER_AVG, W.Format("%lf %d"), wire)
        # Min, Max and Avg with one exchange: the same bytes as three sharded filters
        st3, ok3 = D.stats_wire_sharded(codec, W.Format("%lf %d"), wire)
        for k, op in enumerate((W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG)):
            r1, o1 = D.filter_wire_sharded(codec, op, W.Format("%lf %d") if op == W.TFILTER_AVG else W.Format("%lf"),
                                           wire if op == W.TFILTER_AVG else
                                           W.WireWave(codec.to_device(np.ascontiguousarray(rows[:, :8]).reshape(-1)),
                                                      hi - lo, None, None, 8))
            assert int(o1.item()) == int(ok3[k].item()) == 1
            assert torch.equal(r1, st3[16 * k:16 * k + 16]), "fused sharded stats differ from the sharded filter %d" % op
        rc, rv, dc, dv = _perf_wave(N, K)
        fmt = W.Format("%ad %ad %alf")
        cols = codec.columns_from_numpy(fmt, [(rc[lo:hi], rv[lo:hi]), (rc[lo:hi], np.full(hi - lo, K, np.int32)),
                                              (dc[lo:hi], dv[lo * K:hi * K])])
        pd, plen, prank = D.perfdata_sharded(codec, W.TFILTER_AVG, W.DataType.DOUBLE_T, cols[0], cols[2], hi - lo, K)
        torch.cuda.synchronize()
        q.put((rank, res.cpu().numpy().copy(), int(ok.item()), pd.cpu().numpy().copy(), int(plen.item()),
               int(prank.item()), codec.status()))
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
    rc, rv, dc, dv = _perf_wave(N, K)
    tree = O.filter_perfdata_tree(O.AVG, O.DOUBLE_T, (rc, rv), (dc, dv), [N // 2, 0])
    for rank, res, ok, pd, plen, prank, status in got:
        assert ok == 1 and status == 0
        assert res[:8].tobytes() == np.float64(want[0]).tobytes()
        assert int(res[8:12].view(np.int32)[0]) == int(want[1]) == int(cnts.sum())
        assert plen == K and prank == tree[0] == -N
        assert pd.view(np.float64)[:K].tobytes() == tree[1].tobytes()


def _rccl_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        codec = W.Codec(device=rank, max_packets=1 << 16)
        lo, hi = D.shard_range(N, rank, world)
        vals, cnts = _scalar_wave(N)
        rows = np.zeros((hi - lo, 12), np.uint8)
        rows[:, :8] = vals[lo:hi].view(np.uint8).reshape(-1, 8)
        rows[:, 8:] = cnts[lo:hi].view(np.uint8).reshape(-1, 4)
        wire = W.WireWave(codec.to_device(rows.reshape(-1)), hi - lo, None, None, 12)
        res, ok = D.filter_wire_sharded(codec, W.TFILTER_AVG, W.Format("%lf %d"), wire)    # all_gather over RCCL
        torch.cuda.synchronize()
        q.put((rank, res.cpu().numpy().copy(), int(ok.item())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_rccl():
    """The same exchange with the real backend: one rank per GPU, the 20-byte results gathered by
    RCCL on the device.  Needs two GPUs (the single-GPU box skips it; the two-ranks-one-GPU test
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
    parts = [O.filter_tree(O.AVG, O.DOUBLE_T, vals[lo:hi], cnts[lo:hi], W.TREE_FANINS)
             for lo, hi in (D.shard_range(N, r, world) for r in range(world))]
    want = O.filter_flat(O.AVG, O.DOUBLE_T, np.array([p[0] for p in parts]), np.array([p[1] for p in parts], np.int32))
    for rank, res, ok in got:
        assert ok == 1 and res[:8].tobytes() == np.float64(want[0]).tobytes()
        assert int(res[8:12].view(np.int32)[0]) == int(want[1])
