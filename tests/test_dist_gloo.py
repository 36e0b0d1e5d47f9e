"""The N > 1 path on CPU: two gloo ranks shard a wave by packet range, exchange the 16-byte
filter results and fold them in rank order.  No GPU here, so the per-shard reduction and the
final fold are done by the oracle (the checker); what is under test is the sharding rule, the
gather (order, packing) and that the ordered fold equals the declared tree over the whole wave."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wavegen import O, ROOT, random_values

from mrnet_amd import dist as D
from mrnet_amd import wave as W


def test_shard_ranges_partition_the_wave():
    for n in (0, 1, 7, 4096, 1048576, 1000003):
        for world in (1, 2, 3, 8):
            edges = [D.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vals = random_values(np.float64, n, 5) * 1e-300     # finite doubles, order-sensitive sum
        vals = np.where(np.isfinite(vals), vals, 1.0)
        cnts = (np.arange(n) % 3 + 1).astype(np.int32)
        lo, hi = D.shard_range(n, rank, world)
        out = {}
        for name, op in (("sum", O.SUM), ("avg", O.AVG), ("max", O.MAX)):
            r = O.filter_tree(op, O.DOUBLE_T, vals[lo:hi], cnts[lo:hi] if op == O.AVG else None,
                              W.TREE_FANINS)
            res = np.zeros(16, np.uint8)
            ok = np.zeros(1, np.int32)
            if r is not None:
                res[:8] = np.frombuffer(np.float64(r[0]).tobytes(), np.uint8)
                res[8:12] = np.frombuffer(np.int32(r[1]).tobytes(), np.uint8)
                ok[0] = 1
            rows, oks = D.gather_results(torch.from_numpy(res), torch.from_numpy(ok))
            out[name] = (rows.numpy().copy(), oks.numpy().copy())
        q.put((rank, lo, hi, out))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gather_and_ordered_fold():
    world, n = 2, 10007
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals = random_values(np.float64, n, 5) * 1e-300
    vals = np.where(np.isfinite(vals), vals, 1.0)
    cnts = (np.arange(n) % 3 + 1).astype(np.int32)
    # every rank saw the same rows, in rank order
    for name, op in (("sum", O.SUM), ("avg", O.AVG), ("max", O.MAX)):
        rows0, oks0 = got[0][3][name]
        rows1, oks1 = got[1][3][name]
        assert np.array_equal(rows0, rows1) and np.array_equal(oks0, oks1) and oks0.tolist() == [1, 1]
        part_v = rows0[:, :8].copy().view(np.float64).reshape(-1)
        part_c = rows0[:, 8:12].copy().view(np.int32).reshape(-1)
        for r, (rank, lo, hi, _) in enumerate(got):
            want = O.filter_tree(op, O.DOUBLE_T, vals[lo:hi], cnts[lo:hi] if op == O.AVG else None,
                                 W.TREE_FANINS)
            assert part_v[r].tobytes() == np.float64(want[0]).tobytes()
        # last level: ordered fold over the ranks == the reference filter applied to the rank results
        final = O.filter_flat(op, O.DOUBLE_T, part_v, part_c if op == O.AVG else None)
        assert final is not None
        if op == O.MAX:
            assert final[0] == vals.max()


def _perf_worker(rank, world, port, n, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rc, rv, dc, dv = _perf_wave(n, k)
        lo, hi = D.shard_range(n, rank, world)
        orank, odata = O.filter_perfdata(O.AVG, O.DOUBLE_T, (rc[lo:hi], rv[lo:hi]), (dc[lo:hi], dv[lo * k:hi * k]))
        out = np.zeros(k, np.float64)
        out[:odata.size] = odata
        row = D.pack_perfdata(torch.from_numpy(out).view(torch.int64), torch.tensor([odata.size], dtype=torch.int64),
                              torch.tensor([orank], dtype=torch.int32), k)
        rows = D.gather_rows(row)
        rank_col, data_col = D.unpack_perfdata(rows, k)
        q.put((rank, rank_col.values.numpy().copy(), data_col.counts.numpy().copy(), data_col.offs.numpy().copy(),
               data_col.values.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _perf_wave(n, k):
    rc = np.ones(n, np.uint64)
    rv = np.arange(n, dtype=np.int32)
    dc = np.full(n, k, np.uint64)
    dv = (random_values(np.float64, n * k, 9) % 1000.0)
    dv = np.where(np.isfinite(dv), dv, 1.0)
    return rc, rv, dc, dv


def test_two_ranks_perfdata_exchange():
    """The per-rank PerfData packets are gathered in rank order, and folding them once more equals
    the filter run as a two-level tree over the whole wave (fan-ins n/2, then 2)."""
    world, n, k = 2, 1000, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_perf_worker, args=(r, world, port, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(got[0][1:], got[1][1:]):
        assert np.array_equal(a, b)                                   # every rank holds the same wave
    _, ranks, lens, offs, data = got[0]
    assert ranks.tolist() == [-(n // 2), -(n // 2)] and lens.tolist() == [k, k] and offs.tolist() == [0, k, 2 * k]
    rc, rv, dc, dv = _perf_wave(n, k)
    final = O.filter_perfdata(O.AVG, O.DOUBLE_T, (np.ones(2, np.uint64), ranks),
                              (lens.astype(np.uint64), data.view(np.float64)[:2 * k]))
    tree = O.filter_perfdata_tree(O.AVG, O.DOUBLE_T, (rc, rv), (dc, dv), [n // 2, 0])
    assert final[0] == tree[0] == -n and np.array_equal(final[1].view(np.uint8), tree[1].view(np.uint8))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it starts N rank processes itself
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches a GPU.  Here, without a
    GPU, every rank must come up, say that it needs a HIP device, and the parent must hand their
    failure on as its own exit code."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the GPU test of the same command")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert r.stderr.count("needs a HIP device") == 3, r.stderr[-2000:]
