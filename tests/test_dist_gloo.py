"""The N > 1 path on CPU: gloo ranks shard a wave by packet range and exchange 16-byte nodes (or, in
the declared-tree reading, results).  No GPU here, so what a rank computes from its packets is done
by a host model built on the oracle (the checker); what is under test is the sharding rule, the
exchange of mrnet_amd/dist.py itself (gather order and packing, the relay of the floating sums in
rank order, the broadcast) and that the outcome equals ONE flat fold over the whole wave."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wavegen import O, ROOT, random_values, splitmix64

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


class HostNodes:
    """What Codec.filter_partial_wire / filter_combine / is_chained compute, on the host: the node
    rules of include/mrpdr.h (struct mrpdr_node) written with numpy and the oracle's flat fold, so
    that dist.filter_wire_sharded runs unchanged without a device."""

    def is_chained(self, op, t):
        return op in (O.SUM, O.AVG) and t in (O.FLOAT_T, O.DOUBLE_T)

    def empty(self, n, dtype):
        return torch.zeros(n, dtype=dtype)

    @staticmethod
    def _node(bits, count, present):
        raw = np.zeros(16, np.uint8)
        if present:
            b = np.asarray(bits).tobytes()
            raw[:len(b)] = np.frombuffer(b, np.uint8)
            raw[8:12] = np.frombuffer(np.int32(count).tobytes(), np.uint8)
            raw[12:16] = np.frombuffer(np.int32(1).tobytes(), np.uint8)
        return torch.from_numpy(raw)

    def filter_partial_wire(self, op, fmt, wire, sender_bo=0, front=None):
        t, dt = fmt.types[0], O.elem_dtype(fmt.types[0])
        vals, cnt = wire
        f = None if front is None else front.numpy()
        f_present = f is not None and int(f[12:16].view(np.int32)[0]) != 0
        if self.is_chained(op, t):
            x = vals if op == O.SUM else (vals * cnt.astype(dt)).astype(dt)
            if f_present:
                x = np.concatenate([f[:dt.itemsize].view(dt), x])          # 0 + s == s: the chain goes on from s
            total = int(cnt.sum()) if op == O.AVG else 0
            if f_present:
                total += int(f[8:12].view(np.int32)[0])
            if x.size == 0:
                return self._node(0, 0, False)
            return self._node(O.filter_flat(O.SUM, t, x)[0], total, True)
        assert f is None
        if vals.size == 0:
            return self._node(0, 0, False)
        if op == O.AVG:                                                   # int32 here: the undivided wrapping sum
            assert dt == np.int32
            prod = (vals.astype(np.int64) * cnt.astype(np.int64)) & 0xFFFFFFFF
            return self._node(np.array(int(prod.sum()) & 0xFFFFFFFF, np.uint64).astype(np.uint32), int(cnt.sum()), True)
        r = O.filter_flat(op, t, vals)[0]
        mark = int(dt.kind == "f" and op in (O.MIN, O.MAX) and bool(np.isnan(vals).any()))
        return self._node(r, mark, True)

    def filter_combine(self, op, t, nodes):
        dt = O.elem_dtype(t)
        rows = nodes.numpy().reshape(-1, 16)
        rows = [r for r in rows if int(r[12:16].view(np.int32)[0])]
        res, ok = np.zeros(16, np.uint8), np.zeros(1, np.int32)
        if not rows:
            return torch.from_numpy(res), torch.from_numpy(ok)
        val = [r[:dt.itemsize].view(dt)[0] for r in rows]
        cnt = [int(r[8:12].view(np.int32)[0]) for r in rows]
        out_c = None
        if self.is_chained(op, t):
            out = val[-1]
            if op == O.AVG:
                out_c = cnt[-1]
                if out_c == 0:
                    return torch.from_numpy(res), torch.from_numpy(ok)
                out = dt.type(out / dt.type(out_c))
        elif op == O.AVG:
            out_c = sum(cnt)
            if out_c == 0:
                return torch.from_numpy(res), torch.from_numpy(ok)
            tot = np.array(sum(int(v) & 0xFFFFFFFF for v in val) & 0xFFFFFFFF, np.uint64).astype(np.uint32).view(np.int32)
            out = np.int32(int(np.trunc(int(tot) / out_c)))                  # C division truncates toward zero
        elif dt.kind == "f" and op in (O.MIN, O.MAX):
            out = val[0]
            for v, c in zip(val[1:], cnt[1:]):
                keep = (out < v) if op == O.MIN else (out > v)
                out = v if (c or not keep) else out
        else:
            out = O.filter_flat(op, t, np.array(val, dt))[0]
        b = np.asarray(out, dt).tobytes()
        res[:len(b)] = np.frombuffer(b, np.uint8)
        if out_c is not None:
            res[dt.itemsize:dt.itemsize + 4] = np.frombuffer(np.int32(out_c).tobytes(), np.uint8)
        ok[0] = 1
        return torch.from_numpy(res), torch.from_numpy(ok)


class _Fmt:
    def __init__(self, t):
        self.types, self.widths = [t], [O.elem_dtype(t).itemsize]


EXACT_CASES = [(O.INT32_T, O.AVG), (O.INT32_T, O.SUM), (O.FLOAT_T, O.SUM), (O.DOUBLE_T, O.AVG), (O.FLOAT_T, O.AVG),
               (O.FLOAT_T, O.MIN), (O.DOUBLE_T, O.MAX), (O.UINT64_T, O.MIN)]


def _exact_wave(t, op, n):
    dt = O.elem_dtype(t)
    u = splitmix64(1234 + t, n)
    if dt.kind == "f":
        vals = ((u % np.uint64(1_000_000)).astype(np.float64) / 1e3 - 100.0).astype(dt)
        if op in (O.MIN, O.MAX):
            vals[[n // 3 - 1, n // 2, n - 7]] = np.nan
    else:
        vals = (u % np.uint64(1000)).astype(dt)
    cnt = (splitmix64(5, n) % np.uint64(4) + np.uint64(1)).astype(np.int32) if op == O.AVG else None
    return vals, cnt


def _exact_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, out = HostNodes(), []
        lo, hi = D.shard_range(n, rank, world)
        for t, op in EXACT_CASES:
            vals, cnt = _exact_wave(t, op, n)
            shard = (vals[lo:hi], None if cnt is None else cnt[lo:hi])
            res, ok = D.filter_wire_sharded(model, op, _Fmt(t), shard)
            out.append((res.numpy().copy(), int(ok[0])))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_return_one_flat_call(world):
    """dist.filter_wire_sharded over gloo: gathered nodes for the order-free operators, the relayed
    chain for the floating sums; every rank ends with the bits of ONE oracle fold over the whole wave
    (integer Avg divides once; a NaN at a range's end restarts the comparison in the next range)."""
    n = 30011
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exact_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, (t, op) in enumerate(EXACT_CASES):
        vals, cnt = _exact_wave(t, op, n)
        want = O.filter_flat(op, t, vals, cnt)
        dt = O.elem_dtype(t)
        for rank, out in got:
            res, ok = out[k]
            assert ok == 1
            assert res[:dt.itemsize].tobytes() == np.asarray(want[0]).tobytes(), (t, op, rank)
            if op == O.AVG:
                assert int(res[dt.itemsize:dt.itemsize + 4].view(np.int32)[0]) == want[1]
    # the per-range results folded once more are NOT that (why nodes, not results, are exchanged)
    vals, cnt = _exact_wave(O.INT32_T, O.AVG, n)
    parts = [O.filter_flat(O.AVG, O.INT32_T, vals[lo:hi], cnt[lo:hi]) for lo, hi in (D.shard_range(n, r, world) for r in range(world))]
    levels = O.filter_flat(O.AVG, O.INT32_T, np.array([p[0] for p in parts]), np.array([p[1] for p in parts], np.int32))
    assert levels[1] == O.filter_flat(O.AVG, O.INT32_T, vals, cnt)[1]


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
