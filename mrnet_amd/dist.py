"""Multi-GPU side of the wave codec: one process per GPU, torch.distributed (RCCL on GPUs).

Packets of a wave are independent, so encode/decode shard by packet range with no
collective at all.  The only exchange is the filter result.  In the reference a node's wave goes
through ONE filter call whatever its size (src/Filter.C:42-117 hands the whole vector to the filter
function; tfilter_Avg multiplies, sums and divides once, src/FilterDefinitions.C:517-561), so the
GPUs of a node are not tree levels: every rank reduces its range to a 16-byte node that holds
nothing the one call would not hold either (struct mrpdr_node: undivided sums for AVG, the NaN mark
for float MIN / MAX), and the nodes are combined in rank order --
  * integer operators, float / double MIN / MAX: one all_gather of the nodes, one combine;
  * float / double SUM / AVG, where every addition rounds: the chain of additions is relayed in
    rank order (rank r starts from the node rank r - 1 left), the last rank broadcasts.
The result is that of MRPDR_SHAPE_FLAT over the whole wave, bit for bit.  shape=SHAPE_TREE keeps
the other reading, for callers who declare the GPUs a tree level: every rank runs the declared tree
over its shard, the 16-byte RESULTS are gathered and folded once more (fan-in = number of ranks).
"""
import torch
import torch.distributed as dist

from . import wave as W


def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _rank(group=None):
    return dist.get_rank(group) if dist.is_initialized() else 0


def _global(r, group=None):
    return r if group is None else dist.get_global_rank(group, r)


def _via_host(t, group=None):
    """gloo moves host memory: the CPU rehearsal of the exchange.  RCCL moves device memory."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def shard_range(n, rank, world):
    """Contiguous packet range [lo, hi) of `rank`: [rank*n/world, (rank+1)*n/world)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def gather_results(result, result_ok, group=None):
    """all_gather the per-rank filter results.  result: uint8[16], result_ok: int32[1] (on the
    device with RCCL, on the host with gloo).  Returns (uint8[G,16], int32[G]) in rank order."""
    packed = torch.zeros(20, dtype=torch.uint8, device=result.device)
    packed[:16] = result
    packed[16:20] = result_ok.view(torch.uint8)
    rows = gather_rows(packed, group)
    return rows[:, :16].contiguous(), rows[:, 16:20].contiguous().view(torch.int32).view(-1)


def combine_results(codec, op, dtype, elem_bytes, rows, oks):
    """Last tree level: fold the present per-rank results in rank order on the device.
    rows: uint8[G,16] (value, then the int32 count for AVG), oks: int32[G]."""
    present = rows[oks != 0].contiguous()
    g = present.shape[0]
    base = present.data_ptr() if g else None
    return codec.filter_raw(op, dtype, base, 16, (base + elem_bytes) if g else None, 16, g, 0,
                            W.SHAPE_FLAT)


def relay_node(step, empty_node, group=None):
    """The chain of a float / double SUM or AVG through the ranks: rank r waits for the node of the
    packets in front of its range (rank 0 has none), step(front) continues the chain over the rank's
    own packets and returns the node behind them, which goes to rank r + 1; the last rank's node is
    the wave's and is broadcast.  empty_node() makes a uint8[16] buffer where step wants its front.
    Every rank returns the wave's node."""
    world, rank = _world(group), _rank(group)
    if not dist.is_initialized():
        return step(None)
    front = None
    if rank > 0:
        front = empty_node()
        buf = front.cpu() if _via_host(front, group) else front
        dist.recv(buf, src=_global(rank - 1, group), group=group)
        if buf is not front:
            front.copy_(buf)
    node = step(front)
    send = node.cpu() if _via_host(node, group) else node
    if rank < world - 1:
        dist.send(send, dst=_global(rank + 1, group), group=group)
    dist.broadcast(send, src=_global(world - 1, group), group=group)
    if send is not node:
        node.copy_(send)
    return node


def filter_wire_sharded(codec, op, fmt, wire, sender_bo=W.LITTLEENDIAN, group=None, shape=W.SHAPE_FLAT):
    """tfilter_* over a wave whose packets are spread over the ranks of `group`; `wire` is this
    rank's shard (packed fixed-format layout).  Every rank returns the same (result, ok): with
    SHAPE_FLAT that of ONE reference filter call over the whole wave (module docstring), with
    SHAPE_TREE that of the declared tree with the ranks as its last level."""
    dtype = fmt.types[0]
    if shape == W.SHAPE_TREE:
        res, ok = codec.filter_wire(op, fmt, wire, sender_bo, W.SHAPE_TREE)
        rows, oks = gather_results(res, ok, group)
        return combine_results(codec, op, dtype, fmt.widths[0], rows, oks)
    if codec.is_chained(op, dtype):
        node = relay_node(lambda front: codec.filter_partial_wire(op, fmt, wire, sender_bo, front),
                          lambda: codec.empty(16, torch.uint8), group)
        return codec.filter_combine(op, dtype, node)
    rows = gather_rows(codec.filter_partial_wire(op, fmt, wire, sender_bo), group)
    return codec.filter_combine(op, dtype, rows)


def stats_wire_sharded(codec, fmt, wire, sender_bo=W.LITTLEENDIAN, group=None, shape=W.SHAPE_FLAT):
    """Min, Max and Avg of a '<T> %d' wave spread over the ranks of `group` (the three streams a
    PerformanceData-style tool runs over one wave).  Every rank returns (uint8[48], int32[3]): the
    three 16-byte results (MIN, MAX, AVG) and their presence flags.
    SHAPE_FLAT: what three reference filter calls over the whole wave return -- the MIN and MAX nodes
    (and the AVG node of an integer type) travel in ONE all_gather; the AVG of a float / double wave
    is a chain of additions and is relayed in rank order.
    SHAPE_TREE: one fused pass per shard (mrpdr_wave_filter_stats, the declared tree), ONE all_gather
    of 60 bytes per rank, then the ranks as the last tree level of each operator."""
    if len(fmt) != 2 or not fmt.is_fixed or fmt.types[1] != W.DataType.INT32_T or wire.offsets is not None:
        raise W.PdrError("stats_wire_sharded needs a packed '%T %d' wave")
    base, w, dtype = wire.arena.data_ptr(), fmt.widths[0], fmt.types[0]
    swap = int(sender_bo != W.LITTLEENDIAN)
    ops = (W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG)
    out = torch.zeros(48, dtype=torch.uint8, device=wire.arena.device)
    out_ok = torch.zeros(3, dtype=torch.int32, device=wire.arena.device)
    if shape == W.SHAPE_TREE:
        res, oks = codec.filter_stats_raw(dtype, base, wire.stride, base + w, wire.stride, wire.n, swap)
        row = torch.zeros(60, dtype=torch.uint8, device=res.device)
        row[:48] = res
        row[48:60] = oks.view(torch.uint8)
        rows = gather_rows(row, group)                                   # [G, 60], rank order
        flags = rows[:, 48:60].contiguous().view(torch.int32).view(-1, 3)
        for k, op in enumerate(ops):
            r, ok = combine_results(codec, op, dtype, w, rows[:, 16 * k:16 * k + 16].contiguous(),
                                    flags[:, k].contiguous())
            out[16 * k:16 * k + 16] = r
            out_ok[k:k + 1] = ok
        return out, out_ok

    def part(op, front=None):
        return codec.filter_partial_raw(op, dtype, base, wire.stride, base + w, wire.stride, wire.n, swap, front)

    gathered = [op for op in ops if not codec.is_chained(op, dtype)]
    rows = gather_rows(torch.cat([part(op) for op in gathered]), group)      # [G, 16 * len(gathered)]
    for k, op in enumerate(ops):
        if op in gathered:
            j = gathered.index(op)
            nodes = rows[:, 16 * j:16 * j + 16].contiguous()
        else:
            nodes = relay_node(lambda front: part(op, front), lambda: codec.empty(16, torch.uint8), group)
        codec.filter_combine(op, dtype, nodes, out[16 * k:16 * k + 16], out_ok[k:k + 1])
    return out, out_ok


# ---- tfilter_PerfData across ranks ---------------------------------------------------------
# Every rank folds its shard into one output packet ([-nranks], [len], data[len]); the packets of
# all ranks are gathered and folded once more by the same filter, in rank order -- the per-GPU
# results are the children of the root of the tree (src/FilterDefinitions.C:904-1239 runs at
# every level).  The exchange is one all_gather of 16 + 8*cap bytes per rank.

def pack_perfdata(out_data, out_len, out_rank, cap):
    """One rank's output packet as a fixed-size row: i32 rank | i32 pad | i64 len | cap x 8 bytes."""
    row = torch.zeros(16 + 8 * cap, dtype=torch.uint8, device=out_data.device)
    row[0:4] = out_rank.view(torch.uint8)[:4]
    row[8:16] = out_len.view(torch.uint8)[:8]
    row[16:] = out_data.view(torch.uint8)[:8 * cap]
    return row


def gather_rows(row, group=None):
    """all_gather of equal-size byte rows, rank order (on the device with RCCL, via the host
    with gloo).  The collective runs whenever a process group exists, also one of a single rank."""
    if not dist.is_initialized():
        return row.view(1, -1).clone()
    world = _world(group)
    dev = row.device
    send = row.cpu() if _via_host(row, group) else row
    out = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(out, send, group=group)
    return torch.stack(out).to(dev)


def unpack_perfdata(rows, cap):
    """Gathered rows -> the rank and data columns of a wave of len(rows) packets: the ranks are
    packed, the data arrays keep their row positions (offs = g * cap)."""
    g = rows.shape[0]
    dev = rows.device
    ranks = rows[:, 0:4].contiguous().view(torch.int32).view(-1)
    lens = rows[:, 8:16].contiguous().view(torch.int64).view(-1)
    data = rows[:, 16:].contiguous().view(torch.int64).view(-1)
    ones = torch.ones(g, dtype=torch.int64, device=dev)
    rank_col = W.Column(ranks, ones, torch.arange(g + 1, dtype=torch.int64, device=dev), g)
    data_col = W.Column(data, lens, torch.arange(g + 1, dtype=torch.int64, device=dev) * cap, g * cap)
    return rank_col, data_col


def perfdata_sharded(codec, op, dtype, ranks, data, n, cap, group=None, fanin=0, shape=W.SHAPE_FLAT):
    """tfilter_PerfData over a wave spread over the ranks of `group`; ranks / data are this rank's
    decoded columns (n packets), cap bounds the result length.  Every rank returns the same
    (out_data, out_len, out_rank).
    SHAPE_FLAT: ONE filter invocation over the whole wave (src/FilterDefinitions.C:984-1132 over all
    packets): every rank folds its range into an undivided state; integer states are gathered and
    combined, double states -- a chain of roundings -- are relayed in rank order.
    SHAPE_TREE: the ranks as a tree level: every rank folds its shard into one output packet (fanin > 0:
    as a two-level tree of its own first), the packets are gathered and folded once more by the same
    filter, as the children of the root (AVG divides and re-weights per level)."""
    if shape == W.SHAPE_TREE:
        out, out_len, out_rank = codec.filter_perfdata(op, dtype, ranks, data, n, group=fanin, out_cap=cap)
        rows = gather_rows(pack_perfdata(out, out_len, out_rank, cap), group)
        rank_col, data_col = unpack_perfdata(rows, cap)
        return codec.filter_perfdata(op, dtype, rank_col, data_col, rows.shape[0], out_cap=cap)
    nbytes = codec.perfdata_state_bytes(cap)
    if codec.perfdata_is_chained(op, dtype):
        state = relay_node(lambda front: codec.perfdata_partial(op, dtype, ranks, data, n, cap, front),
                           lambda: codec.empty(nbytes, torch.uint8), group)
        return codec.perfdata_combine(op, dtype, state, cap)
    states = gather_rows(codec.perfdata_partial(op, dtype, ranks, data, n, cap), group)
    return codec.perfdata_combine(op, dtype, states, cap)
