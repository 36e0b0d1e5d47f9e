"""Multi-GPU side of the wave codec: one process per GPU, torch.distributed (RCCL on GPUs).

Packets of a wave are independent, so encode/decode shard by packet range with no
collective at all.  The only exchange is the filter result: every rank reduces its shard
with the declared tree shape, the 16-byte results are all-gathered, and the last tree level
(fan-in = number of ranks, rank order) is folded on every rank.  An ordered fold rather than
an all-reduce sum: float Sum and every Avg depend on the fold order
(src/FilterDefinitions.C:1925-2040).
"""
import torch
import torch.distributed as dist

from . import wave as W


def shard_range(n, rank, world):
    """Contiguous packet range [lo, hi) of `rank`: [rank*n/world, (rank+1)*n/world)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def gather_results(result, result_ok, group=None):
    """all_gather the per-rank filter results.  result: uint8[16], result_ok: int32[1] (on the
    device with RCCL, on the host with gloo).  Returns (uint8[G,16], int32[G]) in rank order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return result.view(1, 16).clone(), result_ok.view(1).clone()
    packed = torch.zeros(20, dtype=torch.uint8, device=result.device)
    packed[:16] = result
    packed[16:20] = result_ok.view(torch.uint8)
    dev = packed.device
    if packed.is_cuda and dist.get_backend(group) == "gloo":
        packed = packed.cpu()            # CPU rehearsal of the exchange; RCCL gathers on the device
    out = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(out, packed, group=group)
    rows = torch.stack(out).to(dev)
    return rows[:, :16].contiguous(), rows[:, 16:20].contiguous().view(torch.int32).view(-1)


def combine_results(codec, op, dtype, elem_bytes, rows, oks):
    """Last tree level: fold the present per-rank results in rank order on the device.
    rows: uint8[G,16] (value, then the int32 count for AVG), oks: int32[G]."""
    present = rows[oks != 0].contiguous()
    g = present.shape[0]
    base = present.data_ptr() if g else None
    return codec.filter_raw(op, dtype, base, 16, (base + elem_bytes) if g else None, 16, g, 0,
                            W.SHAPE_FLAT)


def filter_wire_sharded(codec, op, fmt, wire, sender_bo=W.LITTLEENDIAN, group=None):
    """tfilter_* over a wave whose packets are spread over the ranks of `group`; `wire` is this
    rank's shard (packed fixed-format layout).  Every rank returns the same (result, ok)."""
    res, ok = codec.filter_wire(op, fmt, wire, sender_bo, W.SHAPE_TREE)
    rows, oks = gather_results(res, ok, group)
    return combine_results(codec, op, fmt.types[0], fmt.widths[0], rows, oks)


def stats_wire_sharded(codec, fmt, wire, sender_bo=W.LITTLEENDIAN, group=None):
    """Min, Max and Avg of a '<T> %d' wave spread over the ranks of `group` (the three streams a
    PerformanceData-style tool runs over one wave): one fused pass over this rank's shard
    (mrpdr_wave_filter_stats, tree shape), ONE all_gather of 60 bytes per rank, then the last tree
    level of each operator folded in rank order.  Every rank returns (uint8[48], int32[3]): the three
    16-byte results (MIN, MAX, AVG) and their presence flags -- equal to three filter_wire_sharded
    calls, with one exchange instead of three."""
    if len(fmt) != 2 or not fmt.is_fixed or fmt.types[1] != W.DataType.INT32_T or wire.offsets is not None:
        raise W.PdrError("stats_wire_sharded needs a packed '%T %d' wave")
    base, w = wire.arena.data_ptr(), fmt.widths[0]
    swap = int(sender_bo != W.LITTLEENDIAN)
    res, oks = codec.filter_stats_raw(fmt.types[0], base, wire.stride, base + w, wire.stride, wire.n, swap)
    row = torch.zeros(60, dtype=torch.uint8, device=res.device)
    row[:48] = res
    row[48:60] = oks.view(torch.uint8)
    rows = gather_rows(row, group)                                   # [G, 60], rank order
    flags = rows[:, 48:60].contiguous().view(torch.int32).view(-1, 3)
    out = torch.zeros(48, dtype=torch.uint8, device=res.device)
    out_ok = torch.zeros(3, dtype=torch.int32, device=res.device)
    for k, op in enumerate((W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG)):
        r, ok = combine_results(codec, op, fmt.types[0], w, rows[:, 16 * k:16 * k + 16].contiguous(),
                                flags[:, k].contiguous())
        out[16 * k:16 * k + 16] = r
        out_ok[k:k + 1] = ok
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
    with gloo)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return row.view(1, -1).clone()
    dev = row.device
    send = row.cpu() if row.is_cuda and dist.get_backend(group) == "gloo" else row
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


def perfdata_sharded(codec, op, dtype, ranks, data, n, cap, group=None, fanin=0):
    """tfilter_PerfData over a wave spread over the ranks of `group`; ranks / data are this rank's
    decoded columns (n packets), cap bounds the result length.  fanin > 0 folds the rank's own
    packets as a two-level tree (nodes of `fanin` packets, then one node over them) before the
    exchange -- a flat fold of a large shard is one long serial chain per element.  Every rank
    returns the same (out_data, out_len, out_rank)."""
    out, out_len, out_rank = codec.filter_perfdata(op, dtype, ranks, data, n, group=fanin, out_cap=cap)
    rows = gather_rows(pack_perfdata(out, out_len, out_rank, cap), group)
    rank_col, data_col = unpack_perfdata(rows, cap)
    return codec.filter_perfdata(op, dtype, rank_col, data_col, rows.shape[0], out_cap=cap)
