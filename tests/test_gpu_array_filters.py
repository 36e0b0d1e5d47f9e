"""tfilter_ArrayConcat and tfilter_PerfData on the device against the oracle's restatement of
src/FilterDefinitions.C:649-810 and :904-1239.  Integer results and array bytes: bit-exact.
Doubles: bit-exact too, because the oracle folds in the same order (flat) or over the same
declared two-level tree.  Parity unpinned by reference artefacts: the reference's tests never
run these two filters (tests/test_NativeFilters_FE.C covers TFILTER_SUM only)."""
import numpy as np
import pytest
import torch

from wavegen import O, make_wave, random_values, splitmix64

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402
from mrnet_amd import packet as P  # noqa: E402


@pytest.mark.parametrize("fmt_s", ["%ad", "%auc", "%alf", "%Ahd", "%as", "%auld"])
@pytest.mark.parametrize("n", [1, 4, 300])
def test_array_concat(codec, fmt_s, n):
    ocols = make_wave(fmt_s, n, seed=n + len(fmt_s), max_count=19)
    cols = []
    c = ocols[0]
    cols.append((c["counts"], c["sub_counts"], c["values"]) if "sub_counts" in c else (c["counts"], c["values"]))
    wave = P.PacketWave(codec, 77, 101, fmt_s, cols, n)
    out = P.tfilter_ArrayConcat(wave)
    assert out is not None and out.n == 1 and out.get_StreamId() == 77 and out.get_Tag() == 101
    # the reference's output: Packet(stream, tag, fmt, result_array, result_array_size)
    one = dict(c)
    total = int(c["offs"][n])
    one["counts"] = np.array([total], np.uint64)
    one["offs"] = np.array([0, total], np.uint64)
    ref, roffs, rlens = O.wave_encode(fmt_s, [one], 1)
    got = W.Codec.to_numpy(out.wire.arena)[: ref.size]
    assert int(out.wire.lens[0]) == int(rlens[0]) and np.array_equal(got, ref)
    assert codec.status() == 0
    assert P.tfilter_ArrayConcat(P.PacketWave(codec, 1, 1, "%d", [np.arange(3, dtype=np.int32)], 3)) is None


def perf_wave(n, t, seed, kmax=8, same_len=False, aggregated=False):
    dt = np.dtype(O.elem_dtype(t))
    lens = np.full(n, kmax, np.uint64) if same_len else (splitmix64(seed, n) % np.uint64(kmax + 1)).astype(np.uint64)
    if dt.kind == "f":
        vals = ((splitmix64(seed + 1, int(lens.sum())) % np.uint64(2000001)).astype(np.float64) - 1e6) * 1e-3
    else:
        vals = random_values(dt, int(lens.sum()), seed + 1)
        vals = vals >> 3          # sums and the AVG weights still wrap around now and then
    rcnt = np.ones(n, np.uint64)
    ranks = np.arange(n, dtype=np.int32)
    if aggregated:
        ranks = -((splitmix64(seed + 2, n) % np.uint64(5)).astype(np.int32) + 1)
        ranks[::3] = np.arange(n, dtype=np.int32)[::3]
    return (rcnt, ranks), (lens, vals)


PERF_FMT = {O.UINT64_T: "%ad %ad %auld", O.INT64_T: "%ad %ad %ald", O.DOUBLE_T: "%ad %ad %alf"}


@pytest.mark.parametrize("t", [O.UINT64_T, O.INT64_T, O.DOUBLE_T])
@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG])
@pytest.mark.parametrize("n,kmax,same,aggr,group", [(1, 8, True, False, 0), (4, 8, False, False, 0),
                                                     (257, 5, False, True, 0), (1000, 300, False, True, 0),
                                                     (1000, 8, True, True, 64), (5000, 33, False, True, 256)])
def test_perfdata(codec, t, op, n, kmax, same, aggr, group):
    ranks, data = perf_wave(n, t, seed=7 * n + op + t, kmax=kmax, same_len=same, aggregated=aggr)
    if group:
        ref = O.filter_perfdata_tree(op, t, ranks, data, [group, 0])
    else:
        ref = O.filter_perfdata(op, t, ranks, data)
    nel = (np.ones(n, np.uint64), data[0].astype(np.int32))
    wave = P.PacketWave(codec, 5, 100, PERF_FMT[t], [ranks, nel, data], n)
    out = P.tfilter_PerfData(wave, op, group=group)
    assert out is not None and out.n == 1
    cols = out.unpack(PERF_FMT[t])
    torch.cuda.synchronize()
    got_rank = W.Codec.to_numpy(cols[0].values).view(np.int32)[:1]
    got_len = int(W.Codec.to_numpy(cols[2].counts).view(np.uint64)[0])
    got = W.Codec.to_numpy(cols[2].values).view(O.elem_dtype(t))[:got_len]
    assert got_len == ref[1].size and int(got_rank[0]) == ref[0]
    assert int(W.Codec.to_numpy(cols[1].values).view(np.int32)[0]) == got_len
    assert np.array_equal(got.view(np.uint64), ref[1].view(np.uint64)), (got[:4], ref[1][:4])
    assert codec.status() == 0


def test_perfdata_reference_quirks(codec):
    """MIN of int64 never exceeds -1, MAX of negative data is 0, a NaN start for double MIN is
    discarded, packets without data do not count as ranks, AVG divides only the first
    len(last packet) elements."""
    t = O.INT64_T
    ranks = (np.ones(3, np.uint64), np.array([0, 1, 2], np.int32))
    data = (np.array([2, 0, 1], np.uint64), np.array([5, -7, 9], np.int64))
    for op, want in ((W.TFILTER_MIN, [-1, -7]), (W.TFILTER_MAX, [9, 0]), (W.TFILTER_SUM, [14, -7]),
                     (W.TFILTER_AVG, [7, -7])):
        ref = O.filter_perfdata(op, t, ranks, data)
        assert ref[0] == -2 and ref[1].tolist() == want, (op, ref)
        wave = P.PacketWave(codec, 5, 100, PERF_FMT[t], [ranks, (np.ones(3, np.uint64), np.zeros(3, np.int32)), data], 3)
        cols = P.tfilter_PerfData(wave, op).unpack(PERF_FMT[t])
        assert W.Codec.to_numpy(cols[2].values).view(np.int64)[:2].tolist() == want
        assert int(W.Codec.to_numpy(cols[0].values).view(np.int32)[0]) == -2
    dd = (np.array([1, 1], np.uint64), np.array([3.5, 2.25]))
    wave = P.PacketWave(codec, 5, 100, PERF_FMT[O.DOUBLE_T], [(np.ones(2, np.uint64), np.array([4, 9], np.int32)),
                                                            (np.ones(2, np.uint64), np.ones(2, np.int32)), dd], 2)
    cols = P.tfilter_PerfData(wave, W.TFILTER_MIN).unpack(PERF_FMT[O.DOUBLE_T])
    assert float(W.Codec.to_numpy(cols[2].values).view(np.float64)[0]) == 2.25


@pytest.mark.parametrize("n,kmax,vrange,mmax", [(1, 3, 5, 4), (6, 4, 5, 3), (200, 6, 40, 5), (3000, 3, 100000, 4),
                                                (5000, 2, 7, 2)])
def test_int_eq_class(codec, n, kmax, vrange, mmax):
    """tfilter_IntEqClass: classes merged by value (ascending), members in arrival order.
    Unpinned by reference artefacts; the oracle restates the std::map loops."""
    kc = (splitmix64(n, n) % np.uint64(kmax + 1)).astype(np.uint64)                    # classes per packet
    K = int(kc.sum())
    vals = (splitmix64(n + 1, K) % np.uint64(vrange)).astype(np.uint32)
    mc = (splitmix64(n + 2, K) % np.uint64(mmax + 1)).astype(np.uint32)                # members per class
    ko = np.zeros(n + 1, np.int64); np.cumsum(kc, out=ko[1:])
    ec = np.array([int(mc[ko[i]:ko[i + 1]].sum()) for i in range(n)], np.uint64)       # members per packet
    mems = (splitmix64(n + 3, int(ec.sum())) % np.uint64(1 << 20)).astype(np.uint32)
    want = O.filter_int_eq_class((kc, vals), (kc, mc), (ec, mems))
    wave = P.PacketWave(codec, 3, 100, "%aud %aud %aud", [(kc, vals), (kc, mc), (ec, mems)], n)
    out = P.tfilter_IntEqClass(wave)
    assert out is not None and out.n == 1
    cols = out.unpack("%aud %aud %aud")
    torch.cuda.synchronize()
    got = []
    for c in cols:
        cnt = int(W.Codec.to_numpy(c.counts).view(np.uint64)[0])
        got.append(W.Codec.to_numpy(c.values).view(np.uint32)[:cnt])
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert np.all(np.diff(got[0].astype(np.int64)) > 0) and int(got[1].sum()) == got[2].size == int(ec.sum())
    # and the bytes of the output packet
    one = [{"type": O.UINT32_ARRAY_T, "counts": np.array([w.size], np.uint64), "offs": np.array([0, w.size], np.uint64),
            "values": w} for w in want]
    ref, _, rl = O.wave_encode("%aud %aud %aud", one, 1)
    assert np.array_equal(W.Codec.to_numpy(out.wire.arena)[: ref.size], ref)
    assert codec.status() == 0


def _eq_class_by_map(kc, vals, mc, ec, mems):
    """The reference's loops, literally (src/FilterDefinitions.C:824-875): a map from value to the
    members pushed under it; the member index restarts at 0 for every packet; a class is created
    by its first push_back, so a class entry with no members never appears."""
    classes = {}
    ko = np.concatenate([[0], np.cumsum(kc)]).astype(np.int64)
    eo = np.concatenate([[0], np.cumsum(ec)]).astype(np.int64)
    for p in range(len(kc)):
        cur = 0
        for j in range(ko[p], ko[p + 1]):
            for k in range(int(mc[j])):
                classes.setdefault(int(vals[j]), []).append(int(mems[eo[p] + cur + k]))
            cur += int(mc[j])
    keys = sorted(classes)
    return (np.array(keys, np.uint32), np.array([len(classes[k]) for k in keys], np.uint32),
            np.array([m for k in keys for m in classes[k]], np.uint32))


@pytest.mark.parametrize("case", ["zero_member_classes", "surplus_members", "both", "all_empty"])
def test_int_eq_class_follows_the_reference_map(codec, case):
    """Two things the std::map loops do that a sort-based merge has to reproduce on purpose: a class
    entry whose member count is 0 leaves no class behind (also when EVERY entry of that value is
    empty), and a packet whose mems array is longer than its member counts add up to does not
    shift the members of later packets.  Expected side: the map loops written out in the test."""
    n = 700
    kc = (splitmix64(3, n) % np.uint64(4)).astype(np.uint64)
    K = int(kc.sum())
    vals = (splitmix64(4, K) % np.uint64(60)).astype(np.uint32)
    mc = (splitmix64(5, K) % np.uint64(4) + np.uint64(1)).astype(np.uint32)
    if case in ("zero_member_classes", "both"):
        mc[vals % 7 == 3] = 0                         # every entry of these values is empty
        mc[::5] = 0                                   # and some entries of other values
    if case == "all_empty":
        mc[:] = 0
    ko = np.zeros(n + 1, np.int64); np.cumsum(kc, out=ko[1:])
    need = np.array([int(mc[ko[i]:ko[i + 1]].sum()) for i in range(n)], np.uint64)
    ec = need.copy()
    if case in ("surplus_members", "both"):
        ec += (splitmix64(6, n) % np.uint64(3)).astype(np.uint64)     # 0-2 members nobody claims
    mems = (splitmix64(7, int(ec.sum())) % np.uint64(1 << 24)).astype(np.uint32)
    want = _eq_class_by_map(kc, vals, mc, ec, mems)
    orc = O.filter_int_eq_class((kc, vals), (kc, mc), (ec, mems))
    for w, o in zip(want, orc):
        assert np.array_equal(w, o), "oracle differs from the map loops"
    wave = P.PacketWave(codec, 3, 100, "%aud %aud %aud", [(kc, vals), (kc, mc), (ec, mems)], n)
    out = P.tfilter_IntEqClass(wave)
    cols = out.unpack("%aud %aud %aud")
    torch.cuda.synchronize()
    for c, w in zip(cols, want):
        cnt = int(W.Codec.to_numpy(c.counts).view(np.uint64)[0])
        assert cnt == w.size and np.array_equal(W.Codec.to_numpy(c.values).view(np.uint32)[:cnt], w), case
    assert codec.status() == 0


def test_perfdata_concat(codec):
    """TFILTER_ARRAY_CONCAT in tfilter_PerfData (src/FilterDefinitions.C:1000-1008, 1171-1203):
    packets without data drop out with their rank and nelems arrays."""
    n = 500
    dlen = (splitmix64(5, n) % np.uint64(4)).astype(np.uint64)                 # a quarter have no data
    rlen = (splitmix64(6, n) % np.uint64(3) + np.uint64(1)).astype(np.uint64)
    ranks = np.arange(int(rlen.sum()), dtype=np.int32)
    nel = np.arange(int(rlen.sum()), dtype=np.int32) * 3
    data = random_values(np.float64, int(dlen.sum()), 9)
    wave = P.PacketWave(codec, 5, 100, "%ad %ad %alf", [(rlen, ranks), (rlen, nel), (dlen, data)], n)
    out = P.tfilter_PerfData(wave, W.TFILTER_ARRAY_CONCAT)
    cols = out.unpack("%ad %ad %alf")
    torch.cuda.synchronize()
    ro = np.concatenate([[0], np.cumsum(rlen)]).astype(np.int64)
    keep = np.concatenate([np.arange(ro[i], ro[i + 1]) for i in range(n) if dlen[i]]).astype(np.int64)
    for c, want, dt in ((cols[0], ranks[keep], np.int32), (cols[1], nel[keep], np.int32), (cols[2], data, np.float64)):
        cnt = int(W.Codec.to_numpy(c.counts).view(np.uint64)[0])
        assert cnt == want.size
        assert np.array_equal(W.Codec.to_numpy(c.values).view(dt)[:cnt].view(np.uint8), want.view(np.uint8))
    assert codec.status() == 0


import os  # noqa: E402

SOAK = max(1, int(os.environ.get("MRPDR_SOAK", "1")))


@pytest.mark.parametrize("seed", range(40 * SOAK))
def test_random_perfdata(codec, seed):
    """Drawn wave size, element type, operator, array lengths (many empty), rank arrays of 0-3
    entries with aggregate (negative) ranks, and one- or two-level shapes, against the oracle."""
    r = splitmix64(41000 + seed, 16)
    n = [1, 2, 3, 64, 255, 256, 257, 1000, 4000][int(r[0] % np.uint64(9))]
    if int(r[1] % np.uint64(3)) == 0:
        n = int(r[2] % np.uint64(6000)) + 1
    t = [O.UINT64_T, O.INT64_T, O.DOUBLE_T][int(r[3] % np.uint64(3))]
    op = [W.TFILTER_SUM, W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG][int(r[4] % np.uint64(4))]
    kmax = [1, 3, 8, 40, 300, 600][int(r[5] % np.uint64(6))]
    group = [0, 0, 2, 7, 64, 256, 1000][int(r[6] % np.uint64(7))]
    dt = np.dtype(O.elem_dtype(t))
    lens = (splitmix64(seed + 1, n) % np.uint64(kmax + 1)).astype(np.uint64)
    lens[(splitmix64(seed + 2, n) % np.uint64(4)) == 0] = 0                      # a quarter carry no data
    if int(r[7] % np.uint64(2)):
        lens[lens > 0] = kmax                                                   # equal lengths, as real streams have
    if dt.kind == "f":
        vals = ((splitmix64(seed + 3, int(lens.sum())) % np.uint64(2000001)).astype(np.float64) - 1e6) * 1e-3
    else:
        vals = random_values(dt, int(lens.sum()), seed + 3) >> 3
    rcnt = (splitmix64(seed + 4, n) % np.uint64(4)).astype(np.uint64)
    R = int(rcnt.sum())
    rvals = (splitmix64(seed + 5, R) % np.uint64(11)).astype(np.int32) - 6      # -6 .. 4
    ranks, data = (rcnt, rvals), (lens, vals)
    if group and group < n:
        ref = O.filter_perfdata_tree(op, t, ranks, data, [group, 0])
    else:
        group = 0
        ref = O.filter_perfdata(op, t, ranks, data)
    nel = (np.ones(n, np.uint64), lens.astype(np.int32))
    wave = P.PacketWave(codec, 5, 100, PERF_FMT[t], [ranks, nel, data], n)
    out = P.tfilter_PerfData(wave, op, group=group)
    assert out is not None and out.n == 1
    cols = out.unpack(PERF_FMT[t])
    torch.cuda.synchronize()
    got_len = int(W.Codec.to_numpy(cols[2].counts).view(np.uint64)[0])
    got = W.Codec.to_numpy(cols[2].values).view(dt)[:got_len]
    assert ref is not None
    assert got_len == ref[1].size, (n, t, op, kmax, group)
    assert int(W.Codec.to_numpy(cols[0].values).view(np.int32)[0]) == ref[0]
    assert np.array_equal(got.view(np.uint64), ref[1].view(np.uint64)), (n, t, op, kmax, group)
    assert codec.status() == 0


@pytest.mark.parametrize("seed", range(25 * SOAK))
def test_random_int_eq_class(codec, seed):
    r = splitmix64(43000 + seed, 8)
    n = int(r[0] % np.uint64([5, 300, 9000][int(r[1] % np.uint64(3))])) + 1
    kmax = int(r[2] % np.uint64(7))
    vrange = [1, 3, 50, 100000, 1 << 32][int(r[3] % np.uint64(5))]
    mmax = int(r[4] % np.uint64(6))
    kc = (splitmix64(seed + 1, n) % np.uint64(kmax + 1)).astype(np.uint64)
    K = int(kc.sum())
    vals = (splitmix64(seed + 2, K) % np.uint64(vrange)).astype(np.uint32)
    mc = (splitmix64(seed + 3, K) % np.uint64(mmax + 1)).astype(np.uint32)
    ko = np.zeros(n + 1, np.int64); np.cumsum(kc, out=ko[1:])
    ec = np.array([int(mc[ko[i]:ko[i + 1]].sum()) for i in range(n)], np.uint64)
    mems = (splitmix64(seed + 4, int(ec.sum())) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    want = O.filter_int_eq_class((kc, vals), (kc, mc), (ec, mems))
    wave = P.PacketWave(codec, 3, 100, "%aud %aud %aud", [(kc, vals), (kc, mc), (ec, mems)], n)
    out = P.tfilter_IntEqClass(wave)
    cols = out.unpack("%aud %aud %aud")
    torch.cuda.synchronize()
    for c, w in zip(cols, want):
        cnt = int(W.Codec.to_numpy(c.counts).view(np.uint64)[0])
        assert cnt == w.size and np.array_equal(W.Codec.to_numpy(c.values).view(np.uint32)[:cnt], w), (n, kmax, vrange, mmax)
    assert codec.status() == 0


@pytest.mark.parametrize("t", [O.UINT64_T, O.INT64_T])
@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_MIN, W.TFILTER_MAX])
def test_perfdata_integer_folds_do_not_depend_on_the_tree(codec, t, op):
    """One node over 20,000 packets (computed as a two-level tree on the device) against the
    oracle's single flat invocation: identical bits for integer SUM / MIN / MAX."""
    n = 20000
    ranks, data = perf_wave(n, t, seed=op + 3 * t, kmax=9, aggregated=True)
    ref = O.filter_perfdata(op, t, ranks, data)
    nel = (np.ones(n, np.uint64), data[0].astype(np.int32))
    wave = P.PacketWave(codec, 5, 100, PERF_FMT[t], [ranks, nel, data], n)
    cols = P.tfilter_PerfData(wave, op).unpack(PERF_FMT[t])
    torch.cuda.synchronize()
    got_len = int(W.Codec.to_numpy(cols[2].counts).view(np.uint64)[0])
    assert got_len == ref[1].size and int(W.Codec.to_numpy(cols[0].values).view(np.int32)[0]) == ref[0]
    assert np.array_equal(W.Codec.to_numpy(cols[2].values).view(np.uint64)[:got_len], ref[1].view(np.uint64))


def _perf_cols(codec, t, ranks, data, n):
    fmt = W.Format(PERF_FMT[t])
    nel = (np.ones(n, np.uint64), data[0].astype(np.int32))
    return codec.columns_from_numpy(fmt, [ranks, nel, data])


def _perf_result(t, out, out_len, out_rank):
    torch.cuda.synchronize()
    k = int(out_len.item())
    return int(out_rank.item()), W.Codec.to_numpy(out).view(O.elem_dtype(t))[:k]


@pytest.mark.parametrize("t", [O.UINT64_T, O.INT64_T, O.DOUBLE_T])
@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG])
@pytest.mark.parametrize("n,kmax,same,fanins", [(5000, 8, True, [32, 64, 0]), (5000, 8, False, [7, 5, 3]),
                                                (131072, 8, True, [32, 32, 32, 0]), (40001, 8, False, [32, 32, 32, 0]),
                                                (9000, 5, False, [32, 32, 0]), (2000, 13, False, [16, 16, 0]),
                                                (1500, 30, False, [8, 8, 8, 8]), (33, 8, False, [32, 32, 0]),
                                                (40000, 8, False, [32, 64, 0]), (3000, 40, False, [16, 0]),
                                                (3000, 100, False, [16, 4, 0]), (70, 8, True, [32, 64, 0]),
                                                (1, 8, True, [32, 0])])
def test_perfdata_declared_trees_of_any_depth(codec, t, op, n, kmax, same, fanins):
    """mrpdr_wave_perfdata_levels: the filter applied level by level over the declared fan-ins (many
    nodes per workgroup for short arrays), against the oracle applying its one-invocation restatement
    over the same tree.  Bits, doubles included (same order of additions)."""
    ranks, data = perf_wave(n, t, seed=3 * n + op + t + len(fanins), kmax=kmax, same_len=same, aggregated=True)
    # (fan-ins that leave several nodes: the library folds them in one more level)
    ref = O.filter_perfdata_tree(op, t, ranks, data, fanins if fanins[-1] == 0 else fanins + [0])
    cols = _perf_cols(codec, t, ranks, data, n)
    got = _perf_result(t, *codec.filter_perfdata_levels(op, t, cols[0], cols[2], n, fanins, out_cap=kmax))
    assert got[0] == ref[0] and got[1].size == ref[1].size
    assert np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64)), (got[1][:4], ref[1][:4])
    assert codec.status() == 0


@pytest.mark.parametrize("t", [O.UINT64_T, O.INT64_T])
@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG])
def test_perfdata_integer_one_invocation_is_parallel_and_exact(codec, t, op):
    """One invocation over 131,072 integer packets: the library folds it as a tree of undivided nodes
    (wrapping sums of data * nrank and of the ranks, ONE division at the top,
    src/FilterDefinitions.C:1084-1132) -- against the oracle's serial loop over all packets, and against
    the device's own serial fold (no scratch).  Ragged lengths, aggregated ranks, a last packet shorter
    than the longest (AVG divides only its elements), and a last packet without data (no division)."""
    n = 131072
    for variant in range(3):
        ranks, data = perf_wave(n, t, seed=40 + variant + op, kmax=8, same_len=(variant == 0), aggregated=True)
        lens, vals = data
        if variant == 2:
            lens = lens.copy()
            cut = int(lens[:-1].sum())
            lens[-1] = 0
            data = (lens, vals[:cut])
        ref = O.filter_perfdata(op, t, ranks, data)
        cols = _perf_cols(codec, t, ranks, data, n)
        par = _perf_result(t, *codec.filter_perfdata(op, t, cols[0], cols[2], n, group=0, out_cap=8))
        assert par[0] == ref[0] and np.array_equal(par[1].view(np.uint64), ref[1].view(np.uint64)), (variant, par, ref)
        # the serial fold of the same call (no scratch handed over)
        import ctypes as C
        from mrnet_amd import capi
        out = codec.empty(8, torch.int64)
        out_len, out_rank = codec.empty(1, torch.int64), codec.empty(1, torch.int32)
        a, b = cols[0].c_struct(), cols[2].c_struct()
        assert capi.load().mrpdr_wave_perfdata(codec.handle, op, int(t), C.byref(a), C.byref(b), n, 0, out.data_ptr(), 8,
                                               out_len.data_ptr(), out_rank.data_ptr(), None, 0,
                                               torch.cuda.current_stream().cuda_stream) == 1
        ser = _perf_result(t, out, out_len, out_rank)
        assert ser[0] == ref[0] and np.array_equal(ser[1].view(np.uint64), ref[1].view(np.uint64)), variant
    assert codec.status() == 0


@pytest.mark.parametrize("t", [O.UINT64_T, O.INT64_T, O.DOUBLE_T])
@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_MIN, W.TFILTER_MAX, W.TFILTER_AVG])
def test_perfdata_in_ranges_equals_one_invocation(codec, t, op):
    """mrpdr_wave_perfdata_partial / _combine: a wave in 2, 3 and 8 packet ranges (as the GPUs of a node
    hold it) against ONE oracle invocation over the WHOLE wave -- integer states gathered, double states
    relayed (each range continues from the state in front).  Ragged ranges with empty ones, NaNs in
    double data (MIN / MAX keep the later operand when a comparison is false)."""
    for n, kmax in ((20000, 8), (5000, 33), (700, 100)):
        ranks, data = perf_wave(n, t, seed=11 * n + op + t, kmax=kmax, aggregated=True)
        if t == O.DOUBLE_T and op in (W.TFILTER_MIN, W.TFILTER_MAX):
            data[1][:: max(1, data[1].size // 50)] = np.nan
        ref = O.filter_perfdata(op, t, ranks, data)
        ro = np.zeros(n + 1, np.int64); np.cumsum(ranks[0], out=ro[1:])
        do = np.zeros(n + 1, np.int64); np.cumsum(data[0], out=do[1:])
        chained = codec.perfdata_is_chained(op, t)
        assert chained == (t == O.DOUBLE_T)
        for edges in ([0, n // 2, n], [0, n // 3, n // 3, n - 5, n], [0] + [n * r // 8 for r in range(1, 8)] + [n], [0, 0, n, n]):
            states, front = [], None
            for lo, hi in zip(edges[:-1], edges[1:]):
                rk = (ranks[0][lo:hi], ranks[1][ro[lo]:ro[hi]])
                dd = (data[0][lo:hi], data[1][do[lo]:do[hi]])
                if hi > lo:
                    cols = _perf_cols(codec, t, rk, dd, hi - lo)
                    st = codec.perfdata_partial(op, t, cols[0], cols[2], hi - lo, kmax, front if chained else None)
                else:
                    empty = W.Column(None, None, None, 0)
                    st = codec.perfdata_partial(op, t, empty, empty, 0, kmax, front if chained else None)
                states.append(st)
                front = st
            got = _perf_result(t, *codec.perfdata_combine(op, t, torch.stack(states), kmax))
            assert got[0] == ref[0] and got[1].size == ref[1].size, (n, edges, got, ref)
            assert np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64)), (n, edges, got[1][:4], ref[1][:4])
    assert codec.status() == 0


@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_AVG, W.TFILTER_MIN, W.TFILTER_MAX])
@pytest.mark.parametrize("n,k", [(8192, 8), (20001, 3), (131072, 8), (40000, 64), (9000, 1)])
def test_perfdata_double_one_invocation_regular_waves(codec, op, n, k):
    """One invocation over doubles is a chain of roundings per element.  When every packet carries the
    same number of elements, end to end (one metric's stream), the elements are strided columns and the
    device-wide exact fold takes their chains side by side: the bits of the oracle's serial loop over all
    packets.  Checked on regular waves (aggregated ranks as weights), on the same wave with one packet
    made shorter or empty (the device finds out and the serial fold runs), and with a first array that
    does not start the values column."""
    t = O.DOUBLE_T
    for variant in ("regular", "one_short", "one_empty", "offset", "negative"):
        ranks, data = perf_wave(n, t, seed=n + k + op, kmax=k, same_len=True, aggregated=True)
        lens, vals = data
        if op in (W.TFILTER_MIN, W.TFILTER_MAX):
            vals = vals.copy()
            vals[:: max(1, vals.size // 40)] = np.nan          # NaNs restart the comparison
            if variant == "negative":
                vals = -np.abs(vals)                           # MAX starts from 0: all-negative data gives 0 (or a late NaN)
        if variant == "one_short" and k > 1:
            lens = lens.copy()
            at = int(lens[: n // 2].sum())
            lens[n // 2] = k - 1
            vals = np.concatenate([vals[: at + k - 1], vals[at + k:]])
        elif variant == "one_empty":
            lens = lens.copy()
            at = int(lens[: n - 1].sum())
            lens[n - 1] = 0
            vals = vals[:at]
        data = (lens, vals)
        ref = O.filter_perfdata(op, t, ranks, data)
        cols = _perf_cols(codec, t, ranks, data, n)
        if variant == "offset":
            # the packets' arrays behind five elements nobody owns
            pad = torch.zeros(5, dtype=cols[2].values.dtype, device=codec.device)
            cols[2].values = torch.cat([pad, cols[2].values])
            cols[2].offs = cols[2].offs + 5
            cols[2].capacity = cols[2].values.numel()
        got = _perf_result(t, *codec.filter_perfdata(op, t, cols[0], cols[2], n, group=0, out_cap=k))
        assert got[0] == ref[0] and got[1].size == ref[1].size, (variant, got, ref)
        assert np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64)), (variant, got[1][:4], ref[1][:4])
    assert codec.status() == 0


@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_AVG, W.TFILTER_MIN, W.TFILTER_MAX])
def test_perfdata_double_ranges_relay_the_columns(codec, op):
    """A regular double wave in ranges large enough for the strided exact fold: every range continues
    the elements' chains from the state in front (fronts of the same form go on as columns; a range of
    another form -- ragged lengths -- and whatever follows it take the serial fold from that state on).
    Against ONE oracle invocation over the whole wave."""
    t, k = O.DOUBLE_T, 8
    for n, edges, ragged_range in ((60000, [0, 20000, 40000, 60000], None), (50000, [0, 9000, 30000, 30000, 50000], None),
                                   (45000, [0, 15000, 30000, 45000], 1), (30000, [0, 10000, 20000, 30000], 0)):
        ranks, data = perf_wave(n, t, seed=n + op, kmax=k, same_len=True, aggregated=True)
        lens, vals = data
        if op in (W.TFILTER_MIN, W.TFILTER_MAX):
            vals = -np.abs(vals)
            vals[:: 997] = np.nan
            vals[edges[1] * k - 3: edges[1] * k + 2] = np.nan     # NaNs on both sides of a range boundary
        if ragged_range is not None:
            lo = edges[ragged_range]
            lens = lens.copy()
            at = int(lens[: lo + 7].sum())
            lens[lo + 7] = k - 3                              # one shorter array inside that range
            vals = np.concatenate([vals[: at + k - 3], vals[at + k:]])
        data = (lens, vals)
        ref = O.filter_perfdata(op, t, ranks, data)
        ro = np.zeros(n + 1, np.int64); np.cumsum(ranks[0], out=ro[1:])
        do = np.zeros(n + 1, np.int64); np.cumsum(data[0], out=do[1:])
        front, states = None, []
        for lo, hi in zip(edges[:-1], edges[1:]):
            if hi > lo:
                cols = _perf_cols(codec, t, (ranks[0][lo:hi], ranks[1][ro[lo]:ro[hi]]), (data[0][lo:hi], data[1][do[lo]:do[hi]]), hi - lo)
                front = codec.perfdata_partial(op, t, cols[0], cols[2], hi - lo, k, front)
            else:
                empty = W.Column(None, None, None, 0)
                front = codec.perfdata_partial(op, t, empty, empty, 0, k, front)
            states.append(front)
        got = _perf_result(t, *codec.perfdata_combine(op, t, torch.stack(states), k))
        assert got[0] == ref[0] and got[1].size == ref[1].size, (n, got, ref)
        assert np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64)), (n, ragged_range, got[1][:4], ref[1][:4])
    assert codec.status() == 0


@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_AVG, W.TFILTER_MIN, W.TFILTER_MAX])
@pytest.mark.parametrize("n,kmax", [(8192, 8), (30001, 5), (131072, 8), (20000, 64), (9000, 1)])
def test_perfdata_double_one_invocation_ragged_waves(codec, op, n, kmax):
    """Double waves whose arrays have drawn lengths 0 .. kmax (packets without data among them): the
    elements' chains run over a dense copy in which a missing element is +0.0 -- no bit of a sum changes
    by it -- and MIN / MAX skip what a packet does not carry.  Against the oracle's serial loop, bits.
    Variants: the last packet empty (AVG then divides nothing, src/FilterDefinitions.C:1114), the last
    packet of one element (AVG divides element 0 alone), room for more elements than any packet has, -0.0
    and NaN data, all-negative data for MAX (whose start value is 0)."""
    t = O.DOUBLE_T
    for variant in ("drawn", "last_empty", "last_one", "roomy", "signed_zeros", "negative"):
        ranks, data = perf_wave(n, t, seed=5 * n + kmax + op, kmax=kmax, aggregated=True)
        lens, vals = data[0].copy(), data[1].copy()
        if variant == "last_empty":
            vals = vals[: vals.size - int(lens[-1])]
            lens[-1] = 0
        elif variant == "last_one":
            vals = np.concatenate([vals[: vals.size - int(lens[-1])], np.array([0.25])])
            lens[-1] = 1
        elif variant == "signed_zeros":
            vals[: vals.size // 2] = 0.0
            vals[: vals.size // 2: 2] = -0.0                  # sums that sit at zero while elements go missing
        elif variant == "negative":
            vals = -np.abs(vals) - 1.0
        if op in (W.TFILTER_MIN, W.TFILTER_MAX) and variant != "signed_zeros":
            vals[:: max(1, vals.size // 40)] = np.nan
        data = (lens, vals)
        cap = kmax + 3 if variant == "roomy" else kmax
        ref = O.filter_perfdata(op, t, ranks, data)
        cols = _perf_cols(codec, t, ranks, data, n)
        got = _perf_result(t, *codec.filter_perfdata(op, t, cols[0], cols[2], n, group=0, out_cap=cap))
        assert got[0] == ref[0] and got[1].size == ref[1].size, (variant, got, ref)
        assert np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64)), (variant, got[1][:4], ref[1][:4])
    assert codec.status() == 0


def test_perfdata_double_ragged_wave_longer_than_the_bound(codec):
    """A ragged wave whose longest array does not fit out_cap: the device finds out, no column path takes
    it, and the call reports as it always has (status bit 1, no elements)."""
    t, n, kmax = O.DOUBLE_T, 10000, 8
    ranks, data = perf_wave(n, t, seed=77, kmax=kmax)
    cols = _perf_cols(codec, t, ranks, data, n)
    out, out_len, out_rank = codec.filter_perfdata(W.TFILTER_SUM, t, cols[0], cols[2], n, group=0, out_cap=kmax - 2)
    torch.cuda.synchronize()
    assert int(out_len.item()) == 0 and codec.status() & 1


@pytest.mark.parametrize("op", [W.TFILTER_SUM, W.TFILTER_AVG, W.TFILTER_MIN, W.TFILTER_MAX])
def test_perfdata_double_ragged_ranges_relay_the_columns(codec, op):
    """Ragged double waves in ranges: each range's dense columns continue from the state in front, whatever
    form that has (a regular range behind a ragged one, a ragged one behind a regular one, a range whose
    packets are all shorter than what the front already holds, a range that is all empty packets)."""
    t, k = O.DOUBLE_T, 8
    for n, edges, forms in ((60000, [0, 20000, 40000, 60000], "rgr"), (48000, [0, 16000, 32000, 48000], "grs"),
                            (40000, [0, 10000, 20000, 20000, 40000], "ge-g"), (36000, [0, 12000, 24000, 36000], "szg")):
        ranks, _ = perf_wave(n, t, seed=n + op, kmax=k, aggregated=True)
        lens = np.zeros(n, np.uint64)
        for (lo, hi), f in zip(zip(edges[:-1], edges[1:]), forms):
            m = hi - lo
            if f == "r":   lens[lo:hi] = k                                                            # regular
            elif f == "g": lens[lo:hi] = splitmix64(n + lo, m) % np.uint64(k + 1)                     # ragged, with empty packets
            elif f == "s": lens[lo:hi] = splitmix64(n + lo, m) % np.uint64(3) + np.uint64(1)          # short: 1 .. 3 elements
            elif f == "z": lens[lo:hi] = 0                                                            # nobody carries data
        vals = ((splitmix64(n + 1, int(lens.sum())) % np.uint64(2000001)).astype(np.float64) - 1e6) * 1e-3
        if op in (W.TFILTER_MIN, W.TFILTER_MAX):
            vals = -np.abs(vals)
            vals[:: 991] = np.nan
        data = (lens, vals)
        ref = O.filter_perfdata(op, t, ranks, data)
        ro = np.zeros(n + 1, np.int64); np.cumsum(ranks[0], out=ro[1:])
        do = np.zeros(n + 1, np.int64); np.cumsum(lens, out=do[1:])
        front, states = None, []
        for lo, hi in zip(edges[:-1], edges[1:]):
            if hi > lo:
                cols = _perf_cols(codec, t, (ranks[0][lo:hi], ranks[1][ro[lo]:ro[hi]]), (lens[lo:hi], vals[do[lo]:do[hi]]), hi - lo)
                front = codec.perfdata_partial(op, t, cols[0], cols[2], hi - lo, k, front)
            else:
                empty = W.Column(None, None, None, 0)
                front = codec.perfdata_partial(op, t, empty, empty, 0, k, front)
            states.append(front)
        got = _perf_result(t, *codec.perfdata_combine(op, t, torch.stack(states), k))
        assert got[0] == ref[0] and got[1].size == ref[1].size, (n, forms, got, ref)
        assert np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64)), (n, forms, got[1][:4], ref[1][:4])
    assert codec.status() == 0


@pytest.mark.parametrize("t,op", [(O.DOUBLE_T, W.TFILTER_AVG), (O.INT64_T, W.TFILTER_AVG), (O.DOUBLE_T, W.TFILTER_MIN),
                                  (O.DOUBLE_T, W.TFILTER_SUM)])
def test_perfdata_full_size_wave_in_8_ranges(codec, t, op):
    """The performance-data configuration at its full size: 1,048,576 packets of eight-element arrays held
    as 8 ranges of 131,072 (one per GPU), every range folded into a state (doubles relayed through the
    elements' exact chains, integers gathered), against ONE oracle invocation over all packets.  Filter
    times as the benchmark draws them (normal around 1e-3), a NaN now and then for MIN."""
    n, k, g = 1 << 20, 8, 8
    rng = np.random.default_rng(2026)
    vals = rng.normal(1e-3, 1e-4, n * k)
    if t == O.INT64_T:
        vals = (vals * 1e9).astype(np.int64)
    elif op == W.TFILTER_MIN:
        vals[:: 100003] = np.nan
    lens = np.full(n, k, np.uint64)
    ranks = (np.ones(n, np.uint64), np.arange(n, dtype=np.int32))
    ref = O.filter_perfdata(op, t, ranks, (lens, vals))
    chained = codec.perfdata_is_chained(op, t)
    states, front = [], None
    for r in range(g):
        lo, hi = r * n // g, (r + 1) * n // g
        cols = _perf_cols(codec, t, (ranks[0][lo:hi], ranks[1][lo:hi]), (lens[lo:hi], vals[lo * k:hi * k]), hi - lo)
        front = codec.perfdata_partial(op, t, cols[0], cols[2], hi - lo, k, front if chained else None)
        states.append(front)
    got = _perf_result(t, *codec.perfdata_combine(op, t, torch.stack(states), k))
    assert got[0] == ref[0] == -n and got[1].size == ref[1].size == k
    assert np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64)), (got[1][:4], ref[1][:4])
    assert codec.status() == 0
