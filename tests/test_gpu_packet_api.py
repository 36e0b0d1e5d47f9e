"""The reference's own end-to-end tests, replayed on the wave API: tests/test_basic_FE.C
(every scalar type and the alltypes packet, packed by the back-ends, unpacked by the
front-end), tests/test_arrays_FE.C (ARRAY_LEN elements, memcmp) and
tests/test_NativeFilters_FE.C (TFILTER_SUM of one constant per back-end)."""
import numpy as np
import pytest
import torch

from wavegen import O

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402
from mrnet_amd.packet import PacketWave, tfilter_Avg, tfilter_Max, tfilter_Min, tfilter_Sum  # noqa: E402

TOK_DTYPE = {"c": np.int8, "uc": np.uint8, "hd": np.int16, "uhd": np.uint16, "d": np.int32,
             "ud": np.uint32, "ld": np.int64, "uld": np.uint64, "f": np.float32, "lf": np.float64}
PROT_BASIC = W.FIRST_APPLICATION_TAG + 1


def received(codec, sent):
    """What the front-end holds after Message::recv: header and data buffers of the wave."""
    return PacketWave.from_buffers(codec, sent.get_FormatString(), sent.size(), sent.get_Header(),
                                   sent.get_HeaderLen(), sent.get_Buffer(), inlet_rank=0)


def string_col(s, n):
    b = np.frombuffer(s.encode() + b"\0", np.uint8)
    return np.full(n, b.size, np.uint64), np.tile(b, n)


@pytest.mark.parametrize("num_backends", [1, 4, 16])
def test_basic_scalar_round_trips(codec, golden, num_backends):
    """tests/test_basic_FE.C: send_val -17 / 17 / -17.234f / 17.3421 per type, echo, unpack, compare."""
    vals = {"c": -17, "uc": 17, "hd": -17, "uhd": 17, "d": -17, "ud": 17, "ld": -17, "uld": 17,
            "f": -17.234, "lf": 17.3421}
    for tok, v in vals.items():
        fmt = "%" + tok
        col = np.full(num_backends, v, TOK_DTYPE[tok])
        sent = PacketWave(codec, W.USER_STRM_BASE_ID, PROT_BASIC, fmt, [col],
                          src_ranks=codec.to_device(np.arange(num_backends, dtype=np.uint32)))
        got = received(codec, sent)
        assert not got.has_Error()
        assert got.get_StreamId() == W.USER_STRM_BASE_ID and got.get_Tag() == PROT_BASIC
        assert got.get_FormatString() == fmt
        assert got.unpack("%d %d") == -1                   # format mismatch, src/Packet.C:527-532
        assert got.unpack(fmt) != -1
        assert np.array_equal(got.values(0), col), tok
        assert np.array_equal(W.Codec.to_numpy(got.get_SourceRank()).view(np.uint32), np.arange(num_backends))
    # "%s" alone
    sent = PacketWave(codec, W.USER_STRM_BASE_ID, PROT_BASIC, "%s", [string_col("Test String", num_backends)])
    got = received(codec, sent)
    counts, body = got.values(0)
    assert counts.tolist() == [12] * num_backends and body.tobytes() == b"Test String\0" * num_backends


def test_basic_alltypes_packet(codec, golden):
    g = golden["basic_test"]
    n = 8
    cols = []
    for tok, val in zip([t[1:] for t in g["fmt"].split()], g["values"]):
        if tok == "s":
            cols.append(string_col(val, n))
        else:
            cols.append(np.full(n, ord(val) if isinstance(val, str) else val, TOK_DTYPE[tok]))
    sent = PacketWave(codec, W.USER_STRM_BASE_ID, PROT_BASIC, g["fmt"], cols)
    got = received(codec, sent)
    assert got.unpack(g["fmt"]) != -1 and not got.has_Error()
    for i in range(10):
        assert np.array_equal(got.values(i), cols[i])
    # compare_Float(f1, f2, 3 digits) of the reference is implied: the values are bit-identical
    counts, body = got.values(10)
    assert body.tobytes() == b"Test String\0" * n


def test_arrays_round_trip(codec, golden):
    """tests/test_arrays_FE.C: 5,000 elements of every array type; memcmp(sent, received)."""
    a = golden["arrays_test"]
    m, n = a["array_len"], 3
    for tok, val in a["values"].items():
        dt = TOK_DTYPE[tok[1:]]
        body = np.full(n * m, val, dt)
        sent = PacketWave(codec, W.USER_STRM_BASE_ID, PROT_BASIC, "%" + tok, [(np.full(n, m, np.uint64), body)])
        got = received(codec, sent)
        counts, recv = got.values(0)
        assert counts.tolist() == [m] * n and recv.tobytes() == body.tobytes(), tok


@pytest.mark.parametrize("num_backends", [1, 4, 16, 64])
def test_native_filters_sum(codec, golden, num_backends):
    """tests/test_NativeFilters_FE.C:122-262 with TFILTER_SUM + SFILTER_WAITFORALL."""
    g = golden["native_filters_test"]
    for tok, val in g["values"].items():
        dt = np.dtype(TOK_DTYPE[tok])
        wave = PacketWave(codec, W.USER_STRM_BASE_ID, W.FIRST_APPLICATION_TAG + 2, "%" + tok,
                          [np.full(num_backends, val, dt)])
        out = tfilter_Sum(received(codec, wave), shape=W.SHAPE_FLAT)
        assert out.get_StreamId() == W.USER_STRM_BASE_ID and out.get_Tag() == W.FIRST_APPLICATION_TAG + 2
        assert out.get_FormatString() == "%" + tok and out.size() == 1
        comp = np.zeros(1, dt)
        with np.errstate(over="ignore"):
            for _ in range(num_backends):
                comp[0] = comp[0] + dt.type(val)
        assert out.unpack("%" + tok) != -1
        assert out.values(0)[0].tobytes() == comp[0].tobytes(), (tok, num_backends)


def test_filters_min_max_avg_and_rejections(codec):
    v = np.array([123.45, 1e8, -1e8, 0.1, 3.0], np.float32)
    wave = PacketWave(codec, 7, 100, "%f", [v])
    assert tfilter_Max(wave, W.SHAPE_FLAT).values(0)[0] == np.float32(1e8)
    assert tfilter_Min(wave, W.SHAPE_FLAT).values(0)[0] == np.float32(-1e8)
    assert "%f" % tfilter_Sum(wave, W.SHAPE_FLAT).values(0)[0] == "123.099998"
    avg = tfilter_Avg(PacketWave(codec, 7, 100, "%d %d", [np.array([7, 8, 10], np.int32),
                                                          np.array([1, 2, 4], np.int32)]), W.SHAPE_FLAT)
    assert avg.get_FormatString() == "%d %d"
    assert (int(avg.values(0)[0]), int(avg.values(1)[0])) == (9, 7)
    # formats the built-in filters do not accept produce no output packet
    assert tfilter_Sum(PacketWave(codec, 7, 100, "%d %d", [np.zeros(2, np.int32), np.zeros(2, np.int32)])) is None
    assert tfilter_Avg(PacketWave(codec, 7, 100, "%d", [np.zeros(2, np.int32)])) is None
    assert tfilter_Avg(PacketWave(codec, 7, 100, "%d %d", [np.ones(2, np.int32), np.zeros(2, np.int32)])) is None


@pytest.mark.parametrize("n", [4, 1025, 65536])
@pytest.mark.parametrize("fmt", ["%d %d", "%hd %d", "%f", "%lf %d", "%uc %d", "%uld %d", "%lf", "%f %d"])
def test_filter_mirrors_default_to_one_reference_call(codec, fmt, n):
    """tfilter_Sum / Max / Min / Avg called the way a ported tool calls them -- no shape
    argument -- return what ONE reference filter call over the wave returns
    (src/FilterDefinitions.C:101-149, 237-286, 374-424, 517-561): a single left fold, one division
    for Avg.  Bit for bit, every type, also where the order matters (float Sum, every Avg)."""
    from wavegen import random_values, splitmix64
    toks = [t[1:] for t in fmt.split()]
    dt = np.dtype(TOK_DTYPE[toks[0]])
    t = int(W.Format("%" + toks[0]).types[0])
    if dt.kind == "f":
        u = splitmix64(5 * n + 1, n)
        vals = (((u % np.uint64(2_000_001)).astype(np.float64) / 1e3 - 1000.0)
                * np.array([1e-3, 1.0, 1e3, 1e6])[(u >> np.uint64(40)) % np.uint64(4)]).astype(dt)
    else:
        vals = random_values(dt, n, 5 * n + 1)
        if dt.kind == "i":
            vals = np.where(vals == np.iinfo(dt).min, dt.type(1), vals)
    if len(toks) == 2:
        cnts = (splitmix64(n + 3, n) % np.uint64(5) + np.uint64(1)).astype(np.int32)
        wave = received(codec, PacketWave(codec, 7, 100, fmt, [vals, cnts]))
        out = tfilter_Avg(wave)
        ref = O.filter_flat(O.AVG, t, vals, cnts)
        assert out is not None and out.get_FormatString() == fmt
        assert out.values(0)[0].tobytes() == np.asarray(ref[0]).tobytes(), (fmt, n)
        assert int(out.values(1)[0]) == ref[1]
        return
    wave = received(codec, PacketWave(codec, 7, 100, fmt, [vals]))
    for mirror, op in ((tfilter_Sum, O.SUM), (tfilter_Max, O.MAX), (tfilter_Min, O.MIN)):
        out = mirror(wave)
        ref = O.filter_flat(op, t, vals)
        assert out.values(0)[0].tobytes() == np.asarray(ref[0]).tobytes(), (fmt, n, op)


def test_wave_view_reads_fields_in_place(codec):
    """PacketWave.view: the batched, zero-copy form of Packet::unpack (src/Packet.C:1065-1431 is the
    per-packet malloc + memcpy it removes).  A received wave is decoded once; a second kernel (torch
    reductions here) reads the field columns in place -- no device-to-host copy, no new device
    buffer -- under the same format check as unpack; consumers go first, the wave last."""
    from wavegen import make_wave
    fmt_s, n = "%d %lf %af %s", 5000
    ocols = make_wave(fmt_s, n, seed=4, max_count=9)
    sent = PacketWave(codec, 11, 105, fmt_s, [ocols[0]["values"], ocols[1]["values"],
                                               (ocols[2]["counts"], ocols[2]["values"]),
                                               (ocols[3]["counts"], ocols[3]["values"])])
    got = received(codec, sent)
    assert got.view("%d %lf") == -1 and got.view(None) == -1            # as unpack: strcmp on the format
    views = got.view(fmt_s)
    cols = got.unpack(fmt_s)
    # in place: the views are the decoded columns' own memory
    assert views[0].data_ptr() == cols[0].values.data_ptr() and views[0].dtype == torch.int32
    assert views[1].data_ptr() == cols[1].values.data_ptr() and views[1].dtype == torch.float64
    counts, offs, vals = views[2]
    assert vals.data_ptr() == cols[2].values.data_ptr() and vals.dtype == torch.float32
    assert offs.data_ptr() == cols[2].offs.data_ptr()
    # a consumer kernel on the device: per-packet array sums by offsets, scalar reductions
    total = int(offs[n].item())
    seg = torch.zeros(n, dtype=torch.float64, device=codec.device)
    owner = torch.repeat_interleave(torch.arange(n, device=codec.device), counts)
    seg.index_add_(0, owner, vals[:total].to(torch.float64))
    want = np.zeros(n, np.float64)
    np.add.at(want, np.repeat(np.arange(n), ocols[2]["counts"].astype(np.int64)), ocols[2]["values"].astype(np.float64))
    assert np.allclose(seg.cpu().numpy(), want, rtol=1e-12, atol=0)
    assert int(views[0].sum(dtype=torch.int64).item()) == int(ocols[0]["values"].astype(np.int64).sum())
    assert float(views[1].max().item()) == float(ocols[1]["values"].max())
    scnt, soffs, sbytes = views[3]
    assert sbytes.dtype == torch.uint8 and int(soffs[n].item()) == int(ocols[3]["counts"].sum())
    assert bytes(sbytes[: int(soffs[n].item())].cpu().numpy()) == ocols[3]["values"].tobytes()
    del views, counts, offs, vals, scnt, soffs, sbytes              # consumers first ...
    del got                                                         # ... then the wave


def test_receive_side_errors(codec):
    fmt = "%d %ad"
    n = 50
    sent = PacketWave(codec, 9, 101, fmt, [np.arange(n, dtype=np.int32),
                                           (np.full(n, 3, np.uint64), np.arange(3 * n, dtype=np.int32))])
    # a truncated buffer: ERR_PACKING for that packet, the others decode
    lens = sent.get_Buffer().lens.clone()
    lens[10] = 9
    wire = W.WireWave(sent.get_Buffer().arena, n, sent.get_Buffer().offsets, lens, 0)
    got = PacketWave.from_buffers(codec, fmt, n, sent.get_Header(), sent.get_HeaderLen(), wire)
    assert got.unpack(fmt) != -1 and got.has_Error()
    ok = W.Codec.to_numpy(got.ok)
    assert ok[10] == 0 and ok.sum() == n - 1
    # headers that announce another format are refused
    other = PacketWave(codec, 9, 101, "%d %ald", [np.arange(n, dtype=np.int32),
                                                  (np.zeros(n, np.uint64), np.zeros(0, np.int64))])
    bad = PacketWave.from_buffers(codec, fmt, n, other.get_Header(), other.get_HeaderLen(), sent.get_Buffer())
    assert bad.has_Error()


def test_wave_from_children_of_both_byte_orders(codec):
    """Packets of one wave keep the byte order their own message announced (src/Packet.C:96,118):
    a receive-side wave whose odd packets came from big-endian senders unpacks to the same values,
    and the scalar filters fold it."""
    from wavegen import O, to_sender_order
    fmt = "%d %ad"
    n = 40
    vals = np.arange(n, dtype=np.int32) * 7 - 50
    arr = np.arange(3 * n, dtype=np.int32)
    sent = PacketWave(codec, 9, 101, fmt, [vals, (np.full(n, 3, np.uint64), arr)])
    hl = sent.get_HeaderLen()
    hdrs = W.Codec.to_numpy(sent.get_Header()).reshape(n, hl).copy()
    buf = sent.get_Buffer()
    arena = W.Codec.to_numpy(buf.arena).copy()
    offs = W.Codec.to_numpy(buf.offsets).view(np.uint64)
    lens = W.Codec.to_numpy(buf.lens).view(np.uint64)
    be = to_sender_order(fmt, arena, offs, lens)
    flen = len(fmt) + 1
    for i in range(1, n, 2):
        arena[int(offs[i]):int(offs[i] + lens[i])] = be[int(offs[i]):int(offs[i] + lens[i])]
        for a in (0, 4, 8, 12):
            hdrs[i, a:a + 4] = hdrs[i, a:a + 4][::-1].copy()
        hdrs[i, 16 + flen:24 + flen] = hdrs[i, 16 + flen:24 + flen][::-1].copy()
        hdrs[i, hl - 1] = W.BIGENDIAN
    wire = W.WireWave(codec.to_device(arena), n, buf.offsets, buf.lens, 0)
    got = PacketWave.from_buffers(codec, fmt, n, codec.to_device(hdrs.reshape(-1)), hl, wire)
    assert not got.has_Error() and got.get_StreamId() == 9 and got.get_Tag() == 101
    cols = got.unpack(fmt)
    assert cols != -1 and not got.has_Error()
    assert np.array_equal(W.Codec.to_numpy(cols[0].values).view(np.int32), vals)
    assert np.array_equal(W.Codec.to_numpy(cols[1].values).view(np.int32)[:3 * n], arr)
    # a scalar wave of both orders through tfilter_Sum
    s = PacketWave(codec, 9, 101, "%d", [vals])
    h1 = W.Codec.to_numpy(s.get_Header()).reshape(n, s.get_HeaderLen()).copy()
    a1 = W.Codec.to_numpy(s.get_Buffer().arena).copy()[:4 * n].reshape(n, 4)
    for i in range(1, n, 2):
        a1[i] = a1[i][::-1].copy()
        for a in (0, 4, 8, 12):
            h1[i, a:a + 4] = h1[i, a:a + 4][::-1].copy()
        h1[i, 16 + 3:24 + 3] = h1[i, 16 + 3:24 + 3][::-1].copy()
        h1[i, -1] = W.BIGENDIAN
    w1 = W.WireWave(codec.to_device(a1.reshape(-1)), n, None, None, 4)
    g1 = PacketWave.from_buffers(codec, "%d", n, codec.to_device(h1.reshape(-1)), s.get_HeaderLen(), w1)
    out = tfilter_Sum(g1)
    assert out is not None and int(W.Codec.to_numpy(out.unpack("%d")[0].values).view(np.int32)[0]) == int(vals.sum())


@pytest.mark.parametrize("num_backends", [1, 4, 16, 2000])
def test_dynamic_filters(codec, num_backends):
    """tests/test_DynamicFilters_FE.C:94-124, 152-200 with tests/test_DynamicFilters_BE.C:34-47:
    every back-end sends 1 (aggr_Count must return the number of end points) and (odd, even)
    flags of its rank (aggr_CountOddsAndEvens must return how many of each there are)."""
    from mrnet_amd.packet import aggr_Count, aggr_CountOddsAndEvens
    n = num_backends
    ones = PacketWave(codec, 11, 103, "%d", [np.ones(n, dtype=np.int32)])
    got = PacketWave.from_buffers(codec, "%d", n, ones.get_Header(), ones.get_HeaderLen(), ones.get_Buffer())
    out = aggr_Count(got)
    assert out.get_StreamId() == 11 and out.get_Tag() == 103
    assert int(W.Codec.to_numpy(out.unpack("%d")[0].values).view(np.int32)[0]) == n
    ranks = np.arange(n)
    odd, even = (ranks % 2 == 1).astype(np.int32), (ranks % 2 == 0).astype(np.int32)
    flags = PacketWave(codec, 11, 104, "%d %d", [odd, even])
    got = PacketWave.from_buffers(codec, "%d %d", n, flags.get_Header(), flags.get_HeaderLen(), flags.get_Buffer())
    cols = aggr_CountOddsAndEvens(got).unpack("%d %d")
    assert int(W.Codec.to_numpy(cols[0].values).view(np.int32)[0]) == n // 2
    assert int(W.Codec.to_numpy(cols[1].values).view(np.int32)[0]) == (n + 1) // 2
    # mixed scalar types wrap in their own C type, as "sum += v" in the filter's C code does
    mixed = PacketWave(codec, 11, 105, "%uc %hd %lf", [np.full(n, 200, np.uint8), np.full(n, -30000, np.int16),
                                                      np.full(n, 0.5, np.float64)])
    c3 = aggr_Count(mixed).unpack("%uc %hd %lf")
    assert int(W.Codec.to_numpy(c3[0].values).view(np.uint8)[0]) == (200 * n) % 256
    assert int(W.Codec.to_numpy(c3[1].values).view(np.int16)[0]) == np.int16((-30000 * n + 32768) % 65536 - 32768)
    assert float(W.Codec.to_numpy(c3[2].values).view(np.float64)[0]) == 0.5 * n


def test_filter_objects(codec):
    """Filter::push_Packets (src/Filter.C:42-117): the NULL filter hands the wave on, a stateless
    filter emits one packet, PerfData takes its aggregation from the parameter packet, and a
    stateful filter carries its state from wave to wave and reports it through get_FilterState."""
    from mrnet_amd.packet import BUILTIN_FILTERS
    n = 300
    vals = (np.arange(n, dtype=np.int32) * 3) % 17
    wave = PacketWave(codec, 5, 100, "%d", [vals])
    assert BUILTIN_FILTERS["TFILTER_NULL"]().push_Packets(wave) == [wave]
    out = BUILTIN_FILTERS["TFILTER_SUM"]().push_Packets(wave)
    assert len(out) == 1 and int(W.Codec.to_numpy(out[0].unpack("%d")[0].values).view(np.int32)[0]) == int(vals.sum())
    assert BUILTIN_FILTERS["TFILTER_AVG"]().push_Packets(wave) == []          # "%d" is not an Avg format
    # PerfData: aggregation id in the parameter packet
    ranks = (np.ones(4, np.uint64), np.arange(4, dtype=np.int32))
    nel = (np.ones(4, np.uint64), np.full(4, 2, np.int32))
    data = (np.full(4, 2, np.uint64), np.arange(8, dtype=np.float64))
    perf = PacketWave(codec, 5, 100, "%ad %ad %alf", [ranks, nel, data])
    f = BUILTIN_FILTERS["TFILTER_PERFDATA"]()
    assert f.push_Packets(perf) == []                                         # no parameters yet
    f.set_FilterParams(PacketWave(codec, 5, 100, "%d", [np.array([W.TFILTER_MAX], np.int32)]))
    got = f.push_Packets(perf)[0].unpack("%ad %ad %alf")
    assert W.Codec.to_numpy(got[2].values).view(np.float64)[:2].tolist() == [6.0, 7.0]
    assert int(W.Codec.to_numpy(got[0].values).view(np.int32)[0]) == -4
    # a stateful filter: two waves through one Filter object
    fp = BUILTIN_FILTERS["IntegerPercentiles"]()
    assert fp.get_FilterState(5) is None

    def bins(b, mx, mn):
        return PacketWave(codec, 5, 102, "%uld %ud %ud", [np.array(b, np.uint64), np.array(mx, np.uint32),
                                                          np.array(mn, np.uint32)])
    o1 = fp.push_Packets(bins([1, 4], [10, 30], [5, 7]))[0].unpack("%uld %ud %ud")
    assert [int(W.Codec.to_numpy(c.values).view(dt)[0]) for c, dt in zip(o1, (np.uint64, np.uint32, np.uint32))] == [5, 30, 5]
    o2 = fp.push_Packets(bins([8], [20], [2]))[0].unpack("%uld %ud %ud")
    assert [int(W.Codec.to_numpy(c.values).view(dt)[0]) for c, dt in zip(o2, (np.uint64, np.uint32, np.uint32))] == [13, 30, 2]
    st = fp.get_FilterState(9)
    assert st.get_StreamId() == 9 and st.get_Tag() == 102
    sc = st.unpack("%uld %ud %ud")
    assert [int(W.Codec.to_numpy(c.values).view(dt)[0]) for c, dt in zip(sc, (np.uint64, np.uint32, np.uint32))] == [13, 30, 2]


def test_null_string_makes_the_packet_wave_erroneous(codec):
    """Packet's constructor reports ERR_PACKING when encode_pdr_data fails (src/Packet.C:70-72), as
    it does for a NULL string (src/pdr.c:279-282): a '%s' field without bytes."""
    good = PacketWave(codec, 5, 100, "%d %s", [np.arange(3, dtype=np.int32),
                                              (np.array([3, 1, 3], np.uint64), np.frombuffer(b"hi\0\0yo\0", np.uint8))])
    assert not good.has_Error()
    bad = PacketWave(codec, 5, 100, "%d %s", [np.arange(3, dtype=np.int32),
                                             (np.array([3, 0, 3], np.uint64), np.frombuffer(b"hi\0yo\0", np.uint8))])
    assert bad.has_Error()
    assert codec.status() == 0
