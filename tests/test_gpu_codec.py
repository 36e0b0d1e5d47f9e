"""Parity of the HIP codec with the oracle, through the C ABI, on seeded waves.
Bar: bit-exact (byte and index work)."""
import numpy as np
import pytest
import torch

from wavegen import O, columns_equal, make_wave, splitmix64, to_sender_order

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402


def dev_cols(codec, fmt, ocols):
    out = []
    for k, c in zip(fmt.kinds, ocols):
        if k == W.SCALAR:
            out.append(W.Column(codec.to_device(c["values"])))
        elif k == W.STRING_ARRAY:
            vals, sc = c["values"], c["sub_counts"]
            out.append(W.Column(codec.to_device(vals) if vals.size else codec.empty(1, torch.int8),
                                codec.to_device(c["counts"]), codec.to_device(c["offs"]), vals.size,
                                codec.to_device(sc) if sc.size else codec.empty(1, torch.int64),
                                codec.to_device(c["sub_offs"]), sc.size))
        else:
            vals = c["values"]
            out.append(W.Column(codec.to_device(vals) if vals.size else codec.empty(1, torch.int8),
                                codec.to_device(c["counts"]), codec.to_device(c["offs"]),
                                capacity=vals.size))
    return out


def host_cols(fmt, dcols, n):
    out = []
    for i, (k, c) in enumerate(zip(fmt.kinds, dcols)):
        dt = fmt.np_dtype(i)
        d = {"type": int(fmt.types[i])}
        if k == W.SCALAR:
            d["values"] = W.Codec.to_numpy(c.values).view(dt)[:n].copy()
        else:
            d["counts"] = W.Codec.to_numpy(c.counts).view(np.uint64)
            d["offs"] = W.Codec.to_numpy(c.offs).view(np.uint64)
            tot = int(d["offs"][n])
            if k == W.STRING_ARRAY:
                d["sub_counts"] = W.Codec.to_numpy(c.sub_counts).view(np.uint64)[:tot].copy()
                d["sub_offs"] = W.Codec.to_numpy(c.sub_offs).view(np.uint64)[:tot + 1].copy()
                tot = int(d["sub_offs"][tot])
            d["values"] = (W.Codec.to_numpy(c.values).view(dt)[:tot].copy() if tot
                           else np.zeros(0, dt))
        out.append(d)
    return out


def gpu_encode(codec, fmt_s, ocols, n, align=1, packed=None):
    fmt = W.Format(fmt_s)
    wire = codec.encode(fmt, dev_cols(codec, fmt, ocols), n, align=align, packed=packed)
    torch.cuda.synchronize()
    return fmt, wire


def wire_from_numpy(codec, n, arena, offs=None, lens=None, stride=0):
    a = codec.to_device(arena) if arena.size else codec.empty(1, torch.uint8)
    return W.WireWave(a, n, None if offs is None else codec.to_device(offs),
                      None if lens is None else codec.to_device(lens), stride)


VAR_FORMATS = ["%d %uld %lf %s %auc", "%af", "%s", "%ad %ad %alf", "%auc", "%ahd %s %ald %c",
               "%c %uc %hd %uhd %d %ud %ld %uld %f %lf %s",
               "%Ac %Auc %Ahd %Auhd %Ad %Aud %Ald %Auld %Af %Alf", "%lf %alf %f %af %hd %ahd",
               "%as", "%s %as %auhd", "%d %As %lf %as %ad"]
FIXED_FORMATS = ["%d", "%lf %d", "%c %uc %hd %uhd %d %ud %ld %uld %f %lf", "%c", "%hd %c", "%uld",
                 "%f %f %f"]


@pytest.mark.parametrize("fmt_s", VAR_FORMATS)
@pytest.mark.parametrize("n,align", [(0, 1), (1, 1), (5, 16), (257, 1), (1000, 8)])
def test_var_encode_decode_parity(codec, fmt_s, n, align):
    ocols = make_wave(fmt_s, n, seed=n + 13 * align, max_count=37)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n, align=align)
    fmt, wire = gpu_encode(codec, fmt_s, ocols, n, align=align)
    assert np.array_equal(W.Codec.to_numpy(wire.lens).view(np.uint64), lens)
    assert np.array_equal(W.Codec.to_numpy(wire.offsets).view(np.uint64), offs)
    got = W.Codec.to_numpy(wire.arena)[:arena.size]
    assert np.array_equal(got, arena), "encoded bytes differ"
    for bo, src in ((W.LITTLEENDIAN, arena), (W.BIGENDIAN, to_sender_order(fmt_s, arena, offs, lens))):
        ref_cols, ref_ok = O.wave_decode(fmt_s, src, offs, lens, bo)
        dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, src, offs, lens), bo)
        torch.cuda.synchronize()
        assert np.array_equal(W.Codec.to_numpy(ok), ref_ok)
        assert columns_equal(host_cols(fmt, dcols, n), ref_cols), (fmt_s, n, bo)
        assert columns_equal(ref_cols, ocols)
    assert codec.status() == 0


@pytest.mark.parametrize("fmt_s", FIXED_FORMATS)
@pytest.mark.parametrize("n,align", [(0, 1), (1, 1), (3, 4), (256, 1), (257, 16), (5000, 1)])
def test_fixed_encode_decode_parity(codec, fmt_s, n, align):
    ocols = make_wave(fmt_s, n, seed=3 * n + align)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n, align=align)
    # packed layout (LDS-staged rows path)
    fmt, wire = gpu_encode(codec, fmt_s, ocols, n, align=align)
    assert wire.offsets is None and wire.stride == (int(offs[1]) if n else wire.stride)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
    # explicit offsets (generic walker path)
    fmt, wire2 = gpu_encode(codec, fmt_s, ocols, n, align=align, packed=False)
    assert np.array_equal(W.Codec.to_numpy(wire2.arena)[:arena.size], arena)
    assert np.array_equal(W.Codec.to_numpy(wire2.offsets).view(np.uint64), offs)
    for bo, src in ((W.LITTLEENDIAN, arena), (W.BIGENDIAN, to_sender_order(fmt_s, arena, offs, lens))):
        ref_cols, ref_ok = O.wave_decode(fmt_s, src, offs, lens, bo)
        for w in (wire_from_numpy(codec, n, src, stride=wire.stride),
                  wire_from_numpy(codec, n, src, offs, lens)):
            dcols, ok = codec.decode(fmt, w, bo)
            torch.cuda.synchronize()
            assert np.array_equal(W.Codec.to_numpy(ok), ref_ok)
            assert columns_equal(host_cols(fmt, dcols, n), ref_cols), (fmt_s, n, bo)


def test_golden_packet_through_the_gpu(codec, golden):
    g = golden["packet"]
    v = g["values"]
    fmt_s = g["fmt"]

    def arr(t, dt, vals):
        vals = np.asarray(vals, dt)
        return {"type": t, "counts": np.array([vals.size], np.uint64),
                "offs": np.array([0, vals.size], np.uint64), "values": vals}
    ocols = [{"type": O.INT32_T, "values": np.array([v["d"]], np.int32)},
             {"type": O.UINT64_T, "values": np.array([v["uld"]], np.uint64)},
             {"type": O.DOUBLE_T, "values": np.array([v["lf"]], np.float64)},
             arr(O.STRING_T, np.uint8, np.frombuffer(v["s"].encode() + b"\0", np.uint8)),
             arr(O.UCHAR_ARRAY_T, np.uint8, v["auc"]), arr(O.FLOAT_ARRAY_T, np.float32, v["af"])]
    fmt, wire = gpu_encode(codec, fmt_s, ocols, 1)
    assert W.Codec.to_numpy(wire.arena)[:57].tobytes() == bytes.fromhex(g["buf_hex"])
    hdrs, stride = codec.encode_headers(fmt, 1, stream_id=g["stream_id"], tag=g["tag"],
                                        src_rank=g["src_rank"])
    assert stride == 49 and W.Codec.to_numpy(hdrs)[:49].tobytes() == bytes.fromhex(g["hdr_hex"])
    # the captured bytes decode to the captured values
    raw = np.frombuffer(bytes.fromhex(g["buf_hex"]), np.uint8)
    dcols, ok = codec.decode(fmt, wire_from_numpy(codec, 1, raw, np.array([0, 57], np.uint64),
                                                  np.array([57], np.uint64)))
    assert int(ok[0]) == 1 and columns_equal(host_cols(fmt, dcols, 1), ocols)
    # swap known answer: 00 00 01 02 from a big-endian sender is 258
    s = golden["swap"]
    f1 = W.Format(s["fmt"])
    dcols, ok = codec.decode(f1, wire_from_numpy(codec, 1, np.frombuffer(bytes.fromhex(s["bytes_hex"]), np.uint8),
                                                 stride=4), s["sender_byteorder"])
    assert int(ok[0]) == 1 and int(dcols[0].values[0]) == s["value"]


def test_reference_test_values_round_trip(codec, golden):
    """The packets of tests/test_basic_FE.C and tests/test_arrays_FE.C, n back-ends at once."""
    g = golden["basic_test"]
    fmt_s = g["fmt"]
    n = 16
    toks = [t[1:] for t in fmt_s.split()]
    ocols = []
    for tok, val in zip(toks, g["values"]):
        t = O.fmt_types("%" + tok)[0]
        if tok == "s":
            b = np.frombuffer(val.encode() + b"\0", np.uint8)
            cnt = np.full(n, b.size, np.uint64)
            offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(b.size)
            ocols.append({"type": t, "counts": cnt, "offs": offs, "values": np.tile(b, n)})
        else:
            if isinstance(val, str):
                val = ord(val)
            ocols.append({"type": t, "values": np.full(n, val, O.elem_dtype(t))})
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    fmt, wire = gpu_encode(codec, fmt_s, ocols, n)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
    dcols, ok = codec.decode(fmt, wire)
    assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)
    a = golden["arrays_test"]
    for tok, val in a["values"].items():
        fmt_s = "%" + tok
        t = O.fmt_types(fmt_s)[0]
        dt = O.elem_dtype(t)
        m = a["array_len"]
        ocols = [{"type": t, "counts": np.full(4, m, np.uint64),
                  "offs": np.arange(5, dtype=np.uint64) * np.uint64(m), "values": np.full(4 * m, val, dt)}]
        arena, offs, lens = O.wave_encode(fmt_s, ocols, 4)
        fmt, wire = gpu_encode(codec, fmt_s, ocols, 4)
        assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
        be = to_sender_order(fmt_s, arena, offs, lens)
        dcols, ok = codec.decode(fmt, wire_from_numpy(codec, 4, be, offs, lens), W.BIGENDIAN)
        assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, 4), ocols), tok


@pytest.mark.parametrize("fmt_s,n,maxc", [("%af", 37, 20000), ("%auc", 3000, 600), ("%ald", 9, 9000),
                                          ("%ahd %s %alf", 300, 700), ("%auc", 70000, 3),
                                          ("%s %ald %c %af", 40, 9000), ("%c %auc %hd %ahd", 20, 100000),
                                          ("%hd %s %af", 64, 16384)])
def test_segments_crossing_chunks(codec, fmt_s, n, maxc):
    """Arrays longer and shorter than the 16 KiB streaming chunk, odd alignments."""
    ocols = make_wave(fmt_s, n, seed=n, max_count=maxc)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    fmt, wire = gpu_encode(codec, fmt_s, ocols, n)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
    be = to_sender_order(fmt_s, arena, offs, lens)
    dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, be, offs, lens), W.BIGENDIAN)
    assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)


@pytest.mark.parametrize("fmt_s,n,maxc,align", [("%af", 33, 9000, 128), ("%d %s %ald", 100, 300, 64),
                                                 ("%auc %af", 50, 5000, 256), ("%af", 5, 3, 16)])
def test_arena_phase_layout(codec, fmt_s, n, maxc, align):
    """Slots of `align` bytes with the first payload of every packet on an align boundary:
    same packet bytes, different placement."""
    ocols = make_wave(fmt_s, n, seed=align + n, max_count=maxc)
    fmt = W.Format(fmt_s)
    phase = codec.payload_phase(fmt, align)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n, align=align, phase=phase)
    first_var = [i for i, k in enumerate(fmt.kinds) if k != W.SCALAR][0]
    before = sum(fmt.widths[i] for i in range(first_var)) + (4 if fmt.kinds[first_var] == W.STRING else 8)
    assert all((int(o) + before) % align == 0 for o in offs[:-1])
    wire = codec.encode(fmt, dev_cols(codec, fmt, ocols), n, align=align, phase=phase)
    assert np.array_equal(W.Codec.to_numpy(wire.offsets).view(np.uint64), offs)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
    dcols, ok = codec.decode(fmt, wire)
    assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)


def test_zero_length_and_empty(codec):
    fmt_s = "%ad %s %auc"
    n = 6
    ocols = make_wave(fmt_s, n, 5, counts={0: np.zeros(n, np.uint64), 1: np.zeros(n, np.uint64),
                                           2: np.array([0, 3, 0, 0, 1, 0], np.uint64)})
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    fmt, wire = gpu_encode(codec, fmt_s, ocols, n)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
    dcols, ok = codec.decode(fmt, wire)
    assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)
    # all arrays empty
    ocols = make_wave("%af", 50, 1, max_count=0)
    arena, offs, lens = O.wave_encode("%af", ocols, 50)
    fmt, wire = gpu_encode(codec, "%af", ocols, 50)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
    dcols, ok = codec.decode(fmt, wire)
    assert bool(ok.all()) and int(dcols[0].offs[50]) == 0


def test_truncated_packets_fail_per_packet(codec):
    fmt_s = "%d %ad %lf"
    n = 600
    ocols = make_wave(fmt_s, n, 11, max_count=12)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    short = lens.copy()
    cut = (np.arange(n) % 3 == 0)
    short[cut] = (short[cut] * np.uint64(2)) // np.uint64(3)
    short[5] = 0
    bad = arena.copy()
    bad[int(offs[7]) + 4:int(offs[7]) + 12] = np.frombuffer(np.uint64(1 << 40).tobytes(), np.uint8)
    for bo, src in ((W.LITTLEENDIAN, bad), (W.BIGENDIAN, to_sender_order(fmt_s, arena, offs, lens))):
        ref_cols, ref_ok = O.wave_decode(fmt_s, src, offs, short, bo)
        assert 0 < ref_ok.sum() < n
        fmt = W.Format(fmt_s)
        dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, src, offs, short), bo)
        assert np.array_equal(W.Codec.to_numpy(ok), ref_ok)
        assert columns_equal(host_cols(fmt, dcols, n), ref_cols)
    # string arrays: cut inside the strings, inside a length prefix, inside the count
    fmt_s = "%hd %as %d"
    ocols = make_wave(fmt_s, n, 17)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    short = lens.copy()
    for r, k in ((0, 1), (1, 5), (2, 9), (3, 13)):
        sel = (np.arange(n) % 7 == r)
        short[sel] = np.minimum(short[sel], lens[sel] - np.minimum(lens[sel], np.uint64(k)))
    short[6::7] = np.minimum(lens[6::7], np.uint64(6))
    for bo, src in ((W.LITTLEENDIAN, arena), (W.BIGENDIAN, to_sender_order(fmt_s, arena, offs, lens))):
        ref_cols, ref_ok = O.wave_decode(fmt_s, src, offs, short, bo)
        assert 0 < ref_ok.sum() < n
        fmt = W.Format(fmt_s)
        dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, src, offs, short), bo)
        assert np.array_equal(W.Codec.to_numpy(ok), ref_ok)
        assert columns_equal(host_cols(fmt, dcols, n), ref_cols)
    assert codec.status() == 0
    # fixed format, packed rows path, with explicit short lengths
    fmt_s = "%d %lf %hd"
    ocols = make_wave(fmt_s, n, 12)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    short = lens.copy()
    short[::4] = 9
    short[1::4] = 3
    ref_cols, ref_ok = O.wave_decode(fmt_s, arena, offs, short, O.LITTLE)
    fmt = W.Format(fmt_s)
    w = wire_from_numpy(codec, n, arena, None, short, stride=14)
    dcols, ok = codec.decode(fmt, w)
    assert np.array_equal(W.Codec.to_numpy(ok), ref_ok)
    assert columns_equal(host_cols(fmt, dcols, n), ref_cols)


def test_headers_parity(codec):
    fmt_s = "%d %uld %lf %s %auc"
    fmt = W.Format(fmt_s)
    n = 700
    src = np.arange(n, dtype=np.uint32)
    tags = (100 + (np.arange(n) % 7)).astype(np.int32)
    hdrs, stride = codec.encode_headers(fmt, n, stream_id=W.USER_STRM_BASE_ID, tags=codec.to_device(tags),
                                        src_ranks=codec.to_device(src))
    h = W.Codec.to_numpy(hdrs).reshape(n, stride)
    for i in (0, 1, 255, 256, n - 1):
        assert h[i].tobytes() == O.header_encode(W.USER_STRM_BASE_ID, int(tags[i]), int(src[i]), fmt_s).tobytes()
    out = codec.decode_headers(fmt, n, hdrs, stride=stride)
    assert bool(out["ok"].all())
    assert np.array_equal(W.Codec.to_numpy(out["src_rank"]).view(np.uint32), src)
    assert np.array_equal(W.Codec.to_numpy(out["tag"]), tags)
    assert bool((out["stream_id"] == W.USER_STRM_BASE_ID).all()) and bool((out["byteorder"] == 0).all())
    # headers from a big-endian sender, with a destination list, and one foreign format
    raw = []
    for i in range(5):
        hb = O.header_encode(77 + i, -3 - i, 1000 + i, fmt_s if i != 3 else "%d", dest=[1, 2, 3][: i % 4])
        hb = hb.copy()
        flen = len((fmt_s if i != 3 else "%d")) + 1
        for a, w in ((0, 4), (4, 4), (8, 4), (12, 4), (16 + flen, 8)):
            hb[a:a + w] = hb[a:a + w][::-1]
        nd = i % 4
        for k in range(nd):
            a = 16 + flen + 8 + 4 * k
            hb[a:a + 4] = hb[a:a + 4][::-1]
        hb[-1] = 1
        raw.append(hb)
    lens = np.array([len(x) for x in raw], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    out = codec.decode_headers(fmt, 5, codec.to_device(np.concatenate(raw)),
                               offsets=codec.to_device(offs), lens=codec.to_device(lens), max_dest=2)
    dr = W.Codec.to_numpy(out["dest_ranks"])
    assert dr[1, 0] == 1 and dr[2].tolist() == [1, 2]          # byte-reversed ranks, made right
    assert W.Codec.to_numpy(out["ok"]).tolist() == [1, 1, 1, 0, 1]
    assert W.Codec.to_numpy(out["stream_id"]).tolist() == [77, 78, 79, 80, 81]
    assert W.Codec.to_numpy(out["tag"]).tolist() == [-3, -4, -5, -6, -7]
    assert W.Codec.to_numpy(out["ndest"]).tolist()[:3] == [0, 1, 2]
    assert W.Codec.to_numpy(out["byteorder"]).tolist() == [1, 1, 1, 0, 1]


def test_headers_with_destination_list(codec):
    """A downstream wave: every header carries the same destination ranks (src/Packet.C:579-585)."""
    fmt_s = "%d %af"
    fmt = W.Format(fmt_s)
    n = 300
    dest = np.array([3, 9, 4000000000, 7, 11], np.uint32)
    src = np.arange(n, dtype=np.uint32)
    hdrs, stride = codec.encode_headers(fmt, n, stream_id=12, tag=-5, src_ranks=codec.to_device(src),
                                        dest_ranks=codec.to_device(dest))
    assert stride == len(O.header_encode(12, -5, 0, fmt_s, dest=dest.tolist()))
    h = W.Codec.to_numpy(hdrs).reshape(n, stride)
    for i in (0, 1, 255, 256, n - 1):
        assert h[i].tobytes() == O.header_encode(12, -5, int(src[i]), fmt_s, dest=dest.tolist()).tobytes()
    out = codec.decode_headers(fmt, n, hdrs, stride=stride, max_dest=8)
    assert bool(out["ok"].all()) and bool((out["ndest"] == 5).all())
    assert np.array_equal(W.Codec.to_numpy(out["dest_ranks"]).view(np.uint32)[:, :5], np.tile(dest, (n, 1)))


def test_full_size_array_wave_properties(codec):
    """The headline shape: '%af' with 65,536 floats per packet from a big-endian sender.
    Size-independent properties: decode(BE) == element-wise byte reversal; encode(decode(x))
    is the little-endian wire; decode(encode(decode(x))) == decode(x)."""
    n, m = 1024, 65536
    fmt = W.Format("%af")
    g = torch.Generator(device="cuda").manual_seed(3)
    vals = torch.randint(-2**31, 2**31 - 1, (n, m), dtype=torch.int32, device="cuda", generator=g)
    # wire as the big-endian sender wrote it: u64 count then m floats, every item reversed
    stride = 8 + 4 * m
    be = torch.empty((n, stride), dtype=torch.uint8, device="cuda")
    be[:, :8] = torch.tensor(list(int(m).to_bytes(8, "big")), dtype=torch.uint8, device="cuda")
    be[:, 8:] = vals.view(torch.uint8).view(n, m, 4).flip(-1).reshape(n, 4 * m)
    offs = torch.arange(n + 1, dtype=torch.int64, device="cuda") * stride
    lens = torch.full((n,), stride, dtype=torch.int64, device="cuda")
    wire = W.WireWave(be.view(-1), n, offs, lens, 0)
    cols, ok = codec.decode(fmt, wire, W.BIGENDIAN)
    assert bool(ok.all()) and int(cols[0].offs[n]) == n * m
    assert bool((cols[0].counts == m).all())
    assert torch.equal(cols[0].values.view(n, m), vals)
    le = codec.encode(fmt, cols, n)
    assert torch.equal(le.offsets, offs) and torch.equal(le.lens, lens)
    lw = le.arena.view(n, stride)
    assert torch.equal(lw[:, 8:].contiguous().view(torch.int32).view(n, m), vals)
    assert torch.equal(lw[:, :8].contiguous().view(torch.int64).view(-1), torch.full((n,), m, dtype=torch.int64, device="cuda"))
    cols2, ok2 = codec.decode(fmt, le, W.LITTLEENDIAN)
    assert bool(ok2.all()) and torch.equal(cols2[0].values, cols[0].values)
    assert codec.status() == 0


def test_headline_wave_at_its_full_size(codec):
    """The wave the GB/s figure is quoted on, at its real size: 4,096 packets of '%af' with
    65,536 floats each from a big-endian sender (1 GiB of payload) -- decode, sizeof into the
    cache-line phased arena the benchmark uses, encode.  Checked through properties that do not
    depend on the size (decode = element-wise byte reversal of the wire; encode(decode(x)) = the
    little-endian wire of the same values; counts, offsets, lengths), plus three whole packets
    (first, one from the middle, last) through the oracle, byte for byte, both directions."""
    n, m = 4096, 65536
    fmt_s = "%af"
    fmt = W.Format(fmt_s)
    stride = 8 + 4 * m
    g = torch.Generator(device="cuda").manual_seed(33)
    be = torch.empty((n, stride), dtype=torch.uint8, device="cuda")
    be[:, :8] = torch.tensor(list(int(m).to_bytes(8, "big")), dtype=torch.uint8, device="cuda")
    for a in range(0, n, 512):                       # in slices: the generator's scratch stays small
        v = torch.randint(-2**31, 2**31 - 1, (512, m), dtype=torch.int32, device="cuda", generator=g)
        be[a:a + 512, 8:] = v.view(torch.uint8).view(512, m, 4).flip(-1).reshape(512, 4 * m)
        del v
    offs = torch.arange(n + 1, dtype=torch.int64, device="cuda") * stride
    lens = torch.full((n,), stride, dtype=torch.int64, device="cuda")
    wire = W.WireWave(be.view(-1), n, offs, lens, 0)
    cols = [W.Column(torch.empty(n * m, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.int64, device="cuda"),
                     torch.empty(n + 1, dtype=torch.int64, device="cuda"), capacity=n * m)]
    ok = torch.empty(n, dtype=torch.int32, device="cuda")
    codec.decode_into(fmt, wire, cols, ok, W.BIGENDIAN)
    assert bool(ok.all()) and bool((cols[0].counts == m).all())
    assert torch.equal(cols[0].offs, torch.arange(n + 1, dtype=torch.int64, device="cuda") * m)
    vals = cols[0].values.view(n, m)
    for a in range(0, n, 512):
        want = be[a:a + 512, 8:].reshape(512, m, 4).flip(-1).reshape(512, 4 * m).contiguous().view(torch.int32)
        assert torch.equal(vals[a:a + 512], want), "decode is not the element-wise byte reversal"
        del want
    align = 128
    phase = codec.payload_phase(fmt, align)
    slot = (phase + stride + align - 1) // align * align
    out_lens, out_offs, total = codec.sizeof(fmt, cols, n, align, phase)
    assert total == n * slot and bool((out_lens == stride).all())
    assert torch.equal(out_offs[:n], torch.arange(n, dtype=torch.int64, device="cuda") * slot + phase)
    dst = W.WireWave(torch.full((total,), 0xEE, dtype=torch.uint8, device="cuda"), n, out_offs, out_lens, 0)
    codec.encode_into(fmt, cols, dst)
    rows = dst.arena.view(n, slot)
    assert bool((rows[:, phase:phase + 8].contiguous().view(torch.int64) == m).all())
    for a in range(0, n, 512):
        got = rows[a:a + 512, phase + 8:phase + stride].contiguous().view(torch.int32)
        assert torch.equal(got, vals[a:a + 512]), "encode is not the little-endian wire of the decoded values"
        del got
    assert bool((rows[:, :phase] == 0xEE).all()) and bool((rows[:, phase + stride:] == 0xEE).all())
    assert codec.status() == 0
    # whole packets through the oracle
    for i in (0, 2049, n - 1):
        pkt = W.Codec.to_numpy(be[i])
        dec, ook = O.wave_decode(fmt_s, pkt, np.array([0, stride], np.uint64), np.array([stride], np.uint64), O.BIG)
        assert bool(ook.all()) and int(dec[0]["counts"][0]) == m
        assert W.Codec.to_numpy(vals[i]).tobytes() == dec[0]["values"].tobytes(), i
        enc, _, _ = O.wave_encode(fmt_s, dec, 1)
        assert W.Codec.to_numpy(rows[i, phase:phase + stride]).tobytes() == enc.tobytes(), i


@pytest.mark.parametrize("fmt_s,n,giant_every,giant,small", [("%auc", 3000, 100, 1000000, 40),
                                                             ("%ad", 30000, 10000, 7000000, 1),
                                                             ("%hd %s %ahd", 2000, 500, 3000000, 9)])
def test_skewed_segment_sizes(codec, fmt_s, n, giant_every, giant, small):
    """A few huge arrays among many tiny ones: the chunked payload kernel meets chunks holding
    one packet, dozens of packets (boundaries staged in LDS) and thousands of packets."""
    fmt = W.Format(fmt_s)
    f = len(fmt) - 1
    cnt = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) >> np.uint64(7)) % np.uint64(small + 1)
    cnt[giant_every // 2::giant_every] = giant
    ocols = make_wave(fmt_s, n, seed=n, max_count=5, counts={f: cnt})
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    wire = codec.encode(fmt, dev_cols(codec, fmt, ocols), n)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
    be = to_sender_order(fmt_s, arena, offs, lens)
    dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, be, offs, lens), W.BIGENDIAN)
    assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)


def test_short_packet_tiles_leave_gaps_alone_and_take_any_order(codec):
    """Waves of short packets run through LDS-staged tiles of 64 packets.  Encode must write packet
    bytes only (the bytes between packets may be a message's headers); decode must accept packets
    that are out of order or overlapping in the arena (then the tile reads global memory)."""
    fmt_s = "%d %s %ahd %uc %auld"
    n = 1000
    ocols = make_wave(fmt_s, n, 23, max_count=20)
    for align, phase in ((1, 0), (16, 0), (8, 3), (1, 45), (64, 7)):
        arena, offs, lens = O.wave_encode(fmt_s, ocols, n, align=align, phase=phase)
        fmt = W.Format(fmt_s)
        cols = dev_cols(codec, fmt, ocols)
        dlens, doffs, total = codec.sizeof(fmt, cols, n, align, phase)
        assert total == int(offs[n]) and np.array_equal(W.Codec.to_numpy(doffs).view(np.uint64), offs)
        buf = torch.full((max(total, 1) + 5,), 0xEE, dtype=torch.uint8, device=codec.device)
        for shift in (0, 5):                      # arena base 16-byte aligned, and not
            buf.fill_(0xEE)
            wire = W.WireWave(buf[shift:shift + total], n, doffs, dlens, 0)
            codec.encode_into(fmt, cols, wire)
            got = W.Codec.to_numpy(wire.arena)
            mask = np.zeros(total, bool)
            for i in range(n):
                mask[int(offs[i]):int(offs[i] + lens[i])] = True
            assert np.array_equal(got[mask], arena[:total][mask]), (align, phase, shift)
            assert bool((got[~mask] == 0xEE).all()), "bytes between packets were overwritten"
            assert bool((W.Codec.to_numpy(buf)[:shift] == 0xEE).all())
            dcols, ok = codec.decode(fmt, wire)
            assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)
    # packets stored in reverse order, and two packets sharing the same bytes
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    order = np.arange(n)[::-1]
    roffs = np.zeros(n + 1, np.uint64)
    rarena = np.zeros_like(arena)
    pos = 0
    for k in order:
        roffs[k] = pos
        rarena[pos:pos + int(lens[k])] = arena[int(offs[k]):int(offs[k] + lens[k])]
        pos += int(lens[k])
    roffs[n] = pos
    for bo, src in ((W.LITTLEENDIAN, rarena), (W.BIGENDIAN, to_sender_order(fmt_s, rarena, roffs, lens))):
        ref_cols, ref_ok = O.wave_decode(fmt_s, src, roffs, lens, bo)
        dcols, ok = codec.decode(W.Format(fmt_s), wire_from_numpy(codec, n, src, roffs, lens), bo)
        assert np.array_equal(W.Codec.to_numpy(ok), ref_ok) and ref_ok.all()
        assert columns_equal(host_cols(W.Format(fmt_s), dcols, n), ref_cols)
    soffs, slens = offs.copy(), lens.copy()
    soffs[1::2] = soffs[0:-1:2][:len(soffs[1::2])]          # odd packets alias the even ones
    slens[1::2] = lens[0::2][:len(slens[1::2])]
    soffs[n] = offs[n]
    ref_cols, ref_ok = O.wave_decode(fmt_s, arena, soffs, slens, O.LITTLE)
    dcols, ok = codec.decode(W.Format(fmt_s), wire_from_numpy(codec, n, arena, soffs, slens))
    assert np.array_equal(W.Codec.to_numpy(ok), ref_ok)
    assert columns_equal(host_cols(W.Format(fmt_s), dcols, n), ref_cols)
    assert codec.status() == 0


@pytest.mark.parametrize("fmt_s,n,maxc", [("%af", 40, 30000), ("%d %alf %s %ad", 300, 9000), ("%auc %d", 1000, 20),
                                          ("%d %lf", 500, 0), ("%as %af", 50, 4000)])
def test_encode_in_two_calls_on_two_streams(codec, fmt_s, n, maxc):
    """mrpdr_wave_encode_begin (prefixes + plans, on a side stream, with its own ctx) and
    mrpdr_wave_encode_finish (payloads) give the bytes of the one-call encode."""
    ocols = make_wave(fmt_s, n, seed=n + maxc, max_count=maxc)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n, align=16)
    fmt = W.Format(fmt_s)
    cols = dev_cols(codec, fmt, ocols)
    enc = W.Codec(device=codec.device.index, max_packets=1 << 16)
    packed = fmt.is_fixed
    if packed:
        wire = W.WireWave(torch.zeros(n * fmt.fixed_bytes, dtype=torch.uint8, device=codec.device), n, None, None,
                          fmt.fixed_bytes)
        arena, _, _ = O.wave_encode(fmt_s, ocols, n)
    else:
        dl, do, total = enc.sizeof(fmt, cols, n, 16, 0)
        wire = W.WireWave(torch.zeros(total, dtype=torch.uint8, device=codec.device), n, do, dl, 0)
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())
        enc.encode_begin(fmt, cols, wire)
        ev.record(side)
    torch.cuda.current_stream().wait_event(ev)
    enc.encode_finish(fmt, cols, wire)
    torch.cuda.synchronize()
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[: arena.size], arena)
    if not packed and not any(k == W.STRING_ARRAY for k in fmt.kinds) and maxc >= 4000:
        with pytest.raises(W.PdrError):           # finish without a begin
            enc.encode_finish(fmt, cols, wire)
    assert enc.status() == 0


@pytest.mark.parametrize("n,align,phase", [(32769, 1, 0), (100000, 16, 0), (300001, 8, 3), (1 << 20, 1, 45)])
def test_sizeof_large_waves(codec, n, align, phase):
    """Above 32,768 packets sizing and offsets take two launches (workgroup sums, then offsets);
    repeated calls reuse the same workspace."""
    fmt_s = "%d %s %auc"
    fmt = W.Format(fmt_s)
    slen = (splitmix64(n, n) % np.uint64(20) + np.uint64(1)).astype(np.uint64)
    alen = (splitmix64(n + 1, n) % np.uint64(300)).astype(np.uint64)
    so = np.zeros(n + 1, np.uint64); np.cumsum(slen, out=so[1:])
    ao = np.zeros(n + 1, np.uint64); np.cumsum(alen, out=ao[1:])
    cols = [W.Column(codec.empty(n, torch.int32)),
            W.Column(codec.empty(1, torch.int8), codec.to_device(slen), codec.to_device(so), int(so[n])),
            W.Column(codec.empty(1, torch.int8), codec.to_device(alen), codec.to_device(ao), int(ao[n]))]
    want_lens = np.uint64(4 + 4 + 8) + slen + alen
    want_offs = O.scan_offsets(want_lens, align, phase)
    big = W.Codec(device=codec.device.index, max_packets=1 << 20)
    for _ in range(3):
        lens, offs, total = big.sizeof(fmt, cols, n, align, phase)
        assert np.array_equal(W.Codec.to_numpy(lens).view(np.uint64), want_lens)
        assert np.array_equal(W.Codec.to_numpy(offs).view(np.uint64), want_offs)
        assert total == int(want_offs[n])
    assert big.status() == 0


def test_column_beyond_four_gibibytes(codec):
    """'%Ad' packets whose arrays add up to more than 4 GiB (and more chunk descriptors than one
    launch takes): 64-bit offsets everywhere, the payload kernel run in windows.  Checked through
    size-independent properties, as the oracle's per-element path would take minutes here."""
    counts = [600_000_001, 0, 550_000_000, 3]
    n, total = len(counts), sum(counts)
    assert 4 * total > (1 << 32)
    fmt = W.Format("%Ad")
    g = torch.Generator(device="cuda").manual_seed(41)
    vals = torch.randint(-2**31, 2**31 - 1, (total,), dtype=torch.int32, device="cuda", generator=g)
    lens_h = [8 + 4 * c for c in counts]
    offs_h = [0]
    for l in lens_h:
        offs_h.append(offs_h[-1] + l + 5)                  # 5 bytes of slack between packets
    be = torch.zeros(offs_h[-1], dtype=torch.uint8, device="cuda")
    at = 0
    for i, c in enumerate(counts):
        p = offs_h[i]
        be[p:p + 8] = torch.tensor(list(int(c).to_bytes(8, "big")), dtype=torch.uint8, device="cuda")
        step = 1 << 27
        for a in range(0, c, step):                         # in slices: the reversed copy stays small
            b = min(c, a + step)
            be[p + 8 + 4 * a:p + 8 + 4 * b] = vals[at + a:at + b].view(torch.uint8).view(-1, 4).flip(-1).reshape(-1)
        at += c
    offs = torch.tensor(offs_h, dtype=torch.int64, device="cuda")
    lens = torch.tensor(lens_h, dtype=torch.int64, device="cuda")
    wire = W.WireWave(be, n, offs, lens, 0)
    cols, ok = codec.decode(fmt, wire, W.BIGENDIAN)
    assert bool(ok.all()) and int(cols[0].offs[n]) == total
    assert cols[0].counts.tolist() == counts
    assert torch.equal(cols[0].values.view(torch.int32)[:total], vals)
    del be, wire
    le = codec.encode(fmt, cols, n)
    assert le.lens.tolist() == lens_h
    lo = le.offsets.tolist()
    at = 0
    for i, c in enumerate(counts):
        p = lo[i]
        # clones: a view at an odd offset cannot be reinterpreted in place
        assert int(le.arena[p:p + 8].clone().view(torch.int64).item()) == c
        if c:
            assert torch.equal(le.arena[p + 8:p + 8 + 4 * c].clone().view(torch.int32), vals[at:at + c])
        at += c
    assert codec.status() == 0


@pytest.mark.parametrize("n", [5, 40000])
def test_null_string_is_flagged_by_sizeof(codec, n):
    """A '%s' field with no bytes at all stands for the NULL pointer pdr_string refuses to encode
    (src/pdr.c:279-282): sizeof raises status bit 2; the empty string (one NUL) is fine."""
    fmt = W.Format("%d %s")
    cnt = np.full(n, 3, np.uint64)
    cols = codec.columns_from_numpy(fmt, [np.arange(n, dtype=np.int32), (cnt, np.tile(np.frombuffer(b"hi\0", np.uint8), n))])
    codec.sizeof(fmt, cols, n)
    assert codec.status() == 0
    cnt[n // 2] = 0
    body = np.tile(np.frombuffer(b"hi\0", np.uint8), n - 1)
    cols = codec.columns_from_numpy(fmt, [np.arange(n, dtype=np.int32), (cnt, body)])
    codec.sizeof(fmt, cols, n)
    assert codec.status() == 4


def test_array_count_bound_is_flagged_by_sizeof(codec):
    """'%a' arrays stop at INT32_MAX elements (pdr_array's maxsize, src/pdr.c:429 with
    src/Packet.C:647-810); '%A' arrays do not.  sizeof reads only the counts."""
    n = 3
    big = np.array([1, 2**31 + 5, 2], np.uint64)
    for fmt_s, want in (("%auc", 4), ("%Auc", 0)):
        fmt = W.Format(fmt_s)
        cols = codec.alloc_columns(fmt, n, capacities=[16])
        cols[0].counts = codec.to_device(big)
        cols[0].offs = codec.to_device(np.concatenate([[0], np.cumsum(big)]).astype(np.uint64))
        lens, offs, total = codec.sizeof(fmt, cols, n)
        assert total == 3 * 8 + int(big.sum())
        assert codec.status() == want


@pytest.mark.parametrize("fmt_s,n,max_count", [("%d", 65536, 0), ("%d %uld %lf %s %auc", 65536, 300),
                                               ("%lf %d", 131072, 0), ("%ad %ad %alf", 131072, 8),
                                               ("%d", (1 << 19) + 37, 0), ("%lf %d", (1 << 19) + 1023, 0),
                                               ("%uhd %c", 1 << 19, 0), ("%c", (1 << 19) + 5, 0)])
def test_baseline_configs_at_full_size(codec, fmt_s, n, max_count):
    """The benchmark's wave shapes at their full packet counts, byte for byte against the oracle:
    65,536 x '%d', 65,536 x the mixed format, one GPU's 131,072-packet shard of the '%lf %d' and
    of the performance-data wave; and packed scalar waves of more than 2^19 packets, which take the
    four-rows-per-lane form of the rows kernels.  Encode, then decode what a big-endian sender would
    have sent."""
    big = W.Codec(device=codec.device.index, max_packets=1 << 20)
    ocols = make_wave(fmt_s, n, 17, max_count=max_count)
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    fmt = W.Format(fmt_s)
    wire = big.encode(fmt, dev_cols(big, fmt, ocols), n)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[:arena.size], arena)
    if not fmt.is_fixed:
        assert np.array_equal(W.Codec.to_numpy(wire.lens).view(np.uint64), lens)
    be = to_sender_order(fmt_s, arena, offs, lens)
    src = wire_from_numpy(big, n, be, None, None, stride=fmt.fixed_bytes) if fmt.is_fixed \
        else wire_from_numpy(big, n, be, offs, lens)
    dcols, ok = big.decode(fmt, src, W.BIGENDIAN)
    assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols)
    assert big.status() == 0
