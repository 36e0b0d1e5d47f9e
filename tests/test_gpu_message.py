"""Message framing (src/Message.C): a wave encoded in place inside one message buffer, and the
reverse.  Bar: bit-exact against the oracle's message bytes."""
import numpy as np
import pytest
import torch

from wavegen import O, columns_equal, make_wave
from test_gpu_codec import dev_cols, host_cols

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402

FORMATS = ["%d", "%lf %d", "%d %s %auc", "%af", "%s", "%c %ahd %uld", "%hd %c"]


def oracle_message(fmt_s, ocols, n, stream_id, tag, ranks):
    arena, offs, lens = O.wave_encode(fmt_s, ocols, n)
    hdrs = [O.header_encode(stream_id, tag, int(ranks[i]), fmt_s) for i in range(n)]
    h = len(hdrs[0]) if n else 0
    harena = np.concatenate(hdrs) if n else np.zeros(0, np.uint8)
    return O.message_build(harena, np.arange(n, dtype=np.uint64) * np.uint64(h),
                           np.full(n, h, np.uint64), arena, offs[:n], lens), arena, offs, lens


@pytest.mark.parametrize("fmt_s", FORMATS)
@pytest.mark.parametrize("n", [0, 1, 3, 300, 4097])
def test_message_build_matches_oracle(codec, fmt_s, n):
    ocols = make_wave(fmt_s, n, seed=5 * n + len(fmt_s), max_count=41)
    ranks = (np.arange(n, dtype=np.uint32) * 7 + 3).astype(np.uint32)
    ref, _, _, _ = oracle_message(fmt_s, ocols, n, 1 << 30, 123, ranks)
    fmt = W.Format(fmt_s)
    msg = codec.message_build(fmt, dev_cols(codec, fmt, ocols), n, stream_id=1 << 30, tag=123,
                              src_ranks=codec.to_device(ranks) if n else None)
    torch.cuda.synchronize()
    got = W.Codec.to_numpy(msg)
    assert got.size == ref.size
    assert np.array_equal(got, ref), "message bytes differ"     # spare byte: 0 on both sides
    assert codec.status() == 0


@pytest.mark.parametrize("fmt_s", FORMATS)
@pytest.mark.parametrize("n", [0, 1, 3, 300, 4097])
def test_message_parse_and_decode(codec, fmt_s, n):
    ocols = make_wave(fmt_s, n, seed=11 * n + len(fmt_s), max_count=29)
    ranks = (np.arange(n, dtype=np.uint32) + 100).astype(np.uint32)
    ref, arena, offs, lens = oracle_message(fmt_s, ocols, n, (1 << 30) + 2, -7, ranks)
    parsed = O.message_parse(ref)
    fmt = W.Format(fmt_s)
    cnt, hdr, wire = codec.message_parse(fmt, codec.to_device(ref))
    assert cnt == n == parsed[0]
    pre = 6 + 16 * n
    assert np.array_equal(W.Codec.to_numpy(wire.offsets).view(np.uint64)[:n] + np.uint64(pre), parsed[3])
    assert np.array_equal(W.Codec.to_numpy(wire.lens).view(np.uint64), parsed[4])
    assert W.Codec.to_numpy(hdr["ok"]).tolist() == [1] * n
    assert np.array_equal(W.Codec.to_numpy(hdr["src_rank"]).view(np.uint32), ranks)
    assert np.array_equal(W.Codec.to_numpy(hdr["tag"]), np.full(n, -7, np.int32))
    assert np.array_equal(W.Codec.to_numpy(hdr["stream_id"]).view(np.uint32),
                          np.full(n, (1 << 30) + 2, np.uint32))
    dcols, ok = codec.decode(fmt, wire, W.LITTLEENDIAN)
    torch.cuda.synchronize()
    assert W.Codec.to_numpy(ok).tolist() == [1] * n
    assert columns_equal(host_cols(fmt, dcols, n), ocols)


def test_message_round_trip_large(codec):
    """Full-size property: build -> parse -> decode returns the columns (1024 packets x 64 KiB)."""
    n, cnt = 1024, 16384
    fmt = W.Format("%af")
    vals = torch.arange(n * cnt, dtype=torch.float32, device=codec.device)
    counts = torch.full((n,), cnt, dtype=torch.int64, device=codec.device)
    offs = torch.arange(n + 1, dtype=torch.int64, device=codec.device) * cnt
    msg = codec.message_build(fmt, [W.Column(vals, counts, offs, capacity=n * cnt)], n)
    got_n, hdr, wire = codec.message_parse(fmt, msg)
    assert got_n == n and int(hdr["ok"].sum()) == n
    dcols, ok = codec.decode(fmt, wire, W.LITTLEENDIAN)
    assert int(ok.sum()) == n
    assert torch.equal(dcols[0].values[:n * cnt].view(torch.float32), vals)
    assert torch.equal(dcols[0].counts.view(torch.int64), counts)


def test_message_truncated_is_rejected(codec):
    fmt_s = "%d %s"
    ocols = make_wave(fmt_s, 4, seed=2, max_count=9)
    ref, _, _, _ = oracle_message(fmt_s, ocols, 4, 1 << 30, 100, np.zeros(4, np.uint32))
    with pytest.raises(W.PdrError):
        codec.message_parse(W.Format(fmt_s), codec.to_device(ref[:-3]))
    assert O.message_parse(ref[:-3]) is None


import os  # noqa: E402

from test_gpu_random_formats import draw  # noqa: E402

SOAK = max(1, int(os.environ.get("MRPDR_SOAK", "1")))


@pytest.mark.parametrize("seed", range(30 * SOAK))
def test_random_messages(codec, seed):
    """Drawn format string and wave size: the message built on the device equals the oracle's byte
    for byte, and the oracle's message parses and decodes back to the columns."""
    fmt_s, n, max_count, _, _ = draw(15000 + seed)
    if n * max_count > 200000:
        max_count = 40
    ocols = make_wave(fmt_s, n, seed=seed + 3, max_count=max_count)
    ranks = (np.arange(n, dtype=np.uint32) * 3 + seed).astype(np.uint32)
    ref, arena, offs, lens = oracle_message(fmt_s, ocols, n, (1 << 30) + seed, 100 + seed, ranks)
    fmt = W.Format(fmt_s)
    msg = codec.message_build(fmt, dev_cols(codec, fmt, ocols), n, stream_id=(1 << 30) + seed, tag=100 + seed,
                              src_ranks=codec.to_device(ranks) if n else None)
    torch.cuda.synchronize()
    assert np.array_equal(W.Codec.to_numpy(msg), ref), (fmt_s, n)
    cnt, hdr, wire = codec.message_parse(fmt, codec.to_device(ref))
    assert cnt == n and W.Codec.to_numpy(hdr["ok"]).tolist() == [1] * n
    assert np.array_equal(W.Codec.to_numpy(hdr["src_rank"]).view(np.uint32), ranks)
    dcols, ok = codec.decode(fmt, wire, W.LITTLEENDIAN)
    assert W.Codec.to_numpy(ok).tolist() == [1] * n
    assert columns_equal(host_cols(fmt, dcols, n), ocols), (fmt_s, n)
    assert codec.status() == 0


def test_packet_wave_through_a_message(codec):
    """PacketWave.from_message / to_message: a wave taken out of the oracle's message and sent on
    leaves as the same bytes."""
    from mrnet_amd.packet import PacketWave
    fmt_s, n = "%d %s %auhd", 321
    ocols = make_wave(fmt_s, n, seed=8, max_count=11)
    ranks = (np.arange(n, dtype=np.uint32) + 9).astype(np.uint32)
    ref, _, _, _ = oracle_message(fmt_s, ocols, n, (1 << 30) + 1, 117, ranks)
    wave = PacketWave.from_message(codec, fmt_s, codec.to_device(ref))
    assert wave.size() == n and not wave.has_Error() and wave.get_StreamId() == (1 << 30) + 1 and wave.get_Tag() == 117
    assert columns_equal(host_cols(wave.fmt, wave.unpack(fmt_s), n), ocols)
    out = wave.to_message()
    torch.cuda.synchronize()
    assert np.array_equal(W.Codec.to_numpy(out), ref)


@pytest.mark.parametrize("n", [1, 1023, 1025, 70001])
def test_one_call_framing_equals_the_separate_calls(codec, n):
    """mrpdr_message_write_frame / mrpdr_message_parse against write_preamble + encode_headers and
    parse_sizes + decode_headers: the same bytes, offsets, lengths and header fields, with waves
    that span many of their 1,024-packet workgroups and sizes that differ from packet to packet."""
    import ctypes as C
    from mrnet_amd.wave import _ptr, _stream
    fmt = W.Format("%d %s")
    ocols = make_wave("%d %s", n, seed=n, max_count=23)
    cols = dev_cols(codec, fmt, ocols)
    ranks = codec.to_device((np.arange(n, dtype=np.uint32) * 5 + 1).astype(np.uint32))
    msg = codec.message_build(fmt, cols, n, stream_id=(1 << 30) + 9, tag=77, src_ranks=ranks)
    # the separate calls, into a zeroed buffer of the same size
    h = fmt.header_bytes
    lens, offsets, payload = codec.sizeof(fmt, cols, n, 1, h)
    pre = 6 + 16 * n
    assert msg.numel() == pre + payload
    sep = torch.zeros_like(msg)
    lib = codec._lib
    assert lib.mrpdr_message_write_preamble(codec.handle, n, h, _ptr(lens), _ptr(sep), _stream())
    codec.encode_into(fmt, cols, W.WireWave(sep[pre:], n, offsets, lens, 0))
    codec.encode_headers(fmt, n, stream_id=(1 << 30) + 9, tag=77, src_ranks=ranks, out=sep,
                         out_ptr=sep.data_ptr() + pre - h, offsets=offsets)
    assert torch.equal(msg, sep)
    # and back
    cnt, hdr, wire = codec.message_parse(fmt, msg, n=n, check_length=False)
    ho, bo = codec.empty(n + 1, torch.int64), codec.empty(n + 1, torch.int64)
    hl, bl = codec.empty(n, torch.int64), codec.empty(n, torch.int64)
    total = C.c_uint64(0)
    assert lib.mrpdr_message_parse_sizes(codec.handle, msg.data_ptr() + 5, n, _ptr(ho), _ptr(hl), _ptr(bo), _ptr(bl),
                                         C.byref(total), _stream())
    assert total.value == payload
    assert torch.equal(wire.offsets[:n + 1], bo) and torch.equal(wire.lens[:n], bl)
    assert torch.equal(bo[:n], offsets[:n]) and torch.equal(bl, lens) and int(ho[n]) == payload
    sep_hdr = codec.decode_headers(fmt, n, msg[pre:], offsets=ho, lens=hl)
    for k in ("stream_id", "tag", "src_rank", "byteorder", "ndest", "ok"):
        assert torch.equal(hdr[k], sep_hdr[k]), k
    assert int(hdr["ok"].sum()) == n and torch.equal(hdr["src_rank"].view(torch.int32), ranks.view(torch.int32))
    dcols, ok = codec.decode(fmt, wire, W.LITTLEENDIAN)
    assert int(ok.sum()) == n and columns_equal(host_cols(fmt, dcols, n), ocols)


@pytest.mark.parametrize("n", [700, 4097])
@pytest.mark.parametrize("giant", [False, True])
def test_message_of_short_mixed_packets(codec, n, giant):
    """The mixed short-packet format inside a message: with the 49-byte headers between the data
    buffers the arena holds 320-448 bytes per packet, the range of the tile kernels' wider LDS
    image.  With `giant`, one packet is far longer than the image: its tile (and only its tile)
    runs on global memory.  Build, parse and decode against the oracle, byte for byte."""
    fmt_s = "%d %uld %lf %s %auc"
    rng = np.random.default_rng(n + giant)
    counts = {3: rng.integers(0, 24, n).astype(np.uint64), 4: rng.integers(200, 420, n).astype(np.uint64)}
    if giant:
        counts[4][n // 2] = 70000
    ocols = make_wave(fmt_s, n, seed=n, counts=counts)
    ranks = np.arange(n, dtype=np.uint32)
    ref, arena, offs, lens = oracle_message(fmt_s, ocols, n, 1 << 30, 5, ranks)
    per_packet = (ref.size - (6 + 16 * n)) / n
    assert 320 < per_packet <= 448 or giant
    fmt = W.Format(fmt_s)
    cols = dev_cols(codec, fmt, ocols)
    msg = codec.message_build(fmt, cols, n, stream_id=1 << 30, tag=5, src_ranks=codec.to_device(ranks))
    torch.cuda.synchronize()
    assert np.array_equal(W.Codec.to_numpy(msg), ref)
    ws = codec.message_workspace(n)
    cnt, hdr, wire = codec.message_parse(fmt, codec.to_device(ref), n=n, check_length=False, workspace=ws)
    # both decode forms: the walk that stages the tiles (begin + finish) and the one call
    dcols, ok = codec.decode(fmt, wire, W.LITTLEENDIAN)
    assert int(ok.sum()) == n and int(hdr["ok"].sum()) == n
    assert columns_equal(host_cols(fmt, dcols, n), ocols)
    caps = [None, None, None, int(ocols[3]["offs"][-1]), int(ocols[4]["offs"][-1])]
    d2, ok2 = codec.alloc_columns(fmt, n, caps), codec.empty(n, torch.int32)
    codec.decode_into(fmt, wire, d2, ok2)
    assert int(ok2.sum()) == n and columns_equal(host_cols(fmt, d2, n), ocols)
    assert codec.status() == 0


def test_downstream_message_with_destination_lists(codec):
    """A downstream message: every header carries a destination list (src/Packet.C:579-585), so the
    headers are longer than mrpdr_header_bytes.  Built from the separate calls (sizeof with the longer
    phase, preamble, headers with dest_ranks, data), checked against the oracle's bytes, and taken
    apart by the one-call parse with room for the destinations."""
    fmt_s, n = "%d %auc", 1500
    ocols = make_wave(fmt_s, n, seed=77, max_count=19)
    dest = np.array([4, 8, 15, 16], np.uint32)
    fmt = W.Format(fmt_s)
    cols = dev_cols(codec, fmt, ocols)
    h = len(O.header_encode(1 << 30, 3, 0, fmt_s, dest=dest.tolist()))
    assert h == fmt.header_bytes + 4 * dest.size
    lens, offsets, payload = codec.sizeof(fmt, cols, n, 1, h)
    pre = 6 + 16 * n
    msg = torch.zeros(pre + payload, dtype=torch.uint8, device=codec.device)
    from mrnet_amd.wave import _ptr, _stream
    assert codec._lib.mrpdr_message_write_preamble(codec.handle, n, h, _ptr(lens), _ptr(msg), _stream())
    codec.encode_into(fmt, cols, W.WireWave(msg[pre:], n, offsets, lens, 0))
    ranks = codec.to_device(np.arange(n, dtype=np.uint32))
    codec.encode_headers(fmt, n, stream_id=1 << 30, tag=3, src_ranks=ranks, out=msg, out_ptr=msg.data_ptr() + pre - h,
                         offsets=offsets, dest_ranks=codec.to_device(dest))
    # the oracle's message of the same packets
    arena, aoffs, alens = O.wave_encode(fmt_s, ocols, n)
    hdrs = np.concatenate([O.header_encode(1 << 30, 3, i, fmt_s, dest=dest.tolist()) for i in range(n)])
    ref = O.message_build(hdrs, np.arange(n, dtype=np.uint64) * np.uint64(h), np.full(n, h, np.uint64), arena, aoffs[:n], alens)
    torch.cuda.synchronize()
    assert np.array_equal(W.Codec.to_numpy(msg), ref)
    ws = codec.message_workspace(n, max_dest=6)
    cnt, hdr, wire = codec.message_parse(fmt, msg, n=n, workspace=ws)
    assert int(hdr["ok"].sum()) == n and bool((hdr["ndest"] == dest.size).all())
    assert np.array_equal(W.Codec.to_numpy(hdr["dest_ranks"]).view(np.uint32)[:, :4], np.tile(dest, (n, 1)))
    assert torch.equal(hdr["src_rank"].view(torch.int32), ranks.view(torch.int32))
    dcols, ok = codec.decode(fmt, wire, W.LITTLEENDIAN)
    assert int(ok.sum()) == n and columns_equal(host_cols(fmt, dcols, n), ocols)
