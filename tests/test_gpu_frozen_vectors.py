"""The device against the committed vectors of tests/golden/oracle_vectors.json -- data only,
no oracle call for the expected side (inputs are regenerated from the stored seeds)."""
import numpy as np
import pytest
import torch

from wavegen import columns_equal, make_wave, random_values, to_sender_order
from test_gpu_codec import dev_cols, host_cols, wire_from_numpy

pytestmark = pytest.mark.gpu

from mrnet_amd import wave as W  # noqa: E402

TREE = {"flat": W.SHAPE_FLAT, "tree": W.SHAPE_TREE}
OPS = {"sum": W.TFILTER_SUM, "min": W.TFILTER_MIN, "max": W.TFILTER_MAX, "avg": W.TFILTER_AVG}


def unhex(h):
    return np.frombuffer(bytes.fromhex(h), np.uint8)


def check_wave(codec, case):
    fmt_s, n = case["fmt"], case["n"]
    ocols = make_wave(fmt_s, n, case["seed"], max_count=case["max_count"])
    fmt = W.Format(fmt_s)
    lens = np.array(case["lens"], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    wire = codec.encode(fmt, dev_cols(codec, fmt, ocols), n, packed=False)
    torch.cuda.synchronize()
    le = unhex(case["le_hex"])
    assert np.array_equal(W.Codec.to_numpy(wire.lens).view(np.uint64), lens)
    assert np.array_equal(W.Codec.to_numpy(wire.arena)[: le.size], le), fmt_s
    for bo, h in ((W.LITTLEENDIAN, case["le_hex"]), (W.BIGENDIAN, case["be_hex"])):
        dcols, ok = codec.decode(fmt, wire_from_numpy(codec, n, unhex(h), offs, lens), bo)
        assert bool(ok.all()) and columns_equal(host_cols(fmt, dcols, n), ocols), (fmt_s, bo)


def test_every_token_and_mixed_packets(codec, frozen):
    for case in frozen["tokens"] + frozen["mixed"]:
        check_wave(codec, case)
    assert codec.status() == 0


def test_truncation_at_every_length(codec, frozen):
    t = frozen["truncation"]
    fmt_s = t["fmt"]
    ocols = make_wave(fmt_s, 1, t["seed"], max_count=t["max_count"])
    fmt = W.Format(fmt_s)
    wire = codec.encode(fmt, dev_cols(codec, fmt, ocols), 1, packed=False)
    full = W.Codec.to_numpy(wire.arena)[: int(wire.lens[0])]
    m = full.size + 1
    # m packets that all start at byte 0 and are cut after 0, 1, ... bytes
    lens = np.arange(m, dtype=np.uint64)
    offs = np.zeros(m + 1, np.uint64)
    _, ok = codec.decode(fmt, wire_from_numpy(codec, m, full, offs, lens), W.LITTLEENDIAN)
    assert W.Codec.to_numpy(ok).tolist() == t["ok_le"]
    # the same packet as a big-endian sender wrote it, cut at every length, decoded with reversal
    be = to_sender_order(fmt_s, full, np.array([0, full.size], np.uint64), np.array([full.size], np.uint64))
    _, ok = codec.decode(fmt, wire_from_numpy(codec, m, be, offs, lens), W.BIGENDIAN)
    assert W.Codec.to_numpy(ok).tolist() == t["ok_be"]


def test_three_packet_message(codec, frozen):
    g = frozen["message"]
    fmt = W.Format(g["fmt"])
    ocols = make_wave(g["fmt"], g["n"], g["seed"], max_count=g["max_count"])
    src = codec.to_device(np.arange(g["n"], dtype=np.uint32))
    msg = codec.message_build(fmt, dev_cols(codec, fmt, ocols), g["n"], stream_id=g["stream_id"], tag=g["tag"],
                              src_ranks=src)
    assert np.array_equal(W.Codec.to_numpy(msg), unhex(g["hex"]))
    n, hdr, wire = codec.message_parse(fmt, codec.to_device(unhex(g["hex"])))
    dcols, ok = codec.decode(fmt, wire)
    assert n == g["n"] and bool(ok.all()) and bool(hdr["ok"].all())
    assert columns_equal(host_cols(fmt, dcols, n), ocols)


def test_filters_all_types(codec, frozen):
    for c in frozen["filters"]:
        fmt = W.Format("%" + c["tok"])
        dt = fmt.np_dtype(0)
        vals = random_values(dt, c["n"], c["seed"])
        cnt = ((np.arange(c["n"], dtype=np.int32) % 5) + 1) if c["op"] == "avg" else None
        for shape in ("flat", "tree"):
            res, ok = codec.filter_columns(OPS[c["op"]], fmt.types[0], codec.to_device(vals),
                                           None if cnt is None else codec.to_device(cnt), TREE[shape])
            want = unhex(c[shape + "_hex"])
            got = W.Codec.to_numpy(res)[: want.size]
            assert int(ok.item()) == 1 and np.array_equal(got, want), (c["tok"], c["n"], c["op"], shape)
            if c["op"] == "avg":           # the count follows the value, as in the "<T> %d" output packet
                got_cnt = int(W.Codec.to_numpy(res)[want.size: want.size + 4].view(np.int32)[0])
                assert got_cnt == c[shape + "_count"]
