"""The oracle against everything that pins it: byte vectors captured from the compiled
reference, and the values / expectations the reference's own tests hold
(tests/test_basic_FE.C, tests/test_arrays_FE.C, tests/test_NativeFilters_FE.C)."""
import numpy as np
import pytest

from wavegen import O, columns_equal, make_wave, to_sender_order

TOK_DTYPE = {"c": np.int8, "uc": np.uint8, "hd": np.int16, "uhd": np.uint16, "d": np.int32,
             "ud": np.uint32, "ld": np.int64, "uld": np.uint64, "f": np.float32, "lf": np.float64}
TOK_TYPE = {"c": O.CHAR_T, "uc": O.UCHAR_T, "hd": O.INT16_T, "uhd": O.UINT16_T, "d": O.INT32_T,
            "ud": O.UINT32_T, "ld": O.INT64_T, "uld": O.UINT64_T, "f": O.FLOAT_T, "lf": O.DOUBLE_T}


def arr_col(t, dtype, values):
    values = np.asarray(values, dtype=dtype)
    return {"type": t, "counts": np.array([values.size], np.uint64),
            "offs": np.array([0, values.size], np.uint64), "values": values}


def str_col(s):
    b = np.frombuffer(s.encode() + b"\0", dtype=np.uint8)
    return arr_col(O.STRING_T, np.uint8, b)


def test_packet_bytes_match_reference_capture(golden):
    g = golden["packet"]
    v = g["values"]
    cols = [{"type": O.INT32_T, "values": np.array([v["d"]], np.int32)},
            {"type": O.UINT64_T, "values": np.array([v["uld"]], np.uint64)},
            {"type": O.DOUBLE_T, "values": np.array([v["lf"]], np.float64)},
            str_col(v["s"]),
            arr_col(O.UCHAR_ARRAY_T, np.uint8, v["auc"]),
            arr_col(O.FLOAT_ARRAY_T, np.float32, v["af"])]
    arena, offs, lens = O.wave_encode(g["fmt"], cols, 1)
    assert arena.tobytes() == bytes.fromhex(g["buf_hex"])
    hdr = O.header_encode(g["stream_id"], g["tag"], g["src_rank"], g["fmt"])
    assert hdr.tobytes() == bytes.fromhex(g["hdr_hex"])
    # and back
    dec, ok = O.wave_decode(g["fmt"], arena, offs, lens, O.LITTLE)
    assert ok.tolist() == [1] and columns_equal(dec, cols)
    h = O.header_decode(np.frombuffer(bytes.fromhex(g["hdr_hex"]), np.uint8))
    assert (h["stream_id"], h["tag"], h["src_rank"], h["fmt"], h["byteorder"]) == \
        (g["stream_id"], g["tag"], g["src_rank"], g["fmt"], 0)
    assert h["dest"].size == 0


def test_swap_known_answer(golden):
    g = golden["swap"]
    raw = np.frombuffer(bytes.fromhex(g["bytes_hex"]), np.uint8)
    cols, ok = O.wave_decode(g["fmt"], raw, np.array([0, 4], np.uint64), np.array([4], np.uint64),
                             g["sender_byteorder"])
    assert ok[0] == 1 and int(cols[0]["values"][0]) == g["value"]


def test_filter_known_answers(golden):
    ops = {"sum": O.SUM, "min": O.MIN, "max": O.MAX, "avg": O.AVG}
    for case in golden["filters"]:
        tok = case["fmt"].split()[0][1:]
        vals = np.asarray(case["values"], dtype=TOK_DTYPE[tok])
        cnt = np.asarray(case["counts"], np.int32) if "counts" in case else None
        r, c = O.filter_flat(ops[case["op"]], TOK_TYPE[tok], vals, cnt)
        if "printf_f" in case:
            assert "%f" % r == case["printf_f"]
        else:
            assert (int(r), c) == (case["value"], case["count"])


def test_native_filters_sum_expectation(golden):
    """tests/test_NativeFilters_FE.C:122-262: comp += VAL, num_backends times, in the type."""
    g = golden["native_filters_test"]
    for tok, val in g["values"].items():
        dt = np.dtype(TOK_DTYPE[tok])
        for nb in g["num_backends"]:
            comp = np.zeros(1, dtype=dt)
            with np.errstate(over="ignore"):
                for _ in range(nb):
                    comp[0] = comp[0] + dt.type(val)
            r, _ = O.filter_flat(O.SUM, TOK_TYPE[tok], np.full(nb, val, dtype=dt))
            assert r.tobytes() == comp[0].tobytes(), (tok, nb)


def test_basic_alltypes_round_trip(golden):
    """tests/test_basic_FE.C:1345-1388: pack the alltypes packet, unpack, compare."""
    g = golden["basic_test"]
    v = g["values"]
    toks = [t[1:] for t in g["fmt"].split()]
    cols = []
    for tok, val in zip(toks, v):
        if tok == "s":
            cols.append(str_col(val))
        else:
            if isinstance(val, str):
                val = ord(val)
            cols.append({"type": TOK_TYPE[tok], "values": np.array([val], dtype=TOK_DTYPE[tok])})
    arena, offs, lens = O.wave_encode(g["fmt"], cols, 1)
    # 1+1+2+2+4+4+8+8+4+8 fixed bytes, then u32 12 + "Test String\0"
    assert int(lens[0]) == 42 + 4 + 12
    assert arena[:2].tobytes() == b"AB"
    assert arena[42:46].view(np.uint32)[0] == 12 and arena[46:].tobytes() == b"Test String\0"
    for bo, wire in ((O.LITTLE, arena), (O.BIG, to_sender_order(g["fmt"], arena, offs, lens))):
        dec, ok = O.wave_decode(g["fmt"], wire, offs, lens, bo)
        assert ok[0] == 1 and columns_equal(dec, cols)


def test_arrays_round_trip(golden):
    """tests/test_arrays_FE.C: ARRAY_LEN elements of one value, memcmp of sent vs received."""
    g = golden["arrays_test"]
    n = g["array_len"]
    for tok, val in g["values"].items():
        fmt = "%" + tok
        t = O.fmt_types(fmt)[0]
        dt = O.elem_dtype(t)
        cols = [arr_col(t, dt, np.full(n, val, dtype=dt))]
        arena, offs, lens = O.wave_encode(fmt, cols, 1)
        assert int(lens[0]) == 8 + n * dt.itemsize
        assert arena[:8].view(np.uint64)[0] == n
        assert arena[8:].tobytes() == cols[0]["values"].tobytes()
        for bo, wire in ((O.LITTLE, arena), (O.BIG, to_sender_order(fmt, arena, offs, lens))):
            dec, ok = O.wave_decode(fmt, wire, offs, lens, bo)
            assert ok[0] == 1 and columns_equal(dec, cols), (tok, bo)
        # %A* is the same on the wire (src/Packet.C:698-707)
        big = "%A" + tok[1:]
        cols[0]["type"] = O.fmt_types(big)[0]
        arena2, _, _ = O.wave_encode(big, cols, 1)
        assert arena2.tobytes() == arena.tobytes()


def test_every_token_alone_and_mixed():
    toks = ["c", "uc", "hd", "uhd", "d", "ud", "ld", "uld", "f", "lf", "s", "ac", "auc", "ahd",
            "auhd", "ad", "aud", "ald", "auld", "af", "alf", "as", "Ac", "Auc", "Ahd", "Auhd",
            "Ad", "Aud", "Ald", "Auld", "Af", "Alf", "As"]
    for k, tok in enumerate(toks):
        fmt = "%" + tok
        cols = make_wave(fmt, 17, 100 + k, max_count=9)
        arena, offs, lens = O.wave_encode(fmt, cols, 17)
        for bo, wire in ((O.LITTLE, arena), (O.BIG, to_sender_order(fmt, arena, offs, lens))):
            dec, ok = O.wave_decode(fmt, wire, offs, lens, bo)
            assert ok.all() and columns_equal(dec, cols), (tok, bo)
    fmt = " ".join("%" + t for t in toks)
    cols = make_wave(fmt, 9, 7, max_count=5)
    arena, offs, lens = O.wave_encode(fmt, cols, 9, align=16)
    assert all(int(o) % 16 == 0 for o in offs)
    dec, ok = O.wave_decode(fmt, arena, offs, lens, O.LITTLE)
    assert ok.all() and columns_equal(dec, cols)


def test_format_delimiters_and_unknown_tokens():
    assert O.fmt_types("%d %uld\t%lf\n%s") == [O.INT32_T, O.UINT64_T, O.DOUBLE_T, O.STRING_T]
    assert O.fmt_types("%d%d") == [O.INT32_T, O.INT32_T]      # '%' is a delimiter too
    assert O.fmt_types("") == []
    with pytest.raises(ValueError):
        O.fmt_types("%q")
    with pytest.raises(ValueError):
        O.fmt_types("%lld")


def test_empty_format_has_no_body():
    arena, offs, lens = O.wave_encode("", [], 3)
    assert arena.size == 0 and lens.tolist() == [0, 0, 0]


def test_zero_length_arrays_and_empty_string():
    fmt = "%ad %s %auc"
    cols = make_wave(fmt, 6, 5, counts={0: np.zeros(6, np.uint64), 1: np.zeros(6, np.uint64),
                                        2: np.array([0, 3, 0, 0, 1, 0], np.uint64)})
    arena, offs, lens = O.wave_encode(fmt, cols, 6)
    assert lens.tolist() == [8 + 5 + 8 + c for c in (0, 3, 0, 0, 1, 0)]
    dec, ok = O.wave_decode(fmt, arena, offs, lens, O.LITTLE)
    assert ok.all() and columns_equal(dec, cols)


def test_truncated_packets_fail_per_packet():
    fmt = "%d %ad %lf"
    cols = make_wave(fmt, 4, 11, counts={1: np.array([3, 3, 3, 3], np.uint64)})
    arena, offs, lens = O.wave_encode(fmt, cols, 4)
    short = lens.copy()
    short[0] = 2            # inside the first scalar
    short[1] = 4 + 8 + 5    # inside the array
    short[2] = 4 + 8 + 12 + 7   # inside the trailing double
    dec, ok = O.wave_decode(fmt, arena, offs, short, O.LITTLE)
    assert ok.tolist() == [0, 0, 0, 1]
    assert dec[0]["values"][0] == 0 and dec[1]["counts"][0] == 0
    assert dec[0]["values"][1] == cols[0]["values"][1] and dec[1]["counts"][1] == 0
    assert dec[1]["counts"][2] == 3 and dec[2]["values"][2] == 0
    assert dec[2]["values"][3] == cols[2]["values"][3]
    # a count above the %a bound (INT32_MAX) fails even if the prefix is readable
    bad = arena.copy()
    bad[int(offs[3]) + 4:int(offs[3]) + 12] = np.frombuffer(np.uint64(1 << 40).tobytes(), np.uint8)
    dec, ok = O.wave_decode(fmt, bad, offs, lens, O.LITTLE)
    assert ok.tolist() == [1, 1, 1, 0]


def test_filter_semantics_wrap_nan_avg():
    # wrap-around in the element type
    r, _ = O.filter_flat(O.SUM, O.CHAR_T, np.full(40, 7, np.int8))
    assert int(r) == 24                       # 280 mod 256
    r, _ = O.filter_flat(O.SUM, O.UINT16_T, np.full(5000, 17, np.uint16))
    assert int(r) == (5000 * 17) & 0xFFFF
    r, _ = O.filter_flat(O.SUM, O.INT32_T, np.array([2**31 - 1, 1], np.int32))
    assert int(r) == -2**31
    # NaN survives Max/Min only when it is last (src/FilterDefinitions.C:2090-2136)
    nan = np.float32("nan")
    assert O.filter_flat(O.MAX, O.FLOAT_T, np.array([1, nan, 2], np.float32))[0] == 2
    assert np.isnan(O.filter_flat(O.MAX, O.FLOAT_T, np.array([1, 2, nan], np.float32))[0])
    assert O.filter_flat(O.MIN, O.DOUBLE_T, np.array([np.nan, 5, 7]))[0] == 5
    # Avg: product in T, truncating division, count summed as int
    assert O.filter_flat(O.AVG, O.INT32_T, np.array([-7, 2], np.int32), np.array([1, 1], np.int32)) == (-2, 2)
    r, c = O.filter_flat(O.AVG, O.UCHAR_T, np.array([200, 100], np.uint8), np.array([2, 1], np.int32))
    assert (int(r), c) == (81, 3)             # (uchar)400 = 144, + 100 = 244, / 3 = 81
    assert O.filter_flat(O.AVG, O.INT32_T, np.array([1], np.int32), np.array([0], np.int32)) is None
    # the tree shape with a single level of fan-in 0 is the flat fold
    v = np.array([123.45, 1e8, -1e8, 0.1, 3], np.float32)
    assert O.filter_tree(O.SUM, O.FLOAT_T, v, None, [0])[0].tobytes() == O.filter_flat(O.SUM, O.FLOAT_T, v)[0].tobytes()
    # and a different shape changes a float sum, which is why the shape is declared
    assert O.filter_tree(O.SUM, O.FLOAT_T, v, None, [2, 0])[0] != O.filter_flat(O.SUM, O.FLOAT_T, v)[0]


def test_message_framing_round_trip():
    """src/Message.C: u8 byte order | u32 count | 2n x u64 sizes | spare byte | hdr0 buf0 hdr1 ..."""
    fmt = "%d %s %auc"
    n = 3
    cols = make_wave(fmt, n, 21, max_count=9)
    arena, offs, lens = O.wave_encode(fmt, cols, n)
    hdrs = [O.header_encode(1 << 30, 100, i, fmt) for i in range(n)]
    h = len(hdrs[0])
    hoffs = np.arange(n, dtype=np.uint64) * np.uint64(h)
    msg = O.message_build(np.concatenate(hdrs), hoffs, np.full(n, h, np.uint64), arena, offs[:n], lens)
    assert msg.size == 5 + 16 * n + 1 + n * h + int(lens.sum())
    assert msg[0] == 0 and msg[1:5].view(np.uint32)[0] == n
    sizes = msg[5:5 + 16 * n].view(np.uint64)
    assert sizes.tolist() == [v for i in range(n) for v in (h, int(lens[i]))]
    parsed = O.message_parse(msg)
    assert parsed is not None and parsed[0] == n
    _, ho, hl, bo, bl = parsed
    for i in range(n):
        assert msg[int(ho[i]):int(ho[i] + hl[i])].tobytes() == hdrs[i].tobytes()
        assert msg[int(bo[i]):int(bo[i] + bl[i])].tobytes() == arena[int(offs[i]):int(offs[i] + lens[i])].tobytes()
    # the count honours the sender's flag, the size vector is read in local order (:78 vs :117)
    be = msg.copy()
    be[0] = 1
    be[1:5] = msg[1:5][::-1]
    assert O.message_parse(be)[0] == n
    assert O.message_parse(msg[:-1]) is None          # truncated payload
    assert O.message_parse(O.message_build(np.zeros(0, np.uint8), [], [], np.zeros(0, np.uint8), [], []))[0] == 0


def test_perfdata_filter_reference_rules():
    """src/FilterDefinitions.C:904-1239: MIN starts from bytes 0xFF (-1 for int64), MAX / SUM /
    AVG from 0, packets without data are not ranks, a negative rank is an aggregate of -rank
    ranks, AVG divides only the first len(last packet) elements.  Unpinned by reference tests."""
    ranks = (np.ones(3, np.uint64), np.array([0, 1, 2], np.int32))
    data = (np.array([2, 0, 1], np.uint64), np.array([5, -7, 9], np.int64))
    want = {O.ORC_SUM if hasattr(O, "ORC_SUM") else 0: [14, -7], 1: [-1, -7], 2: [9, 0], 3: [7, -7]}
    for op, w in want.items():
        rank, out = O.filter_perfdata(op, O.INT64_T, ranks, data)
        assert rank == -2 and out.tolist() == w
    # aggregates: ranks -3 and -2 weigh 3 and 2 in AVG; total 5
    ranks = (np.ones(2, np.uint64), np.array([-3, -2], np.int32))
    data = (np.array([1, 1], np.uint64), np.array([10.0, 20.0]))
    rank, out = O.filter_perfdata(3, O.DOUBLE_T, ranks, data)
    assert rank == -5 and out.tolist() == [(10.0 * 3 + 20.0 * 2) / 5]
    # double MIN: the NaN start is discarded by the first comparison
    rank, out = O.filter_perfdata(1, O.DOUBLE_T, ranks, data)
    assert out.tolist() == [10.0]
    # the same filter at every level of a tree: AVG divides per level
    ranks = (np.ones(4, np.uint64), np.arange(4, dtype=np.int32))
    data = (np.full(4, 1, np.uint64), np.array([1, 2, 4, 9], np.uint64))
    flat = O.filter_perfdata(3, O.UINT64_T, ranks, data)
    tree = O.filter_perfdata_tree(3, O.UINT64_T, ranks, data, [2, 0])
    assert flat == (-4, [4]) or (flat[0] == -4 and flat[1].tolist() == [4])       # 16 / 4
    assert tree[0] == -4 and tree[1].tolist() == [((1 + 2) // 2 * 2 + (4 + 9) // 2 * 2) // 4]   # (2 + 12) / 4 = 3


def test_oracle_matches_its_frozen_vectors(frozen, tmp_path):
    """The committed vectors are what the oracle produces today (a change of the oracle's
    behaviour must show up as a diff of tests/golden/oracle_vectors.json)."""
    import json
    import subprocess
    import sys
    import os
    import shutil
    here = os.path.dirname(os.path.abspath(__file__))
    gen = os.path.join(here, "golden", "make_oracle_vectors.py")
    path = os.path.join(here, "golden", "oracle_vectors.json")
    keep = tmp_path / "kept.json"
    shutil.copy(path, keep)
    try:
        subprocess.check_call([sys.executable, gen])
        with open(path) as f:
            again = json.load(f)
    finally:
        shutil.copy(keep, path)
    assert again == frozen


def test_int_eq_class_oracle():
    """src/FilterDefinitions.C:812-902: classes merged by value, members in arrival order."""
    v, c, m = O.filter_int_eq_class(([2, 2], [5, 7, 5, 3]), ([2, 2], [2, 1, 1, 2]), ([3, 3], [1, 2, 3, 9, 4, 4]))
    assert v.tolist() == [3, 5, 7] and c.tolist() == [2, 3, 1] and m.tolist() == [4, 4, 1, 2, 9, 3]


def test_wave_decode_with_one_order_per_packet():
    """Every packet decoded with the order its message announced (src/Packet.C:96,118): a wave
    whose packets alternate between the two orders decodes to the columns it was built from."""
    from wavegen import make_wave, to_sender_order, columns_equal
    fmt = "%hd %s %auld %as %f"
    n = 37
    cols = make_wave(fmt, n, 5, max_count=9)
    arena, offs, lens = O.wave_encode(fmt, cols, n)
    be = to_sender_order(fmt, arena, offs, lens)
    orders = (np.arange(n) % 2).astype(np.uint8)
    src = arena.copy()
    for i in np.nonzero(orders)[0]:
        src[int(offs[i]):int(offs[i] + lens[i])] = be[int(offs[i]):int(offs[i] + lens[i])]
    got, ok = O.wave_decode(fmt, src, offs, lens, orders)
    assert bool(ok.all()) and columns_equal(got, cols)
    # the wrong order for the foreign half: length prefixes read reversed, those packets fail
    _, ok_le = O.wave_decode(fmt, src, offs, lens, O.LITTLE)
    assert np.array_equal(ok_le != 0, orders == 0)


def test_null_string_cannot_be_encoded():
    """pdr_string refuses a NULL pointer on encode (src/pdr.c:279-282); in column form that is a
    '%s' field without a single byte."""
    cols = [{"type": O.fmt_types("%d %s")[0], "values": np.array([1, 2], np.int32)},
            {"type": O.fmt_types("%d %s")[1], "counts": np.array([3, 0], np.uint64),
             "offs": np.array([0, 3, 3], np.uint64), "values": np.frombuffer(b"hi\0", dtype=np.uint8).copy()}]
    with pytest.raises(RuntimeError):
        O.wave_sizeof("%d %s", cols, 2)
    cols[1]["counts"] = np.array([3, 1], np.uint64)           # the empty string "" is fine: one NUL byte
    cols[1]["offs"] = np.array([0, 3, 4], np.uint64)
    cols[1]["values"] = np.frombuffer(b"hi\0\0", dtype=np.uint8).copy()
    arena, offs, lens = O.wave_encode("%d %s", cols, 2)
    assert lens.tolist() == [4 + 4 + 3, 4 + 4 + 1]
