"""What an internal node of the tree does with one wave, end to end (the reference's
PeerNode::recv -> Message::recv -> Stream::push_Packet -> Filter::push_Packets -> Message::send,
src/Message.C:48-335, src/Filter.C:42-117): the messages of its children arrive as bytes,
are taken apart, the packets form one wave, the stream's filter folds it, and the result goes
upstream as a message.  Input messages are built by the oracle (the bytes a reference child would
send); the output message is parsed by the oracle and its packet decoded there."""
import numpy as np
import pytest
import torch

from wavegen import O, make_wave, random_values
from test_gpu_message import oracle_message

pytestmark = pytest.mark.gpu

from mrnet_amd import packet as P  # noqa: E402
from mrnet_amd import wave as W  # noqa: E402

STREAM, TAG = (1 << 30) + 10, 100


def child_messages(fmt_s, children, per_child, seed, values=None):
    """children x per_child packets, one message per child; returns (messages, list of column sets)."""
    msgs, cols = [], []
    for c in range(children):
        oc = make_wave(fmt_s, per_child, seed + c, max_count=6)
        if values is not None:
            oc[0]["values"] = values[c * per_child:(c + 1) * per_child].copy()
        ranks = np.arange(c * per_child, (c + 1) * per_child, dtype=np.uint32)
        msgs.append(oracle_message(fmt_s, oc, per_child, STREAM, TAG, ranks)[0])
        cols.append(oc)
    return msgs, cols


def receive_wave(codec, fmt_s, msgs):
    """Message::recv for every child, then one wave over all packets: the payloads side by side in
    one device arena, the per-message offsets shifted to their place."""
    fmt = W.Format(fmt_s)
    payloads, hdr_o, hdr_l, buf_o, buf_l = [], [], [], [], []
    at = 0
    for m in msgs:
        n, hdr, wire = codec.message_parse(fmt, codec.to_device(m))
        assert bool(hdr["ok"].all())
        pre = 6 + 16 * n
        parsed = O.message_parse(m)
        payloads.append(m[pre:])
        hdr_o.append(parsed[1] - np.uint64(pre) + np.uint64(at)); hdr_l.append(parsed[2])
        buf_o.append(W.Codec.to_numpy(wire.offsets).view(np.uint64)[:n] + np.uint64(at))
        buf_l.append(W.Codec.to_numpy(wire.lens).view(np.uint64))
        at += m.size - pre
    arena = np.concatenate(payloads)
    groups = P.group_by_format(arena, np.concatenate(hdr_o), np.concatenate(hdr_l))
    assert list(groups) == [fmt_s]
    n = sum(len(x) for x in buf_l)
    offs = np.concatenate(buf_o + [np.array([at], np.uint64)])
    return W.WireWave(codec.to_device(arena), n, codec.to_device(offs), codec.to_device(np.concatenate(buf_l)), 0)


def send_upstream(codec, out):
    """Message::send of the filter's one output packet; returns the message bytes on the host."""
    cols = out.unpack(out.fmt.text)
    msg = codec.message_build(out.fmt, cols, 1, stream_id=out.get_StreamId(), tag=out.get_Tag())
    torch.cuda.synchronize()
    return W.Codec.to_numpy(msg)


def oracle_receive(fmt_s, msg):
    n, ho, hl, bo, bl = O.message_parse(msg)
    assert n == 1
    hdr = O.header_decode(msg[int(ho[0]):int(ho[0] + hl[0])])
    cols, ok = O.wave_decode(fmt_s, msg, bo, bl, O.LITTLE)
    assert ok.tolist() == [1] and hdr["fmt"] == fmt_s and hdr["stream_id"] == STREAM and hdr["tag"] == TAG
    return cols


@pytest.mark.parametrize("children,per_child", [(4, 1), (16, 8), (3, 700)])
def test_node_sums_its_children(codec, children, per_child):
    n = children * per_child
    vals = random_values(np.int32, n, 21)
    msgs, _ = child_messages("%d", children, per_child, 3, vals)
    wire = receive_wave(codec, "%d", msgs)
    cols, ok = codec.decode(W.Format("%d"), wire)
    assert bool(ok.all())
    wave = P.PacketWave(codec, STREAM, TAG, "%d", cols, n=n)
    out = P.BUILTIN_FILTERS["TFILTER_SUM"]().push_Packets(wave)[0]
    got = oracle_receive("%d", send_upstream(codec, out))
    assert int(got[0]["values"][0]) == int(vals.sum(dtype=np.int32))


def test_node_concatenates_arrays(codec):
    children, per_child = 5, 40
    msgs, cols = child_messages("%auhd", children, per_child, 9)
    wire = receive_wave(codec, "%auhd", msgs)
    dcols, ok = codec.decode(W.Format("%auhd"), wire)
    assert bool(ok.all())
    wave = P.PacketWave(codec, STREAM, TAG, "%auhd", dcols, n=children * per_child)
    out = P.BUILTIN_FILTERS["TFILTER_ARRAY_CONCAT"]().push_Packets(wave)[0]
    got = oracle_receive("%auhd", send_upstream(codec, out))
    want = np.concatenate([c[0]["values"] for c in cols])
    assert int(got[0]["counts"][0]) == want.size and np.array_equal(got[0]["values"], want)
