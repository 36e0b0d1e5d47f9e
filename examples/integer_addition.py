"""Examples/IntegerAddition of the reference, reduced to what one tree node does per wave.

There (IntegerAddition_FE.C / _BE.C / IntegerAdditionFilter.C) the front-end broadcasts
(send_val, num_iters); every back-end answers iteration i with send_val * i on a stream whose
upstream filter adds the integers of a wave; the front-end expects
num_backends * send_val * i per iteration.  Here the waves of all iterations are built as the
back-ends' packets would arrive at a node (headers + '%d' bodies), and the node's work -- decode the
headers, fold the wave, emit one packet -- runs on the GPU.

    python examples/integer_addition.py [num_backends] [num_iters]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mrnet_amd import packet as P  # noqa: E402
from mrnet_amd import wave as W  # noqa: E402

PROT_SUM = W.FIRST_APPLICATION_TAG + 1


def main(num_backends=65536, num_iters=5, send_val=32, verbose=True):
    codec = W.Codec(max_packets=max(num_backends, 1024))
    add = P.Filter(lambda wave, state, params: ([P.aggr_Count(wave)], state))       # IntegerAdd
    ranks = codec.to_device(np.arange(num_backends, dtype=np.uint32))
    ok = True
    for i in range(num_iters):
        # what the back-ends send: Packet(stream, PROT_SUM, "%d", send_val * i), one each
        sent = P.PacketWave(codec, W.USER_STRM_BASE_ID, PROT_SUM, "%d",
                            [np.full(num_backends, send_val * i, np.int32)], src_rank=0)
        sent.src_ranks = ranks
        # what the node receives: header and data buffers; it knows nothing else
        wave = P.PacketWave.from_buffers(codec, "%d", num_backends, sent.get_Header(), sent.get_HeaderLen(),
                                         sent.get_Buffer())
        out = add.push_Packets(wave)[0]
        recv_val = int(W.Codec.to_numpy(out.unpack("%d")[0].values).view(np.int32)[0])
        want = int(np.array(num_backends * send_val * i, dtype=np.int64).astype(np.int32))   # int wraps, as in C
        good = recv_val == want and out.get_Tag() == PROT_SUM
        ok = ok and good
        if verbose:
            print("Iteration %d: %s! Expected %d. Got %d." % (i, "Success" if good else "Failure", want, recv_val))
    return ok


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:3]]
    sys.exit(0 if main(*a) else 1)
