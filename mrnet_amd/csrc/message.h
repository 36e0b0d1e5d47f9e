// Message framing: the unit MRNet puts on a TCP link (src/Message.C:201-335 send, :48-199 recv)
//   u8 byteorder | u32 npackets | 2*npackets x u64 sizes (hdr0, buf0, hdr1, ...) | 1 spare byte
//   | hdr0 | buf0 | hdr1 | buf1 | ...
// The count is written in the sender's order and flagged by the first byte (:308-312); the size
// vector is written in the sender's order too but the reference reads it back with the LOCAL
// order (:117), so only same-endian peers interoperate -- kept as is.  The spare byte is
// allocated but never written by the reference (:254, :316-324); it is written as 0 here.
//
// The payload part of a message is exactly a wave arena with interleaved headers: packet i's
// data sits hdr_len bytes after its header, so encode and decode work in place on the message
// buffer and the socket can send / receive it with a single writev / readv.
#pragma once

#include "common.h"

#define MSG_COUNT_BYTES 5u      // byteorder flag + u32 count

__global__ __launch_bounds__(MRPDR_TPB) void k_msg_preamble(uint8_t *msg, uint64_t n, uint64_t hdr_len,
                                                            const uint64_t *lens)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i == 0) {
        msg[0] = (uint8_t)MRPDR_LITTLEENDIAN;
        st_u<uint32_t>(msg + 1, (uint32_t)n);
        msg[MSG_COUNT_BYTES + 16 * n] = 0;
    }
    if (i < n) {
        st_u<uint64_t>(msg + MSG_COUNT_BYTES + 16 * i, hdr_len);
        st_u<uint64_t>(msg + MSG_COUNT_BYTES + 16 * i + 8, lens[i]);
    }
}

// sizes[2n] as received -> hdr_lens, buf_lens and (in buf_offsets, as scratch) hdr+buf per packet.
// The sizes are untrusted: each enters the sum as at most `clamp` = 2^63 / (n + 1), so neither a pair
// nor the scan over the pairs can wrap; a clamped size makes the payload total larger than any
// message that exists, which the length check of the caller (and the bounds checks of the header
// and data decoders, who get the sizes as received) then refuse.
__global__ __launch_bounds__(MRPDR_TPB) void k_msg_split(const uint8_t *sizes, uint64_t n, uint64_t clamp,
                                                         uint64_t *hdr_lens, uint64_t *buf_lens, uint64_t *pair)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = ld_u<uint64_t>(sizes + 16 * i), b = ld_u<uint64_t>(sizes + 16 * i + 8);
    hdr_lens[i] = h;
    buf_lens[i] = b;
    pair[i] = (h < clamp ? h : clamp) + (b < clamp ? b : clamp);
}

// hdr_offsets[n+1] = scan of the pairs; the data of packet i follows its header
__global__ __launch_bounds__(MRPDR_TPB) void k_msg_place(const uint64_t *hdr_offsets, const uint64_t *hdr_lens,
                                                         uint64_t n, uint64_t *buf_offsets)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i < n) buf_offsets[i] = hdr_offsets[i] + hdr_lens[i];
    if (i == n) buf_offsets[n] = hdr_offsets[n];
}
