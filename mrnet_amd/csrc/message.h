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
#include "header.h"
#include "scan.h"

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

// The sizes of a received message are untrusted: each enters the sums as at most `clamp` =
// 2^63 / (n + 1), so neither a pair nor the scan over the pairs can wrap; a clamped size makes the
// payload total larger than any message that exists, which the length check of the caller (and the
// bounds checks of the header and data decoders, who get the sizes as received) then refuse.

// ---------------------------------------------------------------------------
// The receive side in two launches (Message::recv, src/Message.C:48-199, which ends by building a
// Packet from every header buffer: Packet::decode_pdr_header, src/Packet.C:87-104)
// ---------------------------------------------------------------------------
#define MSG_BLOCK_PKTS 1024             // packets per workgroup, four per lane
static_assert(MSG_BLOCK_PKTS == 4 * MRPDR_TPB, "a lane owns four consecutive packets");

// The (clamped) header + data bytes of the lane's four packets.
__device__ __forceinline__ void msg_load_sizes(const uint8_t *sizes, uint64_t n, uint64_t clamp, uint64_t first,
                                               uint64_t (&h)[4], uint64_t (&b)[4], uint64_t (&pair)[4])
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = first + k < n ? first + k : n - 1;         // unconditional loads, issued together
        h[k] = ld_u<uint64_t>(sizes + 16 * i);
        b[k] = ld_u<uint64_t>(sizes + 16 * i + 8);
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        pair[k] = first + k < n ? (h[k] < clamp ? h[k] : clamp) + (b[k] < clamp ? b[k] : clamp) : 0;
}

__global__ __launch_bounds__(MRPDR_TPB) void k_msg_sums(const uint8_t *sizes, uint64_t n, uint64_t clamp, uint64_t *block_sums)
{
    uint64_t h[4], b[4], pair[4];
    msg_load_sizes(sizes, n, clamp, (uint64_t)blockIdx.x * MSG_BLOCK_PKTS + threadIdx.x * 4u, h, b, pair);
    uint64_t total;
    block_exclusive_scan(pair[0] + pair[1] + pair[2] + pair[3], &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Offsets and lengths of every header and data buffer (relative to the payload), then the headers
// themselves: the lane that placed a header is not the lane that reads it (reads go lane = packet,
// so that a wavefront's headers are neighbours), the places pass through LDS.
__global__ __launch_bounds__(MRPDR_TPB) void k_msg_place_headers(const uint8_t *sizes, uint64_t n, uint64_t clamp,
                                                                 const uint64_t *block_sums,
                                                                 uint64_t *hdr_offsets, uint64_t *hdr_lens,
                                                                 uint64_t *buf_offsets, uint64_t *buf_lens,
                                                                 HdrDecodeArgs a, HdrImage img)
{
    __shared__ uint64_t s_at[MSG_BLOCK_PKTS], s_len[MSG_BLOCK_PKTS];
    const uint64_t base_pkt = (uint64_t)blockIdx.x * MSG_BLOCK_PKTS;
    const uint64_t first = base_pkt + threadIdx.x * 4u;
    uint64_t h[4], b[4], pair[4];
    msg_load_sizes(sizes, n, clamp, first, h, b, pair);
    uint64_t s = 0;
    for (uint64_t j = threadIdx.x; j < blockIdx.x; j += MRPDR_TPB) s += block_sums[j];
    uint64_t before;
    block_exclusive_scan(s, &before);
    uint64_t total;
    uint64_t run = before + block_exclusive_scan(pair[0] + pair[1] + pair[2] + pair[3], &total);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = first + k;
        if (i < n) {
            hdr_offsets[i] = run;
            hdr_lens[i] = h[k];
            buf_offsets[i] = run + h[k];
            buf_lens[i] = b[k];
            s_at[threadIdx.x * 4u + k] = run;
            s_len[threadIdx.x * 4u + k] = h[k];
        }
        run += pair[k];
        if (i + 1 == n) { hdr_offsets[n] = run; buf_offsets[n] = run; }
    }
    if (!a.hdrs) return;                  // the sizes alone (uniform)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t j = threadIdx.x + (uint32_t)k * MRPDR_TPB;
        if (base_pkt + j < n) hdr_decode_one(a, img, base_pkt + j, s_at[j], s_len[j]);
    }
}

// The send side's framing in one launch: the preamble entry of packet i and its header, which sits
// hdr_len bytes in front of its data (Message::send, src/Message.C:201-335, over the header buffer
// Packet::encode_pdr_header made, src/Packet.C:557-593).
__global__ __launch_bounds__(MRPDR_TPB) void k_msg_frame(uint8_t *msg, uint64_t n, uint64_t hdr_len, const uint64_t *lens,
                                                         HdrEncodeArgs a, HdrImage img)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i == 0) {
        msg[0] = (uint8_t)MRPDR_LITTLEENDIAN;
        st_u<uint32_t>(msg + 1, (uint32_t)n);
        msg[MSG_COUNT_BYTES + 16 * n] = 0;
    }
    if (i >= n) return;
    st_u<uint64_t>(msg + MSG_COUNT_BYTES + 16 * i, hdr_len);
    st_u<uint64_t>(msg + MSG_COUNT_BYTES + 16 * i + 8, lens[i]);
    hdr_encode_one(a, img, i, a.hdrs + a.hdr_offsets[i]);
}
