// Packet headers of a wave (Packet::pdr_packet_header, src/Packet.C:557-593):
//   u32 stream_id | i32 tag | u32 src_rank | u32 strlen(fmt)+1 | fmt bytes, NUL |
//   u64 ndest | ndest x u32 dest_rank | u8 byteorder
// Encoded in the sender's native order; the byte order a header was written in is its
// own last byte (src/Packet.C:96).
#pragma once

#include "common.h"

#define HDR_MAX_FMT 192     // strlen(fmt)+1 that fits the kernel-argument image

struct HdrImage {
    uint32_t fmt_bytes;                 // strlen(fmt)+1
    uint8_t  fmt[HDR_MAX_FMT];
};

struct HdrEncodeArgs {
    uint32_t stream_id; const uint32_t *stream_ids;
    int32_t  tag;       const int32_t  *tags;
    uint32_t src_rank;  const uint32_t *src_ranks;
    const uint32_t *dests; uint64_t ndest;          // one destination list for the whole wave
    uint8_t *hdrs; const uint64_t *hdr_offsets; uint64_t hdr_stride; uint64_t n;
};

struct HdrDecodeArgs {
    const uint8_t *hdrs; uint64_t hdrs_bytes; const uint64_t *hdr_offsets; const uint64_t *hdr_lens; uint64_t hdr_stride;
    uint32_t *stream_ids; int32_t *tags; uint32_t *src_ranks; uint8_t *byteorders; uint64_t *ndests;
    uint32_t *dests; uint64_t dest_stride;
    int32_t *ok; uint64_t n;
};

// One header: packet i's, `space` bytes at byte `at` of the buffer.
__device__ __forceinline__ void hdr_decode_one(const HdrDecodeArgs &a, const HdrImage &img, uint64_t i, uint64_t at, uint64_t space)
{
    // offsets and lengths may come off the wire (a Message's size vector): a header that does
    // not lie inside the buffer fails without being read
    const bool inside = at <= a.hdrs_bytes && space <= a.hdrs_bytes - at;
    const uint8_t *p = a.hdrs + (inside ? at : 0);
    if (!inside) space = 0;
    bool good = space >= 1;
    const bool swap = good && p[space - 1] != (uint8_t)MRPDR_LITTLEENDIAN;
    uint32_t sid = 0, src = 0, flen = 0;
    int32_t tag = 0;
    uint64_t nd = 0;
    uint8_t bo = 0;
    if (good && space >= 16) {
        sid  = (uint32_t)ld_wire<4>(p + 0, swap);
        tag  = (int32_t)(uint32_t)ld_wire<4>(p + 4, swap);
        src  = (uint32_t)ld_wire<4>(p + 8, swap);
        flen = (uint32_t)ld_wire<4>(p + 12, swap);
        p += 16; space -= 16;
    } else good = false;
    // format string: must be the wave's (the built-in filters assert this per packet,
    // src/FilterDefinitions.C:104-105)
    if (good && flen != 0 && flen <= space) {
        bool same = flen == img.fmt_bytes;
        if (same)
            for (uint32_t b = 0; b < flen; b++) same = same && (p[b] == img.fmt[b]);
        good = same;
        p += flen; space -= flen;
    } else good = false;
    if (good && space >= 8) {
        nd = ld_wire<8>(p, swap); p += 8; space -= 8;
        if (nd > space / 4) good = false;
        else {
            if (a.dests)
                for (uint64_t d = 0; d < nd && d < a.dest_stride; d++)
                    a.dests[i * a.dest_stride + d] = (uint32_t)ld_wire<4>(p + 4 * d, swap);
            p += nd * 4; space -= nd * 4;
        }
    } else good = false;
    if (good && space >= 1) bo = p[0];
    else good = false;
    if (a.stream_ids) a.stream_ids[i] = sid;
    if (a.tags) a.tags[i] = tag;
    if (a.src_ranks) a.src_ranks[i] = src;
    if (a.byteorders) a.byteorders[i] = bo;
    if (a.ndests) a.ndests[i] = nd;
    if (a.ok) a.ok[i] = good ? 1 : 0;
}

__global__ __launch_bounds__(MRPDR_TPB) void k_hdr_decode(HdrDecodeArgs a, HdrImage img)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i >= a.n) return;
    hdr_decode_one(a, img, i, a.hdr_offsets ? a.hdr_offsets[i] : i * a.hdr_stride, a.hdr_lens ? a.hdr_lens[i] : a.hdr_stride);
}

// One header written: packet i's at p.
__device__ __forceinline__ void hdr_encode_one(const HdrEncodeArgs &a, const HdrImage &img, uint64_t i, uint8_t *p)
{
    st_u<uint32_t>(p + 0, a.stream_ids ? a.stream_ids[i] : a.stream_id);
    st_u<int32_t>(p + 4, a.tags ? a.tags[i] : a.tag);
    st_u<uint32_t>(p + 8, a.src_ranks ? a.src_ranks[i] : a.src_rank);
    st_u<uint32_t>(p + 12, img.fmt_bytes);
    for (uint32_t b = 0; b < img.fmt_bytes; b++) p[16 + b] = img.fmt[b];
    uint8_t *q = p + 16 + img.fmt_bytes;
    st_u<uint64_t>(q, a.ndest);                 // pdr_uint64(dest count), src/Packet.C:579-585
    q += 8;
    for (uint64_t d = 0; d < a.ndest; d++, q += 4) st_u<uint32_t>(q, a.dests[d]);
    q[0] = (uint8_t)MRPDR_LITTLEENDIAN;         // pdrmem_getbo() of this device
}

__global__ __launch_bounds__(MRPDR_TPB) void k_hdr_encode(HdrEncodeArgs a, HdrImage img)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i >= a.n) return;
    hdr_encode_one(a, img, i, a.hdrs + (a.hdr_offsets ? a.hdr_offsets[i] : i * a.hdr_stride));
}
