// Scalars and length prefixes: packed rows <-> per-field columns.
//
// The reference handles these one primitive at a time: pdr_packet_data switches on
// the element type (src/Packet.C:629-828) and calls pdr_<type>, which calls
// pdrmem_put<type> / pdrmem_get<type>[ _swap ] (src/pdr.c:34-127,
// src/pdr_mem.c:22-315) at a running cursor with a bounds check, no padding.
//
// Two forms here:
//   k_rows_*   scalar-only formats in a packed arena: a workgroup stages 256 (or 1,024) packet
//              rows in LDS with 128-bit coalesced global accesses and transposes
//              them to / from the columns
//   k_walk_*   any format / any arena layout: one lane walks one packet
#pragma once

#include "common.h"

#define ROWS_PER_BLOCK MRPDR_TPB
#define ROWS_MAX_STRIDE 256
// Large waves of narrow rows: a lane takes ROWS_WIDE rows (t, t + 256, ...), so a workgroup moves
// 4 KiB and more instead of 1 KiB and its fixed costs are shared by four times the bytes.
#define ROWS_WIDE 4
#define ROWS_WIDE_MAX_STRIDE 16          // 16 KiB of LDS per workgroup at most
#define ROWS_WIDE_MIN_PACKETS (1u << 19) // below, more workgroups matter more than bigger ones

// ---------------------------------------------------------------------------
// LDS-staged transposition, scalar-only formats, packed arena
// ---------------------------------------------------------------------------
template <bool SWAP, int R>
__global__ __launch_bounds__(MRPDR_TPB) void k_rows_decode(FmtDev fmt, WireDev wire, ColsDev cols,
                                                           uint64_t n, int32_t *ok)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    constexpr uint32_t ROWS = ROWS_PER_BLOCK * R;
    const uint64_t p0 = (uint64_t)blockIdx.x * ROWS;
    const uint32_t np = (uint32_t)((n - p0) < ROWS ? (n - p0) : ROWS);
    const uint32_t stride = (uint32_t)wire.stride;
    const uint32_t tile_bytes = np * stride;
    // p0 * stride is a multiple of 256 bytes, so with a 16-byte aligned arena every
    // 16-byte piece of the tile is an aligned 128-bit load.  Four loads of a lane are issued
    // before the first LDS store (unconditional: a lane past the end re-reads the last piece).
    const uint8_t *g = wire.base + p0 * (uint64_t)stride;
    const uint32_t whole = tile_bytes & ~15u;
    for (uint32_t base = 0; base < whole; base += 4u * MRPDR_TPB * 16u) {
        uint4 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t off = base + ((uint32_t)u * MRPDR_TPB + threadIdx.x) * 16u;
            x[u] = *reinterpret_cast<const uint4 *>(g + (off < whole ? off : whole - 16u));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t off = base + ((uint32_t)u * MRPDR_TPB + threadIdx.x) * 16u;
            if (off < whole) *reinterpret_cast<uint4 *>(tile + off) = x[u];
        }
    }
    if (whole + threadIdx.x < tile_bytes) tile[whole + threadIdx.x] = g[whole + threadIdx.x];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t lr = threadIdx.x + (uint32_t)r * MRPDR_TPB;
        if (lr >= np) break;
        const uint64_t i = p0 + lr;
        const uint8_t *row = tile + lr * stride;
        const uint64_t len = wire.lens ? wire.lens[i] : (uint64_t)fmt.fixed_bytes;
        const bool sw = pkt_swap(wire, i, SWAP);        // the wave's order, or this packet's own
        bool good = true;
        for (int f = 0; f < fmt.nf; f++) {
            const uint32_t w = fmt.f[f].width, at = fmt.f[f].fixed_before;
            uint64_t v = 0;
            // same test as every pdrmem_get*: width > remaining space -> FALSE
            if (good && (uint64_t)at + w <= len && at + w <= stride) {
                if (((at | stride) & (w - 1)) == 0) {                 // naturally aligned in the LDS tile
                    switch (w) {
                    case 1: v = row[at]; break;
                    case 2: v = *reinterpret_cast<const uint16_t *>(row + at); if (sw) v = __builtin_bswap16((uint16_t)v); break;
                    case 4: v = *reinterpret_cast<const uint32_t *>(row + at); if (sw) v = __builtin_bswap32((uint32_t)v); break;
                    default: v = *reinterpret_cast<const uint64_t *>(row + at); if (sw) v = __builtin_bswap64(v); break;
                    }
                } else {
                    for (uint32_t b = 0; b < w; b++) {
                        const uint32_t src = sw ? (w - 1 - b) : b;    // src/byte_order.c:12-16
                        v |= (uint64_t)row[at + src] << (8 * b);
                    }
                }
            } else {
                good = false;
            }
            st_w(reinterpret_cast<uint8_t *>(cols.values[f]) + i * w, (int)w, v);
        }
        if (ok) ok[i] = good ? 1 : 0;
    }
}

template <int R>
__global__ __launch_bounds__(MRPDR_TPB) void k_rows_encode(FmtDev fmt, WireDev wire, ColsDev cols, uint64_t n)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    constexpr uint32_t ROWS = ROWS_PER_BLOCK * R;
    const uint64_t p0 = (uint64_t)blockIdx.x * ROWS;
    const uint32_t np = (uint32_t)((n - p0) < ROWS ? (n - p0) : ROWS);
    const uint32_t stride = (uint32_t)wire.stride;
    const uint32_t tile_bytes = np * stride;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t lr = threadIdx.x + (uint32_t)r * MRPDR_TPB;
        if (lr >= np) break;
        const uint64_t i = p0 + lr;
        uint8_t *row = tile + lr * stride;
        for (int f = 0; f < fmt.nf; f++) {
            const uint32_t w = fmt.f[f].width, at = fmt.f[f].fixed_before;
            const uint64_t v = ld_wire_w(reinterpret_cast<const uint8_t *>(cols.values[f]) + i * w, (int)w, false);
            if (((at | stride) & (w - 1)) == 0) {
                switch (w) {
                case 1: row[at] = (uint8_t)v; break;
                case 2: *reinterpret_cast<uint16_t *>(row + at) = (uint16_t)v; break;
                case 4: *reinterpret_cast<uint32_t *>(row + at) = (uint32_t)v; break;
                default: *reinterpret_cast<uint64_t *>(row + at) = v; break;
                }
            } else {
                for (uint32_t b = 0; b < w; b++) row[at + b] = (uint8_t)(v >> (8 * b));
            }
        }
        for (uint32_t b = fmt.fixed_bytes; b < stride; b++) row[b] = 0;     // arena padding
    }
    __syncthreads();
    uint8_t *g = wire.base + p0 * (uint64_t)stride;
    for (uint32_t off = threadIdx.x * 16u; off < tile_bytes; off += MRPDR_TPB * 16u) {
        if (off + 16u <= tile_bytes) {
            *reinterpret_cast<uint4 *>(g + off) = *reinterpret_cast<const uint4 *>(tile + off);
        } else {
            for (uint32_t b = off; b < tile_bytes; b++) g[b] = tile[b];
        }
    }
}

// ---------------------------------------------------------------------------
// generic walkers: one lane per packet
// ---------------------------------------------------------------------------
// Decode walk.  Scalars go to their columns, the element count of every array /
// string field goes to counts[], ok[i] tells whether the whole packet decoded.
// Failure rules are the reference's: a primitive wider than the remaining space
// fails (src/pdr_mem.c:42,80,130,...), a count above the token's bound fails
// (src/pdr.c:224,429), array elements are checked one by one so the array fails when
// count*width exceeds the space (src/pdr.c:459-462).  After the first failure the
// remaining fields of that packet read as zero.
// String arrays: the lane walks the strings of its packet (their lengths are on the wire one
// behind the other), counts[] receives the number of strings and sa_bytes[f] accumulates the
// bytes of all strings of field f over the wave (what the caller sizes values[] with).
//
// p points at the packet's first byte (global memory, or a copy of the wire staged in LDS).
__device__ __forceinline__ bool walk_decode_packet(const bool swap, const FmtDev &fmt, const ColsDev &cols, uint64_t i,
                                                   const uint8_t *p, uint64_t space, bool good,
                                                   uint64_t *sa_bytes, uint64_t *cnt_out = nullptr,
                                                   uint32_t cnt_stride = 0)
{
    // cnt_out (optional): the count of the k-th variable field also goes to cnt_out[k * cnt_stride]
    for (int f = 0; f < fmt.nf; f++) {
        const FieldDev fd = fmt.f[f];
        const uint32_t w = fd.width;
        if (fd.kind == MRPDR_SCALAR) {
            uint64_t v = 0;
            if (good && w <= space) { v = ld_wire_w(p, (int)w, swap); p += w; space -= w; }
            else good = false;
            st_w(reinterpret_cast<uint8_t *>(cols.values[f]) + i * w, (int)w, v);
        } else if (fd.kind == MRPDR_ARRAY) {
            uint64_t cnt = 0;
            if (good && 8 <= space) {
                cnt = ld_wire<8>(p, swap); p += 8; space -= 8;
                const uint64_t bound = fd.large ? UINT64_MAX : (uint64_t)INT32_MAX;
                if (cnt > bound || cnt > space / w) { good = false; cnt = 0; }
                p += cnt * w; space -= cnt * w;
            } else good = false;
            cols.counts[f][i] = cnt;
            if (cnt_out) { *cnt_out = cnt; cnt_out += cnt_stride; }
        } else if (fd.kind == MRPDR_STRING_ARRAY) {
            // u64 count, then count x [u32 strlen+1 | bytes] (src/Packet.C:794-813)
            uint64_t cnt = 0, bytes = 0;
            if (good && 8 <= space) {
                cnt = ld_wire<8>(p, swap); p += 8; space -= 8;
                const uint64_t bound = fd.large ? UINT64_MAX : (uint64_t)INT32_MAX;
                if (cnt > bound || cnt > space) good = false;
                for (uint64_t k = 0; good && k < cnt; k++) {
                    if (4 > space) { good = false; break; }
                    const uint64_t l = ld_wire<4>(p, swap); p += 4; space -= 4;
                    if (l == 0 || l > space) { good = false; break; }
                    p += l; space -= l; bytes += l;
                }
                if (!good) cnt = 0;
                else if (bytes) atomicAdd((unsigned long long *)&sa_bytes[f], (unsigned long long)bytes);
            } else good = false;
            cols.counts[f][i] = cnt;
            if (cnt_out) { *cnt_out = cnt; cnt_out += cnt_stride; }
        } else {    // MRPDR_STRING: u32 strlen+1, then the bytes, NUL included
            uint64_t cnt = 0;
            if (good && 4 <= space) {
                cnt = ld_wire<4>(p, swap); p += 4; space -= 4;
                if (cnt == 0 || cnt > space) { good = false; cnt = 0; }
                p += cnt; space -= cnt;
            } else good = false;
            cols.counts[f][i] = cnt;
            if (cnt_out) { *cnt_out = cnt; cnt_out += cnt_stride; }
        }
    }
    return good;
}

template <bool SWAP>
__global__ __launch_bounds__(MRPDR_TPB) void k_walk_decode(FmtDev fmt, WireDev wire, ColsDev cols,
                                                           uint64_t n, int32_t *ok, uint64_t *sa_bytes)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i >= n) return;
    const uint64_t at = pkt_off(wire, i);
    const uint64_t space = pkt_len(wire, fmt, i);
    bool good = at <= wire.bytes && space <= wire.bytes - at;
    good = walk_decode_packet(pkt_swap(wire, i, SWAP), fmt, cols, i, wire.base + at, space, good, sa_bytes);
    if (ok) ok[i] = good ? 1 : 0;
}

// Encode walk: scalars and length prefixes at their running cursor.  Array and
// string payloads are moved by the segment copy kernel.
__device__ __forceinline__ void walk_encode_packet(const FmtDev &fmt, const WireDev &wire, const ColsDev &cols, uint64_t i)
{
    uint8_t *p = wire.base + pkt_off(wire, i);
    for (int f = 0; f < fmt.nf; f++) {
        const FieldDev fd = fmt.f[f];
        const uint32_t w = fd.width;
        if (fd.kind == MRPDR_SCALAR) {
            st_w(p, (int)w, ld_wire_w(reinterpret_cast<const uint8_t *>(cols.values[f]) + i * w, (int)w, false));
            p += w;
        } else if (fd.kind == MRPDR_ARRAY || fd.kind == MRPDR_STRING_ARRAY) {
            const uint64_t cnt = cols.counts[f][i];
            st_u<uint64_t>(p, cnt);                      // pdr_uint64(count), src/pdr.c:425
            p += 8 + var_bytes(fmt, cols, f, i);
        } else {
            const uint64_t cnt = cols.counts[f][i];
            st_u<uint32_t>(p, (uint32_t)cnt);            // pdr_uint32(strlen+1), src/pdr.c:295-297
            p += 4 + cnt;
        }
    }
}

__global__ __launch_bounds__(MRPDR_TPB) void k_walk_encode(FmtDev fmt, WireDev wire, ColsDev cols, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i < n) walk_encode_packet(fmt, wire, cols, i);
}
