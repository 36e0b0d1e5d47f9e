// Array and string payloads: the streaming kernel where the bytes go.
//
// Reference: pdr_array calls the element routine once per element through two
// indirections (src/pdr.c:459-462 -> pdr_<type> -> pdrmem_get<type>[ _swap ],
// src/pdr_mem.c:78-315); the _swap variants reverse the bytes of one element with
// byte_swap(out, in, 1, width) (src/byte_order.c:8-17).  pdr_bytes / pdr_string move
// raw bytes with memcpy and never swap them (src/pdr.c:212-325, src/pdr_mem.c:394-412).
//
// Here the payload of field f over the whole wave is one segmented copy between
//   dense side : the packed values column, segment s = [offs[s]*W, offs[s+1]*W)
//   wire side  : packet s, at its payload position inside the packet
// The dense side is cut into 16 KiB chunks; a workgroup owns a run of consecutive
// chunks, so it streams through the column with aligned 128-bit accesses and follows
// the segment boundaries as it goes.  Every lane moves 16 bytes per step; elements
// are byte-reversed in registers (v_perm_b32) when the sender's order differs.
#pragma once

#include "common.h"

#define COPY_UNROLL 4
#define COPY_PIECE 16u
#define COPY_CHUNK (MRPDR_TPB * COPY_PIECE * COPY_UNROLL)     // 16 KiB of dense bytes

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Reverse the bytes of every W-byte element in a 16-byte vector.
template <int W> __device__ inline u32x4 swap_elems(u32x4 v)
{
    if (W == 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __builtin_amdgcn_perm(v[k], v[k], 0x02030001u);
    } else if (W == 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __builtin_bswap32(v[k]);
    } else if (W == 8) {
        u32x4 r;
        r[0] = __builtin_bswap32(v[1]); r[1] = __builtin_bswap32(v[0]);
        r[2] = __builtin_bswap32(v[3]); r[3] = __builtin_bswap32(v[2]);
        v = r;
    }
    return v;
}

struct CopyArgs {
    WireDev  wire;
    uint8_t *dense;            // values column, 16-byte aligned
    const uint64_t *seg_offs;  // offs[n+1] in elements
    uint64_t n;
    uint64_t dense_cap;        // bytes available in dense
    uint32_t *status;          // bit 0 set when the column is too small
    int32_t  field;
};

// Position of field f's payload inside packet s: fixed part, own prefix, and the
// payloads of the variable fields in front of it.
__device__ inline uint64_t payload_pos(const FmtDev &fmt, const ColsDev &cols, int f, uint64_t s)
{
    uint64_t pos = (uint64_t)fmt.f[f].fixed_before + prefix_bytes(fmt.f[f].kind);
    for (int g = 0; g < f; g++)
        if (fmt.f[g].kind != MRPDR_SCALAR) pos += cols.counts[g][s] * (uint64_t)fmt.f[g].width;
    return pos;
}

// First segment whose dense range ends after element index e (offs[s+1] > e).
__device__ inline uint64_t seg_find(const uint64_t *offs, uint64_t lo, uint64_t hi, uint64_t e)
{
    while (lo < hi) {                       // invariant: answer in [lo, hi]
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (offs[mid + 1] > e) hi = mid; else lo = mid + 1;
    }
    return lo;
}

struct Seg {
    uint64_t s;          // segment (= packet) index
    uint64_t lo, hi;     // dense byte range
    uint8_t *wire;       // wire address of dense byte `lo`
};

template <int W>
__device__ inline void seg_load(Seg &sg, uint64_t s, const CopyArgs &a, const FmtDev &fmt, const ColsDev &cols)
{
    sg.s = s;
    sg.lo = a.seg_offs[s] * W;
    sg.hi = a.seg_offs[s + 1] * W;
    sg.wire = a.wire.base + pkt_off(a.wire, s) + payload_pos(fmt, cols, a.field, s);
}

// Move one element at a time: chunk tails and 16-byte pieces that straddle packets.
template <int W, bool SWAP, bool ENCODE>
__device__ inline void copy_elems(Seg &sg, uint64_t b, uint64_t end, const CopyArgs &a,
                                  const FmtDev &fmt, const ColsDev &cols)
{
    typedef typename UIntOf<W>::type U;
    for (; b < end; b += W) {
        while (b >= sg.hi) seg_load<W>(sg, sg.s + 1, a, fmt, cols);   // skips empty arrays
        uint8_t *wp = sg.wire + (b - sg.lo);
        uint8_t *dp = a.dense + b;
        if (ENCODE) {
            st_u<U>(wp, ld_u<U>(dp));
        } else {
            U v = ld_u<U>(wp);
            if (SWAP) v = bswap_u(v);
            st_u<U>(dp, v);
        }
    }
}

template <int W, bool SWAP, bool ENCODE>
__global__ __launch_bounds__(MRPDR_TPB) void k_copy(CopyArgs a, FmtDev fmt, ColsDev cols)
{
    uint64_t total = a.seg_offs[a.n] * W;
    if (total > a.dense_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, 1u);
        return;
    }
    const uint64_t nchunks = (total + COPY_CHUNK - 1) / COPY_CHUNK;
    const uint64_t per_block = (nchunks + gridDim.x - 1) / gridDim.x;
    const uint64_t c0 = (uint64_t)blockIdx.x * per_block;
    const uint64_t c1 = (c0 + per_block < nchunks) ? c0 + per_block : nchunks;

    Seg blk;                        // segment the workgroup is streaming through (uniform)
    blk.s = 0; blk.lo = 1; blk.hi = 0; blk.wire = nullptr;
    bool have = false;

    for (uint64_t c = c0; c < c1; c++) {
        const uint64_t cb = c * COPY_CHUNK;
        const uint64_t ce = (cb + COPY_CHUNK < total) ? cb + COPY_CHUNK : total;
        if (!have || cb < blk.lo || cb >= blk.hi) {
            const uint64_t s = seg_find(a.seg_offs, have && cb >= blk.hi ? blk.s : 0, a.n - 1, cb / W);
            seg_load<W>(blk, s, a, fmt, cols);
            have = true;
        }
        if (ce <= blk.hi) {
            // whole chunk inside one packet: straight 128-bit streaming
            const uint8_t *wbase = blk.wire + (cb - blk.lo);
            u32x4 v[COPY_UNROLL];
            uint64_t b[COPY_UNROLL];
#pragma unroll
            for (int u = 0; u < COPY_UNROLL; u++) {
                b[u] = cb + ((uint64_t)u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
                if (b[u] + COPY_PIECE <= ce) {
                    if (ENCODE) v[u] = *reinterpret_cast<const u32x4 *>(a.dense + b[u]);
                    else        v[u] = ld_u<u32x4>(wbase + (b[u] - cb));
                }
            }
#pragma unroll
            for (int u = 0; u < COPY_UNROLL; u++) {
                if (b[u] + COPY_PIECE <= ce) {
                    if (ENCODE) st_u<u32x4>(const_cast<uint8_t *>(wbase) + (b[u] - cb), v[u]);
                    else        *reinterpret_cast<u32x4 *>(a.dense + b[u]) = SWAP ? swap_elems<W>(v[u]) : v[u];
                } else if (b[u] < ce) {
                    Seg t = blk;
                    copy_elems<W, SWAP, ENCODE>(t, b[u], ce, a, fmt, cols);
                }
            }
        } else {
            // chunk spans several packets: every lane locates its own piece
            const uint64_t s_last = seg_find(a.seg_offs, blk.s, a.n - 1, (ce - 1) / W);
#pragma unroll 1
            for (int u = 0; u < COPY_UNROLL; u++) {
                const uint64_t b = cb + ((uint64_t)u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
                if (b >= ce) continue;
                const uint64_t e = (b + COPY_PIECE < ce) ? b + COPY_PIECE : ce;
                Seg t;
                seg_load<W>(t, seg_find(a.seg_offs, blk.s, s_last, b / W), a, fmt, cols);
                if (e - b == COPY_PIECE && e <= t.hi) {
                    uint8_t *wp = t.wire + (b - t.lo);
                    if (ENCODE) {
                        st_u<u32x4>(wp, *reinterpret_cast<const u32x4 *>(a.dense + b));
                    } else {
                        u32x4 v = ld_u<u32x4>(wp);
                        *reinterpret_cast<u32x4 *>(a.dense + b) = SWAP ? swap_elems<W>(v) : v;
                    }
                } else {
                    copy_elems<W, SWAP, ENCODE>(t, b, e, a, fmt, cols);
                }
            }
        }
    }
}
