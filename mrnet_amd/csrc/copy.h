// Array and string payloads: the streaming kernels where the bytes go.
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
// The dense side is cut into 16 KiB chunks.  k_copy_plan writes one 16-byte descriptor
// per chunk (which packet it starts in, where its first byte lives on the wire, whether
// it stays inside that packet); k_copy runs one workgroup per chunk, in chunk order so
// that the workgroups in flight sweep a contiguous window of HBM: a single descriptor
// load, then 4 x 128-bit loads and stores per lane with nontemporal hints (the data is
// touched once).  Elements are byte-reversed in registers (v_perm_b32) when the sender's
// order differs.  Measured on MI355X: the same loop without any codec logic moves
// 6.3 TB/s (read+write), misaligned wire addresses and the byte reversal cost nothing.
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

// 16-byte accesses.  The wire side may sit at any byte address; the dense side is
// always 16-byte aligned.  NT marks the access nontemporal.
template <bool NT> __device__ inline u32x4 ld16_any(const uint8_t *p)
{
    typedef u32x4 u32x4_u __attribute__((aligned(1)));
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(p));
    return *reinterpret_cast<const u32x4_u *>(p);
}
template <bool NT> __device__ inline void st16_any(uint8_t *p, u32x4 v)
{
    typedef u32x4 u32x4_u __attribute__((aligned(1)));
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_u *>(p));
    else *reinterpret_cast<u32x4_u *>(p) = v;
}
template <bool NT> __device__ inline u32x4 ld16(const uint8_t *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return *reinterpret_cast<const u32x4 *>(p);
}
template <bool NT> __device__ inline void st16(uint8_t *p, u32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
    else *reinterpret_cast<u32x4 *>(p) = v;
}

struct CopyArgs {
    WireDev  wire;
    uint8_t *dense;            // values column, 16-byte aligned
    const uint64_t *seg_offs;  // offs[n+1] in elements
    uint64_t n;
    uint64_t dense_cap;        // bytes available in dense
    uint64_t first_chunk;      // chunk index of descriptor 0 (windows of very large columns)
    uint32_t *status;          // bit 0 set when the column is too small
    int32_t  field;
};

// One 16 KiB chunk of the dense side.
struct ChunkDesc {
    uint64_t wire_off;     // arena offset of the wire byte that matches the chunk's first byte
    uint32_t seg;          // packet the chunk starts in
    uint32_t len_single;   // bytes in the chunk (<= COPY_CHUNK) << 1 | 1 when inside one packet
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

__device__ inline void seg_load(Seg &sg, uint64_t s, uint32_t w, const CopyArgs &a, const FmtDev &fmt,
                                const ColsDev &cols)
{
    sg.s = s;
    sg.lo = a.seg_offs[s] * w;
    sg.hi = a.seg_offs[s + 1] * w;
    sg.wire = a.wire.base + pkt_off(a.wire, s) + payload_pos(fmt, cols, a.field, s);
}

// One lane per chunk: locate it.
__global__ __launch_bounds__(MRPDR_TPB) void k_copy_plan(CopyArgs a, FmtDev fmt, ColsDev cols,
                                                         ChunkDesc *desc, uint64_t ndesc)
{
    const uint64_t k = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (k >= ndesc) return;
    const uint32_t w = fmt.f[a.field].width;
    const uint64_t total = a.seg_offs[a.n] * w;
    ChunkDesc d;
    d.wire_off = 0; d.seg = 0; d.len_single = 0;
    const uint64_t cb = (a.first_chunk + k) * COPY_CHUNK;
    if (total > a.dense_cap) {
        if (k == 0) atomicOr(a.status, 1u);
    } else if (cb < total) {
        const uint64_t ce = (cb + COPY_CHUNK < total) ? cb + COPY_CHUNK : total;
        Seg sg;
        seg_load(sg, seg_find(a.seg_offs, 0, a.n - 1, cb / w), w, a, fmt, cols);
        d.wire_off = (uint64_t)(sg.wire - a.wire.base) + (cb - sg.lo);
        d.seg = (uint32_t)sg.s;
        d.len_single = ((uint32_t)(ce - cb) << 1) | (ce <= sg.hi ? 1u : 0u);
    }
    desc[k] = d;
}

// Move one element at a time: chunk tails and 16-byte pieces that straddle packets.
template <int W, bool SWAP, bool ENCODE>
__device__ inline void copy_elems(Seg &sg, uint64_t b, uint64_t end, const CopyArgs &a,
                                  const FmtDev &fmt, const ColsDev &cols)
{
    typedef typename UIntOf<W>::type U;
    for (; b < end; b += W) {
        while (b >= sg.hi) seg_load(sg, sg.s + 1, W, a, fmt, cols);   // skips empty arrays
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

template <int W, bool SWAP, bool ENCODE, bool NT>
__global__ __launch_bounds__(MRPDR_TPB) void k_copy(CopyArgs a, FmtDev fmt, ColsDev cols, const ChunkDesc *desc)
{
    typedef typename UIntOf<W>::type U;
    const ChunkDesc d = desc[blockIdx.x];
    const uint32_t len = d.len_single >> 1;
    if (len == 0) return;
    const uint64_t cb = (a.first_chunk + blockIdx.x) * COPY_CHUNK;
    uint8_t *dense = a.dense + cb;

    if (d.len_single & 1u) {
        // whole chunk inside one packet: straight 128-bit streaming
        uint8_t *wire = a.wire.base + d.wire_off;
        u32x4 v[COPY_UNROLL];
#pragma unroll
        for (int u = 0; u < COPY_UNROLL; u++) {
            const uint32_t b = (u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
            if (b + COPY_PIECE <= len) v[u] = ENCODE ? ld16<NT>(dense + b) : ld16_any<NT>(wire + b);
        }
#pragma unroll
        for (int u = 0; u < COPY_UNROLL; u++) {
            const uint32_t b = (u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
            if (b + COPY_PIECE <= len) {
                if (ENCODE) st16_any<NT>(wire + b, v[u]);
                else        st16<NT>(dense + b, SWAP ? swap_elems<W>(v[u]) : v[u]);
            } else if (b < len) {
                for (uint32_t e = b; e < len; e += W) {            // tail of the column
                    if (ENCODE) {
                        st_u<U>(wire + e, ld_u<U>(dense + e));
                    } else {
                        U x = ld_u<U>(wire + e);
                        if (SWAP) x = bswap_u(x);
                        st_u<U>(dense + e, x);
                    }
                }
            }
        }
        return;
    }

    // chunk spans several packets: every lane locates its own piece
    const uint64_t ce = cb + len;
    const uint64_t s_last = seg_find(a.seg_offs, d.seg, a.n - 1, (ce - 1) / W);
#pragma unroll 1
    for (int u = 0; u < COPY_UNROLL; u++) {
        const uint64_t b = cb + ((uint64_t)u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
        if (b >= ce) continue;
        const uint64_t e = (b + COPY_PIECE < ce) ? b + COPY_PIECE : ce;
        Seg t;
        seg_load(t, seg_find(a.seg_offs, d.seg, s_last, b / W), W, a, fmt, cols);
        if (e - b == COPY_PIECE && e <= t.hi) {
            uint8_t *wp = t.wire + (b - t.lo);
            if (ENCODE) st16_any<false>(wp, ld16<false>(a.dense + b));
            else {
                u32x4 x = ld16_any<false>(wp);
                st16<false>(a.dense + b, SWAP ? swap_elems<W>(x) : x);
            }
        } else {
            copy_elems<W, SWAP, ENCODE>(t, b, e, a, fmt, cols);
        }
    }
}
