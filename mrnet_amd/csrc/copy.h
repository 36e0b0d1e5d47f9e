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
#include "rows.h"

#define COPY_UNROLL 4
#define COPY_PIECE 16u
#define COPY_CHUNK (MRPDR_TPB * COPY_PIECE * COPY_UNROLL)     // 16 KiB of dense bytes
#define COPY_LDS_SEGS 1024                                    // packets per chunk staged in LDS

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
    uint32_t pos_fixed;        // fixed bytes in front of the field's payload (scalars, prefixes, its own prefix)
    uint32_t var_before;       // bit g set: field g is variable and lies in front of it
};

// One 16 KiB chunk of the dense side.
struct ChunkDesc {
    uint64_t wire_off;     // arena offset of the wire byte that matches the chunk's first byte
                           // (chunks that span packets: index of the last packet they touch)
    uint32_t seg;          // packet the chunk starts in
    uint32_t info;         // bytes in the chunk (<= COPY_CHUNK) << 3 | DESC_* flags
};
#define DESC_SINGLE 1u     // the chunk lies inside one packet
#define DESC_HEAD   2u     // encode: the bytes in front of the first aligned wire piece are ours to write
#define DESC_EXTEND 4u     // encode: the last aligned wire piece may run into the next chunk's bytes
// a full chunk inside one packet with nothing special at either end
#define DESC_COMMON ((COPY_CHUNK << 3) | DESC_EXTEND | DESC_SINGLE)

// First segment whose dense range ends after element index e (offs[s+1] > e).
__device__ inline uint64_t seg_find(const uint64_t *offs, uint64_t lo, uint64_t hi, uint64_t e)
{
    while (lo < hi) {                       // invariant: answer in [lo, hi]
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (offs[mid + 1] > e) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// The same over all n segments, starting from a guess: the packets of a wave are often all
// alike, and then the packet of element e is e * n / total.  Two probes (one round trip) settle
// that case; otherwise the bracket is widened in doubling steps from the guess and the binary
// search runs inside it.  offs[0] = 0 and e < total = offs[n].
__device__ inline uint64_t seg_find_near(const uint64_t *offs, uint64_t n, uint64_t total, uint64_t e)
{
    uint64_t g = (uint64_t)((double)e * (double)n / (double)total);
    if (g > n - 1) g = n - 1;
    const uint64_t a = offs[g], b = offs[g + 1];
    if (a <= e && e < b) return g;          // no segment in front of g ends after e: offs[g] <= e
    uint64_t lo, hi, step = 1;
    if (e < a) {                            // in front of g (g > 0, as offs[0] <= e)
        hi = g - 1;                         // kept true: offs[hi + 1] > e
        for (;;) {
            lo = hi + 1 >= step ? hi + 1 - step : 0;
            if (lo == 0 || offs[lo] <= e) break;
            hi = lo - 1;
            step <<= 1;
        }
    } else {                                // behind g
        lo = g + 1;                         // kept true: offs[lo] <= e
        for (;;) {
            hi = lo + step - 1 < n - 1 ? lo + step - 1 : n - 1;
            if (hi == n - 1 || offs[hi + 1] > e) break;
            lo = hi + 1;
            step <<= 1;
        }
    }
    return seg_find(offs, lo, hi, e);
}

struct Seg {
    uint64_t s;          // segment (= packet) index
    uint64_t lo, hi;     // dense byte range
    uint8_t *wire;       // wire address of dense byte `lo`
};

__device__ inline void seg_load(Seg &sg, uint64_t s, uint32_t w, const CopyArgs &a, const FmtDev &fmt,
                                const ColsDev &cols)
{
    // what depends on the format alone was worked out on the host (pos_fixed, var_before): a
    // format whose only variable field is this one needs no look at the format here, and the
    // three loads below are independent
    sg.s = s;
    const uint64_t o0 = a.seg_offs[s], o1 = a.seg_offs[s + 1];
    uint64_t pos = pkt_off(a.wire, s) + a.pos_fixed;
    for (uint32_t rest = a.var_before; rest; rest &= rest - 1)
        pos += var_bytes(fmt, cols, __builtin_ctz(rest), s);
    sg.lo = o0 * w;
    sg.hi = o1 * w;
    sg.wire = a.wire.base + pos;
}

// One lane per chunk: locate it.
//
// Encode stores go to the wire, whose addresses have arbitrary alignment.  A misaligned
// 128-bit store stream costs about 8 % of the kernel (partial cache lines at every
// wavefront boundary), so an encode chunk stores 16-byte ALIGNED wire pieces and loads the
// dense side misaligned instead (misaligned loads are free).  With delta = bytes from the
// chunk's first wire byte to the next 16-byte boundary, piece k covers wire bytes
// [delta + 16k, delta + 16k + 16).  Piece 1023 runs delta bytes into the next chunk
// (DESC_EXTEND: allowed when those bytes are in the same packet); the delta bytes in front
// of piece 0 were then written by the previous chunk, otherwise DESC_HEAD asks for them.
template <bool ENCODE>
__device__ __forceinline__ void plan_chunk(const CopyArgs &a, const FmtDev &fmt, const ColsDev &cols,
                                           ChunkDesc *desc, uint64_t k)
{
    const uint32_t w = fmt.f[a.field].width;
    const uint64_t total_elems = a.seg_offs[a.n];
    const uint64_t total = total_elems * w;
    ChunkDesc d;
    d.wire_off = 0; d.seg = 0; d.info = 0;
    const uint64_t cb = (a.first_chunk + k) * COPY_CHUNK;
    if (total > a.dense_cap) {
        if (k == 0) atomicOr(a.status, 1u);
    } else if (cb < total) {
        const uint64_t ce = (cb + COPY_CHUNK < total) ? cb + COPY_CHUNK : total;
        Seg sg;
        seg_load(sg, seg_find_near(a.seg_offs, a.n, total_elems, cb / w), w, a, fmt, cols);
        d.wire_off = (uint64_t)(sg.wire - a.wire.base) + (cb - sg.lo);
        d.seg = (uint32_t)sg.s;
        uint32_t flags = ce <= sg.hi ? DESC_SINGLE : 0u;
        const uint32_t delta = (uint32_t)(0 - (uintptr_t)(a.wire.base + d.wire_off)) & 15u;
        // a chunk that spans packets has no single wire address: the slot carries its last packet
        if (!flags) d.wire_off = seg_find_near(a.seg_offs, a.n, total_elems, (ce - 1) / w);
        if (!ENCODE || delta == 0) {
            flags |= DESC_EXTEND;
        } else if (flags && ce - cb == COPY_CHUNK) {
            if (ce + 16 <= sg.hi) flags |= DESC_EXTEND;
            const bool prev_extends = cb >= COPY_CHUNK && cb - COPY_CHUNK >= sg.lo && cb + 16 <= sg.hi;
            if (!prev_extends) flags |= DESC_HEAD;
        }
        d.info = ((uint32_t)(ce - cb) << 3) | flags;
    }
    desc[k] = d;
}

template <bool ENCODE>
__global__ __launch_bounds__(MRPDR_TPB) void k_copy_plan(CopyArgs a, FmtDev fmt, ColsDev cols,
                                                         ChunkDesc *desc, uint64_t ndesc)
{
    const uint64_t k = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (k < ndesc) plan_chunk<ENCODE>(a, fmt, cols, desc, k);
}

// Encode: the prefix walk (one lane per packet) and the chunk plan of one field (one lane per
// chunk) need the same inputs and nothing of each other: one launch for both.
__global__ __launch_bounds__(MRPDR_TPB) void k_walk_plan_encode(FmtDev fmt, WireDev wire, ColsDev cols, uint64_t n,
                                                                CopyArgs a, ChunkDesc *desc, uint64_t ndesc)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i < n) walk_encode_packet(fmt, wire, cols, i);
    if (i < ndesc) plan_chunk<true>(a, fmt, cols, desc, i);
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

// What one workgroup needs to stream a chunk that lies inside one packet: kept first
// and small so the kernel prologue is two scalar loads (arguments, descriptor).
struct CopyFast {
    const ChunkDesc *desc;
    uint8_t *wire_base;
    uint8_t *dense_base;       // column base + first_chunk * COPY_CHUNK
};

template <int W, bool SWAP, bool ENCODE, bool NT>
__global__ __launch_bounds__(MRPDR_TPB) void k_copy(CopyFast fa, CopyArgs a, FmtDev fmt, ColsDev cols)
{
    typedef typename UIntOf<W>::type U;
    // Chunk order = workgroup order: the workgroups in flight sweep one contiguous window of
    // HBM.  (Measured alternatives: a contiguous run of chunks per workgroup or per XCD is
    // 5-8 % slower for this kernel.)
    const uint32_t chunk = blockIdx.x;
    // one 16-byte scalar load: {wire_off, seg | info << 32}
    const ulonglong2 raw = reinterpret_cast<const ulonglong2 *>(fa.desc)[chunk];
    const uint32_t info = (uint32_t)(raw.y >> 32);
    uint8_t *dense = fa.dense_base + (uint64_t)chunk * COPY_CHUNK;

    if ((info | DESC_HEAD | DESC_EXTEND) == (DESC_COMMON | DESC_HEAD)) {
        // a full chunk inside one packet (nearly all of a large-array wave): four 128-bit
        // loads, then four stores, no branches in the decode direction
        uint8_t *wire = fa.wire_base + raw.x;
        // encode: shift to the aligned wire pieces (see k_copy_plan); decode: dense is aligned
        const uint32_t delta = ENCODE ? ((uint32_t)(0 - (uintptr_t)wire) & 15u) : 0u;
        // encode, last chunk of a packet: the last piece would run past the packet
        const bool whole_last = !ENCODE || (info & DESC_EXTEND) || threadIdx.x != MRPDR_TPB - 1;
        u32x4 v[COPY_UNROLL];
#pragma unroll
        for (int u = 0; u < COPY_UNROLL; u++) {
            const uint32_t b = (u * MRPDR_TPB + threadIdx.x) * COPY_PIECE + delta;
            if (u < COPY_UNROLL - 1 || whole_last) v[u] = ENCODE ? ld16_any<NT>(dense + b) : ld16_any<NT>(wire + b);
        }
#pragma unroll
        for (int u = 0; u < COPY_UNROLL; u++) {
            const uint32_t b = (u * MRPDR_TPB + threadIdx.x) * COPY_PIECE + delta;
            if (u < COPY_UNROLL - 1 || whole_last) {
                if (ENCODE) st16<NT>(wire + b, v[u]);
                else        st16<NT>(dense + b, SWAP ? swap_elems<W>(v[u]) : v[u]);
            }
        }
        if (ENCODE) {
            if (!whole_last)
                for (uint32_t e = COPY_CHUNK - COPY_PIECE + delta; e < COPY_CHUNK; e++) wire[e] = dense[e];
            if ((info & DESC_HEAD) && threadIdx.x < delta) wire[threadIdx.x] = dense[threadIdx.x];
        }
        return;
    }

    const uint32_t len = info >> 3;
    if (len == 0) return;
    const uint32_t seg = (uint32_t)raw.y;
    const uint64_t cb = (a.first_chunk + chunk) * COPY_CHUNK;

    if (info & DESC_SINGLE) {
        // a short chunk inside one packet: the last one of the column
        uint8_t *wire = fa.wire_base + raw.x;
#pragma unroll 1
        for (int u = 0; u < COPY_UNROLL; u++) {
            const uint32_t b = (u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
            if (b + COPY_PIECE <= len) {
                if (ENCODE) st16_any<false>(wire + b, ld16<false>(dense + b));
                else {
                    u32x4 x = ld16_any<false>(wire + b);
                    st16<false>(dense + b, SWAP ? swap_elems<W>(x) : x);
                }
            } else if (b < len) {
                for (uint32_t e = b; e < len; e += W) {
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

    // chunk spans several packets.  The workgroup first loads the boundaries and wire addresses
    // of all packets that touch the chunk into LDS (one packet per lane, in parallel); every
    // lane then locates its own 16-byte piece with a binary search in LDS.
    const uint64_t ce = cb + len;
    const uint64_t s_last = raw.x;
    const uint64_t nseg = s_last - seg + 1;
    __shared__ uint64_t s_lo[COPY_LDS_SEGS + 1];
    __shared__ uint64_t s_woff[COPY_LDS_SEGS];      // arena offsets (an address read back from LDS would
                                                    // compile to flat accesses)
    if (nseg <= COPY_LDS_SEGS) {
        for (uint32_t i = threadIdx.x; i < nseg; i += MRPDR_TPB) {
            Seg t;
            seg_load(t, seg + i, W, a, fmt, cols);
            s_lo[i] = t.lo;
            s_woff[i] = (uint64_t)(t.wire - a.wire.base);
            if (i + 1 == nseg) s_lo[nseg] = t.hi;
        }
        __syncthreads();
        // locate the four pieces and issue their loads together, then store
        uint32_t at[COPY_UNROLL];
        bool whole[COPY_UNROLL];
        u32x4 v[COPY_UNROLL];
#pragma unroll
        for (int u = 0; u < COPY_UNROLL; u++) {
            const uint64_t b = cb + ((uint64_t)u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
            whole[u] = false;
            at[u] = 0;
            if (b >= ce) continue;
            uint32_t lo = 0, hi = (uint32_t)nseg - 1;          // first i with s_lo[i+1] > b
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_lo[mid + 1] > b) hi = mid; else lo = mid + 1;
            }
            at[u] = lo;
            whole[u] = b + COPY_PIECE <= ce && b + COPY_PIECE <= s_lo[lo + 1];
            if (whole[u]) v[u] = ENCODE ? ld16<false>(a.dense + b) : ld16_any<false>(fa.wire_base + s_woff[lo] + (b - s_lo[lo]));
        }
#pragma unroll
        for (int u = 0; u < COPY_UNROLL; u++) {
            const uint64_t b = cb + ((uint64_t)u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
            if (b >= ce) continue;
            uint32_t lo = at[u];
            if (whole[u]) {
                if (ENCODE) st16_any<false>(fa.wire_base + s_woff[lo] + (b - s_lo[lo]), v[u]);
                else        st16<false>(a.dense + b, SWAP ? swap_elems<W>(v[u]) : v[u]);
            } else {
                const uint64_t e = (b + COPY_PIECE < ce) ? b + COPY_PIECE : ce;
                for (uint64_t p = b; p < e; p += W) {
                    while (p >= s_lo[lo + 1]) lo++;             // next packet (skips empty arrays)
                    uint8_t *wp = fa.wire_base + s_woff[lo] + (p - s_lo[lo]);
                    if (ENCODE) {
                        st_u<U>(wp, ld_u<U>(a.dense + p));
                    } else {
                        U x = ld_u<U>(wp);
                        if (SWAP) x = bswap_u(x);
                        st_u<U>(a.dense + p, x);
                    }
                }
            }
        }
        return;
    }

    // more packets in one chunk than the LDS table holds (arrays of a few bytes): every lane
    // searches the offsets in global memory
#pragma unroll 1
    for (int u = 0; u < COPY_UNROLL; u++) {
        const uint64_t b = cb + ((uint64_t)u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
        if (b >= ce) continue;
        const uint64_t e = (b + COPY_PIECE < ce) ? b + COPY_PIECE : ce;
        Seg t;
        seg_load(t, seg_find(a.seg_offs, seg, s_last, b / W), W, a, fmt, cols);
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

// ---------------------------------------------------------------------------
// Short payloads: packet-centric copy
// ---------------------------------------------------------------------------
// When the packets of a wave carry short arrays or strings (tens to hundreds of bytes) a
// 16 KiB chunk holds dozens to thousands of them and the column is only a few hundred
// chunks long: the chunk kernel above is short of parallelism and spends its time finding
// packet boundaries.  Here 16 lanes take one packet: 128-bit pieces while they fit, then
// the last few elements one per lane.  n/16 wavefront quarters in flight instead of
// bytes/16Ki workgroups.
#define PKT_LANES 16u

template <int W, bool SWAP, bool ENCODE>
__global__ __launch_bounds__(MRPDR_TPB) void k_copy_packets(CopyArgs a, FmtDev fmt, ColsDev cols)
{
    typedef typename UIntOf<W>::type U;
    const uint64_t s = ((uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x) / PKT_LANES;
    const uint32_t lane = threadIdx.x % PKT_LANES;
    if (a.seg_offs[a.n] * W > a.dense_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, 1u);
        return;
    }
    if (s >= a.n) return;
    Seg sg;
    seg_load(sg, s, W, a, fmt, cols);
    const uint64_t len = sg.hi - sg.lo;
    uint8_t *dense = a.dense + sg.lo;
    for (uint64_t base = 0; base < len; base += PKT_LANES * COPY_PIECE) {
        const uint64_t rem = len - base;
        const uint32_t full = rem >= PKT_LANES * COPY_PIECE ? PKT_LANES : (uint32_t)(rem / COPY_PIECE);
        if (lane < full) {
            const uint64_t b = base + lane * COPY_PIECE;
            if (ENCODE) st16_any<false>(sg.wire + b, ld16_any<false>(dense + b));
            else {
                u32x4 x = ld16_any<false>(sg.wire + b);
                st16_any<false>(dense + b, SWAP ? swap_elems<W>(x) : x);
            }
        } else {
            // the lanes without a whole piece share the last (< 16 byte) run, one element each
            const uint64_t t0 = base + (uint64_t)full * COPY_PIECE;
            for (uint64_t e = t0 + (uint64_t)(lane - full) * W; e < len; e += (uint64_t)(PKT_LANES - full) * W) {
                if (ENCODE) {
                    st_u<U>(sg.wire + e, ld_u<U>(dense + e));
                } else {
                    U x = ld_u<U>(sg.wire + e);
                    if (SWAP) x = bswap_u(x);
                    st_u<U>(dense + e, x);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Waves with one byte order per packet
// ---------------------------------------------------------------------------
// A wave collected from children of both byte orders is decoded as if every sender were
// little-endian (the kernels above, no reversal), then the elements of the packets that came
// from the other kind are reversed in place: one wavefront per such packet, 128-bit pieces,
// then the last few elements one per lane.  The extra pass reads and writes only those packets.
template <int W>
__global__ __launch_bounds__(MRPDR_TPB) void k_swap_foreign(uint8_t *dense, const uint64_t *offs, const uint8_t *orders,
                                                            uint64_t n, uint64_t dense_cap)
{
    typedef typename UIntOf<W>::type U;
    const uint64_t s = ((uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (s >= n || !orders[s] || offs[n] * W > dense_cap) return;
    uint8_t *p = dense + offs[s] * W;
    const uint64_t len = (offs[s + 1] - offs[s]) * W;
    const uint64_t whole = len & ~15ull;
    for (uint64_t b = (uint64_t)lane * COPY_PIECE; b < whole; b += 64 * COPY_PIECE)
        st16_any<false>(p + b, swap_elems<W>(ld16_any<false>(p + b)));
    for (uint64_t e = whole + (uint64_t)lane * W; e < len; e += 64 * W) st_u<U>(p + e, bswap_u(ld_u<U>(p + e)));
}

// ---------------------------------------------------------------------------
// Calibration: the same streaming loop with no codec logic
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(MRPDR_TPB) void k_plain_copy(uint8_t *dst, const uint8_t *src, uint64_t bytes)
{
    const uint64_t base = (uint64_t)blockIdx.x * COPY_CHUNK;
    u32x4 v[COPY_UNROLL];
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; u++) {
        const uint64_t b = base + (uint64_t)(u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
        if (b < bytes) v[u] = ld16<true>(src + b);
    }
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; u++) {
        const uint64_t b = base + (uint64_t)(u * MRPDR_TPB + threadIdx.x) * COPY_PIECE;
        if (b < bytes) st16<true>(dst + b, v[u]);
    }
}
