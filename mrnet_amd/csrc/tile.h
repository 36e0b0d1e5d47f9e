// Waves of SHORT variable-length packets (tens to hundreds of bytes each): LDS-staged codec.
//
// Reference work per packet: pdr_packet_data walks the fields one pdr_* call at a time
// (src/Packet.C:594-845), strings and small arrays through pdr_string / pdr_bytes / pdr_array
// (src/pdr.c:212-333, 415-472) with a malloc each.
//
// Here a workgroup owns a TILE of 64 consecutive packets.  In the arena those packets are
// neighbours, and so are their elements in every values column, so both sides of the tile are
// two short contiguous byte ranges.  The wire range is staged in LDS: global memory sees only
// 128-bit accesses to consecutive addresses (the wire image is loaded / stored as a block, the
// columns are read / written as blocks), and all the irregular work -- walking fields at a
// running cursor, finding which packet a byte belongs to -- happens in LDS.
//
//   encode  k_tile_encode   columns -> LDS image of the tile's packets -> wire        (1 launch)
//   decode  k_tile_walk     wire -> LDS; scalars, counts, ok, per-tile count sums     (begin)
//           k_tile_offs     counts + tile sums -> offs[] of every variable field      (begin)
//           k_tile_decode   wire -> LDS; payloads -> values columns                   (finish)
//   decode in one call: k_count_walk (length prefixes only, nothing staged) in place of
//           k_tile_walk; k_tile_decode then writes the scalar columns too
// The image is 24 KiB (five workgroups per CU) or, for waves whose 64-packet spans are longer,
// 32 KiB (four): a template parameter of the kernels, chosen by the host per wave.
//
// What a tile costs is a chain of dependent memory round trips, not bytes: measured on a wave of
// 2^20 packets the HBM traffic is the algorithmic traffic and nothing in the memory system is
// busy, but a wavefront waits for 85 % of its life.  So every round issues all its loads before
// the first of them is waited for (metadata; the wire image; the column loads of a field, the
// first field's together with the scalars), what depends on the format alone comes from scalar
// loads (one wavefront per field, one lane per packet), and conditional loads are avoided
// where a clamped address will do: a predicated load compiles to a branch and a wait.
//
// A tile whose packets do not fit the LDS image (a giant packet among small ones), overlap or
// are out of order runs the same code straight on global memory: slower, same result.
// Formats with string arrays or more than TILE_MAXV variable fields take the generic kernels.
#pragma once

#include "common.h"
#include "copy.h"
#include "rows.h"
#include "scan.h"

#ifndef TILE_PKTS
#define TILE_PKTS 64                   // measured: 32-packet tiles are 25-45 % slower
#endif
#ifndef TILE_TPB
#define TILE_TPB MRPDR_TPB             // threads per workgroup of the tile kernels
#endif
#define TILE_MAXV 8
#define TILE_IMG (TILE_PKTS * 384)      // the LDS image of a tile: 24 KiB, five workgroups per CU
#define TILE_IMG_WIDE (TILE_PKTS * 512) // for waves whose 64-packet spans are longer: 32 KiB, four per CU
#define TILE_OFFS_PKTS 1024            // packets per workgroup of k_tile_offs
static_assert(TILE_IMG % (TILE_TPB * 16) == 0 && TILE_IMG_WIDE % (TILE_TPB * 16) == 0,
              "the image is a whole number of workgroup-wide rows of 16-byte pieces");
static_assert(TILE_PKTS <= 64 && TILE_PKTS * 4 == TILE_TPB, "a lane of a wavefront per packet; sixteen lanes move a packet, four packets each");

// Diagnostic build only (tools/tile_stamps.py builds the library with -DMRPDR_TILE_STAMPS): lane 0 of
// every workgroup notes the 100 MHz wall clock at the phase boundaries of the tile kernels.
#ifdef MRPDR_TILE_STAMPS
__device__ unsigned long long *g_tile_stamps;
#define TILE_STAMP(k) do { if (threadIdx.x == 0 && g_tile_stamps) g_tile_stamps[(uint64_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TILE_STAMP(k) do { } while (0)
#endif

struct TileVars {                       // the variable fields of the format, in order
    int32_t nv;
    int32_t field[TILE_MAXV];
    int32_t width[TILE_MAXV];           // element width of each (kept here: one scalar load, no chain)
    // Of the wave's columns, per launch (tile_vars_of): offs[] and capacity of variable field v under
    // the index v itself.  cols.offs[tv.field[v]] is two look-ups into the kernel arguments, the
    // second at an index that the first one loaded -- two dependent scalar loads in front of the
    // load that is wanted (and, indexed per lane, two dependent round trips to memory).
    uint64_t *offs[TILE_MAXV];
    uint64_t  cap[TILE_MAXV];
};

__host__ inline TileVars tile_vars_of(const TileVars &fmt_vars, const ColsDev &cols)
{
    TileVars tv = fmt_vars;
    for (int v = 0; v < TILE_MAXV; v++) {
        tv.offs[v] = v < tv.nv ? cols.offs[tv.field[v]] : nullptr;
        tv.cap[v] = v < tv.nv ? cols.cap[tv.field[v]] : 0;
    }
    return tv;
}

// MAXV = variable fields the tables hold: formats with up to 4 take the small form, which leaves
// room for a fifth workgroup per CU (measured on a 1M-packet wave: +11 % throughput).
template <int MAXV>
struct TileLdsT {
    static constexpr int NV = MAXV;
    uint64_t start[TILE_PKTS + 1];      // arena offset of every packet of the tile
    uint64_t end[TILE_PKTS];
    uint64_t seg[MAXV][TILE_PKTS + 1];  // offs[] of every variable field, in elements
    uint64_t vpos[MAXV][TILE_PKTS];     // position of the field's length prefix in its packet
    int32_t  staged;
    int32_t  packed;                    // every packet starts where the one before it ends
    int32_t  too_small;                 // decode: a values column cannot hold the wave
};
#define TILE_SMALLV 4

// Address of arena byte x: in the LDS image (the image keeps the global address modulo 16, so
// aligned 128-bit pieces of global memory are aligned pieces of the image) or in the arena.
template <bool STAGED>
__device__ __forceinline__ uint8_t *tile_at(uint8_t *img, uint8_t *gbase, uint64_t lo, uint32_t mis, uint64_t x)
{
    return STAGED ? img + (uint32_t)(x - lo) + mis : gbase + x;
}

__device__ __forceinline__ u32x4 ld16_u(const uint8_t *p)
{
    u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void st16_u(uint8_t *p, u32x4 v)
{
    __builtin_memcpy(p, &v, 16);
}

// Packet boundaries (and, with SEGS, the offs[] of the variable fields) of the tile into LDS
// in one round of global loads; decides whether the tile can be staged.
// With CAPS the capacities of the values columns are checked against the wave's totals in the
// same round (offs[n] of every variable field).
struct TileNothing { __device__ __forceinline__ void operator()() const {} };

// after_loads(): called when the round's loads have arrived and are on their way into LDS, before the
// barriers and the checks: loads issued there are in flight while the boundaries are checked.  (Issued
// any earlier they would sit in front of nothing -- loads return in issue order -- but the compiler
// has to wait for ALL loads where a conditional one may or may not have been issued.)
template <bool SEGS, bool CAPS, int IMG, typename L, typename F = TileNothing>
__device__ __forceinline__ void tile_load_meta(L &m, const TileVars &tv, const WireDev &wire,
                                               uint64_t p0, uint32_t np, uint64_t n, F after_loads = F())
{
    const uint32_t t = threadIdx.x;
    if (t == 0) { m.staged = 1; m.packed = 1; m.too_small = 0; }
    // (the same for every lane of every tile: scalar loads, issued here and not needed before the
    // second barrier.  As a per-lane look-up -- lane v asking for cols.offs[tv.field[v]][n] -- this was
    // three dependent round trips to memory that the last wavefront brought to the first barrier.)
    bool over = false;
    if (CAPS) {
#pragma unroll
        for (int v = 0; v < L::NV; v++)
            if (v < tv.nv) over = over || tv.offs[v][n] > tv.cap[v];
    }
    // every load of the round is issued before the first LDS store waits for one of them
    uint64_t s0 = 0, l0 = 0;
    if (t < np) { s0 = wire.offsets[p0 + t]; l0 = wire.lens[p0 + t]; }
    uint64_t sg[L::NV];
    if (SEGS) {
#pragma unroll
        for (int v = 0; v < L::NV; v++) sg[v] = (v < tv.nv && t <= np) ? tv.offs[v][p0 + t] : 0;
    }
    if (t < np) {
        m.start[t] = s0;
        m.end[t] = s0 + l0;
    }
    if (SEGS) {
#pragma unroll
        for (int v = 0; v < L::NV; v++)
            if (v < tv.nv && t <= np) m.seg[v][t] = sg[v];
    }
    after_loads();
    __syncthreads();
    if (t < np) {
        const uint64_t s = m.start[t], e = m.end[t];
        bool fine = e >= s && s <= wire.bytes && e - s <= wire.bytes - s;      // inside the arena
        if (t + 1 < np && m.start[t + 1] < e) fine = false;                    // in order, no overlap
        if (t == 0 && (m.end[np - 1] < s || m.end[np - 1] - s + 32 > IMG)) fine = false;   // fits the image
        if (!fine) m.staged = 0;
        if (t + 1 < np && m.start[t + 1] != e) m.packed = 0;
        if (SEGS && CAPS) {
            // offs[] is the caller's between the two halves of a decode: whatever it holds, no packet
            // moves more bytes than it has, backwards, or beyond the column
            bool sane = true;
            const uint64_t room = (e >= s && s <= wire.bytes && e - s <= wire.bytes - s) ? e - s : 0;
            for (int v = 0; v < tv.nv; v++) {
                const uint64_t a = m.seg[v][t], b = m.seg[v][t + 1];
                sane = sane && b >= a && (b - a) <= room && (b - a) * (uint64_t)tv.width[v] <= room && b <= tv.cap[v];
            }
            if (!sane) m.too_small = 1;
        }
    }
    if (over) m.too_small = 1;
    __syncthreads();
}

// Bytes of the variable fields in front of field f in packet i of the tile.
template <typename L>
__device__ __forceinline__ uint64_t tile_var_before(const L &m, const TileVars &tv, const FmtDev &,
                                                    int f, uint32_t i)
{
    uint64_t b = 0;
    for (int v = 0; v < tv.nv && tv.field[v] < f; v++)
        b += (m.seg[v][i + 1] - m.seg[v][i]) * (uint64_t)tv.width[v];
    return b;
}

// Copy of the tile's wire range into the image: aligned 128-bit pieces (the few bytes in front
// of the first packet and behind the last one that share their 16-byte granule come along).
// A lane has at most IMG / (TILE_TPB * 16) = 6 (or 8) pieces; their loads are all issued before
// the first LDS store, so the lane waits for memory once, not once per piece.  (As compiled
// before, load - wait - store per piece, a 19 KiB tile cost five round trips in a row.)
template <int IMG>
__device__ __forceinline__ void tile_stage_in(uint8_t *img, const uint8_t *gbase, uint64_t lo, uint64_t hi, uint32_t mis)
{
    constexpr int PIECES = IMG / (TILE_TPB * 16);
    const uint8_t *g = gbase + lo - mis;
    const uint32_t bytes = (uint32_t)(hi - lo) + mis;          // at most IMG - 16 (see tile_load_meta)
    if (bytes == 0) return;
    // loads are unconditional (a lane past the end re-reads the last piece): predicated loads
    // compile to a branch and a wait each, which is the chain this is here to avoid
    const uint32_t last = (bytes - 1u) & ~15u;
    u32x4 x[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        const uint32_t o = ((uint32_t)k * TILE_TPB + threadIdx.x) * 16u;
        x[k] = *reinterpret_cast<const u32x4 *>(g + (o < last ? o : last));
    }
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        const uint32_t o = ((uint32_t)k * TILE_TPB + threadIdx.x) * 16u;
        if (o < bytes) *reinterpret_cast<u32x4 *>(img + o) = x[k];
    }
}

// The same in two steps, so that the loads can be on their way while something else is done: the
// tile's byte range needs only the first packet's start and the last packet's end (two uniform
// loads), not the checked boundaries of all 64 -- reading the range early is harmless even if the
// tile then turns out not to be stageable.
template <int IMG> struct TileStage {
    static constexpr int PIECES = IMG / (TILE_TPB * 16);
    u32x4 x[PIECES];
    uint32_t bytes;                     // 0: nothing was loaded
    __device__ __forceinline__ void issue(const WireDev &wire, uint64_t p0, uint32_t np)
    {
        const uint64_t lo = wire.offsets[p0], last_at = wire.offsets[p0 + np - 1], last_len = wire.lens[p0 + np - 1];
        const bool fine = lo <= last_at && last_at <= wire.bytes && last_len <= wire.bytes - last_at &&
                          last_at + last_len - lo + 32 <= IMG;
        bytes = 0;
        if (!fine) return;              // (uniform)
        const uint32_t mis = (uint32_t)((uintptr_t)(wire.base + lo) & 15u);
        bytes = (uint32_t)(last_at + last_len - lo) + mis;
        if (bytes == 0) return;
        const uint8_t *g = wire.base + lo - mis;
        const uint32_t last = (bytes - 1u) & ~15u;
#pragma unroll
        for (int k = 0; k < PIECES; k++) {
            const uint32_t o = ((uint32_t)k * TILE_TPB + threadIdx.x) * 16u;
            x[k] = *reinterpret_cast<const u32x4 *>(g + (o < last ? o : last));
        }
    }
    __device__ __forceinline__ void commit(uint8_t *img) const
    {
#pragma unroll
        for (int k = 0; k < PIECES; k++) {
            const uint32_t o = ((uint32_t)k * TILE_TPB + threadIdx.x) * 16u;
            if (o < bytes) *reinterpret_cast<u32x4 *>(img + o) = x[k];
        }
    }
};

// Payload of variable field v between the image (or the arena) and the values column, packet by
// packet: 16 lanes take one packet -- 128-bit pieces while they fit, then the last few elements
// one per lane.  Neighbouring packets are neighbours in the column, so the global side of a
// wavefront is one contiguous run.  A tile that runs on global memory may hold giant packets:
// those are left to a second pass in which all 256 lanes share one packet.
#define TILE_BIG_PAYLOAD 4096u

template <uint32_t GROUP, bool STAGED, bool ENCODE, bool SWAP, int W>
__device__ __forceinline__ void tile_copy_run(uint8_t *w, uint8_t *d, uint64_t len, uint32_t lane)
{
    typedef typename UIntOf<W>::type U;
    const uint64_t whole = len & ~(uint64_t)15;
    for (uint64_t o = (uint64_t)lane * 16u; o < whole; o += GROUP * 16u) {
        if (ENCODE) st16_u(w + o, ld16_any<false>(d + o));
        else {
            const u32x4 x = ld16_u(w + o);
            st16_any<false>(d + o, SWAP ? swap_elems<W>(x) : x);
        }
    }
    for (uint64_t o = whole + (uint64_t)lane * W; o < len; o += (uint64_t)GROUP * W) {
        if (ENCODE) st_u<U>(w + o, ld_u<U>(d + o));
        else {
            U x = ld_u<U>(w + o);
            if (SWAP) x = bswap_u(x);
            st_u<U>(d + o, x);
        }
    }
}

// Encode into the image: the same packet-by-packet copy, split in two.  load() issues the global
// loads of a lane's four packets (two 128-bit pieces each, then one tail byte each); store()
// puts them into the image.  Nothing is reversed on encode (src/pdr_mem.c:22-38 and its twins
// store native order), so the copy is a copy of bytes whatever the element width, and one body
// serves all fields.  Between the two calls the lane can do other work: its wait for memory
// then covers both.  Positions are kept as offsets from the image and from the column, not as
// pointers: the accesses then compile to LDS and global instructions (pointers that pass
// through an array of generic pointers end up as flat accesses).
struct TileFill {
    static constexpr int PK = TILE_PKTS / (TILE_TPB / 16);      // packets per 16-lane group
    uint32_t wo[PK];
    uint64_t dof[PK];
    uint32_t len[PK];
    u32x4 r[PK][2];
    uint8_t tail[PK];
    bool more;

    // vpos_of(i) = where the field's length prefix starts in packet i
    template <typename L, typename VPOS>
    __device__ __forceinline__ void load(const L &m, int v, uint32_t width, uint32_t pre, const uint8_t *dense,
                                         uint32_t np, uint64_t lo, uint32_t mis, VPOS vpos_of)
    {
        const uint32_t g = threadIdx.x >> 4, lane = threadIdx.x & 15u;
        more = false;
#pragma unroll
        for (int k = 0; k < PK; k++) {
            const uint32_t i = g + (uint32_t)k * (TILE_TPB / 16);
            len[k] = 0; wo[k] = 0; dof[k] = 0;
            if (i < np) {
                const uint64_t s_lo = m.seg[v][i] * width;
                len[k] = (uint32_t)(m.seg[v][i + 1] * width - s_lo);      // a staged tile holds less than 24 KiB
                wo[k] = (uint32_t)(m.start[i] + vpos_of(i) + pre - lo) + mis;
                dof[k] = s_lo;
                more |= len[k] > 512u;
            }
        }
#pragma unroll
        for (int k = 0; k < PK; k++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t o = lane * 16u + (uint32_t)q * 256u;
                if (o + 16u <= len[k]) r[k][q] = ld16_any<false>(dense + dof[k] + o);
            }
#pragma unroll
        for (int k = 0; k < PK; k++) {
            const uint32_t o = (len[k] & ~15u) + lane;
            if (o < len[k]) tail[k] = dense[dof[k] + o];
        }
    }

    __device__ __forceinline__ void store(uint8_t *img, const uint8_t *dense) const
    {
        const uint32_t lane = threadIdx.x & 15u;
#pragma unroll
        for (int k = 0; k < PK; k++) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t o = lane * 16u + (uint32_t)q * 256u;
                if (o + 16u <= len[k]) st16_u(img + wo[k] + o, r[k][q]);
            }
            const uint32_t o = (len[k] & ~15u) + lane;
            if (o < len[k]) img[wo[k] + o] = tail[k];
        }
        if (!more) return;
        // payloads beyond 512 bytes: the remaining whole pieces, one at a time
#pragma unroll
        for (int k = 0; k < PK; k++)
            for (uint32_t o = 512u + lane * 16u; o + 16u <= len[k]; o += 256u)
                st16_u(img + wo[k] + o, ld16_any<false>(dense + dof[k] + o));
    }
};

// Decode out of the image: the mirror of TileFill in one piece.  The LDS reads of a lane's four
// packets are all issued before the first global store, so their latencies overlap.
template <bool SWAP, int W, typename L>
__device__ __forceinline__ void tile_drain_field(L &m, const uint8_t *img, int v, uint32_t pre, uint8_t *dense,
                                                 uint32_t np, uint64_t lo, uint32_t mis)
{
    typedef typename UIntOf<W>::type U;
    constexpr int PK = TILE_PKTS / (TILE_TPB / 16);      // packets per 16-lane group
    const uint32_t g = threadIdx.x >> 4, lane = threadIdx.x & 15u;
    uint32_t wo[PK];
    uint64_t dof[PK];
    uint32_t len[PK];
    bool more = false;
#pragma unroll
    for (int k = 0; k < PK; k++) {
        const uint32_t i = g + (uint32_t)k * (TILE_TPB / 16);
        len[k] = 0; wo[k] = 0; dof[k] = 0;
        if (i < np) {
            const uint64_t s_lo = m.seg[v][i] * W;
            len[k] = (uint32_t)(m.seg[v][i + 1] * W - s_lo);      // a staged tile holds less than 24 KiB
            wo[k] = (uint32_t)(m.start[i] + m.vpos[v][i] + pre - lo) + mis;
            dof[k] = s_lo;
            more |= len[k] > 512u;
        }
    }
    u32x4 r[PK][2];
#pragma unroll
    for (int k = 0; k < PK; k++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t o = lane * 16u + (uint32_t)q * 256u;
            if (o + 16u <= len[k]) r[k][q] = ld16_u(img + wo[k] + o);
        }
    U tail[PK];
#pragma unroll
    for (int k = 0; k < PK; k++) {
        const uint32_t o = (len[k] & ~15u) + lane * W;
        if (o < len[k]) tail[k] = ld_u<U>(img + wo[k] + o);
    }
#pragma unroll
    for (int k = 0; k < PK; k++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t o = lane * 16u + (uint32_t)q * 256u;
            if (o + 16u <= len[k]) st16_any<false>(dense + dof[k] + o, SWAP ? swap_elems<W>(r[k][q]) : r[k][q]);
        }
        const uint32_t o = (len[k] & ~15u) + lane * W;
        if (o < len[k]) st_u<U>(dense + dof[k] + o, SWAP ? bswap_u(tail[k]) : tail[k]);
    }
    if (!more) return;
    // payloads beyond 512 bytes: the remaining whole pieces, one at a time
#pragma unroll
    for (int k = 0; k < PK; k++)
        for (uint32_t o = 512u + lane * 16u; o + 16u <= len[k]; o += 256u) {
            const u32x4 x = ld16_u(img + wo[k] + o);
            st16_any<false>(dense + dof[k] + o, SWAP ? swap_elems<W>(x) : x);
        }
}

// Decode of a field that is SHORT in this tile (a dozen bytes per packet: names, small vectors): one
// wavefront, one lane per packet, instead of sixteen lanes per packet in all four.  Sixteen lanes
// on a 13-byte string is fifteen lanes of predicated-off instructions, and every wavefront of the
// workgroup issues them; here the other wavefronts go on to the next field meanwhile.  A lane moves
// 128-bit pieces while they fit (a few iterations at most for the odd long packet), then the rest
// as one masked 8 / 4 / 2 / 1-byte cascade out of a 16-byte LDS read (reading past the packet's
// end stays inside the image, which keeps 32 spare bytes).
#define TILE_SHORT_AVG 32u             // bytes per packet, averaged over the tile, up to which a field counts as short

template <bool SWAP, int W, typename L>
__device__ __forceinline__ void tile_drain_short(L &m, const uint8_t *img, int v, uint32_t pre, uint8_t *dense,
                                                 uint32_t np, uint64_t lo, uint32_t mis)
{
    const uint32_t i = threadIdx.x & 63u;
    if (i >= np) return;
    const uint64_t s_lo = m.seg[v][i] * W;
    const uint32_t len = (uint32_t)(m.seg[v][i + 1] * W - s_lo);
    const uint8_t *src = img + (uint32_t)(m.start[i] + m.vpos[v][i] + pre - lo) + mis;
    uint8_t *dst = dense + s_lo;
    uint32_t o = 0;
    for (; o + 16u <= len; o += 16u) {
        const u32x4 x = ld16_u(src + o);
        st16_any<false>(dst + o, SWAP ? swap_elems<W>(x) : x);
    }
    const uint32_t r = len - o;            // < 16, a multiple of W
    if (r == 0) return;
    u32x4 x = ld16_u(src + o);
    if (SWAP) x = swap_elems<W>(x);
    dst += o;
    uint32_t a = x[0], b = x[1];
    if (r & 8u) { st_u<uint64_t>(dst, (uint64_t)a | ((uint64_t)b << 32)); dst += 8; a = x[2]; b = x[3]; }
    if (r & 4u) { st_u<uint32_t>(dst, a); dst += 4; a = b; }
    if (W <= 2 && (r & 2u)) { st_u<uint16_t>(dst, (uint16_t)a); dst += 2; a >>= 16; }
    if (W == 1 && (r & 1u)) st_u<uint8_t>(dst, (uint8_t)a);
}

template <bool STAGED, bool ENCODE, bool SWAP, int W, typename L>
__device__ __forceinline__ void tile_copy_field(L &m, uint8_t *img, const WireDev &wire, int v, uint32_t pre,
                                                uint8_t *dense, uint32_t np, uint64_t lo, uint32_t mis)
{
    if (STAGED && !ENCODE) {        // (the staged encode is driven by tile_encode_body itself)
        const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)((m.seg[v][np] - m.seg[v][0]) * W));
        if (bytes <= np * TILE_SHORT_AVG) {
            // (wavefront v of the four: with several short fields they go side by side)
            if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == (v & (TILE_TPB / 64 - 1)))
                tile_drain_short<SWAP, W>(m, img, v, pre, dense, np, lo, mis);
            return;
        }
        tile_drain_field<SWAP, W>(m, img, v, pre, dense, np, lo, mis);
        return;
    }
    const uint32_t g = threadIdx.x >> 4, lane = threadIdx.x & 15u;
    for (uint32_t i = g; i < np; i += TILE_TPB / 16) {
        const uint64_t s_lo = m.seg[v][i] * W, len = m.seg[v][i + 1] * W - s_lo;
        if (len == 0 || (!STAGED && len >= TILE_BIG_PAYLOAD)) continue;
        tile_copy_run<16u, STAGED, ENCODE, SWAP, W>(
            tile_at<STAGED>(img, wire.base, lo, mis, m.start[i] + m.vpos[v][i] + pre), dense + s_lo, len, lane);
    }
    if (STAGED) return;
    for (uint32_t i = 0; i < np; i++) {
        const uint64_t s_lo = m.seg[v][i] * W, len = m.seg[v][i + 1] * W - s_lo;
        if (len < TILE_BIG_PAYLOAD) continue;
        tile_copy_run<TILE_TPB, STAGED, ENCODE, SWAP, W>(
            tile_at<STAGED>(img, wire.base, lo, mis, m.start[i] + m.vpos[v][i] + pre), dense + s_lo, len, threadIdx.x);
    }
}

template <bool STAGED, bool ENCODE, bool SWAP, typename L>
__device__ __forceinline__ void tile_copy_fields(L &m, uint8_t *img, const FmtDev &fmt, const TileVars &tv,
                                                 const WireDev &wire, const ColsDev &cols, uint32_t np)
{
    const uint64_t lo = m.start[0];
    const uint32_t mis = (uint32_t)((uintptr_t)(wire.base + lo) & 15u);
    for (int v = 0; v < tv.nv; v++) {
        const int f = tv.field[v];
        const uint32_t pre = prefix_bytes(fmt.f[f].kind);
        uint8_t *dense = reinterpret_cast<uint8_t *>(cols.values[f]);
        // raw bytes and strings are never reversed (src/pdr_mem.c:394-412)
        switch (fmt.f[f].width) {
        case 1: tile_copy_field<STAGED, ENCODE, false, 1>(m, img, wire, v, pre, dense, np, lo, mis); break;
        case 2: tile_copy_field<STAGED, ENCODE, SWAP, 2>(m, img, wire, v, pre, dense, np, lo, mis); break;
        case 4: tile_copy_field<STAGED, ENCODE, SWAP, 4>(m, img, wire, v, pre, dense, np, lo, mis); break;
        default: tile_copy_field<STAGED, ENCODE, SWAP, 8>(m, img, wire, v, pre, dense, np, lo, mis); break;
        }
    }
}

// ---------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------
template <bool STAGED, typename L>
__device__ __forceinline__ void tile_encode_body(L &m, uint8_t *img, const FmtDev &fmt, const TileVars &tv,
                                                 const WireDev &wire, const ColsDev &cols, uint64_t p0, uint32_t np)
{
    const uint32_t t = threadIdx.x;
    const uint64_t lo = m.start[0];
    const uint32_t mis = (uint32_t)((uintptr_t)(wire.base + lo) & 15u);
    // The payload of the first variable field starts at a fixed place in every packet (only
    // scalars lie in front of it), so its loads need nothing but the tile's offs[]: they are
    // issued first and are on their way while the scalars are handled.
    TILE_STAMP(1);
    TileFill fill;
    const uint8_t *dense0 = nullptr;
    if (STAGED && tv.nv > 0) {
        const int f = tv.field[0];
        const uint64_t at0 = fmt.f[f].fixed_before;
        dense0 = reinterpret_cast<const uint8_t *>(cols.values[f]);
        fill.load(m, 0, (uint32_t)tv.width[0], prefix_bytes(fmt.f[f].kind), dense0, np, lo, mis,
                  [at0](uint32_t) { return at0; });
    }
    // One wavefront per field, one lane per packet: what depends on the field alone (its
    // kind, width, column) is the same for all lanes, so it is fetched by scalar loads and the
    // branches on it do not diverge; the column reads of a wavefront are one contiguous run.
    const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint32_t i = t & 63u;
    // where every variable field starts in its packet
    for (int v = wv; v < tv.nv; v += TILE_TPB / 64) {
        const int f = tv.field[v];
        if (i < np) m.vpos[v][i] = fmt.f[f].fixed_before + tile_var_before(m, tv, fmt, f, i);
    }
    // scalars and length prefixes
    for (int f = wv; f < fmt.nf; f += TILE_TPB / 64) {
        const FieldDev fd = fmt.f[f];
        if (i >= np) continue;
        uint8_t *dst = tile_at<STAGED>(img, wire.base, lo, mis, m.start[i] + fd.fixed_before + tile_var_before(m, tv, fmt, f, i));
        if (fd.kind == MRPDR_SCALAR) {
            const uint8_t *src = reinterpret_cast<const uint8_t *>(cols.values[f]) + (p0 + i) * fd.width;
            st_w(dst, (int)fd.width, ld_wire_w(src, (int)fd.width, false));
        } else {
            int v = 0;
            while (tv.field[v] != f) v++;
            // pdr_uint64(count), src/pdr.c:425; strlen+1, :295-297
            st_w(dst, fd.kind == MRPDR_ARRAY ? 8 : 4, m.seg[v][i + 1] - m.seg[v][i]);
        }
    }
    if (!STAGED) {
        __syncthreads();
        tile_copy_fields<false, true, false>(m, img, fmt, tv, wire, cols, np);
        return;
    }
    TILE_STAMP(2);
    if (tv.nv > 0) fill.store(img, dense0);
    TILE_STAMP(3);
    __syncthreads();                     // the positions of the other variable fields are in LDS
    TILE_STAMP(4);
    for (int v = 1; v < tv.nv; v++) {
        const int f = tv.field[v];
        const uint8_t *dense = reinterpret_cast<const uint8_t *>(cols.values[f]);
        fill.load(m, v, (uint32_t)tv.width[v], prefix_bytes(fmt.f[f].kind), dense, np, lo, mis,
                  [&m, v](uint32_t k) { return m.vpos[v][k]; });
        fill.store(img, dense);
    }
    TILE_STAMP(5);
    __syncthreads();
    TILE_STAMP(6);
    // image -> arena, packet by packet (16 lanes each): the bytes in front of the first 16-byte
    // boundary and behind the last one go out one per lane, the pieces between them as aligned
    // 128-bit stores.  Only bytes of packets are written: what lies between packets is not ours
    // (padding of the caller's layout, or the headers of a message).
    if (m.packed) {
        // the tile is one run of bytes: out as a block, aligned 128-bit stores by all lanes
        const uint64_t hi = m.end[np - 1];
        const uint64_t a0 = lo + ((16u - mis) & 15u);
        const uint64_t head_end = a0 < hi ? a0 : hi;
        if (lo + t < head_end) wire.base[lo + t] = img[mis + t];
        if (a0 < hi) {
            const uint64_t a1 = a0 + ((hi - a0) & ~(uint64_t)15);
            for (uint64_t x = a0 + (uint64_t)t * 16u; x < a1; x += TILE_TPB * 16u)
                *reinterpret_cast<u32x4 *>(wire.base + x) = *reinterpret_cast<const u32x4 *>(img + (uint32_t)(x - lo) + mis);
            if (a1 + t < hi) wire.base[a1 + t] = img[(uint32_t)(a1 - lo) + mis + t];
        }
        TILE_STAMP(7);
        return;
    }
    // Positions relative to the tile's first byte (32 bits: a staged tile is shorter than the
    // image); image byte of relative position r is img[r + mis], and r + mis has the alignment
    // of the arena address.  The LDS reads of a lane's four packets are issued before the first
    // global store.
    constexpr int PK = TILE_PKTS / (TILE_TPB / 16);
    const uint32_t grp = t >> 4, lane = t & 15u;
    uint8_t *out = wire.base + lo;
    const uint8_t *src = img + mis;
    uint32_t rs[PK], re[PK], ra0[PK], ra1[PK];
    bool more = false;
#pragma unroll
    for (int k = 0; k < PK; k++) {
        const uint32_t i = grp + (uint32_t)k * (TILE_TPB / 16);
        rs[k] = re[k] = 0;
        if (i < np) { rs[k] = (uint32_t)(m.start[i] - lo); re[k] = (uint32_t)(m.end[i] - lo); }
        ra0[k] = rs[k] + ((0u - (rs[k] + mis)) & 15u);          // first 16-byte boundary at or behind the start
        if (ra0[k] > re[k]) ra0[k] = re[k];
        ra1[k] = ra0[k] + ((re[k] - ra0[k]) & ~15u);
        more |= ra1[k] - ra0[k] > 512u;
    }
    uint8_t hb[PK], tb[PK];
    u32x4 r[PK][2];
#pragma unroll
    for (int k = 0; k < PK; k++) {
        if (rs[k] + lane < ra0[k]) hb[k] = src[rs[k] + lane];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t x = ra0[k] + lane * 16u + (uint32_t)q * 256u;
            if (x < ra1[k]) r[k][q] = *reinterpret_cast<const u32x4 *>(src + x);
        }
        if (ra1[k] + lane < re[k]) tb[k] = src[ra1[k] + lane];
    }
#pragma unroll
    for (int k = 0; k < PK; k++) {
        if (rs[k] + lane < ra0[k]) out[rs[k] + lane] = hb[k];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t x = ra0[k] + lane * 16u + (uint32_t)q * 256u;
            if (x < ra1[k]) *reinterpret_cast<u32x4 *>(out + x) = r[k][q];
        }
        if (ra1[k] + lane < re[k]) out[ra1[k] + lane] = tb[k];
    }
    TILE_STAMP(7);
    if (!more) return;
#pragma unroll
    for (int k = 0; k < PK; k++)
        for (uint32_t x = ra0[k] + 512u + lane * 16u; x < ra1[k]; x += 256u)
            *reinterpret_cast<u32x4 *>(out + x) = *reinterpret_cast<const u32x4 *>(src + x);
}

template <int V, int IMG>
__global__ __launch_bounds__(TILE_TPB) void k_tile_encode(FmtDev fmt, TileVars tv, WireDev wire, ColsDev cols, uint64_t n)
{
    __shared__ TileLdsT<V> m;
    __shared__ __attribute__((aligned(16))) uint8_t img[IMG];
    const uint64_t p0 = (uint64_t)blockIdx.x * TILE_PKTS;
    const uint32_t np = (uint32_t)((n - p0) < TILE_PKTS ? (n - p0) : TILE_PKTS);
    TILE_STAMP(0);
    tile_load_meta<true, false, IMG>(m, tv, wire, p0, np, n);
    if (m.staged) tile_encode_body<true>(m, img, fmt, tv, wire, cols, p0, np);
    else          tile_encode_body<false>(m, img, fmt, tv, wire, cols, p0, np);
}

// ---------------------------------------------------------------------------
// decode, begin: walk
// ---------------------------------------------------------------------------
template <bool SWAP, int V, int IMG>
__global__ __launch_bounds__(TILE_TPB) void k_tile_walk(FmtDev fmt, TileVars tv, WireDev wire, ColsDev cols,
                                                         uint64_t n, int32_t *ok, uint64_t *tile_sums,
                                                         uint64_t ntiles)
{
    __shared__ TileLdsT<V> m;
    __shared__ __attribute__((aligned(16))) uint8_t img[IMG];
    const uint64_t p0 = (uint64_t)blockIdx.x * TILE_PKTS;
    const uint32_t np = (uint32_t)((n - p0) < TILE_PKTS ? (n - p0) : TILE_PKTS);
    const uint32_t t = threadIdx.x;
    tile_load_meta<false, false, IMG>(m, tv, wire, p0, np, n);
    const bool staged = m.staged != 0;
    const uint64_t lo = m.start[0];
    const uint32_t mis = (uint32_t)((uintptr_t)(wire.base + lo) & 15u);
    if (staged) {
        tile_stage_in<IMG>(img, wire.base, lo, m.end[np - 1], mis);
        __syncthreads();
    }
    if (t >= 64) return;                        // one wavefront walks: a lane per packet
    bool good = false;
    if (t < np) {
        const uint64_t at = m.start[t], space = m.end[t] - at;
        if (staged) {
            good = walk_decode_packet(SWAP, fmt, cols, p0 + t, tile_at<true>(img, wire.base, lo, mis, at), space, true,
                                            nullptr, &m.seg[0][t], TILE_PKTS + 1);
        } else {
            // same checks as the generic walker: the packet must lie inside the arena
            const uint64_t len = wire.lens[p0 + t];
            good = at <= wire.bytes && len <= wire.bytes - at;
            good = walk_decode_packet(SWAP, fmt, cols, p0 + t, wire.base + at, len, good, nullptr,
                                            &m.seg[0][t], TILE_PKTS + 1);
        }
        if (ok) ok[p0 + t] = good ? 1 : 0;
    }
    // element total of every variable field over the tile (the input of the offsets scan)
    for (int v = 0; v < tv.nv; v++) {
        uint64_t c = t < np ? m.seg[v][t] : 0;          // left there by the walk of this lane
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor((unsigned long long)c, d, 64);
        if (t == 0) tile_sums[(uint64_t)v * ntiles + blockIdx.x] = c;
    }
}

// ---------------------------------------------------------------------------
// decode in one call: the counts without the wire image
// ---------------------------------------------------------------------------
// When one call decodes the whole wave, nobody needs the scalar columns before the payloads move:
// the payload kernel has every packet in LDS anyway and writes the scalars from there.  What has to
// be known beforehand is only what offs[] is made of -- the length prefixes.  So the first pass does
// not stage anything: one lane per packet reads the packet's prefixes straight from the arena
// (one dependent load per variable field; the bounds of everything in between are arithmetic on
// lengths, the same tests as walk_decode_packet) and leaves counts[], ok[], how many leading
// fields of the packet are good, and the per-tile sums.  It touches the cache lines that hold a
// prefix instead of the whole wire, and a CU runs 2,048 such lanes at once instead of five
// wavefronts walking LDS images.
#define COUNT_WALK_TPB 256                  // packets per workgroup, and per partial sum

template <bool SWAP>
__global__ __launch_bounds__(COUNT_WALK_TPB) void k_count_walk(FmtDev fmt, WireDev wire, ColsDev cols,
                                                               uint64_t n, int32_t *ok, uint8_t *nvalid,
                                                               uint64_t *block_sums, uint64_t stride)
{
    __shared__ uint64_t s_part[TILE_MAXV][COUNT_WALK_TPB / 64];
    const uint64_t i = (uint64_t)blockIdx.x * COUNT_WALK_TPB + threadIdx.x;
    const bool live = i < n;
    uint64_t at = 0, space = 0;
    if (live) { at = wire.offsets[i]; space = wire.lens[i]; }
    bool good = live && at <= wire.bytes && space <= wire.bytes - at;
    const uint8_t *p = wire.base + at;
    uint32_t valid = 0;
    int v = 0;
    // (no barrier inside the walk: a wavefront's dependent loads wait for nobody else's)
    for (int f = 0; f < fmt.nf; f++) {
        const FieldDev fd = fmt.f[f];
        const uint32_t w = fd.width;
        if (fd.kind == MRPDR_SCALAR) {
            if (good && w <= space) { p += w; space -= w; valid++; }
            else good = false;
            continue;
        }
        uint64_t cnt = 0;
        if (fd.kind == MRPDR_ARRAY) {
            if (good && 8 <= space) {
                cnt = ld_wire<8>(p, SWAP); p += 8; space -= 8;
                const uint64_t bound = fd.large ? UINT64_MAX : (uint64_t)INT32_MAX;
                if (cnt > bound || cnt > (space >> __builtin_ctz(w))) { good = false; cnt = 0; }
                else valid++;
                p += cnt * w; space -= cnt * w;
            } else good = false;
        } else {                                          // MRPDR_STRING: u32 strlen+1, then the bytes
            if (good && 4 <= space) {
                cnt = ld_wire<4>(p, SWAP); p += 4; space -= 4;
                if (cnt == 0 || cnt > space) { good = false; cnt = 0; }
                else valid++;
                p += cnt; space -= cnt;
            } else good = false;
        }
        if (live) cols.counts[f][i] = cnt;
        // the wavefront's total of this field
        uint64_t c = cnt;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor((unsigned long long)c, d, 64);
        if ((threadIdx.x & 63u) == 0) s_part[v][threadIdx.x >> 6] = c;
        v++;
    }
    if (live) {
        if (ok) ok[i] = good ? 1 : 0;
        nvalid[i] = (uint8_t)valid;
    }
    __syncthreads();
    if ((int)threadIdx.x < v) {
        uint64_t total = 0;
        for (int k = 0; k < COUNT_WALK_TPB / 64; k++) total += s_part[threadIdx.x][k];
        block_sums[(uint64_t)threadIdx.x * stride + blockIdx.x] = total;
    }
}

// ---------------------------------------------------------------------------
// decode, begin: offsets of every variable field, one launch
// ---------------------------------------------------------------------------
// Workgroup (c, v) owns TILE_OFFS_PKTS packets of variable field v.  Its base is the sum of the
// partial sums in front of it (read in parallel), then a workgroup scan of its own counts.  The
// partial sums cover pkts_per_sum packets each: a tile (k_tile_walk) or a workgroup of
// k_count_walk; workgroup c adds up the sums in front of its first packet (4,096 values at most
// on a wave of 2^20 packets, sixteen per lane).
__global__ __launch_bounds__(MRPDR_TPB) void k_tile_offs(TileVars tv, ColsDev cols, uint64_t n,
                                                         const uint64_t *part_sums, uint64_t stride, uint32_t pkts_per_sum)
{
    const int v = blockIdx.y;
    const int f = tv.field[v];
    const uint64_t first = (uint64_t)blockIdx.x * TILE_OFFS_PKTS;
    const uint64_t sums_before = first / pkts_per_sum;
    const uint64_t *sums = part_sums + (uint64_t)v * stride;
    uint64_t s = 0;
    for (uint64_t j = threadIdx.x; j < sums_before; j += MRPDR_TPB) s += sums[j];
    uint64_t base;
    block_exclusive_scan(s, &base);
    constexpr int IPT = TILE_OFFS_PKTS / MRPDR_TPB;
    uint64_t c[IPT];
    uint64_t mine = 0;
    const uint64_t at = first + (uint64_t)threadIdx.x * IPT;
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        c[k] = at + k < n ? cols.counts[f][at + k] : 0;
        mine += c[k];
    }
    uint64_t total;
    uint64_t run = base + block_exclusive_scan(mine, &total);
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        if (at + k < n) cols.offs[f][at + k] = run;
        run += c[k];
        if (at + k + 1 == n) cols.offs[f][n] = run;
    }
}

// ---------------------------------------------------------------------------
// decode, finish: payloads
// ---------------------------------------------------------------------------
// nvalid (optional): the one-call decode -- the scalar columns are written here too, from the
// image, with k_count_walk's verdict on every packet (field f of packet i is good iff f < nvalid[i];
// the fields behind a failure read as zero).
template <bool SWAP, int V, int IMG>
__global__ __launch_bounds__(TILE_TPB) void k_tile_decode(FmtDev fmt, TileVars tv, WireDev wire, ColsDev cols,
                                                           uint64_t n, uint32_t *status, const uint8_t *nvalid)
{
    __shared__ TileLdsT<V> m;
    __shared__ __attribute__((aligned(16))) uint8_t img[IMG];
    const uint64_t p0 = (uint64_t)blockIdx.x * TILE_PKTS;
    const uint32_t np = (uint32_t)((n - p0) < TILE_PKTS ? (n - p0) : TILE_PKTS);
    const uint32_t t = threadIdx.x;
    // (lane i of every wavefront: how many leading fields of packet i are good -- loaded with the
    // boundaries, not when the scalars need it)
    TILE_STAMP(0);
    const uint32_t good_fields = (nvalid && (t & 63u) < np) ? nvalid[p0 + (t & 63u)] : 0u;
    // a column too small for the wave, or offs[] that do not describe these packets: nothing is
    // written, the status word says so
    // the image's loads go out before the boundaries are checked, and arrive while they are
    TileStage<IMG> early;
    tile_load_meta<true, true, IMG>(m, tv, wire, p0, np, n, [&]() { early.issue(wire, p0, np); });
    TILE_STAMP(1);
    const bool skip = m.too_small != 0;          // no payload moves (the scalar columns are still written)
    if (skip && t == 0) atomicOr(status, 1u);
    if (skip && !nvalid) return;
    const bool staged = m.staged != 0;
    if (staged) early.commit(img);               // (a staged tile passed the same tests on the same two packets)
    TILE_STAMP(2);
    // where every variable field starts in its packet, from the counts (a packet that failed
    // has zero counts from the failed field on, so nothing of it is moved)
    {
        const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));      // a wavefront per field
        const uint32_t i = t & 63u;
        for (int v = wv; v < tv.nv; v += TILE_TPB / 64) {
            const int f = tv.field[v];
            if (i < np) m.vpos[v][i] = fmt.f[f].fixed_before + tile_var_before(m, tv, fmt, f, i);
        }
    }
    TILE_STAMP(3);
    __syncthreads();
    TILE_STAMP(4);
    if (nvalid) {
        // scalars: one wavefront per field, one lane per packet (the field's width and column
        // come from scalar loads, the column stores of a wavefront are one contiguous run)
        const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
        const uint32_t i = t & 63u;
        const uint64_t lo = m.start[0];
        const uint32_t mis = (uint32_t)((uintptr_t)(wire.base + lo) & 15u);
        for (int f = wv; f < fmt.nf; f += TILE_TPB / 64) {
            const FieldDev fd = fmt.f[f];
            if (fd.kind != MRPDR_SCALAR || i >= np) continue;
            uint64_t x = 0;
            const uint64_t pos = fd.fixed_before + tile_var_before(m, tv, fmt, f, i);
            // (the place is checked against the packet once more: offs[] may be anything when the
            // payloads are being skipped)
            if ((uint32_t)f < good_fields && m.end[i] >= m.start[i] && pos + fd.width <= m.end[i] - m.start[i]) {
                const uint64_t at = m.start[i] + pos;
                // (two branches, not one pointer chosen between LDS and the arena: that would be a flat access)
                if (staged) x = ld_wire_w(img + (uint32_t)(at - lo) + mis, (int)fd.width, SWAP);
                else        x = ld_wire_w(wire.base + at, (int)fd.width, SWAP);
            }
            st_w(reinterpret_cast<uint8_t *>(cols.values[f]) + (p0 + i) * fd.width, (int)fd.width, x);
        }
    }
    TILE_STAMP(5);
    if (skip) return;
    if (staged) tile_copy_fields<true, false, SWAP>(m, img, fmt, tv, wire, cols, np);
    else        tile_copy_fields<false, false, SWAP>(m, img, fmt, tv, wire, cols, np);
    TILE_STAMP(6);
}


