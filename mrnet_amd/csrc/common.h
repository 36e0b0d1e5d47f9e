// Shared host/device definitions of the wave codec.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrpdr.h"

#define MRPDR_MAXF 32          // tokens per format string
#define MRPDR_TPB 256          // threads per workgroup: 4 wavefronts of 64

// One format token as the kernels see it.
struct FieldDev {
    uint8_t  kind;          // mrpdr_kind
    uint8_t  width;         // element width in bytes (strings: 1)
    uint8_t  large;         // %A* token: count bound UINT64_MAX instead of INT32_MAX
    uint8_t  type;          // DataType code
    uint32_t fixed_before;  // bytes of scalars and length prefixes that precede this field
};

struct FmtDev {
    int32_t  nf;
    uint32_t fixed_bytes;   // all scalars + all length prefixes
    FieldDev f[MRPDR_MAXF];
};

struct ColsDev {
    void     *values[MRPDR_MAXF];
    uint64_t *counts[MRPDR_MAXF];
    uint64_t *offs[MRPDR_MAXF];
    uint64_t  cap[MRPDR_MAXF];
    uint64_t *sub_counts[MRPDR_MAXF];   // string arrays: strlen+1 of every string
    uint64_t *sub_offs[MRPDR_MAXF];     // string arrays: byte range of every string in values
    uint64_t  sub_cap[MRPDR_MAXF];
};

struct WireDev {
    uint8_t        *base;
    const uint64_t *offsets;
    const uint64_t *lens;
    uint64_t        stride;
    uint64_t        bytes;
    const uint8_t  *orders;     // optional: sender byte order of every packet (a wave collected from
                                // children of both kinds); NULL = one order for the wave
};

// Whether packet i is byte-reversed on decode: its own order differs from the device's
// (little-endian, 0) when the wave carries one order per packet, else the wave's setting.
// pdrmem_create makes the same choice per packet (src/pdr_mem.c:433-449, src/Packet.C:96,118).
__device__ inline bool pkt_swap(const WireDev &w, uint64_t i, bool uniform)
{
    return w.orders ? w.orders[i] != 0 : uniform;
}

// Length prefix of a variable field: u64 element count for arrays (src/pdr.c:425),
// u32 strlen+1 for strings (src/pdr.c:295-297).
__host__ __device__ inline uint32_t prefix_bytes(uint8_t kind)
{
    return kind == MRPDR_SCALAR ? 0u : (kind == MRPDR_STRING ? 4u : 8u);
}

// Bytes that variable field g of packet s puts on the wire behind its length prefix.  A string
// array is count x [u32 strlen+1 | bytes, NUL] (pdr_array over pdr_wrapstring,
// src/Packet.C:794-813).
__device__ inline uint64_t var_bytes(const FmtDev &fmt, const ColsDev &cols, int g, uint64_t s)
{
    const uint64_t cnt = cols.counts[g][s];
    if (fmt.f[g].kind != MRPDR_STRING_ARRAY) return cnt * (uint64_t)fmt.f[g].width;
    const uint64_t first = cols.offs[g][s];
    return 4 * cnt + (cols.sub_offs[g][first + cnt] - cols.sub_offs[g][first]);
}

// pdr_sizeof of packet i: the fixed part plus the payloads of its variable fields
// (src/pdr_sizeof.c:120-140 over Packet::pdr_packet_data).  A '%s' field without a single byte
// -- not even the NUL -- stands for the NULL pointer, which pdr_string refuses to encode
// (src/pdr.c:279-282), and so are counts beyond the token's bound: such a packet has no valid
// encoding and sets *bad.
__device__ inline uint64_t packet_bytes(const FmtDev &fmt, const ColsDev &cols, uint64_t i, bool *bad)
{
    uint64_t len = fmt.fixed_bytes;
    for (int f = 0; f < fmt.nf; f++) {
        if (fmt.f[f].kind == MRPDR_SCALAR) continue;
        len += var_bytes(fmt, cols, f, i);
        const uint64_t cnt = cols.counts[f][i];
        if (fmt.f[f].kind == MRPDR_STRING) {
            if (cnt == 0 || cnt > 0xFFFFFFFEull) *bad = true;      // NULL, or beyond pdr_wrapstring's bound (src/pdr.c:291,332)
        } else if (!fmt.f[f].large && cnt > (uint64_t)INT32_MAX) {
            *bad = true;                                           // '%a' counts stop at INT32_MAX (src/pdr.c:429, src/Packet.C:647-810)
        }
    }
    return len;
}

// The same for K consecutive packets first .. first+K-1 of a lane (those below n), field by
// field: the loads of the K packets are independent and unconditional (a packet past the end
// re-reads packet n-1), so they are issued together; what depends on the field alone is
// fetched once.  Packet by packet, every load was followed by its own wait.
template <int K>
__device__ inline void packets_bytes(const FmtDev &fmt, const ColsDev &cols, uint64_t first, uint64_t n,
                                     uint64_t (&len)[K], bool *bad)
{
#pragma unroll
    for (int k = 0; k < K; k++) len[k] = fmt.fixed_bytes;
    for (int f = 0; f < fmt.nf; f++) {
        const FieldDev fd = fmt.f[f];
        if (fd.kind == MRPDR_SCALAR) continue;
        uint64_t cnt[K];
#pragma unroll
        for (int k = 0; k < K; k++) cnt[k] = cols.counts[f][first + k < n ? first + k : n - 1];
        if (fd.kind == MRPDR_STRING_ARRAY) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (first + k >= n) continue;
                const uint64_t at = cols.offs[f][first + k];
                len[k] += 4 * cnt[k] + (cols.sub_offs[f][at + cnt[k]] - cols.sub_offs[f][at]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++)
                if (first + k < n) len[k] += cnt[k] * (uint64_t)fd.width;
        }
        bool wrong = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (first + k >= n) continue;
            if (fd.kind == MRPDR_STRING) wrong |= cnt[k] == 0 || cnt[k] > 0xFFFFFFFEull;      // as in packet_bytes
            else if (!fd.large) wrong |= cnt[k] > (uint64_t)INT32_MAX;
        }
        if (wrong) *bad = true;
    }
}

__device__ inline uint64_t pkt_off(const WireDev &w, uint64_t i)
{
    return w.offsets ? w.offsets[i] : i * w.stride;
}

__device__ inline uint64_t pkt_len(const WireDev &w, const FmtDev &fmt, uint64_t i)
{
    return w.lens ? w.lens[i] : (uint64_t)fmt.fixed_bytes;
}

// ---- byte-granular loads/stores at any alignment --------------------------------
// gfx950 under the HSA ABI runs with unaligned global access enabled, so these
// compile to single global_load/store instructions.
template <typename U> __device__ inline U ld_u(const uint8_t *p)
{
    U v;
    __builtin_memcpy(&v, p, sizeof(U));
    return v;
}
template <typename U> __device__ inline void st_u(uint8_t *p, U v)
{
    __builtin_memcpy(p, &v, sizeof(U));
}

__device__ inline uint16_t bswap_u(uint16_t v) { return __builtin_bswap16(v); }
__device__ inline uint32_t bswap_u(uint32_t v) { return __builtin_bswap32(v); }
__device__ inline uint64_t bswap_u(uint64_t v) { return __builtin_bswap64(v); }
__device__ inline uint8_t  bswap_u(uint8_t v)  { return v; }

// Load a W-byte scalar from the wire as an unsigned integer, reversing its bytes
// when the sender's order differs (pdrmem_get*_swap, src/pdr_mem.c:96-315).
template <int W> struct UIntOf;
template <> struct UIntOf<1> { typedef uint8_t  type; };
template <> struct UIntOf<2> { typedef uint16_t type; };
template <> struct UIntOf<4> { typedef uint32_t type; };
template <> struct UIntOf<8> { typedef uint64_t type; };

template <int W> __device__ inline uint64_t ld_wire(const uint8_t *p, bool swap)
{
    typedef typename UIntOf<W>::type U;
    U v = ld_u<U>(p);
    if (swap) v = bswap_u(v);
    return (uint64_t)v;
}

__device__ inline uint64_t ld_wire_w(const uint8_t *p, int w, bool swap)
{
    switch (w) {
    case 1: return ld_wire<1>(p, swap);
    case 2: return ld_wire<2>(p, swap);
    case 4: return ld_wire<4>(p, swap);
    default: return ld_wire<8>(p, swap);
    }
}

__device__ inline void st_w(uint8_t *p, int w, uint64_t v)
{
    switch (w) {
    case 1: st_u<uint8_t>(p, (uint8_t)v); break;
    case 2: st_u<uint16_t>(p, (uint16_t)v); break;
    case 4: st_u<uint32_t>(p, (uint32_t)v); break;
    default: st_u<uint64_t>(p, v); break;
    }
}
