// String arrays (%as / %As): pdr_array over pdr_wrapstring (src/Packet.C:794-813,
// src/pdr.c:263-333, 415-472).  On the wire: u64 count | count x [u32 strlen+1 | bytes, NUL].
//
// Decoded form, two levels: counts[n] / offs[n+1] number the strings of every packet; string j
// of the wave owns values[sub_offs[j] .. sub_offs[j+1]) and sub_counts[j] = strlen+1.
//
// The lengths of the strings of one packet sit one behind the other on the wire, so finding them
// is serial per packet (one lane walks), while the bytes are moved by the 64 lanes of a wavefront
// per packet.  MRNet uses these tokens in control messages (src/Network.C:1954-1955), not in
// data waves: the kernels are simple rather than bandwidth-tuned.
#pragma once

#include "common.h"

// Wire address of field f's length prefix in packet i.  Only called for packets that decoded
// through field f, so no bounds checks are repeated here.
__device__ inline const uint8_t *wire_cursor(const bool swap, const FmtDev &fmt, const WireDev &wire, uint64_t i, int f)
{
    const uint8_t *p = wire.base + pkt_off(wire, i);
    for (int g = 0; g < f; g++) {
        const FieldDev fd = fmt.f[g];
        if (fd.kind == MRPDR_SCALAR) p += fd.width;
        else if (fd.kind == MRPDR_ARRAY) p += 8 + ld_wire<8>(p, swap) * fd.width;
        else if (fd.kind == MRPDR_STRING) p += 4 + ld_wire<4>(p, swap);
        else {
            const uint64_t c = ld_wire<8>(p, swap);
            p += 8;
            for (uint64_t k = 0; k < c; k++) p += 4 + ld_wire<4>(p, swap);
        }
    }
    return p;
}

// decode, step 1: one lane per packet writes sub_counts[] of its strings and the packet's
// string bytes (to be scanned into the byte position of the packet's first string).
template <bool SWAP>
__global__ __launch_bounds__(MRPDR_TPB) void k_strarr_sizes(FmtDev fmt, WireDev wire, ColsDev cols, uint64_t n,
                                                            int f, uint64_t *pkt_bytes, uint32_t *status)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i >= n) return;
    uint64_t bytes = 0;
    if (cols.offs[f][n] > cols.sub_cap[f]) {
        if (i == 0) atomicOr(status, 1u);
    } else if (cols.counts[f][i]) {
        const uint64_t cnt = cols.counts[f][i], first = cols.offs[f][i];
        const bool sw = pkt_swap(wire, i, SWAP);
        const uint8_t *p = wire_cursor(sw, fmt, wire, i, f) + 8;
        for (uint64_t k = 0; k < cnt; k++) {
            const uint64_t l = ld_wire<4>(p, sw);
            cols.sub_counts[f][first + k] = l;
            p += 4 + l;
            bytes += l;
        }
    }
    pkt_bytes[i] = bytes;
}

// decode, step 2: one wavefront per packet moves the bytes of its strings (never reversed,
// src/pdr_mem.c:394-412) and writes sub_offs[].
template <bool SWAP>
__global__ __launch_bounds__(MRPDR_TPB) void k_strarr_decode(FmtDev fmt, WireDev wire, ColsDev cols, uint64_t n,
                                                             int f, const uint64_t *pkt_boffs, uint32_t *status)
{
    const uint64_t i = ((uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (cols.offs[f][n] > cols.sub_cap[f] || pkt_boffs[n] > cols.cap[f]) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, 1u);
        return;
    }
    if (i >= n) return;
    const uint64_t cnt = cols.counts[f][i], first = cols.offs[f][i];
    // also the next packet's first entry (same value), and sub_offs[total] from the last packet
    if (lane == 0) cols.sub_offs[f][first + cnt] = pkt_boffs[i + 1];
    if (!cnt) return;
    const bool sw = pkt_swap(wire, i, SWAP);
    const uint8_t *p = wire_cursor(sw, fmt, wire, i, f) + 8;
    uint8_t *values = reinterpret_cast<uint8_t *>(cols.values[f]);
    uint64_t run = pkt_boffs[i];
    for (uint64_t k = 0; k < cnt; k++) {
        const uint64_t l = ld_wire<4>(p, sw);
        if (lane == 0) cols.sub_offs[f][first + k] = run;
        for (uint64_t b = lane; b < l; b += 64) values[run + b] = p[4 + b];
        p += 4 + l;
        run += l;
    }
}

// encode: one wavefront per packet.  The count was written by the walker; string k of the packet
// starts 4(k+1) + (bytes of the strings in front of it) behind the count.
__global__ __launch_bounds__(MRPDR_TPB) void k_strarr_encode(FmtDev fmt, WireDev wire, ColsDev cols, uint64_t n, int f)
{
    const uint64_t i = ((uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (i >= n) return;
    const uint64_t cnt = cols.counts[f][i];
    if (!cnt) return;
    const uint64_t *so = cols.sub_offs[f] + cols.offs[f][i];
    uint64_t pos = (uint64_t)fmt.f[f].fixed_before + 8;
    for (int g = 0; g < f; g++)
        if (fmt.f[g].kind != MRPDR_SCALAR) pos += var_bytes(fmt, cols, g, i);
    uint8_t *dst = wire.base + pkt_off(wire, i) + pos;
    const uint8_t *values = reinterpret_cast<const uint8_t *>(cols.values[f]);
    const uint64_t b0 = so[0], nbytes = so[cnt] - b0;
    for (uint64_t k = lane; k < cnt; k += 64)         // pdr_uint32(strlen+1), src/pdr.c:295-297
        st_u<uint32_t>(dst + 4 * k + (so[k] - b0), (uint32_t)(so[k + 1] - so[k]));
    for (uint64_t b = lane; b < nbytes; b += 64) {
        uint64_t lo = 0, hi = cnt - 1;                // string that holds byte b: so[k+1] > b0 + b
        while (lo < hi) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (so[mid + 1] > b0 + b) hi = mid; else lo = mid + 1;
        }
        dst[4 * (lo + 1) + b] = values[b0 + b];
    }
}
