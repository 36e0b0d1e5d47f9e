// Decode of a packed scalar-only wave and the filter over its first field in one pass.
//
// What a tree node does with an upstream wave (src/Stream.C:543-664): every packet is decoded
// (Packet::decode_pdr_data, src/Packet.C:106-133) because the filter asks for (*pkt)[0]
// (src/FilterDefinitions.C:148), and the filter folds those values.  Done as two calls the rows
// are read twice and the wave costs three launches (rows -> columns; columns -> tree nodes; nodes
// -> result).  Here a workgroup stages its 1,024 rows in LDS once, writes the columns AND folds the
// first field (with the int32 count behind it for AVG) into its tree node; the last level is the
// filter's own second launch (a wave of one tile takes it in the same launch).  Lane t takes rows 4t .. 4t+3 -- the leaf of the declared tree -- so
// the nodes, and with them the result, are those of mrpdr_wave_filter over the decoded column, bit
// for bit, for the tree shape and for every flat result that is computed level by level.
#pragma once

#include "common.h"
#include "filter.h"
#include "rows.h"

#define FUSED_MAX_STRIDE 32          // 32 KiB of LDS per workgroup at most

// One decoded scalar out of a row staged in LDS (the same reads as k_rows_decode).
__device__ __forceinline__ uint64_t fused_field(const uint8_t *row, uint32_t at, uint32_t w, uint32_t stride, bool sw)
{
    uint64_t v = 0;
    if (((at | stride) & (w - 1)) == 0) {
        switch (w) {
        case 1: v = row[at]; break;
        case 2: v = *reinterpret_cast<const uint16_t *>(row + at); if (sw) v = __builtin_bswap16((uint16_t)v); break;
        case 4: v = *reinterpret_cast<const uint32_t *>(row + at); if (sw) v = __builtin_bswap32((uint32_t)v); break;
        default: v = *reinterpret_cast<const uint64_t *>(row + at); if (sw) v = __builtin_bswap64(v); break;
        }
    } else {
        for (uint32_t b = 0; b < w; b++) v |= (uint64_t)row[at + (sw ? (w - 1 - b) : b)] << (8 * b);
    }
    return v;
}

template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_rows_decode_filter(FmtDev fmt, WireDev wire, ColsDev cols, uint64_t n,
                                                                  int32_t swap, int32_t *ok, Partial *partials,
                                                                  uint8_t *result, int32_t *result_ok)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    __shared__ Partial wave_part[MRPDR_TPB / 64];
    const uint64_t p0 = (uint64_t)blockIdx.x * FILTER_TILE;
    const uint32_t np = (uint32_t)((n - p0) < FILTER_TILE ? (n - p0) : FILTER_TILE);
    const uint32_t stride = (uint32_t)wire.stride;
    const uint32_t tile_bytes = np * stride;
    // p0 * stride is a multiple of 1,024 bytes: with a 16-byte aligned arena every 16-byte piece of
    // the tile is an aligned 128-bit load; a lane's loads are issued together (unconditional: a lane
    // past the end re-reads the last piece)
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
    T vals[FILTER_LEAF];
    int32_t cnts[FILTER_LEAF];
    int nvalid = 0;
#pragma unroll
    for (int k = 0; k < FILTER_LEAF; k++) {
        vals[k] = T();
        cnts[k] = 1;
        const uint32_t lr = threadIdx.x * FILTER_LEAF + (uint32_t)k;
        if (lr >= np) continue;
        const uint64_t i = p0 + lr;
        const uint8_t *row = tile + lr * stride;
        const uint64_t len = wire.lens ? wire.lens[i] : (uint64_t)fmt.fixed_bytes;
        const bool sw = pkt_swap(wire, i, swap != 0);
        bool good = true;
        for (int f = 0; f < fmt.nf; f++) {
            const uint32_t w = fmt.f[f].width, at = fmt.f[f].fixed_before;
            uint64_t v = 0;
            // same test as every pdrmem_get*: width > remaining space -> FALSE
            if (good && (uint64_t)at + w <= len && at + w <= stride) v = fused_field(row, at, w, stride, sw);
            else good = false;
            st_w(reinterpret_cast<uint8_t *>(cols.values[f]) + i * w, (int)w, v);
            if (f == 0) vals[k] = from_bits<T>(v);
            if (f == 1 && (OP == MRPDR_AVG || OP == OP_AVGSUM)) cnts[k] = (int32_t)(uint32_t)v;
        }
        if (ok) ok[i] = good ? 1 : 0;
        nvalid = k + 1;
    }
    if (OP == OP_AVGSUM) {
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) vals[k] = mul_cnt<T>(vals[k], cnts[k]);
    } else if (OP == OP_MINR || OP == OP_MAXR) {
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) cnts[k] = vals[k] != vals[k];
    }
    const Partial out = tile_node<OP, T>(vals, cnts, nvalid, wave_part);
    if (threadIdx.x != 0) return;
    // (result != NULL: the wave is this one tile, and the last level is taken here)
    if (result) top_of_one<OP, T>(out, result, result_ok);
    else partials[blockIdx.x] = out;
}
