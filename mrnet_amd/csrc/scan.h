// Exclusive prefix scan over per-packet lengths / element counts.
//
// This is the batched form of the reference's sizing pass: pdr_sizeof() walks one
// packet to learn its length (src/pdr_sizeof.c:120-140) so that encode_pdr_data can
// malloc it (src/Packet.C:61-69).  Here every lane sizes packets, a 64-lane wavefront
// scan turns lengths into offsets, and three small launches stitch the workgroups
// together (partial sums -> scan of the sums -> final offsets).
#pragma once

#include "common.h"

#define SCAN_IPT 8                          // items per thread
#define SCAN_TILE (MRPDR_TPB * SCAN_IPT)    // items per workgroup

__device__ inline uint64_t round_up_pow2(uint64_t v, uint32_t align)
{
    return (v + (uint64_t)(align - 1)) & ~(uint64_t)(align - 1);
}

// Inclusive scan across the 64 lanes of a wavefront.
__device__ inline uint64_t wave_inclusive_scan(uint64_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t up = __shfl_up((unsigned long long)v, d, 64);
        if (lane >= d) v += up;
    }
    return v;
}

// Workgroup-wide exclusive scan of one value per thread (NW wavefronts per workgroup);
// *total = sum over the group.
template <int NW = MRPDR_TPB / 64>
__device__ inline uint64_t block_exclusive_scan(uint64_t v, uint64_t *total)
{
    __shared__ uint64_t wave_sums[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint64_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint64_t s = wave_sums[w];
        if (w < wave) base += s;
        sum += s;
    }
    __syncthreads();
    *total = sum;
    return base + inc - v;
}

// Every item occupies a slot of round_up(phase + item, align) bytes and starts `phase`
// bytes into its slot (phase = 0, align = 1: a plain exclusive scan).
// Pass 1: sum of the slots of each tile.
__global__ __launch_bounds__(MRPDR_TPB) void k_scan_partial(const uint64_t *in, uint64_t n, uint32_t align,
                                                            uint32_t phase, uint64_t *tile_sums)
{
    const uint64_t first = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_IPT;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++)
        if (first + k < n) s += round_up_pow2(in[first + k] + phase, align);
    uint64_t total;
    block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// Pass 2: offsets[i] = sums of the tiles in front (a few thousand at most, added up by the
// workgroup itself: cheaper than a launch in between that scans them) + scan inside the tile;
// offsets[n] = grand total.
__global__ __launch_bounds__(MRPDR_TPB) void k_scan_final(const uint64_t *in, uint64_t n, uint32_t align,
                                                          uint32_t phase, const uint64_t *tile_sums,
                                                          uint64_t *out)
{
    uint64_t before = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += MRPDR_TPB) before += tile_sums[j];
    uint64_t tile_base;
    block_exclusive_scan(before, &tile_base);
    const uint64_t first = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_IPT;
    uint64_t v[SCAN_IPT];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) {
        v[k] = first + k < n ? round_up_pow2(in[first + k] + phase, align) : 0;
        s += v[k];
    }
    uint64_t total;
    uint64_t run = tile_base + block_exclusive_scan(s, &total);
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) {
        if (first + k < n) out[first + k] = run + phase;
        run += v[k];
        if (first + k + 1 == n) out[n] = run;
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

// Small waves: the whole scan in ONE launch by one 1024-lane workgroup that walks the
// items tile by tile with a running carry.  With SIZED the item is not read from `in`
// but computed from the wave's count columns -- the pdr_sizeof dry run of every packet
// (src/pdr_sizeof.c:120-140) -- and also stored to lens[].
#define SCAN1_TPB 1024
#define SCAN1_MAX_N (4u * SCAN1_TPB * SCAN_IPT)     // above this the two-launch scan is faster

template <bool SIZED>
__global__ __launch_bounds__(SCAN1_TPB) void k_scan_single(const uint64_t *in, FmtDev fmt, ColsDev cols,
                                                           uint64_t n, uint32_t align, uint32_t phase,
                                                           uint64_t *lens, uint64_t *out, uint32_t *status)
{
    uint64_t carry = 0;
    bool bad = false;
    for (uint64_t base = 0; base < n; base += (uint64_t)SCAN1_TPB * SCAN_IPT) {
        const uint64_t first = base + (uint64_t)threadIdx.x * SCAN_IPT;
        uint64_t v[SCAN_IPT];
        uint64_t s = 0;
        // (n > 0 inside the loop; all loads of the lane are issued before the first is used)
        if (SIZED) packets_bytes<SCAN_IPT>(fmt, cols, first, n, v, &bad);
        else {
#pragma unroll
            for (int k = 0; k < SCAN_IPT; k++) v[k] = in[first + k < n ? first + k : n - 1];
        }
#pragma unroll
        for (int k = 0; k < SCAN_IPT; k++) {
            uint64_t item = 0;
            if (first + k < n) {
                if (SIZED) lens[first + k] = v[k];
                item = round_up_pow2(v[k] + phase, align);
            }
            v[k] = item;
            s += item;
        }
        uint64_t total;
        uint64_t run = carry + block_exclusive_scan<SCAN1_TPB / 64>(s, &total);
#pragma unroll
        for (int k = 0; k < SCAN_IPT; k++) {
            if (first + k < n) out[first + k] = run + phase;
            run += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) out[n] = carry;
    if (SIZED && bad) atomicOr(status, 4u);
}

// Larger waves, sizing + offsets in TWO launches: every workgroup sizes SIZE_PKTS packets and
// leaves the sum of their slots; the second launch adds up the sums in front of the workgroup
// (a few thousand at most, read in parallel) and scans its own packets.
#define SIZE_PKTS (MRPDR_TPB * 4)

__global__ __launch_bounds__(MRPDR_TPB) void k_size_sums(FmtDev fmt, ColsDev cols, uint64_t n, uint32_t align,
                                                         uint32_t phase, uint64_t *lens, uint64_t *sums,
                                                         uint32_t *status)
{
    bool bad = false;
    const uint64_t first = (uint64_t)blockIdx.x * SIZE_PKTS + (uint64_t)threadIdx.x * 4;
    uint64_t s = 0;
    uint64_t item[4];
    packets_bytes<4>(fmt, cols, first, n, item, &bad);       // the launch has n > 0
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (first + k >= n) break;
        lens[first + k] = item[k];
        s += round_up_pow2(item[k] + phase, align);
    }
    if (bad) atomicOr(status, 4u);
    uint64_t total;
    block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(MRPDR_TPB) void k_size_offs(const uint64_t *lens, uint64_t n, uint32_t align,
                                                         uint32_t phase, const uint64_t *sums, uint64_t *out)
{
    uint64_t s = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += MRPDR_TPB) s += sums[j];
    uint64_t base;
    block_exclusive_scan(s, &base);
    const uint64_t first = (uint64_t)blockIdx.x * SIZE_PKTS + (uint64_t)threadIdx.x * 4;
    uint64_t v[4];
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = first + k < n ? round_up_pow2(lens[first + k] + phase, align) : 0;
        mine += v[k];
    }
    uint64_t total;
    uint64_t run = base + block_exclusive_scan(mine, &total);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (first + k < n) out[first + k] = run + phase;
        run += v[k];
        if (first + k + 1 == n) out[n] = run;
    }
}

// Packed layout of a fixed-length format: closed form, no scan needed.
__global__ __launch_bounds__(MRPDR_TPB) void k_offsets_fixed(uint64_t n, uint64_t len, uint64_t stride,
                                                             uint32_t phase, uint64_t *lens, uint64_t *offsets)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i < n && lens) lens[i] = len;
    if (i <= n && offsets) offsets[i] = i * stride + (i < n ? phase : 0);
}
