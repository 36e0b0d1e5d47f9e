// The chain of float / double additions of ONE reference filter call (flatscan.h has the arithmetic)
// for large waves, with the per-packet work spread over the whole device.
//
// k_filter_flat_scan keeps everything in one workgroup because each round needs the sum the round
// before it left.  What a round needs of that sum, though, is little: its exponent e (which fixes the
// grid) and, at the very end, its integer S.  So the packets are cut into chunks of 1,024, and a
// first pass guesses every chunk's exponent from an ordinary (inexact, parallel) prefix sum of the
// wave -- the exact chain cannot be far from it -- and works out, for the guess and its two
// neighbours, what the chunk does to ANY sum of that exponent:
//     step[p]   the chunk's total step on S, for a sum of parity p in front of it
//     lo[p], hi[p]   the least and the greatest exact partial sum inside the chunk, relative to S
// (a chunk's packets are parity maps; their composition and the running offsets come out of one
// workgroup scan per candidate exponent).  The last pass is one workgroup walking the chunks in
// order: if the sum's exponent is one of the chunk's candidates and S + lo[p] .. S + hi[p] stays inside
// the binade, the whole chunk is S += step[p]; otherwise (the first chunk, a chunk in which the sum
// changes its binade, non-finite values, a bad guess) its 1,024 packets are added the plain way.
// The exact sum is the chain's for every input: a chunk is only ever skipped under the same
// conditions under which flatscan.h's rounds accept packets, checked against the actual S.
//
// Integer range: while a chunk's test holds for the actual S, every partial step inside it is
// within 2^(MANT+2) of S, so nothing wraps; where a partial step wraps, an earlier packet of the
// same chunk has already pushed lo / hi out of every binade, and min / max keep that.
#pragma once

#include "flatscan.h"

#ifndef WIDE_TPB
#define WIDE_TPB 256
#endif
#define WIDE_EPL 4
#define WIDE_CHUNK ((uint32_t)(WIDE_TPB * WIDE_EPL))
#define WIDE_K 3                        // candidate exponents per chunk: guess - 1, guess, guess + 1
#define WIDE_STAGE 128u                 // chunk summaries the walk keeps in LDS at a time
static_assert(WIDE_TPB * WIDE_EPL == WIDE_CHUNK, "a lane owns consecutive packets of its chunk");

template <typename S> struct WideSum { S step[2], lo[2], hi[2]; };
struct WideHead { int32_t e0; uint32_t bad; };      // candidates e0 .. e0 + WIDE_K - 1; bit k: candidate k does not apply

// Workspace of a wave of n packets (16-byte aligned): chunk sums, chunk counts, heads, summaries.
template <typename T> struct WideWs {
    typedef typename Grid<T>::S S;
    double *sums; int32_t *cnts; WideHead *heads; WideSum<S> *maps;
    __host__ __device__ static uint64_t bytes(uint64_t nc)
    {
        return up(nc * sizeof(double)) + up(nc * sizeof(int32_t)) + up(nc * sizeof(WideHead)) + up(nc * WIDE_K * sizeof(WideSum<S>));
    }
    __host__ __device__ static uint64_t up(uint64_t b) { return (b + 15) & ~(uint64_t)15; }
    __host__ __device__ WideWs(void *base, uint64_t nc)
    {
        uint8_t *p = (uint8_t *)base;
        sums = (double *)p; p += up(nc * sizeof(double));
        cnts = (int32_t *)p; p += up(nc * sizeof(int32_t));
        heads = (WideHead *)p; p += up(nc * sizeof(WideHead));
        maps = (WideSum<S> *)p;
    }
};

// packet i of the wave as the fold takes it: the value, times its count for AVG (in T, as mul_cnt)
template <int OP, typename T> __device__ __forceinline__ T wide_value(const FilterIn &in, uint64_t i, int32_t *cnt)
{
    T v = load_value<T>(in, i);
    if (OP == MRPDR_AVG) {
        const int32_t c = load_count(in, i);
        v = mul_cnt<T>(v, c);
        *cnt += c;
    }
    return v;
}

__device__ __forceinline__ double wide_block_sum(double v, double *s_part)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int w = 0; w < WIDE_TPB / 64; w++) r += s_part[w];
    return r;
}

// ---- pass 1: an ordinary sum of every chunk (double, any order) and its AVG count --------------------
template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_sums(FilterIn in, void *ws_base, uint64_t nc)
{
    __shared__ double s_part[WIDE_TPB / 64];
    __shared__ int32_t s_cnt;
    WideWs<T> ws(ws_base, nc);
    const uint64_t c = blockIdx.x;
    if (threadIdx.x == 0) s_cnt = 0;
    double v = 0;
    int32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < WIDE_EPL; k++) {
        const uint64_t i = c * WIDE_CHUNK + (uint32_t)k * WIDE_TPB + threadIdx.x;
        if (i < in.n) v += (double)wide_value<OP, T>(in, i, &cnt);
    }
    const double total = wide_block_sum(v, s_part);
    if (OP == MRPDR_AVG) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&s_cnt, cnt);
        __syncthreads();
    }
    if (threadIdx.x == 0) { ws.sums[c] = total; ws.cnts[c] = s_cnt; }
}

// ---- pass 2: what every chunk does to a sum of its guessed exponent and of the two next to it ------
template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_maps(FilterIn in, void *ws_base, uint64_t nc)
{
    typedef Grid<T> G;
    typedef typename G::S S;
    __shared__ double s_part[WIDE_TPB / 64];
    __shared__ GridMap<S> s_wave[WIDE_TPB / 64];
    __shared__ S s_lo[2][WIDE_TPB / 64], s_hi[2][WIDE_TPB / 64];
    WideWs<T> ws(ws_base, nc);
    const uint64_t c = blockIdx.x;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    // the sum in front of the chunk, roughly
    double a = 0;
    for (uint64_t j = t; j < c; j += WIDE_TPB) a += ws.sums[j];
    a = wide_block_sum(a, s_part);
    WideHead head;
    head.e0 = 0; head.bad = (1u << WIDE_K) - 1u;
    uint64_t abits;
    __builtin_memcpy(&abits, &a, 8);
    const int aef = (int)((abits >> 52) & 0x7ff);
    const bool whole = (c + 1) * WIDE_CHUNK <= in.n;            // (the last, partial chunk is folded plainly)
    if (whole && aef != 0 && aef != 0x7ff) {
        head.e0 = aef - 1023 - 1;
        head.bad = 0;
        for (int k = 0; k < WIDE_K; k++) {
            const int e = head.e0 + k;
            if (e < 1 - G::BIAS || e > G::EMASK - 1 - G::BIAS) head.bad |= 1u << k;     // not a normal number of T
        }
    }
    if (head.bad != (1u << WIDE_K) - 1u) {
        T x[WIDE_EPL];
        int32_t unused = 0;
#pragma unroll
        for (int k = 0; k < WIDE_EPL; k++) x[k] = wide_value<OP, T>(in, c * WIDE_CHUNK + t * WIDE_EPL + (uint32_t)k, &unused);
        for (int cand = 0; cand < WIDE_K; cand++) {
            if ((head.bad >> cand) & 1u) continue;              // (uniform)
            const int e = head.e0 + cand;
            GridStep<T> st[WIDE_EPL];
            GridMap<S> f;
            f.a0 = f.a1 = 0;
            bool cross = false;
#pragma unroll
            for (int k = 0; k < WIDE_EPL; k++) {
                st[k] = grid_step<T>(x[k], e);
                cross = cross || st[k].cross;
                GridMap<S> g;
                g.a0 = st[k].step[0]; g.a1 = st[k].step[1];
                f = grid_then(f, g);
            }
            GridMap<S> inc = f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const GridMap<S> o = grid_shfl_up(inc, d);
                if ((int)lane >= d) inc = grid_then(o, inc);
            }
            GridMap<S> exc = grid_shfl_up(inc, 1);
            if (lane == 0) exc.a0 = exc.a1 = 0;
            __syncthreads();                                    // (s_wave, s_lo, s_hi of the candidate before)
            if (lane == 63) s_wave[wave] = inc;
            const bool any_cross = __syncthreads_or(cross) != 0;
            GridMap<S> pre, total;
            pre.a0 = pre.a1 = total.a0 = total.a1 = 0;
            for (uint32_t w = 0; w < WIDE_TPB / 64; w++) {
                if (w == wave) pre = total;
                total = grid_then(total, s_wave[w]);
            }
            exc = grid_then(pre, exc);
            // the exact partial sums of this lane's packets, relative to S, for both parities of S
            S lo[2], hi[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                S q = p ? exc.a1 : exc.a0;
                lo[p] = hi[p] = q + st[0].floor_i;
#pragma unroll
                for (int k = 0; k < WIDE_EPL; k++) {
                    const S tf = q + st[k].floor_i;
                    lo[p] = tf < lo[p] ? tf : lo[p];
                    hi[p] = tf > hi[p] ? tf : hi[p];
                    q += ((p + (int)(q & 1)) & 1) ? st[k].step[1] : st[k].step[0];      // (a select: an index would send st[] to scratch memory)
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    const S ol = grid_lane(lo[p], (int)(lane ^ (uint32_t)d)), oh = grid_lane(hi[p], (int)(lane ^ (uint32_t)d));
                    lo[p] = ol < lo[p] ? ol : lo[p];
                    hi[p] = oh > hi[p] ? oh : hi[p];
                }
                if (lane == 0) { s_lo[p][wave] = lo[p]; s_hi[p][wave] = hi[p]; }
            }
            __syncthreads();
            if (t == 0) {
                if (any_cross) head.bad |= 1u << cand;
                WideSum<S> out;
                for (int p = 0; p < 2; p++) {
                    S l = s_lo[p][0], h = s_hi[p][0];
                    for (int w = 1; w < WIDE_TPB / 64; w++) { l = s_lo[p][w] < l ? s_lo[p][w] : l; h = s_hi[p][w] > h ? s_hi[p][w] : h; }
                    out.lo[p] = l; out.hi[p] = h;
                    out.step[p] = p ? total.a1 : total.a0;
                }
                ws.maps[c * WIDE_K + cand] = out;
            }
        }
    }
    if (t == 0) ws.heads[c] = head;
}

// ---- pass 3: the chunks in order ---------------------------------------------------------------------
template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_walk(FilterIn in, void *ws_base, uint64_t nc, uint8_t *result, int32_t *result_ok)
{
    typedef Grid<T> G;
    typedef typename G::U U;
    typedef typename G::S S;
    __shared__ __align__(16) T s_val[WIDE_CHUNK + 64];          // (slack: the plain fold reads a batch ahead)
    __shared__ WideHead s_head[WIDE_STAGE];
    __shared__ WideSum<S> s_map[WIDE_STAGE * WIDE_K];
    __shared__ T s_acc;
    __shared__ int32_t s_cnt;
    WideWs<T> ws(ws_base, nc);
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) s_cnt = 0;
    T acc = (T)0;                                               // the same in every lane
#ifdef WIDE_DEBUG
    uint32_t nplain = 0;
#endif
    const S LO = (S)1 << G::MANT, HI = ((S)1 << (G::MANT + 1)) - 2;
    T ahead[WIDE_EPL];                                          // a chunk's packets, loaded ahead
    uint64_t ahead_of = ~(uint64_t)0;
    for (uint64_t c0 = 0; c0 < nc; c0 += WIDE_STAGE) {
        const uint32_t m = (uint32_t)(nc - c0 < WIDE_STAGE ? nc - c0 : WIDE_STAGE);
        __syncthreads();
        for (uint32_t j = t; j < m; j += WIDE_TPB) s_head[j] = ws.heads[c0 + j];
        for (uint32_t j = t; j < m * WIDE_K; j += WIDE_TPB) s_map[j] = ws.maps[c0 * WIDE_K + j];
        __syncthreads();
        for (uint32_t j = 0; j < m; j++) {
            const U ab = (U)to_bits<T>(acc);
            const int aef = (int)((ab >> G::MANT) & (U)G::EMASK);
            const WideHead head = s_head[j];
            bool done = false;
            if (aef != 0 && aef != G::EMASK) {
                const int e = aef - G::BIAS, cand = e - head.e0;
                if (cand >= 0 && cand < WIDE_K && !((head.bad >> cand) & 1u)) {
                    const U afrac = (ab & (((U)1 << G::MANT) - 1)) | ((U)1 << G::MANT);
                    const S s0 = (ab >> (sizeof(U) * 8 - 1)) ? -(S)afrac : (S)afrac;
                    const int p = (int)(s0 & 1);
                    const WideSum<S> *w = &s_map[j * WIDE_K + (uint32_t)cand];      // (read in place: a copy indexed by p would live in scratch memory)
                    const S l = s0 + w->lo[p], h = s0 + w->hi[p];
                    // every exact partial sum of the chunk inside the binade (the same conservative
                    // ends as grid_inside), all on one side of zero
                    if ((l >= LO && h <= HI) || (h <= -LO - 1 && l >= -HI)) {
                        const S sum = s0 + w->step[p];
                        acc = from_bits<T>((uint64_t)(((U)(e + G::BIAS) << G::MANT) | ((U)(sum < 0 ? -sum : sum) & (((U)1 << G::MANT) - 1)) |
                                                      (sum < 0 ? (U)1 << (sizeof(U) * 8 - 1) : (U)0)));
                        done = true;
                    }
                }
            }
            if (done) continue;                                 // (uniform: acc is the same in every lane)
#ifdef WIDE_DEBUG
            nplain++;
#endif
            // the chunk's packets the plain way: staged by everybody, added up by one wavefront.  The
            // next chunk's packets are asked for before the adding starts: if that chunk has to be
            // taken the plain way too (runs of them are the rule: zeros, non-finite sums, a sum that
            // hovers), its values are in registers by the time this one is done.
            const uint64_t cc = c0 + j, first = cc * WIDE_CHUNK;
            const uint32_t len = (uint32_t)(in.n - first < WIDE_CHUNK ? in.n - first : WIDE_CHUNK);
            if (ahead_of != cc) {
#pragma unroll
                for (int k = 0; k < WIDE_EPL; k++) {
                    const uint64_t i = first + (uint32_t)k * WIDE_TPB + t;
                    int32_t unused = 0;
                    ahead[k] = wide_value<OP, T>(in, i < in.n ? i : in.n - 1, &unused);
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < WIDE_EPL; k++) s_val[(uint32_t)k * WIDE_TPB + t] = ahead[k];
            __syncthreads();
            if (cc + 1 < nc) {
#pragma unroll
                for (int k = 0; k < WIDE_EPL; k++) {
                    const uint64_t i = first + WIDE_CHUNK + (uint32_t)k * WIDE_TPB + t;
                    int32_t unused = 0;
                    ahead[k] = wide_value<OP, T>(in, i < in.n ? i : in.n - 1, &unused);
                }
                ahead_of = cc + 1;
            }
            if (wave == 0) {
                acc = plain_fold<T>(acc, s_val, len);           // (values behind len are staged but not added)
                if (lane == 0) s_acc = acc;
            }
            __syncthreads();
            acc = s_acc;
        }
    }
#ifdef WIDE_DEBUG
    if (t == 0) printf("wide walk: %u of %llu chunks plain\n", nplain, (unsigned long long)nc);
#endif
    int32_t cnt = 0;
    if (OP == MRPDR_AVG) {
        for (uint64_t j = t; j < nc; j += WIDE_TPB) cnt += ws.cnts[j];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        __syncthreads();
        if (lane == 0) atomicAdd(&s_cnt, cnt);
        __syncthreads();
    }
    if (t == 0) {
        cnt = s_cnt;
        bool present = in.n != 0;
        if (OP == MRPDR_AVG) {
            if (cnt == 0) present = false;
            else acc = div_cnt<T>(acc, cnt);
        }
        write_result<OP, T>(result, result_ok, present, acc, cnt);
    }
}
