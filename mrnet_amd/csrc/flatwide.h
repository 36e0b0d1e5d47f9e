// The chain of float / double additions of ONE reference filter call (flatscan.h has the arithmetic)
// for large waves, with the per-packet work spread over the whole device.
//
// k_filter_flat_scan keeps everything in one workgroup because each round needs the sum the round
// before it left.  What a round needs of that sum, though, is little: its exponent e (which fixes the
// grid) and, at the very end, its integer S.  So the packets are cut into chunks of 1,024 (four
// sub-chunks of 256, a wavefront each), and a first pass guesses the sum's exponent in front of every
// sub-chunk from an ordinary (inexact, parallel) prefix sum of the wave -- the exact chain cannot be
// far from it -- and works out, for the guess and its two neighbours, what the sub-chunk does to ANY
// sum of that exponent, of either parity p:
//     step   the sub-chunk's total step on S
//     lo, hi the least and the greatest exact partial sum inside it, relative to S
// (the packets are parity maps; their composition and the running offsets come out of a wavefront
// scan per candidate exponent).  Such summaries compose like the maps themselves (wide_then), which
// is used twice: a chunk's summary is put together from those of its four sub-chunks (where they
// agree on the exponent -- they do unless the rough sum changes its binade inside the chunk), and
// the walk puts the summaries of up to 64 chunks together to step over all of them at once.
// The last pass is one workgroup walking the chunks in order: if the sum's exponent is one of the
// chunk's candidates and S + lo .. S + hi stays inside the binade, the whole chunk is S += step;
// otherwise (the first chunk, a chunk in which the sum changes its binade, non-finite values, a bad
// guess) the chunk is opened and gone through sub-chunk by sub-chunk, each in one step where its own
// summary allows and packet by packet where not.  The exact sum is the chain's for every input: packets
// are only ever skipped under the same conditions under which flatscan.h's rounds accept packets,
// checked against the actual S.  The walk is one wavefront's chain of dependent instructions (an
// instruction every ~3.8 ns): everything about it is arranged to keep that chain short -- one trip to
// LDS per chunk, chunks that will probably be opened marked by the second pass and loaded ahead of
// time, loading and staging left to the other wavefronts.
//
// Integer range: every packet's step is below 2^(MANT+2) in size, so a running sum cannot pass from
// inside a binade to beyond the integer range without a partial sum landing in between, where it fails
// the test for every S; lo / hi keep that (min / max) and are clamped to +-2^(MANT+3) wherever they
// are produced, so that adding S or a step in front of them never wraps.
#pragma once

#include "flatscan.h"

#ifndef WIDE_TPB
#define WIDE_TPB 256
#endif
#define WIDE_EPL 4
#define WIDE_CHUNK ((uint32_t)(WIDE_TPB * WIDE_EPL))
#define WIDE_K 3                        // candidate exponents per chunk: guess - 1, guess, guess + 1
#define WIDE_STAGE 128u                 // chunk summaries the walk keeps in LDS at a time
#ifndef WIDE_MANY_MIN
#define WIDE_MANY_MIN 4u                // chunks the hints promise in a row from which the walk takes them many at a time
#endif
// (2, 3, 4 and 6 measured within 1 % of one another)

#define WIDE_SUBS (WIDE_TPB / 64)        // a wavefront's packets (256) are a sub-chunk with summaries of its own
static_assert(WIDE_TPB * WIDE_EPL == WIDE_CHUNK && 4 + WIDE_K * WIDE_SUBS <= 31, "head.bad holds a bit per candidate and sub-chunk");

// What a run of packets does to a sum of parity p in front of it is of[p]: the least and the greatest
// exact partial sum inside the run and the run's total step, all relative to S.  (One side is what the
// walk reads per chunk: 16 or 32 bytes, one or two LDS reads.)
// `e` is the exponent the side is for (WIDE_NO_E: for none -- a packet of the run is out of that
// exponent's reach, or the exponent is not a normal number's): the walk reads all three candidates'
// sides at once and takes the one whose e is the sum's, without looking anything else up.
template <typename S> struct WideSide { S lo, hi, step, e; };
#define WIDE_NO_E(S) ((S)1 << 30)
template <typename S> struct __align__(16) WideSum { WideSide<S> of[2]; };
// Candidates e0 .. e0 + WIDE_K - 1.  bad, bit k: candidate k does not apply to the chunk as a whole;
// bit 4 + k * WIDE_SUBS + w: nor to its sub-chunk w; WIDE_HINT: the walk will probably have to open the chunk.
struct WideHead { int32_t e0; uint32_t bad; };
#define WIDE_HINT 0x80000000u
#define WIDE_CHUNK_BITS ((1u << WIDE_K) - 1u)
#define WIDE_ALL_BAD ((1u << (4 + WIDE_K * WIDE_SUBS)) - 1u)
#define WIDE_SUB_BITS (WIDE_ALL_BAD & ~15u)
__host__ __device__ constexpr uint32_t wide_bad_sub(int cand, uint32_t w) { return 1u << (4 + cand * WIDE_SUBS + (int)w); }
// bit WIDE_SKIP_SHIFT + w: sub-chunk w will probably not go in one step (the walk's own test on the rough
// sum in front of it fails, with a margin): the walk adds it plainly without looking at its summaries.
// Like WIDE_HINT a guess that costs or saves time only.
#define WIDE_SKIP_SHIFT (4 + WIDE_K * WIDE_SUBS)
#ifndef WIDE_SKIP_MARGIN
#define WIDE_SKIP_MARGIN 12             // the margin of that test: 2^-12 of the binade's lower end
#endif
// (measured on five zero-mean draws and the tests' kinds: 2^-5 costs 2-8 %, 2^-8, 2^-11 and 2^-14 lie within
// 2 % of one another)
static_assert(WIDE_SKIP_SHIFT + WIDE_SUBS <= 31 && WIDE_K == 3, "head.bad holds the skip bits below WIDE_HINT; the walk ands three candidates' bits");
__host__ __device__ constexpr uint32_t wide_bad_all_of(int cand) { return (1u << cand) | (((1u << WIDE_SUBS) - 1u) << (4 + cand * WIDE_SUBS)); }

// Workspace of a wave of n packets (16-byte aligned): chunk sums, chunk counts, heads, summaries.
template <typename T> struct WideWs {
    typedef typename Grid<T>::S S;
    double *sums; int32_t *cnts; WideHead *heads; WideSum<S> *maps, *subs;
    __host__ __device__ static uint64_t bytes(uint64_t nc)
    {
        return up(nc * sizeof(double)) + up(nc * sizeof(int32_t)) + up(nc * sizeof(WideHead)) + up(nc * WIDE_K * sizeof(WideSum<S>)) +
               up(nc * WIDE_K * WIDE_SUBS * sizeof(WideSum<S>));
    }
    __host__ __device__ static uint64_t up(uint64_t b) { return (b + 15) & ~(uint64_t)15; }
    __host__ __device__ WideWs(void *base, uint64_t nc)
    {
        uint8_t *p = (uint8_t *)base;
        sums = (double *)p; p += up(nc * sizeof(double));
        cnts = (int32_t *)p; p += up(nc * sizeof(int32_t));
        heads = (WideHead *)p; p += up(nc * sizeof(WideHead));
        maps = (WideSum<S> *)p; p += up(nc * WIDE_K * sizeof(WideSum<S>));
        subs = (WideSum<S> *)p;
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

// A packet as it lies in memory -- its value, its count for AVG, its byte order if the wave carries
// one per packet -- and what the fold makes of it.  Two steps, so that loads issued ahead of time stay
// in flight: turning the bytes round right behind the load would wait for it.
template <typename T> struct WideRaw { typename UIntOf<sizeof(T)>::type v; uint32_t c; uint8_t o; };
template <int OP, typename T> __device__ __forceinline__ WideRaw<T> wide_load(const FilterIn &in, uint64_t i)
{
    WideRaw<T> r;
    r.v = ld_u<typename UIntOf<sizeof(T)>::type>(in.values + i * in.value_stride);
    r.c = 0; r.o = 0;
    if (OP == MRPDR_AVG) r.c = ld_u<uint32_t>(in.counts + i * in.count_stride);
    if (in.orders) r.o = in.orders[i];
    return r;
}
template <int OP, typename T> __device__ __forceinline__ T wide_value_of(const FilterIn &in, const WideRaw<T> &r)
{
    const bool swap = in.orders ? r.o != 0 : in.swap != 0;
    T v = from_bits<T>((uint64_t)(swap ? bswap_u(r.v) : r.v));
    if (OP == MRPDR_AVG) v = mul_cnt<T>(v, (int32_t)(swap ? bswap_u(r.c) : r.c));
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
__device__ __forceinline__ void wide_sums_body(const FilterIn &in, void *ws_base, uint64_t nc)
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
// lo / hi never leave +-2^(MANT+3): anything at or beyond 2^(MANT+2) in size fails the walk's test for
// every S, and kept small it cannot wrap when S or a step in front is added to it.
template <typename T> __device__ __forceinline__ typename Grid<T>::S wide_clamp(typename Grid<T>::S v)
{
    typedef typename Grid<T>::S S;
    const S c = (S)1 << (Grid<T>::MANT + 3);
    return v > c ? c : (v < -c ? -c : v);
}

// What two runs of packets do one after the other (b behind a), from what each does on its own.
template <typename T> __device__ __forceinline__ WideSum<typename Grid<T>::S> wide_then(const WideSum<typename Grid<T>::S> &a, const WideSum<typename Grid<T>::S> &b)
{
    typedef typename Grid<T>::S S;
    WideSum<S> c;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const S st = a.of[p].step;
        const bool odd = ((p + (int)(st & 1)) & 1) != 0;        // the parity of S + st
        const S bl = st + (odd ? b.of[1].lo : b.of[0].lo), bh = st + (odd ? b.of[1].hi : b.of[0].hi);
        c.of[p].lo = wide_clamp<T>(a.of[p].lo < bl ? a.of[p].lo : bl);
        c.of[p].hi = wide_clamp<T>(a.of[p].hi > bh ? a.of[p].hi : bh);
        c.of[p].step = st + (odd ? b.of[1].step : b.of[0].step);
        c.of[p].e = a.of[p].e == b.of[p].e ? a.of[p].e : WIDE_NO_E(S);
    }
    return c;
}

// every exact partial sum S + lo .. S + hi inside the binade (the same conservative ends as
// grid_inside, drawn in by `margin`), all on one side of zero
template <typename T> __device__ __forceinline__ bool wide_inside(typename Grid<T>::S l, typename Grid<T>::S h, typename Grid<T>::S margin)
{
    typedef typename Grid<T>::S S;
    const S LO = (S)1 << Grid<T>::MANT, HI = ((S)1 << (Grid<T>::MANT + 1)) - 2;
    return (l >= LO + margin && h <= HI - margin) || (h <= -LO - 1 - margin && l >= -HI + margin);
}

template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_sums(FilterIn in, void *ws_base, uint64_t nc)
{
    wide_sums_body<OP, T>(in, ws_base, nc);
}

template <int OP, typename T>
__device__ __forceinline__ void wide_maps_body(const FilterIn &in, void *ws_base, uint64_t nc, const FilterLink &link)
{
    typedef Grid<T> G;
    typedef typename G::S S;
    __shared__ double s_part[WIDE_TPB / 64];
    __shared__ WideSum<S> s_sub[WIDE_K * WIDE_SUBS];
    __shared__ uint32_t s_bad;
    WideWs<T> ws(ws_base, nc);
    const uint64_t c = blockIdx.x;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) s_bad = 0;
    const bool whole = (c + 1) * WIDE_CHUNK <= in.n;            // (the last, partial chunk is folded plainly)
    // the chunk's packets first: their loads are in flight while the sums in front are added up
    WideRaw<T> raw[WIDE_EPL];
#pragma unroll
    for (int k = 0; k < WIDE_EPL; k++) raw[k] = wide_load<OP, T>(in, whole ? c * WIDE_CHUNK + t * WIDE_EPL + (uint32_t)k : 0);
    // the sum in front of the chunk, roughly
    double a = 0;
    for (uint64_t j = t; j < c; j += WIDE_TPB) a += ws.sums[j];
    a = wide_block_sum(a, s_part);                              // (its barriers are behind s_bad = 0)
    a += (double)chain_start<T>(link);                          // (a range of a wave: the sum the packets in front left)
    // The candidates are chosen per sub-chunk, from the rough sum in front of THAT sub-chunk: where the
    // sum moves by several binades inside a chunk (an excursion and its way back, zero-mean data near
    // zero, the first chunk of every wave) the sub-chunks behind the move then have summaries for the
    // exponents the sum really has there.  A slot of the chunk's own summary holds only where all four
    // sub-chunks agree on the slot's exponent (wide_then), which they do unless the rough sum changes its
    // binade inside the chunk -- and then the chunk is opened anyway.
    __shared__ double s_wsum[WIDE_SUBS];
    T x[WIDE_EPL];
    double mine = 0;
#pragma unroll
    for (int k = 0; k < WIDE_EPL; k++) { x[k] = wide_value_of<OP, T>(in, raw[k]); mine += (double)x[k]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane == 0) s_wsum[wave] = mine;
    __syncthreads();
    double aw = a;
    for (uint32_t w = 0; w < wave; w++) aw += s_wsum[w];
    WideHead head;
    head.e0 = 0; head.bad = 0;
    int e0w = 0;
    uint32_t badw = WIDE_CHUNK_BITS;                             // bit k: this sub-chunk has no candidate k
    {
        uint64_t bits;
        __builtin_memcpy(&bits, &aw, 8);
        const int ef = (int)((bits >> 52) & 0x7ff);
        if (whole && ef != 0 && ef != 0x7ff) {
            e0w = ef - 1023 - 1;
            badw = 0;
            for (int k = 0; k < WIDE_K; k++) {
                const int e = e0w + k;
                if (e < 1 - G::BIAS || e > G::EMASK - 1 - G::BIAS) badw |= 1u << k;     // not a normal number of T
            }
        }
        __builtin_memcpy(&bits, &a, 8);
        head.e0 = (int)((bits >> 52) & 0x7ff) - 1023 - 1;      // (sub-chunk 0's: what the hint below is computed for)
    }
    if (badw != WIDE_CHUNK_BITS) {
        // The candidates side by side, not one after the other: three independent chains of shuffles.
        // (A candidate already marked bad is computed as well and not looked at.)  Every wavefront
        // sums up its own 256 packets, relative to the first of them.
        GridStep<T> st[WIDE_K][WIDE_EPL];
        GridMap<S> inc[WIDE_K], exc[WIDE_K];
        uint32_t cross = 0;
#pragma unroll
        for (int cand = 0; cand < WIDE_K; cand++) {
            GridMap<S> f;
            f.a0 = f.a1 = 0;
#pragma unroll
            for (int k = 0; k < WIDE_EPL; k++) {
                st[cand][k] = grid_step<T>(x[k], e0w + cand);
                cross |= st[cand][k].cross ? 1u << cand : 0u;
                GridMap<S> g;
                g.a0 = st[cand][k].step[0]; g.a1 = st[cand][k].step[1];
                f = grid_then(f, g);
            }
            inc[cand] = f;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int cand = 0; cand < WIDE_K; cand++) {
                const GridMap<S> o = grid_shfl_up(inc[cand], d);
                if ((int)lane >= d) inc[cand] = grid_then(o, inc[cand]);
            }
        }
        // the exact partial sums of this lane's packets, relative to S, for both parities of S
        S lo[WIDE_K][2], hi[WIDE_K][2];
#pragma unroll
        for (int cand = 0; cand < WIDE_K; cand++) {
            exc[cand] = grid_shfl_up(inc[cand], 1);
            if (lane == 0) exc[cand].a0 = exc[cand].a1 = 0;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                S q = p ? exc[cand].a1 : exc[cand].a0;
                lo[cand][p] = hi[cand][p] = q + st[cand][0].floor_i;
#pragma unroll
                for (int k = 0; k < WIDE_EPL; k++) {
                    const S tf = q + st[cand][k].floor_i;
                    lo[cand][p] = tf < lo[cand][p] ? tf : lo[cand][p];
                    hi[cand][p] = tf > hi[cand][p] ? tf : hi[cand][p];
                    q += ((p + (int)(q & 1)) & 1) ? st[cand][k].step[1] : st[cand][k].step[0];      // (a select: an index would send st[] to scratch memory)
                }
                lo[cand][p] = wide_clamp<T>(lo[cand][p]);
                hi[cand][p] = wide_clamp<T>(hi[cand][p]);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
            for (int cand = 0; cand < WIDE_K; cand++) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const S ol = grid_lane(lo[cand][p], (int)(lane ^ (uint32_t)d)), oh = grid_lane(hi[cand][p], (int)(lane ^ (uint32_t)d));
                    lo[cand][p] = ol < lo[cand][p] ? ol : lo[cand][p];
                    hi[cand][p] = oh > hi[cand][p] ? oh : hi[cand][p];
                }
            }
        }
#pragma unroll
        for (int cand = 0; cand < WIDE_K; cand++) {
            if (__ballot((cross >> cand) & 1u) != 0) badw |= 1u << cand;
        }
        if (lane == 63) {
#pragma unroll
            for (int cand = 0; cand < WIDE_K; cand++) {
                WideSum<S> w;
                w.of[0].step = inc[cand].a0; w.of[1].step = inc[cand].a1;
                w.of[0].lo = lo[cand][0]; w.of[1].lo = lo[cand][1];
                w.of[0].hi = hi[cand][0]; w.of[1].hi = hi[cand][1];
                w.of[0].e = w.of[1].e = (badw >> cand) & 1u ? WIDE_NO_E(S) : (S)(e0w + cand);
                s_sub[cand * WIDE_SUBS + wave] = w;
            }
            // Will the walk take this sub-chunk in one step?  Slot 1 is the exponent of the rough sum in
            // front of it: the walk's test on that sum, with a margin for how far the exact chain may be
            // from it.  A sub-chunk that fails it is not worth the walk's look (which costs what some 90
            // additions cost): it is marked, and added plainly.
            bool skip = (badw & 2u) != 0;
            if (!skip) {
                const S sa = (S)scalbn(aw, G::MANT - (e0w + 1));
                const S margin = (S)1 << (G::MANT - WIDE_SKIP_MARGIN);
                for (int p = 0; p < 2; p++) skip = skip || !wide_inside<T>(sa + lo[1][p], sa + hi[1][p], margin);
            }
            if (skip) atomicOr(&s_bad, 1u << (WIDE_SKIP_SHIFT + wave));
        }
    } else if (lane == 63) {
        // no candidate at all for this sub-chunk (the sum in front of it is zero or not finite, or this is
        // the wave's last, partial chunk): the walk reads the summaries before anything else, so they
        // have to say so themselves
        WideSum<S> none;
        for (int p = 0; p < 2; p++) { none.of[p].lo = none.of[p].hi = none.of[p].step = 0; none.of[p].e = WIDE_NO_E(S); }
#pragma unroll
        for (int cand = 0; cand < WIDE_K; cand++) s_sub[cand * WIDE_SUBS + wave] = none;
    }
    if (lane == 63 && badw) {
        uint32_t bits = 0;
#pragma unroll
        for (int cand = 0; cand < WIDE_K; cand++) if ((badw >> cand) & 1u) bits |= (1u << cand) | wide_bad_sub(cand, wave);
        atomicOr(&s_bad, bits);
    }
    __syncthreads();
    if (t < WIDE_K * WIDE_SUBS) ws.subs[c * (WIDE_K * WIDE_SUBS) + t] = s_sub[t];
    if (t < WIDE_K) {
        // (thread `cand` puts slot `cand` of the sub-chunks together; a slot on whose exponent they do not
        // agree comes out as a summary for none, and is crossed out for the chunk)
        WideSum<S> out = s_sub[t * WIDE_SUBS];
        for (uint32_t w = 1; w < WIDE_SUBS; w++) out = wide_then<T>(out, s_sub[t * WIDE_SUBS + w]);
        ws.maps[c * WIDE_K + t] = out;
        const bool none = out.of[0].e == WIDE_NO_E(S);
        if (none) atomicOr(&s_bad, 1u << t);
        if (t == 1) {
            // Is the walk likely to open this chunk?  Slot 1 of sub-chunk 0 is the exponent of the rough
            // sum in front of the chunk: the walk's own test on that sum, with a margin for how far the
            // exact chain may be from it.  (Only a hint: the walk loads hinted chunks ahead of time.)
            bool hint = none || (s_bad & 2u) != 0;
            if (!hint) {
                const S sa = (S)scalbn(a, G::MANT - (head.e0 + 1));
                const S margin = (S)1 << (G::MANT - 8);
                for (int p = 0; p < 2; p++) hint = hint || !wide_inside<T>(sa + out.of[p].lo, sa + out.of[p].hi, margin);
            }
            if (hint) atomicOr(&s_bad, WIDE_HINT);
        }
    }
    __syncthreads();
    head.bad = s_bad;
    if ((head.bad & WIDE_CHUNK_BITS) == WIDE_CHUNK_BITS) head.bad |= WIDE_HINT;
    if (t == 0) ws.heads[c] = head;
}

template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_maps(FilterIn in, void *ws_base, uint64_t nc, FilterLink link = FilterLink{ nullptr, nullptr })
{
    wide_maps_body<OP, T>(in, ws_base, nc, link);
}

// ---- pass 3: the chunks in order ---------------------------------------------------------------------
// acc's exponent if acc is a normal number
template <typename T> __device__ __forceinline__ bool wide_exponent(T acc, int &e)
{
    typedef Grid<T> G;
    typedef typename G::U U;
    const int aef = (int)(((U)to_bits<T>(acc) >> G::MANT) & (U)G::EMASK);
    e = aef - G::BIAS;
    return aef != 0 && aef != G::EMASK;
}

// the number of exponent e (a normal number's) and signed integer significand sum (2^MANT <= |sum| < 2^(MANT+1))
template <typename T> __device__ __forceinline__ T wide_number(int e, typename Grid<T>::S sum)
{
    typedef Grid<T> G;
    typedef typename G::U U;
    return from_bits<T>((uint64_t)(((U)(e + G::BIAS) << G::MANT) | ((U)(sum < 0 ? -sum : sum) & (((U)1 << G::MANT) - 1)) |
                                   (sum < 0 ? (U)1 << (sizeof(U) * 8 - 1) : (U)0)));
}

// lane l's value, l the same in every lane: the answer is known to be the same in every lane too
// (a shuffle's is not, and everything computed from it -- the sum, the walk's branches -- would be
// carried out per lane)
__device__ __forceinline__ int32_t wide_of_lane(int32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, (int)l); }
__device__ __forceinline__ int64_t wide_of_lane(int64_t v, uint32_t l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, (int)l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), (int)l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// one side out of LDS, in 128-bit reads
__device__ __forceinline__ WideSide<int32_t> wide_side(const WideSide<int32_t> *p)
{
    typedef int32_t v4 __attribute__((ext_vector_type(4)));
    const v4 v = *(const v4 *)p;
    WideSide<int32_t> r;
    r.lo = v.x; r.hi = v.y; r.step = v.z; r.e = v.w;
    return r;
}
__device__ __forceinline__ WideSide<int64_t> wide_side(const WideSide<int64_t> *p)
{
    typedef int64_t v2 __attribute__((ext_vector_type(2)));
    const v2 a = ((const v2 *)p)[0], b = ((const v2 *)p)[1];
    WideSide<int64_t> r;
    r.lo = a.x; r.hi = a.y; r.step = b.x; r.e = b.y;
    return r;
}

// The three candidates' summaries of one run of packets (w, w + stride, w + 2 * stride) against the
// actual sum: if one of them is for acc's exponent and every exact partial sum stays inside acc's
// binade, acc takes the whole step and the answer is true.  All three are read at once (a lane each),
// before it is known which one counts: one trip to LDS on the chain from sum to sum.
template <typename T> __device__ __forceinline__ bool wide_take(T &acc, const WideSum<typename Grid<T>::S> *w, uint32_t stride)
{
    typedef Grid<T> G;
    typedef typename G::U U;
    typedef typename G::S S;
    static_assert(WIDE_K == 3, "three candidates are read here");
    int e;
    if (!wide_exponent<T>(acc, e)) return false;
    const U ab = (U)to_bits<T>(acc);
    const U afrac = (ab & (((U)1 << G::MANT) - 1)) | ((U)1 << G::MANT);
    const S s0 = (ab >> (sizeof(U) * 8 - 1)) ? -(S)afrac : (S)afrac;
    const int p = (int)(s0 & 1);
    // lane k reads candidate k (the lanes from WIDE_K on read the last one again); the lane whose side
    // is for acc's exponent hands it to everybody
    const uint32_t k = threadIdx.x & 63u;
    const WideSide<S> c = wide_side(&w[(k < WIDE_K ? k : WIDE_K - 1u) * stride].of[p]);
    const uint64_t is = __ballot(c.e == (S)e);
    if (!is) return false;
    const uint32_t from = (uint32_t)__ffsll((unsigned long long)is) - 1u;
    const S lo = wide_of_lane(c.lo, from), hi = wide_of_lane(c.hi, from), step = wide_of_lane(c.step, from);
    if (!wide_inside<T>(s0 + lo, s0 + hi, (S)0)) return false;
    acc = wide_number<T>(e, s0 + step);
    return true;
}

// A run of chunks at once (w: the first one's summaries, WIDE_K per chunk; avail: how many there are):
// lane l takes the summary of chunk l that is for acc's exponent, the lanes' summaries are put together
// by an inclusive scan (wide_then is associative; a chunk without a summary for the exponent stands as
// one that fails), and lane l's result says whether chunks 0 .. l all go in one step each from the
// sum as it is.  acc takes the steps of the first chunks that do; the answer is how many those were
// (the chunk behind them, if there is one, failed its test).  About four single steps' worth of
// instructions, whatever the run's length.
template <typename T> __device__ __forceinline__ uint32_t wide_take_many(T &acc, const WideSum<typename Grid<T>::S> *w, uint32_t avail, uint32_t lane)
{
    typedef Grid<T> G;
    typedef typename G::U U;
    typedef typename G::S S;
    int e;
    if (!wide_exponent<T>(acc, e)) return 0;
    const U ab = (U)to_bits<T>(acc);
    const U afrac = (ab & (((U)1 << G::MANT) - 1)) | ((U)1 << G::MANT);
    const S s0 = (ab >> (sizeof(U) * 8 - 1)) ? -(S)afrac : (S)afrac;
    const bool odd = (s0 & 1) != 0;
    const S FAILS = (S)1 << (G::MANT + 3);                      // (wide_clamp's bound: inside no binade)
    // (plain variables, no structures: a structure copied under a condition ends up in scratch memory)
    S lo0 = -FAILS, hi0 = FAILS, st0 = 0, lo1 = -FAILS, hi1 = FAILS, st1 = 0;      // in front of an even / an odd S
    if (lane < avail) {
        const WideSum<S> *c = w + lane * WIDE_K;
#pragma unroll
        for (int k = 0; k < WIDE_K; k++) {
            const WideSide<S> even = wide_side(&c[k].of[0]), odd_side = wide_side(&c[k].of[1]);
            const bool is = even.e == (S)e;
            lo0 = is ? even.lo : lo0; hi0 = is ? even.hi : hi0; st0 = is ? even.step : st0;
            lo1 = is ? odd_side.lo : lo1; hi1 = is ? odd_side.hi : hi1; st1 = is ? odd_side.step : st1;
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        // the run of chunks ending d lanes down (a), this lane's run behind it (b): as wide_then
        const S alo0 = grid_up(lo0, d), ahi0 = grid_up(hi0, d), ast0 = grid_up(st0, d);
        const S alo1 = grid_up(lo1, d), ahi1 = grid_up(hi1, d), ast1 = grid_up(st1, d);
        const bool odd0 = (ast0 & 1) != 0, odd1 = (ast1 & 1) == 0;      // the parity of S + a's step, S even / odd
        const S bl0 = ast0 + (odd0 ? lo1 : lo0), bh0 = ast0 + (odd0 ? hi1 : hi0), bs0 = ast0 + (odd0 ? st1 : st0);
        const S bl1 = ast1 + (odd1 ? lo1 : lo0), bh1 = ast1 + (odd1 ? hi1 : hi0), bs1 = ast1 + (odd1 ? st1 : st0);
        if ((int)lane >= d) {
            lo0 = wide_clamp<T>(alo0 < bl0 ? alo0 : bl0); hi0 = wide_clamp<T>(ahi0 > bh0 ? ahi0 : bh0); st0 = bs0;
            lo1 = wide_clamp<T>(alo1 < bl1 ? alo1 : bl1); hi1 = wide_clamp<T>(ahi1 > bh1 ? ahi1 : bh1); st1 = bs1;
        }
    }
    const S lo = odd ? lo1 : lo0, hi = odd ? hi1 : hi0, step = odd ? st1 : st0;
    const uint64_t ok = __ballot(wide_inside<T>(s0 + lo, s0 + hi, (S)0));
    const uint32_t took = ~ok ? (uint32_t)__ffsll((unsigned long long)~ok) - 1u : 64u;
    if (took) acc = wide_number<T>(e, s0 + wide_of_lane(step, took - 1u));
    return took;
}

// A chunk the walk could not take in one step (one wavefront; the chunk's packets in s_val, its
// sub-chunk summaries in s_sub): sub-chunk by sub-chunk, each in one step where its test holds for the
// sum in front of it, the plain way otherwise.  Sub-chunks that no sum can take in one step (every
// candidate crossed out, or marked by the second pass as not worth a look) are added in one go with their
// neighbours.
template <typename T> __device__ __forceinline__ T wide_open(T acc, const WideHead head, const WideSum<typename Grid<T>::S> *s_sub, const T *s_val, uint32_t len)
{
    uint32_t forced = (1u << WIDE_SUBS) - 1u;                   // bit w: sub-chunk w is crossed out for every candidate ...
#pragma unroll
    for (int cand = 0; cand < WIDE_K; cand++) forced &= head.bad >> (4 + cand * WIDE_SUBS);
    forced |= (head.bad >> WIDE_SKIP_SHIFT) & ((1u << WIDE_SUBS) - 1u);        // ... or not expected to go in one step
    uint32_t w = 0;
    while (w * 64u * WIDE_EPL < len) {
        const uint32_t at = w * 64u * WIDE_EPL;
        if (!((forced >> w) & 1u) && len - at >= 64u * WIDE_EPL) {
            if (wide_take<T>(acc, &s_sub[w], WIDE_SUBS)) { w++; continue; }
            acc = plain_fold<T>(acc, s_val + at, 64u * WIDE_EPL);
            w++;
            continue;
        }
        uint32_t w1 = w + 1;
        while (w1 < WIDE_SUBS && ((forced >> w1) & 1u)) w1++;
        const uint32_t end = w1 * 64u * WIDE_EPL < len ? w1 * 64u * WIDE_EPL : len;
        acc = plain_fold<T>(acc, s_val + at, end - at);
        w = w1;
    }
    return acc;
}

// a value that is the same in every lane, said so
__device__ __forceinline__ uint64_t wide_uniform(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// the first hinted chunk of a stage (a bit per chunk in h0, h1) at or behind `from`; m if there is none
__device__ __forceinline__ uint32_t wide_next_hint(uint64_t h0, uint64_t h1, uint32_t from, uint32_t m)
{
    if (from < 64) h0 &= ~(uint64_t)0 << from;
    else { h0 = 0; h1 = from < 128 ? h1 & (~(uint64_t)0 << (from - 64)) : 0; }
    return h0 ? (uint32_t)__ffsll((unsigned long long)h0) - 1u : (h1 ? 63u + (uint32_t)__ffsll((unsigned long long)h1) : m);
}

template <int OP, typename T>
__device__ __forceinline__ void wide_walk_body(const FilterIn &in, void *ws_base, uint64_t nc, uint8_t *result, int32_t *result_ok,
                                               const FilterLink &link)
{
    typedef Grid<T> G;
    typedef typename G::S S;
    constexpr uint32_t SUB_DW = (uint32_t)(sizeof(WideSum<S>) * WIDE_K * WIDE_SUBS / 4);     // a chunk's sub-chunk summaries, in 32-bit words
    static_assert(SUB_DW <= WIDE_TPB && WIDE_STAGE <= WIDE_TPB && WIDE_STAGE == 128, "one word / one head per thread, two 64-bit hint masks");
    // Two of everything an opened chunk needs: while one wavefront adds up the chunk in one set, the
    // others may already stage the next chunk into the other.
    __shared__ __align__(16) T s_val[2][WIDE_CHUNK + 64];       // (slack: the plain fold reads a batch ahead)
    __shared__ WideHead s_head[WIDE_STAGE];
    __shared__ WideSum<S> s_map[WIDE_STAGE * WIDE_K];
    __shared__ __align__(16) WideSum<S> s_sub[2][WIDE_K * WIDE_SUBS];
    __shared__ uint64_t s_hint[WIDE_STAGE / 64], s_dead[WIDE_STAGE / 64];
    __shared__ T s_acc[2];
    __shared__ int32_t s_cnt;
    WideWs<T> ws(ws_base, nc);
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) s_cnt = 0;
    T acc = chain_start<T>(link);                               // the same in every lane
    uint32_t cur = 0;                                           // the set the chunk being added is in
    uint64_t staged_of = ~(uint64_t)0;                          // the chunk already staged in the other set
#ifdef WIDE_DEBUG
    uint32_t nplain = 0, nmiss = 0;
#endif
    // A chunk's packets and sub-chunk summaries, loaded ahead of the walk: an opened chunk costs little
    // once its packets are at hand, and most of a microsecond of waiting if they are not.
    // Wavefront 0, the one that adds, takes no part in loading and staging: every instruction it issues
    // is time on the chain.
    constexpr uint32_t LOADERS = WIDE_TPB - 64, LEPL = (WIDE_CHUNK + LOADERS - 1) / LOADERS;
    static_assert(SUB_DW <= LOADERS, "one word of the sub-chunk summaries per loading thread");
    const uint32_t u = t - 64u;                                 // (loading threads: 0 .. LOADERS - 1)
    WideRaw<T> ahead[LEPL];
    uint32_t sub_ahead = 0;
    uint64_t ahead_of = ~(uint64_t)0;
#define WIDE_FETCH(cc_) do { \
        const uint64_t fc_ = (cc_), f0_ = fc_ * WIDE_CHUNK; \
        if (wave != 0) { \
            _Pragma("unroll") for (uint32_t k = 0; k < LEPL; k++) { \
                const uint32_t at_ = k * LOADERS + u; \
                const uint64_t i = f0_ + (at_ < WIDE_CHUNK ? at_ : WIDE_CHUNK - 1u); \
                ahead[k] = wide_load<OP, T>(in, i < in.n ? i : in.n - 1); \
            } \
            if (u < SUB_DW) sub_ahead = ((const uint32_t *)(ws.subs + fc_ * (WIDE_K * WIDE_SUBS)))[u]; \
        } \
        ahead_of = fc_; \
    } while (0)
    for (uint64_t c0 = 0; c0 < nc; c0 += WIDE_STAGE) {
        const uint32_t m = (uint32_t)(nc - c0 < WIDE_STAGE ? nc - c0 : WIDE_STAGE);
        __syncthreads();
        for (uint32_t j = t; j < m; j += WIDE_TPB) s_head[j] = ws.heads[c0 + j];
        for (uint32_t j = t; j < m * WIDE_K; j += WIDE_TPB) s_map[j] = ws.maps[c0 * WIDE_K + j];
        if (t < WIDE_STAGE) {
            const uint32_t bad = ws.heads[c0 + (t < m ? t : 0)].bad;
            const uint64_t b = __ballot(t < m && (bad & WIDE_HINT) != 0);
            const uint64_t d = __ballot(t < m && (bad & WIDE_CHUNK_BITS) == WIDE_CHUNK_BITS);
            if (lane == 0) { s_hint[wave] = b; s_dead[wave] = d; }
        }
        __syncthreads();
        // (kept in scalar registers: the walk tests a bit of these per chunk)
        const uint64_t hint0 = wide_uniform(s_hint[0]), hint1 = wide_uniform(s_hint[1]);
        const uint64_t dead0 = wide_uniform(s_dead[0]), dead1 = wide_uniform(s_dead[1]);       // chunks without a summary for any exponent: not worth a look
        uint32_t calm = wide_next_hint(hint0, hint1, 0u, m);       // the next chunk that is likely to be opened
        if (calm < m && ahead_of != c0 + calm) WIDE_FETCH(c0 + calm);
        for (uint32_t j = 0; j < m; j++) {
            // (uniform: acc is the same in every lane)
            if (j > calm) calm = wide_next_hint(hint0, hint1, j, m);
            if (((j < 64u ? dead0 >> j : dead1 >> (j - 64u)) & 1u) != 0) { }        // (straight to the opening)
            else if (calm > j && calm - j >= WIDE_MANY_MIN) {
                const uint32_t took = wide_take_many<T>(acc, &s_map[j * WIDE_K], m - j, lane);
                j += took;
                if (took == 64u || j >= m) { j--; continue; }
            }
            else if (wide_take<T>(acc, &s_map[j * WIDE_K], 1u)) continue;
            const WideHead head = s_head[j];
            // Opened: staged by everybody, gone through by one wavefront.
            const uint64_t cc = c0 + j, first = cc * WIDE_CHUNK;
            const uint32_t len = (uint32_t)(in.n - first < WIDE_CHUNK ? in.n - first : WIDE_CHUNK);
#ifdef WIDE_DEBUG
            nplain++;
            if (ahead_of != cc) nmiss++;
#endif
            // Into LDS, unless it was staged while the chunk before was being added (ordered by the
            // barrier behind that chunk).  No barrier in front of the staging: the set written here was
            // last read two opened chunks ago, in front of the barrier that closed that chunk.
#define WIDE_STAGE_AHEAD(set_) do { \
        if (wave != 0) { \
            _Pragma("unroll") for (uint32_t k = 0; k < LEPL; k++) { \
                const uint32_t at = k * LOADERS + u; \
                if (at < WIDE_CHUNK) s_val[set_][at] = wide_value_of<OP, T>(in, ahead[k]); \
            } \
            if (u < SUB_DW) ((uint32_t *)s_sub[set_])[u] = sub_ahead; \
        } \
    } while (0)
            cur ^= 1u;
            if (staged_of != cc) {
                if (ahead_of != cc) WIDE_FETCH(cc);
                WIDE_STAGE_AHEAD(cur);
                __syncthreads();
            }
            staged_of = ~(uint64_t)0;
            // What to load next, before the adding starts: the next hinted chunk if the hints are holding
            // (this chunk was one), the chunk behind this one if not (runs of opened chunks are the rule
            // where the rough sums mislead).  The chunk right behind this one is staged as well, under the
            // additions of this one: in a run of opened chunks the adding wavefront goes from chunk to
            // chunk with one barrier in between.
            {
                uint32_t nx = j + 1;
                if (head.bad & WIDE_HINT) nx = wide_next_hint(hint0, hint1, j + 1, m);
                if (nx < m) {
                    WIDE_FETCH(c0 + nx);
                    if (nx == j + 1) {
                        WIDE_STAGE_AHEAD(cur ^ 1u);
                        staged_of = c0 + nx;
                    }
                }
            }
            if (wave == 0) {
                // (values behind len are staged but not added)
                // Looking at a sub-chunk's summaries costs what some 90 additions cost, and saves 256 if
                // the sub-chunk goes in one step: the second pass has marked the sub-chunks that will
                // probably not (a running balance of looks that paid, kept by the walk on top of those
                // marks, made no measurable difference).
                // Every sub-chunk crossed out for every candidate, or not expected to go in one step:
                const bool hopeless = (((head.bad >> 4) & (head.bad >> (4 + WIDE_SUBS)) & (head.bad >> (4 + 2 * WIDE_SUBS))) | (head.bad >> WIDE_SKIP_SHIFT)) % (1u << WIDE_SUBS)
                                      == (1u << WIDE_SUBS) - 1u;
                if (hopeless) acc = plain_fold<T>(acc, s_val[cur], len);
                else acc = wide_open<T>(acc, head, s_sub[cur], s_val[cur], len);
                if (lane == 0) s_acc[cur] = acc;
            }
            __syncthreads();
            acc = s_acc[cur];                                   // (two of these as well: the next write of this one is two barriers away)
        }
    }
#undef WIDE_FETCH
#undef WIDE_STAGE_AHEAD
#ifdef WIDE_DEBUG
    if (t == 0) printf("wide walk: %u of %llu chunks opened, %u of them not loaded ahead\n", nplain, (unsigned long long)nc, nmiss);
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
    if (t == 0) chain_finish<OP, T>(link, in.n, acc, s_cnt, result, result_ok);
}

template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_walk(FilterIn in, void *ws_base, uint64_t nc, uint8_t *result, int32_t *result_ok,
                                                        FilterLink link = FilterLink{ nullptr, nullptr })
{
    wide_walk_body<OP, T>(in, ws_base, nc, result, result_ok, link);
}

// ---- several columns at once ---------------------------------------------------------------------------
// The same three passes over strided columns side by side (blockIdx.y picks the column).  Used for
// performance-data packets whose arrays all have one length k and lie end to end: element y of every
// packet is a column (values at base + (first + y) * sizeof(T), k * sizeof(T) apart), and the elements'
// chains run next to each other, one walking workgroup each.  What the host does not know is read on the
// device: shape[0] != 0 if the wave has that form (0 switches the launch off), shape[1] = columns, shape[2] =
// first (the element index of packet 0's array in the values column), shape[6] = elements from packet to packet.  Column y has its summaries at
// ws_base + y * ws_stride and its result at result + 16 y.
// fronts / nodes (both or neither): column y continues from the node fronts[y] and leaves the node nodes[y]
// instead of a result (a packet range of a wave, FilterLink).
// dense / row_counts: the dense copy of a wave whose arrays are not of one length (shape[7] = 1 puts the
// columns there, shape[6] elements from packet to packet) and the packets' own array lengths.
struct WideMulti { const uint64_t *shape; uint64_t ws_stride; const Partial *fronts; Partial *nodes; const uint8_t *dense; const uint64_t *row_counts; };

__device__ __forceinline__ FilterLink wide_multi_link(const WideMulti &m)
{
    return FilterLink{ m.fronts ? m.fronts + blockIdx.y : nullptr, m.nodes ? m.nodes + blockIdx.y : nullptr };
}

template <typename T> __device__ __forceinline__ bool wide_multi_column(const WideMulti &m, FilterIn &in)
{
    const uint64_t on = m.shape[0], cols = m.shape[1], first = m.shape[2];
    if (!on || blockIdx.y >= cols) return false;
    if (m.shape[7]) in.values = m.dense;
    in.values += (first + blockIdx.y) * sizeof(T);
    in.value_stride = m.shape[6] * sizeof(T);
    return true;
}

template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_sums_multi(FilterIn in, void *ws_base, uint64_t nc, WideMulti m)
{
    if (!wide_multi_column<T>(m, in)) return;
    wide_sums_body<OP, T>(in, (uint8_t *)ws_base + blockIdx.y * m.ws_stride, nc);
}

template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_maps_multi(FilterIn in, void *ws_base, uint64_t nc, WideMulti m)
{
    if (!wide_multi_column<T>(m, in)) return;
    wide_maps_body<OP, T>(in, (uint8_t *)ws_base + blockIdx.y * m.ws_stride, nc, wide_multi_link(m));
}

template <int OP, typename T>
__global__ __launch_bounds__(WIDE_TPB) void k_wide_walk_multi(FilterIn in, void *ws_base, uint64_t nc, uint8_t *result, int32_t *result_ok, WideMulti m)
{
    if (!wide_multi_column<T>(m, in)) return;
    wide_walk_body<OP, T>(in, (uint8_t *)ws_base + blockIdx.y * m.ws_stride, nc, result + 16u * blockIdx.y, result_ok + blockIdx.y, wide_multi_link(m));
}
