// tfilter_Sum / tfilter_Avg over float and double, ONE reference call over the whole wave, without
// walking the chain of additions one link at a time.
//
// Reference: result = result + x_i for every packet in arrival order, in the element's own type
// (src/FilterDefinitions.C:101-149, 517-561, sum() at :1955-1960).  Every addition rounds, so the
// result is the chain's and nothing else's; k_filter_flat walks it at one dependent add per
// packet.  But between two changes of its exponent the running sum lives on a fixed grid: with
// 2^e <= |s| < 2^(e+1) and u = 2^(e - MANT), s = S * u for an integer S, and as long as the exact
// sum s + x stays inside the same binade, round-to-nearest-even of s + x is u * rne(S + x / u):
//   x / u = I + F with I = floor(x / u) and 0 <= F < 1 is exact integer arithmetic on x's bits, and
//   rne(S + I + F) = S + I            if F < 1/2
//                    S + I + 1        if F > 1/2
//                    S + I + (S + I is odd ? 1 : 0)   if F = 1/2  (ties go to the even neighbour).
// So a packet is a map on S that depends on S only through its parity: two integers (the step for
// an even S, the step for an odd S).  Such maps compose associatively -- (a then b)[p] =
// a[p] + b[(p + a[p]) & 1] -- and a workgroup of 1,024 lanes scans 4,096 of them in about three
// microseconds, where the chain of additions needs fifteen.  Every
// lane then checks, for its own packets, that the exact sum really stayed inside the binade (an
// integer range test on S + I, conservative at both ends); the scan's results are taken up to the
// first packet where it did not, that packet is added the ordinary way (a hardware add, which moves
// the sum to its new binade), and the next round starts from there.  Where rounds keep ending early
// (sums hovering around a power of two, cancellations, zeros, infinities, NaNs) stretches of the
// wave are folded the plain way, so the worst case stays near the plain chain (measured: at most
// 13 % over it; 2.2 to 2.9 times faster where the sum grows).
//
// The bits are those of the serial fold for every input: the tests compare both kernels and the
// CPU oracle on ordinary, tie-heavy, cancelling, huge-range, subnormal and non-finite data.
#pragma once

#include "filter.h"

// One workgroup walks the chain, so it is made as wide as a workgroup gets: sixteen wavefronts, four
// per SIMD -- the integer work per packet (some seventy instructions) is spread over 1,024 lanes and
// the wavefronts of a SIMD fill each other's latencies.
#ifndef SCAN_TPB
#define SCAN_TPB 1024
#endif
#ifndef SCAN_EPL_F
#define SCAN_EPL_F 4                                // packets per lane and round: float (8 measured: no faster, dearer when a round fails)
#endif
#ifndef SCAN_EPL_D
#define SCAN_EPL_D 4                                // double (the window holds half as many)
#endif
#define SCAN_WAVES (SCAN_TPB / 64)
#define SCAN_WIN_ELEMS(T) ((61440u / sizeof(T)) / SCAN_TPB * SCAN_TPB)      // LDS window of staged values: 60 KiB
__device__ __forceinline__ int32_t grid_lane(int32_t v, int l) { return __shfl(v, l, 64); }
__device__ __forceinline__ int64_t grid_lane(int64_t v, int l) { return (int64_t)__shfl((unsigned long long)v, l, 64); }
// A round costs about what the plain chain needs for this many packets: a round that got no
// further than this has not paid.
#define SCAN_PAYS_OF(T) (sizeof(T) == 4 ? 1280u : 1792u)       // (measured: round, crossing packet and hand-over)
#define SCAN_PLAIN_MIN 1024u
#define SCAN_PLAIN_MAX 32768u

template <typename T> struct Grid;
template <> struct Grid<float> {
    typedef uint32_t U; typedef int32_t S;
    static constexpr int MANT = 23, EMASK = 0xff, BIAS = 127, MAXSH = 31;
};
template <> struct Grid<double> {
    typedef uint64_t U; typedef int64_t S;
    static constexpr int MANT = 52, EMASK = 0x7ff, BIAS = 1023, MAXSH = 63;
};

// x = m * 2^(ex - MANT) with a signed integer m; false for infinities and NaNs.
template <typename T> __device__ __forceinline__ bool grid_split(T x, typename Grid<T>::S &m, int &ex)
{
    typedef Grid<T> G;
    typedef typename G::U U;
    typedef typename G::S S;
    const U b = (U)to_bits<T>(x);
    const int ef = (int)((b >> G::MANT) & (U)G::EMASK);
    U frac = b & (((U)1 << G::MANT) - 1);
    if (ef == G::EMASK) return false;
    if (ef == 0) ex = 1 - G::BIAS;                  // zero, subnormal
    else { frac |= (U)1 << G::MANT; ex = ef - G::BIAS; }
    m = (b >> (sizeof(U) * 8 - 1)) ? -(S)frac : (S)frac;
    return true;
}

// A packet as a map on the integer S: step[p] for a sum of parity p; cross = the map does not apply
// (the sum leaves the binade for sure, or x is not finite).
template <typename T> struct GridStep {
    typename Grid<T>::S floor_i;                    // I
    typename Grid<T>::S step[2];
    bool cross;
};

template <typename T> __device__ __forceinline__ GridStep<T> grid_step(T x, int e)
{
    typedef Grid<T> G;
    typedef typename G::U U;
    typedef typename G::S S;
    GridStep<T> g;
    g.floor_i = 0; g.step[0] = g.step[1] = 0; g.cross = true;
    S m;
    int ex;
    if (!grid_split<T>(x, m, ex)) return g;
    const int k = ex - e;
    if (k >= 2) return g;                           // |x| >= 4 * 2^e > |s| + 2^(e+1): out of the binade
    g.cross = false;
    if (k >= 0) {
        g.floor_i = (S)((U)m << k);
        g.step[0] = g.step[1] = g.floor_i;
        return g;
    }
    const int sh = -k < G::MAXSH ? -k : G::MAXSH;   // (beyond: |x / u| < 1/4 either way, and the shifts below say so)
    const S i = m >> sh;                            // floor
    const U rem = (U)m & (((U)1 << sh) - 1), half = (U)1 << (sh - 1);
    g.floor_i = i;
    if (rem < half) g.step[0] = g.step[1] = i;
    else if (rem > half) g.step[0] = g.step[1] = i + 1;
    else { g.step[0] = i + (i & 1); g.step[1] = i + ((i + 1) & 1); }      // tie: to the even neighbour of S + i
    return g;
}

// The exact sum (S + I + F) * u stays inside the binade, both ends conservative (nothing that could
// round to the binade's upper end, nothing below its lower end): the rounding rule above holds.
template <typename T> __device__ __forceinline__ bool grid_inside(typename Grid<T>::S tf)
{
    typedef typename Grid<T>::S S;
    const S lo = (S)1 << Grid<T>::MANT, hi = ((S)1 << (Grid<T>::MANT + 1)) - 2;
    return (tf >= lo && tf <= hi) || (tf <= -lo - 1 && tf >= -hi);
}

template <typename S> struct GridMap { S a0, a1; };      // the step for an even / an odd S

template <typename S> __device__ __forceinline__ GridMap<S> grid_then(const GridMap<S> &a, const GridMap<S> &b)
{
    GridMap<S> c;
    c.a0 = a.a0 + ((a.a0 & 1) ? b.a1 : b.a0);
    c.a1 = a.a1 + (((a.a1 + 1) & 1) ? b.a1 : b.a0);
    return c;
}

__device__ __forceinline__ int32_t grid_up(int32_t v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ int64_t grid_up(int64_t v, int d) { return (int64_t)__shfl_up((unsigned long long)v, d, 64); }

template <typename S> __device__ __forceinline__ GridMap<S> grid_shfl_up(const GridMap<S> &f, int d)
{
    GridMap<S> o;
    o.a0 = grid_up(f.a0, d);
    o.a1 = grid_up(f.a1, d);
    return o;
}

// The plain chain over r staged values from s on (any alignment): the first few one by one up to a
// 16-byte boundary, then 128-bit LDS reads, one batch of sixteen values in flight under the additions
// of the batch before (as k_filter_flat does).  s[r .. r+31] may be read (inside the window's slack).
template <typename T> __device__ __forceinline__ T plain_fold(T acc, const T *s, uint32_t r)
{
    constexpr uint32_t PERQ = 16 / sizeof(T);
    uint32_t j = 0;
    while (j < r && (((uintptr_t)(s + j)) & 15u)) acc = acc + s[j++];
    if (j + 32 <= r) {
        T a[16], b[16];
        lds_read16(a, s + j);
        for (; j + 32 <= r; j += 32) {
            lds_read16(b, s + j + 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; k++) acc = acc + a[k];
            lds_read16(a, s + j + 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; k++) acc = acc + b[k];
        }
    }
    (void)PERQ;
    for (; j < r; j++) acc = acc + s[j];
    return acc;
}

template <int OP, typename T>
__global__ __launch_bounds__(SCAN_TPB) void k_filter_flat_scan(FilterIn in, uint8_t *result, int32_t *result_ok,
                                                               FilterLink link = FilterLink{ nullptr, nullptr })
{
    typedef Grid<T> G;
    typedef typename G::U U;
    typedef typename G::S S;
    typedef typename UIntOf<sizeof(T)>::type RawU;
    constexpr uint32_t WIN = SCAN_WIN_ELEMS(T);
    constexpr int SCAN_EPL = sizeof(T) == 4 ? SCAN_EPL_F : SCAN_EPL_D;
    constexpr uint32_t SCAN_ROUND = SCAN_TPB * SCAN_EPL;      // packets per round
    constexpr uint32_t SCAN_PAYS = SCAN_PAYS_OF(T);
    constexpr int PER = WIN / SCAN_TPB;
    static_assert(WIN % SCAN_TPB == 0 && WIN >= SCAN_ROUND, "a window holds whole rounds");
    constexpr bool COUNTS = OP == MRPDR_AVG;
    __shared__ __align__(16) T win[WIN + 64];           // (slack: the plain fold reads one batch ahead)
    __shared__ GridMap<S> s_wave[SCAN_WAVES];
    __shared__ uint32_t s_bad[SCAN_WAVES];
    __shared__ S s_sum[SCAN_WAVES];
    __shared__ T s_acc[2];                              // (two, used in turn: a wavefront may still be reading one
                                                        //  when wavefront 0, one barrier on, writes the next sum)
    __shared__ int32_t s_cnt;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) s_cnt = 0;
    // acc, pos and the back-off state are the same in every lane: every lane computes them
    T acc = chain_start<T>(link);
    uint64_t pos = 0, wbase = 0, wend = 0;
    // length of the next plain stretch.  The first one is a full one: the sum starts at zero, and a
    // sum that grows from nothing changes its binade every few packets at first
    uint32_t plain = SCAN_PLAIN_MIN, miss = 0;
    int32_t credit = 0;
    uint32_t owed = 0;
    int32_t mycnt = 0;
    uint32_t flip = 0;
    while (pos < in.n) {
        if (pos == wend) {
            // the next window of values (times count for AVG), in arrival order.  (Its loads issued a
            // window ahead, into registers, was measured: 2 % for float, nothing for double, whose
            // kernel is at its register limit.)
            __syncthreads();
            wbase = pos;
            wend = wbase + WIN < in.n ? wbase + WIN : in.n;
            RawU raw[PER];
            uint32_t rawc[PER];
            uint8_t ord[PER];
            const uint64_t last = in.n - 1;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                uint64_t i = wbase + (uint64_t)k * SCAN_TPB + t;
                i = i < last ? i : last;
                raw[k] = ld_u<RawU>(in.values + i * in.value_stride);
                rawc[k] = COUNTS ? ld_u<uint32_t>(in.counts + i * in.count_stride) : 1u;
                ord[k] = in.orders ? in.orders[i] : (uint8_t)(in.swap != 0);
            }
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const uint64_t i = wbase + (uint64_t)k * SCAN_TPB + t;
                RawU r = raw[k];
                uint32_t c = rawc[k];
                if (ord[k]) { r = bswap_u(r); c = bswap_u(c); }
                T v = from_bits<T>((uint64_t)r);
                if (COUNTS) v = mul_cnt<T>(v, (int32_t)c);
                if (i < in.n) { if (COUNTS) mycnt += (int32_t)c; } else v = (T)0;
                win[k * SCAN_TPB + t] = v;
            }
            __syncthreads();
        }
        const T *s = win + (uint32_t)(pos - wbase);
        const uint32_t avail = (uint32_t)(wend - pos);
        uint32_t run;                               // packets to fold the plain way after this round
        const U ab = (U)to_bits<T>(acc);
        const int aef = (int)((ab >> G::MANT) & (U)G::EMASK);
        if (owed) {
            // the rest of a plain stretch that the window before this one cut short
            run = owed < avail ? owed : avail;
            owed -= run;
        } else if (aef != 0 && aef != G::EMASK) {
            // ---- one round of the scan: up to SCAN_ROUND packets from pos on ----
            const int e = aef - G::BIAS;
            const U afrac = (ab & (((U)1 << G::MANT) - 1)) | ((U)1 << G::MANT);
            const S s0 = (ab >> (sizeof(U) * 8 - 1)) ? -(S)afrac : (S)afrac;
            const uint32_t len = avail < SCAN_ROUND ? avail : SCAN_ROUND;
            GridStep<T> st[SCAN_EPL];
            GridMap<S> f;
            f.a0 = f.a1 = 0;
#pragma unroll
            for (int k = 0; k < SCAN_EPL; k++) {
                const uint32_t j = t * SCAN_EPL + (uint32_t)k;
                st[k] = grid_step<T>(s[j < len ? j : 0], e);
                if (j >= len) st[k].cross = true;           // behind the end: the round stops there
                GridMap<S> g;
                g.a0 = st[k].step[0]; g.a1 = st[k].step[1];
                f = grid_then(f, g);
            }
            // inclusive scan of the lanes' maps over the wavefront, then over the wavefronts in front
            GridMap<S> inc = f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const GridMap<S> o = grid_shfl_up(inc, d);
                if ((int)lane >= d) inc = grid_then(o, inc);
            }
            if (lane == 63) s_wave[wave] = inc;
            GridMap<S> exc = grid_shfl_up(inc, 1);
            if (lane == 0) exc.a0 = exc.a1 = 0;
            __syncthreads();
            // the wavefronts in front of this one: every wavefront scans the sixteen totals itself
            // (lane w holds wavefront w's; four shuffle steps instead of up to fifteen compositions)
            {
                GridMap<S> tot;
                tot.a0 = tot.a1 = 0;
                if (lane < SCAN_WAVES) tot = s_wave[lane];
#pragma unroll
                for (int d = 1; d < SCAN_WAVES; d <<= 1) {
                    const GridMap<S> o = grid_shfl_up(tot, d);
                    if ((int)lane >= d) tot = grid_then(o, tot);
                }
                GridMap<S> pre;                                   // inclusive total of wavefront wave - 1
                pre.a0 = grid_lane(tot.a0, wave ? (int)wave - 1 : 0);
                pre.a1 = grid_lane(tot.a1, wave ? (int)wave - 1 : 0);
                if (wave == 0) pre.a0 = pre.a1 = 0;
                exc = grid_then(pre, exc);
            }
            S p = s0 + ((s0 & 1) ? exc.a1 : exc.a0);
            // this lane's packets with the actual sums: the first one that leaves the binade
            uint32_t bad = SCAN_EPL;
            S p_bad = 0;
#pragma unroll
            for (int k = 0; k < SCAN_EPL; k++) {
                const bool fine = !st[k].cross && grid_inside<T>(p + st[k].floor_i);
                if (!fine && bad == SCAN_EPL) { bad = (uint32_t)k; p_bad = p; }
                p += (p & 1) ? st[k].step[1] : st[k].step[0];
            }
            uint32_t first = bad < SCAN_EPL ? t * SCAN_EPL + bad : SCAN_ROUND;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)first, d, 64);
                first = o < first ? o : first;
            }
            // the sum in front of that packet is the round's result (behind the last packet if none
            // is bad): within the wavefront it comes from the lane that found the packet
            const S p_first = first == SCAN_ROUND ? grid_lane(p, 63) : grid_lane(p_bad, (int)(first / SCAN_EPL) & 63);
            if (lane == 0) { s_bad[wave] = first; s_sum[wave] = p_first; }
            __syncthreads();
            // the first wavefront that found a bad packet decides; if none did, the last one's sum
            // behind its last packet is the round's
            S sum;
            {
                const uint32_t fw = lane < SCAN_WAVES ? s_bad[lane] : SCAN_ROUND;
                const S sw = s_sum[lane < SCAN_WAVES ? lane : 0];
                const unsigned long long found = __ballot(fw != SCAN_ROUND);
                const int w = found ? __builtin_ctzll(found) : SCAN_WAVES - 1;
                first = (uint32_t)__shfl((int)fw, w, 64);
                sum = grid_lane(sw, w);
            }
            // S * u, exact: |S| < 2^(MANT+1), and the result is a normal number of exponent e
            acc = from_bits<T>((uint64_t)(((U)(e + G::BIAS) << G::MANT) | ((U)(sum < 0 ? -sum : sum) & (((U)1 << G::MANT) - 1)) |
                                          (sum < 0 ? (U)1 << (sizeof(U) * 8 - 1) : (U)0)));
            pos += first;
            // Book-keeping in packets: a round costs about what the plain chain needs for SCAN_PAYS
            // packets.  credit is what the rounds so far have saved (kept within bounds, so that old
            // history fades), miss counts rounds in a row that did not pay.
            // (a round that the window's end cut short and that took all it was given says nothing)
            if (first < len || len == SCAN_ROUND) {
                miss = first >= SCAN_PAYS ? 0u : miss + 1u;
                credit += (int32_t)first - (int32_t)SCAN_PAYS;
                credit = credit < -4 * (int32_t)SCAN_PAYS ? -4 * (int32_t)SCAN_PAYS : credit > 8 * (int32_t)SCAN_PAYS ? 8 * (int32_t)SCAN_PAYS : credit;
            }
            if (first == len) continue;
            // A packet that takes the sum out of its binade: it is added the plain way.  One round
            // that has not paid is no verdict (behind the packet the sum is at the start of a new
            // binade -- for a growing sum the longest stretch it will ever have); two in a row are
            // followed by a plain stretch; and once the rounds have cost clearly more than they
            // saved (sums hovering around powers of two, wide-ranging magnitudes), long plain
            // stretches follow every round that does not pay, until the balance is back
            run = 1u;
            if (first < SCAN_PAYS) {
                if (credit <= -3 * (int32_t)SCAN_PAYS) run = SCAN_PLAIN_MAX;
                else if (miss >= 2u) run = SCAN_PLAIN_MIN;
            }
            s += first;
            if (run > avail - first) { owed = run - (avail - first); run = avail - first; }
        } else {
            // zero, subnormal, infinite or NaN: no grid to scan on
            run = plain;
            plain = plain < SCAN_PLAIN_MAX ? plain * 2u : plain;
            if (run > avail) { owed = run - avail; run = avail; }
        }
        // (one wavefront walks the chain, the others wait: sixteen copies of it would share four SIMDs)
        if (wave == 0) {
            acc = plain_fold<T>(acc, s, run);
            if (lane == 0) s_acc[flip] = acc;
        }
        __syncthreads();
        acc = s_acc[flip];
        flip ^= 1u;
        pos += run;
    }
    if (COUNTS) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) mycnt += __shfl_xor(mycnt, m, 64);
        __syncthreads();
        if (lane == 0) atomicAdd(&s_cnt, mycnt);
        __syncthreads();
    }
    if (t == 0) chain_finish<OP, T>(link, in.n, acc, s_cnt, result, result_ok);
}
