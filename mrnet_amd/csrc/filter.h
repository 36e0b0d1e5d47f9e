// tfilter_Sum / tfilter_Min / tfilter_Max / tfilter_Avg over a wave.
//
// Reference (src/FilterDefinitions.C): each filter is a left fold over the packets
// of a wave in the element's own C type --
//   Sum  result = 0; result = result + x_i                          (:100-149, :1925-1964)
//   Max  result = x_0; result = (result > x_i) ? result : x_i       (:237-286, :2090-2136)
//   Min  result = x_0; result = (result < x_i) ? result : x_i       (:374-424, :2042-2088)
//   Avg  packets are (value, count): result = sum of value*count computed in T,
//        n = sum of counts (int); output (result / n, n)            (:514-561, :1966-2040)
// 8- and 16-bit integers are promoted to int and truncated back; integer division
// truncates toward zero; float stays float.
//
// Folder applies exactly that fold to a short list of children.  The tree shape
// applies it level by level (lane: 4 packets; six pairings across the 64 lanes of a
// wavefront; the 4 wavefronts of a workgroup; all workgroups), the flat shape applies
// it once to the whole wave from one lane.
#pragma once

#include "common.h"

#define FILTER_LEAF 4                                   // packets per lane
#define FILTER_TILE (MRPDR_TPB * FILTER_LEAF)           // packets per workgroup
#define FLAT_TILE 2048                                  // packets staged in LDS per step

// Operators the library uses on its own to compute ONE reference filter call (the flat shape)
// level by level where that gives the same bits as the serial fold:
//   OP_AVGSUM  integer AVG: the sums of value*count (wrap-around, in T) and of the counts do not
//              depend on the order; the one division (src/FilterDefinitions.C:561) is done at the top
//   OP_MINR /  float MIN / MAX: the comparison keeps the new element whenever it is false, so a NaN
//   OP_MAXR    replaces the result and the element after it replaces the NaN -- a NaN restarts the
//              fold.  A node therefore carries "a NaN lies below me" (in cnt): joining two nodes, a
//              right child with the flag wins outright; otherwise the comparison decides, ties going
//              to the right child as they go to the later packet in the serial fold.
#define OP_AVGSUM 16
#define OP_MINR   17
#define OP_MAXR   18

template <int OP> struct OpTraits {
    static constexpr bool counted = OP == MRPDR_AVG || OP == OP_AVGSUM || OP == OP_MINR || OP == OP_MAXR;   // cnt travels with the node
    static constexpr bool pair_result = OP == MRPDR_AVG || OP == OP_AVGSUM;                                   // output packet is (value, count)
};

struct Partial {            // one node of the tree, as stored between launches
    uint64_t bits;          // value, low bytes
    int32_t  cnt;           // AVG: summed count
    int32_t  present;       // 0 = no packet below this node
};

template <typename T> __device__ inline T mul_cnt(T v, int32_t n);     // value * count, reference casts
template <typename T> __device__ inline T div_cnt(T v, int32_t n);     // value / count, reference casts

#define SMALL_INT_OPS(T)                                                              \
    template <> __device__ inline T mul_cnt<T>(T v, int32_t n) { return (T)((int32_t)v * n); } \
    template <> __device__ inline T div_cnt<T>(T v, int32_t n) { return (T)((int32_t)v / n); }
SMALL_INT_OPS(int8_t) SMALL_INT_OPS(uint8_t) SMALL_INT_OPS(int16_t) SMALL_INT_OPS(uint16_t)
#undef SMALL_INT_OPS
template <> __device__ inline int32_t  mul_cnt<int32_t>(int32_t v, int32_t n)   { return v * n; }
template <> __device__ inline int32_t  div_cnt<int32_t>(int32_t v, int32_t n)   { return v / n; }
template <> __device__ inline uint32_t mul_cnt<uint32_t>(uint32_t v, int32_t n) { return v * (uint32_t)n; }
template <> __device__ inline uint32_t div_cnt<uint32_t>(uint32_t v, int32_t n) { return v / (uint32_t)n; }
template <> __device__ inline int64_t  mul_cnt<int64_t>(int64_t v, int32_t n)   { return v * (int64_t)n; }
template <> __device__ inline int64_t  div_cnt<int64_t>(int64_t v, int32_t n)   { return v / (int64_t)n; }
template <> __device__ inline uint64_t mul_cnt<uint64_t>(uint64_t v, int32_t n) { return v * (uint64_t)(int64_t)n; }
template <> __device__ inline uint64_t div_cnt<uint64_t>(uint64_t v, int32_t n) { return v / (uint64_t)(int64_t)n; }
template <> __device__ inline float    mul_cnt<float>(float v, int32_t n)       { return v * (float)n; }
template <> __device__ inline float    div_cnt<float>(float v, int32_t n)       { return v / (float)n; }
template <> __device__ inline double   mul_cnt<double>(double v, int32_t n)     { return v * (double)n; }
template <> __device__ inline double   div_cnt<double>(double v, int32_t n)     { return v / (double)n; }

// aggr_val |= cval (Examples/FaultRecovery/FaultRecoveryFilter.C:112-122); the host admits
// integer types only
template <bool AND, typename T> __device__ inline T or_bits(T a, T b)
{
    uint64_t x = 0, y = 0;
    __builtin_memcpy(&x, &a, sizeof(T));
    __builtin_memcpy(&y, &b, sizeof(T));
    x = AND ? (x & y) : (x | y);
    T r;
    __builtin_memcpy(&r, &x, sizeof(T));
    return r;
}

template <int OP, typename T> struct Folder {
    T res;
    int32_t cnt;
    bool any;
    __device__ inline void init() { res = (T)0; cnt = 0; any = false; }
    __device__ inline void add(T x, int32_t c)
    {
        if (OP == MRPDR_SUM) {
            res = (T)(res + x);
        } else if (OP == MRPDR_MAX) {
            res = any ? ((res > x) ? res : x) : x;
        } else if (OP == MRPDR_MIN) {
            res = any ? ((res < x) ? res : x) : x;
        } else if (OP == MRPDR_OR) {
            res = or_bits<false>(res, x);
        } else if (OP == MRPDR_AND) {
            res = any ? or_bits<true>(res, x) : x;
        } else if (OP == OP_AVGSUM) {
            res = (T)(res + x);             // x is a product already, or a node's sum of products
            cnt += c;
        } else if (OP == OP_MAXR) {
            res = (any && !c) ? ((res > x) ? res : x) : x;
            cnt |= c;
        } else if (OP == OP_MINR) {
            res = (any && !c) ? ((res < x) ? res : x) : x;
            cnt |= c;
        } else {
            res = (T)(res + mul_cnt<T>(x, c));
            cnt += c;
        }
        any = true;
    }
    // add() for a fold that has taken at least one element: the same arithmetic, no flag to test
    __device__ inline void add_more(T x, int32_t c)
    {
        if (OP == MRPDR_SUM) res = (T)(res + x);
        else if (OP == MRPDR_MAX) res = (res > x) ? res : x;
        else if (OP == MRPDR_MIN) res = (res < x) ? res : x;
        else if (OP == MRPDR_OR) res = or_bits<false>(res, x);
        else if (OP == MRPDR_AND) res = or_bits<true>(res, x);
        else if (OP == OP_AVGSUM) { res = (T)(res + x); cnt += c; }
        else if (OP == OP_MAXR) { res = c ? x : ((res > x) ? res : x); cnt |= c; }
        else if (OP == OP_MINR) { res = c ? x : ((res < x) ? res : x); cnt |= c; }
        else { res = (T)(res + mul_cnt<T>(x, c)); cnt += c; }
    }
    // the one division of the flat integer AVG, at the top of its tree
    __device__ inline bool finish_top()
    {
        if (OP != OP_AVGSUM) return finish();
        if (!any || cnt == 0) return false;
        res = div_cnt<T>(res, cnt);
        return true;
    }
    // false: the node is absent (no child, or an AVG whose counts sum to zero)
    __device__ inline bool finish()
    {
        if (!any) return false;
        if (OP == MRPDR_AVG) {
            if (cnt == 0) return false;
            res = div_cnt<T>(res, cnt);
        }
        return true;
    }
};

template <typename T> __device__ inline uint64_t to_bits(T v)
{
    uint64_t b = 0;
    __builtin_memcpy(&b, &v, sizeof(T));
    return b;
}
template <typename T> __device__ inline T from_bits(uint64_t b)
{
    T v;
    __builtin_memcpy(&v, &b, sizeof(T));
    return v;
}

struct FilterIn {
    const uint8_t *values;  uint64_t value_stride;
    const uint8_t *counts;  uint64_t count_stride;
    uint64_t n;
    int32_t  swap;
    const uint8_t *orders;      // optional: sender byte order of every packet (then swap is unused)
};

__device__ inline bool filter_swap(const FilterIn &in, uint64_t i)
{
    return in.orders ? in.orders[i] != 0 : in.swap != 0;
}

template <typename T> __device__ inline T load_value(const FilterIn &in, uint64_t i)
{
    return from_bits<T>(ld_wire<sizeof(T)>(in.values + i * in.value_stride, filter_swap(in, i)));
}
__device__ inline int32_t load_count(const FilterIn &in, uint64_t i)
{
    return (int32_t)(uint32_t)ld_wire<4>(in.counts + i * in.count_stride, filter_swap(in, i));
}

// The data bytes of the filter's output packet: T value [, int32 count], no padding
// between them (Packet::pdr_packet_data ENCODE of "%T" or "%T %d").
// result and result_ok may lie in host memory that a node thread polls instead of synchronising
// (it presets *result_ok to a value that is neither 0 nor 1 and waits for it to change): the flag is
// therefore stored last, behind a system-scope fence, with release semantics -- whoever sees the
// flag sees the sixteen result bytes.
template <int OP, typename T>
__device__ inline void write_result(uint8_t *result, int32_t *result_ok, bool present, T v, int32_t cnt)
{
    uint64_t lo = 0, hi = 0;
    if (present) {
        uint8_t b[16] = { 0 };
        __builtin_memcpy(b, &v, sizeof(T));
        if (OpTraits<OP>::pair_result) __builtin_memcpy(b + sizeof(T), &cnt, 4);
        __builtin_memcpy(&lo, b, 8);
        __builtin_memcpy(&hi, b + 8, 8);
    }
    st_u<uint64_t>(result, lo);
    st_u<uint64_t>(result + 8, hi);
    if (result_ok) {
        __threadfence_system();             // (measured: 0.15-0.25 us of a 4.6 us call; the release store alone 0.2 more than a plain one)
        __hip_atomic_store(result_ok, present ? 1 : 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// A filter call over a RANGE of a wave's packets (the wave is spread over several GPUs, or comes in
// pieces): `front` is the node of the packets in front of the range (NULL: the range starts the
// wave), and the call leaves in `node` the node of front + range instead of a result.  What a node
// holds depends on how the operator's single reference call is computed (see OP_AVGSUM / OP_MINR /
// OP_MAXR above): the value so far, for AVG the undivided sum of value * count with the summed
// count, for float MIN / MAX the NaN mark; float and double SUM / AVG hold the running sum of the
// chain of additions itself, which is why their ranges have to be taken in order, each from the
// node the range before it left.
struct FilterLink { const Partial *front; Partial *node; };

template <typename T> __device__ inline void write_node(Partial *node, bool present, T v, int32_t cnt)
{
    Partial out;
    out.bits = present ? to_bits<T>(v) : 0;
    out.cnt = present ? cnt : 0;
    out.present = present ? 1 : 0;
    *node = out;
}

// ---- tree shape, levels 0..7: one partial per workgroup -------------------------
// Levels 0..7 for one operator over the packets this lane loaded: level 0 folds the lane's
// packets in order, levels 1..6 pair neighbours across the wavefront (the lower lane is the left
// child), level 7 folds the wavefronts of the workgroup.  Thread 0 returns the workgroup's node.
// Contains workgroup barriers: every thread calls it.
template <int OP, typename T>
__device__ inline Partial tile_node(const T *vals, const int32_t *cnts, int nvalid, Partial *wave_part)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Folder<OP, T> fo;
    fo.init();
#pragma unroll
    for (int k = 0; k < FILTER_LEAF; k++)
        if (k < nvalid) fo.add(vals[k], OpTraits<OP>::counted ? cnts[k] : 1);
    bool present = fo.finish();
    T v = fo.res;
    int32_t cnt = fo.cnt;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const uint64_t ob = __shfl_xor((unsigned long long)to_bits<T>(v), m, 64);
        const int32_t  oc = __shfl_xor(cnt, m, 64);
        const int      op = __shfl_xor((int)present, m, 64);
        const bool right = (lane & m) != 0;
        const T lv = right ? from_bits<T>(ob) : v;   const T rv = right ? v : from_bits<T>(ob);
        const int32_t lc = right ? oc : cnt;         const int32_t rc = right ? cnt : oc;
        const bool lp = right ? (op != 0) : present; const bool rp = right ? present : (op != 0);
        Folder<OP, T> g;
        g.init();
        if (lp) g.add(lv, lc);
        if (rp) g.add(rv, rc);
        present = g.finish();
        v = g.res; cnt = g.cnt;
    }
    __syncthreads();                     // wave_part may still be read from the previous operator
    if (lane == 0) {
        wave_part[wave].bits = to_bits<T>(v);
        wave_part[wave].cnt = cnt;
        wave_part[wave].present = present ? 1 : 0;
    }
    __syncthreads();
    Partial out;
    out.bits = 0; out.cnt = 0; out.present = 0;
    if (threadIdx.x == 0) {
        Folder<OP, T> g;
        g.init();
        for (int w = 0; w < MRPDR_TPB / 64; w++)
            if (wave_part[w].present) g.add(from_bits<T>(wave_part[w].bits), wave_part[w].cnt);
        out.present = g.finish() ? 1 : 0;
        out.bits = to_bits<T>(g.res);
        out.cnt = g.cnt;
    }
    return out;
}

// This lane's packets: values and (with COUNTS) counts; returns how many there are.
// All loads of the lane (values, counts, byte orders) are issued before the first of them is
// used: they are unconditional (a lane past the end re-reads the last packet) and the bytes are
// put in order afterwards.  Written packet by packet with the swap decided at the load, this
// compiled to one wait per load, up to sixteen memory round trips in a row.
template <typename T, bool COUNTS>
__device__ inline int tile_load(const FilterIn &in, T *vals, int32_t *cnts)
{
    typedef typename UIntOf<sizeof(T)>::type U;
    const uint64_t first = (uint64_t)blockIdx.x * FILTER_TILE + (uint64_t)threadIdx.x * FILTER_LEAF;
    U raw[FILTER_LEAF];
    uint32_t rawc[FILTER_LEAF];
    uint8_t ord[FILTER_LEAF];
    const uint64_t last = in.n - 1;              // the launch has a workgroup only where there are packets
#pragma unroll
    for (int k = 0; k < FILTER_LEAF; k++) {
        const uint64_t i = first + k < last ? first + k : last;
        raw[k] = ld_u<U>(in.values + i * in.value_stride);
        rawc[k] = COUNTS ? ld_u<uint32_t>(in.counts + i * in.count_stride) : 1u;
    }
    if (in.orders) {
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) ord[k] = in.orders[first + k < last ? first + k : last];
    } else {
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) ord[k] = in.swap != 0;
    }
    int nvalid = 0;
#pragma unroll
    for (int k = 0; k < FILTER_LEAF; k++) {
        vals[k] = T();
        cnts[k] = 1;
        if (first + k < in.n) {
            U r = raw[k];
            uint32_t c = rawc[k];
            if (ord[k]) { r = bswap_u(r); c = bswap_u(c); }
            vals[k] = from_bits<T>((uint64_t)r);
            if (COUNTS) cnts[k] = (int32_t)c;
            nvalid = k + 1;
        }
    }
    return nvalid;
}

// The last level over a wave of one tile, by the thread that holds the tile's node: the same
// fold as tree_top over one staged node, so a wave of up to FILTER_TILE packets needs no second
// launch (the reference's own microbenchmark wave has four packets).
template <int OP, typename T>
__device__ inline void top_of_one(const Partial &node, uint8_t *result, int32_t *result_ok)
{
    Folder<OP, T> g;
    g.init();
    if (node.present) g.add(from_bits<T>(node.bits), node.cnt);
    const bool present = g.finish_top();
    write_result<OP, T>(result, result_ok, present, g.res, g.cnt);
}

// result != NULL: the launch has one workgroup and writes the filter's result itself.
template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_filter_tree_tile(FilterIn in, Partial *partials, uint8_t *result,
                                                                int32_t *result_ok)
{
    __shared__ Partial wave_part[MRPDR_TPB / 64];
    T vals[FILTER_LEAF];
    int32_t cnts[FILTER_LEAF];
    const int nvalid = tile_load<T, OP == MRPDR_AVG || OP == OP_AVGSUM>(in, vals, cnts);
    if (OP == OP_AVGSUM) {
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) vals[k] = mul_cnt<T>(vals[k], cnts[k]);
    } else if (OP == OP_MINR || OP == OP_MAXR) {
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) cnts[k] = vals[k] != vals[k];
    }
    const Partial out = tile_node<OP, T>(vals, cnts, nvalid, wave_part);
    if (threadIdx.x != 0) return;
    if (result) top_of_one<OP, T>(out, result, result_ok);
    else partials[blockIdx.x] = out;
}

// One lane folds m staged nodes in order.  Eight LDS reads are issued before the first of them is
// used: the fold is a dependent chain, the reads need not be.
template <int OP, typename T>
__device__ inline void fold_nodes(Folder<OP, T> &g, const Partial *stage, uint32_t m, bool dense)
{
    if (dense && m) {
        // every staged node is present (all but degenerate waves): a straight dependent chain of
        // the operator itself, nothing else between two links
        g.add(from_bits<T>(stage[0].bits), stage[0].cnt);
        for (uint32_t k = 1; k < m; k += 8) {
            Partial p[8];
#pragma unroll
            for (int j = 0; j < 8; j++) p[j] = stage[(k + j < m) ? k + j : k];
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (k + j < m) g.add_more(from_bits<T>(p[j].bits), p[j].cnt);
        }
        return;
    }
    for (uint32_t k = 0; k < m; k += 8) {
        Partial p[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            p[j].present = 0;
            if (k + j < m) p[j] = stage[k + j];
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (p[j].present) g.add(from_bits<T>(p[j].bits), p[j].cnt);
    }
}

// ---- tree shape, last level: all workgroup partials, in order ---------------------
// Contains workgroup barriers: every thread calls it.
template <int OP, typename T>
__device__ inline void tree_top(const Partial *partials, uint64_t np, uint8_t *result, int32_t *result_ok,
                                Partial *stage, FilterLink link = FilterLink{ nullptr, nullptr })
{
    Folder<OP, T> g;
    g.init();
    if (link.front && threadIdx.x == 0) {            // the packets in front of this range: the leftmost child
        const Partial f = *link.front;
        if (f.present) g.add(from_bits<T>(f.bits), f.cnt);
    }
    for (uint64_t base = 0; base < np; base += FLAT_TILE) {
        const uint32_t m = (uint32_t)((np - base) < FLAT_TILE ? (np - base) : FLAT_TILE);
        __syncthreads();
        int all = 1;
        for (uint32_t k = threadIdx.x; k < m; k += MRPDR_TPB) {
            stage[k] = partials[base + k];
            all &= stage[k].present;
        }
        const bool dense = __syncthreads_and(all) != 0;
        if (threadIdx.x == 0) fold_nodes<OP, T>(g, stage, m, dense);
    }
    if (threadIdx.x == 0) {
        if (link.node) {
            write_node<T>(link.node, g.finish(), g.res, g.cnt);
            return;
        }
        const bool present = g.finish_top();
        write_result<OP, T>(result, result_ok, present, g.res, g.cnt);
    }
}

template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_filter_tree_top(const Partial *partials, uint64_t np,
                                                               uint8_t *result, int32_t *result_ok)
{
    __shared__ Partial stage[FLAT_TILE];
    tree_top<OP, T>(partials, np, result, result_ok, stage);
}

// The last level of a range's own tree, with the node in front as its leftmost child, left as a node
// (link.node), or -- np nodes of consecutive ranges, link empty -- the filter's result.
template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_filter_tree_link(const Partial *partials, uint64_t np, FilterLink link,
                                                                uint8_t *result, int32_t *result_ok)
{
    __shared__ Partial stage[FLAT_TILE];
    tree_top<OP, T>(partials, np, result, result_ok, stage, link);
}

// ---- the chain of float / double additions, taken up and handed on ----------------------------------
// Where the chain starts (the sum the packets in front left, 0 without any) ...
template <typename T> __device__ inline T chain_start(const FilterLink &link)
{
    if (!link.front) return (T)0;
    const Partial f = *link.front;
    return f.present ? from_bits<T>(f.bits) : (T)0;
}
// ... and where it ends (one thread): the node of front + range, or the filter's result
template <int OP, typename T>
__device__ inline void chain_finish(const FilterLink &link, uint64_t n, T acc, int32_t cnt, uint8_t *result, int32_t *result_ok)
{
    bool present = n != 0;
    if (link.front) {
        const Partial f = *link.front;
        if (f.present) { present = true; cnt += f.cnt; }
    }
    if (link.node) {
        write_node<T>(link.node, present, acc, OP == MRPDR_AVG ? cnt : 0);
        return;
    }
    if (OP == MRPDR_AVG) {
        if (cnt == 0) present = false;
        else acc = div_cnt<T>(acc, cnt);
    }
    write_result<OP, T>(result, result_ok, present, acc, cnt);
}

// ---- Min, Max and Avg of one '<T> %d' wave in one pass ------------------------------
// The packets are read once; each operator then goes through the same tree as on its own
// (same fan-ins, same order), so the three results equal those of three separate filters.
// partials: three runs of `stride` nodes (MIN, MAX, AVG); results: 3 x 16 bytes; oks: 3.
template <typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_filter_stats_tile(FilterIn in, Partial *partials, uint64_t stride)
{
    __shared__ Partial wave_part[MRPDR_TPB / 64];
    T vals[FILTER_LEAF];
    int32_t cnts[FILTER_LEAF];
    const int nvalid = tile_load<T, true>(in, vals, cnts);
    const Partial a = tile_node<MRPDR_MIN, T>(vals, cnts, nvalid, wave_part);
    const Partial b = tile_node<MRPDR_MAX, T>(vals, cnts, nvalid, wave_part);
    const Partial c = tile_node<MRPDR_AVG, T>(vals, cnts, nvalid, wave_part);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = a;
        partials[stride + blockIdx.x] = b;
        partials[2 * stride + blockIdx.x] = c;
    }
}

#define STATS_TILE 1024                 // nodes of each operator staged per step (3 x 16 KiB of LDS)

template <typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_filter_stats_top(const Partial *partials, uint64_t np, uint64_t stride,
                                                                uint8_t *results, int32_t *oks)
{
    // the three folds are independent chains: lane 0 of wavefronts 0, 1, 2 takes one each
    __shared__ Partial stage[3][STATS_TILE];
    const int mine = (threadIdx.x & 63) == 0 ? (int)(threadIdx.x >> 6) : 3;
    Folder<MRPDR_MIN, T> gmin;
    Folder<MRPDR_MAX, T> gmax;
    Folder<MRPDR_AVG, T> gavg;
    gmin.init(); gmax.init(); gavg.init();
    for (uint64_t base = 0; base < np; base += STATS_TILE) {
        const uint32_t m = (uint32_t)((np - base) < STATS_TILE ? (np - base) : STATS_TILE);
        __syncthreads();
        int all = 1;
        for (uint32_t k = threadIdx.x; k < 3 * m; k += MRPDR_TPB) {
            const uint32_t o = k / m, j = k - o * m;
            stage[o][j] = partials[o * stride + base + j];
            all &= stage[o][j].present;
        }
        const bool dense = __syncthreads_and(all) != 0;
        if (mine == 0) fold_nodes<MRPDR_MIN, T>(gmin, stage[0], m, dense);
        else if (mine == 1) fold_nodes<MRPDR_MAX, T>(gmax, stage[1], m, dense);
        else if (mine == 2) fold_nodes<MRPDR_AVG, T>(gavg, stage[2], m, dense);
    }
    if (mine == 0) { const bool p = gmin.finish(); write_result<MRPDR_MIN, T>(results, oks, p, gmin.res, gmin.cnt); }
    if (mine == 1) { const bool p = gmax.finish(); write_result<MRPDR_MAX, T>(results + 16, oks + 1, p, gmax.res, gmax.cnt); }
    if (mine == 2) { const bool p = gavg.finish(); write_result<MRPDR_AVG, T>(results + 32, oks + 2, p, gavg.res, gavg.cnt); }
}

// ---- flat shape: one left fold over the whole wave ------------------------------
// Only what depends on the order runs here: SUM and AVG of float and double (every other flat
// result is computed level by level, see OP_AVGSUM / OP_MINR / OP_MAXR and the host side).
// The fold is one dependent chain of additions, a few clocks each -- that chain is the floor;
// everything around it is kept off it: the whole workgroup loads the next tile (byte order,
// value * count for AVG) into registers while wavefront 0 adds up the current one out of LDS,
// sixteen values per batch of 128-bit LDS reads issued one batch ahead; the AVG count is an
// integer sum and is taken in parallel.
template <typename T> __device__ inline void lds_read16(T (&v)[16], const T *s)
{
    struct alignas(16) Q { T x[16 / sizeof(T)]; };
    constexpr int PERQ = 16 / sizeof(T);
#pragma unroll
    for (int q = 0; q < 16 / PERQ; q++) {
        const Q a = *reinterpret_cast<const Q *>(s + q * PERQ);
#pragma unroll
        for (int k = 0; k < PERQ; k++) v[q * PERQ + k] = a.x[k];
    }
}

// A tile's packets as loaded: raw bits, put in order only when they are stored to LDS, so that
// nothing waits for the loads before the fold of the tile in front has run.  The loads are
// unconditional (a lane past the end re-reads the last packet).
template <typename T, bool COUNTS> struct FlatRaw {
    typedef typename UIntOf<sizeof(T)>::type U;
    static constexpr int PER = FLAT_TILE / MRPDR_TPB;
    U raw[PER];
    uint32_t rawc[PER];
    uint8_t ord[PER];
    __device__ inline void issue(const FilterIn &in, uint64_t base)
    {
        const uint64_t last = in.n - 1;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            uint64_t i = base + (uint64_t)k * MRPDR_TPB + threadIdx.x;
            i = i < last ? i : last;
            raw[k] = ld_u<U>(in.values + i * in.value_stride);
            rawc[k] = COUNTS ? ld_u<uint32_t>(in.counts + i * in.count_stride) : 1u;
        }
        if (in.orders) {
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const uint64_t i = base + (uint64_t)k * MRPDR_TPB + threadIdx.x;
                ord[k] = in.orders[i < last ? i : last];
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++) ord[k] = (uint8_t)(in.swap != 0);
        }
    }
    // values (times count with COUNTS) to dst[FLAT_TILE]; returns the sum of this lane's counts
    __device__ inline int32_t commit(const FilterIn &in, uint64_t base, T *dst) const
    {
        int32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint64_t i = base + (uint64_t)k * MRPDR_TPB + threadIdx.x;
            U r = raw[k];
            uint32_t c = rawc[k];
            if (ord[k]) { r = bswap_u(r); c = bswap_u(c); }
            T v = from_bits<T>((uint64_t)r);
            if (COUNTS) v = mul_cnt<T>(v, (int32_t)c);
            if (i < in.n) { if (COUNTS) cnt += (int32_t)c; } else v = (T)0;
            dst[k * MRPDR_TPB + threadIdx.x] = v;
        }
        return cnt;
    }
};

template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_filter_flat(FilterIn in, uint8_t *result, int32_t *result_ok,
                                                           FilterLink link = FilterLink{ nullptr, nullptr })
{
    __shared__ __align__(16) T sv[2][FLAT_TILE];
    __shared__ int32_t s_cnt;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_cnt = 0;
    const uint64_t ntiles = (in.n + FLAT_TILE - 1) / FLAT_TILE;
    T acc = chain_start<T>(link);
    int32_t mycnt = 0;
    FlatRaw<T, OP == MRPDR_AVG> regs;
    if (ntiles) {
        regs.issue(in, 0);
        mycnt += regs.commit(in, 0, sv[0]);
    }
    __syncthreads();
    for (uint64_t tile = 0; tile < ntiles; tile++) {
        const uint64_t base = tile * FLAT_TILE;
        const uint32_t m = (uint32_t)((in.n - base) < FLAT_TILE ? (in.n - base) : FLAT_TILE);
        const bool more = tile + 1 < ntiles;
        if (more) regs.issue(in, base + FLAT_TILE);
        if (t < 64) {           // every lane of wavefront 0 adds the same values: the reads are broadcasts
            const T *s = sv[tile & 1];
            uint32_t j = 0;
            // two register sets take turns, each read one batch ahead of its use (a read past the
            // tile's m values, still inside the buffer, is not used)
            T a[16], b[16];
            lds_read16(a, s);
            for (; j + 32 <= m; j += 32) {
                lds_read16(b, s + j + 16);
                __builtin_amdgcn_sched_barrier(0);          // keep the reads in front of the additions they overlap
#pragma unroll
                for (int k = 0; k < 16; k++) acc = acc + a[k];
                lds_read16(a, s + (j + 48 <= FLAT_TILE ? j + 32 : j));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; k++) acc = acc + b[k];
            }
            if (j + 16 <= m) {
#pragma unroll
                for (int k = 0; k < 16; k++) acc = acc + a[k];
                j += 16;
            }
            for (; j < m; j++) acc = acc + s[j];
        }
        if (more) mycnt += regs.commit(in, base + FLAT_TILE, sv[(tile + 1) & 1]);
        __syncthreads();
    }
    if (OP == MRPDR_AVG) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) mycnt += __shfl_xor(mycnt, m, 64);
        if ((t & 63u) == 0) atomicAdd(&s_cnt, mycnt);
        __syncthreads();
    }
    if (t == 0) chain_finish<OP, T>(link, in.n, acc, s_cnt, result, result_ok);
}

// ---------------------------------------------------------------------------
// tfilter_PerfData: element-wise Sum / Min / Max / Avg over array packets
// ---------------------------------------------------------------------------
// Reference: src/FilterDefinitions.C:904-1239, packets "%ad %ad %a{uld,ld,lf}" = (rank array,
// nelems array, data array).  One invocation folds the data arrays element by element in packet
// order; packets without data are skipped; rank[0] < 0 means "already an aggregate of -rank
// ranks".  Literal quirks kept: the MIN accumulator starts as bytes 0xFF (UINT64_MAX, -1 for
// int64, a NaN for double that the first '<' discards), every other one as 0 (so MAX of negative
// data is 0); AVG adds x * nrank and divides the first data_len(LAST packet) elements by the
// total number of ranks; the output rank array is [-total_nranks].
//
// One lane per element u; workgroup row g folds the packets of group g (group = all packets:
// one invocation; smaller groups: the first level of a tree whose second level is a second
// launch over the group results, which this kernel leaves in column form).
struct PerfIn {
    const int32_t  *rank_vals; const uint64_t *rank_counts; const uint64_t *rank_offs;
    const void     *data_vals; const uint64_t *data_counts; const uint64_t *data_offs;
    uint64_t n, group;
};
struct PerfOut {
    void     *data; uint64_t stride;      // group g writes data[g * stride + u]
    uint64_t *lens;                       // [groups] aggr_results.size() of the group
    uint64_t *offs;                       // [groups] g * stride (column form for the next level), or NULL
    int32_t  *ranks;                      // [groups] -total_nranks
    uint64_t *rank_counts, *rank_offs;    // [groups] 1 / g, or NULL
    uint32_t *status;
};

#define PERF_STAGE_ELEMS 4096        // 32 KB of LDS for the data of one tile of packets

// Raw nodes (perfnodes.h has the whole story): with in_lasts the packets are raw nodes of a level
// below (added with weight one, `last` taken from the last of them); with out_lasts the node is left
// raw (no AVG division, its `last` written); front = a raw node handed over by whoever folded the
// packets in front of these (header, then the accumulators): the fold goes on from it.
struct PerfRaw {
    const uint64_t *in_lasts;
    uint64_t *out_lasts;
    const uint8_t *front;             // struct mrpdr_perf_state + accumulators, or NULL
    uint32_t *out_packets;            // where a state's `packets` word goes, or NULL
    const uint64_t *unless;           // the launch leaves at once when *unless != 0 (another path has the wave), or NULL
};
struct PerfFront { uint64_t longest, last; int32_t rank; uint32_t packets; uint64_t reserved; };

template <int OP, typename T> __device__ __forceinline__ T perf_fold(T acc, T x, int32_t nrank)
{
    if (OP == MRPDR_SUM) return acc + x;
    if (OP == MRPDR_MIN) return (acc < x) ? acc : x;
    if (OP == MRPDR_MAX) return (acc > x) ? acc : x;
    return acc + x * (T)nrank;
}

template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_perfdata(PerfIn in, PerfOut out, PerfRaw raw = PerfRaw{ nullptr, nullptr, nullptr, nullptr, nullptr })
{
    // The packets of the node are taken in tiles of 256: their lengths, offsets and rank counts
    // go to LDS (one packet per lane), and when the tile's data -- one contiguous run of the
    // values column -- fits, so does the data, with coalesced loads.  The fold itself is serial
    // in packet order per element, as the reference's, and then reads LDS only.
    __shared__ uint64_t s_len[MRPDR_TPB], s_off[MRPDR_TPB];
    __shared__ int32_t s_nrank[MRPDR_TPB];
    __shared__ uint4 s_rec[MRPDR_TPB];           // staged tiles: {position in s_data, length, rank count, -}
    __shared__ T s_data[PERF_STAGE_ELEMS];
    __shared__ uint32_t s_total;
    __shared__ unsigned long long s_longest;
    if (raw.unless && *raw.unless) return;
    const uint64_t g = blockIdx.y;
    const uint32_t t = threadIdx.x;
    const uint64_t u = (uint64_t)blockIdx.x * MRPDR_TPB + t;
    const uint64_t i0 = g * in.group, i1 = (i0 + in.group < in.n) ? i0 + in.group : in.n;
    const T *data = reinterpret_cast<const T *>(in.data_vals);
    T acc;
    __builtin_memset(&acc, OP == MRPDR_MIN ? 0xFF : 0, sizeof acc);
    uint32_t total = 0;
    uint64_t longest = 0, last = 0;
    PerfFront front = { 0, 0, 0, 0, 0 };
    if (raw.front) {
        front = *reinterpret_cast<const PerfFront *>(raw.front);
        if (!front.packets) { front.longest = 0; front.last = 0; front.rank = 0; }
        if (u < front.longest) acc = reinterpret_cast<const T *>(raw.front + sizeof(PerfFront))[u];
        last = front.last;
    }
    if (t == 0) { s_total = (uint32_t)(0 - front.rank); s_longest = front.longest; }
    for (uint64_t p0 = i0; p0 < i1; p0 += MRPDR_TPB) {
        const uint32_t np = (uint32_t)((i1 - p0) < MRPDR_TPB ? (i1 - p0) : MRPDR_TPB);
        __syncthreads();
        // The rank total and the longest array do not depend on the order of the packets: the
        // lanes that load the tile's lengths add them up (wavefront first, then LDS), which
        // leaves the serial loop below with the fold alone.
        uint32_t my_ranks = 0;
        uint64_t my_len = 0;
        // (the four loads are unconditional, a lane past the end re-reads the tile's last packet:
        // they are issued together, and only the rank itself is a second, dependent round trip)
        const uint64_t pi = p0 + (t < np ? t : np - 1);
        const uint64_t ld_len = in.data_counts[pi], ld_off = in.data_offs[pi];
        const uint64_t ld_rc = in.rank_counts[pi], ld_ro = in.rank_offs[pi];
        // where the tile's data lies in the column, read by every lane itself (three loads of one
        // address each): whether the tile is staged is known without a trip through LDS, and
        // the data loads go out together with the rank load
        const uint64_t base = in.data_offs[p0], tail_off = in.data_offs[p0 + np - 1];
        last = in.data_counts[p0 + np - 1];
        const uint64_t run = tail_off + last - base;
        // staged only by the workgroup of the first 256 elements: longer arrays read the column
        // with full wavefronts anyway
        const bool staged = gridDim.x == 1 && tail_off >= base && run <= PERF_STAGE_ELEMS;
        T x[8];
        if (staged && run) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint64_t k = (uint64_t)q * MRPDR_TPB + t;
                x[q] = data[base + (k < run ? k : run - 1)];
            }
        }
        if (t < np) {
            const uint64_t len = ld_len;
            s_len[t] = len;
            s_off[t] = ld_off;
            int32_t rank = 0;
            if (len && ld_rc) rank = in.rank_vals[ld_ro];
            const int32_t nrank = rank < 0 ? -rank : 1;
            s_nrank[t] = raw.in_lasts ? 1 : nrank;         // (a raw node below is added as it is)
            my_ranks = len ? (uint32_t)nrank : 0u;
            my_len = len;
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            my_ranks += __shfl_xor(my_ranks, m, 64);
            const uint64_t o = __shfl_xor((unsigned long long)my_len, m, 64);
            my_len = o > my_len ? o : my_len;
        }
        if ((t & 63u) == 0) {
            atomicAdd(&s_total, my_ranks);
            atomicMax(&s_longest, (unsigned long long)my_len);
        }
        if (staged) {
            if (t < np) s_rec[t] = make_uint4((uint32_t)(s_off[t] - base), (uint32_t)s_len[t], (uint32_t)s_nrank[t], 0u);
            if (run) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint64_t k = (uint64_t)q * MRPDR_TPB + t;
                    if (k < run) s_data[k] = x[q];
                }
            }
            // what lies beyond the first eight loads of a lane, eight at a time again
            for (uint64_t kb = 8u * MRPDR_TPB; kb < run; kb += 8u * MRPDR_TPB) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint64_t k = kb + (uint64_t)q * MRPDR_TPB + t;
                    x[q] = data[base + (k < run ? k : run - 1)];
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint64_t k = kb + (uint64_t)q * MRPDR_TPB + t;
                    if (k < run) s_data[k] = x[q];
                }
            }
            __syncthreads();
            // branch-free so that the LDS reads of several packets are in flight together; the
            // fold itself stays in packet order
#pragma unroll 8
            for (uint32_t j = 0; j < np; j++) {
                const uint4 r = s_rec[j];
                const bool active = t < r.y;
                const T xv = s_data[active ? r.x + t : 0u];
                const T f = perf_fold<OP, T>(acc, xv, (int32_t)r.z);
                acc = active ? f : acc;
            }
        } else {
            __syncthreads();
#pragma unroll 4
            for (uint32_t j = 0; j < np; j++) {
                const uint64_t len = s_len[j];
                if (u < len) acc = perf_fold<OP, T>(acc, data[s_off[j] + u], s_nrank[j]);
            }
        }
    }
    __syncthreads();
    total = s_total;
    longest = s_longest;
    if (raw.in_lasts && i1 > i0) last = raw.in_lasts[i1 - 1];
    if (!raw.out_lasts && OP == MRPDR_AVG && last && u < last) {
        acc = acc / (T)total;          // unsigned total converted to the element type, as the reference's /=
    }
    if (longest > out.stride) {                    // the result does not fit: say so, write nothing
        if (u == 0) atomicOr(out.status, 1u);
        longest = 0;
    }
    if (u < longest) reinterpret_cast<T *>(out.data)[g * out.stride + u] = acc;
    if (u == 0) {
        out.lens[g] = longest;
        out.ranks[g] = 0 - (int32_t)total;
        if (out.offs) out.offs[g] = g * out.stride;
        if (out.rank_counts) { out.rank_counts[g] = 1; out.rank_offs[g] = g; }
        if (raw.out_lasts) raw.out_lasts[g] = last;
        if (raw.out_packets) *raw.out_packets = 1u;
    }
}
