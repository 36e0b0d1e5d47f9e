// tfilter_PerfData over many nodes at once: the levels of a tree of filter invocations, and the
// order-free form of ONE invocation over integer data.
//
// Reference: src/FilterDefinitions.C:904-1239 (see k_perfdata in filter.h for the fold's rules).  One
// invocation folds its packets one after the other, element by element.  k_perfdata gives a node a
// workgroup and an element a lane, which leaves 8 of 256 lanes busy on the arrays performance data
// really carries (eight doubles per packet, src/Stream.C:1086-1110) and makes a node of 256 packets a
// chain of 256 dependent steps.  Here a workgroup holds as many nodes as fit -- a node takes W lanes,
// W the array bound rounded up to a power of two -- so that the levels of a tree are short chains
// side by side, and when the nodes of a workgroup are exactly the children of one node of the next
// level (fan-in 32 for arrays of up to eight elements) that node is folded in the same launch:
// 131,072 packets as 32 x 32 x 32 x 4 are two launches whose chains are 32 + 32 and 32 + 4 steps long.
//
// Two readings of "levels":
//   * the declared tree: every node is a filter invocation of its own, as in a tree of processes --
//     AVG divides at every node and the node's output carries rank -total, which weights it one level
//     up (in_lasts == out_lasts == NULL);
//   * raw nodes: nothing is divided and nothing is re-weighted on the way up -- a node's output is the
//     accumulator itself, its total of ranks and the data length of the last packet below it
//     (out_lasts), and a level over raw nodes (in_lasts) adds them with weight one and keeps the last
//     node's `last`.  For uint64 / int64 data (wrapping adds, one division at the very top; MIN and MAX
//     with their fixed start values) that is ONE invocation over all packets, bit for bit, whatever the
//     grouping.  Doubles round at every addition: their single invocation stays the serial k_perfdata.
#pragma once

#include "filter.h"

#ifdef MRPDR_TILE_STAMPS
#define PERF_STAMP(k) do { if (threadIdx.x == 0 && g_tile_stamps) g_tile_stamps[(uint64_t)blockIdx.y * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PERF_STAMP(k) do { } while (0)
#endif

#define PERF_CHUNK 32u          // packets of a node whose records are staged at a time
#define PERF_MIN_W 8u           // lanes per node at least: at most 32 nodes per workgroup

// header of a raw node handed between GPUs (struct mrpdr_perf_state), followed by its accumulators
struct PerfState {
    uint64_t longest;           // aggr_results.size(): elements that have taken at least one packet
    uint64_t last;              // data_len of the last packet below (what AVG divides in the end)
    int32_t  rank;              // 0 - total_nranks, the rank value of the output packet
    uint32_t packets;           // 0: no packet below (then nothing else counts, not even `last`)
    uint64_t reserved;
};

struct PerfLevel {
    PerfIn in;                  // the packets of this level (columns), in.group of them per node
    PerfOut out;                // node g: data[g * stride + u], lens[g], offs[g], ranks[g], rank_counts / offs[g]
    const uint64_t *in_lasts;   // packets are raw nodes: `last` of each (NULL: packets as they come off the wire)
    uint64_t *out_lasts;        // leave raw nodes: `last` of each (NULL: the filter's own output)
    uint32_t *out_packets;      // raw output of ONE node that is a state: its `packets` word (may be NULL)
    uint32_t W;                 // lanes per node: a power of two, PERF_MIN_W .. MRPDR_TPB
    uint32_t nodes;             // ceil(n / group)
    uint64_t regular;           // the packets are nodes this library wrote itself: packet q has its data at
                                // q * regular and its one rank at q (0: columns of any layout)
    uint32_t fuse;              // the nodes of a workgroup are the children of ONE node of the level above,
                                // which is folded here as well (out is that level's: node = workgroup)
    uint32_t raw_mid;           // fuse: the nodes in between are raw
};

// a packet's length as the records keep it (a length that does not fit is longer than any bound the
// kernel runs with, and stays so)
__device__ __forceinline__ uint32_t perf_len32(uint64_t len) { return len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len; }

template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_perf_nodes(PerfLevel a)
{
    // per staged packet: where its data lies and how long it is (0: nothing to add) ...
    __shared__ uint4 s_rec[(MRPDR_TPB / PERF_MIN_W) * PERF_CHUNK];
    // ... and what it weighs (0: no packet at all)
    __shared__ int32_t s_nrank[(MRPDR_TPB / PERF_MIN_W) * PERF_CHUNK];
    const uint32_t t = threadIdx.x, W = a.W, npb = MRPDR_TPB / W;
    const uint32_t ln = t / W, lu = t % W;
    const uint64_t node0 = (uint64_t)blockIdx.y * npb, node = node0 + ln;
    const uint64_t u = (uint64_t)blockIdx.x * W + lu;
    const bool raw_in = a.in_lasts != nullptr, raw_out = a.out_lasts != nullptr;
    const uint64_t group = a.in.group, n = a.in.n;
    const uint64_t i0 = node * group;
    const uint64_t i1 = node < a.nodes ? (i0 + group < n ? i0 + group : n) : i0;
    const T *data = reinterpret_cast<const T *>(a.in.data_vals);
    // (a load that is not wanted reads this instead: an address that is there whatever the wave holds)
    const uint64_t *idle = a.in.data_counts;
    T acc;
    __builtin_memset(&acc, OP == MRPDR_MIN ? 0xFF : 0, sizeof acc);
    uint32_t total = 0;
    uint64_t longest = 0, last = 0;
    // A node's records are fetched by the node's own lanes: lane lu takes packets lu, lu + W, ... of the
    // chunk (W <= 32: PERF_CHUNK / W each; wider nodes: the first 32 lanes one each).
    PERF_STAMP(0);
    constexpr int RMAX = PERF_CHUNK / PERF_MIN_W;
    const uint32_t per_lane = W < PERF_CHUNK ? PERF_CHUNK / W : 1u;
    uint4 *rec = s_rec + ln * PERF_CHUNK;
    int32_t *nrk = s_nrank + ln * PERF_CHUNK;
    for (uint64_t c = 0; c < group; c += PERF_CHUNK) {
        const uint64_t at = i0 + c;
        const uint32_t m = at < i1 ? (uint32_t)(i1 - at < PERF_CHUNK ? i1 - at : PERF_CHUNK) : 0u;
        __syncthreads();                        // the chunk before has been folded
        uint64_t len[RMAX], off[RMAX], rc[RMAX], ro[RMAX];
        int32_t rank[RMAX];
        T x[PERF_CHUNK];
        if (a.regular) {
            // nodes of a level below, laid out by this library: everything is one round trip -- lengths,
            // ranks and (whatever the lengths will say: a node's room is always there) the values
#pragma unroll
            for (int k = 0; k < RMAX; k++) {
                const uint32_t j = lu + (uint32_t)k * W;
                const bool mine = (uint32_t)k < per_lane && j < m;
                const uint64_t q = mine ? at + j : n - 1;
                len[k] = a.in.data_counts[q]; rank[k] = a.in.rank_vals[q];
                off[k] = q * a.regular; rc[k] = 1; ro[k] = q;
            }
#pragma unroll
            for (uint32_t j = 0; j < PERF_CHUNK; j++) {
                const uint64_t q = j < m ? at + j : (m ? at + m - 1 : n - 1);
                x[j] = data[q * a.regular + u];
            }
#pragma unroll
            for (int k = 0; k < RMAX; k++) {
                const uint32_t j = lu + (uint32_t)k * W;
                const bool mine = (uint32_t)k < per_lane && j < m;
                if ((uint32_t)k < per_lane && j < PERF_CHUNK)
                    rec[j] = mine ? make_uint4((uint32_t)off[k], (uint32_t)(off[k] >> 32), perf_len32(len[k]), 1u) : make_uint4(0u, 0u, 0u, 0u);
            }
        } else {
            // round trip one: lengths, places, rank counts (unconditional: a record past the node's end
            // re-reads packet n - 1)
#pragma unroll
            for (int k = 0; k < RMAX; k++) {
                const uint32_t j = lu + (uint32_t)k * W;
                const bool mine = (uint32_t)k < per_lane && j < m;
                const uint64_t q = mine ? at + j : n - 1;
                len[k] = a.in.data_counts[q]; off[k] = a.in.data_offs[q];
                rc[k] = a.in.rank_counts[q]; ro[k] = a.in.rank_offs[q];
            }
            // round trip two, first half: the ranks (asked for now, looked at behind the data loads)
#pragma unroll
            for (int k = 0; k < RMAX; k++) {
                const uint32_t j = lu + (uint32_t)k * W;
                const bool mine = (uint32_t)k < per_lane && j < m;
                const int32_t *p = (mine && len[k] && rc[k]) ? a.in.rank_vals + ro[k] : reinterpret_cast<const int32_t *>(idle);
                rank[k] = *p;
                if ((uint32_t)k < per_lane && j < PERF_CHUNK)
                    rec[j] = mine ? make_uint4((uint32_t)off[k], (uint32_t)(off[k] >> 32), perf_len32(len[k]), 1u) : make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
            PERF_STAMP(1);
            // round trip two, second half: every value of the chunk this lane will add.  (All records out of
            // LDS first, then all loads: read and load by turns, every load waited for its record's trip to LDS.)
            uint4 r[PERF_CHUNK];
#pragma unroll
            for (uint32_t j = 0; j < PERF_CHUNK; j++) r[j] = rec[j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t j = 0; j < PERF_CHUNK; j++) {
                const bool wanted = j < m && u < r[j].z;
                const T *p = wanted ? data + (((uint64_t)r[j].y << 32) | r[j].x) + u : reinterpret_cast<const T *>(idle);
                x[j] = *p;
            }
        }
#pragma unroll
        for (int k = 0; k < RMAX; k++) {
            const uint32_t j = lu + (uint32_t)k * W;
            const bool mine = (uint32_t)k < per_lane && j < m;
            const int32_t r = (mine && len[k] && rc[k]) ? rank[k] : 0;
            if ((uint32_t)k < per_lane && j < PERF_CHUNK) nrk[j] = mine ? (r < 0 ? 0 - r : 1) : 0;
        }
        __syncthreads();
        PERF_STAMP(2);
        uint32_t ls[PERF_CHUNK];
        int32_t ws[PERF_CHUNK];
#pragma unroll
        for (uint32_t j = 0; j < PERF_CHUNK; j++) { ls[j] = rec[j].z; ws[j] = nrk[j]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (uint32_t j = 0; j < PERF_CHUNK; j++) {
            if (j < m) {
                const uint32_t l = ls[j];
                const int32_t nrank = ws[j];
                total += l ? (uint32_t)nrank : 0u;
                longest = l > longest ? l : longest;
                last = l;
                if (u < l) acc = perf_fold<OP, T>(acc, x[j], raw_in ? 1 : nrank);
            }
        }
    }
#ifdef MRPDR_TILE_STAMPS
    if (__builtin_amdgcn_readfirstlane((int)(acc != acc)) == 12345) return;      // (the fold's result is needed in front of the stamp)
#endif
    PERF_STAMP(3);
    const bool valid = node < a.nodes;
    if (raw_in && valid && i1 > i0) last = a.in_lasts[i1 - 1];
    uint64_t at_node = node;
    if (a.fuse) {
        // The workgroup's nodes are finished as outputs of their own level (divided unless raw), parked
        // in LDS, and the lanes of its first node fold them in order as the ONE node of the level above.
        __shared__ T s_acc2[MRPDR_TPB];
        __shared__ uint32_t s_len2[MRPDR_TPB / PERF_MIN_W], s_tot2[MRPDR_TPB / PERF_MIN_W];
        __shared__ uint64_t s_last2[MRPDR_TPB / PERF_MIN_W];
        const bool raw1 = a.raw_mid != 0;
        if (valid && !raw1 && OP == MRPDR_AVG && last && u < last) acc = acc / (T)total;
        if (longest > a.out.stride) {
            if (u == 0) atomicOr(a.out.status, 1u);
            longest = 0;
        }
        s_acc2[t] = acc;
        if (lu == 0) { s_len2[ln] = valid ? (uint32_t)longest : 0u; s_tot2[ln] = total; s_last2[ln] = last; }
        __syncthreads();
        if (ln != 0) return;
        const uint32_t children = (uint32_t)(a.nodes - node0 < npb ? a.nodes - node0 : npb);
        __builtin_memset(&acc, OP == MRPDR_MIN ? 0xFF : 0, sizeof acc);
        total = 0; longest = 0; last = 0;
        // (everything out of LDS first -- a workgroup holds 32 nodes at most -- then the chain)
        constexpr uint32_t CMAX = MRPDR_TPB / PERF_MIN_W;
        uint32_t cl[CMAX], ct[CMAX];
        uint64_t ck[CMAX];
        T cx[CMAX];
#pragma unroll
        for (uint32_t j = 0; j < CMAX; j++) {
            const uint32_t jj = j < children ? j : 0u;
            cl[j] = s_len2[jj]; ct[j] = s_tot2[jj]; ck[j] = s_last2[jj];
            cx[j] = s_acc2[jj * W + lu];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (uint32_t j = 0; j < CMAX; j++) {
            if (j < children) {
                const uint32_t l = cl[j];
                const int32_t nrank = (int32_t)ct[j];           // the child's rank is -total: it weighs `total` ranks
                total += l ? (uint32_t)nrank : 0u;
                longest = l > longest ? l : longest;
                last = raw1 ? ck[j] : l;
                if (u < l) acc = perf_fold<OP, T>(acc, cx[j], raw1 ? 1 : nrank);
            }
        }
        at_node = blockIdx.y;
        PERF_STAMP(4);
    } else if (!valid) return;
    if (!raw_out && OP == MRPDR_AVG && last && u < last) acc = acc / (T)total;
    if (longest > a.out.stride) {                   // the result does not fit: say so, write nothing
        if (u == 0) atomicOr(a.out.status, 1u);
        longest = 0;
    }
    if (u < longest) reinterpret_cast<T *>(a.out.data)[at_node * a.out.stride + u] = acc;
    if (u == 0) {
        a.out.lens[at_node] = longest;
        a.out.ranks[at_node] = 0 - (int32_t)total;
        if (a.out.offs) a.out.offs[at_node] = at_node * a.out.stride;
        if (a.out.rank_counts) { a.out.rank_counts[at_node] = 1; a.out.rank_offs[at_node] = at_node; }
        if (raw_out) a.out_lasts[at_node] = last;
        if (a.out_packets) *a.out_packets = 1u;
    }
}

// The raw nodes of consecutive packet ranges of one wave (one per GPU; g of them, `stride` bytes apart,
// header first), in range order, into the output packet.  Integer data: the accumulators are added up /
// compared (any grouping gives the one invocation's bits), the ranks summed, AVG divided once.  Double
// data arrives relayed -- every range continued the fold from the state in front of it -- so the caller
// passes the last state alone and this is its division and its way into the packet's form.
template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_perf_combine(const uint8_t *states, uint64_t g, uint64_t stride, PerfOut out, int32_t relayed)
{
    const uint64_t u = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    T acc;
    __builtin_memset(&acc, OP == MRPDR_MIN ? 0xFF : 0, sizeof acc);
    uint32_t total = 0;
    uint64_t longest = 0, last = 0;
    for (uint64_t s = 0; s < g; s++) {
        const PerfState h = *reinterpret_cast<const PerfState *>(states + s * stride);
        if (!h.packets) continue;
        total += (uint32_t)(0 - h.rank);
        longest = h.longest > longest ? h.longest : longest;
        last = h.last;
        if (u < h.longest) {
            const T x = reinterpret_cast<const T *>(states + s * stride + sizeof(PerfState))[u];
            // (a relayed state IS the accumulator, start value and all: a MAX that a NaN restarted may lie below 0)
            acc = relayed ? x : perf_fold<OP, T>(acc, x, 1);
        }
    }
    if (OP == MRPDR_AVG && last && u < last) acc = acc / (T)total;
    if (longest > out.stride) {
        if (u == 0) atomicOr(out.status, 1u);
        longest = 0;
    }
    if (u < longest) reinterpret_cast<T *>(out.data)[u] = acc;
    if (u == 0) {
        out.lens[0] = longest;
        out.ranks[0] = 0 - (int32_t)total;
    }
}

// ---- one invocation over doubles: the elements' chains side by side ----------------------------------
// Doubles round at every addition, so one invocation of SUM / AVG is a chain per element -- 131,072
// dependent steps for a GPU's shard.  The exact chain has a fast form (flatwide.h), and when every
// packet carries the same number of elements, end to end in the values column (what a stream of one
// metric's performance data looks like), element u of all packets IS a strided column: the elements'
// chains are taken by the device-wide fold side by side (REGULAR).  A wave of another form -- arrays
// of different lengths, packets without data -- is first copied into that form (RAGGED): a dense
// n x cap matrix in which a packet's missing elements are +0.0.  Adding +0.0 (or +0.0 times a rank
// count) changes no bit of a sum that started from +0.0 (such a sum is never -0.0: only -0.0 + -0.0
// gives it), so the chains over the dense columns are the reference's chains over the packets that do
// carry the element; MIN / MAX, which have no neutral value (any bits may be data), skip the missing
// elements by the packets' lengths.  Which form the wave has is found out on the device: k_perf_shape
// checks it and leaves the weights (nrank, 0 for a packet without data) as an int32 column; the serial
// k_perfdata runs only where neither form applies (each launch looks at shape[0] and leaves at once
// when it is not its turn).
//   shape[0] 0: another path has the wave, 1: REGULAR, 2: RAGGED     shape[1] columns
//   shape[2] first element of column 0    shape[3] total ranks of these packets
//   shape[4] longest array of these packets    shape[5] data length of the last packet
//   shape[6] elements from one packet to the next within a column    shape[7] 1: the columns are in the dense copy
//   shape[8] (low word) set by any packet that breaks the regular form
#define PERF_SHAPE_WORDS 16
#define PERF_SHAPE_BLOCKS 256u          // at most: every workgroup ends in three atomics on the same words
__global__ __launch_bounds__(MRPDR_TPB) void k_perf_shape(PerfIn in, uint64_t cap, uint64_t *shape, int32_t *nranks)
{
    __shared__ uint32_t s_ranks[MRPDR_TPB / 64];
    __shared__ unsigned long long s_len[MRPDR_TPB / 64];
    __shared__ uint32_t s_bad[MRPDR_TPB / 64];
    const uint64_t k = in.data_counts[0], first = in.data_offs[0];
    bool bad = k == 0 || k > cap;
    uint32_t nrank_sum = 0;
    uint64_t longest = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x; i < in.n; i += (uint64_t)gridDim.x * MRPDR_TPB) {
        const uint64_t len = in.data_counts[i];
        bad = bad || len != k || in.data_offs[i] != first + i * k;
        int32_t rank = 0;
        if (in.rank_counts[i]) rank = in.rank_vals[in.rank_offs[i]];
        const uint32_t nrank = len ? (uint32_t)(rank < 0 ? 0 - rank : 1) : 0u;         // a packet without data is nobody's rank
        nranks[i] = (int32_t)nrank;
        nrank_sum += nrank;
        longest = len > longest ? len : longest;
        if (i == in.n - 1) shape[5] = len;
    }
    // the rank total, the longest array and the verdict: wavefront, workgroup, then one atomic each
    // (one per wavefront, all on the same words, took 50 us for 131,072 packets)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        nrank_sum += __shfl_xor(nrank_sum, d, 64);
        const uint64_t o = __shfl_xor((unsigned long long)longest, d, 64);
        longest = o > longest ? o : longest;
    }
    const bool any_bad = __ballot(bad) != 0;
    if ((threadIdx.x & 63u) == 0) { s_ranks[threadIdx.x >> 6] = nrank_sum; s_len[threadIdx.x >> 6] = longest; s_bad[threadIdx.x >> 6] = any_bad ? 1u : 0u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = 0, b = 0;
        unsigned long long l = 0;
        for (uint32_t w = 0; w < MRPDR_TPB / 64; w++) { r += s_ranks[w]; l = s_len[w] > l ? s_len[w] : l; b |= s_bad[w]; }
        atomicAdd(reinterpret_cast<unsigned long long *>(&shape[3]), (unsigned long long)r);
        atomicMax(reinterpret_cast<unsigned long long *>(&shape[4]), l);
        if (b) atomicOr(reinterpret_cast<uint32_t *>(&shape[8]), 1u);
        if (blockIdx.x == 0) { shape[1] = k; shape[2] = first; }
    }
}

// (a launch of its own: every workgroup of k_perf_shape has to have spoken)
// front (a state handed over by whoever folded the packets in front, or NULL): REGULAR columns go on from
// it only if it has the same form -- every element covered by every packet so far, k of them; RAGGED
// columns go on from any state that fits the bound.  Its accumulators become the columns' front nodes
// (fronts, if given: SUM / AVG; MIN / MAX make theirs in k_perf_minmax_fronts).
__global__ void k_perf_shape_verdict(uint64_t *shape, const uint8_t *front, Partial *fronts, uint64_t cap)
{
    const uint64_t k = shape[1], first = shape[2], longest = shape[4];
    bool regular = (uint32_t)shape[8] == 0;
    PerfState h = { 0, 0, 0, 0, 0 };
    if (front) h = *reinterpret_cast<const PerfState *>(front);
    if (!h.packets) { h.longest = 0; h.last = 0; h.rank = 0; }
    if (h.packets && (h.longest != k || h.last != k)) regular = false;
    const uint64_t cols = regular ? k : (h.longest > longest ? h.longest : longest);
    const bool ragged = !regular && longest >= 1 && cols <= cap;
    __syncthreads();
    if ((regular || ragged) && fronts && threadIdx.x < cols) {
        const bool has = threadIdx.x < h.longest;
        Partial p;
        p.bits = has ? reinterpret_cast<const uint64_t *>(front + sizeof(PerfState))[threadIdx.x] : 0;
        p.cnt = has ? 0 - h.rank : 0;
        p.present = has ? 1 : 0;
        fronts[threadIdx.x] = p;
    }
    if (threadIdx.x == 0) {
        shape[0] = regular ? 1u : (ragged ? 2u : 0u);
        shape[1] = cols;
        shape[2] = regular ? first : 0;
        shape[6] = regular ? k : cap;
        shape[7] = regular ? 0u : 1u;
    }
}

// RAGGED: the packets' arrays into the rows of a dense n x cap matrix, +0.0 where a packet has no element
__global__ __launch_bounds__(MRPDR_TPB) void k_perf_densify(PerfIn in, const uint64_t *shape, uint64_t cap, uint64_t *dense)
{
    if (shape[0] != 2) return;
    const uint64_t all = in.n * cap;
    for (uint64_t idx = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x; idx < all; idx += (uint64_t)gridDim.x * MRPDR_TPB) {
        const uint64_t i = idx / cap, u = idx - i * cap;
        const uint64_t len = in.data_counts[i], off = in.data_offs[i];
        dense[idx] = u < len ? reinterpret_cast<const uint64_t *>(in.data_vals)[off + u] : 0;
    }
}

// the columns' nodes behind a packet range into the range's state (struct mrpdr_perf_state + accumulators)
__global__ void k_perf_state_from_columns(const uint64_t *shape, const Partial *nodes, const uint8_t *front, uint8_t *state)
{
    if (!shape[0]) return;
    const uint64_t cols = shape[1];
    if (threadIdx.x < cols) reinterpret_cast<uint64_t *>(state + sizeof(PerfState))[threadIdx.x] = nodes[threadIdx.x].bits;
    if (threadIdx.x == 0) {
        uint32_t total = (uint32_t)shape[3];
        if (front) {
            const PerfState h = *reinterpret_cast<const PerfState *>(front);
            if (h.packets) total += (uint32_t)(0 - h.rank);
        }
        PerfState out;
        out.longest = cols; out.last = shape[5]; out.rank = 0 - (int32_t)total; out.packets = 1; out.reserved = 0;
        *reinterpret_cast<PerfState *>(state) = out;
    }
}

// the columns' nodes behind a whole wave into the output packet: AVG's one division, over the elements
// the last packet carries (src/FilterDefinitions.C:1114-1132), by the unsigned rank total
template <int OP>
__global__ __launch_bounds__(MRPDR_TPB) void k_perf_from_columns(const uint64_t *shape, const Partial *nodes, PerfOut out)
{
    if (!shape[0]) return;
    const uint64_t cols = shape[1], last = shape[5], u = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    const uint32_t total = (uint32_t)shape[3];
    if (cols > out.stride) {
        if (u == 0) { atomicOr(out.status, 1u); out.lens[0] = 0; out.ranks[0] = 0 - (int32_t)total; }
        return;
    }
    if (u < cols) {
        double acc = from_bits<double>(nodes[u].bits);
        if (OP == MRPDR_AVG && last && u < last) acc = acc / (double)total;
        reinterpret_cast<double *>(out.data)[u] = acc;
    }
    if (u == 0) { out.lens[0] = cols; out.ranks[0] = 0 - (int32_t)total; }
}

// MIN / MAX of the same double waves: the comparison keeps the later operand whenever it is false,
// so a NaN restarts the fold -- the (value, restarted) nodes of OP_MINR / OP_MAXR carry that through a
// tree, and the elements' trees run side by side (blockIdx.y = element).  tfilter_PerfData's MAX starts
// from 0 rather than from the first packet (src/FilterDefinitions.C:961-965): that 0 is the front node.
template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_tree_tile_multi(FilterIn in, Partial *partials, uint64_t pstride, WideMulti m)
{
    __shared__ Partial wave_part[MRPDR_TPB / 64];
    if (!wide_multi_column<T>(m, in)) return;
    T vals[FILTER_LEAF];
    int32_t cnts[FILTER_LEAF];
    int nvalid = tile_load<T, false>(in, vals, cnts);
    if (m.shape[7]) {
        // RAGGED: only the packets that carry element blockIdx.y take part -- the lane's values move up,
        // in order, over the ones that do not (selects, no indexed registers)
        const uint64_t first = (uint64_t)blockIdx.x * FILTER_TILE + (uint64_t)threadIdx.x * FILTER_LEAF;
        uint64_t lens[FILTER_LEAF];
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) lens[k] = m.row_counts[first + k < in.n ? first + k : in.n - 1];
        T kept[FILTER_LEAF];
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) kept[k] = T();
        int nkept = 0;
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) {
            const bool has = k < nvalid && blockIdx.y < lens[k];
#pragma unroll
            for (int j = 0; j < FILTER_LEAF; j++) kept[j] = (has && j == nkept) ? vals[k] : kept[j];
            nkept += has ? 1 : 0;
        }
#pragma unroll
        for (int k = 0; k < FILTER_LEAF; k++) vals[k] = kept[k];
        nvalid = nkept;
    }
#pragma unroll
    for (int k = 0; k < FILTER_LEAF; k++) cnts[k] = vals[k] != vals[k];
    const Partial out = tile_node<OP, T>(vals, cnts, nvalid, wave_part);
    if (threadIdx.x == 0) partials[blockIdx.y * pstride + blockIdx.x] = out;
}

template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_tree_top_multi(const Partial *partials, uint64_t np, uint64_t pstride, uint8_t *results,
                                                              int32_t *oks, WideMulti m)
{
    __shared__ Partial stage[FLAT_TILE];
    if (!m.shape[0] || blockIdx.y >= m.shape[1]) return;
    tree_top<OP, T>(partials + blockIdx.y * pstride, np, results + 16u * blockIdx.y, oks + blockIdx.y, stage, wide_multi_link(m));
}

// the front nodes of MIN / MAX columns: the state in front if there is one, else MAX's start value 0
// (MIN's start, a NaN, is the same as no front at all: the first comparison discards it)
template <int OP>
__global__ void k_perf_minmax_fronts(const uint64_t *shape, const uint8_t *front, Partial *fronts)
{
    if (!shape[0] || threadIdx.x >= shape[1]) return;
    Partial p;
    p.bits = 0; p.cnt = 0; p.present = OP == MRPDR_MAX ? 1 : 0;
    if (front) {
        const PerfState h = *reinterpret_cast<const PerfState *>(front);
        if (h.packets && threadIdx.x < h.longest) { p.bits = reinterpret_cast<const uint64_t *>(front + sizeof(PerfState))[threadIdx.x]; p.present = 1; }
    }
    fronts[threadIdx.x] = p;
}
