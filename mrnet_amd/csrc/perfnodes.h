// tfilter_PerfData over many nodes at once: the levels of a tree of filter invocations, and the
// order-free form of ONE invocation over integer data.
//
// Reference: src/FilterDefinitions.C:904-1239 (see k_perfdata in filter.h for the fold's rules).  One
// invocation folds its packets one after the other, element by element.  k_perfdata gives a node a
// workgroup and an element a lane, which leaves 8 of 256 lanes busy on the arrays performance data
// really carries (eight doubles per packet, src/Stream.C:1086-1110) and makes a node of 256 packets a
// chain of 256 dependent steps.  Here a workgroup holds as many nodes as fit -- a node takes W lanes,
// W the array bound rounded up to a power of two -- so that the levels of a tree are short chains
// side by side: 131,072 packets as 32 x 64 x 64 are three launches whose chains are 32, 64 and 64
// steps long.
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

#define PERF_CHUNK 32u          // packets of a node whose records are staged at a time
#define PERF_BATCH 16           // packets whose values a lane has in flight together
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
};

template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_perf_nodes(PerfLevel a)
{
    // one record per staged packet: where its data lies, how long it is, what it weighs (0: no packet)
    __shared__ uint4 s_rec[(MRPDR_TPB / PERF_MIN_W) * PERF_CHUNK];
    const uint32_t t = threadIdx.x, W = a.W, npb = MRPDR_TPB / W;
    const uint32_t ln = t / W;
    const uint64_t node0 = (uint64_t)blockIdx.y * npb, node = node0 + ln;
    const uint64_t u = (uint64_t)blockIdx.x * W + (t % W);
    const bool raw_in = a.in_lasts != nullptr, raw_out = a.out_lasts != nullptr;
    const uint64_t group = a.in.group, n = a.in.n;
    const uint64_t i0 = node * group;
    const uint64_t i1 = node < a.nodes ? (i0 + group < n ? i0 + group : n) : i0;
    const T *data = reinterpret_cast<const T *>(a.in.data_vals);
    // (a load that is not wanted reads this instead: an address that is there whatever the wave holds)
    const T *idle = reinterpret_cast<const T *>(a.in.data_counts);
    T acc;
    __builtin_memset(&acc, OP == MRPDR_MIN ? 0xFF : 0, sizeof acc);
    uint32_t total = 0;
    uint64_t longest = 0, last = 0;
    const uint32_t nrec = npb * PERF_CHUNK;
    for (uint64_t c = 0; c < group; c += PERF_CHUNK) {
        __syncthreads();
        // records of this chunk, all nodes of the workgroup: the four loads of a packet are independent
        // (and unconditional: a record past its node's end re-reads packet n - 1), the rank is one more trip
        for (uint32_t r0 = 0; r0 < nrec; r0 += MRPDR_TPB) {
            const uint32_t r = r0 + t;
            const uint32_t rn = r / PERF_CHUNK, rj = r % PERF_CHUNK;
            const uint64_t gn = node0 + rn;
            const uint64_t pi = gn * group + c + rj;
            const uint64_t pend = gn < a.nodes ? (gn * group + group < n ? gn * group + group : n) : 0;
            const bool have = r < nrec && c + rj < group && pi < pend;
            const uint64_t q = have ? pi : n - 1;
            const uint64_t len = a.in.data_counts[q], off = a.in.data_offs[q];
            const uint64_t rc = a.in.rank_counts[q], ro = a.in.rank_offs[q];
            int32_t rank = 0;
            if (have && len && rc) rank = a.in.rank_vals[ro];
            const int32_t nrank = rank < 0 ? 0 - rank : 1;
            if (r < nrec) s_rec[r] = have ? make_uint4((uint32_t)off, (uint32_t)(off >> 32), (uint32_t)len, (uint32_t)nrank)
                                          : make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        const uint64_t at = i0 + c;
        const uint32_t m = at < i1 ? (uint32_t)(i1 - at < PERF_CHUNK ? i1 - at : PERF_CHUNK) : 0u;
        const uint4 *rec = s_rec + ln * PERF_CHUNK;
        for (uint32_t j0 = 0; j0 < m; j0 += PERF_BATCH) {
            uint4 r[PERF_BATCH];
            T x[PERF_BATCH];
#pragma unroll
            for (int k = 0; k < PERF_BATCH; k++) {
                r[k] = rec[j0 + k < m ? j0 + k : m - 1];
                const bool wanted = j0 + k < m && u < r[k].z;
                const T *p = wanted ? data + (((uint64_t)r[k].y << 32) | r[k].x) + u : idle;
                x[k] = *p;
            }
#pragma unroll
            for (int k = 0; k < PERF_BATCH; k++) {
                if (j0 + k < m) {
                    const uint32_t len = r[k].z;
                    const int32_t nrank = (int32_t)r[k].w;
                    total += len ? (uint32_t)nrank : 0u;
                    longest = len > longest ? len : longest;
                    last = len;
                    if (u < len) acc = perf_fold<OP, T>(acc, x[k], raw_in ? 1 : nrank);
                }
            }
        }
    }
    if (node >= a.nodes) return;
    if (raw_in && i1 > i0) last = a.in_lasts[i1 - 1];
    if (!raw_out && OP == MRPDR_AVG && last && u < last) acc = acc / (T)total;
    if (longest > a.out.stride) {                   // the result does not fit: say so, write nothing
        if (u == 0) atomicOr(a.out.status, 1u);
        longest = 0;
    }
    if (u < longest) reinterpret_cast<T *>(a.out.data)[node * a.out.stride + u] = acc;
    if (u == 0) {
        a.out.lens[node] = longest;
        a.out.ranks[node] = 0 - (int32_t)total;
        if (a.out.offs) a.out.offs[node] = node * a.out.stride;
        if (a.out.rank_counts) { a.out.rank_counts[node] = 1; a.out.rank_offs[node] = node; }
        if (raw_out) a.out_lasts[node] = last;
        if (a.out_packets) *a.out_packets = 1u;
    }
}

// The raw nodes of consecutive packet ranges of one wave (one per GPU; g of them, `stride` bytes apart,
// header first), in range order, into the output packet.  Integer data: the accumulators are added up /
// compared (any grouping gives the one invocation's bits), the ranks summed, AVG divided once.  Double
// data arrives relayed -- every range continued the fold from the state in front of it -- so the caller
// passes the last state alone and this is its division and its way into the packet's form.
template <int OP, typename T>
__global__ __launch_bounds__(MRPDR_TPB) void k_perf_combine(const uint8_t *states, uint64_t g, uint64_t stride, PerfOut out)
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
        if (u < h.longest) acc = perf_fold<OP, T>(acc, reinterpret_cast<const T *>(states + s * stride + sizeof(PerfState))[u], 1);
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
