// tfilter_IntEqClass (src/FilterDefinitions.C:812-902): packets "%aud %aud %aud" = (class values,
// members per class, members).  The reference pours every (value -> members) pair of every packet
// into a std::map<unsigned, std::vector<unsigned>> and emits the map: classes by ascending value,
// the members of a class in the order they arrived (packet order, then class order in a packet).
//
// Here: every class entry of the wave gets the key (value << 32 | arrival index); sorting the keys
// orders entries by value and, inside a value, by arrival -- what the map does -- without needing a
// stable sort.  A bitonic network sorts the keys (this filter is rare and small: simplicity over
// speed), then flags / scans turn runs of equal values into classes and a segmented copy gathers
// the members.
#pragma once

#include "common.h"
#include "scan.h"

#define EQ_LOCAL 2048                 // keys one workgroup sorts in LDS (8 per thread)

// keys[i] = value << 32 | i for the K = vals_offs[n] entries, all-ones padding up to kpad;
// src_off is filled by a scan of the member counts afterwards.  An entry without members leaves no
// trace in the reference's map (the value is inserted by the first push_back, :841-845): it gets the
// padding key and sorts out of the way with the padding.
__global__ __launch_bounds__(MRPDR_TPB) void k_eq_keys(const uint32_t *vals, const uint64_t *vals_offs, uint64_t n,
                                                       uint64_t kpad, uint64_t *keys, const uint32_t *memcnts,
                                                       uint64_t *cnt64)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i >= kpad) return;
    const uint64_t k = vals_offs[n];
    const uint32_t c = i < k ? memcnts[i] : 0;
    keys[i] = c ? ((uint64_t)vals[i] << 32 | i) : ~0ull;
    cnt64[i] = c;
}

// one compare-exchange step of the bitonic network over global memory (partner distance j >= EQ_LOCAL / 2)
__global__ __launch_bounds__(MRPDR_TPB) void k_eq_bitonic_global(uint64_t *keys, uint64_t kpad, uint64_t j, uint64_t k)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    const uint64_t p = i ^ j;
    if (i >= kpad || p <= i) return;
    const uint64_t a = keys[i], b = keys[p];
    const bool up = (i & k) == 0;
    if ((a > b) == up) { keys[i] = b; keys[p] = a; }
}

// all steps with partner distance < EQ_LOCAL of merge size k (k >= EQ_LOCAL: the tail of a merge;
// k < EQ_LOCAL is handled by first_k..: the whole network up to EQ_LOCAL) inside LDS
__global__ __launch_bounds__(MRPDR_TPB) void k_eq_bitonic_local(uint64_t *keys, uint64_t kpad, uint64_t k_first, uint64_t k_last)
{
    __shared__ uint64_t s[EQ_LOCAL];
    const uint64_t base = (uint64_t)blockIdx.x * EQ_LOCAL;
    for (uint32_t t = threadIdx.x; t < EQ_LOCAL; t += MRPDR_TPB) s[t] = base + t < kpad ? keys[base + t] : ~0ull;
    __syncthreads();
    for (uint64_t k = k_first; k <= k_last; k <<= 1) {
        for (uint64_t j = (k >> 1) < EQ_LOCAL ? (k >> 1) : EQ_LOCAL >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < EQ_LOCAL; t += MRPDR_TPB) {
                const uint32_t p = t ^ (uint32_t)j;
                if (p > t) {
                    const uint64_t a = s[t], b = s[p];
                    const bool up = ((base + t) & k) == 0;
                    if ((a > b) == up) { s[t] = b; s[p] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t t = threadIdx.x; t < EQ_LOCAL; t += MRPDR_TPB)
        if (base + t < kpad) keys[base + t] = s[t];
}

// sorted entry i: head[i] = 1 where a new value starts; sorted_cnt[i] = members of that entry
__global__ __launch_bounds__(MRPDR_TPB) void k_eq_heads(const uint64_t *keys, const uint64_t *vals_offs, uint64_t n,
                                                        uint64_t kpad, const uint64_t *cnt64, uint64_t *head,
                                                        uint64_t *sorted_cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i >= kpad) return;
    (void)vals_offs; (void)n;
    uint64_t h = 0, c = 0;
    if (keys[i] != ~0ull) {          // entries with members come first, sorted; the rest is padding
        h = (i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32)) ? 1 : 0;
        c = cnt64[keys[i] & 0xFFFFFFFFull];
    }
    head[i] = h;
    sorted_cnt[i] = c;
}

// class_id = exclusive scan of head (+1 at heads), dst_off = exclusive scan of sorted_cnt,
// src_off = exclusive scan of the member counts in arrival order
struct EqOut {
    uint32_t *values; uint32_t *memcnts; uint32_t *mems;
    uint64_t values_cap, mems_cap;
    uint64_t *counts[3]; uint64_t *offs[3];          // the three one-packet columns
    uint32_t *status;
};

__global__ __launch_bounds__(MRPDR_TPB) void k_eq_emit(const uint64_t *keys, const uint64_t *vals_offs, uint64_t n,
                                                       uint64_t kpad, const uint64_t *head, const uint64_t *class_scan,
                                                       const uint64_t *dst_off, const uint64_t *src_off,
                                                       const uint64_t *sorted_cnt, const uint32_t *mems,
                                                       const uint64_t *mems_offs, EqOut o)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    const uint64_t k = vals_offs[n];
    const uint64_t nclasses = class_scan[kpad], nmems = dst_off[kpad];
    const bool fits = nclasses <= o.values_cap && nmems <= o.mems_cap;
    if (i == 0) {
        if (!fits) atomicOr(o.status, 1u);
        for (int c = 0; c < 3; c++) {
            const uint64_t cnt = fits ? (c == 2 ? nmems : nclasses) : 0;
            o.counts[c][0] = cnt; o.offs[c][0] = 0; o.offs[c][1] = cnt;
        }
    }
    if (i >= k || !fits || keys[i] == ~0ull) return;
    const uint64_t cls = class_scan[i];                  // heads: index of the class this entry opens
    if (head[i]) {
        o.values[cls] = (uint32_t)(keys[i] >> 32);
        // members of the class: up to the next head
        uint64_t e = i + 1;
        while (e < k && !head[e]) e++;
        o.memcnts[cls] = (uint32_t)(dst_off[e] - dst_off[i]);
    }
    // The members of entry e of packet p start at mems_offs[p] + (members of p's entries before e):
    // the reference restarts its member index for every packet (curClassMemIdx = 0, :834), so a
    // packet whose mems array is longer than its counts add up to does not shift later packets.
    const uint64_t e = keys[i] & 0xFFFFFFFFull;
    uint64_t lo = 0, hi = n;                             // packet of entry e: vals_offs[p] <= e < vals_offs[p + 1]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (vals_offs[mid] <= e) lo = mid; else hi = mid;
    }
    const uint64_t src = mems_offs[lo] + (src_off[e] - src_off[vals_offs[lo]]), dst = dst_off[i];
    const uint64_t have = mems_offs[n];
    // counts that promise more members than the wave holds: the reference reads past its array;
    // here such a member reads as 0
    for (uint64_t m = 0; m < sorted_cnt[i]; m++) o.mems[dst + m] = src + m < have ? mems[src + m] : 0u;
}
