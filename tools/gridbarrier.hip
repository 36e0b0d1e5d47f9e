// Microbenchmark: what a grid-wide barrier costs on this part when every workgroup of the launch is
// resident (1,024 workgroups of 256 threads with 24 KiB of LDS each -- five per CU -- the shape of the
// short-packet tile decode at 65,536 packets), against a kernel boundary.  A one-launch decode would
// need two such barriers (sizing -> offsets -> payloads).
//
// Barrier: one arrival counter per group of GROUP workgroups, a root counter the last arrival of a
// group bumps, and a generation word everybody spins on (device-scope atomics; release on arrival,
// acquire on the generation).  Every spin is BOUNDED: a workgroup that does not see the generation
// move within SPIN_LIMIT polls sets an error flag and leaves, so the grid drains whatever happens.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define SPIN_LIMIT 2000000u

struct Bar { unsigned int leaf[64 * 32]; unsigned int root; unsigned int gen; unsigned int err; unsigned int pad[29]; unsigned int gens[64 * 32]; };   // (leaf / gens: one 128-byte line per group)

// RELAXED: the polls are relaxed loads (no cache invalidation per poll), one acquire fence when the
// generation has moved.
template <int GROUP, bool SPREAD, bool RELAXED = false>
__device__ inline bool grid_barrier(Bar *b, unsigned int nblocks, unsigned int target_gen)
{
    __syncthreads();
    bool fine = true;
    if (threadIdx.x == 0) {
        const unsigned int g = blockIdx.x / GROUP, ngroups = (nblocks + GROUP - 1) / GROUP;
        const unsigned int in_group = (g + 1) * GROUP <= nblocks ? GROUP : nblocks - g * GROUP;
        __threadfence();
        const unsigned int at = __hip_atomic_fetch_add(&b->leaf[g * 32], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if ((at + 1) % in_group == 0) {                         // the group's last arrival goes on to the root
            const unsigned int r = __hip_atomic_fetch_add(&b->root, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if ((r + 1) % ngroups == 0) {
                // SPREAD: every group polls a word of its own (a line of its own), the last arrival writes them all
                if (SPREAD) for (unsigned int k = 0; k < ngroups; k++) __hip_atomic_store(&b->gens[k * 32], target_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(&b->gen, target_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned int spins = 0;
        const unsigned int *word = SPREAD ? &b->gens[g * 32] : &b->gen;
        if (RELAXED) {
            while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target_gen) {
                if (++spins > SPIN_LIMIT) { b->err = 1; fine = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        } else {
            while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target_gen) {
                if (++spins > SPIN_LIMIT) { b->err = 1; fine = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    return fine;
}

// `rounds` barriers in a row, a little work in between (one load and one store per thread)
template <int GROUP, bool SPREAD, bool RELAXED>
__global__ __launch_bounds__(256) void k_barriers(Bar *b, unsigned int nblocks, unsigned int base_gen, int rounds, const uint32_t *in, uint32_t *out)
{
    __shared__ uint8_t image[24576];
    image[threadIdx.x] = (uint8_t)in[blockIdx.x * 256 + threadIdx.x];
    for (int r = 0; r < rounds; r++) {
        if (!grid_barrier<GROUP, SPREAD, RELAXED>(b, nblocks, base_gen + (unsigned int)r + 1)) break;
        out[blockIdx.x * 256 + threadIdx.x] = image[(threadIdx.x + r) & 255] + (uint32_t)r;
    }
}

__global__ __launch_bounds__(256) void k_plain(const uint32_t *in, uint32_t *out, int r)
{
    __shared__ uint8_t image[24576];
    image[threadIdx.x] = (uint8_t)in[blockIdx.x * 256 + threadIdx.x];
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = image[(threadIdx.x + r) & 255] + (uint32_t)r;
}

template <typename F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int GROUP, bool SPREAD, bool RELAXED = false> static void run(Bar *bar, unsigned int nblocks, const uint32_t *in, uint32_t *out)
{
    unsigned int gen = 0;
    CK(hipMemset(bar, 0, sizeof(Bar)));
    float t[3];
    const int rounds[3] = { 0, 2, 10 };
    for (int k = 0; k < 3; k++) {
        const int r = rounds[k];
        t[k] = timeit([&] { k_barriers<GROUP, SPREAD, RELAXED><<<nblocks, 256>>>(bar, nblocks, gen, r, in, out); gen += (unsigned int)r; }, 50);
    }
    Bar h;
    CK(hipMemcpy(&h, bar, sizeof h, hipMemcpyDeviceToHost));
    printf("groups of %4d%s: launch alone %.2f us, with 2 barriers %.2f us, with 10 barriers %.2f us -> %.2f us per barrier%s\n", GROUP, SPREAD ? (RELAXED ? ", a poll word per group, relaxed polls" : ", a poll word per group") : (RELAXED ? ", relaxed polls" : ""),
           t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, (t[2] - t[0]) * 1e3 / 10, h.err ? "  (A SPIN RAN OUT: barrier failed)" : "");
}

int main()
{
    const unsigned int nblocks = 1024;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_barriers<32, false, false>, 256, 0));
    printf("%d CUs, %d workgroups of the barrier kernel fit a CU: %d resident at once (%u needed)\n", prop.multiProcessorCount, per_cu,
           per_cu * prop.multiProcessorCount, nblocks);
    if ((unsigned int)(per_cu * prop.multiProcessorCount) < nblocks) { printf("not all workgroups would be resident: not run\n"); return 1; }
    Bar *bar; uint32_t *in, *out;
    CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMalloc(&in, nblocks * 256 * 4)); CK(hipMalloc(&out, nblocks * 256 * 4));
    CK(hipMemset(in, 1, nblocks * 256 * 4));
    run<1024, false>(bar, nblocks, in, out);
    run<32, false>(bar, nblocks, in, out);
    run<128, true>(bar, nblocks, in, out);
    run<32, true>(bar, nblocks, in, out);
    run<16, true>(bar, nblocks, in, out);
    run<1024, false, true>(bar, nblocks, in, out);
    run<32, false, true>(bar, nblocks, in, out);
    run<128, true, true>(bar, nblocks, in, out);
    const float tp = timeit([&] { k_plain<<<nblocks, 256>>>(in, out, 1); k_plain<<<nblocks, 256>>>(in, out, 2); k_plain<<<nblocks, 256>>>(in, out, 3); }, 50);
    const float t1 = timeit([&] { k_plain<<<nblocks, 256>>>(in, out, 1); }, 50);
    printf("three dependent launches of the same work: %.2f us, one: %.2f us -> %.2f us per kernel boundary\n", tp * 1e3, t1 * 1e3, (tp - t1) * 1e3 / 2);
    return 0;
}
