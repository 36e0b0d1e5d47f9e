// Third microbenchmark: cost of misaligned 128-bit STORES (the encode direction).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE 0: aligned loads, stores at +MIS.  MODE 1: loads at +MIS, aligned stores, chunk
// head/tail (16 bytes per chunk) written by lane 0 with dword stores.
template <int MIS, int MODE, bool NT>
__global__ __launch_bounds__(256) void k_st(const uint8_t *in, uint8_t *out)
{
    const uint64_t cb = (uint64_t)blockIdx.x * 16384;
    u32x4 v[4];
    if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint8_t *p = in + cb + (u * 256 + threadIdx.x) * 16;
            v[u] = NT ? __builtin_nontemporal_load((const u32x4 *)p) : *(const u32x4 *)p;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint8_t *q = out + MIS + cb + (u * 256 + threadIdx.x) * 16;
            if (NT) __builtin_nontemporal_store(v[u], (u32x4_u *)q); else *(u32x4_u *)q = v[u];
        }
    } else {
        // destination bytes [out+MIS+cb, +16384); aligned pieces start at delta = (16-MIS)%16
        const uint32_t delta = (16 - MIS) % 16;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t k = u * 256 + threadIdx.x;
            const uint8_t *p = in + cb + delta + k * 16;
            if (k < 1023 || delta == 0) v[u] = NT ? __builtin_nontemporal_load((const u32x4_u *)p) : *(const u32x4_u *)p;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t k = u * 256 + threadIdx.x;
            uint8_t *q = out + MIS + cb + delta + k * 16;
            if (k < 1023 || delta == 0) { if (NT) __builtin_nontemporal_store(v[u], (u32x4 *)q); else *(u32x4 *)q = v[u]; }
        }
        if (delta != 0 && threadIdx.x == 255) {
            for (uint32_t e = 0; e < delta; e += 4)
                *(uint32_t *)(out + MIS + cb + e) = *(const uint32_t *)(in + cb + e);
            for (uint32_t e = delta + 1023 * 16; e < 16384; e += 4)
                *(uint32_t *)(out + MIS + cb + e) = *(const uint32_t *)(in + cb + e);
        }
    }
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

int main()
{
    const uint64_t bytes = 1ull << 30;
    uint8_t *in, *out;
    CK(hipMalloc(&in, bytes + 4096)); CK(hipMalloc(&out, bytes + 4096));
    CK(hipMemset(in, 1, bytes + 4096)); CK(hipMemset(out, 0, bytes + 4096));
#define RUN(name, MIS, MODE, NT) { float ms = timeit([&] { k_st<MIS, MODE, NT><<<65536, 256>>>(in, out); }, 20); \
    printf("%-44s %.4f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); }
    for (int r = 0; r < 2; r++) {
        RUN("st aligned nt", 0, 0, true);
        RUN("st mis8 nt", 8, 0, true);
        RUN("st mis4 nt", 4, 0, true);
        RUN("st mis8 plain", 8, 0, false);
        RUN("st aligned plain", 0, 0, false);
        RUN("ld mis8, st aligned + head/tail nt", 8, 1, true);
        RUN("ld mis4, st aligned + head/tail nt", 4, 1, true);
        RUN("ld mis8, st aligned + head/tail plain", 8, 1, false);
    }
    return 0;
}
