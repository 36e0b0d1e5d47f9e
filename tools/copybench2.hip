// Second microbenchmark: grid-stride streaming with nontemporal hints and a prefetched
// per-chunk descriptor (the structure considered for the payload kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Desc { uint64_t wire_off; uint32_t seg; uint32_t single; };

template <int U, bool NTL, bool NTS, bool DESC>
__global__ __launch_bounds__(256) void k_gs(const uint8_t *in, uint8_t *out, uint64_t bytes, const Desc *desc)
{
    const uint64_t CH = 256ull * 16 * U;
    const uint64_t nch = bytes / CH;
    uint64_t c = blockIdx.x;
    Desc d; d.wire_off = 0; d.seg = 0; d.single = 1;
    if (DESC && c < nch) d = desc[c];
    for (; c < nch; c += gridDim.x) {
        Desc nd = d;
        if (DESC && c + gridDim.x < nch) nd = desc[c + gridDim.x];
        const uint8_t *src = DESC ? in + d.wire_off : in + c * CH;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint8_t *p = src + ((uint64_t)u * 256 + threadIdx.x) * 16;
            if (NTL) v[u] = __builtin_nontemporal_load((const u32x4 *)p);
            else __builtin_memcpy(&v[u], p, 16);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            u32x4 w = v[u];
            for (int k = 0; k < 4; k++) w[k] = __builtin_bswap32(w[k]);
            u32x4 *q = (u32x4 *)(out + c * CH + ((uint64_t)u * 256 + threadIdx.x) * 16);
            if (NTS) __builtin_nontemporal_store(w, q); else *q = w;
        }
        d = nd;
    }
}

template <int U> __global__ void k_mkdesc(Desc *d, uint64_t nch, uint64_t mis)
{
    uint64_t c = blockIdx.x * 256ull + threadIdx.x;
    if (c < nch) { d[c].wire_off = c * 256ull * 16 * U + mis; d[c].seg = (uint32_t)c; d[c].single = 1; }
}

__global__ void k_fill_random(uint32_t *p, uint64_t n)
{
    uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    for (; i < n; i += (uint64_t)gridDim.x * 256) {
        uint64_t z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
        p[i] = (uint32_t)(z >> 16);
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
    uint8_t *in, *out; Desc *desc;
    CK(hipMalloc(&in, bytes + 4096)); CK(hipMalloc(&out, bytes + 4096)); CK(hipMalloc(&desc, (bytes / 4096) * sizeof(Desc)));
    CK(hipMemset(in, 1, bytes + 4096)); CK(hipMemset(out, 0, bytes + 4096));
#define RUN(name, U, NTL, NTS, DESC, grid) { \
    uint64_t nch = bytes / (4096ull * U); \
    if (DESC) { k_mkdesc<U><<<(nch + 255) / 256, 256>>>(desc, nch, 8); CK(hipDeviceSynchronize()); } \
    int gg = (grid) == 0 ? (int)nch : (grid); \
    float ms = timeit([&] { k_gs<U, NTL, NTS, DESC><<<gg, 256>>>(in, out, bytes, desc); }, 20); \
    printf("%-36s U%d grid %6d  %.4f ms  %7.1f GB/s\n", name, U, gg, ms, 2.0 * bytes / ms / 1e6); }
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) { k_fill_random<<<4096, 256>>>((uint32_t *)in, bytes / 4); CK(hipDeviceSynchronize()); printf("-- random data --\n"); }
        else printf("-- constant data --\n");
        const int g = 0;
        RUN("nt ld+st", 1, true, true, false, g);
        RUN("nt ld+st", 2, true, true, false, g);
        RUN("nt ld+st", 4, true, true, false, g);
        RUN("nt ld+st", 8, true, true, false, g);
        RUN("nt ld only", 4, true, false, false, g);
        RUN("nt st only", 4, false, true, false, g);
        RUN("nt ld+st desc(mis8)", 1, true, true, true, g);
        RUN("nt ld+st desc(mis8)", 2, true, true, true, g);
        RUN("nt ld+st desc(mis8)", 4, true, true, true, g);
        RUN("plain desc(mis8)", 4, false, false, true, g);
    }
    return 0;
}
