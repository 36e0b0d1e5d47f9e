// Streaming-copy microbenchmark: what the payload kernel can hope for on this box.
// Variants: load alignment, unroll, nontemporal hints, block count, byte swap.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool NT, bool SWAP, int MIS>
__global__ __launch_bounds__(256) void k_stream(const uint8_t *in, uint8_t *out, uint64_t bytes)
{
    const uint64_t CH = 256ull * 16 * U;
    const uint64_t nch = bytes / CH;
    const uint64_t per = (nch + gridDim.x - 1) / gridDim.x;
    uint64_t c0 = blockIdx.x * per, c1 = c0 + per < nch ? c0 + per : nch;
    for (uint64_t c = c0; c < c1; c++) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint8_t *p = in + MIS + c * CH + ((uint64_t)u * 256 + threadIdx.x) * 16;
            if (NT) { v[u] = __builtin_nontemporal_load((const u32x4 *)p); }
            else { __builtin_memcpy(&v[u], p, 16); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            u32x4 w = v[u];
            if (SWAP) for (int k = 0; k < 4; k++) w[k] = __builtin_bswap32(w[k]);
            u32x4 *q = (u32x4 *)(out + c * CH + ((uint64_t)u * 256 + threadIdx.x) * 16);
            if (NT) __builtin_nontemporal_store(w, q); else *q = w;
        }
    }
}

// interleaved grid-stride version (chunk c handled by block c % grid)
template <int U, bool NT, int MIS>
__global__ __launch_bounds__(256) void k_stream_gs(const uint8_t *in, uint8_t *out, uint64_t bytes)
{
    const uint64_t CH = 256ull * 16 * U;
    const uint64_t nch = bytes / CH;
    for (uint64_t c = blockIdx.x; c < nch; c += gridDim.x) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint8_t *p = in + MIS + c * CH + ((uint64_t)u * 256 + threadIdx.x) * 16;
            if (NT) { v[u] = __builtin_nontemporal_load((const u32x4 *)p); }
            else { __builtin_memcpy(&v[u], p, 16); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            u32x4 *q = (u32x4 *)(out + c * CH + ((uint64_t)u * 256 + threadIdx.x) * 16);
            if (NT) __builtin_nontemporal_store(v[u], q); else *q = v[u];
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
#define RUN(name, kern, grid) { float ms = timeit([&] { kern<<<grid, 256>>>(in, out, bytes); }, 20); \
    printf("%-44s grid %5d  %.4f ms  %7.1f GB/s (rd+wr)\n", name, grid, ms, 2.0 * bytes / ms / 1e6); }
    int grids[] = {1024, 2048, 4096, 8192};
    for (int g : grids) {
        RUN("aligned U4", (k_stream<4, false, false, 0>), g);
        RUN("aligned U4 swap", (k_stream<4, false, true, 0>), g);
        RUN("mis8 U4", (k_stream<4, false, false, 8>), g);
        RUN("mis4 U4", (k_stream<4, false, false, 4>), g);
        RUN("mis1 U4", (k_stream<4, false, false, 1>), g);
        RUN("aligned U4 nt", (k_stream<4, true, false, 0>), g);
        RUN("aligned U8", (k_stream<8, false, false, 0>), g);
        RUN("aligned U2", (k_stream<2, false, false, 0>), g);
        RUN("aligned U1", (k_stream<1, false, false, 0>), g);
        RUN("mis8 U8", (k_stream<8, false, false, 8>), g);
        RUN("gs aligned U4", (k_stream_gs<4, false, 0>), g);
        RUN("gs aligned U4 nt", (k_stream_gs<4, true, 0>), g);
        RUN("gs mis8 U4", (k_stream_gs<4, false, 8>), g);
        RUN("gs aligned U1", (k_stream_gs<1, false, 0>), g);
    }
    { int g = 65536; RUN("gs aligned U4", (k_stream_gs<4, false, 0>), g); RUN("gs aligned U1", (k_stream_gs<1, false, 0>), g);
      g = 262144; RUN("gs aligned U1", (k_stream_gs<1, false, 0>), g); }
    float ms = timeit([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }, 20);
    printf("%-44s             %.4f ms  %7.1f GB/s (rd+wr)\n", "hipMemcpyAsync D2D", ms, 2.0 * bytes / ms / 1e6);
    return 0;
}
