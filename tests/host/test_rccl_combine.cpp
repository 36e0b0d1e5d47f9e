// One filter call over a wave that the GPUs of a node hold in packet ranges, from C++ with RCCL:
// one thread per visible device, ncclCommInitAll over all of them, and the only exchange the wave
// needs -- the 16-byte nodes of mrpdr_wave_filter_partial, gathered (ncclAllGather) for the operators
// whose nodes combine in any grouping, relayed in rank order (ncclSend / ncclRecv, then ncclBroadcast
// from the last rank) for float and double SUM / AVG, whose chain of additions has to be continued.
// Every rank ends with the result of ONE reference filter call over the whole wave
// (src/Filter.C:42-117 hands a node's whole wave to the filter function; the loops are
// src/FilterDefinitions.C:101-149, 237-286, 374-424, 517-561), which this program computes on the
// host with those loops and compares bit for bit.  With one device it is a communicator of one rank
// (the collectives still run through RCCL on device memory); with eight it needs no change.
// Output in the reference's test-harness style (tests/test_common.C:41-187).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mrpdr.h"

static int failures = 0;
static std::mutex print_lock;
static void end_SubTest(const char *name, int rank, bool ok)
{
    std::lock_guard<std::mutex> hold(print_lock);
    printf("%s [rank %d]: %s\n", name, rank, ok ? "MRNTEST_SUCCESS" : "MRNTEST_FAILURE");
    if (!ok) failures++;
}

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    printf("%s: %s\nMRNTEST_FAILURE\n", #call, hipGetErrorString(e_)); exit(2); } } while (0)
#define NCCL_CHECK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    printf("%s: %s\nMRNTEST_FAILURE\n", #call, ncclGetErrorString(r_)); exit(2); } } while (0)

static uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---- the reference's loops on the host, in the element's own type -------------------------------
template <typename T> static T host_sum(const std::vector<T> &v)
{
    T r = 0;
    for (T x : v) r = (T)(r + x);
    return r;
}
template <typename T> static T host_min(const std::vector<T> &v)
{
    T r = v[0];
    for (size_t i = 1; i < v.size(); i++) r = (r < v[i]) ? r : v[i];
    return r;
}
template <typename T> static T host_max(const std::vector<T> &v)
{
    T r = v[0];
    for (size_t i = 1; i < v.size(); i++) r = (r > v[i]) ? r : v[i];
    return r;
}
template <typename T> static T host_avg(const std::vector<T> &v, const std::vector<int32_t> &c, int32_t *n_out)
{
    T r = 0;
    int32_t n = 0;
    for (size_t i = 0; i < v.size(); i++) { r = (T)(r + (T)(v[i] * (T)c[i])); n += c[i]; }
    *n_out = n;
    return (T)(r / (T)n);
}

template <typename T> struct TypeOf;
template <> struct TypeOf<int32_t> { static const int code = MRPDR_INT32_T; };
template <> struct TypeOf<uint64_t> { static const int code = MRPDR_UINT64_T; };
template <> struct TypeOf<float> { static const int code = MRPDR_FLOAT_T; };
template <> struct TypeOf<double> { static const int code = MRPDR_DOUBLE_T; };

struct Rank {
    int rank, world, device;
    ncclComm_t comm;
    hipStream_t stream;
    mrpdr_ctx *ctx;
    mrpdr_node *front, *node, *nodes;       // device
    uint8_t *result; int32_t *result_ok;    // device
};

// One wave (values, counts or none) through partial -> exchange -> combine on this rank.
template <typename T>
static void sharded_filter(Rank &r, int op, const std::vector<T> &vals, const std::vector<int32_t> &cnts, uint8_t out[16], int32_t *ok)
{
    const uint64_t n = vals.size();
    const uint64_t lo = (uint64_t)r.rank * n / r.world, hi = (uint64_t)(r.rank + 1) * n / r.world, m = hi - lo;
    T *dv = nullptr;
    int32_t *dc = nullptr;
    HIP_CHECK(hipMalloc(&dv, (m ? m : 1) * sizeof(T)));
    HIP_CHECK(hipMemcpyAsync(dv, vals.data() + lo, m * sizeof(T), hipMemcpyHostToDevice, r.stream));
    if (op == MRPDR_AVG) {
        HIP_CHECK(hipMalloc(&dc, (m ? m : 1) * sizeof(int32_t)));
        HIP_CHECK(hipMemcpyAsync(dc, cnts.data() + lo, m * sizeof(int32_t), hipMemcpyHostToDevice, r.stream));
    }
    const int type = TypeOf<T>::code;
    bool fine = true;
    if (mrpdr_filter_is_chained(op, type)) {
        // the chain of additions goes through the ranks in order
        const mrpdr_node *front = nullptr;
        if (r.rank > 0) {
            NCCL_CHECK(ncclRecv(r.front, sizeof(mrpdr_node), ncclUint8, r.rank - 1, r.comm, r.stream));
            front = r.front;
        }
        fine = fine && mrpdr_wave_filter_partial(r.ctx, op, type, dv, sizeof(T), dc, 4, m, 0, front, r.node, r.stream);
        if (r.rank + 1 < r.world) NCCL_CHECK(ncclSend(r.node, sizeof(mrpdr_node), ncclUint8, r.rank + 1, r.comm, r.stream));
        NCCL_CHECK(ncclBroadcast(r.node, r.node, sizeof(mrpdr_node), ncclUint8, r.world - 1, r.comm, r.stream));
        fine = fine && mrpdr_wave_filter_combine(r.ctx, op, type, r.node, 1, r.result, r.result_ok, r.stream);
    } else {
        fine = fine && mrpdr_wave_filter_partial(r.ctx, op, type, dv, sizeof(T), dc, 4, m, 0, nullptr, r.node, r.stream);
        NCCL_CHECK(ncclAllGather(r.node, r.nodes, sizeof(mrpdr_node), ncclUint8, r.comm, r.stream));
        fine = fine && mrpdr_wave_filter_combine(r.ctx, op, type, r.nodes, (uint64_t)r.world, r.result, r.result_ok, r.stream);
    }
    if (!fine) { printf("%s\n", mrpdr_last_error()); }
    HIP_CHECK(hipMemcpyAsync(out, r.result, 16, hipMemcpyDeviceToHost, r.stream));
    HIP_CHECK(hipMemcpyAsync(ok, r.result_ok, 4, hipMemcpyDeviceToHost, r.stream));
    HIP_CHECK(hipStreamSynchronize(r.stream));
    if (!fine) *ok = -1;
    HIP_CHECK(hipFree(dv));
    if (dc) HIP_CHECK(hipFree(dc));
}

template <typename T> static bool same_bits(const uint8_t *got, T want)
{
    if (want != want) { T g; memcpy(&g, got, sizeof(T)); return g != g; }      // a NaN is a NaN
    return memcmp(got, &want, sizeof(T)) == 0;
}

static void rank_main(Rank r, uint64_t n)
{
    HIP_CHECK(hipSetDevice(r.device));
    HIP_CHECK(hipStreamCreate(&r.stream));
    r.ctx = mrpdr_ctx_create(r.device, n + 1);
    if (!r.ctx) { printf("%s\nMRNTEST_FAILURE\n", mrpdr_last_error()); exit(2); }
    HIP_CHECK(hipMalloc(&r.front, sizeof(mrpdr_node)));
    HIP_CHECK(hipMalloc(&r.node, sizeof(mrpdr_node)));
    HIP_CHECK(hipMalloc(&r.nodes, sizeof(mrpdr_node) * r.world));
    HIP_CHECK(hipMalloc(&r.result, 16));
    HIP_CHECK(hipMalloc(&r.result_ok, 4));

    // every rank draws the same wave and keeps its own range of it
    uint64_t seed = 20240607;
    std::vector<int32_t> vi(n), cnt(n);
    std::vector<uint64_t> vu(n);
    std::vector<float> vf(n);
    std::vector<double> vd(n);
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t u = splitmix64(seed);
        vi[i] = (int32_t)(u % 1000);
        vu[i] = u;
        vf[i] = (float)((double)(u % 1000000) / 1e3 - 20.0);
        vd[i] = (double)((int64_t)(u % 2000001) - 1000000) * 1e-3;
        cnt[i] = (int32_t)((u >> 40) % 4 + 1);
    }
    std::vector<float> vnan = vf;
    for (int g = 1; g <= 8; g++) vnan[(uint64_t)g * n / 8 - 1] = NAN;     // the last packet of a range (8, 4 or 2 ranks) and of the wave
    vnan[n / 3] = NAN;
    vnan[n - 1] = 7.0f;

    uint8_t out[16];
    int32_t ok = 0, want_n = 0;

    sharded_filter<int32_t>(r, MRPDR_AVG, vi, cnt, out, &ok);
    { const int32_t w = host_avg(vi, cnt, &want_n); int32_t gn; memcpy(&gn, out + 4, 4);
      end_SubTest("tfilter_Avg '%d %d': one division over the whole wave", r.rank, ok == 1 && same_bits(out, w) && gn == want_n); }
    sharded_filter<int32_t>(r, MRPDR_SUM, vi, cnt, out, &ok);
    end_SubTest("tfilter_Sum '%d'", r.rank, ok == 1 && same_bits(out, host_sum(vi)));
    sharded_filter<uint64_t>(r, MRPDR_MAX, vu, cnt, out, &ok);
    end_SubTest("tfilter_Max '%uld'", r.rank, ok == 1 && same_bits(out, host_max(vu)));
    sharded_filter<float>(r, MRPDR_SUM, vf, cnt, out, &ok);
    end_SubTest("tfilter_Sum '%f': the chain of roundings relayed in rank order", r.rank, ok == 1 && same_bits(out, host_sum(vf)));
    sharded_filter<double>(r, MRPDR_AVG, vd, cnt, out, &ok);
    { const double w = host_avg(vd, cnt, &want_n); int32_t gn; memcpy(&gn, out + 8, 4);
      end_SubTest("tfilter_Avg '%lf %d'", r.rank, ok == 1 && same_bits(out, w) && gn == want_n); }
    sharded_filter<float>(r, MRPDR_MIN, vnan, cnt, out, &ok);
    end_SubTest("tfilter_Min '%f' with NaNs at range ends", r.rank, ok == 1 && same_bits(out, host_min(vnan)));
    sharded_filter<float>(r, MRPDR_MAX, vnan, cnt, out, &ok);
    end_SubTest("tfilter_Max '%f' with NaNs at range ends", r.rank, ok == 1 && same_bits(out, host_max(vnan)));
    { std::vector<int32_t> none(n, 0);
      sharded_filter<double>(r, MRPDR_AVG, vd, none, out, &ok);
      end_SubTest("tfilter_Avg without weight: no packet", r.rank, ok == 0); }

    mrpdr_ctx_destroy(r.ctx);
    HIP_CHECK(hipStreamDestroy(r.stream));
}

int main(int argc, char **argv)
{
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (argc > 1) ndev = atoi(argv[1]) < ndev ? atoi(argv[1]) : ndev;
    if (ndev < 1) { printf("no HIP device\nMRNTEST_FAILURE\n"); return 2; }
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) devs[i] = i;
    std::vector<ncclComm_t> comms(ndev);
    NCCL_CHECK(ncclCommInitAll(comms.data(), ndev, devs.data()));
    int version = 0;
    NCCL_CHECK(ncclGetVersion(&version));
    printf("RCCL %d, %d rank(s), one thread per device\n", version, ndev);
    const uint64_t n = 1u << 18;
    std::vector<std::thread> threads;
    for (int i = 0; i < ndev; i++) {
        Rank r = {};
        r.rank = i; r.world = ndev; r.device = devs[i]; r.comm = comms[i];
        threads.emplace_back(rank_main, r, n);
    }
    for (auto &t : threads) t.join();
    for (int i = 0; i < ndev; i++) ncclCommDestroy(comms[i]);
    printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
