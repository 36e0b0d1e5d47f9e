// mrpdr_filters.C -- MRNet filter plugin: the built-in transformation filters TFILTER_SUM /
// MIN / MAX / AVG, TFILTER_ARRAY_CONCAT and TFILTER_INT_EQ_CLASS computed on an MI355X through libmrpdr.
//
// Loaded like any user filter (src/Network.C:1887, src/Filter.C:139-207):
//
//     int id = net->load_FilterFunc("libmrpdr_filters.so", "mrpdr_Sum");
//     Stream* s = net->new_Stream(comm, id, SFILTER_WAITFORALL);
//
// and called by Filter::push_Packets (src/Filter.C:42-117) with one wave: the packets the
// synchronisation filter released, one per child.  The plugin ABI is src/Filter.h:64-68; the
// `<name>_format_string` symbols are required by the loader (src/Filter.C:175-183).
//
// Each function accepts what its built-in twin accepts (src/FilterDefinitions.C:90-645): packets of
// exactly one scalar token ("%d", "%lf", ...), for Avg a scalar followed by "%d"; all packets of a
// wave must share one format (the built-ins assert it, :104-105; here a wave that does not is
// dropped with a message).  The result is ONE packet with the stream id, tag and format of the
// first input, holding what one call of the built-in over the same wave computes, bit for bit
// (mrpdr_wave_filter with MRPDR_SHAPE_FLAT).  Packets from senders of the other byte order are
// handled per packet, as Packet::decode_pdr_data does (src/Packet.C:118).
//
// mrpdr_ArrayConcat is tfilter_ArrayConcat (src/FilterDefinitions.C:649-810): packets of exactly one
// array token ("%ad", "%Alf", "%as", ...) in, one packet with the arrays of all of them end to end
// out.  The wave is decoded on the device (length prefixes stripped, elements of foreign senders
// reversed); the decoded values column IS the concatenation, so the only other work is its way back.
// mrpdr_IntEqClass is tfilter_IntEqClass (:812-902) over "%aud %aud %aud" packets, through
// mrpdr_wave_int_eq_class.
//
// Build inside an MRNet tree:
//     g++ -shared -fPIC integration/mrpdr_filters.C -I<mrnet>/include -I<mrnet>/xplat/include
//         -I<mrnet>/external/boost/include -I<repo>/include -I/opt/rocm/include
//         -D__HIP_PLATFORM_AMD__ -L<repo>/mrnet_amd/lib -lmrpdr -L/opt/rocm/lib -lamdhip64 -lpthread
//         -o libmrpdr_filters.so
// `make plugin-check` compiles this file against integration/buildcheck/mrnet/MRNet.h, a
// stand-in that declares only the handful of MRNet names used below (it is a build check of this
// file, not MRNet).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "mrnet/MRNet.h"
#include "mrpdr.h"
#include <hip/hip_runtime_api.h>

using namespace MRN;

namespace {

// Per (stream, filter) state, kept by MRNet in `void** filter_state` between waves.
struct GpuFilterState {
    mrpdr_ctx  *ctx;
    mrpdr_fmt  *fmt;          // compiled format of the stream's packets
    char       *fmt_text;
    hipStream_t stream;
    uint8_t    *h_rows;       // pinned: the data buffers of the wave, one row per packet
    uint8_t    *h_orders;     // pinned: sender byte order of every packet
    uint8_t    *d_rows, *d_orders;
    uint8_t    *h_result;     // pinned, 16 result bytes + the int32 presence flag: the last kernel of a
                              // filter call writes them here itself (no copy back is enqueued)
    size_t      cap_packets;
};

const uint64_t MAX_WAVE = 1u << 22;

// ---- helpers for the gather of large waves ------------------------------------------------------------
// What a filter does on the host is a walk over the wave's Packet objects, a few dependent cache lines
// each; from some ten thousand packets on that walk is most of the call (measured: 306 of 316 us at
// 65,536 packets).  A few threads, kept between waves, take a part of the wave each.  One pool for the
// whole plugin; a wave that finds it taken (another stream's filter is gathering) walks alone.
// MRPDR_FILTER_THREADS = 0 .. 16 sets the number of helpers (default 4, never more than the host's
// other hardware threads).
class GatherPool {
public:
    typedef void (*Part)(void *arg, unsigned part, unsigned parts);
    static GatherPool &get() { static GatherPool pool; return pool; }
    unsigned helpers() const { return (unsigned)workers_.size(); }
    // fn(arg, p, parts) for p = 0 .. parts - 1, part 0 on the calling thread; false: not run at all
    bool run(Part fn, void *arg)
    {
        if (workers_.empty() || !busy_.try_lock()) return false;
        {
            std::lock_guard<std::mutex> hold(m_);
            fn_ = fn; arg_ = arg; pending_ = (int)workers_.size(); gen_++;
        }
        wake_.notify_all();
        fn(arg, 0, (unsigned)workers_.size() + 1);
        {
            std::unique_lock<std::mutex> hold(m_);
            done_.wait(hold, [this] { return pending_ == 0; });
        }
        busy_.unlock();
        return true;
    }
private:
    GatherPool() : fn_(NULL), arg_(NULL), gen_(0), pending_(0), quit_(false)
    {
        unsigned want = 4;
        if (const char *e = getenv("MRPDR_FILTER_THREADS")) want = (unsigned)atoi(e);
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && want > hw - 1) want = hw - 1;
        if (want > 16) want = 16;
        for (unsigned k = 0; k < want; k++) workers_.push_back(std::thread(&GatherPool::loop, this, k + 1));
    }
    ~GatherPool()
    {
        { std::lock_guard<std::mutex> hold(m_); quit_ = true; }
        wake_.notify_all();
        for (size_t k = 0; k < workers_.size(); k++) workers_[k].join();
    }
    void loop(unsigned part)
    {
        unsigned seen = 0;
        std::unique_lock<std::mutex> hold(m_);
        for (;;) {
            wake_.wait(hold, [&] { return quit_ || gen_ != seen; });
            if (quit_) return;
            seen = gen_;
            const Part fn = fn_;
            void *arg = arg_;
            const unsigned parts = (unsigned)workers_.size() + 1;
            hold.unlock();
            fn(arg, part, parts);
            hold.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::mutex m_, busy_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    Part fn_;
    void *arg_;
    unsigned gen_;
    int pending_;
    bool quit_;
};
const size_t GATHER_THREADS_MIN = 8192;          // packets from which the helpers are woken

// a large block moved by the helpers as well (the destination is fresh from malloc: its pages are
// touched for the first time here, which is most of what such a copy costs)
struct CopyJob { uint8_t *dst; const uint8_t *src; size_t bytes; };
void copy_part(void *arg, unsigned part, unsigned parts)
{
    const CopyJob &c = *static_cast<const CopyJob *>(arg);
    const size_t lo = c.bytes / parts * part, hi = part + 1 == parts ? c.bytes : c.bytes / parts * (part + 1);
    memcpy(c.dst + lo, c.src + lo, hi - lo);
}
void copy_in_parts(void *dst, const void *src, size_t bytes)
{
    CopyJob job = { static_cast<uint8_t *>(dst), static_cast<const uint8_t *>(src), bytes };
    if (bytes < ((size_t)1 << 19) || !GatherPool::get().run(copy_part, &job)) memcpy(dst, src, bytes);
}

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    fprintf(stderr, "mrpdr filter: %s: %s\n", what, hipGetErrorString(e));
    return false;
}

void release(GpuFilterState *st)
{
    if (!st) return;
    if (st->h_rows) (void)hipHostFree(st->h_rows);
    if (st->h_orders) (void)hipHostFree(st->h_orders);
    if (st->d_rows) (void)hipFree(st->d_rows);
    if (st->d_orders) (void)hipFree(st->d_orders);
    if (st->h_result) (void)hipHostFree(st->h_result);
    if (st->stream) (void)hipStreamDestroy(st->stream);
    if (st->fmt) mrpdr_fmt_free(st->fmt);
    if (st->ctx) mrpdr_ctx_destroy(st->ctx);
    free(st->fmt_text);
    delete st;
}

// The state of this stream, created on the first wave, re-made when the stream's format changes.
GpuFilterState *state_for(void **slot, const char *fmt_text)
{
    GpuFilterState *st = static_cast<GpuFilterState *>(*slot);
    if (st && strcmp(st->fmt_text, fmt_text) == 0) return st;
    release(st);
    *slot = NULL;
    st = new GpuFilterState();
    memset(st, 0, sizeof *st);
    int device = 0;
    (void)hipGetDevice(&device);
    st->ctx = mrpdr_ctx_create(device, MAX_WAVE);
    st->fmt = mrpdr_fmt_compile(fmt_text);
    st->fmt_text = strdup(fmt_text);
    if (!st->ctx || !st->fmt || !st->fmt_text ||
        !hip_ok(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking), "hipStreamCreate") ||
        !hip_ok(hipHostMalloc((void **)&st->h_result, 32, hipHostMallocDefault), "hipHostMalloc")) {
        fprintf(stderr, "mrpdr filter: %s\n", mrpdr_last_error());
        release(st);
        return NULL;
    }
    *slot = st;
    return st;
}

bool grow(GpuFilterState *st, size_t n, size_t row)
{
    if (n <= st->cap_packets) return true;
    size_t cap = st->cap_packets ? st->cap_packets : 1024;
    while (cap < n) cap *= 2;
    if (st->h_rows) (void)hipHostFree(st->h_rows);
    if (st->h_orders) (void)hipHostFree(st->h_orders);
    if (st->d_rows) (void)hipFree(st->d_rows);
    if (st->d_orders) (void)hipFree(st->d_orders);
    st->h_rows = st->h_orders = st->d_rows = st->d_orders = NULL;
    st->cap_packets = 0;
    if (!hip_ok(hipHostMalloc((void **)&st->h_rows, cap * row, hipHostMallocDefault), "hipHostMalloc") ||
        !hip_ok(hipHostMalloc((void **)&st->h_orders, cap, hipHostMallocDefault), "hipHostMalloc") ||
        !hip_ok(hipMalloc((void **)&st->d_rows, cap * row), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&st->d_orders, cap), "hipMalloc"))
        return false;
    st->cap_packets = cap;
    return true;
}

// The gather of a scalar wave (or of packets [n * part / parts, n * (part + 1) / parts) of it):
// get_Buffer() is the encoded data of a packet (value[, count], no padding) and becomes a row.  A
// received packet carries its sender's byte order as the last byte of its header (src/Packet.C:96); a
// packet made on this node has native order and possibly no header yet.
struct ScalarGather {
    const std::vector<PacketPtr> *in;
    uint8_t *rows, *orders;
    const char *fmt_text;
    size_t row;
    uint8_t native;
    std::atomic<bool> foreign;          // a packet of the other byte order was seen
    std::atomic<size_t> bad;            // the first packet that is not of the wave's format (n: none)
};

void gather_scalar_part(void *arg, unsigned part, unsigned parts)
{
    ScalarGather &g = *static_cast<ScalarGather *>(arg);
    const std::vector<PacketPtr> &in = *g.in;
    const size_t n = in.size(), lo = n * part / parts, hi = n * (part + 1) / parts, row = g.row;
    bool foreign = false;
    const char *same_fmt = g.fmt_text;    // the last format pointer found equal: packets that share it are not compared again
    for (size_t i = lo; i < hi; i++) {
        // every packet is an object of its own somewhere on the heap: the walk over them is a chain of
        // cache misses, so the packet objects a few places ahead are asked for now (the pointer alone:
        // no accessor is called on them, accessors take the packet's lock)
        if (i + 12 < hi) __builtin_prefetch(in[i + 12].get());
        const Packet &p = *in[i];
        const char *pf = p.get_FormatString();
        const char *buf = p.get_Buffer();
        if ((pf != same_fmt && strcmp(pf, g.fmt_text) != 0) || p.get_BufferLen() != row || !buf) {
            size_t seen = g.bad.load();
            while (i < seen && !g.bad.compare_exchange_weak(seen, i)) { }
            return;
        }
        same_fmt = pf;
        // rows are 1 to 12 bytes: fixed-size copies (a store or two) instead of a call with a run-time length
        uint8_t *dst = g.rows + i * row;
        switch (row) {
        case 4:  memcpy(dst, buf, 4); break;
        case 8:  memcpy(dst, buf, 8); break;
        case 12: memcpy(dst, buf, 12); break;
        case 2:  memcpy(dst, buf, 2); break;
        case 1:  *dst = (uint8_t)*buf; break;
        default: memcpy(dst, buf, row); break;
        }
        const char *hdr = p.get_Header();
        const unsigned hdr_len = p.get_HeaderLen();
        g.orders[i] = (hdr && hdr_len) ? (uint8_t)hdr[hdr_len - 1] : g.native;
        foreign = foreign || g.orders[i] != g.native;
    }
    if (foreign) g.foreign.store(true);
}

// One wave through TFILTER_<op>.  Nothing is pushed to `out` when the filter has nothing to emit
// (a format it does not accept, a failure), as the built-ins do after logging.
void run(int op, const char *name, const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, void **state)
{
    const size_t n = in.size();
    if (n == 0) return;
    if (n > MAX_WAVE) { fprintf(stderr, "%s: a wave of %zu packets is beyond this plugin's limit\n", name, n); return; }
    const char *fmt_text = in[0]->get_FormatString();
    GpuFilterState *st = state_for(state, fmt_text);
    if (!st) return;
    // "%T" for Sum / Min / Max, "%T %d" for Avg (src/FilterDefinitions.C:108-146, 525-544)
    const int want = op == MRPDR_AVG ? 2 : 1;
    if (mrpdr_fmt_nfields(st->fmt) != want || mrpdr_fmt_field_kind(st->fmt, 0) != MRPDR_SCALAR ||
        (op == MRPDR_AVG && mrpdr_fmt_field_type(st->fmt, 1) != MRPDR_INT32_T)) {
        fprintf(stderr, "ERROR: %s() - invalid packet type: (%s)\n", name, fmt_text);
        return;
    }
    const int type = mrpdr_fmt_field_type(st->fmt, 0);
    const size_t width = (size_t)mrpdr_fmt_field_width(st->fmt, 0);
    const size_t row = (size_t)mrpdr_fmt_fixed_bytes(st->fmt);
    if (!grow(st, n, row)) return;
    ScalarGather job = { &in, st->h_rows, st->h_orders, fmt_text, row, (uint8_t)mrpdr_local_byteorder(), { false }, { n } };
    if (n < GATHER_THREADS_MIN || !GatherPool::get().run(gather_scalar_part, &job)) gather_scalar_part(&job, 0, 1);
    if (job.bad.load() < n) {
        fprintf(stderr, "%s: packet %zu of the wave is not a '%s' packet\n", name, job.bad.load(), fmt_text);
        return;
    }
    const bool foreign = job.foreign.load();
    if (!hip_ok(hipMemcpyAsync(st->d_rows, st->h_rows, n * row, hipMemcpyHostToDevice, st->stream), "H2D")) return;
    // The result comes back by itself: the filter's last kernel writes the 16 bytes and then, behind a
    // system-scope fence, the presence flag into pinned memory; this thread presets the flag to
    // "pending" and polls it (a bounded spin, then the stream is waited for the ordinary way).
    uint8_t *result = st->h_result;
    volatile int32_t *flag = (volatile int32_t *)(st->h_result + 16);
    *flag = MRPDR_RESULT_PENDING;
    const void *counts = op == MRPDR_AVG ? st->d_rows + width : NULL;
    int32_t called;
    if (foreign) {
        if (!hip_ok(hipMemcpyAsync(st->d_orders, st->h_orders, n, hipMemcpyHostToDevice, st->stream), "H2D")) return;
        called = mrpdr_wave_filter_mixed(st->ctx, op, type, MRPDR_SHAPE_FLAT, st->d_rows, row, counts, row, n,
                                         st->d_orders, result, (int32_t *)(st->h_result + 16), st->stream);
    } else {
        called = mrpdr_wave_filter(st->ctx, op, type, MRPDR_SHAPE_FLAT, st->d_rows, row, counts, row, n, 0,
                                   result, (int32_t *)(st->h_result + 16), st->stream);
    }
    if (!called) { fprintf(stderr, "%s: %s\n", name, mrpdr_last_error()); return; }
    for (unsigned spins = 0; *flag == MRPDR_RESULT_PENDING && spins < (1u << 22); spins++)
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (*flag == MRPDR_RESULT_PENDING && !hip_ok(hipStreamSynchronize(st->stream), "hipStreamSynchronize")) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const int32_t ok = *flag;
    if (ok != 1) return;        // 0: e.g. an Avg whose counts sum to zero (the reference divides by zero there)
    // the 16 result bytes are the output packet's fields in order: value, then for Avg the int count
    const void *fields[2] = { result, result + width };
    out.push_back(PacketPtr(new Packet(in[0]->get_StreamId(), in[0]->get_Tag(), fields, fmt_text)));
}

// ---- filters over packets of arrays: TFILTER_ARRAY_CONCAT, TFILTER_INT_EQ_CLASS ------------------
const int MAX_ARRAY_FIELDS = 3;

struct VarState {
    mrpdr_ctx  *ctx;
    mrpdr_fmt  *fmt;
    char       *fmt_text;
    hipStream_t stream;
    // the wave as received: data buffers end to end in a pinned arena, offsets / lengths / byte orders
    uint8_t    *h_arena, *d_arena, *h_orders, *d_orders;
    uint64_t   *h_meta, *d_meta;       // offsets[n+1] then lens[n]
    size_t      cap_bytes, cap_packets;
    // the decoded columns, one per field
    uint8_t    *d_values[MAX_ARRAY_FIELDS];
    uint64_t   *d_counts[MAX_ARRAY_FIELDS], *d_offs[MAX_ARRAY_FIELDS], *d_sub_counts, *d_sub_offs;
    int32_t    *d_ok;
    size_t      cap_values[MAX_ARRAY_FIELDS], cap_sub;
    // TFILTER_INT_EQ_CLASS: the columns of the output packet and the call's scratch
    uint32_t   *d_out[3];
    uint64_t   *d_out_meta;            // per output column: counts[1], offs[2]
    uint8_t    *d_scratch;
    size_t      cap_out[3], cap_scratch;
    // TFILTER_ARRAY_CONCAT: where the concatenation and its element count come back to (pinned)
    uint8_t    *h_out;
    uint64_t   *h_total;
    size_t      cap_h_out, wave_bytes;  // wave_bytes: data bytes of the wave gather_and_decode took last
};

void release_var(VarState *st)
{
    if (!st) return;
    if (st->h_arena) (void)hipHostFree(st->h_arena);
    if (st->h_orders) (void)hipHostFree(st->h_orders);
    if (st->h_meta) (void)hipHostFree(st->h_meta);
    if (st->h_out) (void)hipHostFree(st->h_out);
    if (st->h_total) (void)hipHostFree(st->h_total);
    void *dev[] = { st->d_arena, st->d_orders, st->d_meta, st->d_values[0], st->d_values[1], st->d_values[2],
                    st->d_counts[0], st->d_counts[1], st->d_counts[2], st->d_offs[0], st->d_offs[1], st->d_offs[2],
                    st->d_sub_counts, st->d_sub_offs, st->d_ok, st->d_out[0], st->d_out[1], st->d_out[2],
                    st->d_out_meta, st->d_scratch };
    for (size_t k = 0; k < sizeof dev / sizeof dev[0]; k++)
        if (dev[k]) (void)hipFree(dev[k]);
    if (st->stream) (void)hipStreamDestroy(st->stream);
    if (st->fmt) mrpdr_fmt_free(st->fmt);
    if (st->ctx) mrpdr_ctx_destroy(st->ctx);
    free(st->fmt_text);
    delete st;
}

VarState *var_state_for(void **slot, const char *fmt_text)
{
    VarState *st = static_cast<VarState *>(*slot);
    if (st && strcmp(st->fmt_text, fmt_text) == 0) return st;
    release_var(st);
    *slot = NULL;
    st = new VarState();
    memset(st, 0, sizeof *st);
    int device = 0;
    (void)hipGetDevice(&device);
    st->ctx = mrpdr_ctx_create(device, MAX_WAVE);
    st->fmt = mrpdr_fmt_compile(fmt_text);
    st->fmt_text = strdup(fmt_text);
    if (!st->ctx || !st->fmt || !st->fmt_text ||
        !hip_ok(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking), "hipStreamCreate")) {
        fprintf(stderr, "mrpdr filter: %s\n", mrpdr_last_error());
        release_var(st);
        return NULL;
    }
    *slot = st;
    return st;
}

template <typename T> bool regrow(T **p, size_t bytes, bool pinned)
{
    if (*p) { if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); }
    *p = NULL;
    return pinned ? hip_ok(hipHostMalloc((void **)p, bytes, hipHostMallocDefault), "hipHostMalloc")
                  : hip_ok(hipMalloc((void **)p, bytes), "hipMalloc");
}

size_t grown(size_t cap, size_t need, size_t first)
{
    if (cap == 0) cap = first;
    while (cap < need) cap *= 2;
    return cap;
}

// Room for a wave of n packets and `bytes` data bytes whose format has nf array fields.
bool var_grow(VarState *st, size_t n, size_t bytes, int nf, bool strings)
{
    if (n > st->cap_packets) {
        const size_t cap = grown(st->cap_packets, n, 1024);
        st->cap_packets = 0;
        if (!regrow(&st->h_orders, cap, true) || !regrow(&st->d_orders, cap, false) ||
            !regrow(&st->h_meta, 8 * (2 * cap + 1), true) || !regrow(&st->d_meta, 8 * (2 * cap + 1), false) ||
            !regrow(&st->d_ok, 4 * cap, false))
            return false;
        for (int f = 0; f < MAX_ARRAY_FIELDS; f++)
            if (!regrow(&st->d_counts[f], 8 * cap, false) || !regrow(&st->d_offs[f], 8 * (cap + 1), false)) return false;
        st->cap_packets = cap;
    }
    if (bytes > st->cap_bytes) {
        const size_t cap = grown(st->cap_bytes, bytes, 1 << 16);
        st->cap_bytes = 0;
        if (!regrow(&st->h_arena, cap, true) || !regrow(&st->d_arena, cap, false)) return false;
        st->cap_bytes = cap;
    }
    // every element takes at least its width on the wire, every string at least five bytes
    for (int f = 0; f < nf; f++) {
        const size_t width = (size_t)mrpdr_fmt_field_width(st->fmt, f), values = bytes / width + 1;
        if (values <= st->cap_values[f]) continue;
        const size_t cap = grown(st->cap_values[f], values, 1 << 14);
        st->cap_values[f] = 0;
        if (!regrow(&st->d_values[f], cap * width, false)) return false;
        st->cap_values[f] = cap;
    }
    const size_t subs = strings ? bytes / 5 + 1 : 0;
    if (subs > st->cap_sub) {
        const size_t cap = grown(st->cap_sub, subs, 1 << 12);
        st->cap_sub = 0;
        if (!regrow(&st->d_sub_counts, 8 * cap, false) || !regrow(&st->d_sub_offs, 8 * (cap + 1), false)) return false;
        st->cap_sub = cap;
    }
    return true;
}

// The gather of a wave of array packets, in two walks (pass 0: format, lengths, byte totals per part;
// pass 1: the copies into the arena, byte orders, offsets), each over packets
// [n * part / parts, n * (part + 1) / parts).
struct VarGather {
    enum { MAX_PARTS = 17 };
    const std::vector<PacketPtr> *in;
    const char *fmt_text;
    uint8_t native;
    int pass;
    unsigned parts;                     // of pass 0 (pass 1 must use the same)
    std::vector<uint64_t> lens;
    size_t part_bytes[MAX_PARTS];       // pass 0: bytes of each part; then: where each part starts in the arena
    uint8_t *arena, *orders;
    uint64_t *offsets, *out_lens;
    std::atomic<size_t> bad;
    std::atomic<bool> mixed;            // not every packet has the byte order of packet 0
};

void gather_var_part(void *arg, unsigned part, unsigned parts)
{
    VarGather &g = *static_cast<VarGather *>(arg);
    const std::vector<PacketPtr> &in = *g.in;
    const size_t n = in.size(), lo = n * part / parts, hi = n * (part + 1) / parts;
    if (g.pass == 0) {
        if (part == 0) g.parts = parts;
        size_t bytes = 0;
        const char *same_fmt = g.fmt_text;
        for (size_t i = lo; i < hi; i++) {
            if (i + 12 < hi) __builtin_prefetch(in[i + 12].get());
            const Packet &p = *in[i];
            const char *pf = p.get_FormatString();
            if ((pf != same_fmt && strcmp(pf, g.fmt_text) != 0) || !p.get_Buffer()) {
                size_t seen = g.bad.load();
                while (i < seen && !g.bad.compare_exchange_weak(seen, i)) { }
                return;
            }
            same_fmt = pf;
            g.lens[i] = (uint64_t)p.get_BufferLen();
            bytes += (size_t)g.lens[i];
        }
        g.part_bytes[part] = bytes;
        return;
    }
    size_t at = g.part_bytes[part];
    // the byte order every packet is compared with: packet 0's (a received packet carries its sender's
    // byte order as the last byte of its header, src/Packet.C:96; a packet made on this node has native
    // order and possibly no header yet)
    const char *hdr0 = in[0]->get_Header();
    const unsigned hdr0_len = in[0]->get_HeaderLen();
    const uint8_t order0 = (hdr0 && hdr0_len) ? (uint8_t)hdr0[hdr0_len - 1] : g.native;
    bool mixed = false;
    for (size_t i = lo; i < hi; i++) {
        if (i + 12 < hi) __builtin_prefetch(in[i + 12].get());
        const Packet &p = *in[i];
        const size_t len = (size_t)g.lens[i];
        memcpy(g.arena + at, p.get_Buffer(), len);
        g.offsets[i] = at; g.out_lens[i] = len; at += len;
        const char *hdr = p.get_Header();
        const unsigned hdr_len = p.get_HeaderLen();
        g.orders[i] = (hdr && hdr_len) ? (uint8_t)hdr[hdr_len - 1] : g.native;
        mixed = mixed || g.orders[i] != order0;
    }
    if (mixed) g.mixed.store(true);
}

// The wave into the arena and onto the device, decoded into cols[nf] (every packet with its own
// sender's byte order).  False: nothing to emit (a message has been printed).
bool gather_and_decode(const char *name, VarState *st, const std::vector<PacketPtr> &in, int nf, bool strings,
                       mrpdr_column *cols)
{
    const size_t n = in.size();
    const char *fmt_text = st->fmt_text;
    // two walks over the packets, each in parts (helper threads for large waves): lengths and format
    // first -- the arena is sized by their sum, and a part starts where the parts in front end -- then
    // the copies
    VarGather job;
    job.in = &in; job.fmt_text = fmt_text; job.native = (uint8_t)mrpdr_local_byteorder();
    job.bad.store(n); job.mixed.store(false);
    job.lens.resize(n);
    memset(job.part_bytes, 0, sizeof job.part_bytes);
    const bool threads = n >= GATHER_THREADS_MIN;
    job.pass = 0;
    if (!threads || !GatherPool::get().run(gather_var_part, &job)) gather_var_part(&job, 0, 1);
    if (job.bad.load() < n) {
        fprintf(stderr, "%s: packet %zu of the wave is not a '%s' packet\n", name, job.bad.load(), fmt_text);
        return false;
    }
    size_t bytes = 0;
    for (unsigned k = 0; k < VarGather::MAX_PARTS; k++) { const size_t b = job.part_bytes[k]; job.part_bytes[k] = bytes; bytes += b; }
    if (!var_grow(st, n, bytes, nf, strings)) return false;
    job.arena = st->h_arena; job.orders = st->h_orders; job.offsets = st->h_meta; job.out_lens = st->h_meta + n + 1;
    job.pass = 1;
    // (the same number of parts as in the first walk: the parts' starts were added up for it)
    if (job.parts == 1 || !GatherPool::get().run(gather_var_part, &job)) {
        if (job.parts != 1) {           // the helpers were there for the first walk and are not now: one part over all
            memset(job.part_bytes, 0, sizeof job.part_bytes);
            job.parts = 1;
        }
        gather_var_part(&job, 0, 1);
    }
    const bool same = !job.mixed.load();
    st->h_meta[n] = bytes;
    st->wave_bytes = bytes;
    uint64_t *d_offsets = st->d_meta, *d_lens = st->d_meta + n + 1;
    if (!hip_ok(hipMemcpyAsync(st->d_arena, st->h_arena, bytes ? bytes : 1, hipMemcpyHostToDevice, st->stream), "H2D") ||
        !hip_ok(hipMemcpyAsync(st->d_meta, st->h_meta, 8 * (2 * n + 1), hipMemcpyHostToDevice, st->stream), "H2D"))
        return false;
    mrpdr_wire wire = { st->d_arena, d_offsets, d_lens, 0, bytes };
    memset(cols, 0, nf * sizeof *cols);
    for (int f = 0; f < nf; f++) {
        const size_t width = (size_t)mrpdr_fmt_field_width(st->fmt, f);
        cols[f].values = st->d_values[f]; cols[f].counts = st->d_counts[f]; cols[f].offs = st->d_offs[f];
        cols[f].capacity = strings ? st->cap_values[f] * width : st->cap_values[f];
    }
    if (strings) { cols[0].sub_counts = st->d_sub_counts; cols[0].sub_offs = st->d_sub_offs; cols[0].sub_capacity = st->cap_sub; }
    int32_t called;
    if (!same) {
        if (!hip_ok(hipMemcpyAsync(st->d_orders, st->h_orders, n, hipMemcpyHostToDevice, st->stream), "H2D")) return false;
        called = mrpdr_wave_decode_mixed(st->ctx, st->fmt, &wire, n, st->d_orders, cols, st->d_ok, st->stream);
    } else {
        called = mrpdr_wave_decode(st->ctx, st->fmt, &wire, n, st->h_orders[0], cols, st->d_ok, st->stream);
    }
    if (!called) { fprintf(stderr, "%s: %s\n", name, mrpdr_last_error()); return false; }
    return true;
}

// Reads and clears the ctx's status word (this also waits for the stream); false when it is set.
bool device_fine(const char *name, VarState *st)
{
    uint32_t status = 0;
    if (!mrpdr_ctx_status(st->ctx, st->stream, &status)) { fprintf(stderr, "%s: %s\n", name, mrpdr_last_error()); return false; }
    if (status) { fprintf(stderr, "%s: device status %u\n", name, status); return false; }
    return true;
}

void run_concat(const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, void **state)
{
    const char *name = "mrpdr_ArrayConcat";
    const size_t n = in.size();
    if (n == 0) return;
    if (n > MAX_WAVE) { fprintf(stderr, "%s: a wave of %zu packets is beyond this plugin's limit\n", name, n); return; }
    const char *fmt_text = in[0]->get_FormatString();
    VarState *st = var_state_for(state, fmt_text);
    if (!st) return;
    const int kind = mrpdr_fmt_nfields(st->fmt) == 1 ? mrpdr_fmt_field_kind(st->fmt, 0) : -1;
    if (kind != MRPDR_ARRAY && kind != MRPDR_STRING_ARRAY) {        // src/FilterDefinitions.C:672-769
        fprintf(stderr, "ERROR: tfilter_ArrayConcat() - invalid packet type: (%s)\n", fmt_text);
        return;
    }
    const bool strings = kind == MRPDR_STRING_ARRAY;
    const size_t width = (size_t)mrpdr_fmt_field_width(st->fmt, 0);
    mrpdr_column col;
    if (!gather_and_decode(name, st, in, 1, strings, &col)) return;
    // a packet that does not decode adds nothing, as one whose unpack fails (:771-782): its count is 0
    uint64_t total = 0;
    const bool large = fmt_text[1] == 'A';
    void *result = NULL;
    if (!strings) {
        // One wait for everything: every element takes `width` bytes of the wave at least, so the
        // concatenation is copied back in that length together with its element count (both into pinned
        // memory), and what the count says is then moved into the array the output packet will own.
        const size_t bound = st->wave_bytes / width;
        if (!st->h_total && !regrow(&st->h_total, 8, true)) return;
        if (bound * width > st->cap_h_out) {
            const size_t cap = grown(st->cap_h_out, bound * width, 1 << 16);
            st->cap_h_out = 0;
            if (!regrow(&st->h_out, cap, true)) return;
            st->cap_h_out = cap;
        }
        if (!hip_ok(hipMemcpyAsync(st->h_total, st->d_offs[0] + n, 8, hipMemcpyDeviceToHost, st->stream), "D2H")) return;
        if (bound && !hip_ok(hipMemcpyAsync(st->h_out, st->d_values[0], bound * width, hipMemcpyDeviceToHost, st->stream), "D2H")) return;
        if (!device_fine(name, st)) return;
        total = *st->h_total;
        if (total > 0xFFFFFFFFull) { fprintf(stderr, "%s: %llu elements do not fit the filter's 32-bit length\n", name, (unsigned long long)total); return; }
        if (total > bound) { fprintf(stderr, "%s: %llu elements from %zu bytes\n", name, (unsigned long long)total, st->wave_bytes); return; }
        result = malloc(total ? total * width : 1);
        if (!result) return;
        copy_in_parts(result, st->h_out, total * width);
    } else {
        if (!hip_ok(hipMemcpyAsync(&total, st->d_offs[0] + n, 8, hipMemcpyDeviceToHost, st->stream), "D2H")) return;
        if (!device_fine(name, st)) return;
        if (total > 0xFFFFFFFFull) { fprintf(stderr, "%s: %llu elements do not fit the filter's 32-bit length\n", name, (unsigned long long)total); return; }
        // the strings come back as one block and the places of each; the packet wants one malloc per string
        std::vector<uint64_t> sub_offs(total + 1);
        if (!hip_ok(hipMemcpyAsync(sub_offs.data(), st->d_sub_offs, 8 * (total + 1), hipMemcpyDeviceToHost, st->stream), "D2H") ||
            !hip_ok(hipStreamSynchronize(st->stream), "hipStreamSynchronize")) return;
        std::vector<char> text(sub_offs[total] ? sub_offs[total] : 1);
        if (sub_offs[total] && (!hip_ok(hipMemcpyAsync(text.data(), st->d_values[0], sub_offs[total], hipMemcpyDeviceToHost, st->stream), "D2H") ||
                                !hip_ok(hipStreamSynchronize(st->stream), "hipStreamSynchronize"))) return;
        char **strs = (char **)malloc((total ? total : 1) * sizeof(char *));
        if (!strs) return;
        for (uint64_t j = 0; j < total; j++) strs[j] = strdup(text.data() + sub_offs[j]);
        result = strs;
    }
    // Packet(stream, tag, fmt, array, length): an unsigned length for '%a', a uint64_t one for '%A'
    // (src/Packet.C:1478-1793); the packet frees the array (and its strings) with it
    PacketPtr p(large ? new Packet(in[0]->get_StreamId(), in[0]->get_Tag(), fmt_text, result, (uint64_t)total)
                      : new Packet(in[0]->get_StreamId(), in[0]->get_Tag(), fmt_text, result, (unsigned)total));
    p->set_DestroyData(true);
    out.push_back(p);
}

// tfilter_IntEqClass (src/FilterDefinitions.C:812-902): "%aud %aud %aud" packets of (class values,
// members per class, members) in, the classes of all of them merged by value out.
void run_eq_class(const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, void **state)
{
    const char *name = "mrpdr_IntEqClass";
    const size_t n = in.size();
    if (n == 0) return;
    if (n > MAX_WAVE) { fprintf(stderr, "%s: a wave of %zu packets is beyond this plugin's limit\n", name, n); return; }
    const char *fmt_text = in[0]->get_FormatString();
    VarState *st = var_state_for(state, fmt_text);
    if (!st) return;
    bool fits = mrpdr_fmt_nfields(st->fmt) == 3;
    for (int f = 0; fits && f < 3; f++)
        fits = mrpdr_fmt_field_kind(st->fmt, f) == MRPDR_ARRAY && mrpdr_fmt_field_width(st->fmt, f) == 4;
    if (!fits) { fprintf(stderr, "ERROR: %s() - invalid packet type: (%s)\n", name, fmt_text); return; }
    mrpdr_column cols[3];
    if (!gather_and_decode(name, st, in, 3, false, cols)) return;
    // outputs are bounded by the inputs: no more classes than class entries, no more members than came in
    const size_t want[3] = { st->cap_values[0], st->cap_values[0], st->cap_values[2] };
    for (int k = 0; k < 3; k++) {
        if (want[k] <= st->cap_out[k]) continue;
        st->cap_out[k] = 0;
        if (!regrow(&st->d_out[k], 4 * want[k], false)) return;
        st->cap_out[k] = want[k];
    }
    if (!st->d_out_meta && !regrow(&st->d_out_meta, 8 * 9, false)) return;
    const size_t scratch = (size_t)mrpdr_eqclass_scratch_bytes(cols[0].capacity);
    if (scratch > st->cap_scratch) {
        st->cap_scratch = 0;
        if (!regrow(&st->d_scratch, scratch, false)) return;
        st->cap_scratch = scratch;
    }
    mrpdr_column oc[3];
    memset(oc, 0, sizeof oc);
    for (int k = 0; k < 3; k++) {
        oc[k].values = st->d_out[k]; oc[k].counts = st->d_out_meta + 3 * k; oc[k].offs = st->d_out_meta + 3 * k + 1;
        oc[k].capacity = st->cap_out[k];
    }
    if (!mrpdr_wave_int_eq_class(st->ctx, &cols[0], &cols[1], &cols[2], n, &oc[0], &oc[1], &oc[2], st->d_scratch, scratch, st->stream)) {
        fprintf(stderr, "%s: %s\n", name, mrpdr_last_error());
        return;
    }
    uint64_t meta[9];
    if (!hip_ok(hipMemcpyAsync(meta, st->d_out_meta, sizeof meta, hipMemcpyDeviceToHost, st->stream), "D2H")) return;
    if (!device_fine(name, st)) return;
    const uint64_t nclasses = meta[0], nmems = meta[6];
    std::vector<uint32_t> vals(nclasses ? nclasses : 1), memcnts(nclasses ? nclasses : 1), mems(nmems ? nmems : 1);
    if ((nclasses && (!hip_ok(hipMemcpyAsync(vals.data(), st->d_out[0], 4 * nclasses, hipMemcpyDeviceToHost, st->stream), "D2H") ||
                      !hip_ok(hipMemcpyAsync(memcnts.data(), st->d_out[1], 4 * nclasses, hipMemcpyDeviceToHost, st->stream), "D2H"))) ||
        (nmems && !hip_ok(hipMemcpyAsync(mems.data(), st->d_out[2], 4 * nmems, hipMemcpyDeviceToHost, st->stream), "D2H")) ||
        !hip_ok(hipStreamSynchronize(st->stream), "hipStreamSynchronize"))
        return;
    // (the packet encodes its arguments when it is made, :877-882; the arrays need not outlive this call)
    out.push_back(PacketPtr(new Packet(in[0]->get_StreamId(), in[0]->get_Tag(), "%aud %aud %aud",
                                       vals.data(), (unsigned)nclasses, memcnts.data(), (unsigned)nclasses,
                                       mems.data(), (unsigned)nmems)));
}

} // namespace

extern "C" {

// The loader wants these symbols; MRNet stores the strings and never compares them with a packet
// (src/Filter.C:35): the functions check the format themselves.
const char *mrpdr_Sum_format_string = "";
const char *mrpdr_Min_format_string = "";
const char *mrpdr_Max_format_string = "";
const char *mrpdr_Avg_format_string = "";
const char *mrpdr_ArrayConcat_format_string = "";
const char *mrpdr_IntEqClass_format_string = "";

void mrpdr_Sum(const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, std::vector<PacketPtr> & /* out_reverse */,
               void **state, PacketPtr & /* params */, const TopologyLocalInfo &)
{
    run(MRPDR_SUM, "mrpdr_Sum", in, out, state);
}

void mrpdr_Min(const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, std::vector<PacketPtr> &, void **state,
               PacketPtr &, const TopologyLocalInfo &)
{
    run(MRPDR_MIN, "mrpdr_Min", in, out, state);
}

void mrpdr_Max(const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, std::vector<PacketPtr> &, void **state,
               PacketPtr &, const TopologyLocalInfo &)
{
    run(MRPDR_MAX, "mrpdr_Max", in, out, state);
}

void mrpdr_Avg(const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, std::vector<PacketPtr> &, void **state,
               PacketPtr &, const TopologyLocalInfo &)
{
    run(MRPDR_AVG, "mrpdr_Avg", in, out, state);
}

void mrpdr_ArrayConcat(const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, std::vector<PacketPtr> &, void **state,
                       PacketPtr &, const TopologyLocalInfo &)
{
    run_concat(in, out, state);
}

void mrpdr_IntEqClass(const std::vector<PacketPtr> &in, std::vector<PacketPtr> &out, std::vector<PacketPtr> &, void **state,
                      PacketPtr &, const TopologyLocalInfo &)
{
    run_eq_class(in, out, state);
}

} // extern "C"
