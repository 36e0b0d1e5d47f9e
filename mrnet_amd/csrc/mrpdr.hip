// C ABI of the wave codec (include/mrpdr.h): argument checks, format compiler,
// kernel launches.  gfx950 only.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <stddef.h>
#include <type_traits>
#include <mutex>

#include "common.h"
#include "scan.h"
#include "rows.h"
#include "copy.h"
#include "filter.h"
#include "flatscan.h"
#include "flatwide.h"
#include "header.h"
#include "message.h"
#include "strarr.h"
#include "tile.h"
#include "eqclass.h"
#include "fused.h"
#include "perfnodes.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int32_t fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return 0;
}

#define HIP_OK(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) return fail("%s: %s", #call, hipGetErrorString(e_));    \
    } while (0)

#define LAUNCHED(name)                                                                \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) return fail("launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *mrpdr_version(void) { return "mrpdr 0.1 (gfx950)"; }
extern "C" const char *mrpdr_last_error(void) { return g_err; }
extern "C" int32_t mrpdr_local_byteorder(void) { return MRPDR_LITTLEENDIAN; }

// ---------------------------------------------------------------------------
// format strings
// ---------------------------------------------------------------------------
struct mrpdr_fmt {
    FmtDev dev;
    char  *text;
    int    fixed;        // all scalar
    int    nvar;
    int    nstrarr;      // string-array tokens
    TileVars tv;         // the variable fields, for the tile kernels
};

// Fmt2Type, src/DataElement.C:205-297
static int token_type(const char *tok)
{
    static const struct { const char *tok; int type; } table[] = {
        {"c", MRPDR_CHAR_T}, {"uc", MRPDR_UCHAR_T}, {"hd", MRPDR_INT16_T}, {"uhd", MRPDR_UINT16_T},
        {"d", MRPDR_INT32_T}, {"ud", MRPDR_UINT32_T}, {"ld", MRPDR_INT64_T}, {"uld", MRPDR_UINT64_T},
        {"f", MRPDR_FLOAT_T}, {"lf", MRPDR_DOUBLE_T}, {"s", MRPDR_STRING_T},
        {"ac", MRPDR_CHAR_ARRAY_T}, {"auc", MRPDR_UCHAR_ARRAY_T}, {"ahd", MRPDR_INT16_ARRAY_T},
        {"auhd", MRPDR_UINT16_ARRAY_T}, {"ad", MRPDR_INT32_ARRAY_T}, {"aud", MRPDR_UINT32_ARRAY_T},
        {"ald", MRPDR_INT64_ARRAY_T}, {"auld", MRPDR_UINT64_ARRAY_T}, {"af", MRPDR_FLOAT_ARRAY_T},
        {"alf", MRPDR_DOUBLE_ARRAY_T}, {"as", MRPDR_STRING_ARRAY_T},
        {"Ac", MRPDR_CHAR_LRG_ARRAY_T}, {"Auc", MRPDR_UCHAR_LRG_ARRAY_T},
        {"Ahd", MRPDR_INT16_LRG_ARRAY_T}, {"Auhd", MRPDR_UINT16_LRG_ARRAY_T},
        {"Ad", MRPDR_INT32_LRG_ARRAY_T}, {"Aud", MRPDR_UINT32_LRG_ARRAY_T},
        {"Ald", MRPDR_INT64_LRG_ARRAY_T}, {"Auld", MRPDR_UINT64_LRG_ARRAY_T},
        {"Af", MRPDR_FLOAT_LRG_ARRAY_T}, {"Alf", MRPDR_DOUBLE_LRG_ARRAY_T},
        {"As", MRPDR_STRING_LRG_ARRAY_T},
    };
    for (size_t i = 0; i < sizeof table / sizeof table[0]; i++)
        if (strcmp(tok, table[i].tok) == 0) return table[i].type;
    return MRPDR_UNKNOWN_T;
}

static int type_kind(int t)
{
    switch (t) {
    case MRPDR_CHAR_T: case MRPDR_UCHAR_T: case MRPDR_INT16_T: case MRPDR_UINT16_T:
    case MRPDR_INT32_T: case MRPDR_UINT32_T: case MRPDR_INT64_T: case MRPDR_UINT64_T:
    case MRPDR_FLOAT_T: case MRPDR_DOUBLE_T: return MRPDR_SCALAR;
    case MRPDR_STRING_T: return MRPDR_STRING;
    case MRPDR_STRING_ARRAY_T: case MRPDR_STRING_LRG_ARRAY_T: return MRPDR_STRING_ARRAY;
    case MRPDR_UNKNOWN_T: return -1;
    default: return MRPDR_ARRAY;
    }
}

static int type_width(int t)
{
    switch (t) {
    case MRPDR_INT16_T: case MRPDR_UINT16_T: case MRPDR_INT16_ARRAY_T: case MRPDR_UINT16_ARRAY_T:
    case MRPDR_INT16_LRG_ARRAY_T: case MRPDR_UINT16_LRG_ARRAY_T: return 2;
    case MRPDR_INT32_T: case MRPDR_UINT32_T: case MRPDR_FLOAT_T: case MRPDR_INT32_ARRAY_T:
    case MRPDR_UINT32_ARRAY_T: case MRPDR_FLOAT_ARRAY_T: case MRPDR_INT32_LRG_ARRAY_T:
    case MRPDR_UINT32_LRG_ARRAY_T: case MRPDR_FLOAT_LRG_ARRAY_T: return 4;
    case MRPDR_INT64_T: case MRPDR_UINT64_T: case MRPDR_DOUBLE_T: case MRPDR_INT64_ARRAY_T:
    case MRPDR_UINT64_ARRAY_T: case MRPDR_DOUBLE_ARRAY_T: case MRPDR_INT64_LRG_ARRAY_T:
    case MRPDR_UINT64_LRG_ARRAY_T: case MRPDR_DOUBLE_LRG_ARRAY_T: return 8;
    case MRPDR_UNKNOWN_T: return 0;
    default: return 1;
    }
}

extern "C" mrpdr_fmt *mrpdr_fmt_compile(const char *fmt)
{
    if (!fmt) { fail("mrpdr_fmt_compile: NULL format"); return nullptr; }
    static const char delim[] = " \t\n%";          // src/Packet.C:611
    mrpdr_fmt *f = (mrpdr_fmt *)calloc(1, sizeof *f);
    if (!f) { fail("mrpdr_fmt_compile: out of memory"); return nullptr; }
    const char *p = fmt;
    uint32_t fixed = 0;
    for (;;) {
        p += strspn(p, delim);
        if (!*p) break;
        size_t len = strcspn(p, delim);
        char tok[16];
        if (len >= sizeof tok) { fail("mrpdr_fmt_compile: unknown token in '%s'", fmt); free(f); return nullptr; }
        memcpy(tok, p, len); tok[len] = 0;
        p += len;
        int t = token_type(tok);
        if (t == MRPDR_UNKNOWN_T) { fail("mrpdr_fmt_compile: unknown token '%s' in '%s'", tok, fmt); free(f); return nullptr; }
        if (f->dev.nf == MRPDR_MAXF) { fail("mrpdr_fmt_compile: more than %d tokens in '%s'", MRPDR_MAXF, fmt); free(f); return nullptr; }
        FieldDev &fd = f->dev.f[f->dev.nf++];
        fd.type = (uint8_t)t;
        fd.kind = (uint8_t)type_kind(t);
        fd.width = (uint8_t)type_width(t);
        fd.large = (uint8_t)(t >= MRPDR_CHAR_LRG_ARRAY_T);
        fd.fixed_before = fixed;
        if (fd.kind == MRPDR_SCALAR) fixed += fd.width;
        else {
            fixed += prefix_bytes(fd.kind);
            if (f->nvar < TILE_MAXV) { f->tv.field[f->nvar] = f->dev.nf - 1; f->tv.width[f->nvar] = fd.width; }
            f->nvar++;
            f->nstrarr += fd.kind == MRPDR_STRING_ARRAY;
        }
    }
    f->dev.fixed_bytes = fixed;
    f->tv.nv = f->nvar <= TILE_MAXV ? f->nvar : 0;
    f->fixed = f->nvar == 0;
    f->text = strdup(fmt);
    return f;
}

extern "C" void mrpdr_fmt_free(mrpdr_fmt *f)
{
    if (f) { free(f->text); free(f); }
}
extern "C" int32_t mrpdr_fmt_nfields(const mrpdr_fmt *f) { return f ? f->dev.nf : 0; }
extern "C" int32_t mrpdr_fmt_field_type(const mrpdr_fmt *f, int32_t i)
{
    return (f && i >= 0 && i < f->dev.nf) ? f->dev.f[i].type : MRPDR_UNKNOWN_T;
}
extern "C" int32_t mrpdr_fmt_field_kind(const mrpdr_fmt *f, int32_t i)
{
    return (f && i >= 0 && i < f->dev.nf) ? f->dev.f[i].kind : -1;
}
extern "C" int32_t mrpdr_fmt_field_width(const mrpdr_fmt *f, int32_t i)
{
    return (f && i >= 0 && i < f->dev.nf) ? f->dev.f[i].width : 0;
}
extern "C" int32_t mrpdr_fmt_is_fixed(const mrpdr_fmt *f) { return f ? f->fixed : 0; }
extern "C" uint64_t mrpdr_fmt_fixed_bytes(const mrpdr_fmt *f) { return f ? f->dev.fixed_bytes : 0; }
extern "C" uint32_t mrpdr_fmt_payload_phase(const mrpdr_fmt *f, uint32_t align)
{
    if (!f || align == 0) return 0;
    for (int i = 0; i < f->dev.nf; i++)
        if (f->dev.f[i].kind != MRPDR_SCALAR) {
            const uint32_t at = f->dev.f[i].fixed_before + prefix_bytes(f->dev.f[i].kind);
            return (align - at % align) % align;
        }
    return 0;
}
extern "C" const char *mrpdr_fmt_string(const mrpdr_fmt *f) { return f ? f->text : ""; }

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct mrpdr_ctx {
    int32_t   device;
    uint64_t  max_packets;
    uint8_t  *ws;             // device workspace
    uint32_t *status;         // ws + 0
    uint64_t *tile_sums;      // scan partials
    uint64_t  ntile_sums;
    Partial  *partials;       // filter partials
    uint64_t  npartials;
    uint64_t *totals;         // per-field totals gathered for the host
    uint64_t *sa_bytes;       // per-field byte totals of string arrays (accumulated by the walk)
    uint64_t *pkt_bytes;      // string arrays: string bytes per packet, and their scan
    uint64_t *pkt_boffs;
    // chunk plans left by mrpdr_wave_encode_begin for mrpdr_wave_encode_finish
    struct {
        const mrpdr_fmt *fmt; uint64_t n; const uint8_t *wire_base;
        uint32_t used;                      // descriptors taken
        int32_t  closed;                    // a field did not fit: no further plans
        uint32_t off[MRPDR_MAXF], nd[MRPDR_MAXF];   // per field: where its plan starts, how long (0 = none)
    } plan;
    uint64_t *tile_cnt;       // tile kernels: element totals per (variable field, tile)
    uint64_t  ntile_cnt;      // tiles the workspace holds per field
    ChunkDesc *desc;          // chunk descriptors of the payload kernels
    int32_t   num_cu;
    // optional timing of the payload kernels (mrpdr_ctx_perf_*)
    int32_t     perf_on;
    hipEvent_t *perf_ev;      // PERF_SLOTS pairs
    uint8_t    *perf_kind;    // 0 = decode payload, 1 = encode payload
    uint32_t    perf_used;
    uint32_t    perf_seen[2];  // payload launches seen per kind since enable (sampling)
};

#define PERF_SLOTS 4096

#define DESC_CAP (1u << 18)   // descriptors per launch: 4 GiB of payload per window

extern "C" mrpdr_ctx *mrpdr_ctx_create(int32_t device, uint64_t max_packets)
{
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) { fail("mrpdr_ctx_create: no HIP device available"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail("mrpdr_ctx_create: hipSetDevice(%d) failed", device); return nullptr; }
    mrpdr_ctx *c = (mrpdr_ctx *)calloc(1, sizeof *c);
    c->device = device;
    c->max_packets = max_packets ? max_packets : 1;
    c->ntile_sums = (c->max_packets + SIZE_PKTS - 1) / SIZE_PKTS + 1;     // SIZE_PKTS < SCAN_TILE
    c->npartials = (c->max_packets + FILTER_TILE - 1) / FILTER_TILE + 1;
    const size_t o_status = 0, o_totals = 256;
    const size_t o_sums = o_totals + 256 * sizeof(uint64_t);
    const size_t o_part = o_sums + ((c->ntile_sums * sizeof(uint64_t) + 255) & ~(size_t)255);
    const size_t o_desc = (o_part + 3 * c->npartials * sizeof(Partial) + 255) & ~(size_t)255;   // x3: the stats filter
    const size_t o_pktb = o_desc + (size_t)DESC_CAP * sizeof(ChunkDesc);
    const size_t o_pkto = o_pktb + (((c->max_packets + 1) * sizeof(uint64_t) + 255) & ~(size_t)255);
    c->ntile_cnt = (c->max_packets + TILE_PKTS - 1) / TILE_PKTS + 1;
    const size_t o_tile = o_pkto + (((c->max_packets + 2) * sizeof(uint64_t) + 255) & ~(size_t)255);
    const size_t bytes = o_tile + (size_t)TILE_MAXV * c->ntile_cnt * sizeof(uint64_t);
    hipError_t e = hipMalloc((void **)&c->ws, bytes);
    if (e == hipSuccess) e = hipMemset(c->ws, 0, bytes);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        fail("mrpdr_ctx_create: %s", hipGetErrorString(e));
        free(c);
        return nullptr;
    }
    c->status = (uint32_t *)(c->ws + o_status);
    c->totals = (uint64_t *)(c->ws + o_totals);
    c->tile_sums = (uint64_t *)(c->ws + o_sums);
    c->partials = (Partial *)(c->ws + o_part);
    c->desc = (ChunkDesc *)(c->ws + o_desc);
    c->sa_bytes = c->totals + 128;
    c->pkt_bytes = (uint64_t *)(c->ws + o_pktb);
    c->pkt_boffs = (uint64_t *)(c->ws + o_pkto);
    c->tile_cnt = (uint64_t *)(c->ws + o_tile);
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return c;
}

extern "C" void mrpdr_ctx_destroy(mrpdr_ctx *c)
{
    if (!c) return;
    if (c->perf_ev) {
        for (uint32_t i = 0; i < 2 * PERF_SLOTS; i++) (void)hipEventDestroy(c->perf_ev[i]);
        free(c->perf_ev);
        free(c->perf_kind);
    }
    (void)hipFree(c->ws);
    free(c);
}

extern "C" int32_t mrpdr_ctx_status(mrpdr_ctx *c, void *stream, uint32_t *status_host)
{
    if (!c || !status_host) return fail("mrpdr_ctx_status: NULL argument");
    HIP_OK(hipMemcpyAsync(status_host, c->status, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_OK(hipMemsetAsync(c->status, 0, sizeof(uint32_t), (hipStream_t)stream));     // read and clear
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    return 1;
}

extern "C" int32_t mrpdr_ctx_perf_enable(mrpdr_ctx *c, int32_t on)
{
    if (!c) return fail("mrpdr_ctx_perf_enable: NULL ctx");
    if (on && !c->perf_ev) {
        c->perf_ev = (hipEvent_t *)calloc(2 * PERF_SLOTS, sizeof(hipEvent_t));
        c->perf_kind = (uint8_t *)calloc(PERF_SLOTS, 1);
        if (!c->perf_ev || !c->perf_kind) return fail("mrpdr_ctx_perf_enable: out of memory");
        for (uint32_t i = 0; i < 2 * PERF_SLOTS; i++)   // timing events without the system-scope cache flush
            HIP_OK(hipEventCreateWithFlags(&c->perf_ev[i], hipEventDisableSystemFence));
    }
    c->perf_on = on > 0 ? on : 0;          // sampling period: every on-th payload launch of a kind
    c->perf_used = 0;
    c->perf_seen[0] = c->perf_seen[1] = 0;
    return 1;
}

extern "C" int32_t mrpdr_ctx_perf_collect(mrpdr_ctx *c, void *stream, double *ms_sum_host, uint64_t *launches_host)
{
    if (!c || !ms_sum_host || !launches_host) return fail("mrpdr_ctx_perf_collect: NULL argument");
    ms_sum_host[0] = ms_sum_host[1] = 0.0;
    launches_host[0] = launches_host[1] = 0;
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    for (uint32_t i = 0; i < c->perf_used; i++) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, c->perf_ev[2 * i], c->perf_ev[2 * i + 1]));
        ms_sum_host[c->perf_kind[i]] += ms;
        launches_host[c->perf_kind[i]]++;
    }
    c->perf_used = 0;
    return 1;
}

extern "C" int32_t mrpdr_ctx_perf_plain_copy(mrpdr_ctx *c, void *dst, const void *src, uint64_t bytes,
                                             int32_t reps, double *ms_avg_host, void *stream)
{
    if (!c || !dst || !src || !ms_avg_host) return fail("mrpdr_ctx_perf_plain_copy: NULL argument");
    if (bytes == 0 || bytes % 16 || ((uintptr_t)dst & 15u) || ((uintptr_t)src & 15u))
        return fail("mrpdr_ctx_perf_plain_copy: bytes and both pointers must be multiples of 16");
    if (reps < 1) return fail("mrpdr_ctx_perf_plain_copy: reps must be positive");
    const uint64_t chunks = (bytes + COPY_CHUNK - 1) / COPY_CHUNK;
    if (chunks > 0x7FFFFFFFull) return fail("mrpdr_ctx_perf_plain_copy: more than 32 TiB");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    k_plain_copy<<<(uint32_t)chunks, MRPDR_TPB, 0, st>>>((uint8_t *)dst, (const uint8_t *)src, bytes);   // warm-up
    HIP_OK(hipEventRecord(e0, st));
    for (int32_t r = 0; r < reps; r++)
        k_plain_copy<<<(uint32_t)chunks, MRPDR_TPB, 0, st>>>((uint8_t *)dst, (const uint8_t *)src, bytes);
    HIP_OK(hipEventRecord(e1, st));
    LAUNCHED("k_plain_copy");
    HIP_OK(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_avg_host = (double)ms / reps;
    return 1;
}

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
static inline uint32_t blocks_for(uint64_t n, uint32_t per_block)
{
    return (uint32_t)((n + per_block - 1) / per_block);
}

// The workspace of a ctx lives on one device, and kernels launch on the calling thread's current
// device: the two must be the same (a kernel given another device's pointers faults).
static int32_t ctx_ready(const char *who, const mrpdr_ctx *c)
{
    if (!c) return fail("%s: NULL ctx", who);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail("%s: no current HIP device", who);
    if (cur != c->device)
        return fail("%s: the ctx belongs to device %d but device %d is current (hipSetDevice first)", who, c->device, cur);
    return 1;
}

static int32_t check_common(const char *who, const mrpdr_ctx *c, const mrpdr_fmt *f, uint64_t n)
{
    if (!ctx_ready(who, c)) return 0;
    if (!f) return fail("%s: NULL format", who);
    if (n > c->max_packets) return fail("%s: n=%llu exceeds the ctx limit %llu", who,
                                        (unsigned long long)n, (unsigned long long)c->max_packets);
    return 1;
}

static int32_t fill_cols(const char *who, const mrpdr_fmt *f, const mrpdr_column *cols, uint64_t n,
                         bool need_values, ColsDev *out, bool need_sub = true)
{
    memset(out, 0, sizeof *out);
    if (f->dev.nf && !cols) return fail("%s: NULL columns", who);
    for (int i = 0; i < f->dev.nf; i++) {
        const FieldDev &fd = f->dev.f[i];
        out->values[i] = cols[i].values;
        out->counts[i] = cols[i].counts;
        out->offs[i] = cols[i].offs;
        out->cap[i] = cols[i].capacity;
        out->sub_counts[i] = cols[i].sub_counts;
        out->sub_offs[i] = cols[i].sub_offs;
        out->sub_cap[i] = cols[i].sub_capacity;
        if (fd.kind == MRPDR_SCALAR) {
            if (n && !cols[i].values) return fail("%s: field %d has no values buffer", who, i);
        } else {
            if (n && (!cols[i].counts || !cols[i].offs)) return fail("%s: field %d needs counts and offs", who, i);
            if (need_values && !cols[i].values && cols[i].capacity)
                return fail("%s: field %d has no values buffer", who, i);
            if (((uintptr_t)cols[i].values & 15u) != 0)
                return fail("%s: field %d values must be 16-byte aligned", who, i);
            if (fd.kind == MRPDR_STRING_ARRAY && n && need_sub && !cols[i].sub_offs)
                return fail("%s: field %d (string array) needs sub_offs", who, i);
        }
    }
    return 1;
}

static int32_t fill_wire(const char *who, const mrpdr_fmt *f, const mrpdr_wire *w, uint64_t n, WireDev *out)
{
    if (!w) return fail("%s: NULL wire", who);
    if (n && !w->base) return fail("%s: NULL arena", who);
    out->base = w->base; out->offsets = w->offsets; out->lens = w->lens;
    out->stride = w->stride; out->bytes = w->bytes; out->orders = nullptr;
    if (n && !w->lens && !f->fixed) return fail("%s: lens is required for variable-length formats", who);
    if (n && !w->offsets) {
        if (!f->fixed) return fail("%s: offsets are required for variable-length formats", who);
        if (w->stride < f->dev.fixed_bytes)
            return fail("%s: stride %llu is shorter than the packet (%u bytes)", who,
                        (unsigned long long)w->stride, f->dev.fixed_bytes);
        if (n && w->stride > w->bytes / n) return fail("%s: arena too small for %llu packets", who, (unsigned long long)n);
    }
    return 1;
}

// exclusive scan of in[n] (rounded up to align) into out[n+1]
static int32_t run_scan(mrpdr_ctx *c, const uint64_t *in, uint64_t n, uint32_t align, uint32_t phase,
                        uint64_t *out, hipStream_t st)
{
    if (n <= SCAN1_MAX_N) {
        k_scan_single<false><<<1, SCAN1_TPB, 0, st>>>(in, FmtDev(), ColsDev(), n, align, phase, nullptr, out, nullptr);
        LAUNCHED("k_scan_single");
        return 1;
    }
    const uint32_t tiles = n ? blocks_for(n, SCAN_TILE) : 1;
    if (n) {
        k_scan_partial<<<tiles, MRPDR_TPB, 0, st>>>(in, n, align, phase, c->tile_sums);
        LAUNCHED("k_scan_partial");
    }
    k_scan_final<<<tiles, MRPDR_TPB, 0, st>>>(in, n, align, phase, c->tile_sums, out);
    LAUNCHED("k_scan_final");
    return 1;
}

static bool rows_path_ok(const mrpdr_fmt *f, const WireDev &w)
{
    return f->fixed && !w.offsets && w.stride <= ROWS_MAX_STRIDE && w.stride > 0 &&
           ((uintptr_t)w.base & 15u) == 0;
}

// Waves of short packets (a few hundred bytes on average) take the LDS-staged tile kernels.
// What has to fit the image is the span of 64 consecutive packets in the arena, whatever lies
// between them (alignment padding; the packet headers when the arena is a Message): the arena's
// bytes per packet stand for it, with one part in six of slack for tiles above the average.
#define TILE_AVG_BYTES 320
#define TILE_AVG_BYTES_WIDE 448
static bool tile_path_ok(const mrpdr_fmt *f, const WireDev &w, uint64_t n)
{
    return !f->fixed && f->nstrarr == 0 && f->nvar <= TILE_MAXV && w.offsets && w.lens && n > 0 &&
           w.bytes / n <= TILE_AVG_BYTES_WIDE;
}
static bool tile_wide(const WireDev &w, uint64_t n) { return w.bytes / n > TILE_AVG_BYTES; }

// Calls launch(V, IMG) with the format's table size and the wave's image size as constants.
template <typename F>
static void tile_variant(bool small, bool wide, F &&launch)
{
    typedef std::integral_constant<int, TILE_SMALLV> vs;
    typedef std::integral_constant<int, TILE_MAXV> vm;
    typedef std::integral_constant<int, TILE_IMG> in;
    typedef std::integral_constant<int, TILE_IMG_WIDE> iw;
    if (small) { if (wide) launch(vs(), iw()); else launch(vs(), in()); }
    else       { if (wide) launch(vm(), iw()); else launch(vm(), in()); }
}

static int32_t perf_begin(mrpdr_ctx *c, hipStream_t st, bool *timed, int kind)
{
    *timed = c->perf_on && c->perf_used < PERF_SLOTS && (c->perf_seen[kind]++ % (uint32_t)c->perf_on) == 0;
    if (*timed) HIP_OK(hipEventRecord(c->perf_ev[2 * c->perf_used], st));
    return 1;
}
static int32_t perf_end(mrpdr_ctx *c, hipStream_t st, bool timed, int kind)
{
    if (timed) {
        HIP_OK(hipEventRecord(c->perf_ev[2 * c->perf_used + 1], st));
        c->perf_kind[c->perf_used++] = (uint8_t)kind;
    }
    return 1;
}

// ---------------------------------------------------------------------------
// sizing
// ---------------------------------------------------------------------------
extern "C" int32_t mrpdr_wave_sizeof(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_column *cols,
                                     uint64_t n, uint32_t align, uint32_t phase, uint64_t *lens,
                                     uint64_t *offsets, uint64_t *total_host, void *stream)
{
    if (!check_common("mrpdr_wave_sizeof", c, f, n)) return 0;
    if (align == 0 || align > 256 || (align & (align - 1))) return fail("mrpdr_wave_sizeof: align must be a power of two <= 256");
    if (!offsets) return fail("mrpdr_wave_sizeof: NULL offsets");
    hipStream_t st = (hipStream_t)stream;
    if (f->fixed) {
        const uint64_t len = f->dev.fixed_bytes;
        const uint64_t stride = (len + phase + align - 1) & ~(uint64_t)(align - 1);
        k_offsets_fixed<<<blocks_for(n + 1, MRPDR_TPB), MRPDR_TPB, 0, st>>>(n, len, stride, phase, lens, offsets);
        LAUNCHED("k_offsets_fixed");
        if (total_host) *total_host = n * stride;
        return 1;
    }
    if (n && !lens) return fail("mrpdr_wave_sizeof: NULL lens");
    ColsDev cd;
    if (!fill_cols("mrpdr_wave_sizeof", f, cols, n, false, &cd)) return 0;
    if (n <= SCAN1_MAX_N) {
        // sizing and scan fused in one launch
        k_scan_single<true><<<1, SCAN1_TPB, 0, st>>>(nullptr, f->dev, cd, n, align, phase, lens, offsets, c->status);
        LAUNCHED("k_scan_single");
    } else {
        const uint32_t blocks = blocks_for(n, SIZE_PKTS);
        k_size_sums<<<blocks, MRPDR_TPB, 0, st>>>(f->dev, cd, n, align, phase, lens, c->tile_sums, c->status);
        LAUNCHED("k_size_sums");
        k_size_offs<<<blocks, MRPDR_TPB, 0, st>>>(lens, n, align, phase, c->tile_sums, offsets);
        LAUNCHED("k_size_offs");
    }
    if (total_host) {
        HIP_OK(hipMemcpyAsync(total_host, offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    return 1;
}

// ---------------------------------------------------------------------------
// segment copies
// ---------------------------------------------------------------------------
enum { COPY_BOTH = 0, COPY_PLAN = 1, COPY_RUN = 2 };   // plan + copy; plan only (encode_begin); copy a planned field

template <int W, bool SWAP, bool ENCODE>
static int32_t launch_copy(mrpdr_ctx *c, CopyArgs a, const FmtDev &fmt, const ColsDev &cd, hipStream_t st,
                           int mode = COPY_BOTH)
{
    // upper bound of the payload: it cannot exceed the arena or the column
    const uint64_t bound = a.wire.bytes < a.dense_cap ? a.wire.bytes : a.dense_cap;
    const uint64_t nchunks = (bound + COPY_CHUNK - 1) / COPY_CHUNK;
    const bool by_packet = bound / a.n < 2048;
    if (mode == COPY_PLAN) {
        // the plan of this field behind the plans already taken, when it fits the descriptor buffer
        if (by_packet) return 1;
        if (c->plan.closed || nchunks > DESC_CAP - c->plan.used) { c->plan.closed = 1; return 1; }
        a.first_chunk = 0;
        k_copy_plan<ENCODE><<<blocks_for(nchunks, MRPDR_TPB), MRPDR_TPB, 0, st>>>(a, fmt, cd, c->desc + c->plan.used, nchunks);
        LAUNCHED("k_copy_plan");
        c->plan.off[a.field] = c->plan.used;
        c->plan.nd[a.field] = (uint32_t)nchunks;
        c->plan.used += (uint32_t)nchunks;
        return 1;
    }
    if (mode == COPY_RUN && c->plan.nd[a.field]) {
        const uint32_t nd = c->plan.nd[a.field];
        bool timed;
        if (!perf_begin(c, st, &timed, ENCODE ? 1 : 0)) return 0;
        a.first_chunk = 0;
        CopyFast fa = { c->desc + c->plan.off[a.field], a.wire.base, a.dense };
        if (bound >= (32ull << 20)) k_copy<W, SWAP, ENCODE, true><<<nd, MRPDR_TPB, 0, st>>>(fa, a, fmt, cd);
        else                        k_copy<W, SWAP, ENCODE, false><<<nd, MRPDR_TPB, 0, st>>>(fa, a, fmt, cd);
        LAUNCHED("k_copy");
        return perf_end(c, st, timed, ENCODE ? 1 : 0);
    }
    // short payloads (under 2 KiB per packet on average): 16 lanes per packet
    if (by_packet) {
        const uint64_t threads = a.n * PKT_LANES;
        bool timed;
        if (!perf_begin(c, st, &timed, ENCODE ? 1 : 0)) return 0;
        k_copy_packets<W, SWAP, ENCODE><<<blocks_for(threads, MRPDR_TPB), MRPDR_TPB, 0, st>>>(a, fmt, cd);
        LAUNCHED("k_copy_packets");
        return perf_end(c, st, timed, ENCODE ? 1 : 0);
    }
    const uint64_t chunks = nchunks;
    // data this large is touched once per kernel: keep it out of the caches
    const bool nt = bound >= (32ull << 20);
    for (uint64_t first = 0; first < chunks; first += DESC_CAP) {
        const uint32_t nd = (uint32_t)((chunks - first) < DESC_CAP ? (chunks - first) : DESC_CAP);
        a.first_chunk = first;
        k_copy_plan<ENCODE><<<blocks_for(nd, MRPDR_TPB), MRPDR_TPB, 0, st>>>(a, fmt, cd, c->desc, nd);
        LAUNCHED("k_copy_plan");
        bool timed;
        if (!perf_begin(c, st, &timed, ENCODE ? 1 : 0)) return 0;
        CopyFast fa = { c->desc, a.wire.base, a.dense + first * COPY_CHUNK };
        if (nt) k_copy<W, SWAP, ENCODE, true><<<nd, MRPDR_TPB, 0, st>>>(fa, a, fmt, cd);
        else    k_copy<W, SWAP, ENCODE, false><<<nd, MRPDR_TPB, 0, st>>>(fa, a, fmt, cd);
        LAUNCHED("k_copy");
        if (!perf_end(c, st, timed, ENCODE ? 1 : 0)) return 0;
    }
    return 1;
}

// Rows kernels: four rows per lane for large waves of narrow rows (see rows.h).
static bool rows_wide(const WireDev &wd, uint64_t n)
{
    return wd.stride <= ROWS_WIDE_MAX_STRIDE && n >= ROWS_WIDE_MIN_PACKETS;
}

static CopyArgs make_copy_args(mrpdr_ctx *c, const mrpdr_fmt *f, const WireDev &wd, const ColsDev &cd, int field,
                               uint64_t n)
{
    CopyArgs a;
    a.wire = wd;
    a.dense = (uint8_t *)cd.values[field];
    a.seg_offs = cd.offs[field];
    a.n = n;
    a.dense_cap = cd.cap[field] * f->dev.f[field].width;
    a.status = c->status;
    a.field = field;
    a.first_chunk = 0;
    a.pos_fixed = f->dev.f[field].fixed_before + prefix_bytes(f->dev.f[field].kind);
    a.var_before = 0;
    for (int g = 0; g < field; g++)
        if (f->dev.f[g].kind != MRPDR_SCALAR) a.var_before |= 1u << g;
    return a;
}

// a column without room while the wave holds elements of the field: status bit 0
__global__ void k_cap_check(const uint64_t *offs, uint64_t n, uint32_t *status)
{
    if (offs[n] != 0) atomicOr(status, 1u);
}

template <bool ENCODE>
static int32_t copy_field(mrpdr_ctx *c, const mrpdr_fmt *f, const WireDev &wd, const ColsDev &cd, int field,
                          uint64_t n, bool swap, hipStream_t st, int mode = COPY_BOTH)
{
    const FieldDev &fd = f->dev.f[field];
    CopyArgs a = make_copy_args(c, f, wd, cd, field, n);
    if (n == 0) return 1;
    if (a.dense_cap == 0) {
        // nothing can be moved; a decode must still say so when the wave does hold elements
        if (!ENCODE) {
            k_cap_check<<<1, 1, 0, st>>>(a.seg_offs, n, c->status);
            LAUNCHED("k_cap_check");
        }
        return 1;
    }
    // raw bytes and strings are never reversed (src/pdr_mem.c:394-412)
    const bool sw = swap && fd.width > 1 && !ENCODE;
    switch (fd.width) {
    case 1: return launch_copy<1, false, ENCODE>(c, a, f->dev, cd, st, mode);
    case 2: return sw ? launch_copy<2, true, ENCODE>(c, a, f->dev, cd, st, mode) : launch_copy<2, false, ENCODE>(c, a, f->dev, cd, st, mode);
    case 4: return sw ? launch_copy<4, true, ENCODE>(c, a, f->dev, cd, st, mode) : launch_copy<4, false, ENCODE>(c, a, f->dev, cd, st, mode);
    default: return sw ? launch_copy<8, true, ENCODE>(c, a, f->dev, cd, st, mode) : launch_copy<8, false, ENCODE>(c, a, f->dev, cd, st, mode);
    }
}

// ---------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------
// phase 1 = begin (everything but the payload streams), 2 = finish, 3 = both
static int32_t encode_phases(const char *who, mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_column *cols,
                             uint64_t n, const mrpdr_wire *wire, void *stream, int phases)
{
    if (!check_common(who, c, f, n)) return 0;
    ColsDev cd;
    WireDev wd;
    if (!fill_cols(who, f, cols, n, true, &cd)) return 0;
    if (!fill_wire(who, f, wire, n, &wd)) return 0;
    if (n == 0 || f->dev.nf == 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    // one-kernel paths have nothing to prepare: they run in the finish phase
    if (rows_path_ok(f, wd)) {
        if (!(phases & 2)) return 1;
        if (rows_wide(wd, n))
            k_rows_encode<ROWS_WIDE><<<blocks_for(n, ROWS_PER_BLOCK * ROWS_WIDE), MRPDR_TPB, ROWS_PER_BLOCK * ROWS_WIDE * wd.stride, st>>>(f->dev, wd, cd, n);
        else
            k_rows_encode<1><<<blocks_for(n, ROWS_PER_BLOCK), MRPDR_TPB, ROWS_PER_BLOCK * wd.stride, st>>>(f->dev, wd, cd, n);
        LAUNCHED("k_rows_encode");
        return 1;
    }
    if (tile_path_ok(f, wd, n)) {
        if (!(phases & 2)) return 1;
        bool timed;
        if (!perf_begin(c, st, &timed, 1)) return 0;
        tile_variant(f->tv.nv <= TILE_SMALLV, tile_wide(wd, n), [&](auto v, auto img) {
            k_tile_encode<decltype(v)::value, decltype(img)::value><<<blocks_for(n, TILE_PKTS), TILE_TPB, 0, st>>>(f->dev, tile_vars_of(f->tv, cd), wd, cd, n);
        });
        LAUNCHED("k_tile_encode");
        return perf_end(c, st, timed, 1);
    }
    if (phases & 1) {
        memset(&c->plan, 0, sizeof c->plan);
        c->plan.fmt = f; c->plan.n = n; c->plan.wire_base = wd.base;
        // the first field whose payload goes through the chunk kernel: its plan rides along with
        // the prefix walk, one launch for both
        int fused = -1;
        uint64_t nch = 0;
        for (int i = 0; i < f->dev.nf && fused < 0; i++) {
            if (f->dev.f[i].kind != MRPDR_ARRAY && f->dev.f[i].kind != MRPDR_STRING) continue;
            const uint64_t cap = cd.cap[i] * f->dev.f[i].width;
            const uint64_t bound = wd.bytes < cap ? wd.bytes : cap;
            if (cap == 0 || bound / n < 2048) continue;        // no payload / the packet-centric kernel
            nch = (bound + COPY_CHUNK - 1) / COPY_CHUNK;
            if (nch <= DESC_CAP) fused = i;
            break;
        }
        if (fused >= 0) {
            const uint64_t lanes = n > nch ? n : nch;
            k_walk_plan_encode<<<blocks_for(lanes, MRPDR_TPB), MRPDR_TPB, 0, st>>>(
                f->dev, wd, cd, n, make_copy_args(c, f, wd, cd, fused, n), c->desc, nch);
            LAUNCHED("k_walk_plan_encode");
            c->plan.off[fused] = 0;
            c->plan.nd[fused] = (uint32_t)nch;
            c->plan.used = (uint32_t)nch;
        } else {
            k_walk_encode<<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, st>>>(f->dev, wd, cd, n);
            LAUNCHED("k_walk_encode");
        }
        if (phases == 1)            // the other plans for the finish call; a combined call plans them as it goes
            for (int i = 0; i < f->dev.nf; i++)
                if (i != fused && (f->dev.f[i].kind == MRPDR_ARRAY || f->dev.f[i].kind == MRPDR_STRING))
                    if (!copy_field<true>(c, f, wd, cd, i, n, false, st, COPY_PLAN)) return 0;
    }
    if (!(phases & 2)) return 1;
    if (phases == 2 && (c->plan.fmt != f || c->plan.n != n || c->plan.wire_base != wd.base))
        return fail("%s: no matching mrpdr_wave_encode_begin on this ctx", who);
    for (int i = 0; i < f->dev.nf; i++) {
        if (f->dev.f[i].kind == MRPDR_STRING_ARRAY) {
            k_strarr_encode<<<blocks_for(n * 64, MRPDR_TPB), MRPDR_TPB, 0, st>>>(f->dev, wd, cd, n, i);
            LAUNCHED("k_strarr_encode");
        } else if (f->dev.f[i].kind != MRPDR_SCALAR) {
            if (!copy_field<true>(c, f, wd, cd, i, n, false, st, (phases == 2 || c->plan.nd[i]) ? COPY_RUN : COPY_BOTH)) return 0;
        }
    }
    if (phases == 2) c->plan.fmt = nullptr;
    return 1;
}

extern "C" int32_t mrpdr_wave_encode(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_column *cols,
                                     uint64_t n, const mrpdr_wire *wire, void *stream)
{
    return encode_phases("mrpdr_wave_encode", c, f, cols, n, wire, stream, 3);
}

extern "C" int32_t mrpdr_wave_encode_begin(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_column *cols,
                                           uint64_t n, const mrpdr_wire *wire, void *stream)
{
    return encode_phases("mrpdr_wave_encode_begin", c, f, cols, n, wire, stream, 1);
}

extern "C" int32_t mrpdr_wave_encode_finish(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_column *cols,
                                            uint64_t n, const mrpdr_wire *wire, void *stream)
{
    return encode_phases("mrpdr_wave_encode_finish", c, f, cols, n, wire, stream, 2);
}

// ---------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------
// The three launches of the tile decode.
static int32_t launch_tile_walk(mrpdr_ctx *c, const mrpdr_fmt *f, const WireDev &wd, const ColsDev &cd, uint64_t n,
                                bool swap, int32_t *ok, hipStream_t st)
{
    const uint32_t tiles = blocks_for(n, TILE_PKTS);
    const bool small = f->tv.nv <= TILE_SMALLV;
    const bool wide = tile_wide(wd, n);
    tile_variant(small, wide, [&](auto v, auto img) {
        constexpr int V = decltype(v)::value, IMG = decltype(img)::value;
        if (swap) k_tile_walk<true, V, IMG><<<tiles, TILE_TPB, 0, st>>>(f->dev, tile_vars_of(f->tv, cd), wd, cd, n, ok, c->tile_cnt, c->ntile_cnt);
        else      k_tile_walk<false, V, IMG><<<tiles, TILE_TPB, 0, st>>>(f->dev, tile_vars_of(f->tv, cd), wd, cd, n, ok, c->tile_cnt, c->ntile_cnt);
    });
    LAUNCHED("k_tile_walk");
    return 1;
}

static int32_t launch_tile_offs(mrpdr_ctx *c, const mrpdr_fmt *f, const ColsDev &cd, uint64_t n, uint32_t pkts_per_sum,
                                hipStream_t st)
{
    k_tile_offs<<<dim3(blocks_for(n, TILE_OFFS_PKTS), (uint32_t)f->tv.nv), MRPDR_TPB, 0, st>>>(tile_vars_of(f->tv, cd), cd, n, c->tile_cnt,
                                                                                                 c->ntile_cnt, pkts_per_sum);
    LAUNCHED("k_tile_offs");
    return 1;
}

static int32_t launch_tile_decode(mrpdr_ctx *c, const mrpdr_fmt *f, const WireDev &wd, const ColsDev &cd, uint64_t n,
                                  bool swap, hipStream_t st, const uint8_t *nvalid = nullptr)
{
    bool timed;
    if (!perf_begin(c, st, &timed, 0)) return 0;
    const bool small = f->tv.nv <= TILE_SMALLV;
    const uint32_t tiles = blocks_for(n, TILE_PKTS);
    const bool wide = tile_wide(wd, n);
    tile_variant(small, wide, [&](auto v, auto img) {
        constexpr int V = decltype(v)::value, IMG = decltype(img)::value;
        if (swap) k_tile_decode<true, V, IMG><<<tiles, TILE_TPB, 0, st>>>(f->dev, tile_vars_of(f->tv, cd), wd, cd, n, c->status, nvalid);
        else      k_tile_decode<false, V, IMG><<<tiles, TILE_TPB, 0, st>>>(f->dev, tile_vars_of(f->tv, cd), wd, cd, n, c->status, nvalid);
    });
    LAUNCHED("k_tile_decode");
    return perf_end(c, st, timed, 0);
}

// orders (optional, device, n bytes): the sender byte order of every packet; sender_bo is then unused.
// Such a wave takes the one-lane-per-packet walk and the chunk copy (no LDS-staged rows or tiles):
// waves of both kinds come from trees that span machine types, not from the bandwidth-bound case.
static int32_t decode_begin(const char *who, mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire,
                            uint64_t n, int32_t sender_bo, const uint8_t *orders, mrpdr_column *cols,
                            int32_t *ok, uint64_t *totals_host, void *stream)
{
    if (!check_common(who, c, f, n)) return 0;
    if (sender_bo != MRPDR_LITTLEENDIAN && sender_bo != MRPDR_BIGENDIAN)
        return fail("%s: bad byte order %d", who, sender_bo);
    ColsDev cd;
    WireDev wd;
    if (!fill_cols(who, f, cols, n, false, &cd, false)) return 0;
    if (!fill_wire(who, f, wire, n, &wd)) return 0;
    wd.orders = orders;
    hipStream_t st = (hipStream_t)stream;
    const bool swap = sender_bo != MRPDR_LITTLEENDIAN;
    if (totals_host) memset(totals_host, 0, 2 * sizeof(uint64_t) * (size_t)f->dev.nf);
    if (f->dev.nf == 0) return 1;
    if (f->nstrarr) HIP_OK(hipMemsetAsync(c->sa_bytes, 0, MRPDR_MAXF * sizeof(uint64_t), st));
    if (n) {
        if (rows_path_ok(f, wd)) {
            if (rows_wide(wd, n)) {
                const size_t lds = ROWS_PER_BLOCK * ROWS_WIDE * wd.stride;
                const uint32_t nb = blocks_for(n, ROWS_PER_BLOCK * ROWS_WIDE);
                if (swap) k_rows_decode<true, ROWS_WIDE><<<nb, MRPDR_TPB, lds, st>>>(f->dev, wd, cd, n, ok);
                else      k_rows_decode<false, ROWS_WIDE><<<nb, MRPDR_TPB, lds, st>>>(f->dev, wd, cd, n, ok);
            } else {
                const size_t lds = ROWS_PER_BLOCK * wd.stride;
                if (swap) k_rows_decode<true, 1><<<blocks_for(n, ROWS_PER_BLOCK), MRPDR_TPB, lds, st>>>(f->dev, wd, cd, n, ok);
                else      k_rows_decode<false, 1><<<blocks_for(n, ROWS_PER_BLOCK), MRPDR_TPB, lds, st>>>(f->dev, wd, cd, n, ok);
            }
            LAUNCHED("k_rows_decode");
        } else if (!orders && tile_path_ok(f, wd, n)) {
            if (!launch_tile_walk(c, f, wd, cd, n, swap, ok, st)) return 0;
            if (!launch_tile_offs(c, f, cd, n, TILE_PKTS, st)) return 0;
            for (int i = 0; totals_host && i < f->dev.nf; i++)
                if (f->dev.f[i].kind != MRPDR_SCALAR)
                    HIP_OK(hipMemcpyAsync(&totals_host[i], cd.offs[i] + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
            if (totals_host) HIP_OK(hipStreamSynchronize(st));
            return 1;
        } else {
            if (swap) k_walk_decode<true><<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, st>>>(f->dev, wd, cd, n, ok, c->sa_bytes);
            else      k_walk_decode<false><<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, st>>>(f->dev, wd, cd, n, ok, c->sa_bytes);
            LAUNCHED("k_walk_decode");
        }
    }
    if (f->fixed) return 1;
    for (int i = 0; i < f->dev.nf; i++) {
        if (f->dev.f[i].kind == MRPDR_SCALAR) continue;
        if (!run_scan(c, cd.counts[i], n, 1, 0, cd.offs[i], st)) return 0;
        if (totals_host) {
            HIP_OK(hipMemcpyAsync(&totals_host[i], cd.offs[i] + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
            if (f->dev.f[i].kind == MRPDR_STRING_ARRAY)
                HIP_OK(hipMemcpyAsync(&totals_host[f->dev.nf + i], c->sa_bytes + i, sizeof(uint64_t),
                                      hipMemcpyDeviceToHost, st));
        }
    }
    if (totals_host) HIP_OK(hipStreamSynchronize(st));
    return 1;
}

static int32_t decode_finish(const char *who, mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire,
                             uint64_t n, int32_t sender_bo, const uint8_t *orders, mrpdr_column *cols, void *stream)
{
    if (!check_common(who, c, f, n)) return 0;
    if (n == 0) {                                   // an empty wave still has sub_offs[0] = 0
        for (int i = 0; cols && i < f->dev.nf; i++)
            if (f->dev.f[i].kind == MRPDR_STRING_ARRAY && cols[i].sub_offs)
                HIP_OK(hipMemsetAsync(cols[i].sub_offs, 0, sizeof(uint64_t), (hipStream_t)stream));
        return 1;
    }
    if (f->fixed) return 1;
    ColsDev cd;
    WireDev wd;
    if (!fill_cols(who, f, cols, n, true, &cd)) return 0;
    if (!fill_wire(who, f, wire, n, &wd)) return 0;
    wd.orders = orders;
    hipStream_t st = (hipStream_t)stream;
    const bool swap = !orders && sender_bo != MRPDR_LITTLEENDIAN;
    if (!orders && tile_path_ok(f, wd, n)) {
        return launch_tile_decode(c, f, wd, cd, n, swap, st);
    }
    for (int i = 0; i < f->dev.nf; i++) {
        if (f->dev.f[i].kind == MRPDR_STRING_ARRAY) {
            if (!cd.sub_counts[i]) return fail("%s: field %d (string array) needs sub_counts", who, i);
            if (swap) k_strarr_sizes<true><<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, st>>>(f->dev, wd, cd, n, i, c->pkt_bytes, c->status);
            else      k_strarr_sizes<false><<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, st>>>(f->dev, wd, cd, n, i, c->pkt_bytes, c->status);
            LAUNCHED("k_strarr_sizes");
            if (!run_scan(c, c->pkt_bytes, n, 1, 0, c->pkt_boffs, st)) return 0;
            if (swap) k_strarr_decode<true><<<blocks_for(n * 64, MRPDR_TPB), MRPDR_TPB, 0, st>>>(f->dev, wd, cd, n, i, c->pkt_boffs, c->status);
            else      k_strarr_decode<false><<<blocks_for(n * 64, MRPDR_TPB), MRPDR_TPB, 0, st>>>(f->dev, wd, cd, n, i, c->pkt_boffs, c->status);
            LAUNCHED("k_strarr_decode");
        } else if (f->dev.f[i].kind != MRPDR_SCALAR) {
            if (!copy_field<false>(c, f, wd, cd, i, n, swap, st)) return 0;
            const uint32_t w = f->dev.f[i].width;
            if (orders && w > 1 && cd.cap[i]) {        // reverse the elements of the foreign packets in place
                const uint32_t blocks = blocks_for(n * 64, MRPDR_TPB);
                uint8_t *v = (uint8_t *)cd.values[i];
                if (w == 2)      k_swap_foreign<2><<<blocks, MRPDR_TPB, 0, st>>>(v, cd.offs[i], orders, n, cd.cap[i] * w);
                else if (w == 4) k_swap_foreign<4><<<blocks, MRPDR_TPB, 0, st>>>(v, cd.offs[i], orders, n, cd.cap[i] * w);
                else             k_swap_foreign<8><<<blocks, MRPDR_TPB, 0, st>>>(v, cd.offs[i], orders, n, cd.cap[i] * w);
                LAUNCHED("k_swap_foreign");
            }
        }
    }
    return 1;
}

extern "C" int32_t mrpdr_wave_decode_begin(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                           uint64_t n, int32_t sender_bo, mrpdr_column *cols,
                                           int32_t *ok, uint64_t *totals_host, void *stream)
{
    return decode_begin("mrpdr_wave_decode_begin", c, f, wire, n, sender_bo, nullptr, cols, ok, totals_host, stream);
}

extern "C" int32_t mrpdr_wave_decode_finish(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                            uint64_t n, int32_t sender_bo, mrpdr_column *cols, void *stream)
{
    return decode_finish("mrpdr_wave_decode_finish", c, f, wire, n, sender_bo, nullptr, cols, stream);
}

// Waves of short packets decoded by ONE call: the first pass reads only the length prefixes
// (k_count_walk), the payload kernel writes the scalar columns as well -- the wire is read once.
// Returns -1 when the wave is not of that kind (the caller takes begin + finish).
static int32_t decode_short_packets(const char *who, mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire, uint64_t n,
                                    int32_t sender_bo, mrpdr_column *cols, int32_t *ok, void *stream)
{
    if (!c || !f || !wire || !cols || f->fixed || n == 0 || n > c->max_packets || f->dev.nf > 255) return -1;
    if (sender_bo != MRPDR_LITTLEENDIAN && sender_bo != MRPDR_BIGENDIAN) return -1;
    if (!check_common(who, c, f, n)) return 0;
    ColsDev cd;
    WireDev wd;
    if (!fill_cols(who, f, cols, n, true, &cd)) return 0;
    if (!fill_wire(who, f, wire, n, &wd)) return 0;
    if (!tile_path_ok(f, wd, n)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const bool swap = sender_bo != MRPDR_LITTLEENDIAN;
    uint8_t *nvalid = (uint8_t *)c->pkt_bytes;             // one byte per packet of the ctx workspace
    if (swap) k_count_walk<true><<<blocks_for(n, COUNT_WALK_TPB), COUNT_WALK_TPB, 0, st>>>(f->dev, wd, cd, n, ok, nvalid, c->tile_cnt, c->ntile_cnt);
    else      k_count_walk<false><<<blocks_for(n, COUNT_WALK_TPB), COUNT_WALK_TPB, 0, st>>>(f->dev, wd, cd, n, ok, nvalid, c->tile_cnt, c->ntile_cnt);
    LAUNCHED("k_count_walk");
    if (!launch_tile_offs(c, f, cd, n, COUNT_WALK_TPB, st)) return 0;
    return launch_tile_decode(c, f, wd, cd, n, swap, st, nvalid);
}

extern "C" int32_t mrpdr_wave_decode(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                     uint64_t n, int32_t sender_bo, mrpdr_column *cols,
                                     int32_t *ok, void *stream)
{
    const int32_t one = decode_short_packets("mrpdr_wave_decode", c, f, wire, n, sender_bo, cols, ok, stream);
    if (one >= 0) return one;
    if (!decode_begin("mrpdr_wave_decode", c, f, wire, n, sender_bo, nullptr, cols, ok, nullptr, stream)) return 0;
    return decode_finish("mrpdr_wave_decode", c, f, wire, n, sender_bo, nullptr, cols, stream);
}

extern "C" int32_t mrpdr_wave_decode_mixed_begin(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                                 uint64_t n, const uint8_t *byteorders, mrpdr_column *cols,
                                                 int32_t *ok, uint64_t *totals_host, void *stream)
{
    if (n && !byteorders) return fail("mrpdr_wave_decode_mixed_begin: NULL byteorders");
    return decode_begin("mrpdr_wave_decode_mixed_begin", c, f, wire, n, 0, n ? byteorders : nullptr, cols, ok,
                        totals_host, stream);
}

extern "C" int32_t mrpdr_wave_decode_mixed_finish(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                                  uint64_t n, const uint8_t *byteorders, mrpdr_column *cols,
                                                  void *stream)
{
    if (n && !byteorders) return fail("mrpdr_wave_decode_mixed_finish: NULL byteorders");
    return decode_finish("mrpdr_wave_decode_mixed_finish", c, f, wire, n, 0, n ? byteorders : nullptr, cols, stream);
}

extern "C" int32_t mrpdr_wave_decode_mixed(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                           uint64_t n, const uint8_t *byteorders, mrpdr_column *cols,
                                           int32_t *ok, void *stream)
{
    if (!mrpdr_wave_decode_mixed_begin(c, f, wire, n, byteorders, cols, ok, nullptr, stream)) return 0;
    return mrpdr_wave_decode_mixed_finish(c, f, wire, n, byteorders, cols, stream);
}

// ---------------------------------------------------------------------------
// column views (host-side only: no device work)
// ---------------------------------------------------------------------------
extern "C" int32_t mrpdr_wave_view(const mrpdr_fmt *f, const char *fmt_expected, const mrpdr_column *cols, uint64_t n,
                                   mrpdr_field_view *views)
{
    if (!f || !fmt_expected || !views) return fail("mrpdr_wave_view: NULL argument");
    if (strcmp(fmt_expected, f->text) != 0)           // ExtractVaList, src/Packet.C:527-532
        return fail("mrpdr_wave_view: format '%s' is not the wave's '%s'", fmt_expected, f->text);
    if (f->dev.nf && !cols) return fail("mrpdr_wave_view: NULL columns");
    for (int i = 0; i < f->dev.nf; i++) {
        const FieldDev &fd = f->dev.f[i];
        if (n && fd.kind == MRPDR_SCALAR && !cols[i].values) return fail("mrpdr_wave_view: field %d has no values", i);
        if (n && fd.kind != MRPDR_SCALAR && (!cols[i].counts || !cols[i].offs))
            return fail("mrpdr_wave_view: field %d has no counts / offs", i);
    }
    for (int i = 0; i < f->dev.nf; i++) {
        const FieldDev &fd = f->dev.f[i];
        mrpdr_field_view &v = views[i];
        memset(&v, 0, sizeof v);
        v.values = cols[i].values;
        if (fd.kind != MRPDR_SCALAR) { v.counts = cols[i].counts; v.offs = cols[i].offs; }
        if (fd.kind == MRPDR_STRING_ARRAY) { v.sub_counts = cols[i].sub_counts; v.sub_offs = cols[i].sub_offs; }
        v.n = n;
        v.type = fd.type; v.kind = fd.kind; v.width = fd.width;
    }
    return 1;
}

// ---------------------------------------------------------------------------
// headers
// ---------------------------------------------------------------------------
static int32_t make_image(const char *who, const mrpdr_fmt *f, HdrImage *img)
{
    if (!f) return fail("%s: NULL format", who);
    size_t l = strlen(f->text) + 1;
    if (l > HDR_MAX_FMT) return fail("%s: format string longer than %d bytes", who, HDR_MAX_FMT - 1);
    memset(img, 0, sizeof *img);
    img->fmt_bytes = (uint32_t)l;
    memcpy(img->fmt, f->text, l);
    return 1;
}

extern "C" uint64_t mrpdr_header_bytes(const mrpdr_fmt *f)
{
    return f ? 16 + strlen(f->text) + 1 + 8 + 1 : 0;     // src/Packet.C:557-593
}

extern "C" uint64_t mrpdr_header_bytes_dest(const mrpdr_fmt *f, uint64_t ndest)
{
    return f ? mrpdr_header_bytes(f) + 4 * ndest : 0;
}

extern "C" int32_t mrpdr_wave_encode_headers(mrpdr_ctx *c, const mrpdr_fmt *f, uint64_t n,
                                             uint32_t stream_id, const uint32_t *stream_ids,
                                             int32_t tag, const int32_t *tags,
                                             uint32_t src_rank, const uint32_t *src_ranks,
                                             const uint32_t *dest_ranks, uint64_t ndest,
                                             uint8_t *hdrs, const uint64_t *hdr_offsets, uint64_t hdr_stride,
                                             void *stream)
{
    if (!ctx_ready("mrpdr_wave_encode_headers", c)) return 0;
    HdrImage img;
    if (!make_image("mrpdr_wave_encode_headers", f, &img)) return 0;
    if (ndest && !dest_ranks) return fail("mrpdr_wave_encode_headers: NULL destination list");
    if (!hdr_offsets && hdr_stride < mrpdr_header_bytes_dest(f, ndest)) return fail("mrpdr_wave_encode_headers: stride shorter than the header");
    if (n == 0) return 1;
    if (!hdrs) return fail("mrpdr_wave_encode_headers: NULL output");
    HdrEncodeArgs a = { stream_id, stream_ids, tag, tags, src_rank, src_ranks, dest_ranks, ndest,
                        hdrs, hdr_offsets, hdr_stride, n };
    k_hdr_encode<<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, (hipStream_t)stream>>>(a, img);
    LAUNCHED("k_hdr_encode");
    return 1;
}

extern "C" int32_t mrpdr_wave_decode_headers(mrpdr_ctx *c, const mrpdr_fmt *f, uint64_t n,
                                             const uint8_t *hdrs, uint64_t hdrs_bytes, const uint64_t *hdr_offsets,
                                             const uint64_t *hdr_lens, uint64_t hdr_stride,
                                             uint32_t *stream_ids, int32_t *tags, uint32_t *src_ranks,
                                             uint8_t *byteorders, uint64_t *ndests,
                                             uint32_t *dest_ranks, uint64_t dest_stride, int32_t *ok,
                                             void *stream)
{
    if (!ctx_ready("mrpdr_wave_decode_headers", c)) return 0;
    HdrImage img;
    if (!make_image("mrpdr_wave_decode_headers", f, &img)) return 0;
    if (n == 0) return 1;
    if (!hdrs) return fail("mrpdr_wave_decode_headers: NULL input");
    HdrDecodeArgs a = { hdrs, hdrs_bytes, hdr_offsets, hdr_lens, hdr_stride, stream_ids, tags, src_ranks,
                        byteorders, ndests, dest_ranks, dest_stride, ok, n };
    k_hdr_decode<<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, (hipStream_t)stream>>>(a, img);
    LAUNCHED("k_hdr_decode");
    return 1;
}

#ifdef MRPDR_TILE_STAMPS
// diagnostic build: where the tile kernels note their phase boundaries (8 x u64 per workgroup), or NULL
extern "C" int32_t mrpdr_debug_tile_stamps(void *buffer)
{
    unsigned long long *p = (unsigned long long *)buffer;
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), &p, sizeof p));
    return 1;
}
#endif

// ---------------------------------------------------------------------------
// message framing
// ---------------------------------------------------------------------------
extern "C" uint64_t mrpdr_message_preamble_bytes(uint64_t n) { return MSG_COUNT_BYTES + 16 * n + 1; }

extern "C" int32_t mrpdr_message_write_preamble(mrpdr_ctx *c, uint64_t n, uint64_t hdr_len, const uint64_t *lens,
                                                uint8_t *msg, void *stream)
{
    if (!ctx_ready("mrpdr_message_write_preamble", c)) return 0;
    if (!msg || (n && !lens)) return fail("mrpdr_message_write_preamble: NULL argument");
    if (n > 0xFFFFFFFFull) return fail("mrpdr_message_write_preamble: the packet count is a uint32 on the wire");
    k_msg_preamble<<<blocks_for(n + 1, MRPDR_TPB), MRPDR_TPB, 0, (hipStream_t)stream>>>(msg, n, hdr_len, lens);
    LAUNCHED("k_msg_preamble");
    return 1;
}

// Size vector -> places (and, with a.hdrs, the headers): two launches.
static int32_t message_places(const char *who, mrpdr_ctx *c, const uint8_t *sizes, uint64_t n,
                              uint64_t *hdr_offsets, uint64_t *hdr_lens, uint64_t *buf_offsets, uint64_t *buf_lens,
                              const HdrDecodeArgs &a, const HdrImage &img, hipStream_t st)
{
    if (n > c->max_packets) return fail("%s: n exceeds the ctx limit", who);
    if (!hdr_offsets || !buf_offsets || (n && (!sizes || !hdr_lens || !buf_lens))) return fail("%s: NULL argument", who);
    if (n == 0) {
        HIP_OK(hipMemsetAsync(hdr_offsets, 0, sizeof(uint64_t), st));
        HIP_OK(hipMemsetAsync(buf_offsets, 0, sizeof(uint64_t), st));
        return 1;
    }
    const uint64_t clamp = (1ull << 63) / (n + 1);
    const uint32_t blocks = blocks_for(n, MSG_BLOCK_PKTS);
    k_msg_sums<<<blocks, MRPDR_TPB, 0, st>>>(sizes, n, clamp, c->tile_cnt);
    LAUNCHED("k_msg_sums");
    k_msg_place_headers<<<blocks, MRPDR_TPB, 0, st>>>(sizes, n, clamp, c->tile_cnt, hdr_offsets, hdr_lens, buf_offsets, buf_lens, a, img);
    LAUNCHED("k_msg_place_headers");
    return 1;
}

extern "C" int32_t mrpdr_message_parse_sizes(mrpdr_ctx *c, const uint8_t *sizes, uint64_t n,
                                             uint64_t *hdr_offsets, uint64_t *hdr_lens,
                                             uint64_t *buf_offsets, uint64_t *buf_lens,
                                             uint64_t *payload_bytes_host, void *stream)
{
    if (!ctx_ready("mrpdr_message_parse_sizes", c)) return 0;
    hipStream_t st = (hipStream_t)stream;
    HdrDecodeArgs none;
    HdrImage img;
    memset(&none, 0, sizeof none);
    memset(&img, 0, sizeof img);
    if (!message_places("mrpdr_message_parse_sizes", c, sizes, n, hdr_offsets, hdr_lens, buf_offsets, buf_lens, none, img, st)) return 0;
    if (payload_bytes_host) {
        HIP_OK(hipMemcpyAsync(payload_bytes_host, hdr_offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    return 1;
}

extern "C" int32_t mrpdr_message_parse(mrpdr_ctx *c, const mrpdr_fmt *f, const uint8_t *msg, uint64_t msg_bytes, uint64_t n,
                                       uint64_t *hdr_offsets, uint64_t *hdr_lens,
                                       uint64_t *buf_offsets, uint64_t *buf_lens,
                                       uint32_t *stream_ids, int32_t *tags, uint32_t *src_ranks,
                                       uint8_t *byteorders, uint64_t *ndests,
                                       uint32_t *dest_ranks, uint64_t dest_stride, int32_t *hdr_ok, void *stream)
{
    const char *who = "mrpdr_message_parse";
    if (!ctx_ready(who, c)) return 0;
    HdrImage img;
    if (!make_image(who, f, &img)) return 0;
    if (!msg) return fail("%s: NULL message", who);
    if (n > 0xFFFFFFFFull) return fail("%s: the packet count is a uint32 on the wire", who);
    const uint64_t pre = mrpdr_message_preamble_bytes(n);
    if (pre > msg_bytes) return fail("%s: message shorter than its size vector", who);
    HdrDecodeArgs a = { msg + pre, msg_bytes - pre, nullptr, nullptr, 0, stream_ids, tags, src_ranks,
                        byteorders, ndests, dest_ranks, dest_stride, hdr_ok, n };
    return message_places(who, c, msg + MSG_COUNT_BYTES, n, hdr_offsets, hdr_lens, buf_offsets, buf_lens, a, img,
                          (hipStream_t)stream);
}

extern "C" int32_t mrpdr_message_write_frame(mrpdr_ctx *c, const mrpdr_fmt *f, uint64_t n,
                                             const uint64_t *lens, const uint64_t *offsets,
                                             uint32_t stream_id, const uint32_t *stream_ids,
                                             int32_t tag, const int32_t *tags,
                                             uint32_t src_rank, const uint32_t *src_ranks,
                                             uint8_t *msg, void *stream)
{
    const char *who = "mrpdr_message_write_frame";
    if (!ctx_ready(who, c)) return 0;
    HdrImage img;
    if (!make_image(who, f, &img)) return 0;
    if (!msg || (n && (!lens || !offsets))) return fail("%s: NULL argument", who);
    if (n > 0xFFFFFFFFull) return fail("%s: the packet count is a uint32 on the wire", who);
    const uint64_t h = mrpdr_header_bytes(f);
    // header i sits h bytes in front of packet i's data: offsets[] come from mrpdr_wave_sizeof with phase = h
    HdrEncodeArgs a = { stream_id, stream_ids, tag, tags, src_rank, src_ranks, nullptr, 0,
                        msg + mrpdr_message_preamble_bytes(n) - h, offsets, 0, n };
    k_msg_frame<<<blocks_for(n + 1, MRPDR_TPB), MRPDR_TPB, 0, (hipStream_t)stream>>>(msg, n, h, lens, a, img);
    LAUNCHED("k_msg_frame");
    return 1;
}

// ---------------------------------------------------------------------------
// filters
// ---------------------------------------------------------------------------
// Device memory a filter call may use between its launches: the tree nodes, and (ctx callers only)
// room for the chunk summaries of the device-wide float fold (flatwide.h).
struct FilterWs { Partial *partials; void *wide; uint64_t wide_bytes; };

static FilterWs ctx_filter_ws(const mrpdr_ctx *c)
{
    // (the string-array scratch doubles as the summaries' room: a filter and a decode of one ctx are
    // calls on one stream, one after the other)
    return FilterWs{ c->partials, c->pkt_bytes, (c->max_packets + 1) * sizeof(uint64_t) };
}

template <int OP, typename T>
static int32_t run_tree(const FilterWs &fw, const FilterIn &in, uint8_t *result, int32_t *result_ok, hipStream_t st)
{
    const uint32_t tiles = in.n ? blocks_for(in.n, FILTER_TILE) : 0;
    if (tiles == 1) {           // one tile: its workgroup takes the last level too
        k_filter_tree_tile<OP, T><<<1, MRPDR_TPB, 0, st>>>(in, fw.partials, result, result_ok);
        LAUNCHED("k_filter_tree_tile");
        return 1;
    }
    if (tiles) {
        k_filter_tree_tile<OP, T><<<tiles, MRPDR_TPB, 0, st>>>(in, fw.partials, nullptr, nullptr);
        LAUNCHED("k_filter_tree_tile");
    }
    k_filter_tree_top<OP, T><<<1, MRPDR_TPB, 0, st>>>(fw.partials, tiles, result, result_ok);
    LAUNCHED("k_filter_tree_top");
    return 1;
}

// The same over a range of a wave: the range's own tree, then its last level with the node in front as
// leftmost child, left as a node.
template <int OP, typename T>
static int32_t run_tree_link(const FilterWs &fw, const FilterIn &in, FilterLink link, hipStream_t st)
{
    const uint32_t tiles = in.n ? blocks_for(in.n, FILTER_TILE) : 0;
    if (tiles) {
        k_filter_tree_tile<OP, T><<<tiles, MRPDR_TPB, 0, st>>>(in, fw.partials, nullptr, nullptr);
        LAUNCHED("k_filter_tree_tile");
    }
    k_filter_tree_link<OP, T><<<1, MRPDR_TPB, 0, st>>>(fw.partials, tiles, link, nullptr, nullptr);
    LAUNCHED("k_filter_tree_link");
    return 1;
}

#define SHAPE_COMBINE 64      // (inside the library only: "values" are the nodes of consecutive ranges)

// One reference filter call over the whole wave (MRPDR_SHAPE_FLAT).  Only SUM and AVG of float
// and double depend on the order of the additions and take the serial kernel; every other result
// is computed level by level with the same bits: integer SUM / MIN / MAX / OR / AND as they are,
// integer AVG as sums with one division at the top, float MIN / MAX with the NaN rule carried
// through the levels.
// Below this many packets the scan's rounds do not pay (measured, growing sums: 4,096 floats 22 us
// against the plain chain's 17, 8,192: 31 against 33, 16,383: 44 against 64).
#define FLAT_SCAN_MIN 8192
// From here on three launches over the whole device beat the one workgroup, for both types (measured,
// floats, the one workgroup against the device-wide form against the plain chain: growing sums of
// 8,192 packets 31 / 26 / 35 us, of 16,383 packets 47 / 36 / 68; zero-mean draws 52 / 45 / 35 and
// 86 / 55-75 / 68 -- between these sizes neither exact form catches the chain on data whose sum hovers
// around zero, the two passes being a third of such a call; data that opens every chunk 37-49 / 42-47 / 35
// and 71-83 / 76-80 / 68).  The one workgroup is left with callers that have no room for the summaries.
#ifndef FLAT_WIDE_MIN
#define FLAT_WIDE_MIN 8192
#endif
#define FLAT_WIDE_MIN_DOUBLE 8192

template <int OP, typename T>
static int32_t run_flat(const FilterWs &fw, const FilterIn &in, uint8_t *result, int32_t *result_ok, hipStream_t st,
                        bool serial = false, FilterLink link = FilterLink{ nullptr, nullptr })
{
    constexpr bool fp = std::is_floating_point<T>::value;
    const bool ranged = link.front || link.node;      // a range of a wave (mrpdr_wave_filter_partial)
    if constexpr (fp && (OP == MRPDR_SUM || OP == MRPDR_AVG)) {
        if (!serial && in.n >= (sizeof(T) == 8 ? FLAT_WIDE_MIN_DOUBLE : FLAT_WIDE_MIN) && fw.wide) {
            // large waves: the per-packet work on the whole device, one workgroup walks the chunks
            const uint64_t nc = (in.n + WIDE_CHUNK - 1) / WIDE_CHUNK;
            if (WideWs<T>::bytes(nc) <= fw.wide_bytes) {
                k_wide_sums<OP, T><<<(uint32_t)nc, WIDE_TPB, 0, st>>>(in, fw.wide, nc);
                LAUNCHED("k_wide_sums");
                k_wide_maps<OP, T><<<(uint32_t)nc, WIDE_TPB, 0, st>>>(in, fw.wide, nc, link);
                LAUNCHED("k_wide_maps");
                k_wide_walk<OP, T><<<1, WIDE_TPB, 0, st>>>(in, fw.wide, nc, result, result_ok, link);
                LAUNCHED("k_wide_walk");
                return 1;
            }
        }
        if (!serial && in.n >= FLAT_SCAN_MIN) {
            // the same chain, taken a binade at a time by a scan (flatscan.h): same bits
            k_filter_flat_scan<OP, T><<<1, SCAN_TPB, 0, st>>>(in, result, result_ok, link);
            LAUNCHED("k_filter_flat_scan");
            return 1;
        }
        k_filter_flat<OP, T><<<1, MRPDR_TPB, 0, st>>>(in, result, result_ok, link);
        LAUNCHED("k_filter_flat");
        return 1;
    } else if constexpr (fp && OP == MRPDR_MIN) {
        return ranged ? run_tree_link<OP_MINR, T>(fw, in, link, st) : run_tree<OP_MINR, T>(fw, in, result, result_ok, st);
    } else if constexpr (fp && OP == MRPDR_MAX) {
        return ranged ? run_tree_link<OP_MAXR, T>(fw, in, link, st) : run_tree<OP_MAXR, T>(fw, in, result, result_ok, st);
    } else if constexpr (!fp && OP == MRPDR_AVG) {
        return ranged ? run_tree_link<OP_AVGSUM, T>(fw, in, link, st) : run_tree<OP_AVGSUM, T>(fw, in, result, result_ok, st);
    } else {
        return ranged ? run_tree_link<OP, T>(fw, in, link, st) : run_tree<OP, T>(fw, in, result, result_ok, st);
    }
}

// The nodes of consecutive ranges of one wave (mrpdr_wave_filter_partial), in order, into the result
// of the one reference call over the whole wave.
template <int OP, typename T>
static int32_t run_combine(const Partial *nodes, uint64_t g, uint8_t *result, int32_t *result_ok, hipStream_t st)
{
    constexpr bool fp = std::is_floating_point<T>::value;
    if constexpr (fp && (OP == MRPDR_SUM || OP == MRPDR_AVG)) {
        // the chain went through the ranges one after the other: the last node holds all of it
        // (a fold of no packets from that node on: the division of AVG and the result's form)
        FilterIn none = { nullptr, 0, nullptr, 0, 0, 0, nullptr };
        k_filter_flat<OP, T><<<1, MRPDR_TPB, 0, st>>>(none, result, result_ok, FilterLink{ g ? nodes + (g - 1) : nullptr, nullptr });
        LAUNCHED("k_filter_flat");
        return 1;
    } else {
        constexpr int TOP = (fp && OP == MRPDR_MIN) ? OP_MINR : (fp && OP == MRPDR_MAX) ? OP_MAXR : (!fp && OP == MRPDR_AVG) ? OP_AVGSUM : OP;
        k_filter_tree_link<TOP, T><<<1, MRPDR_TPB, 0, st>>>(nodes, g, FilterLink{ nullptr, nullptr }, result, result_ok);
        LAUNCHED("k_filter_tree_link");
        return 1;
    }
}

template <int OP, typename T>
static int32_t run_filter(const FilterWs &fw, int32_t shape, const FilterIn &in, uint8_t *result,
                          int32_t *result_ok, hipStream_t st, FilterLink link)
{
    if (shape == SHAPE_COMBINE) return run_combine<OP, T>((const Partial *)in.values, in.n, result, result_ok, st);
    if (shape == MRPDR_SHAPE_FLAT) return run_flat<OP, T>(fw, in, result, result_ok, st, false, link);
    if (shape == MRPDR_SHAPE_FLAT_SERIAL) return run_flat<OP, T>(fw, in, result, result_ok, st, true);
    return run_tree<OP, T>(fw, in, result, result_ok, st);
}

template <typename T>
static int32_t run_filter_op(const FilterWs &fw, int32_t op, int32_t shape, const FilterIn &in, uint8_t *result,
                             int32_t *result_ok, hipStream_t st, FilterLink link)
{
    switch (op) {
    case MRPDR_SUM: return run_filter<MRPDR_SUM, T>(fw, shape, in, result, result_ok, st, link);
    case MRPDR_MIN: return run_filter<MRPDR_MIN, T>(fw, shape, in, result, result_ok, st, link);
    case MRPDR_MAX: return run_filter<MRPDR_MAX, T>(fw, shape, in, result, result_ok, st, link);
    case MRPDR_AVG: return run_filter<MRPDR_AVG, T>(fw, shape, in, result, result_ok, st, link);
    case MRPDR_OR:
        if constexpr (!std::is_floating_point<T>::value) return run_filter<MRPDR_OR, T>(fw, shape, in, result, result_ok, st, link);
        break;
    case MRPDR_AND:
        if constexpr (!std::is_floating_point<T>::value) return run_filter<MRPDR_AND, T>(fw, shape, in, result, result_ok, st, link);
        break;
    }
    return fail("mrpdr_wave_filter: unknown operator %d", op);
}

// Argument checks and type dispatch shared by the ctx entry points and the one that allocates nothing.
static int32_t filter_dispatch(const FilterWs &fw, int32_t op, int32_t type, int32_t shape,
                               const void *values, uint64_t value_stride,
                               const void *counts, uint64_t count_stride,
                               uint64_t n, int32_t swap, const uint8_t *orders,
                               void *result, int32_t *result_ok, void *stream,
                               FilterLink link = FilterLink{ nullptr, nullptr })
{
    if (!result && !link.node) return fail("mrpdr_wave_filter: NULL result");
    if (n && !values) return fail("mrpdr_wave_filter: NULL values");
    if (op == MRPDR_AVG && n && !counts && shape != SHAPE_COMBINE) return fail("mrpdr_wave_filter: AVG needs counts");
    if (shape != MRPDR_SHAPE_FLAT && shape != MRPDR_SHAPE_TREE && shape != MRPDR_SHAPE_FLAT_SERIAL && shape != SHAPE_COMBINE)
        return fail("mrpdr_wave_filter: unknown shape %d", shape);
    if ((op == MRPDR_OR || op == MRPDR_AND) && (type == MRPDR_FLOAT_T || type == MRPDR_DOUBLE_T))
        return fail("mrpdr_wave_filter: bitwise operators need an integer type");
    FilterIn in = { (const uint8_t *)values, value_stride, (const uint8_t *)counts, count_stride, n, swap, orders };
    uint8_t *res = (uint8_t *)result;
    hipStream_t st = (hipStream_t)stream;
    switch (type) {
    case MRPDR_CHAR_T:   return run_filter_op<int8_t>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_UCHAR_T:  return run_filter_op<uint8_t>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_INT16_T:  return run_filter_op<int16_t>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_UINT16_T: return run_filter_op<uint16_t>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_INT32_T:  return run_filter_op<int32_t>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_UINT32_T: return run_filter_op<uint32_t>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_INT64_T:  return run_filter_op<int64_t>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_UINT64_T: return run_filter_op<uint64_t>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_FLOAT_T:  return run_filter_op<float>(fw, op, shape, in, res, result_ok, st, link);
    case MRPDR_DOUBLE_T: return run_filter_op<double>(fw, op, shape, in, res, result_ok, st, link);
    }
    // the built-in filters accept exactly these ten types (src/FilterDefinitions.C:110-146)
    return fail("mrpdr_wave_filter: type %d is not a scalar the filters accept", type);
}

static int32_t wave_filter(mrpdr_ctx *c, int32_t op, int32_t type, int32_t shape,
                           const void *values, uint64_t value_stride,
                           const void *counts, uint64_t count_stride,
                           uint64_t n, int32_t swap, const uint8_t *orders,
                           void *result, int32_t *result_ok, void *stream)
{
    if (!ctx_ready("mrpdr_wave_filter", c)) return 0;
    if (n > c->max_packets) return fail("mrpdr_wave_filter: n exceeds the ctx limit");
    return filter_dispatch(ctx_filter_ws(c), op, type, shape, values, value_stride, counts, count_stride, n, swap, orders,
                           result, result_ok, stream);
}

extern "C" int32_t mrpdr_wave_filter(mrpdr_ctx *c, int32_t op, int32_t type, int32_t shape,
                                     const void *values, uint64_t value_stride,
                                     const void *counts, uint64_t count_stride,
                                     uint64_t n, int32_t swap,
                                     void *result, int32_t *result_ok, void *stream)
{
    return wave_filter(c, op, type, shape, values, value_stride, counts, count_stride, n, swap, nullptr,
                       result, result_ok, stream);
}

extern "C" int32_t mrpdr_wave_filter_mixed(mrpdr_ctx *c, int32_t op, int32_t type, int32_t shape,
                                           const void *values, uint64_t value_stride,
                                           const void *counts, uint64_t count_stride,
                                           uint64_t n, const uint8_t *byteorders,
                                           void *result, int32_t *result_ok, void *stream)
{
    if (n && !byteorders) return fail("mrpdr_wave_filter_mixed: NULL byteorders");
    return wave_filter(c, op, type, shape, values, value_stride, counts, count_stride, n, 0, n ? byteorders : nullptr,
                       result, result_ok, stream);
}

// ---------------------------------------------------------------------------
// a wave in ranges: one reference filter call over packets that several GPUs (or several calls) hold
// ---------------------------------------------------------------------------
static_assert(sizeof(mrpdr_node) == sizeof(Partial) && offsetof(mrpdr_node, count) == offsetof(Partial, cnt) &&
              offsetof(mrpdr_node, present) == offsetof(Partial, present), "mrpdr_node is the tree node the kernels exchange");

extern "C" int32_t mrpdr_filter_is_chained(int32_t op, int32_t type)
{
    return (op == MRPDR_SUM || op == MRPDR_AVG) && (type == MRPDR_FLOAT_T || type == MRPDR_DOUBLE_T);
}

extern "C" int32_t mrpdr_wave_filter_partial(mrpdr_ctx *c, int32_t op, int32_t type,
                                             const void *values, uint64_t value_stride,
                                             const void *counts, uint64_t count_stride,
                                             uint64_t n, int32_t swap,
                                             const mrpdr_node *front, mrpdr_node *node, void *stream)
{
    if (!ctx_ready("mrpdr_wave_filter_partial", c)) return 0;
    if (n > c->max_packets) return fail("mrpdr_wave_filter_partial: n exceeds the ctx limit");
    if (!node) return fail("mrpdr_wave_filter_partial: NULL node");
    return filter_dispatch(ctx_filter_ws(c), op, type, MRPDR_SHAPE_FLAT, values, value_stride, counts, count_stride, n, swap,
                           nullptr, nullptr, nullptr, stream, FilterLink{ (const Partial *)front, (Partial *)node });
}

extern "C" int32_t mrpdr_wave_filter_partial_mixed(mrpdr_ctx *c, int32_t op, int32_t type,
                                                   const void *values, uint64_t value_stride,
                                                   const void *counts, uint64_t count_stride,
                                                   uint64_t n, const uint8_t *byteorders,
                                                   const mrpdr_node *front, mrpdr_node *node, void *stream)
{
    if (!ctx_ready("mrpdr_wave_filter_partial_mixed", c)) return 0;
    if (n > c->max_packets) return fail("mrpdr_wave_filter_partial_mixed: n exceeds the ctx limit");
    if (!node) return fail("mrpdr_wave_filter_partial_mixed: NULL node");
    if (n && !byteorders) return fail("mrpdr_wave_filter_partial_mixed: NULL byteorders");
    return filter_dispatch(ctx_filter_ws(c), op, type, MRPDR_SHAPE_FLAT, values, value_stride, counts, count_stride, n, 0,
                           n ? byteorders : nullptr, nullptr, nullptr, stream, FilterLink{ (const Partial *)front, (Partial *)node });
}

extern "C" int32_t mrpdr_wave_filter_combine(mrpdr_ctx *c, int32_t op, int32_t type, const mrpdr_node *nodes, uint64_t g,
                                             void *result, int32_t *result_ok, void *stream)
{
    if (!ctx_ready("mrpdr_wave_filter_combine", c)) return 0;
    if (g && !nodes) return fail("mrpdr_wave_filter_combine: NULL nodes");
    if (!result) return fail("mrpdr_wave_filter_combine: NULL result");
    return filter_dispatch(ctx_filter_ws(c), op, type, SHAPE_COMBINE, nodes, sizeof(mrpdr_node), nullptr, 0, g, 0, nullptr,
                           result, result_ok, stream);
}

// ---------------------------------------------------------------------------
// decode + filter in one pass over the rows (packed scalar-only waves)
// ---------------------------------------------------------------------------
// The operator the tree kernels run for (op, shape) on T, or -1 when the result needs the serial
// kernel (one reference call of SUM / AVG over float or double).
static int tree_op_for(int32_t op, int32_t shape, bool fp)
{
    if (shape == MRPDR_SHAPE_TREE) return op;
    if (fp) return (op == MRPDR_SUM || op == MRPDR_AVG) ? -1 : op == MRPDR_MIN ? OP_MINR : op == MRPDR_MAX ? OP_MAXR : op;
    return op == MRPDR_AVG ? OP_AVGSUM : op;
}

template <int OP, typename T>
static int32_t launch_fused(mrpdr_ctx *c, const mrpdr_fmt *f, const WireDev &wd, const ColsDev &cd, uint64_t n,
                            int32_t swap, int32_t *ok, uint8_t *result, int32_t *result_ok, hipStream_t st)
{
    const uint32_t tiles = blocks_for(n, FILTER_TILE);
    if (tiles == 1) {           // one tile: its workgroup takes the last level too
        k_rows_decode_filter<OP, T><<<1, MRPDR_TPB, FILTER_TILE * wd.stride, st>>>(f->dev, wd, cd, n, swap, ok, c->partials,
                                                                                    result, result_ok);
        LAUNCHED("k_rows_decode_filter");
        return 1;
    }
    k_rows_decode_filter<OP, T><<<tiles, MRPDR_TPB, FILTER_TILE * wd.stride, st>>>(f->dev, wd, cd, n, swap, ok, c->partials,
                                                                                    nullptr, nullptr);
    LAUNCHED("k_rows_decode_filter");
    k_filter_tree_top<OP, T><<<1, MRPDR_TPB, 0, st>>>(c->partials, tiles, result, result_ok);
    LAUNCHED("k_filter_tree_top");
    return 1;
}

template <typename T>
static int32_t launch_fused_op(int top, mrpdr_ctx *c, const mrpdr_fmt *f, const WireDev &wd, const ColsDev &cd, uint64_t n,
                               int32_t swap, int32_t *ok, uint8_t *result, int32_t *result_ok, hipStream_t st)
{
    constexpr bool fp = std::is_floating_point<T>::value;
    switch (top) {
    case MRPDR_SUM: return launch_fused<MRPDR_SUM, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st);
    case MRPDR_MIN: return launch_fused<MRPDR_MIN, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st);
    case MRPDR_MAX: return launch_fused<MRPDR_MAX, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st);
    case MRPDR_AVG: return launch_fused<MRPDR_AVG, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st);
    case MRPDR_OR:  if constexpr (!fp) return launch_fused<MRPDR_OR, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st); break;
    case MRPDR_AND: if constexpr (!fp) return launch_fused<MRPDR_AND, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st); break;
    case OP_AVGSUM: if constexpr (!fp) return launch_fused<OP_AVGSUM, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st); break;
    case OP_MINR:   if constexpr (fp) return launch_fused<OP_MINR, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st); break;
    case OP_MAXR:   if constexpr (fp) return launch_fused<OP_MAXR, T>(c, f, wd, cd, n, swap, ok, result, result_ok, st); break;
    }
    return fail("mrpdr_wave_decode_filter: unknown operator");
}

extern "C" int32_t mrpdr_wave_decode_filter(mrpdr_ctx *c, const mrpdr_fmt *f, const mrpdr_wire *wire, uint64_t n,
                                            int32_t sender_bo, mrpdr_column *cols, int32_t *ok,
                                            int32_t op, int32_t shape, void *result, int32_t *result_ok, void *stream)
{
    const char *who = "mrpdr_wave_decode_filter";
    if (!check_common(who, c, f, n)) return 0;
    if (sender_bo != MRPDR_LITTLEENDIAN && sender_bo != MRPDR_BIGENDIAN) return fail("%s: bad byte order %d", who, sender_bo);
    if (!f->fixed || f->dev.nf < 1) return fail("%s: needs a format of scalars only, the filtered value first", who);
    if (op < MRPDR_SUM || op > MRPDR_AND) return fail("%s: unknown operator %d", who, op);
    if (shape != MRPDR_SHAPE_FLAT && shape != MRPDR_SHAPE_TREE && shape != MRPDR_SHAPE_FLAT_SERIAL)
        return fail("%s: unknown shape %d", who, shape);
    if (op == MRPDR_AVG && (f->dev.nf < 2 || f->dev.f[1].type != MRPDR_INT32_T))
        return fail("%s: AVG needs the int32 count as second field", who);
    if (!result) return fail("%s: NULL result", who);
    const int32_t type = f->dev.f[0].type;
    const bool fp = type == MRPDR_FLOAT_T || type == MRPDR_DOUBLE_T;
    if ((op == MRPDR_OR || op == MRPDR_AND) && fp) return fail("%s: bitwise operators need an integer type", who);
    ColsDev cd;
    WireDev wd;
    if (!fill_cols(who, f, cols, n, false, &cd, false)) return 0;
    if (!fill_wire(who, f, wire, n, &wd)) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int top = tree_op_for(op, shape, fp);
    const int32_t swap = sender_bo != MRPDR_LITTLEENDIAN;
    if (n == 0 || top < 0 || !rows_path_ok(f, wd) || wd.stride > FUSED_MAX_STRIDE) {
        // what the one pass cannot do: decode, then the filter over the decoded columns
        if (!decode_begin(who, c, f, wire, n, sender_bo, nullptr, cols, ok, nullptr, stream)) return 0;
        return filter_dispatch(ctx_filter_ws(c), op, type, shape, n ? cols[0].values : nullptr, f->dev.f[0].width,
                               (n && op == MRPDR_AVG) ? cols[1].values : nullptr, 4, n, 0, nullptr, result, result_ok, stream);
    }
    uint8_t *res = (uint8_t *)result;
    switch (type) {
    case MRPDR_CHAR_T:   return launch_fused_op<int8_t>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_UCHAR_T:  return launch_fused_op<uint8_t>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_INT16_T:  return launch_fused_op<int16_t>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_UINT16_T: return launch_fused_op<uint16_t>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_INT32_T:  return launch_fused_op<int32_t>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_UINT32_T: return launch_fused_op<uint32_t>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_INT64_T:  return launch_fused_op<int64_t>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_UINT64_T: return launch_fused_op<uint64_t>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_FLOAT_T:  return launch_fused_op<float>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    case MRPDR_DOUBLE_T: return launch_fused_op<double>(top, c, f, wd, cd, n, swap, ok, res, result_ok, st);
    }
    return fail("%s: type %d is not a scalar the filters accept", who, type);
}

// ---------------------------------------------------------------------------
// mrpdr_wave_filter without a ctx and without any allocation
// ---------------------------------------------------------------------------
// The tree nodes between the two launches live in a file-scope device array instead of a ctx
// workspace.  Every GPU the library runs on holds its own copy of the array, at its own address,
// until the process ends; the name below is only the host-side handle of it, the address the
// kernels get comes from hipGetSymbolAddress, once per device.
#define NOALLOC_MAX_PACKETS (1ull << 24)
#define NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[(NOALLOC_MAX_PACKETS / FILTER_TILE + 1) * sizeof(Partial)];

static void *g_scratch_ptr[NOALLOC_MAX_DEVICES];       // per device: where its copy of scratch lives
static std::mutex g_scratch_lock;

// The current device's copy of scratch (resolved on the first call on that device).
static int32_t scratch_of_current_device(const char *who, void **out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail("%s: no current HIP device", who);
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return fail("%s: device %d is beyond the %d this library keeps scratch for", who, dev, NOALLOC_MAX_DEVICES);
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail("%s: hipGetSymbolAddress(scratch): %s", who, hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return 1;
}

extern "C" int32_t mrpdr_wave_filter_noalloc(int32_t op, int32_t type, int32_t shape,
                                             const void *values, uint64_t value_stride,
                                             const void *counts, uint64_t count_stride,
                                             uint64_t n, int32_t swap,
                                             void *result, int32_t *result_ok, void *stream)
{
    if (n > NOALLOC_MAX_PACKETS)
        return fail("mrpdr_wave_filter_noalloc: n=%llu exceeds the %llu packets the built-in scratch holds nodes for",
                    (unsigned long long)n, (unsigned long long)NOALLOC_MAX_PACKETS);
    void *ptr = nullptr;
    if (!scratch_of_current_device("mrpdr_wave_filter_noalloc", &ptr)) return 0;
    // nothing in the array has to start from zero: the first launch writes every node the second reads
    // (a call uses the array either for tree nodes or for the chunk summaries of a large float fold,
    // never for both; waves whose summaries do not fit take the one-workgroup scan)
    return filter_dispatch(FilterWs{ (Partial *)ptr, ptr, sizeof scratch }, op, type, shape, values, value_stride, counts, count_stride, n, swap, nullptr,
                           result, result_ok, stream);
}

template <typename T>
static int32_t run_stats(mrpdr_ctx *c, const FilterIn &in, uint8_t *results, int32_t *oks, hipStream_t st)
{
    const uint32_t tiles = in.n ? blocks_for(in.n, FILTER_TILE) : 0;
    if (tiles) {
        k_filter_stats_tile<T><<<tiles, MRPDR_TPB, 0, st>>>(in, c->partials, c->npartials);
        LAUNCHED("k_filter_stats_tile");
    }
    k_filter_stats_top<T><<<1, MRPDR_TPB, 0, st>>>(c->partials, tiles, c->npartials, results, oks);
    LAUNCHED("k_filter_stats_top");
    return 1;
}

extern "C" int32_t mrpdr_wave_filter_stats(mrpdr_ctx *c, int32_t type, const void *values, uint64_t value_stride,
                                           const void *counts, uint64_t count_stride, uint64_t n, int32_t swap,
                                           void *results, int32_t *results_ok, void *stream)
{
    if (!ctx_ready("mrpdr_wave_filter_stats", c)) return 0;
    if (n > c->max_packets) return fail("mrpdr_wave_filter_stats: n exceeds the ctx limit");
    if (!results || !results_ok) return fail("mrpdr_wave_filter_stats: NULL result");
    if (n && (!values || !counts)) return fail("mrpdr_wave_filter_stats: NULL values or counts");
    FilterIn in = { (const uint8_t *)values, value_stride, (const uint8_t *)counts, count_stride, n, swap, nullptr };
    uint8_t *res = (uint8_t *)results;
    hipStream_t st = (hipStream_t)stream;
    switch (type) {
    case MRPDR_CHAR_T:   return run_stats<int8_t>(c, in, res, results_ok, st);
    case MRPDR_UCHAR_T:  return run_stats<uint8_t>(c, in, res, results_ok, st);
    case MRPDR_INT16_T:  return run_stats<int16_t>(c, in, res, results_ok, st);
    case MRPDR_UINT16_T: return run_stats<uint16_t>(c, in, res, results_ok, st);
    case MRPDR_INT32_T:  return run_stats<int32_t>(c, in, res, results_ok, st);
    case MRPDR_UINT32_T: return run_stats<uint32_t>(c, in, res, results_ok, st);
    case MRPDR_INT64_T:  return run_stats<int64_t>(c, in, res, results_ok, st);
    case MRPDR_UINT64_T: return run_stats<uint64_t>(c, in, res, results_ok, st);
    case MRPDR_FLOAT_T:  return run_stats<float>(c, in, res, results_ok, st);
    case MRPDR_DOUBLE_T: return run_stats<double>(c, in, res, results_ok, st);
    }
    return fail("mrpdr_wave_filter_stats: type %d is not a scalar the filters accept", type);
}

// ---------------------------------------------------------------------------
// array filters
// ---------------------------------------------------------------------------
__global__ void k_concat_column(const uint64_t *in_offs, uint64_t n, uint64_t *out_counts, uint64_t *out_offs,
                                uint32_t *status)
{
    const uint64_t total = in_offs[n];
    out_counts[0] = total;
    out_offs[0] = 0;
    out_offs[1] = total;
    if (total > 0xFFFFFFFFull) atomicOr(status, 2u);
}

extern "C" int32_t mrpdr_wave_array_concat(mrpdr_ctx *c, const mrpdr_column *in, uint64_t n,
                                           mrpdr_column *out, void *stream)
{
    if (!c || !in || !out) return fail("mrpdr_wave_array_concat: NULL argument");
    if (!ctx_ready("mrpdr_wave_array_concat", c)) return 0;
    if (n > c->max_packets) return fail("mrpdr_wave_array_concat: n exceeds the ctx limit");
    if (!in->offs || !out->counts || !out->offs) return fail("mrpdr_wave_array_concat: counts / offs missing");
    k_concat_column<<<1, 1, 0, (hipStream_t)stream>>>(in->offs, n, out->counts, out->offs, c->status);
    LAUNCHED("k_concat_column");
    return 1;
}

// tfilter_PerfData with TFILTER_ARRAY_CONCAT (src/FilterDefinitions.C:1000-1008, 1171-1203): the
// arrays of the packets that carry data, back to back; packets without data contribute nothing,
// not even their rank and nelems arrays.
__global__ __launch_bounds__(MRPDR_TPB) void k_masked_counts(const uint64_t *counts, const uint64_t *mask_counts,
                                                             uint64_t n, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    if (i < n) out[i] = mask_counts[i] ? counts[i] : 0;
}

__global__ __launch_bounds__(MRPDR_TPB) void k_masked_gather(const uint8_t *values, const uint64_t *offs,
                                                             const uint64_t *kept, const uint64_t *kept_offs,
                                                             uint64_t n, uint32_t width, uint8_t *out, uint64_t out_cap,
                                                             uint64_t *out_counts, uint64_t *out_offs, uint32_t *status)
{
    const uint64_t i = (uint64_t)blockIdx.x * MRPDR_TPB + threadIdx.x;
    const uint64_t total = kept_offs[n];
    const bool fits = total <= out_cap;
    if (i == 0) {
        if (!fits) atomicOr(status, 1u);
        out_counts[0] = fits ? total : 0;
        out_offs[0] = 0;
        out_offs[1] = fits ? total : 0;
    }
    if (i >= n || !fits) return;
    const uint8_t *src = values + offs[i] * width;
    uint8_t *dst = out + kept_offs[i] * width;
    for (uint64_t b = 0; b < kept[i] * width; b++) dst[b] = src[b];
}

extern "C" int32_t mrpdr_wave_array_concat_masked(mrpdr_ctx *c, const mrpdr_column *in, uint32_t width,
                                                  const uint64_t *mask_counts, uint64_t n,
                                                  mrpdr_column *out, void *stream)
{
    if (!c || !in || !out || !mask_counts) return fail("mrpdr_wave_array_concat_masked: NULL argument");
    if (!ctx_ready("mrpdr_wave_array_concat_masked", c)) return 0;
    if (n == 0 || n > c->max_packets) return fail("mrpdr_wave_array_concat_masked: bad n");
    if (!in->counts || !in->offs || !out->counts || !out->offs || (out->capacity && !out->values))
        return fail("mrpdr_wave_array_concat_masked: columns need counts, offs and values");
    if (width != 1 && width != 2 && width != 4 && width != 8) return fail("mrpdr_wave_array_concat_masked: bad width");
    hipStream_t st = (hipStream_t)stream;
    k_masked_counts<<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, st>>>(in->counts, mask_counts, n, c->pkt_bytes);
    LAUNCHED("k_masked_counts");
    if (!run_scan(c, c->pkt_bytes, n, 1, 0, c->pkt_boffs, st)) return 0;
    k_masked_gather<<<blocks_for(n, MRPDR_TPB), MRPDR_TPB, 0, st>>>((const uint8_t *)in->values, in->offs, c->pkt_bytes,
                                                                    c->pkt_boffs, n, width, (uint8_t *)out->values,
                                                                    out->capacity, out->counts, out->offs, c->status);
    LAUNCHED("k_masked_gather");
    return 1;
}

static inline uint64_t up16(uint64_t v) { return (v + 15) & ~(uint64_t)15; }

// ---- tfilter_PerfData: levels of nodes ---------------------------------------------------------------
#define PERF_MAX_LEVELS 8
struct PerfPlan { int levels; uint64_t group[PERF_MAX_LEVELS + 1], nodes[PERF_MAX_LEVELS + 1]; };

// Level l folds groups of fanins[l] consecutive packets of the level below (0: all of them); a level of
// one node ends the tree, and if the fan-ins given leave several nodes one more level folds them all.
static PerfPlan perf_plan(uint64_t n, const uint64_t *fanins, int nlevels)
{
    PerfPlan p;
    p.levels = 0;
    uint64_t m = n ? n : 1;
    for (int l = 0; l < nlevels && l < PERF_MAX_LEVELS && (m > 1 || l == 0); l++) {
        const uint64_t f = (fanins[l] == 0 || fanins[l] >= m) ? m : fanins[l];
        p.group[p.levels] = f;
        p.nodes[p.levels] = (m + f - 1) / f;
        m = p.nodes[p.levels++];
    }
    if (m > 1 || p.levels == 0) { p.group[p.levels] = m; p.nodes[p.levels] = 1; p.levels++; }
    return p;
}

// what the nodes of one level take in the scratch: data, lens, offs, rank counts, rank offs, lasts, ranks
static uint64_t perf_level_bytes(uint64_t nodes, uint64_t out_cap)
{
    // (the data part ends in room for one workgroup's lanes: the level above reads a node's W elements
    //  whatever its length, MRPDR_TPB of them at most)
    return up16(nodes * out_cap * 8 + MRPDR_TPB * 8) + 5 * up16(nodes * 8) + up16(nodes * 4);
}

static uint64_t perf_plan_bytes(const PerfPlan &p, uint64_t out_cap)
{
    uint64_t b = 0;
    for (int l = 0; l + 1 < p.levels; l++) b += perf_level_bytes(p.nodes[l], out_cap);
    return b;
}

// The order-free form of one invocation over integer data: raw nodes of 16 packets, of 32 nodes, of 32, then all
// (two launches for short arrays: k_perf_nodes folds two levels at a time).
static const uint64_t PERF_FREE_FANINS[4] = { 16, 32, 32, 0 };
#define PERF_FREE_LEVELS 4
#define PERF_FREE_MIN 2048          // packets from which the raw tree is taken instead of the serial fold

extern "C" uint64_t mrpdr_perfdata_levels_scratch_bytes(uint64_t n, const uint64_t *fanins, int32_t nlevels, uint64_t out_cap)
{
    if (!fanins || nlevels <= 0) return 0;
    return perf_plan_bytes(perf_plan(n, fanins, nlevels), out_cap);
}

// One invocation over doubles: the elements as strided columns through the device-wide exact fold, in place
// when the arrays all have one length, from a dense copy otherwise (perfnodes.h, k_perf_shape).
#define PERF_COLUMNS_MIN FLAT_WIDE_MIN_DOUBLE       // packets from which the strided fold is tried
#define PERF_COLUMNS_MAX_CAP 64u                    // elements per packet at most (one walking workgroup each)
static uint64_t perf_columns_ws_stride(uint64_t n)
{
    return up16(WideWs<double>::bytes((n + WIDE_CHUNK - 1) / WIDE_CHUNK));
}
struct PerfColumns {            // the scratch of that path, carved up
    uint64_t *shape; int32_t *nranks; Partial *fronts, *nodes; void *ws; uint64_t *dense; uint64_t bytes;
};
static PerfColumns perf_columns(void *scratch, uint64_t n, uint64_t cap)
{
    uint64_t at = 0;
    const uint64_t o_shape = at;    at += PERF_SHAPE_WORDS * 8;
    const uint64_t o_nranks = at;   at += up16(4 * n);
    const uint64_t o_fronts = at;   at += up16(16 * cap);
    const uint64_t o_nodes = at;    at += up16(16 * cap);
    const uint64_t o_ws = at;       at += cap * perf_columns_ws_stride(n);
    const uint64_t o_dense = at;    at += up16(8 * n * cap);
    PerfColumns c = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, at };
    if (scratch) {
        uint8_t *p = (uint8_t *)scratch;
        c.shape = (uint64_t *)(p + o_shape);
        c.nranks = (int32_t *)(p + o_nranks);
        c.fronts = (Partial *)(p + o_fronts);
        c.nodes = (Partial *)(p + o_nodes);
        c.ws = p + o_ws;
        c.dense = (uint64_t *)(p + o_dense);
    }
    return c;
}
static uint64_t perf_columns_bytes(uint64_t n, uint64_t out_cap) { return perf_columns(nullptr, n, out_cap).bytes; }

// The column passes of one PerfData operator over the packets `in` (doubles): shape and weights, the dense
// copy if the device finds the wave ragged, then the elements' chains (SUM / AVG) or trees (MIN / MAX) side
// by side, each from its front node (a state handed over, or the operator's start value) to its node.
static int32_t perf_columns_run(int32_t op, const PerfIn &in, uint64_t cap, const void *front, const PerfColumns &pc, hipStream_t st)
{
    const uint64_t n = in.n, nc = (n + WIDE_CHUNK - 1) / WIDE_CHUNK;
    HIP_OK(hipMemsetAsync(pc.shape, 0, PERF_SHAPE_WORDS * 8, st));
    const uint32_t shape_blocks = blocks_for(n, MRPDR_TPB);
    k_perf_shape<<<shape_blocks < PERF_SHAPE_BLOCKS ? shape_blocks : PERF_SHAPE_BLOCKS, MRPDR_TPB, 0, st>>>(in, cap, pc.shape, pc.nranks);
    LAUNCHED("k_perf_shape");
    const bool sums = op == MRPDR_SUM || op == MRPDR_AVG;
    k_perf_shape_verdict<<<1, 64, 0, st>>>(pc.shape, (const uint8_t *)front, sums ? pc.fronts : nullptr, cap);
    LAUNCHED("k_perf_shape_verdict");
    const uint32_t dense_blocks = blocks_for(n * cap, MRPDR_TPB);
    k_perf_densify<<<dense_blocks < 2048u ? dense_blocks : 2048u, MRPDR_TPB, 0, st>>>(in, pc.shape, cap, pc.dense);
    LAUNCHED("k_perf_densify");
    FilterIn fin = { (const uint8_t *)in.data_vals, 0, (const uint8_t *)pc.nranks, 4, n, 0, nullptr };
    WideMulti m = { pc.shape, perf_columns_ws_stride(n), pc.fronts, pc.nodes, (const uint8_t *)pc.dense, in.data_counts };
    const dim3 wide((uint32_t)nc, (uint32_t)cap), one(1, (uint32_t)cap);
    if (op == MRPDR_SUM) {
        k_wide_sums_multi<MRPDR_SUM, double><<<wide, WIDE_TPB, 0, st>>>(fin, pc.ws, nc, m);
        k_wide_maps_multi<MRPDR_SUM, double><<<wide, WIDE_TPB, 0, st>>>(fin, pc.ws, nc, m);
        k_wide_walk_multi<MRPDR_SUM, double><<<one, WIDE_TPB, 0, st>>>(fin, pc.ws, nc, nullptr, nullptr, m);
    } else if (op == MRPDR_AVG) {
        k_wide_sums_multi<MRPDR_AVG, double><<<wide, WIDE_TPB, 0, st>>>(fin, pc.ws, nc, m);
        k_wide_maps_multi<MRPDR_AVG, double><<<wide, WIDE_TPB, 0, st>>>(fin, pc.ws, nc, m);
        k_wide_walk_multi<MRPDR_AVG, double><<<one, WIDE_TPB, 0, st>>>(fin, pc.ws, nc, nullptr, nullptr, m);
    } else {
        // MIN / MAX: the elements' trees side by side (the summaries' room holds their nodes)
        const uint32_t tiles = blocks_for(n, FILTER_TILE);
        const dim3 tg(tiles, (uint32_t)cap);
        if (op == MRPDR_MIN) {
            k_perf_minmax_fronts<MRPDR_MIN><<<1, 64, 0, st>>>(pc.shape, (const uint8_t *)front, pc.fronts);
            k_tree_tile_multi<OP_MINR, double><<<tg, MRPDR_TPB, 0, st>>>(fin, (Partial *)pc.ws, tiles, m);
            k_tree_top_multi<OP_MINR, double><<<one, MRPDR_TPB, 0, st>>>((const Partial *)pc.ws, tiles, tiles, nullptr, nullptr, m);
        } else {
            k_perf_minmax_fronts<MRPDR_MAX><<<1, 64, 0, st>>>(pc.shape, (const uint8_t *)front, pc.fronts);
            k_tree_tile_multi<OP_MAXR, double><<<tg, MRPDR_TPB, 0, st>>>(fin, (Partial *)pc.ws, tiles, m);
            k_tree_top_multi<OP_MAXR, double><<<one, MRPDR_TPB, 0, st>>>((const Partial *)pc.ws, tiles, tiles, nullptr, nullptr, m);
        }
    }
    LAUNCHED("PerfData column passes");
    return 1;
}

extern "C" uint64_t mrpdr_perfdata_scratch_bytes(uint64_t n, uint64_t group, uint64_t out_cap)
{
    if (group == 0 || group >= n) {         // one invocation: integer data may take the raw tree, regular doubles the strided fold
        const uint64_t a = n >= PERF_FREE_MIN ? perf_plan_bytes(perf_plan(n, PERF_FREE_FANINS, PERF_FREE_LEVELS), out_cap) : 0;
        const uint64_t b = (n >= PERF_COLUMNS_MIN && out_cap && out_cap <= PERF_COLUMNS_MAX_CAP) ? perf_columns_bytes(n, out_cap) : 0;
        return a > b ? a : b;
    }
    const uint64_t two[2] = { group, 0 };
    return perf_plan_bytes(perf_plan(n, two, 2), out_cap);
}

extern "C" uint64_t mrpdr_perfdata_state_bytes(uint64_t cap) { return sizeof(PerfState) + 8 * cap; }

// every PerfDataMgr metric is folded in order where the order can change a bit: doubles round at every
// addition, and their MIN / MAX keep the later operand whenever a comparison with a NaN is false
extern "C" int32_t mrpdr_perfdata_is_chained(int32_t op, int32_t type)
{
    (void)op;
    return type == MRPDR_DOUBLE_T;
}

static uint32_t perf_lanes(uint64_t out_cap)
{
    uint32_t w = PERF_MIN_W;
    while (w < out_cap && w < MRPDR_TPB) w <<= 1;
    return w;
}

// One level -- or two: several nodes of short arrays go to k_perf_nodes (many nodes per workgroup), which
// folds the level above in the same launch when a workgroup's nodes are exactly one node's children;
// one node, long nodes or long arrays go to k_perfdata (a node per workgroup row, its packets staged
// 256 at a time).  Returns the number of levels done (0: failure).
static bool perf_nodes_kernel(uint64_t nodes, uint64_t group, uint64_t out_cap)
{
    // (long nodes are better off with a workgroup each: k_perf_nodes takes a node's packets 32 at a
    //  time, two to three dependent round trips per round; measured, nodes of 256: 69 against 39 us per wave)
    return nodes > 1 && perf_lanes(out_cap) <= 64 && group <= 4 * PERF_CHUNK;
}

template <int OP, typename T>
static int32_t launch_perf_level(const PerfIn &in, uint64_t nodes, const PerfOut &o, uint64_t out_cap, const PerfRaw &raw,
                                 uint64_t regular, bool fuse, bool raw_mid, hipStream_t st)
{
    const uint32_t w = perf_lanes(out_cap);
    if (perf_nodes_kernel(nodes, in.group, out_cap)) {
        PerfLevel a = { in, o, raw.in_lasts, raw.out_lasts, raw.out_packets, w, (uint32_t)nodes, regular, fuse ? 1u : 0u, raw_mid ? 1u : 0u };
        const uint32_t npb = MRPDR_TPB / w;
        k_perf_nodes<OP, T><<<dim3(1, (uint32_t)((nodes + npb - 1) / npb)), MRPDR_TPB, 0, st>>>(a);
        LAUNCHED("k_perf_nodes");
        return 1;
    }
    const uint32_t xb = blocks_for(out_cap ? out_cap : 1, MRPDR_TPB);
    k_perfdata<OP, T><<<dim3(xb, (uint32_t)nodes), MRPDR_TPB, 0, st>>>(in, o, raw);
    LAUNCHED("k_perfdata");
    return 1;
}

// The levels of a plan, from the wave's columns to `final` (the one node of the last level).  raw: the
// nodes in between are raw (nothing divided or re-weighted on the way up); raw_final / front: the last
// node is left raw as well (a state: final_raw.out_lasts / out_packets say where), and, for a plan of
// one level, continues from the state `front`.
template <int OP, typename T>
static int32_t run_perf_levels(mrpdr_ctx *c, PerfIn in, const PerfPlan &plan, bool raw, uint64_t out_cap, const PerfOut &final,
                               const PerfRaw &final_raw, uint8_t *scratch, hipStream_t st)
{
    const uint64_t *in_lasts = nullptr;
    uint64_t regular = 0;
    uint8_t *p = scratch;
    const uint32_t npb = MRPDR_TPB / perf_lanes(out_cap);
    for (int l = 0; l < plan.levels; l++) {
        uint64_t nodes = plan.nodes[l];
        in.group = plan.group[l];
        // two levels in one launch: the workgroups of level l hold npb nodes each, and those are the
        // children of one node of level l + 1
        const bool fuse = l + 1 < plan.levels && perf_nodes_kernel(nodes, in.group, out_cap) &&
                          (plan.group[l + 1] == npb || (nodes <= npb && plan.nodes[l + 1] == 1));
        const int top = fuse ? l + 1 : l;           // the level whose nodes this launch leaves
        const bool last_level = top + 1 == plan.levels;
        const uint64_t out_nodes = plan.nodes[top];
        PerfOut o;
        PerfRaw r = { in_lasts, nullptr, nullptr, nullptr, nullptr };
        PerfIn next = in;
        if (last_level) {
            o = final;
            r.out_lasts = final_raw.out_lasts;
            r.out_packets = final_raw.out_packets;
            if (plan.levels == 1) { r.front = final_raw.front; r.unless = final_raw.unless; }
        } else {
            void *d2 = p;                       p += up16(out_nodes * out_cap * 8 + MRPDR_TPB * 8);
            uint64_t *lens2 = (uint64_t *)p;    p += up16(out_nodes * 8);
            uint64_t *offs2 = (uint64_t *)p;    p += up16(out_nodes * 8);
            uint64_t *rcnt2 = (uint64_t *)p;    p += up16(out_nodes * 8);
            uint64_t *roff2 = (uint64_t *)p;    p += up16(out_nodes * 8);
            uint64_t *lasts2 = (uint64_t *)p;   p += up16(out_nodes * 8);
            int32_t *ranks2 = (int32_t *)p;     p += up16(out_nodes * 4);
            o = PerfOut{ d2, out_cap, lens2, offs2, ranks2, rcnt2, roff2, c->status };
            if (raw) r.out_lasts = lasts2;
            next = PerfIn{ ranks2, rcnt2, roff2, d2, lens2, offs2, out_nodes, out_nodes };
        }
        if (!launch_perf_level<OP, T>(in, nodes, o, out_cap, r, regular, fuse, raw, st)) return 0;
        if (!last_level) { in_lasts = raw ? r.out_lasts : nullptr; regular = out_cap; }
        in = next;
        l = top;
    }
    return 1;
}

template <typename T, typename F>
static int32_t perf_op(const char *who, int32_t op, F &&run)
{
    switch (op) {
    case MRPDR_SUM: return run(std::integral_constant<int, MRPDR_SUM>());
    case MRPDR_MIN: return run(std::integral_constant<int, MRPDR_MIN>());
    case MRPDR_MAX: return run(std::integral_constant<int, MRPDR_MAX>());
    case MRPDR_AVG: return run(std::integral_constant<int, MRPDR_AVG>());
    }
    return fail("%s: unknown aggregation %d", who, op);
}

// type dispatch: PerfDataMgr metrics are uint64, int64 or double (src/FilterDefinitions.C:937-953)
template <typename F>
static int32_t perf_type(const char *who, int32_t type, F &&run)
{
    switch (type) {
    case MRPDR_UINT64_T: return run((uint64_t)0);
    case MRPDR_INT64_T:  return run((int64_t)0);
    case MRPDR_DOUBLE_T: return run((double)0);
    }
    return fail("%s: type %d is not a performance-data type", who, type);
}

static int32_t perf_check(const char *who, mrpdr_ctx *c, const mrpdr_column *ranks, const mrpdr_column *data, uint64_t n)
{
    if (!c || !ranks || !data) return fail("%s: NULL argument", who);
    if (!ctx_ready(who, c)) return 0;
    if (n > c->max_packets) return fail("%s: n exceeds the ctx limit", who);
    if (n && (!ranks->counts || !ranks->offs || !data->counts || !data->offs)) return fail("%s: columns need counts and offs", who);
    return 1;
}

static int32_t perf_run_plan(const char *who, mrpdr_ctx *c, int32_t op, int32_t type, const PerfIn &in, const PerfPlan &plan, bool raw,
                             uint64_t out_cap, const PerfOut &final, const PerfRaw &final_raw, void *scratch, uint64_t scratch_bytes,
                             hipStream_t st)
{
    const uint64_t need = perf_plan_bytes(plan, out_cap);
    if (need && (!scratch || scratch_bytes < need || ((uintptr_t)scratch & 15u)))
        return fail("%s: this fold needs %llu bytes of 16-byte aligned scratch", who, (unsigned long long)need);
    for (int l = 0; l < plan.levels; l++)
        if (plan.nodes[l] > 65535ull * (MRPDR_TPB / PERF_MIN_W)) return fail("%s: more than %u nodes on one level", who, 65535u * (MRPDR_TPB / PERF_MIN_W));
    return perf_type(who, type, [&](auto tv) {
        typedef decltype(tv) T;
        return perf_op<T>(who, op, [&](auto opv) {
            return run_perf_levels<decltype(opv)::value, T>(c, in, plan, raw, out_cap, final, final_raw, (uint8_t *)scratch, st);
        });
    });
}

extern "C" int32_t mrpdr_wave_perfdata_levels(mrpdr_ctx *c, int32_t op, int32_t type,
                                              const mrpdr_column *ranks, const mrpdr_column *data, uint64_t n,
                                              const uint64_t *fanins, int32_t nlevels, void *out_data, uint64_t out_cap,
                                              uint64_t *out_len, int32_t *out_rank,
                                              void *scratch, uint64_t scratch_bytes, void *stream)
{
    const char *who = "mrpdr_wave_perfdata_levels";
    if (!perf_check(who, c, ranks, data, n)) return 0;
    if (!out_len || !out_rank) return fail("%s: NULL argument", who);
    if (n == 0) return fail("%s: empty wave (the reference emits nothing)", who);
    if (!fanins || nlevels <= 0 || nlevels > PERF_MAX_LEVELS) return fail("%s: 1 to %d levels", who, PERF_MAX_LEVELS);
    if (out_cap && !out_data) return fail("%s: NULL out_data", who);
    PerfIn in = { (const int32_t *)ranks->values, ranks->counts, ranks->offs, data->values, data->counts, data->offs, n, n };
    PerfOut final = { out_data, out_cap, out_len, nullptr, out_rank, nullptr, nullptr, c->status };
    // a level of nodes of more than 65,535 ... is checked in perf_run_plan
    const PerfPlan plan = perf_plan(n, fanins, nlevels);
    if (perf_lanes(out_cap) > 64)
        for (int l = 0; l < plan.levels; l++)
            if (plan.nodes[l] > 65535) return fail("%s: more than 65535 nodes of long arrays on one level", who);
    return perf_run_plan(who, c, op, type, in, plan, false, out_cap, final, PerfRaw{ nullptr, nullptr, nullptr, nullptr, nullptr }, scratch,
                         scratch_bytes, (hipStream_t)stream);
}

extern "C" int32_t mrpdr_wave_perfdata(mrpdr_ctx *c, int32_t op, int32_t type,
                                       const mrpdr_column *ranks, const mrpdr_column *data, uint64_t n,
                                       uint64_t group, void *out_data, uint64_t out_cap,
                                       uint64_t *out_len, int32_t *out_rank,
                                       void *scratch, uint64_t scratch_bytes, void *stream)
{
    const char *who = "mrpdr_wave_perfdata";
    if (!perf_check(who, c, ranks, data, n)) return 0;
    if (!out_len || !out_rank) return fail("%s: NULL argument", who);
    if (n == 0) return fail("%s: empty wave (the reference emits nothing)", who);
    if (out_cap && !out_data) return fail("%s: NULL out_data", who);
    if (type != MRPDR_UINT64_T && type != MRPDR_INT64_T && type != MRPDR_DOUBLE_T)
        return fail("%s: type %d is not a performance-data type", who, type);
    PerfIn in = { (const int32_t *)ranks->values, ranks->counts, ranks->offs, data->values, data->counts, data->offs, n, n };
    PerfOut final = { out_data, out_cap, out_len, nullptr, out_rank, nullptr, nullptr, c->status };
    const PerfRaw none = { nullptr, nullptr, nullptr, nullptr, nullptr };
    if (group == 0 || group >= n) {
        // One invocation.  Integer data does not depend on the grouping (raw nodes, one division at the
        // top): with room for the nodes it is computed as a tree of short chains; doubles, and callers
        // without scratch, fold the packets one after the other.
        const PerfPlan free_plan = perf_plan(n, PERF_FREE_FANINS, PERF_FREE_LEVELS);
        const bool can = type != MRPDR_DOUBLE_T && n >= PERF_FREE_MIN && scratch && !((uintptr_t)scratch & 15u) &&
                         scratch_bytes >= perf_plan_bytes(free_plan, out_cap) && perf_lanes(out_cap) <= 64;
        if (can) return perf_run_plan(who, c, op, type, in, free_plan, true, out_cap, final, none, scratch, scratch_bytes, (hipStream_t)stream);
        const uint64_t all[1] = { 0 };
        const PerfPlan serial = perf_plan(n, all, 1);
        hipStream_t st = (hipStream_t)stream;
        // Doubles: the elements are columns -- in place if every packet carries the same number of them,
        // end to end, from a dense copy otherwise -- and their chains of roundings are taken by the
        // device-wide exact fold, side by side; the device finds out which (k_perf_shape), and the serial
        // fold below runs only if neither applies (a result longer than out_cap).
        if (type == MRPDR_DOUBLE_T && n >= PERF_COLUMNS_MIN && out_cap &&
            out_cap <= PERF_COLUMNS_MAX_CAP && scratch && !((uintptr_t)scratch & 15u) && scratch_bytes >= perf_columns_bytes(n, out_cap)) {
            const PerfColumns pc = perf_columns(scratch, n, out_cap);
            if (!perf_columns_run(op, in, out_cap, nullptr, pc, st)) return 0;
            if (op == MRPDR_SUM) k_perf_from_columns<MRPDR_SUM><<<1, MRPDR_TPB, 0, st>>>(pc.shape, pc.nodes, final);
            else if (op == MRPDR_AVG) k_perf_from_columns<MRPDR_AVG><<<1, MRPDR_TPB, 0, st>>>(pc.shape, pc.nodes, final);
            else k_perf_from_columns<MRPDR_MIN><<<1, MRPDR_TPB, 0, st>>>(pc.shape, pc.nodes, final);
            LAUNCHED("k_perf_from_columns");
            const uint64_t *shape = pc.shape;
            PerfRaw unless_regular = none;
            unless_regular.unless = shape;
            return perf_run_plan(who, c, op, type, in, serial, false, out_cap, final, unless_regular, nullptr, 0, st);
        }
        return perf_run_plan(who, c, op, type, in, serial, false, out_cap, final, none, nullptr, 0, st);
    }
    if ((n + group - 1) / group > 65535 && perf_lanes(out_cap) > 64) return fail("%s: more than 65535 first-level nodes", who);
    const uint64_t two[2] = { group, 0 };
    return perf_run_plan(who, c, op, type, in, perf_plan(n, two, 2), false, out_cap, final, none, scratch, scratch_bytes, (hipStream_t)stream);
}

// ---- a wave's PerfData fold in ranges (one per GPU) --------------------------------------------------
extern "C" int32_t mrpdr_wave_perfdata_partial(mrpdr_ctx *c, int32_t op, int32_t type,
                                               const mrpdr_column *ranks, const mrpdr_column *data, uint64_t n, uint64_t cap,
                                               const void *front, void *state, void *scratch, uint64_t scratch_bytes, void *stream)
{
    const char *who = "mrpdr_wave_perfdata_partial";
    if (!perf_check(who, c, ranks, data, n)) return 0;
    if (!state || ((uintptr_t)state & 7u)) return fail("%s: state must be an 8-byte aligned device buffer", who);
    if (type != MRPDR_UINT64_T && type != MRPDR_INT64_T && type != MRPDR_DOUBLE_T)
        return fail("%s: type %d is not a performance-data type", who, type);
    const bool chained = mrpdr_perfdata_is_chained(op, type) != 0;
    if (front && !chained) return fail("%s: integer ranges need no order: front must be NULL (combine all states)", who);
    if (front && ((uintptr_t)front & 7u)) return fail("%s: front must be 8-byte aligned", who);
    hipStream_t st = (hipStream_t)stream;
    const uint64_t bytes = mrpdr_perfdata_state_bytes(cap);
    if (n == 0) {           // no packet here: the state in front passes through (or "nothing" does)
        if (front) HIP_OK(hipMemcpyAsync(state, front, bytes, hipMemcpyDeviceToDevice, st));
        else HIP_OK(hipMemsetAsync(state, 0, bytes, st));
        return 1;
    }
    uint8_t *sp = (uint8_t *)state;
    PerfState *h = (PerfState *)sp;
    HIP_OK(hipMemsetAsync(state, 0, sizeof(PerfState), st));
    PerfIn in = { (const int32_t *)ranks->values, ranks->counts, ranks->offs, data->values, data->counts, data->offs, n, n };
    PerfOut final = { sp + sizeof(PerfState), cap, &h->longest, nullptr, &h->rank, nullptr, nullptr, c->status };
    PerfRaw raw = { nullptr, &h->last, (const uint8_t *)front, &h->packets, nullptr };
    const uint64_t all[1] = { 0 };
    PerfPlan plan = perf_plan(n, all, 1);
    if (chained && n >= PERF_COLUMNS_MIN && cap && cap <= PERF_COLUMNS_MAX_CAP && scratch &&
        !((uintptr_t)scratch & 15u) && scratch_bytes >= perf_columns_bytes(n, cap)) {
        // doubles: the elements' chains as columns, each continued from the front's accumulator (see
        // mrpdr_wave_perfdata); the serial fold below runs only if the device finds that front and range
        // do not fit the bound
        const PerfColumns pc = perf_columns(scratch, n, cap);
        if (!perf_columns_run(op, in, cap, front, pc, st)) return 0;
        k_perf_state_from_columns<<<1, 64, 0, st>>>(pc.shape, pc.nodes, (const uint8_t *)front, sp);
        LAUNCHED("k_perf_state_from_columns");
        const uint64_t *shape = pc.shape;
        raw.unless = shape;
    }
    if (!chained && n >= PERF_FREE_MIN && perf_lanes(cap) <= 64) {
        const PerfPlan free_plan = perf_plan(n, PERF_FREE_FANINS, PERF_FREE_LEVELS);
        if (scratch && !((uintptr_t)scratch & 15u) && scratch_bytes >= perf_plan_bytes(free_plan, cap)) plan = free_plan;
    }
    return perf_run_plan(who, c, op, type, in, plan, true, cap, final, raw, scratch, scratch_bytes, st);
}

extern "C" int32_t mrpdr_wave_perfdata_combine(mrpdr_ctx *c, int32_t op, int32_t type, const void *states, uint64_t g, uint64_t cap,
                                               void *out_data, uint64_t out_cap, uint64_t *out_len, int32_t *out_rank, void *stream)
{
    const char *who = "mrpdr_wave_perfdata_combine";
    if (!c || !out_len || !out_rank) return fail("%s: NULL argument", who);
    if (!ctx_ready(who, c)) return 0;
    if (g == 0 || !states) return fail("%s: no states", who);
    if ((uintptr_t)states & 7u) return fail("%s: states must be 8-byte aligned", who);
    if (out_cap && !out_data) return fail("%s: NULL out_data", who);
    const uint64_t stride = mrpdr_perfdata_state_bytes(cap);
    const uint8_t *sp = (const uint8_t *)states;
    const int32_t relayed = mrpdr_perfdata_is_chained(op, type);
    if (relayed) { sp += (g - 1) * stride; g = 1; }     // the last state holds the whole fold
    PerfOut o = { out_data, out_cap, out_len, nullptr, out_rank, nullptr, nullptr, c->status };
    const uint64_t span = cap > out_cap ? cap : out_cap;
    const uint32_t xb = blocks_for(span ? span : 1, MRPDR_TPB);
    hipStream_t st = (hipStream_t)stream;
    return perf_type(who, type, [&](auto tv) {
        typedef decltype(tv) T;
        return perf_op<T>(who, op, [&](auto opv) {
            k_perf_combine<decltype(opv)::value, T><<<xb, MRPDR_TPB, 0, st>>>(sp, g, stride, o, relayed);
            LAUNCHED("k_perf_combine");
            return 1;
        });
    });
}

// ---------------------------------------------------------------------------
// tfilter_IntEqClass
// ---------------------------------------------------------------------------
static uint64_t eq_pad(uint64_t entries)
{
    uint64_t k = EQ_LOCAL;
    while (k < entries) k <<= 1;
    return k;
}

extern "C" uint64_t mrpdr_eqclass_scratch_bytes(uint64_t class_capacity)
{
    return 7 * (eq_pad(class_capacity) + 2) * sizeof(uint64_t);
}

extern "C" int32_t mrpdr_wave_int_eq_class(mrpdr_ctx *c, const mrpdr_column *vals, const mrpdr_column *memcnts,
                                           const mrpdr_column *mems, uint64_t n,
                                           mrpdr_column *out_vals, mrpdr_column *out_memcnts, mrpdr_column *out_mems,
                                           void *scratch, uint64_t scratch_bytes, void *stream)
{
    if (!c || !vals || !memcnts || !mems || !out_vals || !out_memcnts || !out_mems)
        return fail("mrpdr_wave_int_eq_class: NULL argument");
    if (!ctx_ready("mrpdr_wave_int_eq_class", c)) return 0;
    if (n == 0) return fail("mrpdr_wave_int_eq_class: empty wave (the reference emits nothing useful)");
    if (n > c->max_packets) return fail("mrpdr_wave_int_eq_class: n exceeds the ctx limit");
    if (!vals->offs || !mems->offs) return fail("mrpdr_wave_int_eq_class: columns need offs");
    const mrpdr_column *outs[3] = { out_vals, out_memcnts, out_mems };
    for (int k = 0; k < 3; k++)
        if (!outs[k]->counts || !outs[k]->offs || (outs[k]->capacity && !outs[k]->values))
            return fail("mrpdr_wave_int_eq_class: output column %d needs counts, offs and values", k);
    if (out_memcnts->capacity < out_vals->capacity) return fail("mrpdr_wave_int_eq_class: out_memcnts smaller than out_vals");
    const uint64_t entries = vals->capacity;          // upper bound of the class entries of the wave
    if (entries >= 0xFFFFFFFFull) return fail("mrpdr_wave_int_eq_class: 2^32 - 1 class entries or more");
    const uint64_t kpad = eq_pad(entries);
    if (kpad > c->max_packets) return fail("mrpdr_wave_int_eq_class: %llu class entries exceed the ctx limit",
                                           (unsigned long long)entries);
    if (!scratch || scratch_bytes < mrpdr_eqclass_scratch_bytes(entries) || ((uintptr_t)scratch & 15u))
        return fail("mrpdr_wave_int_eq_class: needs %llu bytes of 16-byte aligned scratch",
                    (unsigned long long)mrpdr_eqclass_scratch_bytes(entries));
    hipStream_t st = (hipStream_t)stream;
    uint64_t *keys = (uint64_t *)scratch, *cnt64 = keys + kpad + 2, *head = cnt64 + kpad + 2,
             *sorted_cnt = head + kpad + 2, *class_scan = sorted_cnt + kpad + 2, *dst_off = class_scan + kpad + 2,
             *src_off = dst_off + kpad + 2;
    const uint32_t blocks = blocks_for(kpad, MRPDR_TPB), tiles = blocks_for(kpad, EQ_LOCAL);
    k_eq_keys<<<blocks, MRPDR_TPB, 0, st>>>((const uint32_t *)vals->values, vals->offs, n, kpad, keys,
                                            (const uint32_t *)memcnts->values, cnt64);
    LAUNCHED("k_eq_keys");
    if (!run_scan(c, cnt64, kpad, 1, 0, src_off, st)) return 0;
    // bitonic network: merges up to EQ_LOCAL inside LDS, then per merge size the far steps over
    // global memory and the near steps in LDS again
    k_eq_bitonic_local<<<tiles, MRPDR_TPB, 0, st>>>(keys, kpad, 2, EQ_LOCAL);
    LAUNCHED("k_eq_bitonic_local");
    for (uint64_t k = 2 * EQ_LOCAL; k <= kpad; k <<= 1) {
        for (uint64_t j = k >> 1; j >= EQ_LOCAL; j >>= 1) {
            k_eq_bitonic_global<<<blocks, MRPDR_TPB, 0, st>>>(keys, kpad, j, k);
            LAUNCHED("k_eq_bitonic_global");
        }
        k_eq_bitonic_local<<<tiles, MRPDR_TPB, 0, st>>>(keys, kpad, k, k);
        LAUNCHED("k_eq_bitonic_local");
    }
    k_eq_heads<<<blocks, MRPDR_TPB, 0, st>>>(keys, vals->offs, n, kpad, cnt64, head, sorted_cnt);
    LAUNCHED("k_eq_heads");
    if (!run_scan(c, head, kpad, 1, 0, class_scan, st)) return 0;
    if (!run_scan(c, sorted_cnt, kpad, 1, 0, dst_off, st)) return 0;
    EqOut o;
    o.values = (uint32_t *)out_vals->values; o.memcnts = (uint32_t *)out_memcnts->values; o.mems = (uint32_t *)out_mems->values;
    o.values_cap = out_vals->capacity; o.mems_cap = out_mems->capacity;
    for (int k = 0; k < 3; k++) { o.counts[k] = outs[k]->counts; o.offs[k] = outs[k]->offs; }
    o.status = c->status;
    k_eq_emit<<<blocks, MRPDR_TPB, 0, st>>>(keys, vals->offs, n, kpad, head, class_scan, dst_off, src_off, sorted_cnt,
                                            (const uint32_t *)mems->values, mems->offs, o);
    LAUNCHED("k_eq_emit");
    return 1;
}
