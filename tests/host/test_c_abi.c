/* The C ABI from plain C (C99, gcc): what a C back-end library in the style of the reference's
 * src/lightweight/ would write.  One wave of "%d %s %ad" packets is encoded, decoded again,
 * summed and sent through a Message; a "%f" wave is summed in three ranges; a wave of performance-data
 * packets goes through tfilter_PerfData; output in the reference's test-harness style (tests/test_common.C:41-187). */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mrpdr.h"

/* the two HIP runtime calls this program needs, as libamdhip64 exports them (C linkage);
 * 0 = hipSuccess, copy kinds 1 = host to device, 2 = device to host */
extern int hipMalloc(void **ptr, size_t bytes);
extern int hipMemcpy(void *dst, const void *src, size_t bytes, int kind);
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };

static int failures = 0;
static void end_SubTest(const char *name, int ok)
{
    printf("%s: %s\n", name, ok ? "MRNTEST_SUCCESS" : "MRNTEST_FAILURE");
    if (!ok) failures++;
}

static void *to_device(const void *src, size_t bytes)
{
    void *p = NULL;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { printf("hipMalloc failed\n"); exit(2); }
    int rc = (bytes && src) ? hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) : hipSuccess;   /* src NULL: allocate only */
    if (rc != hipSuccess) { printf("hipMemcpy failed: %d (%zu bytes, src %p)\n", rc, bytes, src); exit(2); }
    return p;
}

int main(void)
{
    enum { N = 1000 };
    static int32_t v[N], arr[3 * N];
    static uint64_t scnt[N], soffs[N + 1], acnt[N], aoffs[N + 1];
    static char text[6 * N];
    int64_t want = 0;
    for (int i = 0; i < N; i++) {
        v[i] = i * 7 - 3000; want += v[i];
        scnt[i] = 6; soffs[i] = 6u * i; memcpy(text + 6 * i, "hello", 6);
        acnt[i] = 3; aoffs[i] = 3u * i;
        for (int k = 0; k < 3; k++) arr[3 * i + k] = 100 * i + k;
    }
    soffs[N] = 6u * N; aoffs[N] = 3u * N;

    mrpdr_ctx *ctx = mrpdr_ctx_create(0, 1 << 16);
    mrpdr_fmt *fmt = mrpdr_fmt_compile("%d %s %ad");
    end_SubTest("mrpdr_ctx_create / mrpdr_fmt_compile", ctx != NULL && fmt != NULL && mrpdr_fmt_nfields(fmt) == 3);
    if (!ctx || !fmt) { printf("%s\n", mrpdr_last_error()); return 1; }

    mrpdr_column cols[3];
    memset(cols, 0, sizeof cols);
    cols[0].values = to_device(v, sizeof v);          cols[0].capacity = N;
    cols[1].values = to_device(text, sizeof text);    cols[1].capacity = 6 * N;
    cols[1].counts = to_device(scnt, sizeof scnt);    cols[1].offs = to_device(soffs, sizeof soffs);
    cols[2].values = to_device(arr, sizeof arr);      cols[2].capacity = 3 * N;
    cols[2].counts = to_device(acnt, sizeof acnt);    cols[2].offs = to_device(aoffs, sizeof aoffs);

    /* Packet::encode_pdr_data for the wave: size, then encode */
    uint64_t *lens = to_device(NULL, 8 * N), *offs = to_device(NULL, 8 * (N + 1)), total = 0;
    int ok = mrpdr_wave_sizeof(ctx, fmt, cols, N, 1, 0, lens, offs, &total, NULL);
    ok = ok && total == (uint64_t)N * (4 + 4 + 6 + 8 + 12);
    uint8_t *arena = to_device(NULL, total);
    mrpdr_wire wire = { arena, offs, lens, 0, total };
    ok = ok && mrpdr_wave_encode(ctx, fmt, cols, N, &wire, NULL);
    static uint8_t first[34];
    hipMemcpy(first, arena, sizeof first, hipMemcpyDeviceToHost);
    const uint8_t expect[34] = { 0x48, 0xf4, 0xff, 0xff, 6, 0, 0, 0, 'h', 'e', 'l', 'l', 'o', 0,
                                 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0 };
    end_SubTest("mrpdr_wave_sizeof / mrpdr_wave_encode", ok && !memcmp(first, expect, sizeof expect));

    /* Packet::decode_pdr_data for the wave, into fresh columns */
    mrpdr_column out[3];
    memset(out, 0, sizeof out);
    out[0].values = to_device(NULL, sizeof v);        out[0].capacity = N;
    out[1].values = to_device(NULL, sizeof text);     out[1].capacity = 6 * N;
    out[1].counts = to_device(NULL, sizeof scnt);     out[1].offs = to_device(NULL, sizeof soffs);
    out[2].values = to_device(NULL, sizeof arr);      out[2].capacity = 3 * N;
    out[2].counts = to_device(NULL, sizeof acnt);     out[2].offs = to_device(NULL, sizeof aoffs);
    int32_t *okd = to_device(NULL, 4 * N);
    ok = mrpdr_wave_decode(ctx, fmt, &wire, N, MRPDR_LITTLEENDIAN, out, okd, NULL);
    static int32_t v2[N], arr2[3 * N], okh[N];
    static char text2[6 * N];
    hipMemcpy(v2, out[0].values, sizeof v2, hipMemcpyDeviceToHost);
    hipMemcpy(text2, out[1].values, sizeof text2, hipMemcpyDeviceToHost);
    hipMemcpy(arr2, out[2].values, sizeof arr2, hipMemcpyDeviceToHost);
    hipMemcpy(okh, okd, sizeof okh, hipMemcpyDeviceToHost);
    int all_ok = 1;
    for (int i = 0; i < N; i++) all_ok = all_ok && okh[i] == 1;
    end_SubTest("mrpdr_wave_decode", ok && all_ok && !memcmp(v, v2, sizeof v) && !memcmp(text, text2, sizeof text) &&
                                      !memcmp(arr, arr2, sizeof arr));

    /* tfilter_Sum over field 0, straight on the wire bytes is not possible here (packets differ in
     * nothing but could): fold the decoded column */
    uint8_t *res = to_device(NULL, 16);
    int32_t *rok = to_device(NULL, 4), sum = 0, present = 0;
    ok = mrpdr_wave_filter(ctx, MRPDR_SUM, MRPDR_INT32_T, MRPDR_SHAPE_FLAT, out[0].values, 4, NULL, 0, N, 0, res, rok, NULL);
    hipMemcpy(&sum, res, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&present, rok, 4, hipMemcpyDeviceToHost);
    end_SubTest("mrpdr_wave_filter(MRPDR_SUM)", ok && present == 1 && sum == (int32_t)want);

    /* Message::send / Message::recv for the wave (src/Message.C:201-335, 48-199): frame and data into
     * one buffer, then size vector, headers and data out of it again, no host read in between */
    {
        const uint64_t h = mrpdr_header_bytes(fmt), pre = mrpdr_message_preamble_bytes(N);
        uint64_t payload = 0;
        ok = mrpdr_wave_sizeof(ctx, fmt, cols, N, 1, (uint32_t)h, lens, offs, &payload, NULL);
        uint8_t *msg = to_device(NULL, pre + payload);
        ok = ok && mrpdr_message_write_frame(ctx, fmt, N, lens, offs, 0x40000010u, NULL, 101, NULL, 7, NULL, msg, NULL);
        mrpdr_wire mw = { msg + pre, offs, lens, 0, payload };
        ok = ok && mrpdr_wave_encode(ctx, fmt, cols, N, &mw, NULL);
        static uint8_t head[5 + 16];
        hipMemcpy(head, msg, sizeof head, hipMemcpyDeviceToHost);
        uint64_t h0 = 0, b0 = 0;
        memcpy(&h0, head + 5, 8); memcpy(&b0, head + 13, 8);
        ok = ok && head[0] == MRPDR_LITTLEENDIAN && head[1] == (N & 255) && head[2] == (N >> 8) && h0 == h && b0 == 34;
        uint64_t *ho = to_device(NULL, 8 * (N + 1)), *hl = to_device(NULL, 8 * N);
        uint64_t *bo = to_device(NULL, 8 * (N + 1)), *bl = to_device(NULL, 8 * N);
        uint32_t *sid = to_device(NULL, 4 * N), *src = to_device(NULL, 4 * N);
        int32_t *tag = to_device(NULL, 4 * N), *hok = to_device(NULL, 4 * N);
        ok = ok && mrpdr_message_parse(ctx, fmt, msg, pre + payload, N, ho, hl, bo, bl, sid, tag, src, NULL, NULL,
                                       NULL, 0, hok, NULL);
        mrpdr_wire rw = { msg + pre, bo, bl, 0, payload };
        ok = ok && mrpdr_wave_decode(ctx, fmt, &rw, N, MRPDR_LITTLEENDIAN, out, okd, NULL);
        static uint32_t sidh[N], srch[N];
        static int32_t tagh[N], hokh[N];
        uint64_t announced = 0;
        hipMemcpy(sidh, sid, sizeof sidh, hipMemcpyDeviceToHost); hipMemcpy(srch, src, sizeof srch, hipMemcpyDeviceToHost);
        hipMemcpy(tagh, tag, sizeof tagh, hipMemcpyDeviceToHost); hipMemcpy(hokh, hok, sizeof hokh, hipMemcpyDeviceToHost);
        hipMemcpy(&announced, ho + N, 8, hipMemcpyDeviceToHost);
        hipMemcpy(v2, out[0].values, sizeof v2, hipMemcpyDeviceToHost);
        hipMemcpy(arr2, out[2].values, sizeof arr2, hipMemcpyDeviceToHost);
        hipMemcpy(okh, okd, sizeof okh, hipMemcpyDeviceToHost);
        for (int i = 0; i < N; i++)
            ok = ok && hokh[i] == 1 && okh[i] == 1 && sidh[i] == 0x40000010u && tagh[i] == 101 && srch[i] == 7;
        end_SubTest("mrpdr_message_write_frame / mrpdr_message_parse", ok && announced == payload &&
                                                                       !memcmp(v, v2, sizeof v) && !memcmp(arr, arr2, sizeof arr));
    }

    /* One tfilter_Sum call over a "%f" wave that three GPUs would hold: every range leaves a node, the
     * chain of additions goes from node to node (mrpdr_wave_filter_partial), the last node is the
     * result (mrpdr_wave_filter_combine) -- the bits of the loop src/FilterDefinitions.C:148-200 runs. */
    {
        enum { M = 40000 };
        static float x[M];
        float chain = 0;
        for (int i = 0; i < M; i++) { x[i] = (float)((i * 2654435761u) >> 8) * 1e-4f - 800.0f; chain = chain + x[i]; }
        float *xd = to_device(x, sizeof x);
        mrpdr_node *nodes = to_device(NULL, 3 * sizeof(mrpdr_node));
        const int edges[4] = { 0, 9000, 9000 + 20000, M };
        ok = mrpdr_filter_is_chained(MRPDR_SUM, MRPDR_FLOAT_T) == 1 && mrpdr_filter_is_chained(MRPDR_SUM, MRPDR_INT32_T) == 0;
        for (int r = 0; r < 3; r++)
            ok = ok && mrpdr_wave_filter_partial(ctx, MRPDR_SUM, MRPDR_FLOAT_T, xd + edges[r], 4, NULL, 0,
                                                 (uint64_t)(edges[r + 1] - edges[r]), 0, r ? nodes + r - 1 : NULL, nodes + r, NULL);
        ok = ok && mrpdr_wave_filter_combine(ctx, MRPDR_SUM, MRPDR_FLOAT_T, nodes, 3, res, rok, NULL);
        float got = 0;
        hipMemcpy(&got, res, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&present, rok, 4, hipMemcpyDeviceToHost);
        end_SubTest("mrpdr_wave_filter_partial / mrpdr_wave_filter_combine", ok && present == 1 && !memcmp(&got, &chain, 4));
    }

    /* tfilter_PerfData, TFILTER_AVG over doubles, one invocation over 10,000 packets whose arrays have
     * 0 .. 8 elements (src/FilterDefinitions.C:984-1132): packets without data are nobody's rank, every
     * element is a chain of additions over the packets that carry it, the elements the LAST packet
     * carries are divided by the rank total. */
    {
        enum { P = 10000, K = 8 };
        static int32_t rank[P];
        static uint64_t one[P], roffs[P + 1], dcnt[P], doffs[P + 1];
        static double dv[P * K], acc[K];
        uint64_t at = 0, longest = 0;
        unsigned total = 0;
        for (int i = 0; i < P; i++) {
            rank[i] = (i % 5 == 0) ? -(1 + i % 3) : i;
            one[i] = 1; roffs[i] = (uint64_t)i;
            dcnt[i] = (uint64_t)((i * 7 + 3) % (K + 1)); doffs[i] = at;
            if (i == P - 1) dcnt[i] = 3;
            const int nrank = rank[i] < 0 ? -rank[i] : 1;
            if (dcnt[i]) total += (unsigned)nrank;
            if (dcnt[i] > longest) longest = dcnt[i];
            for (uint64_t u = 0; u < dcnt[i]; u++) {
                double v = 1e-3 + 1e-7 * (double)((i * 31 + (int)u * 17) % 1000);
                dv[at + u] = v;
                v *= nrank;
                acc[u] += v;
            }
            at += dcnt[i];
        }
        roffs[P] = P; doffs[P] = at;
        for (uint64_t u = 0; u < dcnt[P - 1]; u++) acc[u] /= total;
        mrpdr_column rk, da;
        memset(&rk, 0, sizeof rk); memset(&da, 0, sizeof da);
        rk.values = to_device(rank, sizeof rank); rk.counts = to_device(one, sizeof one); rk.offs = to_device(roffs, sizeof roffs); rk.capacity = P;
        da.values = to_device(dv, 8 * (at ? at : 1)); da.counts = to_device(dcnt, sizeof dcnt); da.offs = to_device(doffs, sizeof doffs); da.capacity = at;
        const uint64_t need = mrpdr_perfdata_scratch_bytes(P, 0, K);
        void *scratch = to_device(NULL, need);
        double *outd = to_device(NULL, 8 * K);
        uint64_t *outl = to_device(NULL, 8), len = 0;
        int32_t *outr = to_device(NULL, 4), orank = 0;
        static double gotd[K];
        ok = mrpdr_wave_perfdata(ctx, MRPDR_AVG, MRPDR_DOUBLE_T, &rk, &da, P, 0, outd, K, outl, outr, scratch, need, NULL);
        hipMemcpy(gotd, outd, sizeof gotd, hipMemcpyDeviceToHost);
        hipMemcpy(&len, outl, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&orank, outr, 4, hipMemcpyDeviceToHost);
        end_SubTest("mrpdr_wave_perfdata(MRPDR_AVG, doubles, ragged arrays)", ok && need > 0 && len == longest && orank == -(int32_t)total &&
                                                                             !memcmp(gotd, acc, 8 * (size_t)longest));
    }

    uint32_t status = 1;
    ok = mrpdr_ctx_status(ctx, NULL, &status) && status == 0;
    ok = ok && !mrpdr_wave_decode(ctx, fmt, NULL, N, MRPDR_LITTLEENDIAN, out, okd, NULL) && strstr(mrpdr_last_error(), "NULL wire");
    end_SubTest("mrpdr_ctx_status / refusal with a reason", ok);

    mrpdr_fmt_free(fmt);
    mrpdr_ctx_destroy(ctx);
    printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
