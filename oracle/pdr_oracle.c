/*
 * pdr_oracle.c -- CPU restatement of MRNet's PDR packet marshalling path.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP codec and the timed CPU
 * baseline.  The product library never links or calls this file.
 *
 * The structure deliberately follows the reference so that it is the same
 * algorithm, not just the same bytes: one stream object with a table of
 * put/get function pointers, one call through that table per scalar or array
 * element, a dry-run sizing pass before every encode, the format string
 * tokenised again for every packet, one heap object per field per packet and
 * one heap block per decoded array.
 *
 * Build with -fwrapv -ffp-contract=off: the reference's integer filters rely
 * on two's-complement wrap-around and its float filters on separate multiply
 * and add roundings.
 */
#define _POSIX_C_SOURCE 200809L      /* clock_gettime under -std=c11 */
#include "pdr_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ======================================================================= */
/* byte order (src/byte_order.c:8-17)                                      */
/* ======================================================================= */
void orc_byte_swap(char *out, const char *in, uint32_t nelems, uint32_t elsize)
{
    uint32_t total = nelems * elsize;
    for (uint32_t base = 0; base < total; base += elsize)
        for (uint32_t k = 0; k < elsize; k++)
            out[base + k] = in[base + elsize - 1 - k];
}

int orc_local_bo(void)                       /* src/pdr_mem.c:421-428 */
{
    const uint16_t probe = 1;
    return (*(const unsigned char *)&probe) ? ORC_LE : ORC_BE;
}

/* ======================================================================= */
/* memory stream (src/pdr_mem.c:22-372)                                    */
/* every primitive: width > space -> FALSE; unaligned native copy; advance */
/* ======================================================================= */
#define MEM_PUT(NAME, W)                                                       \
    static int32_t NAME(orc_stream *s, void *p)                                \
    {                                                                          \
        if ((uint64_t)(W) > s->space) return 0;                                \
        memcpy(s->cur, p, (W));                                                \
        s->cur += (W); s->space -= (W);                                        \
        return 1;                                                              \
    }
#define MEM_GET(NAME, W)                                                       \
    static int32_t NAME(orc_stream *s, void *p)                                \
    {                                                                          \
        if ((uint64_t)(W) > s->space) return 0;                                \
        memcpy(p, s->cur, (W));                                                \
        s->cur += (W); s->space -= (W);                                        \
        return 1;                                                              \
    }
/* swap variants call the generic byte reverser with nelems = 1, as
 * pdrmem_get*_swap do through swap_int16/32/64/float/double
 * (src/pdr_mem.c:96-106,146-158,198-210,250-262,303-315; src/byte_order.h:37-46) */
#define MEM_GET_SWAP(NAME, W)                                                  \
    static int32_t NAME(orc_stream *s, void *p)                                \
    {                                                                          \
        if ((uint64_t)(W) > s->space) return 0;                                \
        orc_byte_swap((char *)p, s->cur, 1, (W));                              \
        s->cur += (W); s->space -= (W);                                        \
        return 1;                                                              \
    }

MEM_PUT(mem_put8, 1)   MEM_GET(mem_get8, 1)
MEM_PUT(mem_put16, 2)  MEM_GET(mem_get16, 2)  MEM_GET_SWAP(mem_get16_swap, 2)
MEM_PUT(mem_put32, 4)  MEM_GET(mem_get32, 4)  MEM_GET_SWAP(mem_get32_swap, 4)
MEM_PUT(mem_put64, 8)  MEM_GET(mem_get64, 8)  MEM_GET_SWAP(mem_get64_swap, 8)
MEM_PUT(mem_putf, 4)   MEM_GET(mem_getf, 4)   MEM_GET_SWAP(mem_getf_swap, 4)
MEM_PUT(mem_putd, 8)   MEM_GET(mem_getd, 8)   MEM_GET_SWAP(mem_getd_swap, 8)

static int32_t mem_putbytes(orc_stream *s, char *addr, uint64_t len)   /* src/pdr_mem.c:332-343 */
{
    if (len > s->space) return 0;
    memcpy(s->cur, addr, (size_t)len);
    s->cur += len; s->space -= len;
    return 1;
}
static int32_t mem_getbytes(orc_stream *s, char *addr, uint64_t len)   /* src/pdr_mem.c:317-330 */
{
    if (len > s->space) return 0;
    memcpy(addr, s->cur, (size_t)len);
    s->cur += len; s->space -= len;
    return 1;
}

/* src/pdr_mem.c:374-412: chars and raw bytes are shared by both tables */
static const struct orc_ops mem_ops = {
    mem_put8, mem_get8, mem_put16, mem_get16, mem_put32, mem_get32, mem_put64, mem_get64,
    mem_putf, mem_getf, mem_putd, mem_getd, mem_putbytes, mem_getbytes };
static const struct orc_ops mem_ops_swap = {
    mem_put8, mem_get8, mem_put16, mem_get16_swap, mem_put32, mem_get32_swap,
    mem_put64, mem_get64_swap, mem_putf, mem_getf_swap, mem_putd, mem_getd_swap,
    mem_putbytes, mem_getbytes };

void orc_mem_create(orc_stream *s, char *addr, uint64_t size, int op, int sender_bo)
{                                                        /* src/pdr_mem.c:433-451 */
    s->cur = s->base = addr;
    s->space = size;
    s->op = op;
    s->ops = &mem_ops;
    if (op == ORC_DECODE && (char)sender_bo != (char)orc_local_bo())
        s->ops = &mem_ops_swap;
}

/* ======================================================================= */
/* sizing stream (src/pdr_sizeof.c:9-140): puts add widths, gets fail      */
/* ======================================================================= */
#define CNT_PUT(NAME, W) \
    static int32_t NAME(orc_stream *s, void *p) { (void)p; s->space += (W); return 1; }
static int32_t cnt_get(orc_stream *s, void *p) { (void)s; (void)p; return 0; }
CNT_PUT(cnt_put8, 1) CNT_PUT(cnt_put16, 2) CNT_PUT(cnt_put32, 4) CNT_PUT(cnt_put64, 8)
static int32_t cnt_putbytes(orc_stream *s, char *a, uint64_t len) { (void)a; s->space += len; return 1; }
static int32_t cnt_getbytes(orc_stream *s, char *a, uint64_t len) { (void)s; (void)a; (void)len; return 0; }
static const struct orc_ops cnt_ops = {
    cnt_put8, cnt_get, cnt_put16, cnt_get, cnt_put32, cnt_get, cnt_put64, cnt_get,
    cnt_put32, cnt_get, cnt_put64, cnt_get, cnt_putbytes, cnt_getbytes };

int32_t orc_sizeof(orc_proc fn, void *obj, uint64_t *size)   /* src/pdr_sizeof.c:120-140 */
{
    orc_stream s;
    s.op = ORC_ENCODE; s.ops = &cnt_ops; s.space = 0; s.cur = s.base = NULL;
    int32_t rc = fn(&s, obj);
    *size = (rc == 1) ? s.space : 0;
    return rc;
}

/* ======================================================================= */
/* generic routines (src/pdr.c:34-127): switch on op, call through table   */
/* ======================================================================= */
#define GENERIC(NAME, PUT, GET)                                                \
    int32_t NAME(orc_stream *s, void *p)                                       \
    {                                                                          \
        switch (s->op) {                                                       \
        case ORC_FREE:   return 1;                                             \
        case ORC_ENCODE: return s->ops->PUT(s, p);                             \
        case ORC_DECODE: return s->ops->GET(s, p);                             \
        }                                                                      \
        return 0;                                                              \
    }
GENERIC(orc_char, put8, get8)
GENERIC(orc_i16, put16, get16)
GENERIC(orc_i32, put32, get32)
GENERIC(orc_i64, put64, get64)
GENERIC(orc_f32, putf, getf)
GENERIC(orc_f64, putd, getd)

int32_t orc_opaque(orc_stream *s, char *cp, uint64_t cnt)    /* src/pdr.c:187-205: no round-up */
{
    if (cnt == 0) return 1;
    switch (s->op) {
    case ORC_DECODE: return s->ops->getbytes(s, cp, cnt);
    case ORC_ENCODE: return s->ops->putbytes(s, cp, cnt);
    case ORC_FREE:   return 1;
    }
    return 0;
}

int32_t orc_bytes(orc_stream *s, char **cpp, uint64_t *sizep, uint64_t maxsize)   /* src/pdr.c:212-252 */
{
    char *sp = *cpp;
    if (!orc_i64(s, sizep)) return 0;
    uint64_t n = *sizep;
    if (n > maxsize && s->op != ORC_FREE) return 0;
    switch (s->op) {
    case ORC_DECODE:
        if (n == 0) return 1;                     /* *cpp stays untouched (NULL) */
        /* the reference mallocs the count it read before pdr_opaque finds the stream too short;
         * a corrupt count asks for exabytes.  Same verdict without the allocation: */
        if (n > s->space) return 0;
        if (sp == NULL) *cpp = sp = (char *)malloc((size_t)n);
        if (sp == NULL) return 0;
        /* fall through */
    case ORC_ENCODE:
        return orc_opaque(s, sp, n);
    case ORC_FREE:
        if (sp) { free(sp); *cpp = NULL; }
        return 1;
    }
    return 0;
}

int32_t orc_string(orc_stream *s, char **cpp, uint32_t maxsize)    /* src/pdr.c:263-325 */
{
    char *sp = *cpp;
    size_t n = 0;
    switch (s->op) {
    case ORC_FREE:
        if (sp) { free(sp); *cpp = NULL; }
        return 1;
    case ORC_ENCODE:
        if (sp == NULL) return 0;                 /* NULL string cannot be encoded */
        n = strlen(sp) + 1;
        break;
    case ORC_DECODE:
        break;
    }
    if (n > maxsize) return 0;
    uint32_t n32 = (uint32_t)n;
    if (!orc_i32(s, &n32)) return 0;
    switch (s->op) {
    case ORC_FREE: return 1;
    case ORC_DECODE:
        if (n32 == 0) return 0;   /* the reference writes sp[-1] here (undefined); refuse instead */
        if (n32 > s->space) return 0;   /* pdr_opaque would refuse it after a malloc of up to 4 GiB */
        if (sp == NULL) { sp = (char *)malloc((size_t)n32); *cpp = sp; }
        if (sp == NULL) return 0;
        sp[n32 - 1] = '\0';
        /* fall through */
    case ORC_ENCODE:
        return orc_opaque(s, sp, n32);
    }
    return 0;
}

int32_t orc_wrapstring(orc_stream *s, void *cpp)                    /* src/pdr.c:330-333 */
{
    return orc_string(s, (char **)cpp, 0xFFFFFFFEu);
}

int32_t orc_array(orc_stream *s, void **addrp, uint64_t *sizep, uint64_t maxsize,
                  uint32_t elsize, orc_proc elproc)                 /* src/pdr.c:415-472 */
{
    char *target = (char *)*addrp;
    int32_t rc = 1;
    if (!orc_i64(s, sizep)) return 0;
    uint64_t c = *sizep;
    if (c > maxsize && s->op != ORC_FREE) return 0;
    uint64_t nodesize = c * elsize;
    if (target == NULL) {
        switch (s->op) {
        case ORC_DECODE:
            if (c == 0) return 1;
            /* The reference callocs count*elsize before looking at the remaining
             * space; a corrupt count would ask for petabytes, and one of 2^61 or more
             * wraps count*elsize to a small block that the element loop then overruns
             * (undefined behaviour in the reference).  Fail such packets up front: the
             * element loop below would fail them anyway. */
            if (elsize != sizeof(char *) && c > s->space / elsize) return 0;    /* (count*elsize may wrap) */
            if (elsize == sizeof(char *) && c > s->space) return 0;
            *addrp = calloc((size_t)nodesize, 1);
            target = (char *)*addrp;
            if (target == NULL) return 0;
            break;
        case ORC_FREE:
            return 1;
        case ORC_ENCODE:
            break;
        }
    }
    for (uint64_t i = 0; i < c && rc; i++) {       /* the hot loop, src/pdr.c:459-462 */
        rc = elproc(s, target);
        target += elsize;
    }
    if (s->op == ORC_FREE) { free(*addrp); *addrp = NULL; }
    return rc;
}

int32_t orc_vector(orc_stream *s, char *basep, uint64_t nelem, uint64_t elsize, orc_proc proc)
{                                                                   /* src/pdr.c:484-498 */
    for (uint64_t i = 0; i < nelem; i++, basep += elsize)
        if (!proc(s, basep)) return 0;
    return 1;
}

/* ======================================================================= */
/* format tokens (src/DataElement.C:205-297, xplat/src/Tokenizer.C:23-58)  */
/* ======================================================================= */
static const struct { const char *tok; int type; } fmt_table[] = {
    {"c", ORC_CHAR}, {"uc", ORC_UCHAR}, {"hd", ORC_INT16}, {"uhd", ORC_UINT16},
    {"d", ORC_INT32}, {"ud", ORC_UINT32}, {"ld", ORC_INT64}, {"uld", ORC_UINT64},
    {"f", ORC_FLOAT}, {"lf", ORC_DOUBLE}, {"s", ORC_STRING},
    {"ac", ORC_CHAR_ARRAY}, {"auc", ORC_UCHAR_ARRAY}, {"ahd", ORC_INT16_ARRAY},
    {"auhd", ORC_UINT16_ARRAY}, {"ad", ORC_INT32_ARRAY}, {"aud", ORC_UINT32_ARRAY},
    {"ald", ORC_INT64_ARRAY}, {"auld", ORC_UINT64_ARRAY}, {"af", ORC_FLOAT_ARRAY},
    {"alf", ORC_DOUBLE_ARRAY}, {"as", ORC_STRING_ARRAY},
    {"Ac", ORC_CHAR_LRG_ARRAY}, {"Auc", ORC_UCHAR_LRG_ARRAY}, {"Ahd", ORC_INT16_LRG_ARRAY},
    {"Auhd", ORC_UINT16_LRG_ARRAY}, {"Ad", ORC_INT32_LRG_ARRAY}, {"Aud", ORC_UINT32_LRG_ARRAY},
    {"Ald", ORC_INT64_LRG_ARRAY}, {"Auld", ORC_UINT64_LRG_ARRAY}, {"Af", ORC_FLOAT_LRG_ARRAY},
    {"Alf", ORC_DOUBLE_LRG_ARRAY}, {"As", ORC_STRING_LRG_ARRAY},
};

int orc_fmt2type(const char *tok)
{
    for (size_t i = 0; i < sizeof fmt_table / sizeof fmt_table[0]; i++)
        if (strcmp(tok, fmt_table[i].tok) == 0) return fmt_table[i].type;
    return ORC_UNKNOWN;
}

static const char FMT_DELIM[] = " \t\n%";           /* src/Packet.C:611 */

/* next token of *pp; copies it into tok (cap bytes); returns 0 at end */
static int next_token(const char **pp, char *tok, size_t cap)
{
    const char *p = *pp + strspn(*pp, FMT_DELIM);
    if (*p == '\0') { *pp = p; return 0; }
    size_t len = strcspn(p, FMT_DELIM);
    size_t cp = len < cap - 1 ? len : cap - 1;
    memcpy(tok, p, cp); tok[cp] = '\0';
    *pp = p + len;
    return 1;
}

int orc_fmt_types(const char *fmt, int *types, int max)
{
    char tok[16];
    int n = 0;
    const char *p = fmt;
    while (next_token(&p, tok, sizeof tok)) {
        int t = orc_fmt2type(tok);
        if (t == ORC_UNKNOWN) return -1;
        if (n < max) types[n] = t;
        n++;
    }
    return n;
}

int orc_type_width(int t)
{
    switch (t) {
    case ORC_CHAR: case ORC_UCHAR: case ORC_CHAR_ARRAY: case ORC_UCHAR_ARRAY:
    case ORC_CHAR_LRG_ARRAY: case ORC_UCHAR_LRG_ARRAY:
    case ORC_STRING: case ORC_STRING_ARRAY: case ORC_STRING_LRG_ARRAY: return 1;
    case ORC_INT16: case ORC_UINT16: case ORC_INT16_ARRAY: case ORC_UINT16_ARRAY:
    case ORC_INT16_LRG_ARRAY: case ORC_UINT16_LRG_ARRAY: return 2;
    case ORC_INT32: case ORC_UINT32: case ORC_FLOAT: case ORC_INT32_ARRAY: case ORC_UINT32_ARRAY:
    case ORC_FLOAT_ARRAY: case ORC_INT32_LRG_ARRAY: case ORC_UINT32_LRG_ARRAY:
    case ORC_FLOAT_LRG_ARRAY: return 4;
    case ORC_INT64: case ORC_UINT64: case ORC_DOUBLE: case ORC_INT64_ARRAY: case ORC_UINT64_ARRAY:
    case ORC_DOUBLE_ARRAY: case ORC_INT64_LRG_ARRAY: case ORC_UINT64_LRG_ARRAY:
    case ORC_DOUBLE_LRG_ARRAY: return 8;
    }
    return 0;
}

int orc_type_kind(int t)
{
    switch (t) {
    case ORC_CHAR: case ORC_UCHAR: case ORC_INT16: case ORC_UINT16: case ORC_INT32:
    case ORC_UINT32: case ORC_INT64: case ORC_UINT64: case ORC_FLOAT: case ORC_DOUBLE: return 0;
    case ORC_STRING: return 2;
    case ORC_STRING_ARRAY: case ORC_STRING_LRG_ARRAY: return 3;
    case ORC_UNKNOWN: return -1;
    }
    return 1;
}

static int is_large(int t) { return t >= ORC_CHAR_LRG_ARRAY; }

/* ======================================================================= */
/* one packet (src/Packet.C:557-845)                                       */
/* ======================================================================= */
int32_t orc_packet_header(orc_stream *s, void *vp)               /* src/Packet.C:557-593 */
{
    orc_packet *pkt = (orc_packet *)vp;
    if (!orc_i32(s, &pkt->stream_id)) return 0;
    if (!orc_i32(s, &pkt->tag)) return 0;
    if (!orc_i32(s, &pkt->src_rank)) return 0;
    if (!orc_wrapstring(s, &pkt->fmt)) return 0;
    if (!orc_array(s, (void **)&pkt->dest, &pkt->ndest, UINT64_MAX, sizeof(uint32_t), orc_i32)) return 0;
    if (!orc_char(s, &pkt->byteorder)) return 0;
    return 1;
}

static void push_elem(orc_packet *pkt, orc_elem *e)
{
    if (pkt->nelems == pkt->cap) {
        pkt->cap = pkt->cap ? pkt->cap * 2 : 4;
        pkt->elems = (orc_elem **)realloc(pkt->elems, pkt->cap * sizeof *pkt->elems);
    }
    pkt->elems[pkt->nelems++] = e;
}

static void free_elem(orc_elem *e)                                /* src/DataElement.C:17-71 */
{
    if (e->destroy && e->val.p) {
        int k = orc_type_kind(e->type);
        if (k == 3) {
            char **arr = (char **)e->val.p;
            for (uint64_t i = 0; i < e->array_len; i++) free(arr[i]);
            free(arr);
        } else if (k == 1 || k == 2) {
            free(e->val.p);
        }
    }
    free(e);
}

int32_t orc_packet_data(orc_stream *s, void *vp)                  /* src/Packet.C:594-845 */
{
    orc_packet *pkt = (orc_packet *)vp;
    if (pkt->fmt == NULL) return 1;
    char tok[16];
    const char *p = pkt->fmt;
    unsigned i = 0;
    while (next_token(&p, tok, sizeof tok)) {      /* format re-parsed for every packet */
        orc_elem *e = NULL;
        if (s->op == ORC_ENCODE) {
            e = pkt->elems[i];
        } else {
            e = (orc_elem *)calloc(1, sizeof *e);  /* new DataElement per field */
            e->type = orc_fmt2type(tok);
            e->destroy = 1;
        }
        int32_t rc = 0;
        uint64_t maxn = is_large(e->type) ? UINT64_MAX : (uint64_t)INT32_MAX;
        switch (e->type) {
        case ORC_CHAR: case ORC_UCHAR:
            rc = orc_char(s, &e->val.c); break;
        case ORC_CHAR_ARRAY: case ORC_UCHAR_ARRAY:
        case ORC_CHAR_LRG_ARRAY: case ORC_UCHAR_LRG_ARRAY:      /* src/Packet.C:638-661 */
            if (s->op == ORC_DECODE) e->val.p = NULL;
            rc = orc_bytes(s, (char **)&e->val.p, &e->array_len, maxn); break;
        case ORC_INT16: case ORC_UINT16:
            rc = orc_i16(s, &e->val.hd); break;
        case ORC_INT16_ARRAY: case ORC_UINT16_ARRAY:
        case ORC_INT16_LRG_ARRAY: case ORC_UINT16_LRG_ARRAY:
            if (s->op == ORC_DECODE) e->val.p = NULL;
            rc = orc_array(s, &e->val.p, &e->array_len, maxn, 2, orc_i16); break;
        case ORC_INT32: case ORC_UINT32:
            rc = orc_i32(s, &e->val.d); break;
        case ORC_INT32_ARRAY: case ORC_UINT32_ARRAY:
        case ORC_INT32_LRG_ARRAY: case ORC_UINT32_LRG_ARRAY:
            if (s->op == ORC_DECODE) e->val.p = NULL;
            rc = orc_array(s, &e->val.p, &e->array_len, maxn, 4, orc_i32); break;
        case ORC_INT64: case ORC_UINT64:
            rc = orc_i64(s, &e->val.ld); break;
        case ORC_INT64_ARRAY: case ORC_UINT64_ARRAY:
        case ORC_INT64_LRG_ARRAY: case ORC_UINT64_LRG_ARRAY:
            if (s->op == ORC_DECODE) e->val.p = NULL;
            rc = orc_array(s, &e->val.p, &e->array_len, maxn, 8, orc_i64); break;
        case ORC_FLOAT:
            rc = orc_f32(s, &e->val.f); break;
        case ORC_DOUBLE:
            rc = orc_f64(s, &e->val.lf); break;
        case ORC_FLOAT_ARRAY: case ORC_FLOAT_LRG_ARRAY:
            if (s->op == ORC_DECODE) e->val.p = NULL;
            rc = orc_array(s, &e->val.p, &e->array_len, maxn, 4, orc_f32); break;
        case ORC_DOUBLE_ARRAY: case ORC_DOUBLE_LRG_ARRAY:
            if (s->op == ORC_DECODE) e->val.p = NULL;
            rc = orc_array(s, &e->val.p, &e->array_len, maxn, 8, orc_f64); break;
        case ORC_STRING_ARRAY: case ORC_STRING_LRG_ARRAY:
            if (s->op == ORC_DECODE) e->val.p = NULL;
            rc = orc_array(s, &e->val.p, &e->array_len, maxn, sizeof(char *), orc_wrapstring); break;
        case ORC_STRING: {
            char *before = s->cur;
            if (s->op == ORC_DECODE) e->val.p = NULL;
            rc = orc_wrapstring(s, &e->val.p);
            /* remember strlen+1 as sent, so that the copy-out below moves exactly the
             * bytes that travelled even when they are not a well-formed C string */
            if (rc && s->op == ORC_DECODE) e->array_len = (uint64_t)(s->cur - before) - 4;
            break;
        }
        default:
            rc = 0; break;
        }
        if (!rc) {
            if (s->op == ORC_DECODE) free_elem(e);   /* the reference leaks it, src/Packet.C:829-834 */
            return 0;
        }
        if (s->op == ORC_DECODE) push_elem(pkt, e);
        i++;
    }
    return 1;
}

static void packet_clear(orc_packet *pkt)
{
    for (unsigned i = 0; i < pkt->nelems; i++) free_elem(pkt->elems[i]);
    free(pkt->elems);
    pkt->elems = NULL; pkt->nelems = pkt->cap = 0;
}

uint64_t orc_header_encode(uint32_t stream_id, int32_t tag, uint32_t src_rank, const char *fmt,
                           const uint32_t *dest, uint64_t ndest, char *out, uint64_t cap)
{                                                                   /* src/Packet.C:20-52 */
    orc_packet pkt;
    memset(&pkt, 0, sizeof pkt);
    pkt.stream_id = stream_id; pkt.tag = tag; pkt.src_rank = src_rank;
    pkt.fmt = (char *)fmt; pkt.dest = (uint32_t *)dest; pkt.ndest = ndest;
    pkt.byteorder = (char)orc_local_bo();
    uint64_t len = 0;
    if (!orc_sizeof(orc_packet_header, &pkt, &len) || len > cap) return 0;
    orc_stream s;
    orc_mem_create(&s, out, len, ORC_ENCODE, orc_local_bo());
    if (!orc_packet_header(&s, &pkt)) return 0;
    return len;
}

int32_t orc_header_decode(const char *hdr, uint64_t hdr_len, uint32_t *stream_id, int32_t *tag,
                          uint32_t *src_rank, char *fmt_out, uint64_t fmt_cap,
                          uint32_t *dest_out, uint64_t dest_cap, uint64_t *ndest, int *byteorder)
{                                                                   /* src/Packet.C:87-104 */
    if (hdr_len == 0) return 0;
    orc_packet pkt;
    memset(&pkt, 0, sizeof pkt);
    orc_stream s;
    /* byte order of the header is its own last byte (src/Packet.C:96) */
    orc_mem_create(&s, (char *)hdr, hdr_len, ORC_DECODE, hdr[hdr_len - 1]);
    int32_t rc = orc_packet_header(&s, &pkt);
    if (rc) {
        *stream_id = pkt.stream_id; *tag = pkt.tag; *src_rank = pkt.src_rank;
        *byteorder = pkt.byteorder; *ndest = pkt.ndest;
        if (fmt_out && fmt_cap) {
            size_t l = strlen(pkt.fmt);
            if (l >= fmt_cap) l = fmt_cap - 1;
            memcpy(fmt_out, pkt.fmt, l); fmt_out[l] = '\0';
        }
        for (uint64_t i = 0; i < pkt.ndest && i < dest_cap; i++) dest_out[i] = pkt.dest[i];
    }
    free(pkt.fmt);
    free(pkt.dest);
    return rc;
}

/* ======================================================================= */
/* waves: columns <-> per-packet element lists                             */
/* ======================================================================= */
/* Borrow column storage for packet i, as ArgList2DataElementArray borrows the
 * caller's arrays (src/Packet.C:847-954). */
static int32_t build_elems(orc_packet *pkt, const char *fmt, const orc_column *cols, uint64_t i)
{
    char tok[16];
    const char *p = fmt;
    int f = 0;
    while (next_token(&p, tok, sizeof tok)) {
        orc_elem *e = (orc_elem *)calloc(1, sizeof *e);
        e->type = orc_fmt2type(tok);
        const orc_column *c = &cols[f];
        int w = orc_type_width(e->type);
        switch (orc_type_kind(e->type)) {
        case 0:
            memcpy(&e->val, (const char *)c->values + i * (uint64_t)w, (size_t)w);
            break;
        case 1:
            e->array_len = c->counts[i];
            e->val.p = (char *)c->values + c->offs[i] * (uint64_t)w;
            break;
        case 2:
            /* a string without a single byte (not even the NUL) stands for the NULL pointer,
             * which pdr_string refuses to encode (src/pdr.c:279-282) */
            e->val.p = c->counts[i] ? (char *)c->values + c->offs[i] : NULL;
            e->array_len = c->counts[i] ? c->counts[i] - 1 : 0;
            break;
        case 3: {
            uint64_t ns = c->counts[i], first = c->offs[i];
            char **arr = (char **)malloc((ns ? ns : 1) * sizeof(char *));
            for (uint64_t k = 0; k < ns; k++)
                arr[k] = (char *)c->values + c->sub_offs[first + k];
            e->val.p = arr; e->array_len = ns;
            break;
        }
        default:
            free(e);
            return 0;
        }
        push_elem(pkt, e);
        f++;
    }
    return 1;
}

static void drop_borrowed(orc_packet *pkt)
{
    for (unsigned k = 0; k < pkt->nelems; k++) {
        if (orc_type_kind(pkt->elems[k]->type) == 3) free(pkt->elems[k]->val.p);
        free(pkt->elems[k]);
    }
    free(pkt->elems);
    pkt->elems = NULL; pkt->nelems = pkt->cap = 0;
}

int32_t orc_wave_sizeof(const char *fmt, const orc_column *cols, uint64_t n, uint64_t *lens)
{
    for (uint64_t i = 0; i < n; i++) {
        orc_packet pkt;
        memset(&pkt, 0, sizeof pkt);
        pkt.fmt = (char *)fmt;
        if (!build_elems(&pkt, fmt, cols, i)) { drop_borrowed(&pkt); return 0; }
        int32_t rc = orc_sizeof(orc_packet_data, &pkt, &lens[i]);
        drop_borrowed(&pkt);
        if (!rc) return 0;
    }
    return 1;
}

int32_t orc_wave_encode(const char *fmt, const orc_column *cols, uint64_t n,
                        const uint64_t *offsets, char *arena, uint64_t *lens)
{                                                                   /* src/Packet.C:54-85 per packet */
    for (uint64_t i = 0; i < n; i++) {
        orc_packet pkt;
        memset(&pkt, 0, sizeof pkt);
        pkt.fmt = (char *)fmt;
        if (!build_elems(&pkt, fmt, cols, i)) { drop_borrowed(&pkt); return 0; }
        uint64_t len = 0;
        int32_t rc = orc_sizeof(orc_packet_data, &pkt, &len);       /* sizing pass */
        if (rc && len) {
            orc_stream s;
            orc_mem_create(&s, arena + offsets[i], len, ORC_ENCODE, orc_local_bo());
            rc = orc_packet_data(&s, &pkt);                         /* encode pass */
        }
        drop_borrowed(&pkt);
        if (!rc) return 0;
        if (lens) lens[i] = len;
    }
    return 1;
}

/* orders (optional): the byte order each packet's message announced; every packet is decoded
 * with its own (src/Packet.C:118 passes the packet's _byteorder to pdrmem_create). */
static uint64_t wave_decode(const char *fmt, const char *arena, const uint64_t *offsets,
                            const uint64_t *lens, uint64_t n, int sender_bo, const uint8_t *orders,
                            orc_column *cols, int32_t *ok)
{                                                                   /* src/Packet.C:106-133 per packet */
    int types[64];
    int nf = orc_fmt_types(fmt, types, 64);
    if (nf < 0 || nf > 64) return 0;
    uint64_t good = 0;
    for (int f = 0; f < nf; f++) {
        if (orc_type_kind(types[f]) != 0) cols[f].offs[0] = 0;
        if (orc_type_kind(types[f]) == 3) cols[f].sub_offs[0] = 0;
    }
    for (uint64_t i = 0; i < n; i++) {
        orc_packet pkt;
        memset(&pkt, 0, sizeof pkt);
        pkt.fmt = (char *)fmt;
        int32_t rc = 1;
        if (lens[i] != 0) {
            orc_stream s;
            orc_mem_create(&s, (char *)arena + offsets[i], lens[i], ORC_DECODE, orders ? orders[i] : sender_bo);
            rc = orc_packet_data(&s, &pkt);
        } else if (nf != 0) {
            rc = 0;   /* a non-empty format cannot have an empty body */
        }
        if (ok) ok[i] = rc;
        good += (uint64_t)(rc != 0);
        /* copy out, as unpack's DataElementArray2ArgList does (src/Packet.C:1065-1431) */
        for (int f = 0; f < nf; f++) {
            orc_column *c = &cols[f];
            int w = orc_type_width(types[f]);
            const orc_elem *e = (unsigned)f < pkt.nelems ? pkt.elems[f] : NULL;
            switch (orc_type_kind(types[f])) {
            case 0:
                if (e) memcpy((char *)c->values + i * (uint64_t)w, &e->val, (size_t)w);
                else   memset((char *)c->values + i * (uint64_t)w, 0, (size_t)w);
                break;
            case 1: {
                uint64_t cnt = e ? e->array_len : 0;
                c->counts[i] = cnt;
                if (cnt) memcpy((char *)c->values + c->offs[i] * (uint64_t)w, e->val.p, (size_t)(cnt * (uint64_t)w));
                c->offs[i + 1] = c->offs[i] + cnt;
                break;
            }
            case 2: {
                uint64_t cnt = e ? e->array_len : 0;
                c->counts[i] = cnt;
                if (cnt) memcpy((char *)c->values + c->offs[i], e->val.p, (size_t)cnt);
                c->offs[i + 1] = c->offs[i] + cnt;
                break;
            }
            case 3: {
                uint64_t ns = e ? e->array_len : 0, first = c->offs[i];
                c->counts[i] = ns;
                for (uint64_t k = 0; k < ns; k++) {
                    const char *str = ((char **)e->val.p)[k];
                    uint64_t l = strlen(str) + 1;
                    c->sub_counts[first + k] = l;
                    memcpy((char *)c->values + c->sub_offs[first + k], str, (size_t)l);
                    c->sub_offs[first + k + 1] = c->sub_offs[first + k] + l;
                }
                c->offs[i + 1] = first + ns;
                break;
            }
            }
        }
        packet_clear(&pkt);
    }
    return good;
}

uint64_t orc_wave_decode(const char *fmt, const char *arena, const uint64_t *offsets,
                         const uint64_t *lens, uint64_t n, int sender_bo,
                         orc_column *cols, int32_t *ok)
{
    return wave_decode(fmt, arena, offsets, lens, n, sender_bo, NULL, cols, ok);
}

uint64_t orc_wave_decode_orders(const char *fmt, const char *arena, const uint64_t *offsets,
                                const uint64_t *lens, uint64_t n, const uint8_t *orders,
                                orc_column *cols, int32_t *ok)
{
    return wave_decode(fmt, arena, offsets, lens, n, 0, orders, cols, ok);
}

/* ======================================================================= */
/* message framing (src/Message.C)                                         */
/* ======================================================================= */
uint64_t orc_message_build(uint32_t n, const char *hdr_arena, const uint64_t *hdr_offs,
                           const uint64_t *hdr_lens, const char *buf_arena, const uint64_t *buf_offs,
                           const uint64_t *buf_lens, char *out, uint64_t cap)
{                                                                   /* src/Message.C:250-327 */
    const uint64_t nbuf = 2ull * n;
    const uint64_t sizes_len = nbuf * sizeof(uint64_t) + 1;           /* "1 extra bytes overhead" */
    uint64_t total = 5 + sizes_len;
    for (uint32_t i = 0; i < n; i++) total += hdr_lens[i] + buf_lens[i];
    if (total > cap) return 0;
    uint64_t *sizes = (uint64_t *)malloc((nbuf ? nbuf : 1) * sizeof(uint64_t));
    for (uint32_t i = 0; i < n; i++) { sizes[2 * i] = hdr_lens[i]; sizes[2 * i + 1] = buf_lens[i]; }
    orc_stream s;
    uint32_t count = n;
    orc_mem_create(&s, out + 1, 4, ORC_ENCODE, orc_local_bo());      /* packet count, :308-312 */
    orc_i32(&s, &count);
    out[0] = (char)orc_local_bo();
    orc_mem_create(&s, out + 5, sizes_len, ORC_ENCODE, orc_local_bo());   /* packet sizes, :316-322 */
    int32_t rc = orc_vector(&s, (char *)sizes, nbuf, sizeof(uint64_t), orc_i64);
    free(sizes);
    if (!rc) return 0;
    char *p = out + 5 + sizes_len;                                    /* the writev gather, :294-299 */
    for (uint32_t i = 0; i < n; i++) {
        memcpy(p, hdr_arena + hdr_offs[i], (size_t)hdr_lens[i]); p += hdr_lens[i];
        memcpy(p, buf_arena + buf_offs[i], (size_t)buf_lens[i]); p += buf_lens[i];
    }
    return total;
}

int32_t orc_message_parse(const char *msg, uint64_t msg_len, uint32_t *n, uint64_t *hdr_offs,
                          uint64_t *hdr_lens, uint64_t *buf_offs, uint64_t *buf_lens, uint32_t cap_n)
{                                                                   /* src/Message.C:71-143 */
    if (msg_len < 5) return 0;
    orc_stream s;
    uint32_t count = 0;
    orc_mem_create(&s, (char *)msg + 1, 4, ORC_DECODE, msg[0]);       /* count: the sender's flag, :78 */
    if (!orc_i32(&s, &count)) return 0;
    *n = count;
    if (count > cap_n) return 0;
    const uint64_t nbuf = 2ull * count, sizes_len = nbuf * sizeof(uint64_t) + 1;
    if (msg_len < 5 + sizes_len) return 0;
    uint64_t *sizes = (uint64_t *)malloc((nbuf ? nbuf : 1) * sizeof(uint64_t));
    orc_mem_create(&s, (char *)msg + 5, sizes_len, ORC_DECODE, orc_local_bo());   /* LOCAL order, :117 */
    int32_t rc = orc_vector(&s, (char *)sizes, nbuf, sizeof(uint64_t), orc_i64);
    uint64_t at = 5 + sizes_len;
    for (uint32_t i = 0; rc && i < count; i++) {
        hdr_offs[i] = at; hdr_lens[i] = sizes[2 * i];     at += sizes[2 * i];
        buf_offs[i] = at; buf_lens[i] = sizes[2 * i + 1]; at += sizes[2 * i + 1];
    }
    free(sizes);
    return rc && at <= msg_len;
}

/* ======================================================================= */
/* filters (src/FilterDefinitions.C)                                       */
/* ======================================================================= */
typedef union { char c; unsigned char uc; int16_t hd; uint16_t uhd; int32_t d; uint32_t ud;
                int64_t ld; uint64_t uld; float f; double lf; char raw[8]; } acc_t;

#define TYPE_SWITCH(BODY)                                                      \
    switch (type) {                                                            \
    case ORC_CHAR:   { typedef char T;          BODY(c)   } break;             \
    case ORC_UCHAR:  { typedef unsigned char T; BODY(uc)  } break;             \
    case ORC_INT16:  { typedef int16_t T;       BODY(hd)  } break;             \
    case ORC_UINT16: { typedef uint16_t T;      BODY(uhd) } break;             \
    case ORC_INT32:  { typedef int32_t T;       BODY(d)   } break;             \
    case ORC_UINT32: { typedef uint32_t T;      BODY(ud)  } break;             \
    case ORC_INT64:  { typedef int64_t T;       BODY(ld)  } break;             \
    case ORC_UINT64: { typedef uint64_t T;      BODY(uld) } break;             \
    case ORC_FLOAT:  { typedef float T;         BODY(f)   } break;             \
    case ORC_DOUBLE: { typedef double T;        BODY(lf)  } break;             \
    default: return 0;                                                         \
    }

/* out = in1 + in2 in the element type (src/FilterDefinitions.C:1925-1964) */
static int f_sum(const acc_t *a, const acc_t *b, acc_t *out, int type)
{
#define BODY(m) out->m = (T)(a->m + b->m);
    TYPE_SWITCH(BODY)
#undef BODY
    return 1;
}
/* out = (in1 < in2) ? in1 : in2   (src/FilterDefinitions.C:2042-2088) */
static int f_min(const acc_t *a, const acc_t *b, acc_t *out, int type)
{
#define BODY(m) out->m = (T)((a->m < b->m) ? a->m : b->m);
    TYPE_SWITCH(BODY)
#undef BODY
    return 1;
}
/* out = (in1 > in2) ? in1 : in2   (src/FilterDefinitions.C:2090-2136) */
static int f_max(const acc_t *a, const acc_t *b, acc_t *out, int type)
{
#define BODY(m) out->m = (T)((a->m > b->m) ? a->m : b->m);
    TYPE_SWITCH(BODY)
#undef BODY
    return 1;
}
/* out = in1 * in2, count cast as the reference casts it (src/FilterDefinitions.C:1966-2002) */
static int f_mult(const acc_t *a, int n, acc_t *out, int type)
{
    switch (type) {
    case ORC_CHAR:   out->c   = (char)(a->c * n); break;
    case ORC_UCHAR:  out->uc  = (unsigned char)(a->uc * n); break;
    case ORC_INT16:  out->hd  = (int16_t)(a->hd * n); break;
    case ORC_UINT16: out->uhd = (uint16_t)(a->uhd * n); break;
    case ORC_INT32:  out->d   = a->d * n; break;
    case ORC_UINT32: out->ud  = a->ud * n; break;
    case ORC_INT64:  out->ld  = a->ld * (int64_t)n; break;
    case ORC_UINT64: out->uld = a->uld * (int64_t)n; break;
    case ORC_FLOAT:  out->f   = a->f * (float)n; break;
    case ORC_DOUBLE: out->lf  = a->lf * (double)n; break;
    default: return 0;
    }
    return 1;
}
/* out = in1 / in2 (src/FilterDefinitions.C:2004-2040); integer division truncates toward zero */
static int f_div(const acc_t *a, int n, acc_t *out, int type)
{
    switch (type) {
    case ORC_CHAR:   out->c   = (char)(a->c / n); break;
    case ORC_UCHAR:  out->uc  = (unsigned char)(a->uc / n); break;
    case ORC_INT16:  out->hd  = (int16_t)(a->hd / n); break;
    case ORC_UINT16: out->uhd = (uint16_t)(a->uhd / n); break;
    case ORC_INT32:  out->d   = a->d / n; break;
    case ORC_UINT32: out->ud  = a->ud / n; break;
    case ORC_INT64:  out->ld  = a->ld / (int64_t)n; break;
    case ORC_UINT64: out->uld = a->uld / (int64_t)n; break;
    case ORC_FLOAT:  out->f   = a->f / (float)n; break;
    case ORC_DOUBLE: out->lf  = a->lf / (double)n; break;
    default: return 0;
    }
    return 1;
}

/* One application of tfilter_Sum / tfilter_Min / tfilter_Max / tfilter_Avg to
 * m packets given as (value, count) pairs. */
static int32_t fold_once(int op, int type, const acc_t *v, const int32_t *cnt, uint64_t m,
                         acc_t *out, int32_t *out_cnt)
{
    acc_t res, prod;
    memset(&res, 0, sizeof res);                 /* memset(result, 0, 8) */
    memset(&prod, 0, sizeof prod);
    if (m == 0) return 0;
    switch (op) {
    case ORC_SUM:                                /* src/FilterDefinitions.C:100-149 */
        for (uint64_t i = 0; i < m; i++) if (!f_sum(&res, &v[i], &res, type)) return 0;
        break;
    case ORC_MAX:                                /* src/FilterDefinitions.C:237-286 */
        for (uint64_t i = 0; i < m; i++)
            if (!(i == 0 ? f_max(&v[0], &v[0], &res, type) : f_max(&res, &v[i], &res, type))) return 0;
        break;
    case ORC_MIN:                                /* src/FilterDefinitions.C:374-424 */
        for (uint64_t i = 0; i < m; i++)
            if (!(i == 0 ? f_min(&v[0], &v[0], &res, type) : f_min(&res, &v[i], &res, type))) return 0;
        break;
    case ORC_AVG: {                              /* src/FilterDefinitions.C:514-561 */
        int nres = 0;
        for (uint64_t i = 0; i < m; i++) {
            if (!f_mult(&v[i], cnt[i], &prod, type)) return 0;
            f_sum(&res, &prod, &res, type);
            nres += cnt[i];
        }
        if (nres == 0) return 0;                 /* division by zero: undefined in the reference */
        f_div(&res, nres, &res, type);
        *out_cnt = nres;
        break;
    }
    case ORC_OR:                                 /* aggr_val |= cval, Examples/FaultRecovery/FaultRecoveryFilter.C:112-122 */
        if (type == ORC_FLOAT || type == ORC_DOUBLE) return 0;
        for (uint64_t i = 0; i < m; i++) {
            uint64_t a = 0, b = 0;
            memcpy(&a, &res, (size_t)orc_type_width(type));
            memcpy(&b, &v[i], (size_t)orc_type_width(type));
            a |= b;
            memcpy(&res, &a, (size_t)orc_type_width(type));
        }
        break;
    case ORC_AND:                                /* std::bitset<N>(v): v & mask */
        if (type == ORC_FLOAT || type == ORC_DOUBLE) return 0;
        res = v[0];
        for (uint64_t i = 1; i < m; i++) {
            uint64_t a = 0, b = 0;
            memcpy(&a, &res, (size_t)orc_type_width(type));
            memcpy(&b, &v[i], (size_t)orc_type_width(type));
            a &= b;
            memcpy(&res, &a, (size_t)orc_type_width(type));
        }
        break;
    default:
        return 0;
    }
    *out = res;
    return 1;
}

static void load_vals(int type, const void *vals, uint64_t n, acc_t *dst)
{
    int w = orc_type_width(type);
    for (uint64_t i = 0; i < n; i++) {
        memset(&dst[i], 0, sizeof dst[i]);
        memcpy(&dst[i], (const char *)vals + i * (uint64_t)w, (size_t)w);
    }
}

int32_t orc_filter_flat(int op, int type, const void *vals, const int32_t *cnt, uint64_t n,
                        void *out_val, int32_t *out_cnt)
{
    if (n == 0 || orc_type_kind(type) != 0) return 0;
    acc_t *v = (acc_t *)malloc(n * sizeof *v);
    load_vals(type, vals, n, v);
    acc_t res; int32_t rc_cnt = 0;
    int32_t rc = fold_once(op, type, v, cnt, n, &res, &rc_cnt);
    free(v);
    if (!rc) return 0;
    memcpy(out_val, &res, (size_t)orc_type_width(type));
    if (out_cnt) *out_cnt = rc_cnt;
    return 1;
}

int32_t orc_filter_tree(int op, int type, const void *vals, const int32_t *cnt, uint64_t n,
                        const uint32_t *fanin, int nlevels, void *out_val, int32_t *out_cnt)
{
    if (n == 0 || orc_type_kind(type) != 0) return 0;
    acc_t *v = (acc_t *)malloc(n * sizeof *v);
    int32_t *c = (int32_t *)malloc(n * sizeof *c);
    char *present = (char *)malloc(n);
    load_vals(type, vals, n, v);
    for (uint64_t i = 0; i < n; i++) { c[i] = cnt ? cnt[i] : 1; present[i] = 1; }
    uint64_t m = n;
    acc_t *gv = (acc_t *)malloc(n * sizeof *gv);
    int32_t *gc = (int32_t *)malloc(n * sizeof *gc);
    for (int L = 0; L < nlevels && m > 0; L++) {
        uint64_t f = fanin[L] ? fanin[L] : m;
        uint64_t groups = (m + f - 1) / f;
        for (uint64_t g = 0; g < groups; g++) {
            uint64_t lo = g * f, hi = lo + f < m ? lo + f : m, k = 0;
            for (uint64_t j = lo; j < hi; j++)
                if (present[j]) { gv[k] = v[j]; gc[k] = c[j]; k++; }
            acc_t r; int32_t rc_cnt = 0;
            memset(&r, 0, sizeof r);
            int32_t okg = k ? fold_once(op, type, gv, gc, k, &r, &rc_cnt) : 0;
            /* writes land at index g <= lo: never ahead of the reads of later groups */
            v[g] = r; c[g] = rc_cnt; present[g] = (char)(okg != 0);
        }
        m = groups;
    }
    int32_t rc = (m == 1 && present[0]);
    if (rc) {
        memcpy(out_val, &v[0], (size_t)orc_type_width(type));
        if (out_cnt) *out_cnt = c[0];
    }
    free(v); free(c); free(present); free(gv); free(gc);
    return rc;
}

/* ======================================================================= */
/* tfilter_PerfData (src/FilterDefinitions.C:904-1239)                      */
/* ======================================================================= */
typedef union { uint64_t u; int64_t i; double d; } perf_t;                 /* perfdata_t */

int32_t orc_filter_perfdata(int op, int type, const int32_t *rank_vals, const uint64_t *rank_counts,
                            const uint64_t *rank_offs, const void *data_vals,
                            const uint64_t *data_counts, const uint64_t *data_offs, uint64_t n,
                            void *out_data, uint64_t out_cap, uint64_t *out_len, int32_t *out_rank)
{
    if (type != ORC_UINT64 && type != ORC_INT64 && type != ORC_DOUBLE) return 0;
    if (op < ORC_SUM || op > ORC_AVG) return 0;
    perf_t init;
    memset(&init, op == ORC_MIN ? 0xFF : 0, sizeof init);                  /* :961-965 */
    perf_t *aggr = NULL;
    uint64_t sz = 0;
    unsigned total_nranks = 0;
    uint64_t data_len = 0;
    for (uint64_t p = 0; p < n; p++) {                                      /* :984-1113 */
        data_len = data_counts[p];
        if (!data_len) continue;
        int rank = rank_counts[p] ? rank_vals[rank_offs[p]] : 0;
        int nrank;
        if (rank < 0) { nrank = -rank; total_nranks += (unsigned)nrank; }   /* :1012-1021 */
        else          { total_nranks++; nrank = 1; }
        if (sz < data_len) {                                                /* :1023-1025 */
            aggr = (perf_t *)realloc(aggr, (size_t)data_len * sizeof *aggr);
            for (uint64_t u = sz; u < data_len; u++) aggr[u] = init;
            sz = data_len;
        }
        const perf_t *x = (const perf_t *)data_vals + data_offs[p];
        for (uint64_t u = 0; u < data_len; u++) {
            perf_t *ag = &aggr[u];
            switch (op) {
            case ORC_SUM:                                                   /* sum(), :1027-1045 */
                if (type == ORC_UINT64) ag->u = ag->u + x[u].u;
                else if (type == ORC_INT64) ag->i = (int64_t)((uint64_t)ag->i + (uint64_t)x[u].i);
                else ag->d = ag->d + x[u].d;
                break;
            case ORC_MIN:                                                   /* mrn_min(), :1046-1064 */
                if (type == ORC_UINT64) ag->u = (ag->u < x[u].u) ? ag->u : x[u].u;
                else if (type == ORC_INT64) ag->i = (ag->i < x[u].i) ? ag->i : x[u].i;
                else ag->d = (ag->d < x[u].d) ? ag->d : x[u].d;
                break;
            case ORC_MAX:                                                   /* mrn_max(), :1065-1083 */
                if (type == ORC_UINT64) ag->u = (ag->u > x[u].u) ? ag->u : x[u].u;
                else if (type == ORC_INT64) ag->i = (ag->i > x[u].i) ? ag->i : x[u].i;
                else ag->d = (ag->d > x[u].d) ? ag->d : x[u].d;
                break;
            default:                                                        /* AVG, :1084-1111 */
                if (type == ORC_UINT64) { uint64_t v = x[u].u; v *= (uint64_t)nrank; ag->u += v; }
                else if (type == ORC_INT64) {
                    uint64_t v = (uint64_t)x[u].i * (uint64_t)(int64_t)nrank;
                    ag->i = (int64_t)((uint64_t)ag->i + v);
                } else { double v = x[u].d; v *= nrank; ag->d += v; }
                break;
            }
        }
    }
    if (data_len && op == ORC_AVG) {                /* the LAST packet's length, :1114-1132 */
        for (uint64_t u = 0; u < data_len; u++) {
            if (type == ORC_UINT64) aggr[u].u /= total_nranks;
            else if (type == ORC_INT64) aggr[u].i /= total_nranks;         /* unsigned -> int64 */
            else aggr[u].d /= total_nranks;
        }
    }
    int32_t rc = sz <= out_cap;
    if (rc) {
        memcpy(out_data, aggr, (size_t)sz * sizeof *aggr);                  /* :1205-1221 */
        *out_len = sz;
        *out_rank = 0 - (int)total_nranks;
    }
    free(aggr);
    return rc;
}

/* ======================================================================= */
/* tfilter_IntEqClass (src/FilterDefinitions.C:812-902)                     */
/* ======================================================================= */
/* The reference pours members into std::map<unsigned, std::vector<unsigned>> one push_back at a
 * time (:841-845): a class exists only once a member has been pushed under its value, so a class
 * entry with a member count of 0 leaves no trace.  Restated as the list of (value, member) pairs in
 * push order, ordered by value afterwards (the map's iteration order; push order inside a value). */
typedef struct { uint32_t val; uint64_t order; uint32_t member; } eq_pair;

static int eq_cmp(const void *a, const void *b)
{
    const eq_pair *x = (const eq_pair *)a, *y = (const eq_pair *)b;
    if (x->val != y->val) return x->val < y->val ? -1 : 1;
    return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}

int32_t orc_filter_int_eq_class(const uint32_t *vals, const uint64_t *val_counts, const uint64_t *val_offs,
                                const uint32_t *memcnts, const uint32_t *mems, const uint64_t *mem_offs,
                                uint64_t n, uint32_t *out_vals, uint32_t *out_memcnts, uint32_t *out_mems,
                                uint64_t *nclasses, uint64_t *nmems)
{
    uint64_t total = 0;
    for (uint64_t p = 0; p < n; p++)
        for (uint64_t j = 0; j < val_counts[p]; j++) total += memcnts[val_offs[p] + j];
    eq_pair *e = (eq_pair *)malloc((total ? total : 1) * sizeof *e);
    uint64_t k = 0;
    for (uint64_t p = 0; p < n; p++) {                     /* :824-849 */
        uint64_t cur = mem_offs[p];                        /* curClassMemIdx = 0: the packet's own mems array */
        for (uint64_t j = 0; j < val_counts[p]; j++) {
            uint64_t at = val_offs[p] + j;
            for (uint32_t m = 0; m < memcnts[at]; m++) {   /* classes[vals[j]].push_back(mems[cur + m]) */
                /* a packet whose member counts promise more than the wave's mems hold: the
                 * reference reads past its array; here such a member reads as 0 */
                e[k].val = vals[at]; e[k].order = k;
                e[k].member = cur + m < mem_offs[n] ? mems[cur + m] : 0;
                k++;
            }
            cur += memcnts[at];
        }
    }
    qsort(e, (size_t)total, sizeof *e, eq_cmp);
    uint64_t nc = 0;
    for (uint64_t i = 0; i < total; i++) {                 /* :852-875 */
        if (i == 0 || e[i].val != e[i - 1].val) { out_vals[nc] = e[i].val; out_memcnts[nc] = 0; nc++; }
        out_memcnts[nc - 1]++;
        out_mems[i] = e[i].member;
    }
    *nclasses = nc; *nmems = total;
    free(e);
    return 1;
}

/* ======================================================================= */
/* timing helper                                                           */
/* ======================================================================= */
double orc_bench_small_wave(uint64_t reps, uint64_t n, int32_t value, int32_t *sum)
{
    if (n == 0) return -1.0;
    int32_t *in = (int32_t *)malloc(n * sizeof *in), *dec = (int32_t *)malloc(n * sizeof *dec);
    int32_t *ok = (int32_t *)malloc(n * sizeof *ok), res = 0, cnt = 0;
    uint64_t *offs = (uint64_t *)malloc((n + 1) * sizeof *offs), *lens = (uint64_t *)malloc(n * sizeof *lens);
    uint64_t one_off[2] = { 0, 4 }, one_len[1];
    char *arena = (char *)malloc(4 * n), out[4] = { 0, 0, 0, 0 };
    double per_wave = -1.0;
    if (!in || !dec || !ok || !offs || !lens || !arena) goto done;
    for (uint64_t i = 0; i < n; i++) { in[i] = value; offs[i] = 4u * i; }
    offs[n] = 4 * n;
    {
        orc_column src = { ORC_INT32, in, NULL, NULL, NULL, NULL };
        if (!orc_wave_encode("%d", &src, n, offs, arena, lens)) goto done;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint64_t r = 0; r < reps; r++) {
        orc_column col = { ORC_INT32, dec, NULL, NULL, NULL, NULL };
        if (orc_wave_decode("%d", arena, offs, lens, n, orc_local_bo(), &col, ok) != n) goto done;
        if (!orc_filter_flat(ORC_SUM, ORC_INT32, dec, NULL, n, &res, &cnt)) goto done;
        orc_column o = { ORC_INT32, &res, NULL, NULL, NULL, NULL };
        if (!orc_wave_encode("%d", &o, 1, one_off, out, one_len)) goto done;
        __asm__ volatile("" : : "r"(out) : "memory");          /* the result is used */
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    memcpy(sum, out, 4);
    per_wave = ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec)) / (double)(reps ? reps : 1);
done:
    free(in); free(dec); free(ok); free(offs); free(lens); free(arena);
    return per_wave;
}

/* The left fold of tfilter_Sum over an already decoded float / double column with everything the
 * reference does per packet taken away (no shared_ptr, no lock, no type switch: sum() of
 * src/FilterDefinitions.C:1925-1964 inlined for one type): the dependent chain of additions and
 * nothing else, i.e. the fastest a host core can produce the reference's bits.  Seconds per fold. */
double orc_bench_chain(int is_double, const void *vals, uint64_t n, uint64_t reps, void *out)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (is_double) {
        const double *v = (const double *)vals;
        double r = 0;
        for (uint64_t k = 0; k < reps; k++) {
            r = 0;
            for (uint64_t i = 0; i < n; i++) r = r + v[i];
            __asm__ volatile("" : : "x"(r) : "memory");
        }
        memcpy(out, &r, 8);
    } else {
        const float *v = (const float *)vals;
        float r = 0;
        for (uint64_t k = 0; k < reps; k++) {
            r = 0;
            for (uint64_t i = 0; i < n; i++) r = r + v[i];
            __asm__ volatile("" : : "x"(r) : "memory");
        }
        memcpy(out, &r, 4);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec)) / (double)(reps ? reps : 1);
}
