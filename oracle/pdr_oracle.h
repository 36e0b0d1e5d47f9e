/*
 * pdr_oracle.h -- CPU restatement of MRNet's PDR packet marshalling path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP codec is compared
 * against (and the timed "cpu_baseline" leg of bench.py).  Nothing in the
 * product library links, loads or calls it.
 *
 * Every routine names the reference file:line whose behaviour it restates
 * (paths relative to the MRNet 5.0.1 source tree).
 *
 * Pinning: the wire bytes this file produces are checked against the byte
 * vectors captured from the compiled reference (tests/golden/) and the
 * known answers the reference's own tests hold (tests/test_basic_FE.C,
 * tests/test_arrays_FE.C, tests/test_NativeFilters_FE.C).
 */
#ifndef PDR_ORACLE_H
#define PDR_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/mrnet/DataElement.h:27-45 -- same order, same numeric values */
enum orc_type {
    ORC_UNKNOWN = 0, ORC_CHAR, ORC_UCHAR,
    ORC_CHAR_ARRAY, ORC_UCHAR_ARRAY,
    ORC_STRING, ORC_STRING_ARRAY,
    ORC_INT16, ORC_UINT16,
    ORC_INT16_ARRAY, ORC_UINT16_ARRAY,
    ORC_INT32, ORC_UINT32,
    ORC_INT32_ARRAY, ORC_UINT32_ARRAY,
    ORC_INT64, ORC_UINT64,
    ORC_INT64_ARRAY, ORC_UINT64_ARRAY,
    ORC_FLOAT, ORC_DOUBLE,
    ORC_FLOAT_ARRAY, ORC_DOUBLE_ARRAY,
    ORC_CHAR_LRG_ARRAY, ORC_UCHAR_LRG_ARRAY,
    ORC_STRING_LRG_ARRAY,
    ORC_INT16_LRG_ARRAY, ORC_UINT16_LRG_ARRAY,
    ORC_INT32_LRG_ARRAY, ORC_UINT32_LRG_ARRAY,
    ORC_INT64_LRG_ARRAY, ORC_UINT64_LRG_ARRAY,
    ORC_FLOAT_LRG_ARRAY, ORC_DOUBLE_LRG_ARRAY
};

enum orc_op { ORC_FREE = 0, ORC_ENCODE = 1, ORC_DECODE = 2 };   /* src/pdr.h:64-68 */
enum orc_bo { ORC_LE = 0, ORC_BE = 1 };                          /* src/pdr.h:70-73 */

/* ---- stream (src/pdr.h:83-110) ---------------------------------------- */
typedef struct orc_stream orc_stream;
struct orc_ops {
    int32_t (*put8)(orc_stream *, void *);   int32_t (*get8)(orc_stream *, void *);
    int32_t (*put16)(orc_stream *, void *);  int32_t (*get16)(orc_stream *, void *);
    int32_t (*put32)(orc_stream *, void *);  int32_t (*get32)(orc_stream *, void *);
    int32_t (*put64)(orc_stream *, void *);  int32_t (*get64)(orc_stream *, void *);
    int32_t (*putf)(orc_stream *, void *);   int32_t (*getf)(orc_stream *, void *);
    int32_t (*putd)(orc_stream *, void *);   int32_t (*getd)(orc_stream *, void *);
    int32_t (*putbytes)(orc_stream *, char *, uint64_t);
    int32_t (*getbytes)(orc_stream *, char *, uint64_t);
};
struct orc_stream {
    int op;
    const struct orc_ops *ops;
    char *cur, *base;
    uint64_t space;
};
typedef int32_t (*orc_proc)(orc_stream *, void *);

void    orc_mem_create(orc_stream *s, char *addr, uint64_t size, int op, int sender_bo);
int     orc_local_bo(void);
int32_t orc_sizeof(orc_proc fn, void *obj, uint64_t *size);

int32_t orc_char(orc_stream *, void *);   int32_t orc_i16(orc_stream *, void *);
int32_t orc_i32(orc_stream *, void *);    int32_t orc_i64(orc_stream *, void *);
int32_t orc_f32(orc_stream *, void *);    int32_t orc_f64(orc_stream *, void *);
int32_t orc_opaque(orc_stream *, char *, uint64_t);
int32_t orc_bytes(orc_stream *, char **, uint64_t *, uint64_t maxsize);
int32_t orc_string(orc_stream *, char **, uint32_t maxsize);
int32_t orc_wrapstring(orc_stream *, void *);
int32_t orc_array(orc_stream *, void **, uint64_t *, uint64_t maxsize, uint32_t elsize, orc_proc);
int32_t orc_vector(orc_stream *, char *, uint64_t nelem, uint64_t elsize, orc_proc);

void orc_byte_swap(char *out, const char *in, uint32_t nelems, uint32_t elsize);

/* ---- format strings --------------------------------------------------- */
int orc_fmt2type(const char *tok);
/* Tokenise on " \t\n%"; writes up to max type codes; returns token count, -1 on unknown token. */
int orc_fmt_types(const char *fmt, int *types, int max);
int orc_type_width(int type);      /* element width in bytes; strings -> 1 */
int orc_type_kind(int type);       /* 0 scalar, 1 array, 2 string, 3 string array */

/* ---- one packet -------------------------------------------------------- */
typedef union { char c; unsigned char uc; int16_t hd; uint16_t uhd; int32_t d; uint32_t ud;
                int64_t ld; uint64_t uld; float f; double lf; void *p; } orc_value;
typedef struct { orc_value val; int type; uint64_t array_len; int destroy; } orc_elem;

typedef struct {
    uint32_t stream_id; int32_t tag; uint32_t src_rank;
    char *fmt;
    uint32_t *dest; uint64_t ndest;
    char byteorder;
    orc_elem **elems; unsigned nelems, cap;
} orc_packet;

int32_t orc_packet_header(orc_stream *, void *pkt);
int32_t orc_packet_data(orc_stream *, void *pkt);

/* Encode a header into out (capacity cap); returns length or 0 on failure. */
uint64_t orc_header_encode(uint32_t stream_id, int32_t tag, uint32_t src_rank, const char *fmt,
                           const uint32_t *dest, uint64_t ndest, char *out, uint64_t cap);
/* Decode a header.  fmt_out must hold fmt_cap bytes.  Returns 1/0. */
int32_t orc_header_decode(const char *hdr, uint64_t hdr_len, uint32_t *stream_id, int32_t *tag,
                          uint32_t *src_rank, char *fmt_out, uint64_t fmt_cap,
                          uint32_t *dest_out, uint64_t dest_cap, uint64_t *ndest, int *byteorder);

/* ---- a wave of packets that share one format --------------------------- */
/*
 * Column form of field f over n packets:
 *   scalar        values = T[n]
 *   array/string  counts[n] (elements; a string counts strlen+1, NUL included),
 *                 offs[n+1] (exclusive scan of counts), values = packed elements
 *   string array  counts[n] = strings per packet, offs[n+1] over strings,
 *                 sub_counts[S] = strlen+1 of each string, sub_offs[S+1], values = bytes
 */
typedef struct {
    int type;
    void *values;
    uint64_t *counts;
    uint64_t *offs;
    uint64_t *sub_counts;
    uint64_t *sub_offs;
} orc_column;

/* Encoded data length of every packet (dry-run encode, per packet).  Returns 1/0. */
int32_t orc_wave_sizeof(const char *fmt, const orc_column *cols, uint64_t n, uint64_t *lens);
/* Encode packet i at arena+offsets[i]; lens[i] receives its length.  Returns 1/0. */
int32_t orc_wave_encode(const char *fmt, const orc_column *cols, uint64_t n,
                        const uint64_t *offsets, char *arena, uint64_t *lens);
/*
 * Decode packet i from arena+offsets[i], length lens[i], sender byte order sender_bo.
 * Columns must be pre-sized by the caller; offs[] are written as packets are appended.
 * ok[i] = 1/0 per packet; a failed packet contributes zero scalars / zero counts for the
 * failed field and every later one.  Returns number of packets that decoded.
 */
uint64_t orc_wave_decode(const char *fmt, const char *arena, const uint64_t *offsets,
                         const uint64_t *lens, uint64_t n, int sender_bo,
                         orc_column *cols, int32_t *ok);
/* The same with one byte order per packet (orders[i], as each packet's message announced it). */
uint64_t orc_wave_decode_orders(const char *fmt, const char *arena, const uint64_t *offsets,
                                const uint64_t *lens, uint64_t n, const uint8_t *orders,
                                orc_column *cols, int32_t *ok);

/* ---- message framing (src/Message.C:201-335 send, :48-199 recv) -------------------- */
/* Gathers n (header, buffer) pairs into one message the way Message::send hands them to writev.
 * The spare byte behind the size vector is left untouched, as in the reference.  Returns the
 * message length, 0 on failure. */
uint64_t orc_message_build(uint32_t n, const char *hdr_arena, const uint64_t *hdr_offs,
                           const uint64_t *hdr_lens, const char *buf_arena, const uint64_t *buf_offs,
                           const uint64_t *buf_lens, char *out, uint64_t cap);
/* Splits a message: positions (inside msg) and lengths of every header and buffer. */
int32_t orc_message_parse(const char *msg, uint64_t msg_len, uint32_t *n, uint64_t *hdr_offs,
                          uint64_t *hdr_lens, uint64_t *buf_offs, uint64_t *buf_lens, uint32_t cap_n);

/* ---- filters (src/FilterDefinitions.C:90-645, 1925-2136) ---------------- */
enum orc_filter_op { ORC_SUM = 0, ORC_MIN = 1, ORC_MAX = 2, ORC_AVG = 3,
                     ORC_OR = 4 /* bitwise, integer types: the BitsetOr example filter */, ORC_AND = 5 };
/*
 * Flat left fold over n values of scalar `type` (cnt != NULL only for AVG).
 * out_val receives the result in the element type, out_cnt the summed count (AVG).
 * Returns 1, or 0 when n == 0 / bad type / AVG count sum is zero.
 */
int32_t orc_filter_flat(int op, int type, const void *vals, const int32_t *cnt, uint64_t n,
                        void *out_val, int32_t *out_cnt);
/*
 * The same fold applied level by level over a tree: level L folds groups of
 * fanin[L] consecutive nodes of the level below (fanin 0 = all remaining nodes).
 * Absent nodes are skipped; a group without nodes is absent.
 */
int32_t orc_filter_tree(int op, int type, const void *vals, const int32_t *cnt, uint64_t n,
                        const uint32_t *fanin, int nlevels, void *out_val, int32_t *out_cnt);

/* ---- array filters ----------------------------------------------------------------- */
/*
 * tfilter_PerfData, aggregations SUM / MIN / MAX / AVG (src/FilterDefinitions.C:904-1239), one
 * invocation over n packets "%ad %ad %a{uld,ld,lf}" given as columns: rank arrays (int32) and
 * data arrays of `type` (ORC_UINT64 / ORC_INT64 / ORC_DOUBLE).  Packets without data are skipped.
 * out_data receives aggr_results (out_len elements, at most out_cap), out_rank the single
 * element of the output rank array (-total_nranks).  Returns 1, 0 on bad arguments / overflow.
 * A packet with data but an empty rank array makes the reference dereference NULL; here its
 * rank reads as 0.
 */
int32_t orc_filter_perfdata(int op, int type, const int32_t *rank_vals, const uint64_t *rank_counts,
                            const uint64_t *rank_offs, const void *data_vals,
                            const uint64_t *data_counts, const uint64_t *data_offs, uint64_t n,
                            void *out_data, uint64_t out_cap, uint64_t *out_len, int32_t *out_rank);

/*
 * tfilter_IntEqClass (src/FilterDefinitions.C:812-902): columns of "%aud %aud %aud" packets in,
 * the three arrays of the output packet out (capacities: as many classes / members as came in).
 */
int32_t orc_filter_int_eq_class(const uint32_t *vals, const uint64_t *val_counts, const uint64_t *val_offs,
                                const uint32_t *memcnts, const uint32_t *mems, const uint64_t *mem_offs,
                                uint64_t n, uint32_t *out_vals, uint32_t *out_memcnts, uint32_t *out_mems,
                                uint64_t *nclasses, uint64_t *nmems);

/* ---- timing helper (bench.py's CPU baseline leg) ----------------------------------- */
/*
 * What one node of a reduction tree does per wave: n "%d" packets decoded, tfilter_Sum, the one
 * output packet encoded -- `reps` times in a C loop, so that no foreign-function call is in the
 * figure.  n = 4 is the reference's microbenchmark (tests/microbench_FE.C:61,145 over
 * tests/topology_files/local-1x4.top).  Returns seconds per wave (negative on failure); *sum
 * receives the last result (n x value, wrapping).
 */
double orc_bench_small_wave(uint64_t reps, uint64_t n, int32_t value, int32_t *sum);
double orc_bench_chain(int is_double, const void *vals, uint64_t n, uint64_t reps, void *out);

#ifdef __cplusplus
}
#endif
#endif
