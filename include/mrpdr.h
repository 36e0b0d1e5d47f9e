/*
 * mrpdr.h -- C ABI of the MI355X wave-at-a-time PDR codec.
 *
 * This is the drop-in boundary for MRNet's packet marshalling path.  Where the
 * reference packs one packet at a time through Packet::encode_pdr_data /
 * decode_pdr_data and the pdr_* primitives, a caller hands this library a WAVE:
 * n packets that share one format string, as device-resident buffers.
 *
 *   wire side     arena (bytes) + per-packet offsets/lengths; the bytes of every
 *                 packet are exactly what src/pdr.c writes (no padding inside a
 *                 packet; sender's native byte order; receiver makes right)
 *   decoded side  one column per format token (struct mrpdr_column)
 *
 * Conventions
 *   - every function returns bool_t-style int32_t: 1 = TRUE, 0 = FALSE
 *     (src/pdr.h:40-58); on FALSE mrpdr_last_error() describes the reason
 *   - all data pointers are DEVICE pointers unless the name ends in _host
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on
 *     it unless a *_host output is requested (then the call synchronises it)
 *   - nothing is thrown across this boundary; no torch/C++ types appear in it
 *   - a mrpdr_ctx owns a small device workspace; calls using the same ctx must
 *     not overlap in time (use one ctx per calling thread / stream)
 *   - a mrpdr_ctx belongs to the device it was created on; that device must be the
 *     calling thread's current HIP device (hipSetDevice) whenever the ctx is used,
 *     otherwise the call returns FALSE
 *   - the device is little-endian: mrpdr_local_byteorder() == 0
 */
#ifndef MRPDR_H
#define MRPDR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DataType codes -- identical to include/mrnet/DataElement.h:27-45 */
enum mrpdr_type {
    MRPDR_UNKNOWN_T = 0, MRPDR_CHAR_T, MRPDR_UCHAR_T,
    MRPDR_CHAR_ARRAY_T, MRPDR_UCHAR_ARRAY_T,
    MRPDR_STRING_T, MRPDR_STRING_ARRAY_T,
    MRPDR_INT16_T, MRPDR_UINT16_T,
    MRPDR_INT16_ARRAY_T, MRPDR_UINT16_ARRAY_T,
    MRPDR_INT32_T, MRPDR_UINT32_T,
    MRPDR_INT32_ARRAY_T, MRPDR_UINT32_ARRAY_T,
    MRPDR_INT64_T, MRPDR_UINT64_T,
    MRPDR_INT64_ARRAY_T, MRPDR_UINT64_ARRAY_T,
    MRPDR_FLOAT_T, MRPDR_DOUBLE_T,
    MRPDR_FLOAT_ARRAY_T, MRPDR_DOUBLE_ARRAY_T,
    MRPDR_CHAR_LRG_ARRAY_T, MRPDR_UCHAR_LRG_ARRAY_T,
    MRPDR_STRING_LRG_ARRAY_T,
    MRPDR_INT16_LRG_ARRAY_T, MRPDR_UINT16_LRG_ARRAY_T,
    MRPDR_INT32_LRG_ARRAY_T, MRPDR_UINT32_LRG_ARRAY_T,
    MRPDR_INT64_LRG_ARRAY_T, MRPDR_UINT64_LRG_ARRAY_T,
    MRPDR_FLOAT_LRG_ARRAY_T, MRPDR_DOUBLE_LRG_ARRAY_T
};

/* pdr_byteorder, src/pdr.h:70-73 */
enum mrpdr_byteorder { MRPDR_LITTLEENDIAN = 0, MRPDR_BIGENDIAN = 1 };

/* field kinds */
enum mrpdr_kind { MRPDR_SCALAR = 0, MRPDR_ARRAY = 1, MRPDR_STRING = 2, MRPDR_STRING_ARRAY = 3 };

/* filter operators: TFILTER_SUM / MIN / MAX / AVG (src/FilterDefinitions.C:90-645) */
enum mrpdr_filter_op { MRPDR_SUM = 0, MRPDR_MIN = 1, MRPDR_MAX = 2, MRPDR_AVG = 3,
                       /* bitwise OR of integer values: the BitsetOr filter and the bit field of
                        * IntegerPercentiles (Examples/FaultRecovery/FaultRecoveryFilter.C:29-132) */
                       MRPDR_OR = 4,
                       /* bitwise AND (std::bitset<N> keeps N bits: AND with the mask) */
                       MRPDR_AND = 5 };

/*
 * Shape of the reduction a filter call applies to a wave.
 *   MRPDR_SHAPE_FLAT  one left fold over all n packets, i.e. exactly one call of the
 *                     reference's tfilter_* on the whole wave (exact for every type and
 *                     operator at any n).  How the bits are reached is the library's
 *                     business: integer SUM / MIN / MAX / OR / AND, integer AVG and float
 *                     MIN / MAX level by level; float and double SUM / AVG -- where every
 *                     addition rounds -- by the chain of additions itself, which waves
 *                     of 8,192 packets and more take a binade at a time with a scan over
 *                     the sum's integer grid instead of one dependent addition per packet
 *                     (with that work spread over the whole device wherever the caller's ctx,
 *                     or the no-allocation entry point's built-in array, has room for it)
 *   MRPDR_SHAPE_FLAT_SERIAL  the same result with float / double SUM and AVG computed by
 *                     the plain chain only, one addition after the other as the reference's
 *                     loop is written (for cross-checks; identical bits, slower on large
 *                     waves); every other operator and type as MRPDR_SHAPE_FLAT
 *   MRPDR_SHAPE_TREE  the same fold applied level by level over a fixed tree, the way
 *                     an MRNet tree applies it at every internal node:
 *                     fan-in 4, then 2,2,2,2,2,2, then 4, then all remaining nodes
 *                     (groups are consecutive packets; a group with no packet is
 *                     skipped).  Integer SUM and NaN-free MIN/MAX give the same bits
 *                     as FLAT; float SUM and every AVG depend on the shape.
 */
enum mrpdr_shape { MRPDR_SHAPE_FLAT = 0, MRPDR_SHAPE_TREE = 1, MRPDR_SHAPE_FLAT_SERIAL = 2 };

/* fan-ins of MRPDR_SHAPE_TREE, 0 = all remaining nodes */
#define MRPDR_TREE_LEVELS 9
#define MRPDR_TREE_FANINS { 4, 2, 2, 2, 2, 2, 2, 4, 0 }

typedef struct mrpdr_ctx mrpdr_ctx;
typedef struct mrpdr_fmt mrpdr_fmt;

/*
 * One decoded field of a wave (device memory).
 *   scalar        values = T[n]
 *   array         counts[n] elements per packet, offs[n+1] = exclusive scan of counts,
 *                 values = the packed elements (16-byte aligned base)
 *   string        as an array of bytes; a string counts strlen+1 (the NUL travels)
 *   string array  (%as / %As, src/Packet.C:794-813) two levels: counts[n] = strings per
 *                 packet, offs[n+1] over strings; string j (over the whole wave) occupies
 *                 values[sub_offs[j] .. sub_offs[j+1]) bytes, NUL included;
 *                 sub_counts[j] = strlen+1 (written on decode, not read on encode)
 * capacity = number of ELEMENTS values can hold, sub_capacity = number of strings
 * sub_counts / sub_offs (+1) can hold (both checked on decode).
 * Replaces DataElement / DataValue (include/mrnet/DataElement.h:13-108).
 */
typedef struct mrpdr_column {
    void     *values;
    uint64_t *counts;
    uint64_t *offs;
    uint64_t  capacity;
    uint64_t *sub_counts;      /* string arrays only, else NULL */
    uint64_t *sub_offs;
    uint64_t  sub_capacity;
} mrpdr_column;

/*
 * Where the wire bytes of the wave live.  Packet i occupies
 *   base + offsets[i] .. + lens[i]            when offsets != NULL
 *   base + i*stride   .. + lens[i]            when offsets == NULL (packed layout)
 * lens == NULL means "every packet has the format's fixed length" (scalar-only
 * formats).  The reference keeps each packet's buffer in its own malloc block
 * (src/Packet.C:69, src/Message.C:130-135); the arena replaces those blocks.
 */
typedef struct mrpdr_wire {
    uint8_t        *base;
    const uint64_t *offsets;
    const uint64_t *lens;
    uint64_t        stride;
    uint64_t        bytes;     /* size of the arena in bytes (bounds) */
} mrpdr_wire;

/* ---- library ----------------------------------------------------------- */
const char *mrpdr_version(void);
/* Text of the last failure on the calling thread. */
const char *mrpdr_last_error(void);
/* pdrmem_getbo(), src/pdr_mem.c:421-428 */
int32_t     mrpdr_local_byteorder(void);

/* Workspace for scans and filter partials on `device`; max_packets bounds n of
 * every later call with this ctx. */
mrpdr_ctx  *mrpdr_ctx_create(int32_t device, uint64_t max_packets);
void        mrpdr_ctx_destroy(mrpdr_ctx *ctx);
/* Device-side status, accumulated over the calls made with this ctx since it was last read:
 * 0 = fine, bit 0 = a decode (or array filter) ran out of column capacity (values or, for string
 * arrays, sub_counts / sub_offs), bit 1 = an array concatenation reached 2^32 elements,
 * bit 2 = mrpdr_wave_sizeof met a packet the reference refuses to encode: a '%s' field of
 * count 0 (no bytes at all, not even the NUL: the NULL string, src/pdr.c:279-282), or a '%a'
 * array of more than INT32_MAX elements (src/pdr.c:429; '%A' has no bound); that packet's
 * bytes are not a valid packet.  Reading clears it.  Synchronises `stream`. */
int32_t     mrpdr_ctx_status(mrpdr_ctx *ctx, void *stream, uint32_t *status_host);

/* Timing of the payload kernels, the counterpart of the reference's per-filter / per-packet
 * timers (PerfDataMgr, src/PerfDataEvent.C; Packet::start_Timer, src/Filter.C:60-112).
 * While enabled (on > 0), every on-th payload-kernel launch of each direction made through this
 * ctx is bracketed by HIP events on the caller's stream (at most 4096 timed launches between two
 * collects); an event pair costs a few microseconds of stream time, so a caller that also times
 * the whole stream samples (on = 5: every fifth launch).
 * collect synchronises the stream and returns, for [0] decode and [1] encode payload
 * kernels, the summed elapsed milliseconds and the number of launches, then resets. */
int32_t     mrpdr_ctx_perf_enable(mrpdr_ctx *ctx, int32_t on);
int32_t     mrpdr_ctx_perf_collect(mrpdr_ctx *ctx, void *stream, double *ms_sum_host,
                                   uint64_t *launches_host);
/* Calibration: the plain copy the payload kernels are measured against -- bytes (a multiple of
 * 16, both pointers 16-byte aligned) moved with 128-bit nontemporal loads and stores, one
 * 16 KiB chunk per workgroup, no codec logic.  `reps` launches are timed with HIP events on
 * `stream`; ms_avg_host receives the average launch time.  (No counterpart in the reference:
 * it tells what this device sustains for read+write streaming.) */
int32_t     mrpdr_ctx_perf_plain_copy(mrpdr_ctx *ctx, void *dst, const void *src, uint64_t bytes,
                                      int32_t reps, double *ms_avg_host, void *stream);

/* ---- format strings ------------------------------------------------------
 * Fmt2Type (src/DataElement.C:205-297) + the " \t\n%" tokenising of
 * Packet::pdr_packet_data (src/Packet.C:608-619), done once per format instead of
 * once per packet.  NULL on an unknown token.
 * Deliberate deviation: a format of more than 32 tokens is refused (NULL) where the reference's
 * tokenizer has no bound (src/Packet.C:608-619).  The field table travels to every kernel as an
 * argument, which fixes its size; the reference's own formats stop at eleven tokens
 * (tests/test_basic_FE.C:1345) and the formats of its own protocol have at most a handful (e.g. src/Network.C:1955). */
mrpdr_fmt  *mrpdr_fmt_compile(const char *fmt);
void        mrpdr_fmt_free(mrpdr_fmt *f);
int32_t     mrpdr_fmt_nfields(const mrpdr_fmt *f);
int32_t     mrpdr_fmt_field_type(const mrpdr_fmt *f, int32_t field);
int32_t     mrpdr_fmt_field_kind(const mrpdr_fmt *f, int32_t field);
int32_t     mrpdr_fmt_field_width(const mrpdr_fmt *f, int32_t field);
/* 1 when every token is a scalar: all packets then have the same length */
int32_t     mrpdr_fmt_is_fixed(const mrpdr_fmt *f);
/* bytes of scalars and length prefixes per packet (= data length for fixed formats) */
uint64_t    mrpdr_fmt_fixed_bytes(const mrpdr_fmt *f);
/* Arena phase that puts the first array/string payload of every packet on an `align`
 * boundary (0 for scalar-only formats); pass it to mrpdr_wave_sizeof. */
uint32_t    mrpdr_fmt_payload_phase(const mrpdr_fmt *f, uint32_t align);
const char *mrpdr_fmt_string(const mrpdr_fmt *f);

/* ---- sizing: pdr_sizeof for a whole wave -----------------------------------
 * Replaces the dry-run encode pdr_sizeof(Packet::pdr_packet_data) that
 * Packet::encode_pdr_data runs per packet (src/Packet.C:61, src/pdr_sizeof.c:120-140).
 * lens[i] = encoded data length of packet i.  The arena layout: packet i owns a slot of
 * round_up(phase + lens[i], align) bytes and starts `phase` bytes into it, so
 * offsets[i] = (sum of the slots before i) + phase and offsets[n] = the arena size
 * (align: 1 or a power of two up to 256; align 1 / phase 0 = packed).
 * Placement is free because the reference keeps every packet in its own malloc block;
 * the bytes of each packet do not change.  A phase from mrpdr_fmt_payload_phase lines
 * the array payloads up with cache lines, which the encode kernel's stores like.
 * total_host (optional) receives offsets[n] and synchronises the stream. */
int32_t mrpdr_wave_sizeof(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_column *cols,
                          uint64_t n, uint32_t align, uint32_t phase, uint64_t *lens,
                          uint64_t *offsets, uint64_t *total_host, void *stream);

/* ---- encode: Packet::encode_pdr_data for a whole wave -------------------------
 * (src/Packet.C:54-85 -> pdr_packet_data ENCODE, src/Packet.C:594-845 -> pdrmem_put*,
 * src/pdr_mem.c:22-343, pdr_array/pdr_bytes/pdr_string, src/pdr.c:212-472).
 * Always writes native (little-endian) order, like the reference encoder.
 * Padding bytes of a packed fixed layout (stride > length) are written as zero. */
int32_t mrpdr_wave_encode(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_column *cols,
                          uint64_t n, const mrpdr_wire *wire, void *stream);
/* The same in two calls, for callers that overlap work on two streams: begin writes scalars
 * and length prefixes and plans the payload streams -- it reads only counts / offs and the
 * scalar columns, so it may run while the array values are still being produced (e.g. by
 * mrpdr_wave_decode_finish of ANOTHER ctx on another stream); finish moves the payloads.
 * finish must follow begin on the same ctx with the same arguments. */
int32_t mrpdr_wave_encode_begin(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_column *cols,
                                uint64_t n, const mrpdr_wire *wire, void *stream);
int32_t mrpdr_wave_encode_finish(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_column *cols,
                                 uint64_t n, const mrpdr_wire *wire, void *stream);

/* ---- decode: Packet::decode_pdr_data for a whole wave -------------------------
 * (src/Packet.C:106-133 -> pdr_packet_data DECODE -> pdrmem_get* / pdrmem_get*_swap,
 * src/pdr_mem.c:40-315; byte reversal of src/byte_order.c:8-17 when
 * sender_byteorder != local).
 *
 * begin : walks every packet: scalar columns, counts[] of array/string fields and
 *         ok[i] (1 = packet decoded, 0 = truncated / oversize count; a failed packet
 *         yields zero for the failed field and all later ones), then scans counts
 *         into offs[].  totals_host (optional, TWO entries per field: [i] and [nfields+i])
 *         receives in [i] the element total of every array/string field (string arrays:
 *         the number of strings) and in [nfields+i] the byte total of a string-array
 *         field, and synchronises the stream.
 * finish: moves (and byte-reverses) the array/string payloads into values[].
 * mrpdr_wave_decode = begin + finish. */
int32_t mrpdr_wave_decode_begin(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                uint64_t n, int32_t sender_byteorder, mrpdr_column *cols,
                                int32_t *ok, uint64_t *totals_host, void *stream);
int32_t mrpdr_wave_decode_finish(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                 uint64_t n, int32_t sender_byteorder, mrpdr_column *cols,
                                 void *stream);
int32_t mrpdr_wave_decode(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_wire *wire,
                          uint64_t n, int32_t sender_byteorder, mrpdr_column *cols,
                          int32_t *ok, void *stream);

/* Waves whose packets come from senders of both byte orders (a node whose children run on
 * machines of both kinds).  The reference decodes every packet with the order its own message
 * carried (src/Packet.C:96,118 -> pdrmem_create, src/pdr_mem.c:433-449).  byteorders: device,
 * n bytes, as mrpdr_wave_decode_headers writes them; packet i is byte-reversed iff
 * byteorders[i] != mrpdr_local_byteorder().  Everything else as the calls above. */
int32_t mrpdr_wave_decode_mixed_begin(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                      uint64_t n, const uint8_t *byteorders, mrpdr_column *cols,
                                      int32_t *ok, uint64_t *totals_host, void *stream);
int32_t mrpdr_wave_decode_mixed_finish(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                       uint64_t n, const uint8_t *byteorders, mrpdr_column *cols,
                                       void *stream);
int32_t mrpdr_wave_decode_mixed(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_wire *wire,
                                uint64_t n, const uint8_t *byteorders, mrpdr_column *cols,
                                int32_t *ok, void *stream);

/* ---- packet headers --------------------------------------------------------
 * Packet::pdr_packet_header (src/Packet.C:557-593) for n packets of one format:
 *   u32 stream_id | i32 tag | u32 src_rank | u32 strlen(fmt)+1 | fmt,NUL |
 *   u64 ndest | ndest x u32 dest_rank | u8 byteorder
 * stream_ids / tags / src_ranks are device columns; a NULL column means the scalar
 * argument applies to every packet.  dest_ranks (device, ndest entries, may be NULL with
 * ndest = 0) is the destination list every packet of the wave carries: upstream waves have
 * none, a downstream multicast has one list for all its packets (src/Packet.C:579-585).
 * Header i is written at hdrs + hdr_offsets[i], or at hdrs + i*hdr_stride when hdr_offsets
 * is NULL.  mrpdr_header_bytes gives the header length without a destination list,
 * mrpdr_header_bytes_dest with ndest destinations. */
uint64_t mrpdr_header_bytes(const mrpdr_fmt *f);
uint64_t mrpdr_header_bytes_dest(const mrpdr_fmt *f, uint64_t ndest);
int32_t  mrpdr_wave_encode_headers(mrpdr_ctx *ctx, const mrpdr_fmt *f, uint64_t n,
                                   uint32_t stream_id, const uint32_t *stream_ids,
                                   int32_t tag, const int32_t *tags,
                                   uint32_t src_rank, const uint32_t *src_ranks,
                                   const uint32_t *dest_ranks, uint64_t ndest,
                                   uint8_t *hdrs, const uint64_t *hdr_offsets, uint64_t hdr_stride,
                                   void *stream);
/* Packet::decode_pdr_header (src/Packet.C:87-104): byte order of a header is its own
 * last byte.  Header i is hdrs + hdr_offsets[i] (or i*hdr_stride when hdr_offsets is
 * NULL), hdr_lens[i] long (or hdr_stride when NULL); hdrs_bytes is the size of the buffer hdrs
 * points into -- offsets and lengths taken from a Message's size vector are untrusted, and a
 * header that does not lie inside the buffer fails (ok[i] = 0) without being read.
 * Outputs (any may be NULL):
 * stream_ids, tags, src_ranks, byteorders (u8), ndests (u64), dest_ranks (packet i's first
 * min(ndest, dest_stride) destinations at dest_ranks + i*dest_stride) and ok[i] = 1 when the
 * header parsed AND its format string equals f's. */
int32_t  mrpdr_wave_decode_headers(mrpdr_ctx *ctx, const mrpdr_fmt *f, uint64_t n,
                                   const uint8_t *hdrs, uint64_t hdrs_bytes, const uint64_t *hdr_offsets,
                                   const uint64_t *hdr_lens, uint64_t hdr_stride,
                                   uint32_t *stream_ids, int32_t *tags, uint32_t *src_ranks,
                                   uint8_t *byteorders, uint64_t *ndests,
                                   uint32_t *dest_ranks, uint64_t dest_stride, int32_t *ok,
                                   void *stream);

/* ---- message framing -----------------------------------------------------------
 * Message::send / Message::recv (src/Message.C:201-335, 48-199).  A message is
 *   u8 byteorder | u32 npackets | 2n x u64 sizes (hdr0, buf0, hdr1, ...) | 1 spare byte |
 *   hdr0 | buf0 | hdr1 | buf1 | ...
 * Its payload part (everything after the preamble) IS a wave arena with interleaved headers,
 * so a wave is encoded straight into the message buffer and decoded straight out of it:
 *
 *   send  lens/offsets = mrpdr_wave_sizeof(align 1, phase = mrpdr_header_bytes(f))
 *         -> offsets[i] is the position of packet i's data in the payload, its header sits
 *            hdr_len bytes in front of it, offsets[n] is the payload size;
 *         mrpdr_message_write_preamble(msg);  payload = msg + mrpdr_message_preamble_bytes(n);
 *         mrpdr_wave_encode(wire{payload, offsets, lens});
 *         mrpdr_wave_encode_headers(hdrs = payload - hdr_len, hdr_offsets = offsets)
 *   recv  the host reads the 5-byte count, then the size vector (as the reference does);
 *         mrpdr_message_parse_sizes turns the size vector into hdr/buf offsets and lengths
 *         (arrays of n+1 offsets; [n] = payload size) for mrpdr_wave_decode_headers and
 *         mrpdr_wave_decode on the payload.
 * The size vector is read in the local byte order, exactly like src/Message.C:117. */
uint64_t mrpdr_message_preamble_bytes(uint64_t n);
int32_t  mrpdr_message_write_preamble(mrpdr_ctx *ctx, uint64_t n, uint64_t hdr_len,
                                      const uint64_t *lens, uint8_t *msg, void *stream);
int32_t  mrpdr_message_parse_sizes(mrpdr_ctx *ctx, const uint8_t *sizes, uint64_t n,
                                   uint64_t *hdr_offsets, uint64_t *hdr_lens,
                                   uint64_t *buf_offsets, uint64_t *buf_lens,
                                   uint64_t *payload_bytes_host, void *stream);

/* The framing of both directions as one call each (no host synchronisation in either).
 *
 * mrpdr_message_write_frame = mrpdr_message_write_preamble + mrpdr_wave_encode_headers for a wave
 * without destination lists, in one launch: what Message::send adds around the data buffers
 * (src/Message.C:201-335 over the header buffers of Packet::encode_pdr_header, src/Packet.C:557-593).
 * lens / offsets are mrpdr_wave_sizeof's (align 1, phase = mrpdr_header_bytes(f)); msg is the whole
 * message buffer, mrpdr_message_preamble_bytes(n) + offsets[n] bytes; stream_ids / tags / src_ranks
 * as in mrpdr_wave_encode_headers.
 *
 * mrpdr_message_parse = mrpdr_message_parse_sizes + mrpdr_wave_decode_headers in two launches: what
 * Message::recv does with a received message before the packets' data is looked at
 * (src/Message.C:48-199, which builds a Packet from every header buffer: Packet::decode_pdr_header,
 * src/Packet.C:87-104).  msg / msg_bytes: the whole message as received; n: its packet count (the
 * receiver read the 5-byte count off the link before it could receive the rest, src/Message.C:71-86).
 * Offsets are relative to the payload, msg + mrpdr_message_preamble_bytes(n); [n] of both offset
 * arrays is the payload size the size vector announces.  Sizes are untrusted: a header or data
 * buffer that does not lie inside msg fails as a packet (hdr_ok[i] = 0, ok[i] = 0 from the decode),
 * nothing outside msg is read.  Header outputs as in mrpdr_wave_decode_headers (any may be NULL). */
int32_t  mrpdr_message_write_frame(mrpdr_ctx *ctx, const mrpdr_fmt *f, uint64_t n,
                                   const uint64_t *lens, const uint64_t *offsets,
                                   uint32_t stream_id, const uint32_t *stream_ids,
                                   int32_t tag, const int32_t *tags,
                                   uint32_t src_rank, const uint32_t *src_ranks,
                                   uint8_t *msg, void *stream);
int32_t  mrpdr_message_parse(mrpdr_ctx *ctx, const mrpdr_fmt *f, const uint8_t *msg, uint64_t msg_bytes,
                             uint64_t n, uint64_t *hdr_offsets, uint64_t *hdr_lens,
                             uint64_t *buf_offsets, uint64_t *buf_lens,
                             uint32_t *stream_ids, int32_t *tags, uint32_t *src_ranks,
                             uint8_t *byteorders, uint64_t *ndests,
                             uint32_t *dest_ranks, uint64_t dest_stride, int32_t *hdr_ok, void *stream);

/* ---- column views: Packet::unpack for a whole wave, without the copies ----------------
 * The reference hands a tool its data one packet at a time: Packet::unpack(fmt, ...) checks the
 * format string and then copies -- scalars by value, every array with malloc + memcpy, every string
 * with strdup, all owned by the caller afterwards (DataElementArray2ArgList, src/Packet.C:1065-1431;
 * ExtractVaList, src/Packet.C:522-542).  A consumer of whole waves (a filter, a tool's own kernel)
 * does not need copies: the decoded columns ARE the fields of all packets.  mrpdr_wave_view gives
 * them out under the same format check, with these rules:
 *   - a view aliases the columns handed to mrpdr_wave_decode*: values / counts / offs are the
 *     caller's own device buffers, nothing is allocated, copied or synchronised, nothing needs
 *     to be freed for the view itself;
 *   - it describes the wave once the decode calls that were enqueued on `stream` have run: read it
 *     from work enqueued on the same stream behind them, or after synchronising;
 *   - it stays valid until the owner of the columns frees them or decodes another wave into them --
 *     consumers first, columns last; the codec itself keeps no reference to them after a call returns;
 *   - it is read-only by contract: a filter that wants to change a value makes its own output
 *     column, as the reference's filters build a new packet.
 * views: one entry per field.  scalar field: values = T[n]; array / string field: element j of
 * packet i is values[offs[i] + j], j < counts[i] (strings: bytes, NUL included); string arrays as
 * in struct mrpdr_column.  Returns FALSE (and fills nothing) when fmt_expected is not the wave's
 * format string, as unpack returns -1 (src/Packet.C:527-532). */
typedef struct mrpdr_field_view {
    const void     *values;
    const uint64_t *counts;        /* NULL for scalar fields */
    const uint64_t *offs;          /* NULL for scalar fields; n + 1 entries */
    const uint64_t *sub_counts;    /* string arrays only */
    const uint64_t *sub_offs;
    uint64_t        n;             /* packets */
    int32_t         type;          /* enum mrpdr_type */
    int32_t         kind;          /* enum mrpdr_kind */
    int32_t         width;         /* bytes per element */
    int32_t         reserved;
} mrpdr_field_view;

int32_t mrpdr_wave_view(const mrpdr_fmt *f, const char *fmt_expected, const mrpdr_column *cols, uint64_t n,
                        mrpdr_field_view *views);

/* ---- filters: tfilter_Sum / Min / Max / Avg over a wave ---------------------------
 * (src/FilterDefinitions.C:90-645; arithmetic helpers :1925-2136.)
 * `type` is a scalar DataType.  values = T[n] with a byte stride of value_stride
 * (sizeof(T) for a decoded column); counts (AVG only) = int32[n] with count_stride.
 * swap != 0 byte-reverses every value and count while loading, so the call can run
 * straight on the wire bytes of a fixed-format wave (fused decode + reduce):
 *   values = wire.base, value_stride = packet stride, counts = wire.base + sizeof(T).
 * result (device, 16 bytes): the data bytes of the output packet the reference filter
 * would build -- T value, then for AVG the int32 count -- in native order, zero padded.
 * result_ok (device int32, optional): 1 = result present, 0 = absent (n == 0, or an AVG whose
 * counts sum to zero: the reference emits no packet).
 * result and result_ok may point into pinned host memory.  The sixteen result bytes are written
 * first, then a system-scope fence, then result_ok with release semantics: a host thread that
 * polls instead of synchronising presets *result_ok to MRPDR_RESULT_PENDING, waits until it reads
 * 0 or 1 (an acquire load), and then finds the result bytes complete.  (0 is a final value --
 * "absent" -- so it cannot serve as the pending mark.) */
#define MRPDR_RESULT_PENDING (-1)
int32_t mrpdr_wave_filter(mrpdr_ctx *ctx, int32_t op, int32_t type, int32_t shape,
                          const void *values, uint64_t value_stride,
                          const void *counts, uint64_t count_stride,
                          uint64_t n, int32_t swap,
                          void *result, int32_t *result_ok, void *stream);

/* The same over a wave whose packets come from senders of both byte orders: byteorders (device,
 * n bytes, as mrpdr_wave_decode_headers writes them) says for every packet whether its value
 * (and count) is byte-reversed, as (*pkt)[0] decodes every packet with the order of its own
 * message (src/Packet.C:118). */
int32_t mrpdr_wave_filter_mixed(mrpdr_ctx *ctx, int32_t op, int32_t type, int32_t shape,
                                const void *values, uint64_t value_stride,
                                const void *counts, uint64_t count_stride,
                                uint64_t n, const uint8_t *byteorders,
                                void *result, int32_t *result_ok, void *stream);

/* ---- one filter call over a wave that is held in ranges ---------------------------------
 * In the reference a node's wave goes through ONE tfilter_* call, however many packets it has
 * (Filter::push_Packets hands the whole vector to the filter function, src/Filter.C:42-117;
 * tfilter_Avg multiplies, sums and divides once, src/FilterDefinitions.C:517-561).  When the packets
 * of that wave lie on several GPUs (contiguous packet ranges, rank order), every GPU reduces its own
 * range to a 16-byte NODE, the nodes are exchanged, and combining them in range order gives the
 * result of that one call -- the same bits as MRPDR_SHAPE_FLAT over the whole wave.  The GPUs are not
 * tree levels: nothing is divided or rounded per range that the one call would not divide or round.
 *
 * What a node holds (native order):
 *   integer SUM / MIN / MAX / OR / AND   bits = the range's result
 *   integer AVG                          bits = sum of (T)(value * count), wrapping, in T;
 *                                        count = sum of the counts (the ONE division is the combine's)
 *   float / double MIN / MAX             bits = the range's result, count = 1 when a NaN restarted the
 *                                        comparison inside the range ((r > x) ? r : x keeps the new
 *                                        element whenever the comparison is false)
 *   float / double SUM / AVG             bits = the running sum of the chain of additions (AVG: of
 *                                        value * count, count = sum of the counts, undivided)
 *   present = 0: no packet below this node.
 * Every addition of a float or double SUM / AVG rounds, so that chain has to be CONTINUED, not
 * re-associated: mrpdr_filter_is_chained(op, type) says so, and for those the ranges are taken one
 * after the other, range r + 1 with `front` = the node range r left (a relay in rank order; the last
 * node is the wave's).  Every other (op, type) needs no order between the calls: front = NULL
 * everywhere, one all-gather of the nodes, one combine.  (`front` is honoured for those too: the node
 * then covers front + range.)
 *
 * mrpdr_wave_filter_partial: arguments as mrpdr_wave_filter; front (device, may be NULL) = node of
 * the packets in front of this range; node (device) receives the node of front + range.  n = 0 is
 * allowed (a rank without packets hands `front` on).
 * mrpdr_wave_filter_combine: nodes = g nodes in range order (device).  Chained operators take the
 * LAST node (it holds the whole chain when the ranges were relayed as above); all others fold the g
 * nodes in order.  result / result_ok as mrpdr_wave_filter. */
typedef struct mrpdr_node {
    uint64_t bits;
    int32_t  count;
    int32_t  present;
} mrpdr_node;

int32_t mrpdr_filter_is_chained(int32_t op, int32_t type);
int32_t mrpdr_wave_filter_partial(mrpdr_ctx *ctx, int32_t op, int32_t type,
                                  const void *values, uint64_t value_stride,
                                  const void *counts, uint64_t count_stride,
                                  uint64_t n, int32_t swap,
                                  const mrpdr_node *front, mrpdr_node *node, void *stream);
/* the same for a range whose packets come from senders of both byte orders (byteorders[i] as in
 * mrpdr_wave_filter_mixed: Packet::decode_pdr_data decodes every packet with its own, src/Packet.C:118) */
int32_t mrpdr_wave_filter_partial_mixed(mrpdr_ctx *ctx, int32_t op, int32_t type,
                                        const void *values, uint64_t value_stride,
                                        const void *counts, uint64_t count_stride,
                                        uint64_t n, const uint8_t *byteorders,
                                        const mrpdr_node *front, mrpdr_node *node, void *stream);
int32_t mrpdr_wave_filter_combine(mrpdr_ctx *ctx, int32_t op, int32_t type,
                                  const mrpdr_node *nodes, uint64_t g,
                                  void *result, int32_t *result_ok, void *stream);

/* What a tree node does with an upstream wave, in one call: decode the packets
 * (Packet::decode_pdr_data, src/Packet.C:106-133, triggered by the filter's (*pkt)[0],
 * src/FilterDefinitions.C:148) AND fold their first field with tfilter_Sum / Min / Max / Avg
 * (src/FilterDefinitions.C:90-645).  f must be a format of scalars only; the filtered value is
 * field 0 (AVG: field 1 is its int32 count); cols / ok as for mrpdr_wave_decode, result /
 * result_ok as for mrpdr_wave_filter over the decoded column -- the same bits.  For a packed arena
 * (offsets NULL, stride <= 32) every row is read once and the call is two launches; other
 * layouts, and the flat SUM / AVG of float and double (a serial fold), run decode and filter one
 * after the other inside the call. */
int32_t mrpdr_wave_decode_filter(mrpdr_ctx *ctx, const mrpdr_fmt *f, const mrpdr_wire *wire, uint64_t n,
                                 int32_t sender_byteorder, mrpdr_column *cols, int32_t *ok,
                                 int32_t op, int32_t shape, void *result, int32_t *result_ok, void *stream);

/* mrpdr_wave_filter for callers that cannot allocate device memory or create a ctx: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  (No counterpart in the
 * reference.)  Same arguments and results as mrpdr_wave_filter, bit for bit, minus the ctx: the
 * nodes the two launches hand over live in a device array built into the library (256 KiB, held by
 * every GPU the library runs on until the process ends), so the call makes no hipMalloc / hipFree
 * and needs no workspace.  (The array also takes the chunk summaries of a flat float / double SUM or
 * AVG, for up to 536,000 float or 273,000 double packets; larger flat float folds fall back to the
 * one-workgroup scan: same bits, slower.)  Limits: n <= 2^24 packets (FALSE beyond); the kernels run on the calling
 * thread's current device, where all pointers must live; all calls on one GPU share that GPU's
 * array, so two calls that overlap in time on one GPU (different streams or threads) race --
 * serialise them.  The first call on a device resolves the array's address
 * (hipGetSymbolAddress): make one call outside graph capture first. */
int32_t mrpdr_wave_filter_noalloc(int32_t op, int32_t type, int32_t shape,
                                  const void *values, uint64_t value_stride,
                                  const void *counts, uint64_t count_stride,
                                  uint64_t n, int32_t swap,
                                  void *result, int32_t *result_ok, void *stream);

/* tfilter_Min, tfilter_Max and tfilter_Avg of one '<T> %d' wave in ONE pass over the packets
 * (a monitoring stream usually carries all three).  Same arguments as mrpdr_wave_filter with the
 * declared tree shape; results = 3 x 16 bytes (MIN, MAX, AVG output packets' data), results_ok = 3
 * flags.  Each result equals that of its own mrpdr_wave_filter(MRPDR_SHAPE_TREE) call bit for bit. */
int32_t mrpdr_wave_filter_stats(mrpdr_ctx *ctx, int32_t type,
                                const void *values, uint64_t value_stride,
                                const void *counts, uint64_t count_stride,
                                uint64_t n, int32_t swap_bytes,
                                void *results, int32_t *results_ok, void *stream);

/* ---- array filters ----------------------------------------------------------
 * tfilter_ArrayConcat (src/FilterDefinitions.C:649-810): the output packet carries the arrays of
 * all input packets back to back -- which is what the decoded values column of the wave already
 * is.  Writes the column form of that ONE packet into out: out->counts[0] = total elements,
 * out->offs[0..1] = {0, total} (string arrays: sub_counts / sub_offs are the input's own); the
 * caller points out->values at the input's values and encodes out with n = 1.  The reference
 * keeps the total in a 32-bit unsigned (:655, :771-772): totals of 2^32 and more are undefined
 * there and set status bit 1 here. */
int32_t mrpdr_wave_array_concat(mrpdr_ctx *ctx, const mrpdr_column *in, uint64_t n,
                                mrpdr_column *out, void *stream);

/* The concatenation tfilter_PerfData performs for TFILTER_ARRAY_CONCAT
 * (src/FilterDefinitions.C:1000-1008, 1171-1203): as mrpdr_wave_array_concat, but packets whose
 * mask_counts[i] is 0 (no data) contribute nothing; the kept elements (width bytes each) are
 * copied into out->values (capacity checked: status bit 0), out->counts[0] / offs[0..1] set.
 * Called once per field (ranks, nelems, data) with the data field's counts as mask. */
int32_t mrpdr_wave_array_concat_masked(mrpdr_ctx *ctx, const mrpdr_column *in, uint32_t width,
                                       const uint64_t *mask_counts, uint64_t n,
                                       mrpdr_column *out, void *stream);

/* tfilter_PerfData with aggregation SUM / MIN / MAX / AVG (src/FilterDefinitions.C:904-1239)
 * over a decoded wave of "%ad %ad %a{uld,ld,lf}" packets: `ranks` is the column of field 0,
 * `data` the column of field 2, type one of MRPDR_UINT64_T / MRPDR_INT64_T / MRPDR_DOUBLE_T.
 * The data arrays are folded element by element in packet order with the reference's rules
 * (packets without data skipped; rank[0] < 0 = an aggregate of -rank ranks; MIN starts from
 * bytes 0xFF, the others from 0; AVG weights by the rank count and divides the first
 * data_len(last packet) elements by the rank total).  The output packet is
 * ([out_rank[0]], [out_len[0]], out_data[0 .. out_len[0])): out_data holds out_cap elements
 * (a longer result sets status bit 0 and out_len 0), out_len / out_rank are device pointers.
 *   group = 0 or >= n : one filter invocation over the n packets (a node with n children), bit
 *                       for bit.  uint64 / int64 data does not depend on how the packets are
 *                       grouped (wrapping sums of data * nrank and of the ranks, ONE division at
 *                       the end, src/FilterDefinitions.C:1084-1132; MIN / MAX from their fixed
 *                       start values): given `scratch` of mrpdr_perfdata_scratch_bytes(n, 0,
 *                       out_cap) bytes, waves of 2,048 packets and more are folded as a tree of
 *                       undivided nodes (short chains side by side) instead of one chain over all
 *                       n packets.  Doubles round at every addition, so their result IS the chain of
 *                       each element: with that scratch, waves of 8,192 packets and more whose result
 *                       has at most 64 elements take the elements' chains side by side through the
 *                       exact parallel fold the scalar filters use for MRPDR_SHAPE_FLAT (arrays of one
 *                       length in place, others from a dense copy in the scratch: n * out_cap * 8
 *                       bytes of it).  Without scratch the packets are folded one after the other;
 *   0 < group < n     : a two-level tree -- nodes of `group` consecutive packets, then one node
 *                       over their outputs (the filter runs at every level, as in an MRNet
 *                       tree: AVG divides and re-weights per level).  Needs `scratch` of
 *                       mrpdr_perfdata_scratch_bytes(n, group, out_cap) bytes, 16-byte aligned. */
uint64_t mrpdr_perfdata_scratch_bytes(uint64_t n, uint64_t group, uint64_t out_cap);
int32_t mrpdr_wave_perfdata(mrpdr_ctx *ctx, int32_t op, int32_t type,
                            const mrpdr_column *ranks, const mrpdr_column *data, uint64_t n,
                            uint64_t group, void *out_data, uint64_t out_cap,
                            uint64_t *out_len, int32_t *out_rank,
                            void *scratch, uint64_t scratch_bytes, void *stream);

/* The same filter as a declared tree of any depth: level l folds groups of fanins[l] consecutive
 * packets of the level below (0: all of them) with one filter invocation per group, the way a tree of
 * MRNet processes with those fan-ins runs tfilter_PerfData at every node (AVG divides and re-weights
 * per node, the output of a node carries rank -total).  If the fan-ins leave more than one node, one
 * more level folds them all.  nlevels <= 8.  Short arrays (out_cap <= 64) put many nodes into one
 * workgroup (256 / W of them, W = out_cap rounded up to a power of two, 8 at least), and a level whose
 * fan-in is exactly that number is folded by the launch of the level below it: 131,072 packets of
 * eight-element arrays as { 16, 32, 32, 0 } are two launches.
 * scratch: mrpdr_perfdata_levels_scratch_bytes(n, fanins, nlevels, out_cap) bytes, 16-byte aligned. */
uint64_t mrpdr_perfdata_levels_scratch_bytes(uint64_t n, const uint64_t *fanins, int32_t nlevels, uint64_t out_cap);
int32_t mrpdr_wave_perfdata_levels(mrpdr_ctx *ctx, int32_t op, int32_t type,
                                   const mrpdr_column *ranks, const mrpdr_column *data, uint64_t n,
                                   const uint64_t *fanins, int32_t nlevels, void *out_data, uint64_t out_cap,
                                   uint64_t *out_len, int32_t *out_rank,
                                   void *scratch, uint64_t scratch_bytes, void *stream);

/* One tfilter_PerfData invocation over a wave that is held in packet ranges (one per GPU, rank order),
 * the counterpart of mrpdr_wave_filter_partial / _combine: every range is folded into a STATE -- this
 * header followed by `cap` accumulators of 8 bytes, nothing divided -- and the states combine into the
 * output packet of the one invocation over the whole wave (src/FilterDefinitions.C:984-1132).
 *   uint64 / int64 : no order between the ranges is needed (front = NULL everywhere); gather the
 *                    states, combine all g of them;
 *   double         : mrpdr_perfdata_is_chained() -- every addition rounds, and MIN / MAX keep the later
 *                    operand when a comparison with a NaN is false: range r + 1 continues from the
 *                    state range r left (front), the last state is the wave's; combine takes the LAST
 *                    of the g states it is given.
 * A range without packets (n = 0) hands `front` on (or an empty state).  scratch as for
 * mrpdr_wave_perfdata(group = 0) (optional: integer ranges of 2,048 packets and more and double ranges
 * of 8,192 packets and more use it). */
typedef struct mrpdr_perf_state {
    uint64_t longest;      /* elements that have taken at least one packet */
    uint64_t last;         /* data length of the last packet below (the elements AVG divides) */
    int32_t  rank;         /* 0 - total number of ranks below: the rank value of the output packet */
    uint32_t packets;      /* 0: no packet below */
    uint64_t reserved;
} mrpdr_perf_state;
uint64_t mrpdr_perfdata_state_bytes(uint64_t cap);        /* sizeof(mrpdr_perf_state) + 8 * cap */
int32_t  mrpdr_perfdata_is_chained(int32_t op, int32_t type);
int32_t  mrpdr_wave_perfdata_partial(mrpdr_ctx *ctx, int32_t op, int32_t type,
                                     const mrpdr_column *ranks, const mrpdr_column *data, uint64_t n, uint64_t cap,
                                     const void *front, void *state,
                                     void *scratch, uint64_t scratch_bytes, void *stream);
int32_t  mrpdr_wave_perfdata_combine(mrpdr_ctx *ctx, int32_t op, int32_t type,
                                     const void *states, uint64_t g, uint64_t cap,
                                     void *out_data, uint64_t out_cap, uint64_t *out_len, int32_t *out_rank,
                                     void *stream);

/* tfilter_IntEqClass (src/FilterDefinitions.C:812-902) over a decoded wave of "%aud %aud %aud"
 * packets (class values, members per class, members): the classes of all packets merged by
 * value, ascending; the members of a class in arrival order (packet order, then class order
 * inside a packet) -- what the reference's std::map<unsigned, std::vector<unsigned>> yields.
 * vals / memcnts / mems are the columns of fields 0, 1, 2 (vals->capacity bounds the number of
 * class entries, 2^32 at most); the three outputs are the columns of the ONE output packet
 * (counts[1], offs[2], values with capacity: too small sets status bit 0 and empty outputs).
 * scratch: mrpdr_eqclass_scratch_bytes(vals->capacity) bytes, 16-byte aligned. */
uint64_t mrpdr_eqclass_scratch_bytes(uint64_t class_capacity);
int32_t mrpdr_wave_int_eq_class(mrpdr_ctx *ctx, const mrpdr_column *vals, const mrpdr_column *memcnts,
                                const mrpdr_column *mems, uint64_t n,
                                mrpdr_column *out_vals, mrpdr_column *out_memcnts, mrpdr_column *out_mems,
                                void *scratch, uint64_t scratch_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MRPDR_H */
