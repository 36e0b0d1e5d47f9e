// C++ host side of the wave codec: the reference's Packet surface for n packets at a time,
// over the C ABI of include/mrpdr.h.  Host buffers in, host buffers out -- the legacy
// boundary of src/Packet.C (malloc'ed hdr/buf blocks, caller-owned unpacked arrays); the HIP
// kernels do all encoding, decoding and filtering.  Header-only; link -lmrpdr -lamdhip64.
//
// Mirrors include/mrnet/Packet.h:41-118:
//   PacketWave(ctx, stream_id, tag, fmt, cols, n)   API constructor, encodes the data at once
//                                                    (src/Packet.C:137-165 -> encode_pdr_data)
//   PacketWave(ctx, fmt, n, hdrs, hdr_len, bufs..)  receive-side constructor: headers decoded at
//                                                    once, data lazily (src/Packet.C:277-295)
//   unpack(fmt, cols)                               -1 when fmt differs (src/Packet.C:527-532)
//   get_Header / get_Buffer                         encoded bytes (header encoded lazily,
//                                                    src/Message.C:288-292)
//   tfilter_Sum / Max / Min / Avg                   one output packet with the stream id and tag
//                                                    of the first input (src/FilterDefinitions.C:151-221)
//   tfilter_ArrayConcat / PerfData / IntEqClass     the array filters (src/FilterDefinitions.C:649-1239)
//   PacketWave(ctx, fmt, msg, bytes) / to_message    Message::recv / Message::send for the wave
//                                                    (src/Message.C:48-335)
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string.h>

#include <stdlib.h>

#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/mrpdr.h"

namespace MRN_GPU {

struct PdrError : std::runtime_error {
    explicit PdrError(const std::string &what) : std::runtime_error(what + ": " + mrpdr_last_error()) {}
};

inline void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// One field of a wave in host memory.  Scalars: values holds n elements.  Arrays and strings:
// counts[i] elements in packet i (a string counts strlen+1), values holds them back to back.
// String arrays (%as): counts[i] strings in packet i, sub_counts[j] = strlen+1 of string j of the
// wave, values holds the bytes of all strings back to back, NULs included.
struct HostColumn {
    std::vector<uint8_t>  values;
    std::vector<uint64_t> counts;
    std::vector<uint64_t> sub_counts;
    template <typename T> void set_scalars(const std::vector<T> &v)
    {
        values.assign((const uint8_t *)v.data(), (const uint8_t *)(v.data() + v.size()));
        counts.clear();
    }
    template <typename T> void set_arrays(const std::vector<uint64_t> &cnt, const std::vector<T> &v)
    {
        values.assign((const uint8_t *)v.data(), (const uint8_t *)(v.data() + v.size()));
        counts = cnt;
    }
    void set_string_arrays(const std::vector<std::vector<std::string> > &per_packet)
    {
        values.clear(); counts.clear(); sub_counts.clear();
        for (const std::vector<std::string> &strs : per_packet) {
            counts.push_back(strs.size());
            for (const std::string &t : strs) {
                sub_counts.push_back(t.size() + 1);
                values.insert(values.end(), t.c_str(), t.c_str() + t.size() + 1);
            }
        }
    }
    template <typename T> const T *as() const { return (const T *)values.data(); }
};

// ---- one packet at a time, the way user code writes it -----------------------------------
// User code creates packets one by one -- Packet(stream_id, tag, "%d %f %s %ad", 5, 2.5f, "txt",
// arr, 3u) (src/Packet.C:137-165; the varargs are read by ArgList2DataElementArray, :847-1063)
// -- and reads them back with unpack(fmt, &a, &b, ...) (:329-353, 1065-1431).  WaveBuilder and
// WaveReader keep those two call shapes on top of the columns of a wave.  Arguments per token:
// a scalar takes one arithmetic value (converted to the token's C type: a variadic template has
// no default promotions to undo), '%s' a const char*, '%aT' / '%AT' a pointer and a length,
// '%as' a const char* const* and a length.
struct WaveArg {
    enum Kind { INT, UINT, FLT, PTR } kind;
    int64_t i; uint64_t u; double d; const void *p;
};
template <typename T> inline WaveArg make_wave_arg(T v)
{
    WaveArg a{};
    if constexpr (std::is_pointer<T>::value || std::is_same<T, std::nullptr_t>::value) { a.kind = WaveArg::PTR; a.p = (const void *)v; }
    else if constexpr (std::is_floating_point<T>::value) { a.kind = WaveArg::FLT; a.d = (double)v; }
    else if constexpr (std::is_unsigned<T>::value) { a.kind = WaveArg::UINT; a.u = (uint64_t)v; }
    else { a.kind = WaveArg::INT; a.i = (int64_t)v; }
    return a;
}

class WaveBuilder {
 public:
    explicit WaveBuilder(const char *fmt) : fmt_(mrpdr_fmt_compile(fmt)), n_(0)
    {
        if (!fmt_) throw PdrError("mrpdr_fmt_compile");
        cols_.resize(mrpdr_fmt_nfields(fmt_));
    }
    ~WaveBuilder() { mrpdr_fmt_free(fmt_); }
    WaveBuilder(const WaveBuilder &) = delete;
    WaveBuilder &operator=(const WaveBuilder &) = delete;

    // One packet.  Returns false (and adds nothing) when the arguments do not fit the format --
    // the reference cannot check that; a NULL string is an encode failure there too (src/pdr.c:280-282).
    template <typename... A> bool add(A... args)
    {
        const WaveArg a[] = { make_wave_arg(args)..., WaveArg{} };
        return add_args(a, sizeof...(A));
    }
    uint64_t size() const { return n_; }
    const char *format() const { return mrpdr_fmt_string(fmt_); }
    const std::vector<HostColumn> &columns() const { return cols_; }

 private:
    static bool scalar_bytes(int type, const WaveArg &a, uint8_t out[8])
    {
        if (a.kind == WaveArg::PTR) return false;
        const double d = a.kind == WaveArg::FLT ? a.d : (a.kind == WaveArg::UINT ? (double)a.u : (double)a.i);
        const int64_t i = a.kind == WaveArg::FLT ? (int64_t)a.d : (a.kind == WaveArg::UINT ? (int64_t)a.u : a.i);
        const uint64_t u = a.kind == WaveArg::UINT ? a.u : (uint64_t)i;
        switch (type) {
        case MRPDR_CHAR_T:   { int8_t v = (int8_t)i; memcpy(out, &v, 1); return true; }
        case MRPDR_UCHAR_T:  { uint8_t v = (uint8_t)u; memcpy(out, &v, 1); return true; }
        case MRPDR_INT16_T:  { int16_t v = (int16_t)i; memcpy(out, &v, 2); return true; }
        case MRPDR_UINT16_T: { uint16_t v = (uint16_t)u; memcpy(out, &v, 2); return true; }
        case MRPDR_INT32_T:  { int32_t v = (int32_t)i; memcpy(out, &v, 4); return true; }
        case MRPDR_UINT32_T: { uint32_t v = (uint32_t)u; memcpy(out, &v, 4); return true; }
        case MRPDR_INT64_T:  { memcpy(out, &i, 8); return true; }
        case MRPDR_UINT64_T: { memcpy(out, &u, 8); return true; }
        case MRPDR_FLOAT_T:  { float v = (float)d; memcpy(out, &v, 4); return true; }
        case MRPDR_DOUBLE_T: { memcpy(out, &d, 8); return true; }
        }
        return false;
    }

    bool add_args(const WaveArg *a, size_t na)
    {
        const int nf = mrpdr_fmt_nfields(fmt_);
        std::vector<HostColumn> next = cols_;          // all or nothing
        size_t k = 0;
        for (int f = 0; f < nf; f++) {
            const int kind = mrpdr_fmt_field_kind(fmt_, f), w = mrpdr_fmt_field_width(fmt_, f);
            HostColumn &c = next[f];
            if (kind == MRPDR_SCALAR) {
                uint8_t b[8];
                if (k >= na || !scalar_bytes(mrpdr_fmt_field_type(fmt_, f), a[k++], b)) return false;
                c.values.insert(c.values.end(), b, b + w);
            } else if (kind == MRPDR_STRING) {
                if (k >= na || a[k].kind != WaveArg::PTR || !a[k].p) return false;
                const char *t = (const char *)a[k++].p;
                const size_t l = strlen(t) + 1;
                c.counts.push_back(l);
                c.values.insert(c.values.end(), (const uint8_t *)t, (const uint8_t *)t + l);
            } else {
                if (k + 2 > na || a[k].kind != WaveArg::PTR || a[k + 1].kind == WaveArg::PTR || a[k + 1].kind == WaveArg::FLT) return false;
                const void *p = a[k].p;
                const uint64_t len = a[k + 1].kind == WaveArg::UINT ? a[k + 1].u : (uint64_t)a[k + 1].i;
                k += 2;
                if (len && !p) return false;
                c.counts.push_back(len);
                if (kind == MRPDR_ARRAY) {
                    c.values.insert(c.values.end(), (const uint8_t *)p, (const uint8_t *)p + len * w);
                } else {                                   // string array
                    const char *const *strs = (const char *const *)p;
                    for (uint64_t j = 0; j < len; j++) {
                        if (!strs[j]) return false;
                        const size_t l = strlen(strs[j]) + 1;
                        c.sub_counts.push_back(l);
                        c.values.insert(c.values.end(), (const uint8_t *)strs[j], (const uint8_t *)strs[j] + l);
                    }
                }
            }
        }
        if (k != na) return false;
        cols_.swap(next);
        n_++;
        return true;
    }

    mrpdr_fmt *fmt_;
    uint64_t n_;
    std::vector<HostColumn> cols_;
};

// Packet::unpack for packet i of an unpacked wave: scalars are copied out, an array comes back as
// a malloc'ed copy the caller frees plus its length, a string as a malloc'ed copy -- the ownership
// rules of DataElementArray2ArgList (src/Packet.C:1065-1431; zero-length array -> NULL).
class WaveReader {
 public:
    WaveReader(const char *fmt, const std::vector<HostColumn> &cols, uint64_t n)
        : fmt_(mrpdr_fmt_compile(fmt)), cols_(cols), n_(n)
    {
        if (!fmt_) throw PdrError("mrpdr_fmt_compile");
        const int nf = mrpdr_fmt_nfields(fmt_);
        offs_.resize(nf); sub_offs_.resize(nf);
        for (int f = 0; f < nf; f++) {
            if (mrpdr_fmt_field_kind(fmt_, f) == MRPDR_SCALAR) continue;
            offs_[f].assign(n + 1, 0);
            for (uint64_t i = 0; i < n; i++) offs_[f][i + 1] = offs_[f][i] + cols[f].counts[i];
            sub_offs_[f].assign(cols[f].sub_counts.size() + 1, 0);
            for (size_t j = 0; j < cols[f].sub_counts.size(); j++) sub_offs_[f][j + 1] = sub_offs_[f][j] + cols[f].sub_counts[j];
        }
    }
    ~WaveReader() { mrpdr_fmt_free(fmt_); }
    WaveReader(const WaveReader &) = delete;
    WaveReader &operator=(const WaveReader &) = delete;

    // unpack(i, fmt, &a, &b, ...): -1 when fmt differs from the wave's format or i is out of range
    template <typename... A> int unpack(uint64_t i, const char *fmt, A *...outs)
    {
        if (!fmt || strcmp(fmt, mrpdr_fmt_string(fmt_)) != 0 || i >= n_) return -1;
        void *o[] = { (void *)outs..., nullptr };
        size_t k = 0;
        const int nf = mrpdr_fmt_nfields(fmt_);
        for (int f = 0; f < nf; f++) {
            const int kind = mrpdr_fmt_field_kind(fmt_, f), w = mrpdr_fmt_field_width(fmt_, f);
            const HostColumn &c = cols_[f];
            if (kind == MRPDR_SCALAR) {
                if (k >= sizeof...(A)) return -1;
                memcpy(o[k++], c.values.data() + i * w, w);
            } else if (kind == MRPDR_STRING) {
                if (k >= sizeof...(A)) return -1;
                const uint64_t l = c.counts[i];
                char *t = (char *)malloc(l ? l : 1);
                memcpy(t, c.values.data() + offs_[f][i], l);
                *(char **)o[k++] = t;
            } else {
                if (k + 2 > sizeof...(A)) return -1;
                const uint64_t len = c.counts[i];
                void *copy = nullptr;
                if (len && kind == MRPDR_ARRAY) {
                    copy = malloc(len * w);
                    memcpy(copy, c.values.data() + offs_[f][i] * w, len * w);
                } else if (len) {
                    char **strs = (char **)malloc(len * sizeof(char *));
                    for (uint64_t j = 0; j < len; j++) {
                        const uint64_t s = offs_[f][i] + j, l = c.sub_counts[s];
                        strs[j] = (char *)malloc(l);
                        memcpy(strs[j], c.values.data() + sub_offs_[f][s], l);
                    }
                    copy = strs;
                }
                *(void **)o[k++] = copy;
                // '%a*' lengths are 32-bit in the API, '%A*' 64-bit (src/Packet.C:904-934)
                if (mrpdr_fmt_field_type(fmt_, f) >= MRPDR_CHAR_LRG_ARRAY_T) *(uint64_t *)o[k++] = len;
                else *(uint32_t *)o[k++] = (uint32_t)len;
            }
        }
        return k == sizeof...(A) ? 0 : -1;
    }

 private:
    mrpdr_fmt *fmt_;
    const std::vector<HostColumn> &cols_;
    uint64_t n_;
    std::vector<std::vector<uint64_t> > offs_, sub_offs_;
};

class DeviceBuffer {
 public:
    DeviceBuffer() : p_(nullptr), bytes_(0) {}
    ~DeviceBuffer() { if (p_) (void)hipFree(p_); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    void resize(size_t bytes)
    {
        if (bytes <= bytes_ && p_) return;
        if (p_) (void)hipFree(p_);
        p_ = nullptr;
        bytes_ = bytes ? bytes : 16;
        hip_check(hipMalloc(&p_, bytes_), "hipMalloc");
    }
    void upload(const void *src, size_t bytes, hipStream_t s)
    {
        resize(bytes);
        if (bytes) hip_check(hipMemcpyAsync(p_, src, bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync H2D");
    }
    void download(void *dst, size_t bytes, hipStream_t s) const
    {
        if (bytes) hip_check(hipMemcpyAsync(dst, p_, bytes, hipMemcpyDeviceToHost, s), "hipMemcpyAsync D2H");
    }
    template <typename T> T *ptr() const { return (T *)p_; }
    size_t bytes() const { return bytes_; }
 private:
    void *p_;
    size_t bytes_;
};

// One per calling thread: device context + stream (the reference runs one filter at a time
// per Filter object, src/Filter.C:76-83, but different streams on different threads).
class WaveContext {
 public:
    explicit WaveContext(int device = 0, uint64_t max_packets = 1 << 21)
    {
        hip_check(hipSetDevice(device), "hipSetDevice");
        hip_check(hipStreamCreate(&stream), "hipStreamCreate");
        ctx = mrpdr_ctx_create(device, max_packets);
        if (!ctx) throw PdrError("mrpdr_ctx_create");
    }
    ~WaveContext()
    {
        mrpdr_ctx_destroy(ctx);
        (void)hipStreamDestroy(stream);
    }
    void sync() { hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize"); }
    mrpdr_ctx *ctx;
    hipStream_t stream;
};

class PacketWave {
 public:
    // ---- API constructor: values -> encoded packets ----------------------------------
    PacketWave(WaveContext &wc, uint32_t stream_id, int32_t tag, const char *fmt,
               const std::vector<HostColumn> &cols, uint64_t n)
        : wc_(wc), fmt_(mrpdr_fmt_compile(fmt)), n_(n), stream_id_(stream_id), tag_(tag),
          byteorder_(MRPDR_LITTLEENDIAN), decoded_(true), error_(false), have_hdrs_(false)
    {
        if (!fmt_) throw PdrError("mrpdr_fmt_compile");
        upload_columns(cols);
        encode_pdr_data();
    }

    // ---- receive-side constructor: encoded packets as Message::recv leaves them --------
    // hdrs: n headers of hdr_len bytes each, back to back.  bufs: the data buffers in one
    // block, packet i at buf_offsets[i], buf_lens[i] long.
    PacketWave(WaveContext &wc, const char *fmt, uint64_t n, const char *hdrs, unsigned hdr_len,
               const char *bufs, const uint64_t *buf_offsets, const uint64_t *buf_lens, uint64_t buf_bytes)
        : wc_(wc), fmt_(mrpdr_fmt_compile(fmt)), n_(n), stream_id_((uint32_t)-1), tag_(-1),
          byteorder_(MRPDR_LITTLEENDIAN), decoded_(false), error_(false), have_hdrs_(true)
    {
        if (!fmt_) throw PdrError("mrpdr_fmt_compile");
        hdr_len_ = hdr_len;
        hdr_region_bytes_ = (uint64_t)n * hdr_len;
        d_hdrs_.upload(hdrs, (size_t)n * hdr_len, wc_.stream);
        d_arena_.upload(bufs, buf_bytes, wc_.stream);
        d_offs_.upload(buf_offsets, 8 * n, wc_.stream);
        d_lens_.upload(buf_lens, 8 * n, wc_.stream);
        arena_bytes_ = buf_bytes;
        decode_pdr_header();
    }

    // ---- receive-side constructor from one Message (src/Message.C:48-199) ------------------
    // msg: the bytes of one message as they came off the link (byte order, packet count, size
    // vector, spare byte, then header and data buffer of every packet).  The host reads the
    // 5-byte count, as Message::recv does before it knows how much more to receive; size
    // vector, headers and data are taken apart on the device, in place.
    PacketWave(WaveContext &wc, const char *fmt, const uint8_t *msg, uint64_t msg_bytes)
        : wc_(wc), fmt_(mrpdr_fmt_compile(fmt)), n_(0), stream_id_((uint32_t)-1), tag_(-1),
          byteorder_(MRPDR_LITTLEENDIAN), decoded_(false), error_(false), have_hdrs_(true)
    {
        if (!fmt_) throw PdrError("mrpdr_fmt_compile");
        if (msg_bytes < 5) throw std::runtime_error("PacketWave: message shorter than its count");
        const bool be = msg[0] != MRPDR_LITTLEENDIAN;           // the count is in the sender's order (:78)
        n_ = be ? ((uint32_t)msg[1] << 24 | (uint32_t)msg[2] << 16 | (uint32_t)msg[3] << 8 | msg[4])
                : ((uint32_t)msg[4] << 24 | (uint32_t)msg[3] << 16 | (uint32_t)msg[2] << 8 | msg[1]);
        const uint64_t pre = mrpdr_message_preamble_bytes(n_);
        if (msg_bytes < pre) throw std::runtime_error("PacketWave: message shorter than its size vector");
        d_msg_.upload(msg, msg_bytes, wc_.stream);
        d_hoffs_.resize(8 * (n_ + 1)); d_hlens_.resize(8 * (n_ ? n_ : 1));
        d_offs_.resize(8 * (n_ + 1));  d_lens_.resize(8 * (n_ ? n_ : 1));
        uint64_t payload = 0;
        if (!mrpdr_message_parse_sizes(wc_.ctx, d_msg_.ptr<uint8_t>() + 5, n_, d_hoffs_.ptr<uint64_t>(),
                                       d_hlens_.ptr<uint64_t>(), d_offs_.ptr<uint64_t>(), d_lens_.ptr<uint64_t>(),
                                       &payload, wc_.stream)) throw PdrError("mrpdr_message_parse_sizes");
        if (pre + payload > msg_bytes) throw std::runtime_error("PacketWave: message shorter than its size vector announces");
        arena_bytes_ = payload;
        in_message_ = true;
        have_hdrs_ = false;             // the headers lie between the data buffers: get_Headers re-encodes them
        msg_pre_ = pre;
        hdr_region_bytes_ = msg_bytes - pre;        // what was received, not what the buffer could hold
        decode_pdr_header();
    }

    // Message::send (src/Message.C:201-335) for this wave: the bytes one write() puts on the link,
    // headers and data encoded at their final place in one device buffer, one copy to the host.
    void to_message(std::vector<uint8_t> &out, const std::vector<uint32_t> *src_ranks = nullptr)
    {
        decode_pdr_data();                                       // a received wave is re-encoded from its columns
        if (!src_ranks && src_ranks_.size() == n_ && n_) src_ranks = &src_ranks_;     // forwarded as received
        const uint64_t h = mrpdr_header_bytes(fmt_);
        const uint64_t pre = mrpdr_message_preamble_bytes(n_);
        DeviceBuffer d_lens, d_offs, d_out, d_src;
        d_lens.resize(8 * (n_ ? n_ : 1)); d_offs.resize(8 * (n_ + 1));
        uint64_t payload = 0;
        if (!mrpdr_wave_sizeof(wc_.ctx, fmt_, cols_.data(), n_, 1, (uint32_t)h, d_lens.ptr<uint64_t>(),
                               d_offs.ptr<uint64_t>(), &payload, wc_.stream)) throw PdrError("mrpdr_wave_sizeof");
        d_out.resize(pre + payload);                             // every byte of it is written below
        uint8_t *m = d_out.ptr<uint8_t>();
        const uint32_t *src = nullptr;
        if (src_ranks && n_) { d_src.upload(src_ranks->data(), 4 * n_, wc_.stream); src = d_src.ptr<uint32_t>(); }
        // preamble and headers in one launch, then the data between the headers
        if (!mrpdr_message_write_frame(wc_.ctx, fmt_, n_, d_lens.ptr<uint64_t>(), d_offs.ptr<uint64_t>(), stream_id_, nullptr,
                                       tag_, nullptr, 0xFFFFFFFFu, src, m, wc_.stream))
            throw PdrError("mrpdr_message_write_frame");
        if (n_) {
            mrpdr_wire w{ m + pre, d_offs.ptr<uint64_t>(), d_lens.ptr<uint64_t>(), 0, payload };
            if (!mrpdr_wave_encode(wc_.ctx, fmt_, cols_.data(), n_, &w, wc_.stream)) throw PdrError("mrpdr_wave_encode");
        }
        out.resize(pre + payload);
        d_out.download(out.data(), pre + payload, wc_.stream);
        wc_.sync();
    }

    ~PacketWave() { mrpdr_fmt_free(fmt_); }
    PacketWave(const PacketWave &) = delete;
    PacketWave &operator=(const PacketWave &) = delete;

    // ---- accessors ---------------------------------------------------------------------
    int32_t get_Tag() const { return tag_; }
    uint32_t get_StreamId() const { return stream_id_; }
    const char *get_FormatString() const { return mrpdr_fmt_string(fmt_); }
    uint64_t size() const { return n_; }
    bool has_Error() const { return error_; }
    const std::vector<uint32_t> &get_SourceRanks() const { return src_ranks_; }
    const std::vector<int32_t> &ok() const { return ok_; }

    // Encoded data buffers in host memory: packet i at offsets[i], lens[i] long.
    void get_Buffers(std::vector<uint8_t> &arena, std::vector<uint64_t> &offsets, std::vector<uint64_t> &lens)
    {
        arena.resize(arena_bytes_);
        offsets.resize(n_);
        lens.resize(n_);
        if (arena_bytes_)
            hip_check(hipMemcpyAsync(arena.data(), wire_.base, arena_bytes_, hipMemcpyDeviceToHost, wc_.stream),
                      "hipMemcpyAsync D2H");
        d_offs_.download(offsets.data(), 8 * n_, wc_.stream);
        d_lens_.download(lens.data(), 8 * n_, wc_.stream);
        wc_.sync();
    }
    unsigned get_HeaderLen() const { return (unsigned)mrpdr_header_bytes(fmt_); }
    // Encoded headers, get_HeaderLen() bytes each (lazy: Message::send asks for them).
    void get_Headers(std::vector<uint8_t> &hdrs, const std::vector<uint32_t> *src_ranks = nullptr)
    {
        if (!src_ranks && src_ranks_.size() == n_ && n_) src_ranks = &src_ranks_;     // as received
        if (!have_hdrs_) encode_pdr_header(src_ranks);
        hdrs.resize((size_t)n_ * hdr_len_);
        d_hdrs_.download(hdrs.data(), hdrs.size(), wc_.stream);
        wc_.sync();
    }

    // Packet::unpack: -1 when the format differs, else the decoded fields in host memory
    // (caller-owned copies, as DataElementArray2ArgList hands out, src/Packet.C:1065-1431).
    int unpack(const char *fmt, std::vector<HostColumn> &out)
    {
        decode_pdr_data();
        if (!fmt || strcmp(fmt, get_FormatString()) != 0) return -1;
        const int nf = mrpdr_fmt_nfields(fmt_);
        out.assign(nf, HostColumn());
        for (int f = 0; f < nf; f++) {
            const int w = mrpdr_fmt_field_width(fmt_, f);
            if (mrpdr_fmt_field_kind(fmt_, f) == MRPDR_SCALAR) {
                out[f].values.resize(n_ * w);
                d_values_[f].download(out[f].values.data(), n_ * w, wc_.stream);
            } else {
                out[f].counts.resize(n_);
                d_counts_[f].download(out[f].counts.data(), 8 * n_, wc_.stream);
                const bool sa = mrpdr_fmt_field_kind(fmt_, f) == MRPDR_STRING_ARRAY;
                const uint64_t elems = sa ? totals_[nf + f] : totals_[f];
                out[f].values.resize(elems * w);
                d_values_[f].download(out[f].values.data(), elems * w, wc_.stream);
                if (sa) {
                    out[f].sub_counts.resize(totals_[f]);
                    d_subc_[f].download(out[f].sub_counts.data(), 8 * totals_[f], wc_.stream);
                }
            }
        }
        wc_.sync();
        return 0;
    }

    // Packet::unpack for the whole wave WITHOUT the copies (mrpdr_wave_view): the fields of all
    // packets as views of the decoded device columns, under unpack's format check (-1 when fmt is
    // not the wave's format string).  The views alias device memory this PacketWave owns: they are
    // valid until it is destroyed (consumers first, the wave last), nothing is copied, allocated or
    // synchronised, and work that reads them goes on context().stream behind the decode.  unpack()
    // above stays what the reference's unpack is: caller-owned host copies.
    int view(const char *fmt, std::vector<mrpdr_field_view> &out)
    {
        decode_pdr_data();
        if (!fmt) return -1;
        out.assign((size_t)mrpdr_fmt_nfields(fmt_), mrpdr_field_view());
        mrpdr_field_view none;
        return mrpdr_wave_view(fmt_, fmt, cols_.data(), n_, out.empty() ? &none : out.data()) ? 0 : -1;
    }

    // device views for the filters
    const mrpdr_wire &wire() const { return wire_; }
    const mrpdr_fmt *format() const { return fmt_; }
    int32_t byteorder() const { return byteorder_; }
    const std::vector<mrpdr_column> &columns() { decode_pdr_data(); return cols_; }
    WaveContext &context() const { return wc_; }
    // elements of field f over the whole wave (strings of a string array)
    uint64_t total(int f) { decode_pdr_data(); return totals_[f]; }

 private:
    void alloc_column_structs()
    {
        const int nf = mrpdr_fmt_nfields(fmt_);
        d_values_ = std::vector<DeviceBuffer>(nf);
        d_counts_ = std::vector<DeviceBuffer>(nf);
        d_coffs_ = std::vector<DeviceBuffer>(nf);
        d_subc_ = std::vector<DeviceBuffer>(nf);
        d_subo_ = std::vector<DeviceBuffer>(nf);
        cols_.assign(nf, mrpdr_column());
        totals_.assign(2 * nf, 0);      // [f] elements (strings for %as), [nf + f] bytes of %as
    }

    void upload_columns(const std::vector<HostColumn> &cols)
    {
        const int nf = mrpdr_fmt_nfields(fmt_);
        if ((int)cols.size() != nf) throw std::runtime_error("PacketWave: one column per format token");
        alloc_column_structs();
        for (int f = 0; f < nf; f++) {
            const int w = mrpdr_fmt_field_width(fmt_, f);
            d_values_[f].upload(cols[f].values.data(), cols[f].values.size(), wc_.stream);
            cols_[f].values = d_values_[f].ptr<void>();
            cols_[f].capacity = cols[f].values.size() / w;
            if (mrpdr_fmt_field_kind(fmt_, f) != MRPDR_SCALAR) {
                std::vector<uint64_t> offs(n_ + 1, 0);
                for (uint64_t i = 0; i < n_; i++) offs[i + 1] = offs[i] + cols[f].counts[i];
                totals_[f] = offs[n_];
                d_counts_[f].upload(cols[f].counts.data(), 8 * n_, wc_.stream);
                d_coffs_[f].upload(offs.data(), 8 * (n_ + 1), wc_.stream);
                wc_.sync();     // offs is a local
                cols_[f].counts = d_counts_[f].ptr<uint64_t>();
                cols_[f].offs = d_coffs_[f].ptr<uint64_t>();
                if (mrpdr_fmt_field_kind(fmt_, f) == MRPDR_STRING_ARRAY) {
                    const uint64_t ns = offs[n_];
                    if (cols[f].sub_counts.size() != ns) throw std::runtime_error("PacketWave: sub_counts size");
                    std::vector<uint64_t> so(ns + 1, 0);
                    for (uint64_t j = 0; j < ns; j++) so[j + 1] = so[j] + cols[f].sub_counts[j];
                    totals_[nf + f] = so[ns];
                    d_subc_[f].upload(cols[f].sub_counts.data(), 8 * ns, wc_.stream);
                    d_subo_[f].upload(so.data(), 8 * (ns + 1), wc_.stream);
                    wc_.sync();
                    cols_[f].sub_counts = d_subc_[f].ptr<uint64_t>();
                    cols_[f].sub_offs = d_subo_[f].ptr<uint64_t>();
                    cols_[f].sub_capacity = ns;
                }
            }
        }
    }

    void encode_pdr_data()                       // src/Packet.C:54-85, for the whole wave
    {
        d_offs_.resize(8 * (n_ + 1));
        d_lens_.resize(8 * n_);
        uint64_t total = 0;
        if (!mrpdr_wave_sizeof(wc_.ctx, fmt_, cols_.data(), n_, 16, 0, d_lens_.ptr<uint64_t>(),
                               d_offs_.ptr<uint64_t>(), &total, wc_.stream)) throw PdrError("mrpdr_wave_sizeof");
        arena_bytes_ = total;
        d_arena_.resize(total);
        hip_check(hipMemsetAsync(d_arena_.ptr<void>(), 0, d_arena_.bytes(), wc_.stream), "hipMemsetAsync");
        wire_ = mrpdr_wire{ d_arena_.ptr<uint8_t>(), d_offs_.ptr<uint64_t>(), d_lens_.ptr<uint64_t>(), 0, total };
        if (!mrpdr_wave_encode(wc_.ctx, fmt_, cols_.data(), n_, &wire_, wc_.stream)) throw PdrError("mrpdr_wave_encode");
        uint32_t status = 0;
        if (!mrpdr_fmt_is_fixed(fmt_)) {
            if (!mrpdr_ctx_status(wc_.ctx, wc_.stream, &status)) throw PdrError("mrpdr_ctx_status");
            if (status) error_ = true;      // e.g. a NULL string: error(ERR_PACKING, ...), src/Packet.C:70-72
        }
    }

    void encode_pdr_header(const std::vector<uint32_t> *src_ranks)      // src/Packet.C:20-52
    {
        hdr_len_ = get_HeaderLen();
        d_hdrs_.resize((size_t)n_ * hdr_len_);
        DeviceBuffer d_src;
        const uint32_t *src = nullptr;
        if (src_ranks) { d_src.upload(src_ranks->data(), 4 * n_, wc_.stream); src = d_src.ptr<uint32_t>(); }
        if (!mrpdr_wave_encode_headers(wc_.ctx, fmt_, n_, stream_id_, nullptr, tag_, nullptr, 0xFFFFFFFFu, src,
                                       nullptr, 0, d_hdrs_.ptr<uint8_t>(), nullptr, hdr_len_, wc_.stream)) throw PdrError("mrpdr_wave_encode_headers");
        wc_.sync();
        have_hdrs_ = true;
    }

    void decode_pdr_header()                     // src/Packet.C:87-104
    {
        DeviceBuffer d_sid, d_tag, d_src, d_bo, d_ok;
        d_sid.resize(4 * n_); d_tag.resize(4 * n_); d_src.resize(4 * n_); d_bo.resize(n_); d_ok.resize(4 * n_);
        const uint8_t *hdr_base = in_message_ ? d_msg_.ptr<uint8_t>() + msg_pre_ : d_hdrs_.ptr<uint8_t>();
        // the bound for untrusted header offsets is the length of what was uploaded (a DeviceBuffer's
        // own size is its capacity: at least 16 bytes, and it never shrinks)
        const uint64_t hdr_bytes = hdr_region_bytes_;
        if (!mrpdr_wave_decode_headers(wc_.ctx, fmt_, n_, hdr_base, hdr_bytes, in_message_ ? d_hoffs_.ptr<uint64_t>() : nullptr,
                                       in_message_ ? d_hlens_.ptr<uint64_t>() : nullptr, hdr_len_,
                                       d_sid.ptr<uint32_t>(), d_tag.ptr<int32_t>(), d_src.ptr<uint32_t>(),
                                       d_bo.ptr<uint8_t>(), nullptr, nullptr, 0, d_ok.ptr<int32_t>(), wc_.stream))
            throw PdrError("mrpdr_wave_decode_headers");
        std::vector<uint32_t> sid(n_);
        std::vector<int32_t> tag(n_), okh(n_);
        std::vector<uint8_t> bo(n_);
        src_ranks_.resize(n_);
        d_sid.download(sid.data(), 4 * n_, wc_.stream); d_tag.download(tag.data(), 4 * n_, wc_.stream);
        d_src.download(src_ranks_.data(), 4 * n_, wc_.stream); d_bo.download(bo.data(), n_, wc_.stream);
        d_ok.download(okh.data(), 4 * n_, wc_.stream);
        wc_.sync();
        if (n_) { stream_id_ = sid[0]; tag_ = tag[0]; byteorder_ = bo[0]; }
        for (uint64_t i = 0; i < n_; i++) {
            if (!okh[i]) error_ = true;                       // a packet of another format
            if (bo[i] != bo[0]) mixed_ = true;                // children of both kinds: each packet keeps its order
        }
        if (mixed_) d_orders_.upload(bo.data(), n_, wc_.stream), wc_.sync();
        uint8_t *arena = in_message_ ? d_msg_.ptr<uint8_t>() + msg_pre_ : d_arena_.ptr<uint8_t>();
        wire_ = mrpdr_wire{ arena, d_offs_.ptr<uint64_t>(), d_lens_.ptr<uint64_t>(), 0, arena_bytes_ };
    }

    void decode_pdr_data()                       // src/Packet.C:106-133, lazy and idempotent
    {
        if (decoded_) return;
        const int nf = mrpdr_fmt_nfields(fmt_);
        alloc_column_structs();
        for (int f = 0; f < nf; f++) {
            const int w = mrpdr_fmt_field_width(fmt_, f);
            if (mrpdr_fmt_field_kind(fmt_, f) == MRPDR_SCALAR) {
                d_values_[f].resize(n_ * w);
                cols_[f].values = d_values_[f].ptr<void>();
                cols_[f].capacity = n_;
            } else {
                d_counts_[f].resize(8 * n_);
                d_coffs_[f].resize(8 * (n_ + 1));
                cols_[f].counts = d_counts_[f].ptr<uint64_t>();
                cols_[f].offs = d_coffs_[f].ptr<uint64_t>();
            }
        }
        DeviceBuffer d_ok;
        d_ok.resize(4 * n_);
        const int began = mixed_
            ? mrpdr_wave_decode_mixed_begin(wc_.ctx, fmt_, &wire_, n_, d_orders_.ptr<uint8_t>(), cols_.data(),
                                            d_ok.ptr<int32_t>(), totals_.data(), wc_.stream)
            : mrpdr_wave_decode_begin(wc_.ctx, fmt_, &wire_, n_, byteorder_, cols_.data(), d_ok.ptr<int32_t>(),
                                      totals_.data(), wc_.stream);
        if (!began) throw PdrError("mrpdr_wave_decode_begin");
        for (int f = 0; f < nf; f++)
            if (mrpdr_fmt_field_kind(fmt_, f) != MRPDR_SCALAR) {
                const bool sa = mrpdr_fmt_field_kind(fmt_, f) == MRPDR_STRING_ARRAY;
                const uint64_t elems = sa ? totals_[nf + f] : totals_[f];
                d_values_[f].resize(elems * mrpdr_fmt_field_width(fmt_, f));
                cols_[f].values = d_values_[f].ptr<void>();
                cols_[f].capacity = elems;
                if (sa) {
                    d_subc_[f].resize(8 * totals_[f]);
                    d_subo_[f].resize(8 * (totals_[f] + 1));
                    cols_[f].sub_counts = d_subc_[f].ptr<uint64_t>();
                    cols_[f].sub_offs = d_subo_[f].ptr<uint64_t>();
                    cols_[f].sub_capacity = totals_[f];
                }
            }
        const int finished = mixed_
            ? mrpdr_wave_decode_mixed_finish(wc_.ctx, fmt_, &wire_, n_, d_orders_.ptr<uint8_t>(), cols_.data(), wc_.stream)
            : mrpdr_wave_decode_finish(wc_.ctx, fmt_, &wire_, n_, byteorder_, cols_.data(), wc_.stream);
        if (!finished) throw PdrError("mrpdr_wave_decode_finish");
        ok_.resize(n_);
        d_ok.download(ok_.data(), 4 * n_, wc_.stream);
        wc_.sync();
        for (uint64_t i = 0; i < n_; i++)
            if (!ok_[i]) error_ = true;          // error(ERR_PACKING, ...), src/Packet.C:122-124
        decoded_ = true;
    }

    WaveContext &wc_;
    mrpdr_fmt *fmt_;
    uint64_t n_;
    uint32_t stream_id_;
    int32_t tag_;
    int32_t byteorder_;
    bool decoded_, error_, have_hdrs_;
    bool mixed_ = false;            // the wave holds packets of both sender byte orders
    unsigned hdr_len_ = 0;
    uint64_t arena_bytes_ = 0;
    DeviceBuffer d_hdrs_, d_arena_, d_offs_, d_lens_, d_orders_;
    DeviceBuffer d_msg_, d_hoffs_, d_hlens_;   // a wave received as one Message: decoded in place
    bool in_message_ = false;
    uint64_t msg_pre_ = 0;
    uint64_t hdr_region_bytes_ = 0; // bytes of the uploaded header block / of the message behind its preamble
    std::vector<DeviceBuffer> d_values_, d_counts_, d_coffs_, d_subc_, d_subo_;
    std::vector<mrpdr_column> cols_;
    std::vector<uint64_t> totals_;
    std::vector<uint32_t> src_ranks_;
    std::vector<int32_t> ok_;
    mrpdr_wire wire_{};
};

// ---- waves of several formats -----------------------------------------------------------------
// The built-in filters only assert that a wave has one format (src/FilterDefinitions.C:104-105);
// a batched codec has to split first.  Host-side, on the header bytes as Message::recv left
// them: the format string sits behind four 32-bit words (src/Packet.C:557-575), its length in
// the header's own byte order (the last byte, src/Packet.C:96).  Returns the packet indices of
// every format in arrival order; headers that cannot hold a format string go under "".
inline std::vector<std::pair<std::string, std::vector<uint64_t> > >
group_by_format(const uint8_t *hdrs, uint64_t hdrs_bytes, const uint64_t *hdr_offsets, const uint64_t *hdr_lens, uint64_t n)
{
    std::vector<std::pair<std::string, std::vector<uint64_t> > > groups;
    for (uint64_t i = 0; i < n; i++) {
        std::string key;
        const uint64_t o = hdr_offsets[i], l = hdr_lens[i];
        if (l >= 16 + 1 + 8 + 1 && o <= hdrs_bytes && l <= hdrs_bytes - o) {
            const uint8_t *h = hdrs + o;
            const bool be = h[l - 1] != MRPDR_LITTLEENDIAN;
            const uint32_t flen = be ? ((uint32_t)h[12] << 24 | (uint32_t)h[13] << 16 | (uint32_t)h[14] << 8 | h[15])
                                     : ((uint32_t)h[15] << 24 | (uint32_t)h[14] << 16 | (uint32_t)h[13] << 8 | h[12]);
            if (flen >= 1 && flen <= l - 16 - 8 - 1 && h[16 + flen - 1] == 0) key.assign((const char *)h + 16, flen - 1);
        }
        size_t g = 0;
        while (g < groups.size() && groups[g].first != key) g++;
        if (g == groups.size()) groups.push_back(std::make_pair(key, std::vector<uint64_t>()));
        groups[g].second.push_back(i);
    }
    return groups;
}

// ---- built-in transformation filters over a wave ------------------------------------------
// Returns 0 and the data bytes of the output packet (T value[, int32 count]) in result16, or -1
// when the filter emits nothing (format it does not accept, empty wave, zero count sum).
inline int run_tfilter(int op, PacketWave &in, int shape, uint8_t result16[16], std::string *out_fmt)
{
    const mrpdr_fmt *f = in.format();
    const int nf = mrpdr_fmt_nfields(f);
    const int want = op == MRPDR_AVG ? 2 : 1;
    if (in.size() == 0 || in.has_Error() || nf != want || mrpdr_fmt_field_kind(f, 0) != MRPDR_SCALAR) return -1;
    if (op == MRPDR_AVG && mrpdr_fmt_field_type(f, 1) != MRPDR_INT32_T) return -1;
    const std::vector<mrpdr_column> &cols = in.columns();
    WaveContext &wc = in.context();
    DeviceBuffer d_res, d_ok;
    d_res.resize(16); d_ok.resize(4);
    const int w = mrpdr_fmt_field_width(f, 0);
    if (!mrpdr_wave_filter(wc.ctx, op, mrpdr_fmt_field_type(f, 0), shape, cols[0].values, (uint64_t)w,
                           op == MRPDR_AVG ? cols[1].values : nullptr, 4, in.size(), 0,
                           d_res.ptr<void>(), d_ok.ptr<int32_t>(), wc.stream)) throw PdrError("mrpdr_wave_filter");
    int32_t ok = 0;
    d_res.download(result16, 16, wc.stream);
    d_ok.download(&ok, 4, wc.stream);
    wc.sync();
    if (out_fmt) *out_fmt = in.get_FormatString();
    return ok ? 0 : -1;
}
inline int tfilter_Sum(PacketWave &in, uint8_t r[16], int shape = MRPDR_SHAPE_FLAT) { return run_tfilter(MRPDR_SUM, in, shape, r, nullptr); }
inline int tfilter_Max(PacketWave &in, uint8_t r[16], int shape = MRPDR_SHAPE_FLAT) { return run_tfilter(MRPDR_MAX, in, shape, r, nullptr); }
inline int tfilter_Min(PacketWave &in, uint8_t r[16], int shape = MRPDR_SHAPE_FLAT) { return run_tfilter(MRPDR_MIN, in, shape, r, nullptr); }
inline int tfilter_Avg(PacketWave &in, uint8_t r[16], int shape = MRPDR_SHAPE_FLAT) { return run_tfilter(MRPDR_AVG, in, shape, r, nullptr); }


// ---- built-in array filters ------------------------------------------------------------------
// Each returns 0 and the fields of the ONE output packet as host columns (what unpack of that
// packet hands out), or -1 when the filter emits nothing.

// The column of a one-packet wave, brought to the host.
inline void download_single(WaveContext &wc, const mrpdr_column &c, int width, bool string_array, HostColumn &out)
{
    out = HostColumn();
    out.counts.resize(1);
    hip_check(hipMemcpyAsync(out.counts.data(), c.counts, 8, hipMemcpyDeviceToHost, wc.stream), "hipMemcpyAsync D2H");
    wc.sync();
    uint64_t bytes = out.counts[0] * (uint64_t)width;
    if (string_array) {
        out.sub_counts.resize(out.counts[0]);
        if (out.counts[0]) {
            hip_check(hipMemcpyAsync(out.sub_counts.data(), c.sub_counts, 8 * out.counts[0], hipMemcpyDeviceToHost, wc.stream),
                      "hipMemcpyAsync D2H");
            wc.sync();
        }
        bytes = 0;
        for (uint64_t l : out.sub_counts) bytes += l;
    }
    out.values.resize(bytes);
    if (bytes) hip_check(hipMemcpyAsync(out.values.data(), c.values, bytes, hipMemcpyDeviceToHost, wc.stream), "hipMemcpyAsync D2H");
    wc.sync();
}

// tfilter_ArrayConcat (src/FilterDefinitions.C:649-810): every input array in packet order.
inline int tfilter_ArrayConcat(PacketWave &in, HostColumn &out)
{
    const mrpdr_fmt *f = in.format();
    if (in.size() == 0 || in.has_Error() || mrpdr_fmt_nfields(f) != 1) return -1;
    const int kind = mrpdr_fmt_field_kind(f, 0);
    if (kind != MRPDR_ARRAY && kind != MRPDR_STRING_ARRAY) return -1;   // "invalid packet type", :764-768
    const std::vector<mrpdr_column> &cols = in.columns();
    if (in.has_Error()) return -1;
    WaveContext &wc = in.context();
    DeviceBuffer d_cnt, d_offs;
    d_cnt.resize(8); d_offs.resize(16);
    mrpdr_column o = cols[0];                       // the values column of the wave IS the output array
    o.counts = d_cnt.ptr<uint64_t>();
    o.offs = d_offs.ptr<uint64_t>();
    if (!mrpdr_wave_array_concat(wc.ctx, &cols[0], in.size(), &o, wc.stream)) throw PdrError("mrpdr_wave_array_concat");
    download_single(wc, o, mrpdr_fmt_field_width(f, 0), kind == MRPDR_STRING_ARRAY, out);
    return 0;
}

// tfilter_PerfData's concatenating aggregation takes this id beside MRPDR_SUM/MIN/MAX/AVG
// (the reference compares aggr_id with the registered id of TFILTER_ARRAY_CONCAT,
// src/FilterDefinitions.C:1000).
enum { PERFDATA_ARRAY_CONCAT = 6 };

// tfilter_PerfData (src/FilterDefinitions.C:904-1239) over "%ad %ad %a{uld,ld,lf}" packets:
// out = (rank array, nelems array, data array) of the output packet.  group > 0 declares a
// two-level tree (nodes of `group` children, then one node over their outputs).
inline int tfilter_PerfData(PacketWave &in, int aggr_id, std::vector<HostColumn> &out, uint64_t group = 0)
{
    const char *fmt = in.get_FormatString();
    int type;
    if (!strcmp(fmt, "%ad %ad %auld")) type = MRPDR_UINT64_T;
    else if (!strcmp(fmt, "%ad %ad %ald")) type = MRPDR_INT64_T;
    else if (!strcmp(fmt, "%ad %ad %alf")) type = MRPDR_DOUBLE_T;
    else return -1;
    const uint64_t n = in.size();
    if (n == 0 || in.has_Error()) return -1;
    const std::vector<mrpdr_column> &cols = in.columns();
    if (in.has_Error()) return -1;
    WaveContext &wc = in.context();
    out.assign(3, HostColumn());
    if (aggr_id == PERFDATA_ARRAY_CONCAT) {
        const uint32_t widths[3] = { 4, 4, 8 };
        for (int f = 0; f < 3; f++) {
            DeviceBuffer d_val, d_cnt, d_offs;
            const uint64_t cap = in.total(f) ? in.total(f) : 1;
            d_val.resize(cap * widths[f]); d_cnt.resize(8); d_offs.resize(16);
            mrpdr_column o = mrpdr_column();
            o.values = d_val.ptr<void>(); o.counts = d_cnt.ptr<uint64_t>(); o.offs = d_offs.ptr<uint64_t>();
            o.capacity = cap;
            if (!mrpdr_wave_array_concat_masked(wc.ctx, &cols[f], widths[f], cols[2].counts, n, &o, wc.stream))
                throw PdrError("mrpdr_wave_array_concat_masked");
            download_single(wc, o, (int)widths[f], false, out[f]);
        }
        return 0;
    }
    if (aggr_id != MRPDR_SUM && aggr_id != MRPDR_MIN && aggr_id != MRPDR_MAX && aggr_id != MRPDR_AVG) return -1;
    // integer SUM / MIN / MAX do not depend on the fold order: a large node is computed as a
    // two-level tree, same bits, without a serial chain over all n packets
    if (group == 0 && n > 2048 && aggr_id != MRPDR_AVG && type != MRPDR_DOUBLE_T) group = 256;
    std::vector<uint64_t> lens(n);
    hip_check(hipMemcpyAsync(lens.data(), cols[2].counts, 8 * n, hipMemcpyDeviceToHost, wc.stream), "hipMemcpyAsync D2H");
    wc.sync();
    uint64_t cap = 0;
    for (uint64_t l : lens) cap = l > cap ? l : cap;
    DeviceBuffer d_out, d_len, d_rank, d_scratch;
    d_out.resize(8 * (cap ? cap : 1)); d_len.resize(8); d_rank.resize(4);
    const uint64_t need = mrpdr_perfdata_scratch_bytes(n, group, cap);
    if (need) d_scratch.resize(need);
    if (!mrpdr_wave_perfdata(wc.ctx, aggr_id, type, &cols[0], &cols[2], n, group, d_out.ptr<void>(), cap,
                             d_len.ptr<uint64_t>(), d_rank.ptr<int32_t>(), need ? d_scratch.ptr<void>() : nullptr, need,
                             wc.stream)) throw PdrError("mrpdr_wave_perfdata");
    uint64_t len = 0;
    int32_t rank = 0;
    d_len.download(&len, 8, wc.stream);
    d_rank.download(&rank, 4, wc.stream);
    wc.sync();
    const int32_t nelems = (int32_t)len;
    out[0].counts.assign(1, 1); out[0].values.assign((const uint8_t *)&rank, (const uint8_t *)(&rank + 1));
    out[1].counts.assign(1, 1); out[1].values.assign((const uint8_t *)&nelems, (const uint8_t *)(&nelems + 1));
    out[2].counts.assign(1, len);
    out[2].values.resize(8 * len);
    d_out.download(out[2].values.data(), 8 * len, wc.stream);
    wc.sync();
    return 0;
}

// tfilter_IntEqClass (src/FilterDefinitions.C:812-902) over "%aud %aud %aud" packets: the
// classes of all inputs merged by value.  out = (values, members per class, members).
inline int tfilter_IntEqClass(PacketWave &in, std::vector<HostColumn> &out)
{
    if (strcmp(in.get_FormatString(), "%aud %aud %aud") != 0 || in.size() == 0 || in.has_Error()) return -1;
    const std::vector<mrpdr_column> &cols = in.columns();
    if (in.has_Error()) return -1;
    WaveContext &wc = in.context();
    const uint64_t kcap = in.total(0) ? in.total(0) : 1, mcap = in.total(2) ? in.total(2) : 1;
    const uint64_t caps[3] = { kcap, kcap, mcap };
    DeviceBuffer d_val[3], d_cnt[3], d_offs[3], d_scratch;
    mrpdr_column o[3];
    for (int k = 0; k < 3; k++) {
        d_val[k].resize(4 * caps[k]); d_cnt[k].resize(8); d_offs[k].resize(16);
        o[k] = mrpdr_column();
        o[k].values = d_val[k].ptr<void>(); o[k].counts = d_cnt[k].ptr<uint64_t>(); o[k].offs = d_offs[k].ptr<uint64_t>();
        o[k].capacity = caps[k];
    }
    const uint64_t need = mrpdr_eqclass_scratch_bytes(cols[0].capacity);
    d_scratch.resize(need);
    if (!mrpdr_wave_int_eq_class(wc.ctx, &cols[0], &cols[1], &cols[2], in.size(), &o[0], &o[1], &o[2],
                                 d_scratch.ptr<void>(), need, wc.stream)) throw PdrError("mrpdr_wave_int_eq_class");
    out.assign(3, HostColumn());
    for (int k = 0; k < 3; k++) download_single(wc, o[k], 4, false, out[k]);
    return 0;
}

} // namespace MRN_GPU
