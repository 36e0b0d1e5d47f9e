// Runs integration/mrpdr_filters.C -- the MRNet filter plugin -- without MRNet: the few Packet
// methods the plugin calls are implemented here over plain host buffers laid out as
// Packet::pdr_packet_header / pdr_packet_data lay them out (src/Packet.C:557-845), waves are handed
// to mrpdr_Sum / Min / Max / Avg / ArrayConcat the way Filter::push_Packets does (src/Filter.C:42-117),
// and the output packet is compared with the reference fold (concatenation) written out on the host
// (src/FilterDefinitions.C:100-149, 517-561, 649-810).  Prints MRNTEST_SUCCESS / MRNTEST_FAILURE lines like
// tests/test_common.C:41-187.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <type_traits>
#include <vector>

#include "../../integration/mrpdr_filters.C"

namespace MRN {

struct Packet::Impl {
    unsigned stream_id; int tag; std::string fmt;
    std::vector<char> hdr, buf;
    void *array; uint64_t array_len; bool array_of_strings, destroy;      // what the array constructor was given
    Impl() : stream_id(0), tag(0), array(NULL), array_len(0), array_of_strings(false), destroy(false) {}
};

static size_t token_width(const std::string &t)
{
    if (t == "c" || t == "uc") return 1;
    if (t == "hd" || t == "uhd") return 2;
    if (t == "d" || t == "ud" || t == "f") return 4;
    return 8;
}

static std::vector<std::string> tokens(const char *fmt)
{
    std::vector<std::string> out;
    std::string cur;
    for (const char *p = fmt;; p++) {
        if (*p == 0 || *p == ' ' || *p == '%' || *p == '\t' || *p == '\n') {
            if (!cur.empty()) out.push_back(cur);
            cur.clear();
            if (*p == 0) break;
        } else cur.push_back(*p);
    }
    return out;
}

// Packet(stream, tag, const void** data, fmt): scalars only, native order (what the plugin builds)
Packet::Packet(unsigned int istream_id, int itag, const void **idata, const char *ifmt) : impl_(new Impl())
{
    impl_->stream_id = istream_id; impl_->tag = itag; impl_->fmt = ifmt;
    size_t k = 0;
    for (const std::string &t : tokens(ifmt)) {
        const char *src = (const char *)idata[k++];
        impl_->buf.insert(impl_->buf.end(), src, src + token_width(t));
    }
}
// Packet(stream, tag, "%a<T> ..." | "%A<T>" | "%as", array, length, ...): array fields, each laid out
// as pdr_array does (u64 count, then the elements; a string as u32 strlen+1 and its bytes, NUL included)
Packet::Packet(unsigned int istream_id, int itag, const char *ifmt, ...) : impl_(new Impl())
{
    impl_->stream_id = istream_id; impl_->tag = itag; impl_->fmt = ifmt;
    va_list ap;
    va_start(ap, ifmt);
    for (const std::string &t : tokens(ifmt)) {
        void *arr = va_arg(ap, void *);
        const uint64_t len = t[0] == 'A' ? va_arg(ap, uint64_t) : (uint64_t)va_arg(ap, unsigned);
        const bool strs = t.substr(1) == "s";
        if (!impl_->array) { impl_->array = arr; impl_->array_len = len; impl_->array_of_strings = strs; }
        impl_->buf.insert(impl_->buf.end(), (const char *)&len, (const char *)&len + 8);
        if (strs) {
            for (uint64_t j = 0; j < len; j++) {
                const char *str = ((char **)arr)[j];
                const uint32_t l = (uint32_t)strlen(str) + 1;
                impl_->buf.insert(impl_->buf.end(), (const char *)&l, (const char *)&l + 4);
                impl_->buf.insert(impl_->buf.end(), str, str + l);
            }
        } else {
            impl_->buf.insert(impl_->buf.end(), (const char *)arr, (const char *)arr + len * token_width(t.substr(1)));
        }
    }
    va_end(ap);
}
void Packet::set_DestroyData(bool b) { impl_->destroy = b; }
Packet::~Packet()
{
    if (impl_->destroy && impl_->array) {
        if (impl_->array_of_strings)
            for (uint64_t j = 0; j < impl_->array_len; j++) free(((char **)impl_->array)[j]);
        free(impl_->array);
    }
    delete impl_;
}
int Packet::get_Tag(void) const { return impl_->tag; }
unsigned int Packet::get_StreamId(void) const { return impl_->stream_id; }
const char *Packet::get_FormatString(void) const { return impl_->fmt.c_str(); }
const char *Packet::get_Buffer(void) const { return impl_->buf.data(); }
uint64_t Packet::get_BufferLen(void) const { return impl_->buf.size(); }
const char *Packet::get_Header(void) const { return impl_->hdr.empty() ? NULL : impl_->hdr.data(); }
unsigned int Packet::get_HeaderLen(void) const { return (unsigned)impl_->hdr.size(); }

class TopologyLocalInfo {};

} // namespace MRN

static int failures = 0;

static void end_SubTest(const std::string &name, bool ok)
{
    printf("%s: %s\n", ok ? "MRNTEST_SUCCESS" : "MRNTEST_FAILURE", name.c_str());
    if (!ok) failures++;
}

// A packet as it looks after Message::recv from a sender of byte order `bo`: header ending in the
// order flag, data bytes reversed field by field when the sender was big-endian.
static PacketPtr received(unsigned stream, int tag, const char *fmt, const void **fields, int bo)
{
    PacketPtr p(new Packet(stream, tag, fields, fmt));
    if (bo) {
        size_t at = 0;
        for (const std::string &t : MRN::tokens(fmt)) {
            const size_t w = MRN::token_width(t);
            for (size_t b = 0; b < w / 2; b++) std::swap(p->impl_->buf[at + b], p->impl_->buf[at + w - 1 - b]);
            at += w;
        }
    }
    p->impl_->hdr.assign(16 + strlen(fmt) + 1 + 8 + 1, 0);
    p->impl_->hdr.back() = (char)bo;
    return p;
}

template <typename T> static void test_type(const char *fmt, uint64_t n, int order_mode)
{
    std::vector<PacketPtr> in, out, rev;
    std::vector<T> vals(n);
    std::vector<int32_t> cnts(n);
    const std::string avg_fmt = std::string(fmt) + " %d";
    uint64_t x = 88172645463325252ull + n;
    for (uint64_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        if (std::is_floating_point<T>::value) vals[i] = (T)((double)(int64_t)(x % 2000001) / 1e3 - 1000.0) * (T)((i % 3) ? 1 : 1000);
        else vals[i] = (T)(x >> 9);
        cnts[i] = (int32_t)(x >> 40) % 5 + 1;
    }
    MRN::TopologyLocalInfo here, *info = &here;
    PacketPtr params;
    for (int op = 0; op < 4; op++) {
        in.clear(); out.clear();
        for (uint64_t i = 0; i < n; i++) {
            const void *f[2] = { &vals[i], &cnts[i] };
            const int bo = order_mode == 2 ? (int)(i % 3 == 1) : order_mode;
            in.push_back(received(1073741834u, 100 + op, op == 3 ? avg_fmt.c_str() : fmt, f, bo));
        }
        void *state = NULL;
        T want; int32_t want_cnt = 0;
        if (op == 0) { want = 0; for (uint64_t i = 0; i < n; i++) want = (T)(want + vals[i]); mrpdr_Sum(in, out, rev, &state, params, *info); }
        else if (op == 1) { want = vals[0]; for (uint64_t i = 1; i < n; i++) want = (want < vals[i]) ? want : vals[i]; mrpdr_Min(in, out, rev, &state, params, *info); }
        else if (op == 2) { want = vals[0]; for (uint64_t i = 1; i < n; i++) want = (want > vals[i]) ? want : vals[i]; mrpdr_Max(in, out, rev, &state, params, *info); }
        else {
            want = 0;
            for (uint64_t i = 0; i < n; i++) { want = (T)(want + (T)(vals[i] * (T)cnts[i])); want_cnt += cnts[i]; }
            want = (T)(want / (T)want_cnt);
            mrpdr_Avg(in, out, rev, &state, params, *info);
        }
        // the same filter state again (second wave of the stream), then release it
        std::vector<PacketPtr> out2;
        if (op == 0) mrpdr_Sum(in, out2, rev, &state, params, *info);
        bool ok = out.size() == 1 && out[0]->get_StreamId() == 1073741834u && out[0]->get_Tag() == 100 + op &&
                  !strcmp(out[0]->get_FormatString(), op == 3 ? avg_fmt.c_str() : fmt) &&
                  out[0]->get_BufferLen() == sizeof(T) + (op == 3 ? 4 : 0) &&
                  memcmp(out[0]->get_Buffer(), &want, sizeof(T)) == 0;
        if (ok && op == 3) ok = memcmp(out[0]->get_Buffer() + sizeof(T), &want_cnt, 4) == 0;
        if (ok && op == 0) ok = out2.size() == 1 && memcmp(out2[0]->get_Buffer(), &want, sizeof(T)) == 0;
        release(static_cast<GpuFilterState *>(state));
        static const char *names[4] = { "mrpdr_Sum", "mrpdr_Min", "mrpdr_Max", "mrpdr_Avg" };
        end_SubTest(std::string(names[op]) + "(" + fmt + ", " + std::to_string(n) + " packets, orders " +
                    (order_mode == 2 ? "mixed" : order_mode ? "big" : "little") + ")", ok);
    }
}

// A wave with one packet of another format is refused as a whole, as the built-ins assert one format per
// wave (src/FilterDefinitions.C:104-105) -- wherever the odd packet lies, also in a helper thread's part
// of a large wave; the stream's state is good for the next, well-formed wave.
static void test_odd_packet(uint64_t n, uint64_t at)
{
    std::vector<PacketPtr> in, out, rev;
    MRN::TopologyLocalInfo here, *info = &here;
    PacketPtr params;
    int32_t want = 0;
    for (uint64_t i = 0; i < n; i++) {
        const int32_t v = (int32_t)(i * 7u + 1u);
        const double d = 0.5;
        const void *f[1] = { i == at ? (const void *)&d : (const void *)&v };
        in.push_back(received(1073741834u, 100, i == at ? "%lf" : "%d", f, 0));
        if (i != at) want = (int32_t)((uint32_t)want + (uint32_t)v);
    }
    void *state = NULL;
    mrpdr_Sum(in, out, rev, &state, params, *info);
    bool ok = out.empty();
    const int32_t one = 1;
    const void *f[1] = { &one };
    in[at] = received(1073741834u, 100, "%d", f, 0);
    mrpdr_Sum(in, out, rev, &state, params, *info);
    want = (int32_t)((uint32_t)want + 1u);
    ok = ok && out.size() == 1 && memcmp(out[0]->get_Buffer(), &want, 4) == 0;
    release(static_cast<GpuFilterState *>(state));
    end_SubTest("mrpdr_Sum(%d, " + std::to_string(n) + " packets, packet " + std::to_string(at) + " of another format)", ok);
}

// ---- mrpdr_ArrayConcat --------------------------------------------------------------------------
// An array packet as it looks after Message::recv from a sender of byte order `bo`.
static PacketPtr received_array(const char *fmt, const void *arr, uint64_t len, size_t w, int bo)
{
    PacketPtr p(fmt[1] == 'A' ? new Packet(5u, 9, fmt, arr, (uint64_t)len) : new Packet(5u, 9, fmt, arr, (unsigned)len));
    std::vector<char> &b = p->impl_->buf;
    if (bo) {
        for (size_t k = 0; k < 4; k++) std::swap(b[k], b[7 - k]);
        if (p->impl_->array_of_strings) {
            size_t at = 8;
            for (uint64_t j = 0; j < len; j++) {
                uint32_t l; memcpy(&l, &b[at], 4);
                std::swap(b[at], b[at + 3]); std::swap(b[at + 1], b[at + 2]);
                at += 4 + l;
            }
        } else {
            for (uint64_t j = 0; j < len; j++)
                for (size_t k = 0; k < w / 2; k++) std::swap(b[8 + j * w + k], b[8 + j * w + w - 1 - k]);
        }
    }
    p->impl_->hdr.assign(16 + strlen(fmt) + 1 + 8 + 1, 0);
    p->impl_->hdr.back() = (char)bo;
    return p;
}

template <typename T> static void test_concat(const char *fmt, uint64_t n, int order_mode)
{
    std::vector<PacketPtr> in, out, rev;
    std::vector<T> all;
    uint64_t x = 1234567 + n * 7 + (uint64_t)order_mode;
    for (uint64_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t len = (x >> 20) % 5 == 0 ? 0 : (x >> 30) % 41;
        std::vector<T> mine(len);
        for (uint64_t j = 0; j < len; j++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; mine[j] = (T)(int64_t)(x >> 11); }
        all.insert(all.end(), mine.begin(), mine.end());
        const int bo = order_mode == 2 ? (int)(i % 3 == 1) : order_mode;
        in.push_back(received_array(fmt, mine.data(), len, sizeof(T), bo));
    }
    void *state = NULL;
    PacketPtr params;
    MRN::TopologyLocalInfo here, *info = &here;
    mrpdr_ArrayConcat(in, out, rev, &state, params, *info);
    const uint64_t total = all.size();
    bool ok = out.size() == 1 && out[0]->get_StreamId() == 5u && out[0]->get_Tag() == 9 && !strcmp(out[0]->get_FormatString(), fmt) &&
              out[0]->get_BufferLen() == 8 + total * sizeof(T) && !memcmp(out[0]->get_Buffer(), &total, 8) &&
              (total == 0 || !memcmp(out[0]->get_Buffer() + 8, all.data(), total * sizeof(T)));
    // the stream's next wave through the same state
    std::vector<PacketPtr> out2;
    mrpdr_ArrayConcat(in, out2, rev, &state, params, *info);
    ok = ok && out2.size() == 1 && out2[0]->get_BufferLen() == out[0]->get_BufferLen() &&
         !memcmp(out2[0]->get_Buffer(), out[0]->get_Buffer(), (size_t)out[0]->get_BufferLen());
    release_var(static_cast<VarState *>(state));
    end_SubTest(std::string("mrpdr_ArrayConcat(") + fmt + ", " + std::to_string(n) + " packets, orders " +
                (order_mode == 2 ? "mixed" : order_mode ? "big" : "little") + ")", ok);
}

static void test_concat_strings(uint64_t n, int order_mode)
{
    std::vector<PacketPtr> in, out, rev;
    std::vector<std::string> all;
    uint64_t x = 99 + n;
    for (uint64_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t len = (x >> 20) % 4;
        std::vector<std::string> mine(len);
        std::vector<char *> ptrs(len);
        for (uint64_t j = 0; j < len; j++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            mine[j] = std::string((size_t)((x >> 9) % 9), (char)('a' + (x >> 40) % 26)) + std::to_string(i);
            ptrs[j] = (char *)mine[j].c_str();
        }
        all.insert(all.end(), mine.begin(), mine.end());
        const int bo = order_mode == 2 ? (int)(i % 3 == 1) : order_mode;
        in.push_back(received_array("%as", ptrs.data(), len, 1, bo));
    }
    void *state = NULL;
    PacketPtr params;
    MRN::TopologyLocalInfo here, *info = &here;
    mrpdr_ArrayConcat(in, out, rev, &state, params, *info);
    std::vector<char> want;
    const uint64_t total = all.size();
    want.insert(want.end(), (const char *)&total, (const char *)&total + 8);
    for (const std::string &str : all) {
        const uint32_t l = (uint32_t)str.size() + 1;
        want.insert(want.end(), (const char *)&l, (const char *)&l + 4);
        want.insert(want.end(), str.c_str(), str.c_str() + l);
    }
    const bool ok = out.size() == 1 && out[0]->get_BufferLen() == want.size() && !memcmp(out[0]->get_Buffer(), want.data(), want.size());
    release_var(static_cast<VarState *>(state));
    end_SubTest(std::string("mrpdr_ArrayConcat(%as, ") + std::to_string(n) + " packets, orders " +
                (order_mode == 2 ? "mixed" : order_mode ? "big" : "little") + ")", ok);
}

// ---- mrpdr_IntEqClass ---------------------------------------------------------------------------
#include <map>

static void test_eq_class(uint64_t n, int order_mode)
{
    std::vector<PacketPtr> in, out, rev;
    std::map<unsigned, std::vector<unsigned> > classes;          // the reference's own structure (:820, :844)
    uint64_t x = 4242 + n * 3 + (uint64_t)order_mode;
    for (uint64_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const unsigned nv = (unsigned)((x >> 20) % 4);
        std::vector<unsigned> vals(nv), cnts(nv), mems;
        for (unsigned j = 0; j < nv; j++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            vals[j] = (unsigned)((x >> 9) % 37);
            cnts[j] = (unsigned)((x >> 30) % 4);                  // empty classes among them
            for (unsigned k = 0; k < cnts[j]; k++) { mems.push_back((unsigned)(i * 16 + j * 4 + k)); classes[vals[j]].push_back(mems.back()); }
        }
        const int bo = order_mode == 2 ? (int)(i % 3 == 1) : order_mode;
        PacketPtr p(new Packet(11u, 3, "%aud %aud %aud", vals.data(), nv, cnts.data(), nv, mems.data(), (unsigned)mems.size()));
        if (bo) {          // as a big-endian sender wrote it: counts and elements reversed
            std::vector<char> &b = p->impl_->buf;
            size_t at = 0;
            for (int f = 0; f < 3; f++) {
                uint64_t len; memcpy(&len, &b[at], 8);
                for (size_t k = 0; k < 4; k++) std::swap(b[at + k], b[at + 7 - k]);
                at += 8;
                for (uint64_t j = 0; j < len; j++, at += 4) { std::swap(b[at], b[at + 3]); std::swap(b[at + 1], b[at + 2]); }
            }
        }
        p->impl_->hdr.assign(16 + 15 + 8 + 1, 0);
        p->impl_->hdr.back() = (char)bo;
        in.push_back(p);
    }
    void *state = NULL;
    PacketPtr params;
    MRN::TopologyLocalInfo here, *info = &here;
    mrpdr_IntEqClass(in, out, rev, &state, params, *info);
    std::vector<unsigned> wv, wc, wm;
    for (std::map<unsigned, std::vector<unsigned> >::iterator it = classes.begin(); it != classes.end(); ++it) {
        wv.push_back(it->first); wc.push_back((unsigned)it->second.size());
        wm.insert(wm.end(), it->second.begin(), it->second.end());
    }
    std::vector<char> want;
    const std::vector<unsigned> *parts[3] = { &wv, &wc, &wm };
    for (int f = 0; f < 3; f++) {
        const uint64_t len = parts[f]->size();
        want.insert(want.end(), (const char *)&len, (const char *)&len + 8);
        want.insert(want.end(), (const char *)parts[f]->data(), (const char *)parts[f]->data() + 4 * len);
    }
    const bool ok = out.size() == 1 && out[0]->get_StreamId() == 11u && out[0]->get_Tag() == 3 &&
                    out[0]->get_BufferLen() == want.size() && !memcmp(out[0]->get_Buffer(), want.data(), want.size());
    release_var(static_cast<VarState *>(state));
    end_SubTest(std::string("mrpdr_IntEqClass(") + std::to_string(n) + " packets, orders " +
                (order_mode == 2 ? "mixed" : order_mode ? "big" : "little") + ")", ok);
}

// ---- --time: what a wave costs through the plugin, host memory in, host memory out ------------------
#include <chrono>

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Wall time per wave of mrpdr_Sum ('%d'), mrpdr_Avg ('%lf %d') and mrpdr_ArrayConcat ('%ad' of eight
// elements) for growing waves, the filter state kept between waves as MRNet keeps it; beside each the
// time of the gather loop alone (the plugin's host part: format check and row copy per packet) and of
// the fold written out on the host over the same packet buffers with nothing else around it (no
// Packet decode, no DataElement, no lock: a lower bound for any CPU filter, far below the reference's
// own per-packet path).  One JSON line on stdout.
static int time_mode()
{
    MRN::TopologyLocalInfo here, *info = &here;
    PacketPtr params;
    printf("{\"plugin_us_per_wave\": {");
    bool first = true;
    for (uint64_t n : { (uint64_t)4, (uint64_t)64, (uint64_t)1024, (uint64_t)16384, (uint64_t)65536 }) {
        std::vector<PacketPtr> in_d, in_lf, in_a, out, rev;
        std::vector<int32_t> vd(n), cnt(n, 1);
        std::vector<double> vlf(n);
        for (uint64_t i = 0; i < n; i++) {
            vd[i] = (int32_t)(i * 2654435761u); vlf[i] = 1e-3 + 1e-7 * (double)(i % 1000);
            const void *f1[1] = { &vd[i] };
            const void *f2[2] = { &vlf[i], &cnt[i] };
            int32_t arr[8];
            for (int k = 0; k < 8; k++) arr[k] = (int32_t)(i + k);
            in_d.push_back(received(1073741834u, 100, "%d", f1, 0));
            in_lf.push_back(received(1073741834u, 100, "%lf %d", f2, 0));
            in_a.push_back(received_array("%ad", arr, 8, 4, 0));
        }
        const int reps = n <= 1024 ? 300 : (n <= 16384 ? 100 : 40);
        double t_sum, t_avg, t_cat, t_gather, t_fold, t_fold_avg;
        void *s1 = NULL, *s2 = NULL, *s3 = NULL;
        for (int w = 0; w < 5; w++) { out.clear(); mrpdr_Sum(in_d, out, rev, &s1, params, *info); }
        double t0 = now_us();
        for (int r = 0; r < reps; r++) { out.clear(); mrpdr_Sum(in_d, out, rev, &s1, params, *info); }
        t_sum = (now_us() - t0) / reps;
        int32_t want = 0;
        for (uint64_t i = 0; i < n; i++) want = (int32_t)((uint32_t)want + (uint32_t)vd[i]);
        bool ok = out.size() == 1 && !memcmp(out[0]->get_Buffer(), &want, 4);
        for (int w = 0; w < 5; w++) { out.clear(); mrpdr_Avg(in_lf, out, rev, &s2, params, *info); }
        t0 = now_us();
        for (int r = 0; r < reps; r++) { out.clear(); mrpdr_Avg(in_lf, out, rev, &s2, params, *info); }
        t_avg = (now_us() - t0) / reps;
        double acc = 0; int32_t total = 0;
        for (uint64_t i = 0; i < n; i++) { acc = acc + vlf[i] * (double)cnt[i]; total += cnt[i]; }
        acc = acc / (double)total;
        ok = ok && out.size() == 1 && !memcmp(out[0]->get_Buffer(), &acc, 8);
        for (int w = 0; w < 3; w++) { out.clear(); mrpdr_ArrayConcat(in_a, out, rev, &s3, params, *info); }
        t0 = now_us();
        for (int r = 0; r < reps / 4 + 1; r++) { out.clear(); mrpdr_ArrayConcat(in_a, out, rev, &s3, params, *info); }
        t_cat = (now_us() - t0) / (reps / 4 + 1);
        ok = ok && out.size() == 1 && out[0]->get_BufferLen() == 8 + 32 * n;
        // the plugin's own host part: what it does per packet before the copy to the device
        std::vector<uint8_t> rows(4 * n), orders(n);
        t0 = now_us();
        for (int r = 0; r < reps; r++) {
            const char *fmt_text = in_d[0]->get_FormatString(), *same = fmt_text;
            for (uint64_t i = 0; i < n; i++) {
                const Packet &p = *in_d[i];
                const char *pf = p.get_FormatString();
                if ((pf != same && strcmp(pf, fmt_text) != 0) || p.get_BufferLen() != 4) { ok = false; break; }
                same = pf;
                memcpy(&rows[4 * i], p.get_Buffer(), 4);
                orders[i] = (uint8_t)p.get_Header()[p.get_HeaderLen() - 1];
            }
            __asm__ volatile("" : : "r"(rows.data()), "r"(orders.data()) : "memory");
        }
        t_gather = (now_us() - t0) / reps;
        // the fold on the host over the packets' own buffers
        t0 = now_us();
        for (int r = 0; r < reps; r++) {
            int32_t sres = 0;
            for (uint64_t i = 0; i < n; i++) { int32_t x; memcpy(&x, in_d[i]->get_Buffer(), 4); sres = (int32_t)((uint32_t)sres + (uint32_t)x); }
            __asm__ volatile("" : : "r"(sres) : "memory");
            ok = ok && sres == want;
        }
        t_fold = (now_us() - t0) / reps;
        t0 = now_us();
        for (int r = 0; r < reps; r++) {
            double a = 0; int32_t tc = 0;
            for (uint64_t i = 0; i < n; i++) {
                double x; int32_t c; memcpy(&x, in_lf[i]->get_Buffer(), 8); memcpy(&c, in_lf[i]->get_Buffer() + 8, 4);
                a = a + x * (double)c; tc += c;
            }
            a = a / (double)tc;
            __asm__ volatile("" : : "x"(a) : "memory");
            ok = ok && a == acc;
        }
        t_fold_avg = (now_us() - t0) / reps;
        release(static_cast<GpuFilterState *>(s1));
        release(static_cast<GpuFilterState *>(s2));
        release_var(static_cast<VarState *>(s3));
        printf("%s\"%llu\": {\"mrpdr_Sum_d\": %.1f, \"mrpdr_Avg_lf\": %.1f, \"mrpdr_ArrayConcat_ad8\": %.1f, "
               "\"gather_loop_only\": %.2f, \"host_fold_sum_d\": %.2f, \"host_fold_avg_lf\": %.2f, \"ok\": %s}",
               first ? "" : ", ", (unsigned long long)n, t_sum, t_avg, t_cat, t_gather, t_fold, t_fold_avg, ok ? "true" : "false");
        first = false;
        if (!ok) failures++;
    }
    printf("}}\n");
    return failures ? 1 : 0;
}

int main(int argc, char **argv)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { printf("MRNTEST_FAILURE: no HIP device\n"); return 1; }
    if (argc > 1 && !strcmp(argv[1], "--time")) return time_mode();
    for (uint64_t n : { (uint64_t)1, (uint64_t)4, (uint64_t)1025, (uint64_t)70000 })
        for (int mode = 0; mode < 3; mode++) {
            test_type<int32_t>("%d", n, mode);
            test_type<int16_t>("%hd", n, mode);
            test_type<uint64_t>("%uld", n, mode);
            test_type<float>("%f", n, mode);
            test_type<double>("%lf", n, mode);
        }
    test_odd_packet(100, 37);
    test_odd_packet(70000, 0);
    test_odd_packet(70000, 33333);
    test_odd_packet(70000, 69999);
    for (uint64_t n : { (uint64_t)1, (uint64_t)5, (uint64_t)1025, (uint64_t)20000 })
        for (int mode = 0; mode < 3; mode++) {
            test_concat<int32_t>("%ad", n, mode);
            test_concat<uint8_t>("%auc", n, mode);
            test_concat<int16_t>("%ahd", n, mode);
            test_concat<double>("%alf", n, mode);
            test_concat<int64_t>("%Ald", n, mode);
            test_concat_strings(n, mode);
        }
    for (uint64_t n : { (uint64_t)1, (uint64_t)7, (uint64_t)1025, (uint64_t)20000 })
        for (int mode = 0; mode < 3; mode++) test_eq_class(n, mode);
    {   // a scalar format is not an array: nothing comes out (src/FilterDefinitions.C:764-768)
        std::vector<PacketPtr> in, out, rev;
        int32_t a = 1;
        const void *f[1] = { &a };
        in.push_back(received(7, 100, "%d", f, 0));
        void *state = NULL;
        PacketPtr params;
        MRN::TopologyLocalInfo here, *info = &here;
        mrpdr_ArrayConcat(in, out, rev, &state, params, *info);
        release_var(static_cast<VarState *>(state));
        end_SubTest("mrpdr_ArrayConcat refuses a scalar format", out.empty());
    }
    // a format the built-ins refuse, and a wave whose packets do not share one format: no output
    {
        std::vector<PacketPtr> in, out, rev;
        int32_t a = 1, b = 2;
        const void *f[2] = { &a, &b };
        in.push_back(received(7, 100, "%d %d", f, 0));
        void *state = NULL;
        PacketPtr params;
        MRN::TopologyLocalInfo here, *info = &here;
        mrpdr_Sum(in, out, rev, &state, params, *info);
        bool ok = out.empty();
        in.clear();
        in.push_back(received(7, 100, "%d", f, 0));
        in.push_back(received(7, 100, "%ud", f, 0));
        mrpdr_Sum(in, out, rev, &state, params, *info);
        ok = ok && out.empty();
        release(static_cast<GpuFilterState *>(state));
        end_SubTest("formats the filter does not accept emit nothing", ok);
    }
    printf("%s\n", failures ? "some sub-tests FAILED" : "all plugin sub-tests passed");
    return failures ? 1 : 0;
}
