// Runs integration/mrpdr_filters.C -- the MRNet filter plugin -- without MRNet: the few Packet
// methods the plugin calls are implemented here over plain host buffers laid out as
// Packet::pdr_packet_header / pdr_packet_data lay them out (src/Packet.C:557-845), waves are handed
// to mrpdr_Sum / Min / Max / Avg the way Filter::push_Packets does (src/Filter.C:42-117), and the
// output packet is compared with the reference fold written out on the host
// (src/FilterDefinitions.C:100-149, 517-561).  Prints MRNTEST_SUCCESS / MRNTEST_FAILURE lines like
// tests/test_common.C:41-187.
#include <math.h>
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
Packet::~Packet() { delete impl_; }
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

int main()
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { printf("MRNTEST_FAILURE: no HIP device\n"); return 1; }
    for (uint64_t n : { (uint64_t)1, (uint64_t)4, (uint64_t)1025, (uint64_t)70000 })
        for (int mode = 0; mode < 3; mode++) {
            test_type<int32_t>("%d", n, mode);
            test_type<int16_t>("%hd", n, mode);
            test_type<uint64_t>("%uld", n, mode);
            test_type<float>("%f", n, mode);
            test_type<double>("%lf", n, mode);
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
