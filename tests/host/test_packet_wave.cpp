// C++ host-side test of mrnet_amd/host/PacketWave.h, in the style of the reference's
// test harness (tests/test_common.C:41-187): every sub-test prints MRNTEST_SUCCESS or
// MRNTEST_FAILURE, and the driver greps the output for "failure" (tests/mrnet_tests.sh:119-130).
//
// Sub-tests follow tests/test_basic_FE.C (alltypes packet), tests/test_arrays_FE.C
// (ARRAY_LEN elements, memcmp) and tests/test_NativeFilters_FE.C (TFILTER_SUM over a
// constant per back-end), plus the byte vectors captured from the compiled reference.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../mrnet_amd/host/PacketWave.h"

using namespace MRN_GPU;

static int failures = 0;
static void end_SubTest(const std::string &name, bool ok)
{
    printf("%s: %s\n", name.c_str(), ok ? "MRNTEST_SUCCESS" : "MRNTEST_FAILURE");
    if (!ok) failures++;
}

static std::vector<uint8_t> from_hex(const char *hex)
{
    std::vector<uint8_t> out;
    for (const char *p = hex; *p;) {
        if (*p == ' ') { p++; continue; }
        unsigned v;
        sscanf(p, "%2x", &v);
        out.push_back((uint8_t)v);
        p += 2;
    }
    return out;
}

template <typename T> static HostColumn scalars(uint64_t n, T v)
{
    HostColumn c;
    c.set_scalars(std::vector<T>(n, v));
    return c;
}

static HostColumn strings(uint64_t n, const char *s)
{
    HostColumn c;
    const size_t l = strlen(s) + 1;
    for (uint64_t i = 0; i < n; i++) {
        c.counts.push_back(l);
        c.values.insert(c.values.end(), (const uint8_t *)s, (const uint8_t *)s + l);
    }
    return c;
}

// What the front-end holds after Message::recv: headers and data buffers of the wave.
static PacketWave *received(WaveContext &wc, PacketWave &sent)
{
    std::vector<uint8_t> hdrs, arena;
    std::vector<uint64_t> offs, lens;
    std::vector<uint32_t> src(sent.size());
    for (uint64_t i = 0; i < sent.size(); i++) src[i] = (uint32_t)i;
    sent.get_Headers(hdrs, &src);
    sent.get_Buffers(arena, offs, lens);
    return new PacketWave(wc, sent.get_FormatString(), sent.size(), (const char *)hdrs.data(),
                          sent.get_HeaderLen(), (const char *)arena.data(), offs.data(), lens.data(), arena.size());
}

static void test_golden(WaveContext &wc)
{
    // new_Packet_t_2(0x40000010, 101, "%d %uld %lf %s %auc %af", -17, 17, 123.5, "hi", {aa,bb}, {1,2.5,-3})
    const char *fmt = "%d %uld %lf %s %auc %af";
    std::vector<HostColumn> cols(6);
    cols[0] = scalars<int32_t>(1, -17);
    cols[1] = scalars<uint64_t>(1, 17);
    cols[2] = scalars<double>(1, 123.5);
    cols[3] = strings(1, "hi");
    cols[4].set_arrays(std::vector<uint64_t>{2}, std::vector<uint8_t>{0xAA, 0xBB});
    cols[5].set_arrays(std::vector<uint64_t>{3}, std::vector<float>{1.0f, 2.5f, -3.0f});
    PacketWave pkt(wc, 0x40000010u, 101, fmt, cols, 1);
    std::vector<uint8_t> hdrs, arena;
    std::vector<uint64_t> offs, lens;
    pkt.get_Headers(hdrs);
    pkt.get_Buffers(arena, offs, lens);
    const std::vector<uint8_t> want_hdr = from_hex(
        "10 00 00 40 65 00 00 00 ff ff ff ff 18 00 00 00 25 64 20 25 75 6c 64 20 25 6c 66 20 25 73 20 25 61 75 63 "
        "20 25 61 66 00 00 00 00 00 00 00 00 00 00");
    const std::vector<uint8_t> want_buf = from_hex(
        "ef ff ff ff 11 00 00 00 00 00 00 00 00 00 00 00 00 e0 5e 40 03 00 00 00 68 69 00 02 00 00 00 00 00 00 00 "
        "aa bb 03 00 00 00 00 00 00 00 00 00 80 3f 00 00 20 40 00 00 40 c0");
    end_SubTest("golden(header bytes)", hdrs == want_hdr);
    end_SubTest("golden(data bytes)", lens[0] == 57 && std::vector<uint8_t>(arena.begin() + offs[0], arena.begin() + offs[0] + 57) == want_buf);
}

static void test_alltypes(WaveContext &wc, uint64_t nb)
{
    const char *fmt = "%c %uc %hd %uhd %d %ud %ld %uld %f %lf %s";
    std::vector<HostColumn> cols(11);
    cols[0] = scalars<char>(nb, 'A');        cols[1] = scalars<unsigned char>(nb, 'B');
    cols[2] = scalars<int16_t>(nb, -17);     cols[3] = scalars<uint16_t>(nb, 17);
    cols[4] = scalars<int32_t>(nb, -17);     cols[5] = scalars<uint32_t>(nb, 17);
    cols[6] = scalars<int64_t>(nb, -17);     cols[7] = scalars<uint64_t>(nb, 17);
    cols[8] = scalars<float>(nb, 123.23412f); cols[9] = scalars<double>(nb, 123.23412);
    cols[10] = strings(nb, "Test String");
    PacketWave sent(wc, (1u << 30) + 10, 101, fmt, cols, nb);
    PacketWave *got = received(wc, sent);
    std::vector<HostColumn> out;
    bool ok = !got->has_Error() && got->get_StreamId() == (1u << 30) + 10 && got->get_Tag() == 101;
    ok = ok && got->unpack("%d", out) == -1;                 // format mismatch
    ok = ok && got->unpack(fmt, out) == 0;
    for (int f = 0; ok && f < 11; f++) ok = out[f].values == cols[f].values && out[f].counts == cols[f].counts;
    for (uint64_t i = 0; ok && i < nb; i++) ok = got->get_SourceRanks()[i] == i;
    end_SubTest("test_alltypes(" + std::to_string(nb) + " back-ends)", ok);
    delete got;
}

template <typename T> static void test_array(WaveContext &wc, const char *fmt, T val)
{
    const uint64_t ARRAY_LEN = 5000, nb = 4;
    std::vector<HostColumn> cols(1);
    cols[0].set_arrays(std::vector<uint64_t>(nb, ARRAY_LEN), std::vector<T>(nb * ARRAY_LEN, val));
    PacketWave sent(wc, (1u << 30) + 10, 102, fmt, cols, nb);
    PacketWave *got = received(wc, sent);
    std::vector<HostColumn> out;
    bool ok = got->unpack(fmt, out) == 0 && out[0].counts == cols[0].counts &&
              out[0].values.size() == cols[0].values.size() &&
              memcmp(out[0].values.data(), cols[0].values.data(), cols[0].values.size()) == 0;
    end_SubTest(std::string("test_array(") + fmt + ")", ok);
    delete got;
}

template <typename T> static void test_sum(WaveContext &wc, const char *fmt, T val, uint64_t nb)
{
    std::vector<HostColumn> cols(1);
    cols[0] = scalars<T>(nb, val);
    PacketWave sent(wc, (1u << 30) + 10, 103, fmt, cols, nb);
    PacketWave *got = received(wc, sent);
    uint8_t res[16];
    T comp = 0;
    for (uint64_t i = 0; i < nb; i++) comp = (T)(comp + val);          // tests/test_NativeFilters_FE.C:122-262
    bool ok = tfilter_Sum(*got, res, MRPDR_SHAPE_FLAT) == 0 && memcmp(res, &comp, sizeof(T)) == 0;
    end_SubTest(std::string("test_Sum(") + fmt + ", " + std::to_string(nb) + ")", ok);
    delete got;
}

static void test_filters_known_answers(WaveContext &wc)
{
    std::vector<HostColumn> cols(1);
    cols[0].set_scalars(std::vector<float>{123.45f, 1e8f, -1e8f, 0.1f, 3.0f});
    PacketWave wave(wc, 7, 100, "%f", cols, 5);
    uint8_t r[16];
    float f;
    char txt[64];
    bool ok = tfilter_Sum(wave, r, MRPDR_SHAPE_FLAT) == 0;
    memcpy(&f, r, 4); snprintf(txt, sizeof txt, "%f", f);
    ok = ok && !strcmp(txt, "123.099998");
    ok = ok && tfilter_Max(wave, r, MRPDR_SHAPE_FLAT) == 0 && (memcpy(&f, r, 4), f == 1e8f);
    ok = ok && tfilter_Min(wave, r, MRPDR_SHAPE_FLAT) == 0 && (memcpy(&f, r, 4), f == -1e8f);
    end_SubTest("tfilter_Sum/Max/Min(captured reference results)", ok);
    std::vector<HostColumn> a(2);
    a[0].set_scalars(std::vector<int32_t>{7, 8, 10});
    a[1].set_scalars(std::vector<int32_t>{1, 2, 4});
    PacketWave avg(wc, 7, 100, "%d %d", a, 3);
    int32_t v[2];
    ok = tfilter_Avg(avg, r, MRPDR_SHAPE_FLAT) == 0 && (memcpy(v, r, 8), v[0] == 9 && v[1] == 7);
    ok = ok && tfilter_Sum(avg, r) == -1;                  // "%d %d" is not a format tfilter_Sum accepts
    end_SubTest("tfilter_Avg(captured reference result)", ok);
}

// The mirrors called without a shape are ONE reference filter call: a single left fold, and for
// Avg a single division (src/FilterDefinitions.C:517-561), whatever the number of packets.
static void test_filters_default_shape(WaveContext &wc)
{
    const uint64_t nb = 1025;
    std::vector<int32_t> v(nb), c(nb);
    std::vector<float> f(nb);
    int32_t sum = 0, cnt = 0;
    float fsum = 0.f;
    for (uint64_t i = 0; i < nb; i++) {
        v[i] = (int32_t)(i * 2654435761u % 1000) - 300;
        c[i] = (int32_t)(i % 5) + 1;
        sum += v[i] * c[i];
        cnt += c[i];
        f[i] = (float)((double)(i * 40503u % 2000001) / 1e3 - 1000.0) * (i % 3 ? 1.f : 1e6f);
        fsum = fsum + f[i];
    }
    std::vector<HostColumn> a(2);
    a[0].set_scalars(v);
    a[1].set_scalars(c);
    PacketWave avg(wc, 7, 100, "%d %d", a, nb);
    uint8_t r[16];
    int32_t got[2];
    bool ok = tfilter_Avg(avg, r) == 0 && (memcpy(got, r, 8), got[0] == sum / cnt && got[1] == cnt);
    std::vector<HostColumn> b(1);
    b[0].set_scalars(f);
    PacketWave fw(wc, 7, 100, "%f", b, nb);
    float fg;
    ok = ok && tfilter_Sum(fw, r) == 0 && (memcpy(&fg, r, 4), memcmp(&fg, &fsum, 4) == 0);
    end_SubTest("tfilter_Avg/Sum(default shape = one reference call, 1025 packets)", ok);
}

// A wave consumer that reads the decoded fields in place (PacketWave::view) instead of taking
// unpack's copies: here the consumer is the Sum kernel, run straight on the viewed column.
static void test_column_views(WaveContext &wc)
{
    const uint64_t nb = 3000;
    std::vector<HostColumn> cols(2);
    std::vector<int32_t> v(nb);
    std::vector<uint64_t> cnt(nb);
    std::vector<float> arr;
    int32_t want = 0;
    for (uint64_t i = 0; i < nb; i++) {
        v[i] = (int32_t)(i * 2654435761u);
        want += v[i];
        cnt[i] = i % 4;
        for (uint64_t j = 0; j < cnt[i]; j++) arr.push_back((float)(i + j));
    }
    cols[0].set_scalars(v);
    cols[1].set_arrays(cnt, arr);
    PacketWave sent(wc, 9, 104, "%d %af", cols, nb);
    PacketWave *got = received(wc, sent);
    std::vector<mrpdr_field_view> views;
    bool ok = got->view("%d %d", views) == -1 && got->view("%d %af", views) == 0 && views.size() == 2;
    ok = ok && views[0].values == got->columns()[0].values && views[0].n == nb && views[0].type == MRPDR_INT32_T &&
         views[0].counts == nullptr && views[1].kind == MRPDR_ARRAY && views[1].width == 4 &&
         views[1].values == got->columns()[1].values && views[1].offs == got->columns()[1].offs;
    DeviceBuffer d_res, d_ok;
    d_res.resize(16); d_ok.resize(4);
    ok = ok && mrpdr_wave_filter(wc.ctx, MRPDR_SUM, views[0].type, MRPDR_SHAPE_FLAT, views[0].values, (uint64_t)views[0].width,
                                 nullptr, 0, views[0].n, 0, d_res.ptr<void>(), d_ok.ptr<int32_t>(), wc.stream);
    // the array field's elements, all packets back to back, summed in place as well
    float fwant = 0.f;
    for (float x : arr) fwant = fwant + x;
    DeviceBuffer d_res2;
    d_res2.resize(16);
    ok = ok && mrpdr_wave_filter(wc.ctx, MRPDR_SUM, MRPDR_FLOAT_T, MRPDR_SHAPE_FLAT, views[1].values, 4, nullptr, 0,
                                 got->total(1), 0, d_res2.ptr<void>(), d_ok.ptr<int32_t>(), wc.stream);
    int32_t sum = 0;
    float fsum = 0.f;
    d_res.download(&sum, 4, wc.stream);
    d_res2.download(&fsum, 4, wc.stream);
    wc.sync();
    ok = ok && sum == want && memcmp(&fsum, &fwant, 4) == 0;
    end_SubTest("PacketWave::view(fields read in place, no host copy)", ok);
    delete got;                       // consumers are done: now the wave, and with it the columns
}

static void test_truncated(WaveContext &wc)
{
    const char *fmt = "%d %ad";
    const uint64_t nb = 40;
    std::vector<HostColumn> cols(2);
    cols[0] = scalars<int32_t>(nb, 5);
    cols[1].set_arrays(std::vector<uint64_t>(nb, 3), std::vector<int32_t>(nb * 3, 9));
    PacketWave sent(wc, 9, 101, fmt, cols, nb);
    std::vector<uint8_t> hdrs, arena;
    std::vector<uint64_t> offs, lens;
    sent.get_Headers(hdrs);
    sent.get_Buffers(arena, offs, lens);
    lens[7] = 9;                                            // cut inside the array count
    PacketWave got(wc, fmt, nb, (const char *)hdrs.data(), sent.get_HeaderLen(), (const char *)arena.data(),
                   offs.data(), lens.data(), arena.size());
    std::vector<HostColumn> out;
    bool ok = got.unpack(fmt, out) == 0 && got.has_Error();
    uint64_t good = 0;
    for (uint64_t i = 0; i < nb; i++) good += got.ok()[i] != 0;
    ok = ok && good == nb - 1 && got.ok()[7] == 0 && out[1].counts[7] == 0 && out[1].counts[8] == 3;
    end_SubTest("truncated buffer fails per packet", ok);
}

// The shape of MRNet's own PROT_NEW_FILTER message, "%s %as %auhd" (src/Network.C:1954-1955).
static void test_string_arrays(WaveContext &wc)
{
    const char *fmt = "%s %as %auhd";
    const uint64_t nb = 5;
    std::vector<HostColumn> cols(3);
    cols[0] = strings(nb, "libfilters.so");
    std::vector<std::vector<std::string> > names(nb);
    std::vector<uint64_t> nids;
    std::vector<uint16_t> ids;
    for (uint64_t i = 0; i < nb; i++) {
        for (uint64_t k = 0; k < i; k++) names[i].push_back(std::string(k, 'f') + "_filter" + std::to_string(i));
        nids.push_back(i);
        for (uint64_t k = 0; k < i; k++) ids.push_back((uint16_t)(100 * i + k));
    }
    cols[1].set_string_arrays(names);
    cols[2].set_arrays(nids, ids);
    PacketWave sent(wc, 1, 9, fmt, cols, nb);
    PacketWave *got = received(wc, sent);
    std::vector<HostColumn> out;
    bool ok = !got->has_Error() && got->unpack(fmt, out) == 0;
    for (int f = 0; ok && f < 3; f++)
        ok = out[f].values == cols[f].values && out[f].counts == cols[f].counts && out[f].sub_counts == cols[f].sub_counts;
    end_SubTest("test_string_arrays(%s %as %auhd)", ok);
    delete got;
}

// Packets written one at a time, as user code does (tests/test_basic_FE.C sends
// "%c %uc %hd %uhd %d %ud %ld %uld %f %lf %s" with these very values), then read back one at a time.
static void test_builder_reader(WaveContext &wc)
{
    const char *fmt = "%c %uc %hd %uhd %d %ud %ld %uld %f %lf %s %ad %As";
    WaveBuilder b(fmt);
    const uint64_t nb = 7;
    bool ok = true;
    for (uint64_t i = 0; i < nb; i++) {
        std::vector<int32_t> arr(i, (int32_t)(100 + i));
        const char *strs[2] = { "first", "second string" };
        ok = ok && b.add('A', (unsigned char)'B', (int16_t)-17, (uint16_t)17, -17 - (int)i, 17u, (int64_t)-17, (uint64_t)17,
                         123.23412f, 123.23412, "Test String", arr.data(), (uint32_t)i, strs, (uint64_t)(i % 3));
    }
    ok = ok && !b.add(1, 2, 3);                              // wrong argument list: nothing added
    ok = ok && b.size() == nb;
    PacketWave sent(wc, (1u << 30) + 10, 101, fmt, b.columns(), nb);
    PacketWave *got = received(wc, sent);
    std::vector<HostColumn> out;
    ok = ok && got->unpack(fmt, out) == 0;
    WaveReader rd(fmt, out, nb);
    for (uint64_t i = 0; ok && i < nb; i++) {
        char c; unsigned char uc; int16_t hd; uint16_t uhd; int32_t d; uint32_t ud; int64_t ld; uint64_t uld;
        float f; double lf; char *s = nullptr; int32_t *arr = nullptr; uint32_t alen = 99; char **strs = nullptr; uint64_t slen = 99;
        ok = rd.unpack(i, fmt, &c, &uc, &hd, &uhd, &d, &ud, &ld, &uld, &f, &lf, &s, &arr, &alen, &strs, &slen) == 0;
        ok = ok && c == 'A' && uc == 'B' && hd == -17 && uhd == 17 && d == -17 - (int)i && ud == 17 && ld == -17 &&
             uld == 17 && f == 123.23412f && lf == 123.23412 && !strcmp(s, "Test String") && alen == i && slen == i % 3;
        ok = ok && ((i == 0) == (arr == nullptr));           // zero-length array -> NULL (src/Packet.C:1163-1164)
        for (uint32_t k = 0; ok && k < alen; k++) ok = arr[k] == (int32_t)(100 + i);
        if (ok && slen > 0) ok = !strcmp(strs[0], "first");
        if (ok && slen > 1) ok = !strcmp(strs[1], "second string");
        for (uint64_t k = 0; strs && k < slen; k++) free(strs[k]);
        free(strs); free(arr); free(s);
    }
    ok = ok && rd.unpack(0, "%d", (int *)nullptr) == -1;     // format mismatch
    end_SubTest("WaveBuilder / WaveReader (one packet at a time)", ok);
    delete got;
}

// The array filters (src/FilterDefinitions.C:649-1239) on small waves whose results follow from
// the reference's rules by hand.
static void test_array_filters(WaveContext &wc)
{
    {   // tfilter_ArrayConcat: arrays of 2, 0 and 3 elements
        std::vector<HostColumn> cols(1);
        cols[0].set_arrays(std::vector<uint64_t>{2, 0, 3}, std::vector<int32_t>{1, 2, 3, 4, 5});
        PacketWave sent(wc, 7, 100, "%ad", cols, 3);
        PacketWave *got = received(wc, sent);
        HostColumn out;
        bool ok = tfilter_ArrayConcat(*got, out) == 0 && out.counts.size() == 1 && out.counts[0] == 5 &&
                  out.values.size() == 20 && !memcmp(out.values.data(), cols[0].values.data(), 20);
        std::vector<HostColumn> sc(1);
        sc[0] = scalars<int32_t>(3, 1);
        PacketWave scalar_wave(wc, 7, 100, "%d", sc, 3);
        ok = ok && tfilter_ArrayConcat(scalar_wave, out) == -1;        // not an array format: nothing is emitted
        end_SubTest("tfilter_ArrayConcat", ok);
        delete got;
    }
    {   // tfilter_PerfData: ranks 0, 1 and an aggregate of 2 ranks
        std::vector<HostColumn> cols(3);
        cols[0].set_arrays(std::vector<uint64_t>{1, 1, 1}, std::vector<int32_t>{0, 1, -2});
        cols[1].set_arrays(std::vector<uint64_t>{1, 1, 1}, std::vector<int32_t>{2, 2, 2});
        cols[2].set_arrays(std::vector<uint64_t>{2, 2, 2}, std::vector<double>{1, 2, 3, 4, 10, 20});
        PacketWave sent(wc, 7, 100, "%ad %ad %alf", cols, 3);
        PacketWave *got = received(wc, sent);
        std::vector<HostColumn> out;
        bool ok = tfilter_PerfData(*got, MRPDR_SUM, out) == 0 && out[0].as<int32_t>()[0] == -4 &&
                  out[1].as<int32_t>()[0] == 2 && out[2].counts[0] == 2 &&
                  out[2].as<double>()[0] == 14.0 && out[2].as<double>()[1] == 26.0;
        ok = ok && tfilter_PerfData(*got, MRPDR_AVG, out) == 0 && out[2].as<double>()[0] == 6.0 && out[2].as<double>()[1] == 11.5;
        ok = ok && tfilter_PerfData(*got, MRPDR_MAX, out) == 0 && out[2].as<double>()[0] == 10.0 && out[2].as<double>()[1] == 20.0;
        // MIN starts from bytes 0xFF, a NaN that the first '<' discards
        ok = ok && tfilter_PerfData(*got, MRPDR_MIN, out) == 0 && out[2].as<double>()[0] == 1.0 && out[2].as<double>()[1] == 2.0;
        ok = ok && tfilter_PerfData(*got, PERFDATA_ARRAY_CONCAT, out) == 0 && out[0].counts[0] == 3 && out[2].counts[0] == 6 &&
             out[0].as<int32_t>()[2] == -2 && out[2].as<double>()[5] == 20.0;
        end_SubTest("tfilter_PerfData(SUM, AVG, MAX, MIN, ARRAY_CONCAT)", ok);
        delete got;
    }
    {   // tfilter_IntEqClass: classes 5 and 3, then 3 and 7
        std::vector<HostColumn> cols(3);
        cols[0].set_arrays(std::vector<uint64_t>{2, 2}, std::vector<uint32_t>{5, 3, 3, 7});
        cols[1].set_arrays(std::vector<uint64_t>{2, 2}, std::vector<uint32_t>{2, 1, 1, 2});
        cols[2].set_arrays(std::vector<uint64_t>{3, 3}, std::vector<uint32_t>{10, 11, 12, 20, 30, 31});
        PacketWave sent(wc, 7, 100, "%aud %aud %aud", cols, 2);
        std::vector<HostColumn> out;
        const uint32_t v[3] = {3, 5, 7}, c[3] = {2, 2, 2}, m[6] = {12, 20, 10, 11, 30, 31};
        bool ok = tfilter_IntEqClass(sent, out) == 0 && out[0].counts[0] == 3 && out[1].counts[0] == 3 && out[2].counts[0] == 6 &&
                  !memcmp(out[0].values.data(), v, 12) && !memcmp(out[1].values.data(), c, 12) &&
                  !memcmp(out[2].values.data(), m, 24);
        end_SubTest("tfilter_IntEqClass", ok);
    }
}

// A wave whose odd packets came from big-endian senders: every packet is decoded with the byte
// order its own header announces (src/Packet.C:96,118).
static void test_both_byte_orders(WaveContext &wc)
{
    const char *fmt = "%d %ad";
    const uint64_t nb = 10;
    std::vector<HostColumn> cols(2);
    std::vector<int32_t> v(nb), arr(3 * nb);
    for (uint64_t i = 0; i < nb; i++) v[i] = (int32_t)(i * 1000 - 7);
    for (uint64_t i = 0; i < 3 * nb; i++) arr[i] = (int32_t)(0x01020300 + i);
    cols[0].set_scalars(v);
    cols[1].set_arrays(std::vector<uint64_t>(nb, 3), arr);
    PacketWave sent(wc, 9, 101, fmt, cols, nb);
    std::vector<uint8_t> hdrs, arena;
    std::vector<uint64_t> offs, lens;
    sent.get_Headers(hdrs);
    sent.get_Buffers(arena, offs, lens);
    const unsigned hl = sent.get_HeaderLen();
    const size_t flen = strlen(fmt) + 1;
    auto reverse = [](uint8_t *p, size_t w) { for (size_t k = 0; k < w / 2; k++) { uint8_t t = p[k]; p[k] = p[w - 1 - k]; p[w - 1 - k] = t; } };
    for (uint64_t i = 1; i < nb; i += 2) {
        uint8_t *h = hdrs.data() + i * hl, *b = arena.data() + offs[i];
        for (int a = 0; a < 16; a += 4) reverse(h + a, 4);      // stream id, tag, source rank, format length
        reverse(h + 16 + flen, 8);                              // destination count
        h[hl - 1] = MRPDR_BIGENDIAN;
        reverse(b, 4);                                          // %d
        reverse(b + 4, 8);                                      // array count
        for (int e = 0; e < 3; e++) reverse(b + 12 + 4 * e, 4);
    }
    PacketWave got(wc, fmt, nb, (const char *)hdrs.data(), hl, (const char *)arena.data(), offs.data(), lens.data(),
                   arena.size());
    std::vector<HostColumn> out;
    bool ok = !got.has_Error() && got.get_StreamId() == 9 && got.get_Tag() == 101 && got.unpack(fmt, out) == 0 && !got.has_Error();
    ok = ok && !memcmp(out[0].values.data(), v.data(), 4 * nb) && !memcmp(out[1].values.data(), arr.data(), 12 * nb);
    end_SubTest("wave with packets of both byte orders", ok);
}

// Packets of two formats in one arrival: split by format string on the host, one wave each.
static void test_group_by_format(WaveContext &wc)
{
    std::vector<HostColumn> a(1), b(2);
    a[0] = scalars<int32_t>(3, 5);
    b[0] = scalars<int32_t>(2, 6);
    b[1] = strings(2, "x");
    PacketWave wa(wc, 7, 100, "%d", a, 3), wb(wc, 7, 100, "%d %s", b, 2);
    std::vector<uint8_t> ha, hb, all;
    wa.get_Headers(ha);
    wb.get_Headers(hb);
    const unsigned la = wa.get_HeaderLen(), lb = wb.get_HeaderLen();
    // arrival order: a0 b0 a1 a2 b1
    std::vector<uint64_t> offs, lens;
    const int order[5] = {0, 100, 1, 2, 101};
    for (int k : order) {
        offs.push_back(all.size());
        if (k < 100) { all.insert(all.end(), ha.begin() + k * la, ha.begin() + (k + 1) * la); lens.push_back(la); }
        else { all.insert(all.end(), hb.begin() + (k - 100) * lb, hb.begin() + (k - 99) * lb); lens.push_back(lb); }
    }
    auto g = group_by_format(all.data(), all.size(), offs.data(), lens.data(), 5);
    bool ok = g.size() == 2 && g[0].first == "%d" && g[1].first == "%d %s" &&
              g[0].second == std::vector<uint64_t>({0, 2, 3}) && g[1].second == std::vector<uint64_t>({1, 4});
    end_SubTest("group_by_format", ok);
}

// Message::send / Message::recv for a wave (src/Message.C:48-335): layout of the message, and the
// wave that comes back out of it.
static void test_message(WaveContext &wc)
{
    const char *fmt = "%d %s %auhd";
    const uint64_t nb = 9;
    std::vector<HostColumn> cols(3);
    std::vector<int32_t> v(nb);
    std::vector<uint16_t> arr;
    std::vector<uint64_t> cnt(nb);
    for (uint64_t i = 0; i < nb; i++) { v[i] = (int32_t)i - 4; cnt[i] = i % 4; for (uint64_t k = 0; k < cnt[i]; k++) arr.push_back((uint16_t)(100 * i + k)); }
    cols[0].set_scalars(v);
    cols[1] = strings(nb, "Test String");
    cols[2].set_arrays(cnt, arr);
    PacketWave sent(wc, 77, 105, fmt, cols, nb);
    std::vector<uint32_t> src(nb);
    for (uint64_t i = 0; i < nb; i++) src[i] = (uint32_t)(1000 + i);
    std::vector<uint8_t> msg;
    sent.to_message(msg, &src);
    // layout: byte order, count, 2n sizes, spare byte, then hdr0 buf0 hdr1 buf1 ...
    std::vector<uint8_t> arena;
    std::vector<uint64_t> offs, lens;
    sent.get_Buffers(arena, offs, lens);
    const uint64_t h = sent.get_HeaderLen();
    bool ok = msg.size() > 5 && msg[0] == 0 && msg[1] == nb && msg[2] == 0 && msg[3] == 0 && msg[4] == 0;
    uint64_t at = 5 + 16 * nb + 1;
    for (uint64_t i = 0; ok && i < nb; i++) {
        uint64_t hs, bs;
        memcpy(&hs, &msg[5 + 16 * i], 8); memcpy(&bs, &msg[5 + 16 * i + 8], 8);
        ok = hs == h && bs == lens[i] && !memcmp(&msg[at + h], &arena[offs[i]], lens[i]);
        at += h + lens[i];
    }
    ok = ok && at == msg.size();
    if (!ok) printf("  (message layout differs)\n");
    PacketWave got(wc, fmt, msg.data(), msg.size());
    std::vector<HostColumn> out;
    ok = ok && !got.has_Error() && got.size() == nb && got.get_StreamId() == 77 && got.get_Tag() == 105 &&
         got.get_SourceRanks() == src && got.unpack(fmt, out) == 0 && !got.has_Error();
    ok = ok && out[0].values == cols[0].values && out[1].values == cols[1].values && out[2].values == cols[2].values &&
         out[2].counts == cnt;
    if (!ok) printf("  (received wave differs)\n");
    std::vector<uint8_t> again;
    got.to_message(again);                                      // forwarded: the same bytes go out again
    ok = ok && again == msg;
    std::vector<uint8_t> cut(msg.begin(), msg.end() - 2);
    bool refused = false;
    try { PacketWave bad(wc, fmt, cut.data(), cut.size()); } catch (const std::exception &) { refused = true; }
    end_SubTest("Message::send / recv of a wave", ok && refused);
}

int main()
{
    try {
        WaveContext wc(0, 1 << 16);
        test_golden(wc);
        for (uint64_t nb : {1, 4, 16}) test_alltypes(wc, nb);
        test_array<char>(wc, "%ac", -17);            test_array<unsigned char>(wc, "%auc", 17);
        test_array<int16_t>(wc, "%ahd", -17);        test_array<uint16_t>(wc, "%auhd", 17);
        test_array<int32_t>(wc, "%ad", -17);         test_array<uint32_t>(wc, "%aud", 17);
        test_array<int64_t>(wc, "%ald", -17);        test_array<uint64_t>(wc, "%auld", 17);
        test_array<float>(wc, "%af", 123.456789f);   test_array<double>(wc, "%alf", -123.456789);
        for (uint64_t nb : {1, 4, 16, 64}) {
            test_sum<char>(wc, "%c", 7, nb);                 test_sum<unsigned char>(wc, "%uc", 7, nb);
            test_sum<int16_t>(wc, "%hd", -17, nb);           test_sum<uint16_t>(wc, "%uhd", 17, nb);
            test_sum<int32_t>(wc, "%d", -117, nb);           test_sum<uint32_t>(wc, "%ud", 117, nb);
            test_sum<int64_t>(wc, "%ld", -177, nb);          test_sum<uint64_t>(wc, "%uld", 177, nb);
            test_sum<float>(wc, "%f", (float)123.450, nb);   test_sum<double>(wc, "%lf", 123.45678, nb);
        }
        test_filters_known_answers(wc);
        test_filters_default_shape(wc);
        test_column_views(wc);
        test_truncated(wc);
        test_string_arrays(wc);
        test_builder_reader(wc);
        test_array_filters(wc);
        test_both_byte_orders(wc);
        test_group_by_format(wc);
        test_message(wc);
    } catch (const std::exception &e) {
        printf("exception: %s\nMRNTEST_FAILURE\n", e.what());
        failures++;
    }
    printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
