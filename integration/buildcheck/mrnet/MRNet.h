// BUILD-CHECK STAND-IN.  This is NOT MRNet and not part of the product: it declares, with the
// signatures of include/mrnet/Packet.h:41-88 and include/mrnet/NetworkTopology.h:227-250 of
// MRNet 5.0.1, only the names integration/mrpdr_filters.C uses, so that `make plugin-check` can
// compile that file where no MRNet tree is installed.  In an MRNet build the real
// "mrnet/MRNet.h" is found first and this file plays no part.
#ifndef MRPDR_BUILDCHECK_MRNET_H
#define MRPDR_BUILDCHECK_MRNET_H

#include <stdint.h>
#include <memory>
#include <vector>

namespace MRN {

typedef uint32_t Rank;

class Packet {
 public:
    Packet(unsigned int istream_id, int itag, const char *ifmt, ...);
    Packet(unsigned int istream_id, int itag, const void **idata, const char *ifmt);
    ~Packet();
    void set_DestroyData(bool b);
    int get_Tag(void) const;
    unsigned int get_StreamId(void) const;
    const char *get_FormatString(void) const;
    const char *get_Buffer(void) const;
    uint64_t get_BufferLen(void) const;
    const char *get_Header(void) const;
    unsigned int get_HeaderLen(void) const;

    struct Impl;                // defined by whoever runs the plugin without MRNet (tests/host/test_plugin.cpp)
    Impl *impl_;
};

typedef std::shared_ptr<Packet> PacketPtr;      // boost::shared_ptr<Packet> in MRNet

class TopologyLocalInfo;

} // namespace MRN

#endif
