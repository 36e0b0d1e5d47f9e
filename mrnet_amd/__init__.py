"""mrnet_amd -- MI355X-native wave-at-a-time PDR codec for MRNet's packet marshalling path.

The compute path is the HIP library mrnet_amd/lib/libmrpdr.so (C ABI: include/mrpdr.h);
this package is the host side that mirrors the reference's Packet / filter surface.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
