// Register / occupancy check of the tile kernels alone (tools/regs_tile.sh): compiles in seconds.
#include "../mrnet_amd/csrc/tile.h"
template __global__ void k_tile_encode<4, TILE_IMG>(FmtDev, TileVars, WireDev, ColsDev, uint64_t);
template __global__ void k_tile_decode<false, 4, TILE_IMG>(FmtDev, TileVars, WireDev, ColsDev, uint64_t, uint32_t *, const uint8_t *);
template __global__ void k_count_walk<false>(FmtDev, WireDev, ColsDev, uint64_t, int32_t *, uint8_t *, uint64_t *, uint64_t);
