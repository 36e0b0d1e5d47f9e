#!/bin/bash
# usage: tools/regs_tile.sh [extra hipcc flags]: registers, LDS and occupancy of the tile kernels (tools/tile_only.hip)
hipcc --offload-arch=gfx950 -O3 -fwrapv -ffp-contract=off -std=c++17 --cuda-device-only -c -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" tools/tile_only.hip 2>&1 |
  grep -E "Function Name|VGPRs:|VGPRs Spill|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//'
