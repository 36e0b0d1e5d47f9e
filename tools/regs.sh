#!/bin/bash
# usage: tools/regs.sh <kernel-name-substring> [extra hipcc flags]: registers, LDS and occupancy of the matching kernels
pat=$1; shift
hipcc --offload-arch=gfx950 -O3 -fwrapv -ffp-contract=off -std=c++17 --cuda-device-only -c -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" mrnet_amd/csrc/mrpdr.hip 2>&1 |
  grep -A9 "Function Name: .*$pat" | grep -E "Function Name|VGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//'
