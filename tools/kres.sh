#!/bin/bash
# kernel resource usage of a .hip file: name, SGPR, VGPR, AGPR, scratch, occupancy (one line per kernel)
# usage: tools/kres.sh file.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/kres.o 2>&1 |
  grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs|ScratchSize|Occupancy" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - | sed -E 's/Function Name: //'
