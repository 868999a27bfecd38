#!/bin/bash
# kernel resource usage (SGPR/VGPR/spills/occupancy/LDS) of one csrc/*.hip file:  tools/kres.sh scan_fwd_chan [filter]
cd /root/repo/vivim_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage \
  -c "$1.hip" -o /tmp/kres.o 2>&1 | grep -E "error|Function Name|SGPRs:|VGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs Spill" \
  | paste - - - - - - - | sed 's/remark: [^ ]* //g; s/\[-Rpass[^]]*\]//g; s/[a-z_]*\.hip:[0-9]*:[0-9]*://g; s/Function Name: //; s/  */ /g' \
  | grep -E "error|${2:-.}"
