#!/bin/bash
# usage: tools/kernel_resources.sh vdm4cdm_amd/csrc/conv.hip  -> VGPR / AGPR / scratch / occupancy per kernel (gfx950)
src=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -c "$src" -o /tmp/_kr.o \
  -Rpass-analysis=kernel-resource-usage 2>/tmp/_kr.txt
grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" /tmp/_kr.txt | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | paste - - - - - | sed -E 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/' \
 | while IFS=$'\t' read n a b c d; do printf "%-64s %s %s %s %s\n" "$(echo $n | c++filt | sed -E 's/vdm:://g; s/\(.*//; s/void //')" "$a" "$b" "$c" "$d"; done
