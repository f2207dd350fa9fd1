#!/bin/bash
out=$PWD/gpurun_out
: > $out/r03_exp6.txt
for v in "" NOLDSREAD NOMFMA NOSTAGE NOSTAGE_NOLDS; do
  echo "== variant: ${v:-baseline}" >> $out/r03_exp6.txt
  if [ -z "$v" ]; then python tools/conv_microbench.py --ops wgrad --graph --only "L0_32_32" >> $out/r03_exp6.txt 2>&1
  else VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_$v.so python tools/conv_microbench.py --ops wgrad --graph --only "L0_32_32" >> $out/r03_exp6.txt 2>&1; fi
done
VDM4CDM_ABLATE_REDUCE=1 python tools/conv_microbench.py --ops wgrad --graph --only "L0_32_32" >> $out/r03_exp6.txt 2>&1
grep -v amdgpu $out/r03_exp6.txt
