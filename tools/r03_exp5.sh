#!/bin/bash
out=$PWD/gpurun_out
echo "== baseline" > $out/r03_exp5.txt
python tools/conv_microbench.py --ops wgrad --graph --only "L0_32_32" >> $out/r03_exp5.txt 2>&1
python tools/conv_microbench.py --ops wgrad --graph --only "L1_64_64" >> $out/r03_exp5.txt 2>&1
echo "== half of the transposed x reads skipped (wrong results)" >> $out/r03_exp5.txt
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_halfreads.so python tools/conv_microbench.py --ops wgrad --graph --only "L0_32_32" >> $out/r03_exp5.txt 2>&1
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_halfreads.so python tools/conv_microbench.py --ops wgrad --graph --only "L1_64_64" >> $out/r03_exp5.txt 2>&1
grep -v amdgpu $out/r03_exp5.txt
