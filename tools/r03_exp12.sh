#!/bin/bash
out=$PWD/gpurun_out
echo "== full staging" > $out/r03_exp12.txt
python tools/conv_microbench.py --ops fwd,dgrad_gn --graph --only "L0_32_32" >> $out/r03_exp12.txt 2>&1
python tools/conv_microbench.py --ops fwd --graph --only "L1_64_64" >> $out/r03_exp12.txt 2>&1
for v in stage4 stage2; do echo "== $v (of 6 sixths of the halo chunks staged; wrong results)" >> $out/r03_exp12.txt
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_$v.so python tools/conv_microbench.py --ops fwd,dgrad_gn --graph --only "L0_32_32" >> $out/r03_exp12.txt 2>&1
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_$v.so python tools/conv_microbench.py --ops fwd --graph --only "L1_64_64" >> $out/r03_exp12.txt 2>&1; done
grep -v amdgpu $out/r03_exp12.txt
