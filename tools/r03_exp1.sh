#!/bin/bash
out=$PWD/gpurun_out
echo "== baseline lib" > $out/r03_exp1.txt
python tools/conv_microbench.py --only L0_ --ops fwd,fwd_gn,dgrad,dgrad_gn,wgrad --graph >> $out/r03_exp1.txt 2>&1
python tools/conv_microbench.py --only L1_64_64 --ops fwd,fwd_gn,dgrad,dgrad_gn,wgrad --graph >> $out/r03_exp1.txt 2>&1
echo "== no weight loads in the tap loop (wrong results, timing only)" >> $out/r03_exp1.txt
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_nowload.so python tools/conv_microbench.py --only L0_ --ops fwd,fwd_gn,dgrad,dgrad_gn --graph >> $out/r03_exp1.txt 2>&1
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_nowload.so python tools/conv_microbench.py --only L1_64_64 --ops fwd,dgrad_gn --graph >> $out/r03_exp1.txt 2>&1
cat $out/r03_exp1.txt
