#!/bin/bash
out=$PWD/gpurun_out
export VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_timeline.so
: > $out/r03_gnb_timeline.txt
for sh in L0_32_32 L0_32_64; do for op in dgrad dgrad_gn; do python tools/conv_timeline.py --shape $sh --op $op >> $out/r03_gnb_timeline.txt 2>&1; done; done
grep -v amdgpu $out/r03_gnb_timeline.txt | grep -E "N=2|tap loop|epilogue|entry->DMA|landed|whole life|per-CU wall|inside the tap"
