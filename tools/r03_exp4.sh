#!/bin/bash
out=$PWD/gpurun_out
export VDM4CDM_WGRAD_GEN=1
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_la2.so timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_grads" > $out/r03_exp4_tests.log 2>&1; rc=$?; tail -3 $out/r03_exp4_tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
echo "== LA=2" > $out/r03_exp4.txt
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_la2.so python tools/conv_microbench.py --ops wgrad --graph --only "_" >> $out/r03_exp4.txt 2>&1
echo "== LA=1" >> $out/r03_exp4.txt
python tools/conv_microbench.py --ops wgrad --graph >> $out/r03_exp4.txt 2>&1
cat $out/r03_exp4.txt
