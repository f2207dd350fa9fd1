#!/bin/bash
out=$PWD/gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_grads or conv_large" > $out/r03_exp3_tests.log 2>&1; rc=$?; tail -5 $out/r03_exp3_tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
echo "== wgrad gen 2 (256 workgroups)" > $out/r03_exp3.txt
python tools/conv_microbench.py --ops wgrad --graph >> $out/r03_exp3.txt 2>&1
echo "== wgrad gen 2, 128 workgroups" >> $out/r03_exp3.txt
VDM4CDM_WGRAD2_WGS=128 python tools/conv_microbench.py --ops wgrad --graph --only L0_ >> $out/r03_exp3.txt 2>&1
echo "== gen 1" >> $out/r03_exp3.txt
VDM4CDM_WGRAD_GEN=1 python tools/conv_microbench.py --ops wgrad --graph >> $out/r03_exp3.txt 2>&1
cat $out/r03_exp3.txt
for r in "2 256" "1 256" "2 192" "2 128" "1 256" "2 256"; do set -- $r; VDM4CDM_WGRAD_GEN=$1 VDM4CDM_WGRAD2_WGS=$2 python bench.py --steps 40 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gen=$1 wgs=$2', d['ms_per_step'])"; done | tee $out/r03_exp3_ab.txt
