#!/bin/bash
out=$PWD/gpurun_out
VDM4CDM_WGRAD_GEN=2 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_grads or conv_large" > $out/r03_exp13_tests.log 2>&1; rc=$?; tail -3 $out/r03_exp13_tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
echo "== wgrad gen 2 with per-run staging tables" > $out/r03_exp13.txt
VDM4CDM_WGRAD_GEN=2 python tools/conv_microbench.py --ops wgrad --graph >> $out/r03_exp13.txt 2>&1
echo "== gen 1" >> $out/r03_exp13.txt
python tools/conv_microbench.py --ops wgrad --graph >> $out/r03_exp13.txt 2>&1
grep -v amdgpu $out/r03_exp13.txt
for r in 2 1 2 1; do VDM4CDM_WGRAD_GEN=$r python bench.py --steps 40 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gen=$r', d['ms_per_step'])"; done | tee $out/r03_exp13_ab.txt
