#!/bin/bash
out=$PWD/gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_grads or conv_large" > $out/r03_exp7_tests.log 2>&1; rc=$?; tail -3 $out/r03_exp7_tests.log
if [ $rc -ne 0 ]; then echo "tests failed"; exit 1; fi
for r in 1 0 1 0; do VDM4CDM_WGRAD_REDUCE2=$r python bench.py --steps 40 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reduce2=$r', d['ms_per_step'])"; done | tee $out/r03_exp7_ab.txt
