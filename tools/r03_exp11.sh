#!/bin/bash
out=$PWD/gpurun_out
timeout -k 10 500 python -m pytest tests/test_unet_gpu.py -x -q -k "backward or training or packed" > $out/r03_exp11_tests.log 2>&1; rc=$?; tail -3 $out/r03_exp11_tests.log
if [ $rc -ne 0 ]; then echo "tests failed"; exit 1; fi
for r in 0 1 2 3 0 1 2; do VDM4CDM_DEFER_WGRAD_LEVELS=$r python bench.py --steps 40 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer levels=$r', d['ms_per_step'])"; done | tee $out/r03_exp11_ab.txt
