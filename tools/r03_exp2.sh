#!/bin/bash
out=$PWD/gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_grads or conv_large or cond_table" > $out/r03_exp2_tests.log 2>&1; tail -3 $out/r03_exp2_tests.log
echo "== rolling wgrad" > $out/r03_exp2.txt
python tools/conv_microbench.py --ops wgrad --graph >> $out/r03_exp2.txt 2>&1
echo "== independent tiles (VDM4CDM_WGRAD_ROLL=0)" >> $out/r03_exp2.txt
VDM4CDM_WGRAD_ROLL=0 python tools/conv_microbench.py --ops wgrad --graph >> $out/r03_exp2.txt 2>&1
cat $out/r03_exp2.txt
for r in 1 0 1 0; do VDM4CDM_WGRAD_ROLL=$r python bench.py --steps 40 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('roll=$r', d['ms_per_step'])"; done | tee $out/r03_exp2_ab.txt
