#!/bin/bash
out=$PWD/gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_fwd or conv_grads or conv_large" > $out/r03_exp14_tests.log 2>&1; rc=$?; tail -3 $out/r03_exp14_tests.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
echo "== two classes per workgroup" > $out/r03_exp14.txt
python tools/conv_microbench.py --ops fwd,dgrad --graph --only "up_64_32" >> $out/r03_exp14.txt 2>&1
python tools/conv_microbench.py --ops dgrad --graph --only "down_32_32" >> $out/r03_exp14.txt 2>&1
echo "== one class per workgroup" >> $out/r03_exp14.txt
VDM4CDM_CLS_CPW=1 python tools/conv_microbench.py --ops fwd,dgrad --graph --only "up_64_32" >> $out/r03_exp14.txt 2>&1
grep -v amdgpu $out/r03_exp14.txt
for r in 2 1 2 1; do VDM4CDM_CLS_CPW=$r python bench.py --steps 40 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cpw=$r', d['ms_per_step'])"; done
for r in 2 1; do VDM4CDM_CLS_CPW=$r python tools/sampler_profile.py --steps 300 2>/dev/null; done
