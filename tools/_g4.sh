set -o pipefail
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/t4a.log 2>&1; echo "rc=$?" >> gpurun_out/t4a.log; tail -3 gpurun_out/t4a.log
bash tools/kernel_counters.sh r04c --only L0_32_32 --ops fwd > gpurun_out/kc_r04c_fwd.txt 2>&1; cat gpurun_out/kc_r04c_fwd.txt
bash tools/kernel_counters.sh r04c --only L0_32_32 --ops dgrad_gn > gpurun_out/kc_r04c_gnb.txt 2>&1; cat gpurun_out/kc_r04c_gnb.txt
for i in 1 2; do python tools/conv_microbench.py --graph --ops fwd,fwd_gn,dgrad,dgrad_gn,wgrad --iters 20 --only L 2>&1 | grep -E "^L[0-1]_"; done > gpurun_out/t4c.log 2>&1; cat gpurun_out/t4c.log
for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events --sample-steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sample']['seconds'])"; done > gpurun_out/t4d.log 2>&1; cat gpurun_out/t4d.log
