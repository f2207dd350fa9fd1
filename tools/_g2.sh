set -o pipefail
python -m pytest tests/test_train_step_gpu.py -x -q -m gpu > gpurun_out/t2a.log 2>&1; echo "rc=$?" >> gpurun_out/t2a.log; tail -4 gpurun_out/t2a.log
VDM4CDM_WG8=3 VDM4CDM_FORCE_TZ=4 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv or gn_bwd_folded" > gpurun_out/t2b.log 2>&1; echo "rc=$?" >> gpurun_out/t2b.log; tail -4 gpurun_out/t2b.log
for m in 0 3 0 3; do echo "WG8=$m"; VDM4CDM_WG8=$m python tools/conv_microbench.py --graph --only L0_ --ops fwd,fwd_gn,dgrad,dgrad_gn --iters 20 2>&1 | grep -E "L0_32_32|L0_64_32 "; done > gpurun_out/t2c.log 2>&1; cat gpurun_out/t2c.log
for m in 0 3 0 3; do echo "WG8=$m"; VDM4CDM_WG8=$m python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events --sample-steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sample']['seconds'])"; done > gpurun_out/t2d.log 2>&1; cat gpurun_out/t2d.log
