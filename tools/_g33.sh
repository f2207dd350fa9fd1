out=$PWD/gpurun_out
VDM4CDM_ROLL_STG=3 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "conv_fwd or conv_grads or gn_bwd_folded or unet_forward or unet_backward or full_size_128 or c3_ or sampler_matches" > $out/t33.log 2>&1; echo "rc=$?" >> $out/t33.log; tail -2 $out/t33.log | cut -c1-200
grep -q "rc=0" $out/t33.log || exit 1
for v in 0 1 3 0 1 3; do echo "ROLL_STG=$v"; VDM4CDM_ROLL_STG=$v python tools/conv_microbench.py --graph --ops fwd,dgrad,dgrad_gn --iters 50 --only L0_32_32 2>&1 | grep L0_32; done
for v in 0 1 3 0 1 3; do echo "ROLL_STG=$v"; VDM4CDM_ROLL_STG=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
for v in 0 1 0 1; do VDM4CDM_ROLL_STG=$v python tools/sampler_profile.py --steps 300; done
