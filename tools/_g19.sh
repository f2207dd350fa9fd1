out=$PWD/gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "test_conv_grads and bf16" > $out/t19b.log 2>&1; echo "rc=$?" >> $out/t19b.log; tail -3 $out/t19b.log
grep -q "rc=0" $out/t19b.log || exit 1
for v in 0 1 0 1; do echo "WGRAD_ROWS=$v"; VDM4CDM_WGRAD_ROWS=$v python tools/conv_microbench.py --graph --ops wgrad --iters 50 2>&1 | grep -i "wgrad" | head -9; done > $out/t19_micro.log 2>&1
cat $out/t19_micro.log
for wgs in 512 256; do VDM4CDM_WGRAD_WGS=$wgs VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_timeline.so timeout -k 10 120 python tools/wgrad_phases.py --shape L0_32_32 2>&1 | grep -v amdgpu.ids; done
for v in 0 1 0 1; do echo "WGRAD_ROWS=$v"; VDM4CDM_WGRAD_ROWS=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done > $out/t19_step.log 2>&1
cat $out/t19_step.log
