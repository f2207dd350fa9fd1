out=$PWD/gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_train_step_gpu.py -x -q -m gpu > $out/t20.log 2>&1; echo "rc=$?" >> $out/t20.log; tail -3 $out/t20.log
grep -q "rc=0" $out/t20.log || exit 1
for v in 1 0 1 0; do echo "FUSED_DGW=$v"; VDM4CDM_FUSED_DGW=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done > $out/t20_step.log 2>&1
cat $out/t20_step.log
