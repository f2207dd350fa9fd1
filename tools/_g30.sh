out=$PWD/gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_train_step_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "dropout or gn_silu or gn_bwd or c3_ or graphed or hipgraph" > $out/t30.log 2>&1; echo "rc=$?" >> $out/t30.log; tail -3 $out/t30.log | cut -c1-200
grep -q "rc=0" $out/t30.log || exit 1
for v in prev cur prev cur; do echo "LIB=$v"; if [ $v = prev ]; then export VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_prev.so; else unset VDM4CDM_LIB; fi; python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
unset VDM4CDM_LIB
python tools/sampler_profile.py --steps 300
