out=$PWD/gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gn_bwd_folded or fused_dgrad or conv_fwd or conv_grads" > $out/t21.log 2>&1; echo "rc=$?" >> $out/t21.log; tail -3 $out/t21.log
grep -q "rc=0" $out/t21.log || exit 1
for v in prev cur prev cur; do echo "LIB=$v"; if [ $v = prev ]; then export VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_prev.so; else unset VDM4CDM_LIB; fi; python tools/conv_microbench.py --graph --ops dgrad_gn --iters 50 2>&1 | grep -i "dgrad_gn" | head -9; done > $out/t21_micro.log 2>&1
cat $out/t21_micro.log
for v in prev cur prev cur; do echo "LIB=$v"; if [ $v = prev ]; then export VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_prev.so; else unset VDM4CDM_LIB; fi; python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done > $out/t21_step.log 2>&1
cat $out/t21_step.log
