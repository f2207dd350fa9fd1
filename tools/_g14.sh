out=$PWD/gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "f32 or fp32" > $out/t14a.log 2>&1; echo "rc=$?" >> $out/t14a.log; tail -3 $out/t14a.log
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "test_conv_grads and bf16" > $out/t14b.log 2>&1; echo "rc=$?" >> $out/t14b.log; tail -3 $out/t14b.log
grep -q "rc=0" $out/t14b.log || exit 1
for v in 0 1 0 1; do echo "ROWS=$v"; VDM4CDM_WGRAD_ROWS=$v python tools/conv_microbench.py --graph --ops wgrad --iters 50 2>&1 | grep -i "wgrad" | head -8; done > $out/t14_micro.log 2>&1
cat $out/t14_micro.log
for v in 0 1 0 1; do echo "ROWS=$v"; VDM4CDM_WGRAD_ROWS=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done > $out/t14_step.log 2>&1
cat $out/t14_step.log
python bench.py --config c2 --steps 40 --warmup 5 --no-cpu-baseline --sample-steps 0 > $out/r04_bench_c2.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r04_bench_c2.json'));print(d['ms_per_step'], d['dtype'], d['roofline'].get('kernel'), d['roofline'].get('frac'))"
