out=$PWD/gpurun_out
timeout -k 10 600 python -m pytest tests/test_unet_gpu.py -x -q > $out/t_u.log 2>&1 || exit 1
rm -f $out/b_ab.log
for i in 1 2; do
for f in 1 0; do
VDM4CDM_WGRAD_STREAM=$f timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --sample-steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side=$f ms_per_step', d['ms_per_step'], d['sample']['seconds'])" >> $out/b_ab.log
done; done
