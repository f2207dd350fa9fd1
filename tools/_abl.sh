out=$PWD/gpurun_out
timeout -k 10 600 python -m pytest tests/test_unet_gpu.py tests/test_entry_gpu.py -x -q > $out/t_u.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --sample-steps 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['sample'])" > $out/b_s.log
