#!/bin/bash
set -e
out=$PWD/gpurun_out; mkdir -p $out
if [ -z "$NOTEST" ]; then timeout -k 10 900 python -m pytest tests/test_unet_gpu.py -x -q > $out/skip_unet_tests.log 2>&1 || { tail -40 $out/skip_unet_tests.log; exit 1; }; fi
[ -z "$NOTEST" ] && tail -2 $out/skip_unet_tests.log
for rep in 1 2; do
for f in 0 1; do
  echo "FUSED_SKIP=$f"
  VDM4CDM_FUSED_SKIP=$f python bench.py --steps 60 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train ms/step', d['ms_per_step'])"
  VDM4CDM_FUSED_SKIP=$f python tools/sampler_profile.py --steps 200 2>/dev/null
done
done
