set -o pipefail
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fused_dgrad_wgrad" > gpurun_out/t10a.log 2>&1; echo "rc=$?" >> gpurun_out/t10a.log; tail -15 gpurun_out/t10a.log
python -m pytest tests/test_unet_gpu.py tests/test_train_step_gpu.py -x -q -m gpu -k "c3_backward or c3_dropout or backward_bf16 or fused_head" > gpurun_out/t10b.log 2>&1; echo "rc=$?" >> gpurun_out/t10b.log; tail -5 gpurun_out/t10b.log
for v in 1 0 1 0; do echo "DGW=$v"; VDM4CDM_FUSED_DGW=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])"; done > gpurun_out/t10d.log 2>&1; cat gpurun_out/t10d.log
