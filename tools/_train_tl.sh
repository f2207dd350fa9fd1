#!/bin/bash
set -e
out=$PWD/gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/tr_tr" -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events --sample-steps 0 > /dev/null 2> "$out/tr_tr.err"
python tools/step_timeline.py "$(ls "$out"/tr_tr/*/*kernel_trace.csv | head -1)" 2 > "$out/train_step_timeline.txt"
rm -rf "$out/tr_tr"
tail -1 "$out/train_step_timeline.txt"
