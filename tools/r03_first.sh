#!/bin/bash
# first GPU checkpoint of round 3: GPU test suite, default bench line, per-launch timeline of the training step
set -o pipefail
out=$PWD/gpurun_out
mkdir -p "$out"
python -m pytest tests -m gpu -x -q -s > "$out/r03a_gputests.log" 2>&1; echo "pytest rc $?" >> "$out/r03a_gputests.log"
tail -5 "$out/r03a_gputests.log"
python bench.py --steps 40 > "$out/r03a_bench_c3.json" 2> "$out/r03a_bench_c3.err"; echo "bench rc $?"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r03a_trace" -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events --sample-steps 0 > "$out/r03a_trace_bench.json" 2> "$out/r03a_trace.err"
f=$(ls "$out"/r03a_trace/*/*kernel_trace.csv | head -1)
python tools/step_timeline.py "$f" 2 > "$out/r03a_step_timeline.txt"
cp "$(ls "$out"/r03a_trace/*/*kernel_stats.csv | head -1)" "$out/r03a_kernel_stats_c3.csv"
rm -rf "$out/r03a_trace"
tail -3 "$out/r03a_step_timeline.txt"
echo done
