#!/bin/bash
# full GPU suite + default bench line + per-launch timeline of one step (tag = $1)
set -o pipefail
tag=${1:-r03b}
out=$PWD/gpurun_out
mkdir -p "$out"
python -m pytest tests -m gpu -x -q -s > "$out/${tag}_gputests.log" 2>&1; echo "pytest rc $?" >> "$out/${tag}_gputests.log"
tail -4 "$out/${tag}_gputests.log"
grep -E "C5 chain|expansive chain" "$out/${tag}_gputests.log"
python bench.py --steps 40 > "$out/${tag}_bench_c3.json" 2> "$out/${tag}_bench_c3.err"; echo "bench rc $?"
python -c "
import json,sys
d=json.loads(open('$out/${tag}_bench_c3.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'sample', d.get('sample'))
print('dominant', d['roofline'].get('kernel'), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'))"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events --sample-steps 0 > "$out/${tag}_trace_bench.json" 2> "$out/${tag}_trace.err"
f=$(ls "$out"/${tag}_trace/*/*kernel_trace.csv | head -1)
python tools/step_timeline.py "$f" 2 > "$out/${tag}_step_timeline.txt"
cp "$(ls "$out"/${tag}_trace/*/*kernel_stats.csv | head -1)" "$out/${tag}_kernel_stats_c3.csv"
rm -rf "$out/${tag}_trace"
tail -1 "$out/${tag}_step_timeline.txt"
echo done
