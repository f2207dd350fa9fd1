#!/bin/bash
# Round checkpoint on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC traffic passes.
# usage (from the repo root on the GPU box): bash tools/round_profile.sh <tag>     outputs under gpurun_out/<tag>_*
set -e
tag=${1:-r01}
out=$PWD/gpurun_out
mkdir -p "$out"
python bench.py > "$out/${tag}_bench_c3.json" 2> "$out/${tag}_bench_c3.err"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --all-kernel-events --sample-steps 0 > "$out/${tag}_stats_bench.json" 2> "$out/${tag}_stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_f" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --sample-steps 0 > /dev/null 2> "$out/${tag}_pmc_f.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_w" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --sample-steps 0 > /dev/null 2> "$out/${tag}_pmc_w.err"
python tools/pmc_traffic.py "$out/${tag}_pmc_f" "$out/${tag}_pmc_w" "$out/${tag}_pmc_bench_traffic.json"
f=$(ls "$out/${tag}_stats"/*/*kernel_stats.csv | head -1)
cp "$f" "$out/${tag}_rocprofv3_kernel_stats_c3.csv"
rm -rf "$out/${tag}_stats"/*/*kernel_trace.csv "$out/${tag}_pmc_f" "$out/${tag}_pmc_w"
# the sampler alone (BASELINE config C5: 128^3, batch 1, hipGraph-captured step)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_samp_stats" -- python3 tools/sampler_profile.py --steps 100 > "$out/${tag}_sampler_profile.json" 2> "$out/${tag}_samp.err"
cp "$(ls "$out/${tag}_samp_stats"/*/*kernel_stats.csv | head -1)" "$out/${tag}_rocprofv3_kernel_stats_sampler_c5.csv"
rm -rf "$out/${tag}_samp_stats" "$out/${tag}_stats"
echo done
