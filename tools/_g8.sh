export TMPDIR=/tmp
out=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04d_stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --sample-steps 0 > "$out/r04d_bench.json" 2> "$out/r04d_stats.err"
cp "$(ls $out/r04d_stats/*/*kernel_stats.csv | head -1)" "$out/r04d_kernel_stats.csv"; rm -rf "$out/r04d_stats"
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04d_kernel_stats.csv')))
steps=15
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms/step', tot/steps/1e6)
for r in rows[:34]:
    c=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{c/steps:6.1f} x {t/c/1e3:7.1f} us = {t/steps/1e6:6.3f} ms  {r['Name'][:110]}")
PY
