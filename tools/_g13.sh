export TMPDIR=/tmp
out=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c2_stats" -- python3 bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --sample-steps 0 > /dev/null 2> "$out/c2_stats.err"
cp "$(ls $out/c2_stats/*/*kernel_stats.csv | head -1)" "$out/r04_c2_kernel_stats.csv"; rm -rf "$out/c2_stats"
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04_c2_kernel_stats.csv')))
steps=15
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms/step', tot/steps/1e6)
for r in rows[:24]:
    c=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{c/steps:6.1f} x {t/c/1e3:7.1f} us = {t/steps/1e6:6.3f} ms  {r['Name'][:105]}")
PY
