#!/bin/bash
# BASELINE side configurations: the reference's one published rate (256^3, 250-step sample, chs 16..128, circular) and the C2 fp32 line
out=$PWD/gpurun_out
python bench.py --config c256 --sample-only --sample-steps 250 > $out/r03_sample_c256.json 2> $out/r03_sample_c256.err; echo "c256 rc $?"
python bench.py --config c2 --steps 40 --sample-steps 0 > $out/r03_bench_c2.json 2> $out/r03_bench_c2.err; echo "c2 rc $?"
python - <<'PY'
import json
for f in ("r03_sample_c256", "r03_bench_c2"):
    d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d.get("ms_per_step"), d.get("sample"), {k: d["roofline"].get(k) for k in ("kernel", "frac", "avg_launch_ms", "step_hbm_frac", "step_mfma_frac")})
PY
