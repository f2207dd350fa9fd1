#!/bin/bash
for cfg in tiny c2 c224; do
python -X faulthandler bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --sample-steps 0 > gpurun_out/r03_g_$cfg.json 2> gpurun_out/r03_g_$cfg.err; echo "$cfg rc $?"
python -c "
import json
try:
    d=json.loads(open('gpurun_out/r03_g_$cfg.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('graph_step'))
except Exception as e: print('no line', e)"
done
