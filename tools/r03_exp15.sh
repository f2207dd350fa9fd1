#!/bin/bash
for r in "1 1" "0 1" "1 0" "0 0" "1 1"; do set -- $r; VDM4CDM_WGRAD_STREAM=$1 VDM4CDM_SKIP_DGRAD_STREAM=$2 python bench.py --steps 40 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_stream=$1 skip_stream=$2', d['ms_per_step'])"; done
