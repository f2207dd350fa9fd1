#!/bin/bash
# Where does the step time go?  bench.py with one kernel family NOT launched at a time (wrong results, timing only): the drop in
# ms/step is what that family costs the step INCLUDING its contention with the other streams (the sum of in-step kernel durations
# over-counts: the streams time-slice).   usage: bash tools/ablate_step.sh > gpurun_out/ablate.txt
run() { VDM4CDM_ABLATE=$1 python bench.py --steps 30 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.3f ms/step' % ('$2', d['ms_per_step']))"; }
run "" "full step"
VDM4CDM_ABLATE_REDUCE=1 run "" "no wgrad slab reduces"
run wgrad "no weight gradients"
run wgrad1 "no 1x1 weight gradients"
run conv1 "no 1x1 convs (fwd+dgrad)"
run gn_fwd "no gn_silu_fwd"
run gn_apply "no gn_bwd_apply"
run pack "no weight re-packing"
run "" "full step (again)"
