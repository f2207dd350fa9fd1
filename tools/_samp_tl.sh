#!/bin/bash
set -e
out=$PWD/gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/samp_tr" -- python3 tools/sampler_profile.py --steps 40 > /dev/null 2> "$out/samp_tr.err"
python tools/step_timeline.py "$(ls "$out"/samp_tr/*/*kernel_trace.csv | head -1)" 5 pack_input_kernel > "$out/sampler_step_timeline.txt"
rm -rf "$out/samp_tr"
tail -1 "$out/sampler_step_timeline.txt"
