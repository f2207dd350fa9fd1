#!/bin/bash
out=$PWD/gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "gn_" > $out/r03_exp8_tests.log 2>&1; rc=$?; tail -3 $out/r03_exp8_tests.log
if [ $rc -ne 0 ]; then echo "tests failed"; exit 1; fi
python tools/sampler_profile.py --steps 300 2>/dev/null
python tools/sampler_profile.py --steps 300 2>/dev/null
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r03d_samp_stats" -- python3 tools/sampler_profile.py --steps 100 > /dev/null 2> "$out/r03d_samp.err"
cp "$(ls "$out/r03d_samp_stats"/*/*kernel_stats.csv | head -1)" "$out/r03d_rocprofv3_kernel_stats_sampler_c5.csv"
rm -rf "$out/r03d_samp_stats"
grep "gn_" "$out/r03d_rocprofv3_kernel_stats_sampler_c5.csv" | cut -c1-200
