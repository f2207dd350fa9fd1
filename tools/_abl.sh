export TMPDIR=/tmp
out=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $out/st_r.json 2> $out/st_r.err
cp $out/st_r/*/*kernel_stats.csv $out/st_r_stats.csv; rm -rf $out/st_r
