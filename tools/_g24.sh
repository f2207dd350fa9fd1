out=$PWD/gpurun_out
timeout -k 10 300 python tools/repro_bits.py 128 2>&1 | grep "rep " 
python -m pytest tests -x -q -m gpu > $out/t24_full.log 2>&1; echo "rc=$?" >> $out/t24_full.log; tail -4 $out/t24_full.log
grep -q "rc=0" $out/t24_full.log || exit 1
python bench.py --steps 60 --warmup 10 --no-cpu-baseline > $out/t24_bench.json 2>$out/t24_bench.err
python -c "import json;d=json.load(open('gpurun_out/t24_bench.json'));print(d['ms_per_step'], d['sample']['seconds'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['largest_by_total_time']['kernel'], d['roofline']['largest_by_total_time']['frac'])"
