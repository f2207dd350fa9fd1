out=$PWD/gpurun_out
for i in 1 2; do
python -X faulthandler -m pytest tests -x -q -m gpu > $out/final2_full_$i.log 2>&1; echo "rc=$?" >> $out/final2_full_$i.log; tail -2 $out/final2_full_$i.log | cut -c1-160
done
