out=$PWD/gpurun_out
python -X faulthandler -m pytest tests -x -v -m gpu > $out/t26_full.log 2>&1; echo "rc=$?" >> $out/t26_full.log; grep -c PASSED $out/t26_full.log; tail -3 $out/t26_full.log | cut -c1-300
