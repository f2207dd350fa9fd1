#!/bin/bash
out=$PWD/gpurun_out
echo "== prefetch before the taps (default)" > $out/r03_exp10.txt
python tools/conv_microbench.py --ops dgrad,dgrad_gn --graph --only "L0_32_32" >> $out/r03_exp10.txt 2>&1
echo "== operands fetched in the epilogue (rolling window)" >> $out/r03_exp10.txt
VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_gnbnopf.so python tools/conv_microbench.py --ops dgrad,dgrad_gn --graph --only "L0_32_32" >> $out/r03_exp10.txt 2>&1
grep -v amdgpu $out/r03_exp10.txt
for r in a b a b; do if [ $r = a ]; then unset VDM4CDM_LIB; else export VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_gnbnopf.so; fi; python bench.py --steps 40 --no-cpu-baseline --no-kernel-events --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $r', d['ms_per_step'])"; done
