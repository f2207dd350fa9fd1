out=$PWD/gpurun_out
export VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_timeline.so
for v in 0 1; do
  VDM4CDM_WGRAD_ROWS=$v timeout -k 10 120 python tools/wgrad_phases.py --shape L0_32_32 || exit 1
  VDM4CDM_WGRAD_ROWS=$v timeout -k 10 120 python tools/wgrad_phases.py --shape L0_64_32 || exit 1
  VDM4CDM_WGRAD_ROWS=$v timeout -k 10 120 python tools/wgrad_phases.py --shape L2_128_128 || exit 1
done > $out/r04_wgrad_phases.txt 2>&1
timeout -k 10 120 python tools/wgrad_phases.py --shape L0_32_32 --stride 2 >> $out/r04_wgrad_phases.txt 2>&1
cat $out/r04_wgrad_phases.txt
