#!/bin/bash
# Round checkpoint on the GPU box: default bench line, rocprofv3 kernel stats + per-launch timeline of the same command, PMC traffic passes,
# the sampler alone, the step ablation.   usage (repo root, GPU box): bash tools/round_profile.sh <tag>     outputs under gpurun_out/<tag>_*
set -e
tag=${1:-r04}
out=$PWD/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --all-kernel-events --sample-steps 0 > "$out/${tag}_bench_c3_all_kernel_events_under_rocprof.json" 2> "$out/${tag}_stats.err"
f=$(ls "$out/${tag}_stats"/*/*kernel_stats.csv | head -1)
cp "$f" "$out/${tag}_rocprofv3_kernel_stats_c3.csv"
rocprofv3 --kernel-trace --output-format csv -d "$out/${tag}_trace" -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events --sample-steps 0 > /dev/null 2> "$out/${tag}_trace.err"
python tools/step_timeline.py "$(ls "$out"/${tag}_trace/*/*kernel_trace.csv | head -1)" 2 train_scalars_kernel > "$out/${tag}_step_timeline.txt"
rm -rf "$out/${tag}_trace" "$out/${tag}_stats"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_pmc_f" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --sample-steps 0 > /dev/null 2> "$out/${tag}_pmc_f.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_pmc_w" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --sample-steps 0 > /dev/null 2> "$out/${tag}_pmc_w.err"
python tools/pmc_traffic.py "$out/${tag}_pmc_f" "$out/${tag}_pmc_w" "$out/${tag}_pmc_bench_traffic.json"
rm -rf "$out/${tag}_pmc_f" "$out/${tag}_pmc_w"
echo "pmc done"
# matrix-pipe / LDS / wait / L2 counters of the same command (one set per pass: 8 SQ + 2 GRBM slots, then 4 TCC slots; no trace domain
# next to --pmc).  A pass whose counter set the device refuses is skipped, the table is built from the passes that ran.
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --sample-steps 0"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$out/${tag}_pmc_sq" -- $B > /dev/null 2> "$out/${tag}_pmc_sq.err" || echo "pmc sq pass failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$out/${tag}_pmc_l2" -- $B > /dev/null 2> "$out/${tag}_pmc_l2.err" || echo "pmc l2 pass failed"
python tools/pmc_counters.py "$out/${tag}_pmc_mfma_util.json" "$out/${tag}_pmc_sq" "$out/${tag}_pmc_l2" || true
rm -rf "$out/${tag}_pmc_sq" "$out/${tag}_pmc_l2"
echo "pmc counters done"
# the sampler alone (BASELINE config C5: 128^3, batch 1, hipGraph-captured step)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_samp_stats" -- python3 tools/sampler_profile.py --steps 100 > "$out/${tag}_sampler_profile.json" 2> "$out/${tag}_samp.err"
cp "$(ls "$out/${tag}_samp_stats"/*/*kernel_stats.csv | head -1)" "$out/${tag}_rocprofv3_kernel_stats_sampler_c5.csv"
rm -rf "$out/${tag}_samp_stats"
rocprofv3 --kernel-trace --output-format csv -d "$out/${tag}_samp_tr" -- python3 tools/sampler_profile.py --steps 40 > /dev/null 2> "$out/${tag}_samp_tr.err"
python tools/step_timeline.py "$(ls "$out"/${tag}_samp_tr/*/*kernel_trace.csv | head -1)" 5 pack_input_kernel > "$out/${tag}_sampler_step_timeline.txt"
rm -rf "$out/${tag}_samp_tr"
python tools/sampler_profile.py --steps 200 > "$out/${tag}_sampler_plain.json" 2>/dev/null
cat "$out/${tag}_sampler_plain.json"
# phase stamps of the weight-gradient kernels (diagnostic build), tap-split kernel next to the row-resident rolling kernel
if [ -f vdm4cdm_amd/libvdm4cdm_hip_timeline.so ]; then
  ( for v in 0 1; do VDM4CDM_WGRAD_ROWS=$v VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_timeline.so python tools/wgrad_phases.py --shape L0_32_32; done
    VDM4CDM_WGRAD_WGS=256 VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_timeline.so python tools/wgrad_phases.py --shape L0_32_32
    VDM4CDM_LIB=$PWD/vdm4cdm_amd/libvdm4cdm_hip_timeline.so python tools/wgrad_phases.py --shape L0_64_32 ) 2>&1 | grep -v amdgpu.ids > "$out/${tag}_wgrad_phases.txt"
fi
python bench.py --config c2 --steps 40 --warmup 5 --no-cpu-baseline --sample-steps 0 > "$out/${tag}_bench_c2.json" 2>/dev/null
# the default bench line LAST: its roofline object reads the counter tables of THIS profile (static files under profiles/)
cp "$out/${tag}_pmc_bench_traffic.json" "$out/${tag}_pmc_mfma_util.json" profiles/ 2>/dev/null || true
python bench.py > "$out/${tag}_bench_c3.json" 2> "$out/${tag}_bench_c3.err"
echo "bench done"
echo done
