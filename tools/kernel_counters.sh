#!/bin/bash
# SQ counter sets (4 passes of 8) on the conv micro-benchmark: which issue port / queue a kernel ALONE on the chip is bound by.
#   bash tools/kernel_counters.sh <tag> [conv_microbench args]      -> gpurun_out/<tag>_kernel_counters.json
tag=${1:-kc}; shift
out=$PWD/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp
B="python3 tools/conv_microbench.py --iters 2 ${@:---only L0_32_32 --ops fwd,dgrad_gn,wgrad}"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/${tag}_kc$i" -- $B > /dev/null 2> "$out/${tag}_kc$i.err" || echo "pass $i failed"
done
python tools/pmc_counters.py "$out/${tag}_kernel_counters.json" "$out/${tag}_kc1" "$out/${tag}_kc2" "$out/${tag}_kc3" "$out/${tag}_kc4" > /dev/null
rm -rf "$out/${tag}_kc"[1-4]
python - "$out/${tag}_kernel_counters.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))["kernels"]
for k,e in d.items():
    c=e["counters"]
    if c.get("SQ_INSTS_MFMA",0)<1: continue
    w=c.get("SQ_WAVES",0) or 1
    print(k, "launches",e["launches"], "waves",int(w))
    print("   per wave: VALU %.0f SALU %.0f MFMA %.0f VMEM %.0f LDS %.0f BRANCH %.0f"%tuple(c.get(n,0)/w for n in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_MFMA","SQ_INSTS_VMEM","SQ_INSTS_LDS","SQ_INSTS_BRANCH")))
    wc=c.get("SQ_WAVE_CYCLES",1)
    print("   wave-cycles(quad) per wave %.0f | frac of wave life: active_any %.2f valu %.2f sca %.2f lds %.2f vmem %.2f | wait_any %.2f wait_inst %.2f wait_inst_lds %.2f"%(
        wc/w, *[c.get(n,0)/wc for n in ("SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_WAIT_INST_LDS")]))
    print("   inst cycles per wave: valu %.0f salu %.0f vmem %.0f | fifo full: ta_addr %.0f ta_cmd %.0f wr_data %.0f lds_data %.0f lds_cmd %.0f | ifetch %.0f"%(
        *[c.get(n,0)/w for n in ("SQ_INST_CYCLES_VALU","SQ_INST_CYCLES_SALU","SQ_INST_CYCLES_VMEM","SQ_VMEM_TA_ADDR_FIFO_FULL","SQ_VMEM_TA_CMD_FIFO_FULL","SQ_VMEM_WR_TA_DATA_FIFO_FULL","SQ_LDS_DATA_FIFO_FULL","SQ_LDS_CMD_FIFO_FULL","SQ_IFETCH")],))
    print("   mfma_busy_frac %.3f coexec/mfma_busy %.3f lds_busy %.3f kernel_cycles %.0f"%(e.get("mfma_busy_frac",0), c.get("SQ_VALU_MFMA_COEXEC_CYCLES",0)/max(c.get("SQ_VALU_MFMA_BUSY_CYCLES",1),1), e.get("lds_busy_frac",0), e.get("kernel_cycles",0)))
PY
