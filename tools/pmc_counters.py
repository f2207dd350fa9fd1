#!/usr/bin/env python
"""Per-kernel SQ / TCC / GRBM counter table from several `rocprofv3 --pmc <set>` passes of the same command (one counter set per
pass: 8 SQ slots, 4 TCC slots, 2 GRBM slots - MI355X_MICROARCH.md "rocprofv3 PMC slots"; never combined with a trace domain).

    python tools/pmc_counters.py profiles/r04_pmc_mfma_util.json gpurun_out/r04_pmc_sq gpurun_out/r04_pmc_l2 [...]

Per kernel family (names reduced as tools/pmc_traffic.py does): the mean of every counter per launch, and
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed
                     over the 8 XCDs); SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (16 per v_mfma_f32_16x16x32_bf16)
  lds_busy_frac    = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles)        (LDS-array cycles, summed over the CUs)
  wait_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked at s_waitcnt / barrier), issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  l2_hit_rate      = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
Counter collection serialises the dispatches: these are the kernels ALONE on the chip, in the cache state the step leaves them.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import simplify  # noqa: E402


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                vals[simplify(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in vals.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        e = {"launches": max(len(v) for v in cs.values()), "counters": m}
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc > 0:
            e["kernel_cycles"] = cyc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
            if "SQ_LDS_IDX_ACTIVE" in m:
                e["lds_busy_frac"] = m["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
        if m.get("SQ_WAVE_CYCLES"):
            for name, c in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("lds_issue_stall_frac", "SQ_WAIT_INST_LDS"),
                            ("active_frac", "SQ_ACTIVE_INST_ANY")):
                if c in m:
                    e[name] = m[c] / m["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in m and m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0.0) > 0:
            e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        res[k] = e
    order = sorted(res, key=lambda k: -res[k]["counters"].get("GRBM_GUI_ACTIVE", res[k]["counters"].get("SQ_BUSY_CYCLES", 0.0)) * res[k]["launches"])
    json.dump({"note": __doc__.strip().split("\n\n")[-1], "kernels": {k: res[k] for k in order}}, open(out, "w"), indent=1)
    for k in order[:14]:
        e = res[k]
        f = lambda x: "   - " if x is None else f"{x:5.2f}"
        print(f"{k:56s} n={e['launches']:4d} mfma {f(e.get('mfma_busy_frac'))} lds {f(e.get('lds_busy_frac'))} wait {f(e.get('wait_frac'))} "
              f"stall {f(e.get('issue_stall_frac'))} l2hit {f(e.get('l2_hit_rate'))}")


if __name__ == "__main__":
    main()
