#!/usr/bin/env python
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs of the same command) into per-kernel HBM traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01_pmc_bench_traffic.json

Units / corrections (MI355X_MICROARCH.md section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
1/2 of the bytes of wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B-per-lane stores.
Kernel names are reduced to the keys bench.py uses (tile-size template arguments dropped).
"""
import collections
import csv
import glob
import json
import re
import sys


def simplify(name):
    """rocprofv3 kernel name -> the key bench.py / hip_ops.py use for the same launch family."""
    # conv_roll_kernel<T, NC, GNB>: the rolling-z form of conv_fwd_kernel<T, T, 3, 1, 0, NC, 4, 8> - same family key
    m = re.search(r"conv_roll_kernel<vdm::(\w+), (\d), (true|false)>", name)
    if m:
        return f"conv_fwd_kernel<{m.group(1)},k3,s1,NC{m.group(2)}>" + ("+gnb" if m.group(3) == "true" else "")
    # conv_fwd_kernel<T, TO, KS, STRIDE, UPS, NC, TZ, TY, SPLIT, GNB, GNP, NW>
    m = re.search(r"conv_fwd_kernel<vdm::(\w+), (?:vdm::)?(\w+), (\d), (\d), (\d), (\d), \d+, \d+(?:, (true|false))?(?:, (true|false))?(?:, (?:true|false))?(?:, \d+)?>", name)
    if m:
        return (f"conv_fwd_kernel<{m.group(1)},k{m.group(3)},s{m.group(4)},NC{m.group(6)}{',split' if m.group(7) == 'true' else ''}>"
                + ("+gnb" if m.group(8) == "true" else ""))
    m = re.search(r"conv_kpack_kernel<vdm::(\w+), (\d)(?:, (true|false))?>", name)
    if m:
        return "conv_kpack_kernel" + ("+gnb" if m.group(3) == "true" else "")
    m = re.search(r"conv_cls_kernel<vdm::(\w+), (\d), (\d)>", name)
    if m:
        return f"conv_cls_kernel<{m.group(1)},NC{m.group(2)},{'B' if m.group(3) == '1' else 'F'}>"
    # conv_wgrad_rows_kernel<T, TZ, TY, ROLL>: the row-resident / rolling form of conv_wgrad_kernel<T, 3, 1, 0, ...> - same family key
    m = re.search(r"conv_wgrad_rows_kernel<vdm::(\w+), \d+, \d+, (?:true|false)>", name)
    if m:
        return f"conv_wgrad_kernel<{m.group(1)},k3,s1,u0>"
    # conv_wgrad_kernel<T, KS, STRIDE, UPS, TZ, TY, NTA, NTB>
    m = re.search(r"conv_wgrad_kernel<vdm::(\w+), (\d), (\d), (\d), \d+, \d+(?:, (\d), (\d))?>", name)
    if m:
        tiles = f",ta{m.group(5)}tb{m.group(6)}" if m.group(5) and (m.group(5), m.group(6)) != ("2", "2") else ""
        return f"conv_wgrad_kernel<{m.group(1)},k{m.group(2)},s{m.group(3)},u{m.group(4)}{tiles}>"
    m = re.search(r"(?:vdm::)?(\w+_kernel)\b", name)          # gn_*_kernel<T>, pack_many_kernel<T>, wgrad_reduce_kernel, ...
    if m:
        return m.group(1)
    return name[:60]


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[simplify(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        res[k] = {"launches": max(len(f), len(w)),
                  "read_bytes_per_launch": 2.0 * 1024.0 * sum(f) / max(len(f), 1),
                  "write_bytes_per_launch": 1024.0 * sum(w) / max(len(w), 1)}
        res[k]["hbm_bytes_per_launch"] = res[k]["read_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    json.dump({"units": "bytes; read = 2 * FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB", "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k:50s} n={v['launches']:4d} read {v['read_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
