#!/usr/bin/env python
"""Gap anatomy of a rocprofv3 --kernel-trace csv: per stream, busy time vs span, and the idle gaps between consecutive kernels.
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/sampler_profile.py --steps 20
    python tools/trace_gaps.py out/*/*kernel_trace.csv [first_kernel_substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the second half of the run (steady state: graph replays)
rows = rows[len(rows) // 2:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
by_q = collections.defaultdict(list)
for r in rows:
    by_q[(r["Queue_Id"], r.get("Stream_Id"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
print(f"window {1e-3 * (t1 - t0):.1f} us, {len(rows)} kernels")
# union busy time over all queues
ev = sorted((s, e) for r in by_q.values() for s, e, _ in r)
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU busy (any queue) {100.0 * busy / (t1 - t0):.1f} %")
for q, ks in by_q.items():
    dur = sum(e - s for s, e, _ in ks)
    gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2][:60], ks[i + 1][2][:60]) for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"queue {q}: {len(ks)} kernels, busy {1e-3 * dur:.1f} us ({100.0 * dur / (t1 - t0):.1f} %), gaps: n={len(pos)} total {1e-3 * sum(g[0] for g in pos):.1f} us, "
          f"median {sorted(g[0] for g in pos)[len(pos) // 2] if pos else 0} ns")
    agg = collections.defaultdict(lambda: [0, 0])
    for g, a, b in pos:
        agg[(a.split("(")[0][-40:], b.split("(")[0][-40:])][0] += g
        agg[(a.split("(")[0][-40:], b.split("(")[0][-40:])][1] += 1
    for k, (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
        print(f"    {1e-3 * tot:9.1f} us in {n:4d} gaps  after {k[0]}  before {k[1]}")
