#!/usr/bin/env python
"""Per-launch view of ONE training step from a rocprofv3 --kernel-trace csv: every kernel of the step in start order with its queue,
start offset and duration - the launches of one template instantiation differ a lot between UNet levels, which the --stats
averages hide.
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events --sample-steps 0
    python tools/step_timeline.py out/*/*kernel_trace.csv [step_index_from_the_end=2] > profiles/rNN_step_timeline.txt
A third argument names the kernel a step starts with (default randn_kernel; `train_scalars_kernel` since the fused step head of round 4;
`pack_input_kernel` for the denoise step of the sampler:
    rocprofv3 --kernel-trace ... -- python3 tools/sampler_profile.py --steps 30;  step_timeline.py trace.csv 5 pack_input_kernel)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
# a training step starts with the randn_kernel pair of the forward diffusion (two launches back to back): take the first of each pair
mark = sys.argv[3] if len(sys.argv) > 3 else "randn_kernel"
starts = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"] and (i == 0 or mark not in rows[i - 1]["Kernel_Name"])]
assert len(starts) > back, f"only {len(starts)} steps in the trace"
lo, hi = starts[-back - 1], starts[-back]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
queues = {}


def short(n):
    n = re.sub(r"vdm::", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::\(anonymous namespace\)::|at::native::", "at::", n)
    return n[:86]


tot = {}
for r in step:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"q{q} {1e-3 * (s - t0):10.1f} us  {1e-3 * (e - s):8.1f} us  grid {r.get('Grid_Size', '?'):>9}  {short(r['Kernel_Name'])}")
    tot[q] = tot.get(q, 0) + (e - s)
end = max(int(r["End_Timestamp"]) for r in step)
print(f"# step span {1e-3 * (end - t0):.1f} us; kernel time per queue: " + ", ".join(f"q{q} {1e-3 * v:.1f} us" for q, v in sorted(tot.items())))
