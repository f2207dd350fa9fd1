#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output.

    hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c vdm4cdm_amd/csrc/conv_fwd.hip -o /tmp/x.o 2> /tmp/res.txt
    python tools/kernel_resources.py /tmp/res.txt
"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
SCR, OCC = r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]"


def field(b, k):
    m = re.search(k + r": (\d+)", b)
    return m.group(1) if m else "?"


for b in blocks:
    name = b.split("\n")[0].strip()
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("vdm::", "").replace("void ", "")
    print("%-100s vgpr %4s agpr %3s scratch %4s occ %s" % (dem[:100], field(b, "VGPRs"), field(b, "AGPRs"), field(b, SCR), field(b, OCC)))
