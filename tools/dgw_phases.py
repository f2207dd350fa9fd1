#!/usr/bin/env python
"""Phase times of the fused dgrad + wgrad kernel (csrc/conv_dgw.hip) from the diagnostic build:
    make -C vdm4cdm_amd/csrc variant NAME=dgwst DEFS=-DVDM_DGW_STAMPS
    VDM4CDM_LIB=vdm4cdm_amd/libvdm4cdm_hip_dgwst.so python tools/dgw_phases.py
Per workgroup and role the s_memrealtime (100 MHz) time of each phase, summed over the steps of its column; printed per step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdm4cdm_amd import hip_ops as ops  # noqa: E402

dev, dt = "cuda:0", torch.bfloat16
N, D = 2, 128
conv = ops.Conv(32, 32, 3)
conv.pack(torch.randn(27, 32, 32, device=dev) * 0.05, dt, need_dgrad=True)
x = torch.randn(N, D, D, D, 32, device=dev).to(dt)
dout = torch.randn(N, D, D, D, 32, device=dev).to(dt)
st = ops.gn_stats(x, None, 8)
gam, bet = torch.ones(32, device=dev), torch.zeros(32, device=dev)
mask = torch.full((N, D ** 3, 4), 0xFF, dtype=torch.uint8, device=dev)
dw = torch.zeros(27, 32, 32, device=dev)
for _ in range(3):
    conv.dgrad_gn_wgrad(dout, x, x, None, 8, st, gam, bet, dw, None, keep_mask=mask, dropout_p=0.1)
torch.cuda.synchronize()
ws = [v for k, v in ops.Conv._ws.items() if k[-1] == "dgw"][0].view(torch.float32)
P = 256
t = ws[P * (27 * 32 * 32 + 32):P * (27 * 32 * 32 + 32) + P * 8].view(P, 8).cpu().double()
steps = D // 2
us = t / steps / 100.0
names0 = ["taps", "wait B2", "epilogue (incl. B3)", "wait B1"]
names1 = ["stage + both k-loop halves (incl. B2)", "wait B3", "DMA wait", "wait B1"]
print(f"per step (us), mean over {P} workgroups, {steps} steps each:")
print("  input-gradient waves : " + " | ".join(f"{n} {us[:, k].mean():.2f}" for k, n in enumerate(names0)) + f" | total {us[:, :4].sum(1).mean():.2f}")
print("  weight-gradient waves: " + " | ".join(f"{n} {us[:, 4 + k].mean():.2f}" for k, n in enumerate(names1)) + f" | total {us[:, 4:].sum(1).mean():.2f}")
