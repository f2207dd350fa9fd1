#!/usr/bin/env python
"""Phase anatomy of conv_duo_kernel (stamped diagnostic build): per phase, how long the loaders need to issue / land the next halo and
how long a team spends in its tap loop and in its epilogue.
    VDM4CDM_LIB=vdm4cdm_amd/libvdm4cdm_hip_timeline.so python tools/duo_timeline.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdm4cdm_amd import _lib, hip_ops as ops  # noqa: E402

L = _lib.lib()
dev, dt = "cuda:0", torch.bfloat16
conv = ops.Conv(32, 32, 3)
conv.pack(torch.randn(27, 32, 32, device=dev) * 0.05, dt, need_dgrad=True)
x = torch.randn(2, 128, 128, 128, 32, device=dev).to(dt)
for _ in range(5):
    conv.fwd(x, gn=True)
torch.cuda.synchronize()
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
L.vdm_debug_set_stamps.argtypes = [_lib.C.c_void_p]
L.vdm_debug_set_stamps(buf.data_ptr())
conv.fwd(x, gn=True)
torch.cuda.synchronize()
L.vdm_debug_set_stamps(None)
s = buf.cpu().numpy().reshape(256, 8, 16).astype(np.int64) * 10.0 / 1e3      # us
ok = s[:, :, 0] > 0
ld, tm = s[:, :, 0:4], s[:, :, 4:8]
f = lambda v: f"mean {v[ok].mean():6.2f}  p10 {np.percentile(v[ok], 10):6.2f}  p90 {np.percentile(v[ok], 90):6.2f} us"
print("loaders: issue DMA      ", f(ld[:, :, 1] - ld[:, :, 0]))
print("loaders: wait landed    ", f(ld[:, :, 2] - ld[:, :, 1]))
print("loaders: wait at barrier", f(ld[:, :, 3] - ld[:, :, 2]))
print("team   : taps           ", f(tm[:, :, 1] - tm[:, :, 0]))
print("team   : wait at barrier", f(tm[:, :, 2] - tm[:, :, 1]))
print("team   : epilogue       ", f(tm[:, :, 3] - tm[:, :, 2]))
