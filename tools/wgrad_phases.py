#!/usr/bin/env python
"""Phase times of the 3^3 weight-gradient kernels (csrc/conv_wgrad.hip) from the stamped diagnostic build:
    make -C vdm4cdm_amd/csrc timeline
    VDM4CDM_LIB=vdm4cdm_amd/libvdm4cdm_hip_timeline.so [VDM4CDM_WGRAD_ROWS=0|1] python tools/wgrad_phases.py [--shape L0_32_32] [--stride 1|2]
Every wave sums the s_memrealtime ticks (100 MHz) it spends per phase over the tiles of its persistent workgroup; printed per tile."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdm4cdm_amd import _lib, hip_ops as ops  # noqa: E402

SHAPES = {"L0_32_32": (128, 32, 32), "L0_64_32": (128, 64, 32), "L1_64_64": (64, 64, 64), "L2_128_128": (32, 128, 128)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="L0_32_32")
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--n", type=int, default=2)
    args = ap.parse_args()
    L = _lib.lib()
    assert hasattr(L, "vdm_debug_set_stamps"), "load the diagnostic build: VDM4CDM_LIB=vdm4cdm_amd/libvdm4cdm_hip_timeline.so"
    D, cin, cout = SHAPES[args.shape]
    dev, dt = "cuda:0", torch.bfloat16
    conv = ops.Conv(cin, cout, 3, stride=args.stride)
    Do = D // args.stride
    x = torch.randn(args.n, D, D, D, cin, device=dev).to(dt)
    dout = torch.randn(args.n, Do, Do, Do, cout, device=dev).to(dt)
    dw = torch.zeros(27, cout, cin, device=dev)
    db = torch.zeros(cout, device=dev)
    fn = lambda: conv.wgrad(x, dout, dw, db)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    nwg_max = 1 << 14
    buf = torch.zeros(nwg_max * 4 * 8, dtype=torch.int64, device=dev)
    L.vdm_debug_set_stamps.argtypes = [_lib.C.c_void_p]
    L.vdm_debug_set_stamps(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    L.vdm_debug_set_stamps(None)
    s = buf.cpu().numpy().reshape(nwg_max, 4, 8).astype(np.int64)
    s = s[s[:, 0, 7] != 0]
    nwg = s.shape[0]
    tiles_total = args.n * -(-Do // (2 if args.stride == 1 else 2)) * -(-Do // (8 if args.stride == 1 else 4)) * -(-Do // 16) * (-(-cin // 32)) * (-(-cout // 32))
    per_wg = tiles_total / nwg
    us = s[:, :, :6] / 100.0                                  # ticks of 10 ns -> us
    life = (s[:, :, 7] - s[:, :, 6]) / 100.0
    names = ["wait: other waves still read the previous tile", "tile decode + LDS-DMA issue", "DMA landed (vmcnt 0 + barrier)",
             "operand reads + MFMAs", "bias column sums", "slab write (once)"]
    rows = os.environ.get("VDM4CDM_WGRAD_ROWS", "1")
    print(f"{args.shape} stride {args.stride} N={args.n} WGRAD_ROWS={rows}: {nwg} workgroups x {per_wg:.1f} tiles, launch {e0.elapsed_time(e1) * 1e3:.1f} us "
          f"(stamped build, incl. the slab reduce), workgroup life {life.mean():.1f} us")
    for k, nm in enumerate(names):
        v = us[:, :, k].reshape(-1)
        if k < 5:
            print(f"  {nm:48s} {v.mean() / per_wg:6.2f} us per tile   ({100 * v.mean() / life.mean():4.1f} % of the life)")
        else:
            print(f"  {nm:48s} {v.mean():6.2f} us            ({100 * v.mean() / life.mean():4.1f} % of the life)")


if __name__ == "__main__":
    main()
