#!/usr/bin/env python
"""Where does a conv workgroup spend its life?  Runs one conv of the 128^3 network on the stamped diagnostic build
(make -C vdm4cdm_amd/csrc timeline -> libvdm4cdm_hip_timeline.so; never used by the product) and prints, per phase, the
distribution of durations over workgroups plus per-CU occupancy of the phases.

    VDM4CDM_LIB=vdm4cdm_amd/libvdm4cdm_hip_timeline.so python tools/conv_timeline.py [--shape L0_32_32] [--op fwd|dgrad|dgrad_gn] [--n 2]

Stamps (s_memrealtime, 100 MHz, per wave): 0 kernel entry, 5 index decode done, 6 staging DMA issued, 1 first weights issued, 2 staging barrier passed (data
landed), 3 tap loop done, 4 epilogue stores retired.  Read the SHARES, not the kernel's length (the stamps fence the schedule).
"""
import argparse
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdm4cdm_amd import _lib, hip_ops as ops  # noqa: E402

SHAPES = {"L0_32_32": (128, 32, 32), "L0_64_32": (128, 64, 32), "L0_32_64": (128, 32, 64), "L1_64_64": (64, 64, 64),
          "L2_128_128": (32, 128, 128), "L3_256_256": (16, 256, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="L0_32_32")
    ap.add_argument("--op", default="fwd")
    ap.add_argument("--n", type=int, default=2)
    args = ap.parse_args()
    L = _lib.lib()
    assert hasattr(L, "vdm_debug_set_stamps"), "load the diagnostic build: VDM4CDM_LIB=vdm4cdm_amd/libvdm4cdm_hip_timeline.so"
    D, cin, cout = SHAPES[args.shape]
    dev, dt = "cuda:0", torch.bfloat16
    conv = ops.Conv(cin, cout, 3)
    conv.pack(torch.randn(27, cout, cin, device=dev) * 0.05, dt, need_dgrad=True)
    x = torch.randn(args.n, D, D, D, cin, device=dev).to(dt)
    dout = torch.randn(args.n, D, D, D, cout, device=dev).to(dt)
    if args.op == "dgrad_gn":                                # dgrad with the folded GroupNorm backward (dropout keep mask on)
        st = ops.gn_stats(x, None, 8)
        gam, bet = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
        mask = torch.full((args.n, D * D * D, cin // 8), 0xFF, dtype=torch.uint8, device=dev)
        fn = lambda: conv.dgrad_gn(dout, x, None, 8, st, gam, bet, keep_mask=mask, dropout_p=0.1)
    else:
        fn = (lambda: conv.fwd(x, gn=True)) if args.op == "fwd" else (lambda: conv.dgrad(dout))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    nwg_max = 1 << 16
    buf = torch.zeros(nwg_max * 4 * 8, dtype=torch.int64, device=dev)
    L.vdm_debug_set_stamps.argtypes = [_lib.C.c_void_p]
    L.vdm_debug_set_stamps(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    L.vdm_debug_set_stamps(None)
    s = buf.cpu().numpy().reshape(nwg_max, 4, 8)
    used = s[:, 0, 0] != 0
    s = s[used].astype(np.int64)
    nwg = s.shape[0]
    t = s[:, :, :7] * 10.0                           # ns
    t0 = t[:, :, 0].min()
    t = t - t0
    print(f"{args.shape} {args.op} N={args.n}: {nwg} workgroups, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build), "
          f"span of stamps {t.max() / 1e3:.1f} us")
    names = ["entry->DMA issued", "DMA issued->barrier (data landed)", "tap loop", "epilogue (stores retired)", "whole life",
             "  entry->setup done (index decode)", "  setup done->staging DMA issued", "  staging issued->weights issued"]
    d = [t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2], t[:, :, 4] - t[:, :, 3], t[:, :, 4] - t[:, :, 0],
         t[:, :, 5] - t[:, :, 0], t[:, :, 6] - t[:, :, 5], t[:, :, 1] - t[:, :, 6]]
    for nm, v in zip(names, d):
        v = v.reshape(-1) / 1e3
        print(f"  {nm:36s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}")
    # per CU: how many workgroups were resident over time, and the summed phase time / CU wall time
    hw = s[:, 0, 7]
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    cu = ((xcc << 12) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf)).astype(np.int64)
    per = collections.defaultdict(list)
    for i in range(nwg):
        per[int(cu[i])].append(i)
    print(f"  distinct CUs seen: {len(per)}; workgroups per CU: min {min(map(len, per.values()))} max {max(map(len, per.values()))}")
    shares = np.zeros(5)
    walls = []
    conc = []
    for c, idx in per.items():
        tt = t[idx][:, 0, :]                             # wave 0 of each workgroup
        wall = tt[:, 4].max() - tt[:, 0].min()
        walls.append(wall)
        for k in range(4):
            shares[k] += (tt[:, k + 1] - tt[:, k]).sum() / wall
        conc.append((tt[:, 4] - tt[:, 0]).sum() / wall)
    shares /= len(per)
    print(f"  per-CU wall {np.mean(walls) / 1e3:.1f} us; mean resident workgroups {np.mean(conc):.2f}")
    print("  phase time summed over a CU's workgroups / that CU's wall time (1.0 = one workgroup always in the phase):")
    for nm, v in zip(names[:4], shares[:4]):
        print(f"    {nm:36s} {v:5.2f}")
    # tap-phase overlap on a CU: fraction of the CU's wall time during which 0 / 1 / 2 workgroups are inside the tap loop
    occ = np.zeros(4)
    for c, idx in list(per.items())[:64]:
        tt = t[idx][:, 0, :]
        ev = sorted([(a, 1) for a in tt[:, 2]] + [(b, -1) for b in tt[:, 3]])
        lo, hi = tt[:, 0].min(), tt[:, 4].max()
        cur, last = 0, lo
        for tm, dlt in ev:
            occ[min(cur, 3)] += tm - last
            last = tm
            cur += dlt
        occ[min(cur, 3)] += hi - last
    occ /= occ.sum()
    print("  CU time with k workgroups inside the tap loop: " + "  ".join(f"k={k}: {occ[k] * 100:4.1f}%" for k in range(4)))


if __name__ == "__main__":
    main()
