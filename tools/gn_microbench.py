#!/usr/bin/env python
"""Micro-benchmark of the GroupNorm + SiLU kernels at the shapes of the 128^3 VDM UNet (B=2): ms and effective GB/s
(algorithmic bytes: every tensor the op must read or write once).
    python tools/gn_microbench.py [--dtype bf16] [--iters 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdm4cdm_amd import hip_ops as ops  # noqa: E402

SHAPES = [("L0_32", 2, 128, 32, 0), ("L0_32+32", 2, 128, 32, 32), ("L1_64", 2, 64, 64, 0), ("L2_128", 2, 32, 128, 0), ("L3_256", 2, 16, 256, 0)]


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


SKIP_SHAPES = [("L0up", 2, 128, 32, 32, 32), ("L0up_b1", 1, 128, 32, 32, 32), ("L1down", 2, 64, 32, 0, 64), ("L1up", 2, 64, 64, 64, 64),
               ("L2down", 2, 32, 64, 0, 128), ("c256_L0up", 1, 256, 16, 16, 16)]


def skip_bench(args):
    """ms per call, alone on the device: (gn_silu_fwd + the 1x1x1 convs) vs gn_silu_skip_fwd; (1x1x1 dgrad + wgrads + finalize/apply) vs
    the fused apply."""
    dt, dev, G = torch.bfloat16, "cuda:0", 8
    for name, N, D, c1, c2, cout in SKIP_SHAPES:
        x1 = torch.randn(N, D, D, D, c1, device=dev).to(dt)
        x2 = torch.randn(N, D, D, D, c2, device=dev).to(dt) if c2 else None
        C = c1 + c2
        gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
        w1, w2 = torch.randn(cout, c1, device=dev) / C ** 0.5, (torch.randn(cout, c2, device=dev) / C ** 0.5 if c2 else None)
        bias = torch.randn(cout, device=dev)
        k1, k2 = ops.Conv(c1, cout, 1), (ops.Conv(c2, cout, 1) if c2 else None)
        k1.pack(w1.view(1, cout, c1), dt, True)
        if k2:
            k2.pack(w2.view(1, cout, c2), dt, True)
        st = ops.gn_stats(x1, x2, G)
        dyh = torch.randn(N, D, D, D, C, device=dev).to(dt)
        dout = torch.randn(N, D, D, D, cout, device=dev).to(dt)
        part = ops.channel_dot_sums(dyh, x1, x2)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dw1, dw2 = torch.zeros(cout, c1, device=dev), (torch.zeros(cout, c2, device=dev) if c2 else None)
        dx1, dx2 = torch.empty_like(x1), (torch.empty_like(x2) if c2 else None)
        dyh.gnb_partials = part

        def sep_f():
            ops.gn_silu_fwd(x1, x2, G, st, gamma, beta)
            s = k1.fwd(x1, bias)
            if k2:
                k2.fwd(x2, None, None, s)

        def sep_b():
            a1 = k1.dgrad(dout)
            a2 = k2.dgrad(dout) if k2 else None
            k1.wgrad(x1, dout, dw1.view(1, cout, c1))
            if k2:
                k2.wgrad(x2, dout, dw2.view(1, cout, c2))
            ops.gn_bwd_fused(x1, x2, G, st, gamma, dyh, dg, db, add1=a1, add2=a2, dx1=dx1, dx2=dx2)

        okf, okb = ops.gn_skip_supported(c1, c2, cout, dt)
        t = [timed(sep_f, args.iters), timed(lambda: ops.gn_silu_skip_fwd(x1, x2, G, st, gamma, beta, w1, w2, bias), args.iters) if okf else float("nan"),
             timed(sep_b, args.iters),
             timed(lambda: ops.gn_bwd_fused(x1, x2, G, st, gamma, dyh, dg, db, dx1=dx1, dx2=dx2, skip=(dout, w1, w2, dw1, dw2)), args.iters) if okb else float("nan")]
        fb = (2 * x1.numel() * C / c1 + dout.numel()) * 2            # x read, y + s written
        bb = (3 * x1.numel() * C / c1 + dout.numel()) * 2            # dyh, x, dout read, dx written
        print(f"{name:10s} fwd separate {t[0] * 1e3:6.3f} ms  fused {t[1] * 1e3:6.3f} ms ({fb / t[1] / 1e9:5.0f} GB/s) | "
              f"bwd separate {t[2] * 1e3:6.3f} ms  fused {t[3] * 1e3:6.3f} ms ({bb / t[3] / 1e9:5.0f} GB/s)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dropout", type=float, default=0.0)
    ap.add_argument("--skip", action="store_true", help="norm1 + 1x1x1 skip conv: the fused passes (csrc/gn_skip.hip) against the separate kernels")
    args = ap.parse_args()
    if args.skip:
        return skip_bench(args)
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda:0"
    G = 8
    for name, N, D, c1, c2 in SHAPES:
        x1 = torch.randn(N, D, D, D, c1, device=dev).to(dt)
        x2 = torch.randn(N, D, D, D, c2, device=dev).to(dt) if c2 else None
        c = c1 + c2
        gamma = torch.randn(c, device=dev)
        beta = torch.randn(c, device=dev)
        dy = torch.randn(N, D, D, D, c, device=dev).to(dt)
        dg = torch.zeros(c, device=dev)
        db = torch.zeros(c, device=dev)
        nbytes = x1.numel() * x1.element_size() * (c / c1)
        st = ops.gn_stats(x1, x2, G)
        t_st = timed(lambda: ops.gn_stats(x1, x2, G), args.iters)
        t_f = timed(lambda: ops.gn_silu_fwd(x1, x2, G, st, gamma, beta, args.dropout, 1), args.iters)
        t_b = timed(lambda: ops.gn_silu_bwd(x1, x2, G, st, gamma, beta, dy, dg, db, dropout_p=args.dropout, seed=1), args.iters)
        print(f"{name:10s} stats {t_st * 1e3:6.3f} ms {nbytes / t_st / 1e9:6.0f} GB/s | fwd {t_f * 1e3:6.3f} ms {2 * nbytes / t_f / 1e9:6.0f} GB/s"
              f" | bwd(reduce+apply) {t_b * 1e3:6.3f} ms {5 * nbytes / t_b / 1e9:6.0f} GB/s (x,dy read twice + dx written)", flush=True)


if __name__ == "__main__":
    main()
