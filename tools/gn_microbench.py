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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dropout", type=float, default=0.0)
    args = ap.parse_args()
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
