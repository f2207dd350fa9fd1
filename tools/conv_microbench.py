#!/usr/bin/env python
"""Micro-benchmark of the conv kernels at the shapes of the 128^3 VDM UNet (B=2): fwd / dgrad / wgrad TFLOP/s.
    python tools/conv_microbench.py [--dtype bf16] [--only L0_32_32] [--iters 20] [--ops fwd,dgrad,wgrad]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdm4cdm_amd import hip_ops as ops  # noqa: E402

SHAPES = [  # name, N, D (output), cin, cout, ks, stride, ups
    ("L0_32_32", 2, 128, 32, 32, 3, 1, 0),
    ("L0_64_32", 2, 128, 64, 32, 3, 1, 0),
    ("L0_up_64_32", 2, 128, 64, 32, 3, 1, 1),
    ("L1_64_64", 2, 64, 64, 64, 3, 1, 0),
    ("L1_128_64", 2, 64, 128, 64, 3, 1, 0),
    ("L2_128_128", 2, 32, 128, 128, 3, 1, 0),
    ("L2_256_128", 2, 32, 256, 128, 3, 1, 0),
    ("L3_128_256", 2, 16, 128, 256, 3, 1, 0),
    ("L3_256_256", 2, 16, 256, 256, 3, 1, 0),
    ("L0_down_32_32", 2, 64, 32, 32, 3, 2, 0),
    ("L1_down_64_64", 2, 32, 64, 64, 3, 2, 0),
    ("L2_down_128_128", 2, 16, 128, 128, 3, 2, 0),
    ("L0_skip_64_32", 2, 128, 64, 32, 1, 1, 0),
    ("L0_in_2_32", 2, 128, 2, 32, 3, 1, 0),
    ("L0_out_32_1", 2, 128, 32, 1, 3, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ops", default="fwd,dgrad,wgrad")
    ap.add_argument("--n", type=int, default=0, help="override the batch size of every shape (1 = the sampler's shapes)")
    ap.add_argument("--graph", action="store_true", help="time a hipGraph replay of the launches (small kernels are host-bound otherwise)")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda:0"
    for name, N, D, cin, cout, ks, stride, ups in SHAPES:
        if args.only and args.only not in name:
            continue
        if args.n:
            N = args.n
        conv = ops.Conv(cin, cout, ks, stride=stride, upsample=ups)
        w = torch.randn(ks ** 3, cout, cin, device=dev) * 0.05
        conv.pack(w, dt, need_dgrad=True)
        iD = D * 2 if stride == 2 else (D // 2 if ups else D)
        x = torch.randn(N, iD, iD, iD, ops.cpad(cin, dt), device=dev).to(dt)
        dout = torch.randn(N, D, D, D, ops.cpad(cout, dt), device=dev).to(dt)
        dw = torch.zeros(ks ** 3, cout, cin, device=dev)
        flops = 2.0 * N * D ** 3 * ks ** 3 * cin * cout
        res = []
        for op in args.ops.split(","):
            if op == "dgrad" and stride == 2:
                continue
            if op == "dgw":                                 # dgrad_gn + wgrad as one launch (csrc/conv_dgw.hip)
                if not conv.dgw_ok(dout, cin, 0):
                    continue
            if op in ("dgrad_gn", "dgw"):                   # dgrad with the GroupNorm backward folded into the epilogue (dropout mask on)
                if not conv.gn_fold_ok(cin, 0, dt) or stride != 1 or ups:
                    continue
                G = 8
                st = ops.gn_stats(x, None, G)
                gam, bet = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
                mask = torch.full((N, D * D * D, cin // ops.epl(dt)), 0xFF, dtype=torch.uint8, device=dev)
            if op in ("fwd_gnp", "gn_silu"):               # inference: GroupNorm + SiLU in the conv's prologue / as the separate pass it replaces
                if ks != 3 or stride != 1 or ups or not conv.gn_in_ok(x):
                    continue
                G = 8
                st = ops.gn_stats(x, None, G)
                gam, bet = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
            fn = {"fwd": lambda: conv.fwd(x), "fwd_gn": lambda: conv.fwd(x, gn=True), "dgrad": lambda: conv.dgrad(dout),
                  "fwd_gnp": lambda: conv.fwd(x, gn=True, gn_in=(G, st, gam, bet)), "gn_silu": lambda: ops.gn_silu_fwd(x, None, G, st, gam, bet),
                  "wgrad": lambda: conv.wgrad(x, dout, dw),
                  "dgw": lambda: conv.dgrad_gn_wgrad(dout, x, x, None, G, st, gam, bet, dw, None, keep_mask=mask, dropout_p=0.1),
                  "dgrad_gn": lambda: conv.dgrad_gn(dout, x, None, G, st, gam, bet, keep_mask=mask, dropout_p=0.1)}[op]
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            if args.graph:                                  # replay a captured graph: device time without the Python launch cost
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fn()
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(args.iters):
                        fn()
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g.replay()
                torch.cuda.synchronize()
                dtm = (time.perf_counter() - t0) / args.iters
            else:
                t0 = time.perf_counter()
                for _ in range(args.iters):
                    fn()
                torch.cuda.synchronize()
                dtm = (time.perf_counter() - t0) / args.iters
            res.append(f"{op} {dtm * 1e3:7.3f} ms {flops / dtm / 1e12:7.1f} TF/s")
        print(f"{name:16s} " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
