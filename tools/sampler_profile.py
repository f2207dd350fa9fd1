#!/usr/bin/env python
"""Reverse-diffusion sampler alone (BASELINE config C5: 128^3, batch 1, hipGraph-captured denoise step) - the program to put behind
rocprofv3 --kernel-trace --stats for the per-kernel anatomy of one sampling step.
    python tools/sampler_profile.py [--steps 100] [--cube 128] [--batch 1]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--cube", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    dev = "cuda:0"
    vdm = bench.build_model(args.cube, [32, 64, 128, 256], "bf16", dev).eval()
    b = bench.make_batch(args.cube, args.batch, 0, dev)
    kw = dict(s_conditioning=b["conditioning"], v_conditionings=b["conditioning_values"])
    vdm.draw_samples(batch_size=args.batch, n_sampling_steps=3, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vdm.draw_samples(batch_size=args.batch, n_sampling_steps=args.steps, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"steps": args.steps, "cube": args.cube, "batch": args.batch, "seconds": dt, "ms_per_step": 1e3 * dt / args.steps}))


if __name__ == "__main__":
    main()
