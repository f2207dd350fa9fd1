#!/usr/bin/env python
"""bench.py - headline benchmark of the vdm4cdm VDM denoising path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full training step of the 128^3 conditional VDM (BASELINE config C3,
trainVDM3D128_c_c thick_lowbatch: chs [32,64,128,256], batch 2 per GPU, bf16 storage / fp32 accumulate):
forward diffusion + UNet forward + ELBO + UNet backward + gradient all-reduce (N>1) + global-norm clip + AdamW,
on synthetic lognormal density cubes already resident in HBM.  Metric: 3D voxels/s = N * B * D^3 / step time.

Launch: under torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) every rank runs main(); a plain `python bench.py --gpus N`
with N > 1 starts the N rank processes itself (fresh children, started before the parent touches the GPU) and forwards rank 0's line.
A run that was asked for N GPUs never prints a line for fewer.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel (3^3 implicit-GEMM conv on bf16 MFMA): algorithmic FLOPs of its launches in the
                 timed region / their summed HIP-event durations, against the 2.5 PFLOP/s dense bf16 MFMA peak;
                 plus the whole-step HBM fraction (algorithmic bytes of SURVEY.md section 8d / step time / 8 TB/s).
  cpu_baseline - the oracle (plain torch CPU fp32, kind "port") timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (D, batch per GPU, chs, precision)
    "c2": (64, 2, [32, 64, 128, 256], "fp32"),
    "c3": (128, 2, [32, 64, 128, 256], "bf16"),
    "c4": (192, 2, [32, 64, 128, 256], "bf16"),
    "c224": (224, 2, [16, 32, 64, 128], "bf16"),
    "c256": (256, 2, [16, 32, 64, 128], "bf16"),          # circular padding (configs.yaml VDM_Mstar_Mcdm_c_c_256)
    "tiny": (32, 2, [16, 32, 64], "bf16"),
}
# algorithmic work per voxel per forward at chs [32,64,128,256]: 720 kFLOP, 1.53 kB (bf16) / 3.05 kB (fp32);
# fwd+bwd = 3x forward.
FLOP_PER_VOXEL_FWD = 720.0e3
BYTES_PER_VOXEL_FWD = {"bf16": 1.53e3, "fp32": 3.05e3}
HBM_PEAK = 8.0e12
MFMA_PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}
# fp32 STORAGE runs on the bf16 matrix pipe by default (three bf16 MFMAs per product, csrc/common.h VDM_FP32_SPLIT): its ceiling is a third
# of the bf16 peak; VDM4CDM_FP32_EXACT=1 selects the library built on v_mfma_f32_16x16x4_f32 and the fp32 MFMA peak above
from vdm4cdm_amd import _lib as _vlib
if not _vlib.FP32_EXACT:
    MFMA_PEAK["fp32"] = 2.5e15 / 3.0
DTYPE_LABEL = {"bf16": "bf16", "fp32": "fp32 (exact fp32 MFMA)" if _vlib.FP32_EXACT else "fp32 storage, bf16x3 MFMA"}


def build_model(D, chs, precision, device, seed=42):
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import LightVDM
    torch.manual_seed(seed)                                     # [REF trainVDM3D128...py:54] seed_everything(42)
    net = CUNet(shape=(1, D, D, D), chs=chs, s_conditioning_channels=1, v_conditioning_dims=[6], t_conditioning=True,
                norm_groups=8, mid_attn=False, dropout_prob=0.1, conv_padding_mode="circular" if D == 256 else "zeros",
                n_attention_heads=4, backend="hip", precision=precision)
    g = torch.Generator().manual_seed(seed)
    net.reset_parameters(generator=g, zero_init_std=0.02)       # zero-init convs -> N(0, 0.02): non-trivial gradients
    return LightVDM(score_model=net, draw_figure=None, gamma_max=13.3, learning_rate=3.0e-4).to(device)


def make_batch(D, B, rank, device):
    from vdm4cdm_amd.data import SyntheticAstroDataModule
    dm = SyntheticAstroDataModule(cropsize=D, batch_size=B, seed=1000 + rank)
    b = dm._make_batch(1000 + rank, B)
    return {"x": b["x"].to(device), "conditioning": b["conditioning"].to(device),
            "conditioning_values": [b["conditioning_values"][0].to(device)]}


def _oracle_fwd_bwd_seconds(chs, D, B, iters):
    """Oracle (torch CPU fp32) forward + backward of the UNet on a (B,1,D,D,D) batch: 1 warm-up + `iters` timed, list of seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_cfg, oracle_params
    from oracle import unet_oracle
    from vdm4cdm_amd.networks import CUNet
    net = CUNet(shape=(1, D, D, D), chs=chs, s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=8, backend="torch")
    net.reset_parameters(generator=torch.Generator().manual_seed(42), zero_init_std=0.02)
    P = {k: v.clone().requires_grad_(True) for k, v in oracle_params(net).items()}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 1, D, D, D, generator=g)
    s = torch.randn(B, 1, D, D, D, generator=g)
    t = torch.rand(B, generator=g)
    v = [torch.rand(B, 6, generator=g)]
    times = []
    for it in range(1 + iters):
        t0 = time.time()
        y = unet_oracle.cunet_forward(P, oracle_cfg(net), x, t, s, v)
        (y * x).sum().backward()
        times.append(time.time() - t0)
        for p in P.values():
            p.grad = None
    return times[1:], times[0]


def cpu_baseline(chs):
    """the oracle (this repo's fp32 torch-CPU restatement of the path; kind "port") on the host cores -
    C2 (64^3, batch 2) and 128^3 batch 1, UNet fwd+bwd, 1 warm-up + 3 timed iterations each, median.  `value` is the 128^3 number
    (the metric's cube size); the C2 number rides along."""
    cores = min(os.cpu_count() or 1, 64)          # oneDNN does not scale these convs past a few dozen threads
    torch.set_num_threads(cores)
    t64, w64 = _oracle_fwd_bwd_seconds(chs, 64, 2, 3)
    t128, w128 = _oracle_fwd_bwd_seconds(chs, 128, 1, 3)
    med = lambda ts: sorted(ts)[len(ts) // 2]
    return {"value": 128 ** 3 / med(t128), "unit": "voxels/s", "cores": cores, "kind": "port",
            "c2_64cube_batch2_voxels_per_s": 2 * 64 ** 3 / med(t64),
            "sample": f"oracle (torch CPU fp32, {cores} threads) UNet fwd+bwd, 1 warm-up + 3 timed, median: 128^3 batch 1 "
                      f"{med(t128):.2f} s (warm-up {w128:.1f} s); C2 64^3 batch 2 {med(t64):.2f} s (warm-up {w64:.1f} s)"}


def sample_leg(args, vdm, batch, D, device, rank, world):
    """One `--sample-steps`-step chain of one 128^3 cube PER RANK (hipGraph-captured denoise step, in-kernel Philox noise).  Returns the
    "sample" object on rank 0 (None elsewhere; every rank gets {"error": ...} if any chain failed).  N = 1: exactly the round-3 leg
    (seed 1234, z_1 from the global generator); N > 1: chain id = rank, seed = entry.chain_seed(rank) for the chain AND its z_1."""
    import torch.distributed as dist
    from vdm4cdm_amd.entry import chain_seed
    res, err = {}, None
    try:
        vdm.eval()
        s = batch["conditioning"][:1]
        v = [batch["conditioning_values"][0][:1]]
        vdm.draw_samples(batch_size=1, n_sampling_steps=3, s_conditioning=s, v_conditionings=v)
        seed = 1234 if world == 1 else chain_seed(rank)
        if world == 1:
            z1 = torch.randn(1, 1, D, D, D, device=device)          # z_1 ~ N(0, I), resident before the clock starts
        else:
            z1 = torch.randn(1, 1, D, D, D, device=device, generator=torch.Generator(device=device).manual_seed(seed))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        z = vdm.draw_samples(batch_size=1, n_sampling_steps=args.sample_steps, z=z1, s_conditioning=s, v_conditionings=v, seed=seed)
        torch.cuda.synchronize()
        secs = time.perf_counter() - t1
        res = {"seconds": secs, "finite": bool(torch.isfinite(z).all()), "std": float(z.std()), "mean": float(z.mean()), "seed": seed}
        if not res["finite"]:
            raise RuntimeError(f"the {args.sample_steps}-step sample of rank {rank} is not finite")
    except Exception as e:                                  # never lose the training line over the secondary measurement
        err = repr(e)
    per = [dict(res, error=err) if err else res]
    if world > 1:
        per = [None] * world
        dist.all_gather_object(per, dict(res, error=err) if err else res)
    if rank != 0:
        return {"error": "a chain failed"} if any("error" in p for p in per) else None
    out = {"steps": args.sample_steps, "cube": D, "batch": 1, "chains": world}
    bad = [(r, p["error"]) for r, p in enumerate(per) if "error" in p]
    if bad:
        out["error"] = f"(rank, error): {bad}"
        return out
    smax = max(p["seconds"] for p in per)
    out.update(seconds=smax, steps_per_s=args.sample_steps / smax, finite=all(p["finite"] for p in per), std=per[0]["std"], mean=per[0]["mean"],
               seconds_max=smax, seconds_per_rank=[p["seconds"] for p in per], seeds=[p["seed"] for p in per],
               std_per_rank=[p["std"] for p in per], aggregate_steps_per_s=world * args.sample_steps / smax,
               note="one chain per rank (independent cubes, no collective), started together; hipGraph-captured denoise step, in-kernel "
                    "Philox noise; seconds = the slowest chain, aggregate = chains * steps / that")
    if args.config == "c256":      # context only: the one rate the reference publishes, other hardware
        out["reference_published"] = {
            "value": 2.50, "unit": "denoise steps/s", "steps": 250, "cube": 256, "chs": [16, 32, 64, 128],
            "hardware": "one NVIDIA GPU of the 80 GB class (model not stated)",
            "source": "/root/reference ICML_figures.ipynb cell 103 (tqdm rate), configs.yaml:127-143, generate_3D.py:61"}
    return out


def self_spawn(n, argv):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as fresh child processes of this one - which has not
    made a single HIP call (and never execs after one) - with the torchrun environment on 127.0.0.1, forward rank 0's stdout (the
    JSON line) and return the worst exit code.  VDM4CDM_SHARE_GPU=1 (+ VDM4CDM_DIST_BACKEND=gloo) rehearses the same path on one GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: rank processes failed (rank, exit code): {bad}", file=sys.stderr)
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (pure wall-clock run)")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="HIP events around EVERY kernel launch (full per-kernel table; ~600 event pairs per step cost ~6 %% of the step)."
                         " Default: only the 3^3 conv fwd/dgrad launches, the candidates for the dominant kernel (~50 per step)")
    ap.add_argument("--per-step", action="store_true", help="diagnostic: synchronise after every timed step and print its duration to stderr")
    ap.add_argument("--graph", action="store_true",
                    help="after the eager timed region (N = 1), also time the same step as ONE hipGraph replay (trainer.GraphedTrainStep), reported "
                         "under \"graph_step\": faster where the host is the limiter (32^3: 1.9 vs 5.2 ms), slower at the BASELINE sizes (128^3: 15.0 vs 14.7 ms)")
    ap.add_argument("--sample-only", action="store_true",
                    help="skip the training steps: only the --sample-steps reverse-diffusion sample is timed (e.g. --config c256 "
                         "--sample-steps 250, the configuration of the reference's one published rate)")
    ap.add_argument("--sample-steps", type=int, default=1000,
                    help="after the training steps, rank 0 also times an n-step reverse-diffusion sample of one cube (second half of the "
                         "BASELINE metric: 1000-step sample wall-clock; reported under \"sample\", never part of \"value\"); 0 = skip")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # plain `python bench.py --gpus N`: become the launcher
        sys.exit(self_spawn(args.gpus, sys.argv[1:]))

    import torch.distributed as dist
    from vdm4cdm_amd import hip_ops
    from vdm4cdm_amd.trainer import allreduce_mean_, clip_grad_norm_flat_, init_distributed

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    rank, local_rank, world = init_distributed("cuda")
    if world != args.gpus:                                        # never report a line for fewer GPUs than were asked for
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to measure a different job than requested")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but this node shows {torch.cuda.device_count()} "
                         "(VDM4CDM_SHARE_GPU=1 VDM4CDM_DIST_BACKEND=gloo rehearses N ranks on one GPU)")
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    D, B, chs, precision = CONFIGS[args.config]
    vdm = build_model(D, chs, precision, device)
    net = vdm.model.score_model
    params = [p for p in vdm.parameters() if p.requires_grad]
    if world > 1:
        for p in params:
            dist.broadcast(p.data, src=0)
    opt = vdm.configure_optimizers(capturable=(world == 1 and args.graph))
    batch = make_batch(D, B, rank, device)
    vdm.train()
    net.enable_ddp(world)            # N > 1: the backward all-reduces the flat gradient in 4 buckets, overlapped with its own tail

    def step():
        loss = vdm.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        synced, net.grad_synced = net.grad_synced, False
        for p in params:
            if p.grad is not None and not (synced and p is net.flat):
                allreduce_mean_(p.grad, world)
        clip_grad_norm_flat_(params, 0.5, use_hip=True, want_norm=False)
        opt.step()
        return loss

    loss, prof, elapsed, ms_per_step, value, per_rank_s = None, None, 0.0, None, None, []
    if not args.sample_only:
        for _ in range(2):      # setup, not warm-up: sizes the caching allocator's pools (main + side streams) and packs the weights once
            step()
        for _ in range(args.warmup):
            step()
        if not args.no_kernel_events:
            prof = hip_ops.PROFILER = hip_ops.KernelProfiler(None if args.all_kernel_events else {"conv3", "wgrad"})
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per_step = []
        for i in range(args.steps):
            if args.per_step:
                torch.cuda.synchronize()
                per_step.append(time.perf_counter())
            # per-launch events perturb the step (~3 % even for the conv launches alone): sample every 5th timed step unless asked for all
            hip_ops.PROFILER = prof if (args.all_kernel_events or i % 5 == 0) else None
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if args.per_step and rank == 0:
            per_step.append(time.perf_counter())
            print("per-step ms:", " ".join(f"{1e3 * (b - a):.2f}" for a, b in zip(per_step, per_step[1:])), file=sys.stderr, flush=True)
        hip_ops.PROFILER = None
        loss = float(loss.detach())          # (drop the autograd graph of the last eager step: a live AccumulateGrad node bound to the
                                             # default stream would be run inside the stream capture of the graphed step below)
        el = torch.tensor([elapsed], device=device, dtype=torch.float64)
        per_rank_s = [elapsed]
        if world > 1:
            gathered = [torch.zeros_like(el) for _ in range(world)]
            dist.all_gather(gathered, el)                    # every rank's own clock around the same K steps: a straggler shows
            per_rank_s = [g.item() for g in gathered]
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = el.item()
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * D ** 3 * args.steps / elapsed

    # Second half of the BASELINE metric: the n-step reverse-diffusion sample.  EVERY rank runs its own chain (independent cubes shard over
    # the GPUs with no collective - /root/reference/generate_3D.py:43-68, the reference's 6-process fan-out), seeded by its global chain
    # id as entry.generate_3d does; the ranks start together (barrier) and rank 0 reports the slowest chain and the aggregate rate.
    sample_res = None
    if args.sample_steps:
        sample_res = sample_leg(args, vdm, batch, D, device, rank, world)

    if rank == 0:
        scale = 1.0 if chs == [32, 64, 128, 256] else None
        out = {
            "metric": f"3D voxels/sec UNet fwd+bwd @{D}^3 (VDM training step)", "value": value, "unit": "voxels/s",
            "n_gpus": world, "steps": 0 if args.sample_only else args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL[precision], "data": "synthetic",
            "config": {"workload": f"{args.config}: {D}^3 conditional VDM (trainVDM3D128_c_c thick_lowbatch), chs {chs}, "
                                   f"batch {B}/GPU, full training step (diffuse + UNet fwd + ELBO + UNet bwd + clip + AdamW), dropout 0.1",
                       "global_batch": world * B, "cube": D, "parallelism": f"dp{world}"},
            "loss": loss,
        }
        if per_rank_s and not args.sample_only:
            out["ms_per_step_per_rank"] = {"min": 1e3 * min(per_rank_s) / args.steps, "max": 1e3 * max(per_rank_s) / args.steps,
                                           "all": [1e3 * t / args.steps for t in per_rank_s]}
        roof = {}
        if prof is not None:
            agg = prof.summary()
            ev_steps = args.steps if args.all_kernel_events else len(range(0, args.steps, 5))
            total_ms = elapsed * 1e3 * ev_steps / args.steps

            def roofline_of(key):
                d = agg[key]
                ach = d["flops"] / (d["ms"] * 1e-3)
                return {"kernel": key, "achieved": ach / 1e12, "peak": MFMA_PEAK[precision] / 1e12, "unit": "TFLOP/s",
                        "frac": ach / MFMA_PEAK[precision],
                        "algorithmic_flop_per_launch": d["flops"] / d["launches"], "executed_flop_per_launch": d["exec_flops"] / d["launches"],
                        "algorithmic_bytes_per_launch": d["bytes"] / d["launches"], "launches_per_step": d["launches"] / ev_steps,
                        "avg_launch_ms": d["ms"] / d["launches"], "ms_per_step": d["ms"] / ev_steps,
                        "stream": "side (overlaps the main stream)" if d["side_launches"] == d["launches"] else "main (critical path)"}

            # candidates: every MFMA conv family with FLOPs (3^3 forward / dgrad incl. the folded-GroupNorm and per-parity-class
            # variants, and the weight gradients); "dominant" = the largest by summed duration on the critical path (main stream),
            # "largest_by_total_time" = over all streams (the weight gradients run on a side stream and overlap the main stream)
            cand = {k: v for k, v in agg.items() if v["flops"] > 0 and v["ms"] > 0}
            main_c = {k: v for k, v in cand.items() if v["side_launches"] < v["launches"]}
            if main_c:
                dom_key = max(main_c, key=lambda k: main_c[k]["ms"])
                # HBM bytes per launch: NOT measured in this run - a static profile from separate rocprofv3 --pmc passes of this same
                # command (tools/round_profile.sh -> tools/pmc_traffic.py, FETCH_SIZE doubled per the gfx950 correction)
                traffic, traffic_src, step_traffic, mfma_busy, mfma_src = None, None, None, None, None
                for tag in ("r04", "r03", "r02"):
                    tpath = os.path.join(ROOT, "profiles", f"{tag}_pmc_bench_traffic.json")
                    if args.config == "c3" and os.path.exists(tpath):
                        tk = json.load(open(tpath))["kernels"]
                        t = tk.get(dom_key)
                        if t:
                            traffic = t["hbm_bytes_per_launch"]
                            traffic_src = f"static profile profiles/{tag}_pmc_bench_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes)"
                            # whole-step counter traffic: every kernel of the profiled run (launch counts / hbm bytes), per step
                            nst = t["launches"] / max(1, round(agg[dom_key]["launches"] / ev_steps))
                            step_traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in tk.values()) / max(nst, 1) / 1e9
                            break
                for tag in ("r04",):
                    mpath = os.path.join(ROOT, "profiles", f"{tag}_pmc_mfma_util.json")
                    if args.config == "c3" and os.path.exists(mpath):
                        mk = json.load(open(mpath))["kernels"].get(dom_key)
                        if mk and "mfma_busy_frac" in mk:
                            mfma_busy = mk["mfma_busy_frac"]
                            mfma_src = (f"static profile profiles/{tag}_pmc_mfma_util.json: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), "
                                        "kernel alone on the chip (counter collection serialises dispatches)")
                roof = {"bound": "mfma", **roofline_of(dom_key), "traffic": traffic, "traffic_source": traffic_src,
                        "mfma_busy_frac": mfma_busy, "mfma_busy_source": mfma_src, "step_traffic_GB": step_traffic,
                        "selection": "largest summed duration among the MFMA conv families launched on the main stream",
                        "share_of_step_time": main_c[dom_key]["ms"] / total_ms,
                        "events": "all launches, every timed step" if args.all_kernel_events
                        else "HIP events (on the launch stream) around the 3^3 conv fwd / dgrad / wgrad launches of every 5th timed step"}
                big_key = max(cand, key=lambda k: cand[k]["ms"])
                roof["largest_by_total_time"] = {"bound": "mfma", **roofline_of(big_key), "share_of_step_time": cand[big_key]["ms"] / total_ms}
            out["kernels"] = {k: {"launches": v["launches"], "ms_per_step": v["ms"] / ev_steps, "side_stream": v["side_launches"] == v["launches"],
                                  "TFLOP/s": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] else None,
                                  "executed_TFLOP/s": (v["exec_flops"] / (v["ms"] * 1e-3) / 1e12) if v["exec_flops"] else None,
                                  "GB/s": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["bytes"] else None}
                              for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        if scale is not None and ms_per_step:
            alg_bytes = 3.0 * BYTES_PER_VOXEL_FWD[precision] * B * D ** 3          # per GPU per step (fwd+bwd)
            alg_flops = 3.0 * FLOP_PER_VOXEL_FWD * B * D ** 3
            roof["step_hbm_frac"] = alg_bytes / (ms_per_step * 1e-3) / HBM_PEAK
            # algorithmic = 27 taps everywhere; the per-parity-class kernels execute 8 merged taps (up-conv fwd / dgrad /
            # wgrad) or 1/8 of the dilated formulation (stride-2 dgrad): this fraction is work rate, not MFMA utilisation
            roof["step_algorithmic_flop_frac_of_mfma_peak"] = alg_flops / (ms_per_step * 1e-3) / MFMA_PEAK[precision]
            roof["step_mfma_frac"] = roof["step_algorithmic_flop_frac_of_mfma_peak"]
            roof["step_algorithmic_GB"] = alg_bytes / 1e9
            roof["step_algorithmic_TFLOP"] = alg_flops / 1e12
        out["roofline"] = roof
        if world == 1 and args.graph and not args.sample_only:
            # the same step as ONE hipGraph replay (trainer.GraphedTrainStep: what Trainer.fit runs at N = 1): K timed replays after W warm-up
            # replays.  `value` above stays the eager run - its kernels carry the per-launch events of the roofline object.
            try:
                from vdm4cdm_amd.trainer import GraphedTrainStep
                vdm.train()                                  # (the sampling leg above left the model in eval mode)
                gs = GraphedTrainStep(vdm, opt, params, 0.5, batch)
                for _ in range(max(args.warmup, 2)):
                    gs(batch)
                torch.cuda.synchronize()
                tg = time.perf_counter()
                for _ in range(args.steps):
                    gl = gs(batch)
                torch.cuda.synchronize()
                tg = time.perf_counter() - tg
                out["graph_step"] = {"ms_per_step": 1e3 * tg / args.steps, "value": B * D ** 3 * args.steps / tg, "unit": "voxels/s", "steps": args.steps,
                                     "loss": float(gl), "note": "whole training step captured once in a hipGraph and replayed (Trainer.fit at N = 1)"}
                hip_ops.SEED_STEP = None
            except Exception as e:                          # never lose the eager line over the extra measurement
                out["graph_step"] = {"error": repr(e)}
                hip_ops.SEED_STEP = None
        if sample_res is not None:
            out["sample"] = sample_res
        if not args.no_cpu_baseline and world == 1 and not args.sample_only:
            try:
                out["cpu_baseline"] = cpu_baseline(chs)
            except Exception as e:                          # never lose the GPU line over the host-side baseline
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if args.sample_only and "error" in (sample_res or {}):
        raise SystemExit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
