"""Minimal fit loop standing in for ``lightning.pytorch.Trainer`` on the VDM training path.

Reproduces what the reference's ``train()`` configures (/root/reference/trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:28-50):
``max_steps``, ``val_check_interval``, ``gradient_clip_val=0.5`` (global L2 norm), a checkpoint every
``every_n_train_steps`` (``{"state_dict": ...}`` - the key the reference reloads, src/utils.py:468-469), an LR / metric log
(JSONL instead of Comet: no network).  Data parallelism is a build-side addition: one process per GPU,
identical weights, ONE all-reduce per step of the flat gradient vector over RCCL (``backend="nccl"`` on ROCm) or gloo on CPU.
"""
import json
import os
import time

import torch
import torch.distributed as dist


def _accepts(fn, name, positional=0):
    """Does `fn` take a parameter called `name` (or **kwargs), or at least `positional` positional arguments (or *args)?  Decided from
    the signature - a TypeError raised INSIDE the user's method must surface, not silently switch graphing / sharding off."""
    import inspect
    try:
        ps = inspect.signature(fn).parameters.values()
    except (TypeError, ValueError):
        return False
    if any(p.kind == p.VAR_KEYWORD or p.name == name for p in ps):
        return True
    npos = sum(p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) for p in ps)
    return positional > 0 and (npos >= positional or any(p.kind == p.VAR_POSITIONAL for p in ps))


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(device_type):
    """Initialise torch.distributed from the torchrun environment if WORLD_SIZE > 1."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("VDM4CDM_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            torch.cuda.set_device(local_device_index(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_device_index(local_rank) if device_type == "cuda" else local_rank, world


def local_device_index(local_rank):
    """GPU of this rank: its LOCAL_RANK - or 0 for every rank under VDM4CDM_SHARE_GPU=1, the rehearsal mode that runs the N > 1 code
    path (shards, gradient buckets, barriers) with several ranks on ONE GPU (use VDM4CDM_DIST_BACKEND=gloo there: RCCL refuses two
    ranks on the same device)."""
    return 0 if os.environ.get("VDM4CDM_SHARE_GPU") == "1" else local_rank


def allreduce_mean_(t, world):
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(world)
    return t


def clip_grad_norm_flat_(params, max_norm, use_hip, want_norm=True):
    """Global-norm clipping without a host sync.  Returns the (device) total norm (None with want_norm=False on the HIP path, where
    the coefficient never exists as a tensor: K10 sums the squares, vdm_clip_scale derives min(1, max_norm / (norm + 1e-6)) inside the
    scaling pass)."""
    grads = [p.grad for p in params if p.grad is not None]
    if use_hip:
        from . import hip_ops as ops
        acc = torch.zeros(1, device=grads[0].device)
        big = [g for g in grads if g.numel() >= 1024 and g.is_contiguous() and g.data_ptr() % 16 == 0 and g.dtype == torch.float32]
        small = [g for g in grads if not any(g is b for b in big)]
        for g in big:
            ops.sumsq(g, acc)                                   # K10
        for g in small:
            acc += (g.float() ** 2).sum()
        for g in big:
            ops.clip_scale_(g, acc, max_norm)
        if small:
            coef = torch.clamp(max_norm / (acc.sqrt() + 1e-6), max=1.0)
            for g in small:
                g.mul_(coef.to(g.device))
        return acc.sqrt() if want_norm else None
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef.to(g.device))
    return total


class GraphedTrainStep:
    """The whole training step - forward diffusion, UNet forward, ELBO, UNet backward, global-norm clip, fused AdamW, weight re-packing -
    captured ONCE in a hipGraph and replayed.  Why: the host needs ~25 us per C-ABI call; in the deep UNet
    levels the kernels are shorter than that, and the main queue idles ~1 ms per step waiting for launches (profiles/r03_step_timeline.txt).
    What makes the step capturable: static shapes (one configuration per run), no host sync on the path, RNG seeds that are kernel
    arguments PLUS a device-side step counter mixed in by the kernels (hip_ops.SEED_STEP: dropout masks, noise fields and the time grid
    change from replay to replay although the host seeds are baked into the graph), AdamW with capturable state.  Single process only:
    with world > 1 the step stays eager (the RCCL buckets are issued from Python).  The caller must not hold the loss tensor of an
    earlier EAGER step when it builds this object: its autograd graph keeps an AccumulateGrad node bound to the default stream alive,
    which would run inside the capture (torch warns; the capture then fails)."""

    def __init__(self, model, opt, params, clip_val, batch, warmup=3):
        from . import hip_ops as ops
        assert ops.PROFILER is None, "per-launch events cannot be recorded inside a captured graph"
        assert ops.SEED_STEP is None, "another graphed step is being built / run in this process"
        self.model, self.opt, self.params, self.clip = model, opt, params, clip_val
        dev = params[0].device
        self.static = {k: ([t.clone() for t in v] if isinstance(v, (list, tuple)) else (None if v is None else v.clone())) for k, v in batch.items()}
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        # The device counter is mixed into the seeds ONLY while this object's own step runs (hip_ops.SEED_STEP is set inside _step and
        # cleared on the way out): eager steps of the same process - a ragged last batch, validation, a later fit - keep drawing u0 and
        # their Philox seeds from the host generators and never see a stale counter.
        inner = getattr(model, "model", model)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        # eager warm-up (allocator pools, workspaces, packed weights, optimizer state): NOT real steps - parameters, optimizer
        # moments / step count and the device counter are put back afterwards, so the first replay is optimizer step 1 on this batch
        keep = [p.detach().clone() for p in params]
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
            with torch.no_grad():
                for p, k in zip(params, keep):
                    p.copy_(k)
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
                self.counter.zero_()
            sm = getattr(inner, "score_model", None)
            if hasattr(sm, "mark_weights_dirty"):
                sm.mark_weights_dirty()
                if hasattr(sm, "repack_weights"):
                    sm.repack_weights()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        del keep
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.gnorm = self._step()
        self.replays = 0

    def _step(self):
        from . import hip_ops as ops
        ops.SEED_STEP = self.counter                            # every seed of THIS step is (host seed, device step counter)
        try:
            loss = self.model.training_step(self.static, 0)
            self.opt.zero_grad(set_to_none=True)
            loss.backward()
            gnorm = clip_grad_norm_flat_(self.params, self.clip, True, want_norm=True) if self.clip else None
            self.opt.step()
            ops.step_inc(self.counter)
        finally:
            ops.SEED_STEP = None
        return loss, gnorm

    def __call__(self, batch):
        for k, v in batch.items():
            if isinstance(v, (list, tuple)):
                for dst, src in zip(self.static[k], v):
                    dst.copy_(src, non_blocking=True)
            elif v is not None:
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        return self.loss


class Trainer:
    def __init__(self, max_steps=1_000_000, val_check_interval=1000, gradient_clip_val=0.5, every_n_train_steps=10_000,
                 default_root_dir="./data/logs", experiment_name="run", limit_val_batches=4, log_every_n_steps=50,
                 n_val_sampling_steps=250, device=None, enable_progress=True, graph_step=None):
        self.max_steps, self.val_check_interval, self.gradient_clip_val = max_steps, val_check_interval, gradient_clip_val
        self.every_n_train_steps = every_n_train_steps
        self.root = os.path.join(default_root_dir, experiment_name)
        self.limit_val_batches, self.log_every_n_steps = limit_val_batches, log_every_n_steps
        self.n_val_sampling_steps = n_val_sampling_steps
        self.device = device
        self.enable_progress = enable_progress
        # capture the training step in a hipGraph (single-process HIP runs); None: $VDM4CDM_GRAPH_STEP (default off: measured 1.9 vs 5.2 ms
        # per step at 32^3 where the host is the limiter, but 15.0 vs 14.7 ms at 128^3 and 23.0 vs 21.7 ms at 64^3 fp32 - bench.py --graph)
        self.graph_step = (os.environ.get("VDM4CDM_GRAPH_STEP", "0") != "0") if graph_step is None else bool(graph_step)
        self.global_step = 0
        self.history = []

    def _log(self, rec):
        self.history.append(rec)
        if self.rank == 0:
            os.makedirs(self.root, exist_ok=True)
            with open(os.path.join(self.root, "metrics.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")

    def save_checkpoint(self, model, epoch):
        if self.rank != 0:
            return None
        d = os.path.join(self.root, "checkpoints")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"epoch={epoch}-step={self.global_step}.ckpt")
        torch.save({"state_dict": model.state_dict(), "global_step": self.global_step, "epoch": epoch}, path)
        return path

    def fit(self, model, datamodule):
        dev = self.device
        if dev is None:
            dev = "cuda" if torch.cuda.is_available() else "cpu"
        dev_type = torch.device(dev).type
        self.rank, local_rank, self.world = init_distributed(dev_type)
        if dev_type == "cuda":
            dev = f"cuda:{local_rank}"
            torch.cuda.set_device(local_rank)
        model.to(dev)
        datamodule.device = dev
        params = [p for p in model.parameters() if p.requires_grad]
        if self.world > 1:                                   # identical weights on every rank
            for p in params:
                dist.broadcast(p.data, src=0)
        sm = model.model.score_model
        if hasattr(sm, "enable_ddp") and dev_type == "cuda":     # HIP backend: bucketed all-reduce inside the backward pass
            sm.enable_ddp(self.world)
        use_hip = dev_type == "cuda" and getattr(model.model.score_model, "backend", "") == "hip"
        graphed = self.graph_step and use_hip and self.world == 1 and getattr(model.model, "noise_schedule", "") == "fixed_linear"
        if use_hip:
            from . import hip_ops as ops
            if ops.ABLATE or "VDM4CDM_ABLATE_REDUCE" in os.environ:      # timing switches of tools/ablate_step.sh skip kernels: never train with them
                raise RuntimeError(f"VDM4CDM_ABLATE={sorted(ops.ABLATE)} is set: kernels are skipped and gradients are garbage - unset it")
        if _accepts(model.configure_optimizers, "capturable"):
            opt = model.configure_optimizers(capturable=graphed)
        else:                                                # (a user model with the reference's zero-argument signature)
            if graphed and self.rank == 0:
                print("Trainer: configure_optimizers() takes no `capturable` argument - the training step stays eager", flush=True)
            opt, graphed = model.configure_optimizers(), False
        gstep = None
        epoch, t0 = 0, time.time()
        model.train()
        while self.global_step < self.max_steps:
            n_batches = 0
            for batch in datamodule.train_dataloader(self.rank, self.world):
                n_batches += 1
                if graphed and gstep is None and self.max_steps - self.global_step >= 8:
                    # captured at the first batch (its eager warm-up steps are undone: parameters / optimizer state restored)
                    gstep = GraphedTrainStep(model, opt, params, self.gradient_clip_val, batch)
                if gstep is not None and gstep.static["x"].shape == batch["x"].shape:
                    loss, gnorm = gstep(batch), gstep.gnorm   # one hipGraph replay = the whole step
                else:                                         # eager step (N > 1 ranks, short runs, a ragged last batch, the torch backend)
                    loss = model.training_step(batch, self.global_step)
                    opt.zero_grad(set_to_none=True)
                    loss.backward()
                    synced = getattr(sm, "grad_synced", False)     # the HIP backward already averaged the flat UNet gradient (in buckets)
                    sm.grad_synced = False
                    for p in params:                              # one collective per remaining parameter tensor (<= 2 schedule scalars)
                        if p.grad is not None and not (synced and p is getattr(sm, "flat", None)):
                            allreduce_mean_(p.grad, self.world)
                    gnorm = None
                    will_log = (self.global_step + 1) % self.log_every_n_steps == 0 or self.global_step == 0
                    if self.gradient_clip_val:
                        gnorm = clip_grad_norm_flat_(params, self.gradient_clip_val, use_hip, want_norm=will_log)
                    opt.step()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0 or self.global_step == 1:
                    rec = {"step": self.global_step, "epoch": epoch, "lr": opt.param_groups[0]["lr"], "time": time.time() - t0,
                           "loss": float(loss), **dict(model.logged)}
                    if gnorm is not None:
                        rec["grad_norm"] = float(gnorm)
                    self._log(rec)
                    if self.enable_progress and self.rank == 0:
                        print(f"step {self.global_step} loss {rec['loss']:.4f} ({rec['time']:.1f}s)", flush=True)
                if self.val_check_interval and self.global_step % self.val_check_interval == 0:
                    self.validate(model, datamodule)
                if self.every_n_train_steps and self.global_step % self.every_n_train_steps == 0:
                    self.save_checkpoint(model, epoch)
                if self.global_step >= self.max_steps:
                    break
            if n_batches == 0:
                raise RuntimeError("empty training dataloader")
            epoch += 1
        return self

    @torch.no_grad()
    def validate(self, model, datamodule):
        model.eval()
        losses, last = [], None
        world = getattr(self, "world", 1)
        if _accepts(datamodule.val_dataloader, "rank", positional=2):      # the validation split is sharded like the training split
            loader = datamodule.val_dataloader(self.rank, world)
        else:                                                  # (a user datamodule with the reference's zero-argument signature)
            if world > 1 and self.rank == 0:
                print("Trainer: val_dataloader() takes no (rank, world): every rank evaluates the whole validation split", flush=True)
            loader = datamodule.val_dataloader()
        nseen = 0
        for i, batch in enumerate(loader):
            if i >= self.limit_val_batches:                    # (per rank: at world N the loss averages up to N * limit_val_batches batches)
                break
            nb = int(batch["x"].shape[0]) if isinstance(batch, dict) and torch.is_tensor(batch.get("x")) else 1
            losses.append(model.validation_step(batch, i).detach().float() * nb)      # sample-weighted: short / uneven last batches
            nseen += nb
            last = batch
        dev = losses[0].device if losses else torch.device("cpu")
        tot = torch.stack([torch.stack(losses).sum() if losses else torch.zeros((), device=dev),
                           torch.tensor(float(nseen), device=dev)])
        if world > 1:                                          # one collective, entered by EVERY rank (also one whose shard was empty)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rec = {"step": self.global_step, "val_loss": float(tot[0] / tot[1].clamp(min=1.0)), "val_samples": int(tot[1].item()), **dict(model.logged)}
        if last is not None and model.draw_figure is not None and self.rank == 0:
            x, kw = model._unpack(last)
            samples = model.draw_samples(batch_size=x.shape[0], n_sampling_steps=self.n_val_sampling_steps, **model._filter(kw))
            try:
                fig = model.draw_figure(last, samples)
                os.makedirs(self.root, exist_ok=True)
                fig.savefig(os.path.join(self.root, f"val_step{self.global_step}.png"))
            except Exception as e:                             # figures are diagnostics, never fatal to the fit loop
                rec["figure_error"] = repr(e)
        if world > 1:            # rank 0 drew the figure samples (n_val_sampling_steps denoise steps): the others wait HERE, explicitly,
            dist.barrier()       # not inside the first gradient all-reduce of the next training step
        self._log(rec)
        model.train()
        return rec
