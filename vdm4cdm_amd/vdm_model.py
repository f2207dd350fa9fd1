"""``vdm_model.VDM`` / ``vdm_model.LightVDM`` - variational diffusion model around the CUNet score network.

Mirrors the interface the reference uses for ``mltools.models.vdm_model``:
* construction ``LightVDM(score_model=, draw_figure=, gamma_min=, gamma_max=, noise_schedule=, learning_rate=)``
  (/root/reference/trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:128-132, train_uc_uc_from_field_name.py:115-120)
* ``.model.score_model``, ``.model.gamma_min/gamma_max/w_cfg``, ``.model.sample_zs_given_zt(zt=, t=, s=, return_ddnm=)``,
  ``.model.sample_zt_given_zs(zs=, t=, s=)``, ``.device`` (/root/reference/src/utils.py:286-299)
* ``.draw_samples(batch_size, n_sampling_steps, verbose, return_all, **kwargs)`` (notebook frame vdm_model.py:531-557;
  call sites /root/reference/generate_3D.py:61)
* ``load_state_dict(torch.load(p)["state_dict"])`` (/root/reference/src/utils.py:468-469)
The arithmetic is spec D9-D12 of SURVEY.md section 8 (the mltools source is not in the reference tree).

On a GPU with the HIP backend the forward diffusion, the ELBO reductions and the ancestral update are the
HIP kernels K7-K9 and the denoise step is captured once in a hipGraph and replayed for every step
(per-step scalars come from a device table indexed by a device-side step counter).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

DATA_NOISE = 1.0e-3
# Fused head of the HIP training step (vdm_diffuse_pack + vdm_loss_terms_rng; VDM4CDM_FUSED_HEAD=0: randn -> diffuse -> pack_input, A/B)
FUSED_HEAD = os.environ.get("VDM4CDM_FUSED_HEAD", "1") != "0"


class _DiffusionLossFn(torch.autograd.Function):
    """sum_n coef_n * sum (eps_hat - eps)^2 / 2 ... via the fused HIP reduction (K8)."""

    @staticmethod
    def forward(ctx, eps_hat, x, eps, eps0, s0a0, coef, sums):
        from . import hip_ops as ops
        d = torch.empty_like(eps_hat)
        ops.loss_terms(x, eps, eps_hat.contiguous(), eps0, s0a0, coef, sums, d)
        ctx.save_for_backward(d)
        # coef_n = 2 w_n  ->  loss = sum_n w_n S_n = 0.5 sum_n coef_n S_n
        return 0.5 * (coef * sums[:, 0]).sum()

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return g * d, None, None, None, None, None, None


class _ElboFn(torch.autograd.Function):
    """(elbo, [diffusion, latent, reconstruction]) in bits/dim: K8 (fused per-sample reductions + d eps_hat) and the scalar assembly
    (vdm_elbo_assemble) - three launches; the metrics are not differentiable, d elbo / d eps_hat = coef_n (eps_hat - eps)."""

    @staticmethod
    def forward(ctx, eps_hat, x, eps, eps0, s0a0, coef, consts, rng=None):
        """rng = ((seed, stream) of eps, (seed, stream) of eps0): fields passed as None are regenerated in the kernel (fused head)."""
        from . import hip_ops as ops
        d = torch.empty_like(eps_hat)
        sums = torch.zeros(x.shape[0], 3, device=x.device)
        ops.loss_terms(x, eps, eps_hat.contiguous(), eps0, s0a0, coef, sums, d, rng=rng)
        out = ops.elbo_assemble(sums, coef, *consts)
        ctx.save_for_backward(d)
        elbo, parts = out[0], out[1:]
        ctx.mark_non_differentiable(parts)
        return elbo, parts

    @staticmethod
    def backward(ctx, g, _):
        (d,) = ctx.saved_tensors
        return g * d, None, None, None, None, None, None, None


class VDM(nn.Module):
    def __init__(self, score_model, noise_schedule="fixed_linear", gamma_min=-13.3, gamma_max=13.3,
                 antithetic_time_sampling=True, data_noise=DATA_NOISE, w_cfg=None):
        super().__init__()
        assert noise_schedule in ("fixed_linear", "learned_linear")
        self.score_model = score_model
        self.noise_schedule = noise_schedule
        self.gamma_min = float(gamma_min)
        self.gamma_max = float(gamma_max)
        self.antithetic_time_sampling = antithetic_time_sampling
        self.data_noise = float(data_noise)
        self.w_cfg = w_cfg
        if noise_schedule == "learned_linear":                   # D9: gamma(t) = b + |w| t
            self.gamma_b = nn.Parameter(torch.tensor(self.gamma_min))
            self.gamma_w = nn.Parameter(torch.tensor(self.gamma_max - self.gamma_min))
        self._graph = None

    # ---------------------------------------------------------------- schedule (D9)
    def gamma(self, t):
        if self.noise_schedule == "learned_linear":
            return self.gamma_b + self.gamma_w.abs() * t
        return self.gamma_min + (self.gamma_max - self.gamma_min) * t

    def dgamma_dt(self, t):
        if self.noise_schedule == "learned_linear":
            return self.gamma_w.abs() * torch.ones_like(t)
        return (self.gamma_max - self.gamma_min) * torch.ones_like(t)

    @staticmethod
    def alpha(g):
        return torch.sqrt(torch.sigmoid(-g))

    @staticmethod
    def sigma(g):
        return torch.sqrt(torch.sigmoid(g))

    def _hip(self, ref):
        return getattr(self.score_model, "backend", None) == "hip" and ref.is_cuda

    # ---------------------------------------------------------------- score
    def get_pred_noise(self, zt, gamma_t, **kwargs):
        """notebook frame vdm_model.py:309-327."""
        t = (gamma_t - self.gamma_min) / (self.gamma_max - self.gamma_min)
        if self.w_cfg is None or self.training:
            return self.score_model(zt, t=t, **kwargs)
        eps_c, eps_u = self._cfg_pair(zt, t, kwargs)
        return (1.0 + self.w_cfg) * eps_c - self.w_cfg * eps_u

    @staticmethod
    def cfg_mask(v_conditionings):
        """The "masked out" vector conditionings of the unguided branch (frame vdm_model.py:327: "Need v_conditionings to mask out";
        the masking value is not in the reference tree - [INFERRED] zeros, the usual null token of a vector conditioning)."""
        return [torch.zeros_like(v) for v in v_conditionings]

    def _cfg_pair(self, zt, t, kwargs):
        """Classifier-free guidance (notebook frame vdm_model.py:318-327): conditional and v-masked noise estimates from ONE
        batch-doubled UNet forward (rows 0..B-1 with the given v_conditionings, rows B..2B-1 with them masked)."""
        assert "v_conditionings" in kwargs, "Need v_conditionings to mask out"
        B = zt.shape[0]
        vs = [v.to(zt.device).expand(B, *v.shape[1:]) if v.shape[0] != B else v.to(zt.device) for v in kwargs["v_conditionings"]]
        kw = dict(kwargs)
        kw["v_conditionings"] = [torch.cat([v, m], dim=0) for v, m in zip(vs, self.cfg_mask(vs))]
        if kw.get("s_conditioning") is not None:
            sc = kw["s_conditioning"]
            sc = sc.expand(B, *sc.shape[1:]) if sc.shape[0] != B else sc
            kw["s_conditioning"] = torch.cat([sc, sc], dim=0)
        t2 = torch.cat([t.expand(B), t.expand(B)]) if torch.is_tensor(t) else t
        eps = self.score_model(torch.cat([zt, zt], dim=0), t=t2, **kw)
        return eps[:B], eps[B:]

    # ---------------------------------------------------------------- loss (D10)
    @staticmethod
    def _rank_world():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _rank_noise_gen(self, device, rank):
        g = getattr(self, "_noise_gen", None)
        if g is None or g.device != torch.device(device) or self._noise_gen_key != (torch.initial_seed(), rank):
            g = torch.Generator(device=device)
            g.manual_seed((torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank + 1)) & 0x7fffffffffffffff)
            self._noise_gen, self._noise_gen_key = g, (torch.initial_seed(), rank)
        return g

    def sample_times(self, B, device):
        return stratified_times(B, device, self.antithetic_time_sampling)

    def get_loss(self, x, times=None, eps=None, eps0=None, **kwargs):
        """Continuous-time ELBO in bits/dim.  Returns (loss, metrics dict)."""
        B = x.shape[0]
        numel = x[0].numel()
        bpd = 1.0 / (numel * math.log(2.0))
        x = x.to(torch.float32).contiguous()
        hip = self._hip(x)
        if not hip:
            if times is None:
                times = self.sample_times(B, x.device)
            g_t = self.gamma(times)
        # 0-dim CPU tensors act as scalars (no device sync for the fixed schedule)
        # (fp64: var1 - log(var1) - 1 ~ 1e-12 cancels catastrophically in fp32)
        g0 = self.gamma(torch.zeros((), dtype=torch.float64))
        g1 = self.gamma(torch.ones((), dtype=torch.float64))
        var1 = torch.sigmoid(g1)
        a0, s0 = self.alpha(g0), self.sigma(g0)
        bc = (B,) + (1,) * (x.dim() - 1)
        if hip:
            from . import hip_ops as ops
            if self.noise_schedule != "fixed_linear":
                raise NotImplementedError("HIP training path supports noise_schedule='fixed_linear' (all 3D scripts); "
                                          "'learned_linear' needs d loss / d z_t which the HIP backward does not emit")
            rank, world = self._rank_world()               # (Philox stream id = 2*rank + {1,2}: different noise fields per rank)
            # Fused head: with no noise supplied, eps and eps0 are never materialised - K7 draws eps from its Philox
            # counters while it forms z_t and writes conv_in's packed input in the same pass, K8 regenerates both fields from the same
            # counters.  The host draws the two seeds exactly as the unfused path does (same generator state -> same noise fields).
            sm = self.score_model
            fuse = (FUSED_HEAD and eps is None and eps0 is None and numel % 4 == 0 and (self.w_cfg is None or self.training)
                    and x.dim() == 5 and getattr(sm, "s_conditioning_channels", 0) <= 1)
            rng = None
            if fuse:
                rng = ((noise_seed(), 2 * rank + 1), (noise_seed(), 2 * rank + 2))
            else:
                if eps is None:
                    eps = ops.randn(torch.empty_like(x), noise_seed(), 2 * rank + 1)
                if eps0 is None:
                    eps0 = ops.randn(torch.empty_like(x), noise_seed(), 2 * rank + 2)
            # the scalar side of the step in ONE launch: time grid (stratified over the global batch), alpha_t, sigma_t, the per-sample
            # loss weight 2 w_n = gamma'(t) bpd / B and the network's normalised time - no ATen launch between the noise draw and K7
            if times is None and self.antithetic_time_sampling and ops.SEED_STEP is not None:
                # graph-captured step (trainer.GraphedTrainStep): u0 comes out of the kernel, from a fixed seed and the device step counter
                if getattr(self, "_graph_seed", None) is None:
                    self._graph_seed = noise_seed()
                sc = ops.train_scalars(B, x.device, rank, world, self.gamma_min, self.gamma_max, bpd / B, seed=self._graph_seed)
            elif times is None and self.antithetic_time_sampling:
                u0 = torch.rand(1, device=x.device, generator=train_generator(x.device))
                sc = ops.train_scalars(B, x.device, rank, world, self.gamma_min, self.gamma_max, bpd / B, u0=u0)
            else:
                if times is None:
                    times = self.sample_times(B, x.device)
                sc = ops.train_scalars(B, x.device, 0, 1, self.gamma_min, self.gamma_max, bpd / B,
                                       times=times.to(device=x.device, dtype=torch.float32).contiguous())
            if fuse:
                s_c = kwargs.get("s_conditioning") if sm.s_conditioning_channels else None
                assert s_c is not None or not sm.s_conditioning_channels, "s_conditioning_channels=1 needs s_conditioning"
                if s_c is not None:
                    s_c = s_c.to(device=x.device, dtype=torch.float32).expand(x.shape).contiguous()
                dt = torch.bfloat16 if sm.precision == "bf16" else torch.float32
                z_t, xin = ops.diffuse_pack(x, s_c, sc[1], sc[2], dt, seed=rng[0][0], stream_id=rng[0][1], want_z=True)
                eps_hat = self.score_model(z_t, t=sc[4], _packed_input=xin, **kwargs)
            else:
                z_t = ops.diffuse(x, eps.contiguous(), sc[1], sc[2])
                if self.w_cfg is None or self.training:    # get_pred_noise's plain branch, t_norm straight from the scalar kernel
                    eps_hat = self.score_model(z_t, t=sc[4], **kwargs)
                else:
                    eps_hat = self.get_pred_noise(z_t, self.gamma(sc[0]), **kwargs)
            dn = self.data_noise
            consts = (float(0.5 * numel * (var1 - torch.log(var1) - 1.0) * bpd), float(0.5 * (1.0 - var1) * bpd),
                      float(0.5 / dn ** 2 * bpd), float(numel * (math.log(dn) + 0.5 * math.log(2 * math.pi)) * bpd))
            loss, parts = _ElboFn.apply(eps_hat, x, eps, None if eps0 is None else eps0.contiguous(), float(s0 / a0), sc[3], consts, rng)
            metrics = {"elbo": loss.detach(), "diffusion_loss": parts[0], "latent_loss": parts[1], "reconstruction_loss": parts[2]}
            return loss, metrics
        else:
            red = tuple(range(1, x.dim()))
            rank, world = self._rank_world()
            if eps is None or eps0 is None:                    # torch backend: a per-rank generator (seed ^ rank), never the global one
                g = self._rank_noise_gen(x.device, rank)
                eps = torch.randn(x.shape, device=x.device, generator=g) if eps is None else eps
                eps0 = torch.randn(x.shape, device=x.device, generator=g) if eps0 is None else eps0
            z_t = self.alpha(g_t).view(bc) * x + self.sigma(g_t).view(bc) * eps
            eps_hat = self.get_pred_noise(z_t, g_t, **kwargs)
            diff = (0.5 * self.dgamma_dt(times) * ((eps - eps_hat) ** 2).sum(red)).mean() * bpd
            sum_x2 = (x ** 2).sum(red)
            sum_r2 = (((s0 / a0).float() * eps0) ** 2).sum(red)
        latent = (0.5 * ((numel * (var1 - torch.log(var1) - 1.0)).float() + (1.0 - var1).float() * sum_x2)).mean() * bpd
        recons = (0.5 * sum_r2 / self.data_noise ** 2
                  + numel * (math.log(self.data_noise) + 0.5 * math.log(2 * math.pi))).mean() * bpd
        loss = diff + latent + recons
        metrics = {"elbo": loss.detach(), "diffusion_loss": diff.detach(), "latent_loss": latent.detach(),
                   "reconstruction_loss": recons.detach()}
        return loss, metrics

    # ---------------------------------------------------------------- sampler (D12)
    def _as_t(self, v, ref):
        return torch.as_tensor(v, dtype=torch.float32, device=ref.device)

    def sample_zs_given_zt(self, zt, t, s, return_ddnm=False, conditioning=None, **kwargs):
        """One ancestral step t -> s (notebook frame vdm_model.py:370-378).  `conditioning` is accepted and
        ignored: /root/reference/src/utils.py:296 still passes the pre-CUNet kwarg ``conditioning=None``."""
        t, s = self._as_t(t, zt), self._as_t(s, zt)
        gamma_t, gamma_s = self.gamma(t), self.gamma(s)
        c = -torch.expm1(gamma_s - gamma_t)
        alpha_t, alpha_s = self.alpha(gamma_t), self.alpha(gamma_s)
        sigma_t, sigma_s = self.sigma(gamma_t), self.sigma(gamma_s)
        pred_noise = self.get_pred_noise(zt=zt, gamma_t=gamma_t, **kwargs)
        if not return_ddnm:
            mean = alpha_s / alpha_t * (zt - c * sigma_t * pred_noise)
            scale = sigma_s * torch.sqrt(c)
            return mean + scale * torch.randn_like(zt)
        x_0t = (zt - sigma_t * pred_noise) / alpha_t
        return (alpha_s / alpha_t) * (1.0 - c), alpha_s * c, x_0t, sigma_s * torch.sqrt(c)

    def sample_zt_given_zs(self, zs, t, s):
        t, s = self._as_t(t, zs), self._as_t(s, zs)
        gamma_t, gamma_s = self.gamma(t), self.gamma(s)
        alpha_ts = self.alpha(gamma_t) / self.alpha(gamma_s)
        var = torch.sigmoid(gamma_t) - alpha_ts ** 2 * torch.sigmoid(gamma_s)
        return alpha_ts * zs + torch.sqrt(var) * torch.randn_like(zs)

    def step_table(self, n_sampling_steps):
        """Host fp64 table [n, 4] = {alpha_s/alpha_t, c*sigma_t, sigma_s*sqrt(c), t_norm} on the fp32 time grid
        ``linspace(1, 0, n+1)`` (/root/reference/src/utils.py:286)."""
        steps = torch.linspace(1.0, 0.0, n_sampling_steps + 1).double()
        with torch.no_grad():
            if self.noise_schedule == "learned_linear":
                g = (self.gamma_b.double().cpu() + self.gamma_w.abs().double().cpu() * steps)
            else:
                g = self.gamma_min + (self.gamma_max - self.gamma_min) * steps
        g_t, g_s = g[:-1], g[1:]
        c = -torch.expm1(g_s - g_t)
        a_t, a_s = torch.sqrt(torch.sigmoid(-g_t)), torch.sqrt(torch.sigmoid(-g_s))
        s_t, s_s = torch.sqrt(torch.sigmoid(g_t)), torch.sqrt(torch.sigmoid(g_s))
        t_norm = (g_t - self.gamma_min) / (self.gamma_max - self.gamma_min)
        return torch.stack([a_s / a_t, c * s_t, s_s * torch.sqrt(c), t_norm], dim=1)

    @torch.no_grad()
    def sample(self, batch_size, n_sampling_steps, device, z=None, return_all=False, verbose=False,
               noises=None, seed=None, use_graph=True, **kwargs):
        """Ancestral sampling (notebook frame vdm_model.py:429-442).  `noises` (optional list of n tensors) and
        `seed` make a chain reproducible; on the HIP backend the step is a replayed hipGraph."""
        shape = (batch_size, *self.score_model.shape)
        if z is None:
            z = torch.randn(shape, device=device) if seed is None else \
                torch.randn(shape, generator=torch.Generator().manual_seed(int(seed))).to(device)
        else:
            z = z.clone()                                    # the HIP path updates z in place: never the caller's tensor
        z = z.to(device=device, dtype=torch.float32).contiguous()
        if self._hip(z):                                     # (return_all: the same captured step, z copied out after every replay)
            return self._sample_hip(z, n_sampling_steps, noises, seed, verbose, use_graph, kwargs, return_all)
        steps = torch.linspace(1.0, 0.0, n_sampling_steps + 1, device=device)
        zs = []
        rng = range(n_sampling_steps)
        if verbose:
            try:
                from tqdm import trange
                rng = trange(n_sampling_steps, desc="sampling")
            except ImportError:
                pass
        gen = None
        if noises is None and seed is not None:              # a seeded chain is reproducible on this path too (per-step noise from
            gen = torch.Generator().manual_seed(int(seed) + 1)   # the chain's own generator, not the global RNG)
        for i in rng:
            if noises is None and gen is None:
                z = self.sample_zs_given_zt(zt=z, t=steps[i], s=steps[i + 1], **kwargs)
            else:
                w_z, w_x, x0, scale = self.sample_zs_given_zt(zt=z, t=steps[i], s=steps[i + 1], return_ddnm=True, **kwargs)
                eps = noises[i].to(z) if noises is not None else torch.randn(z.shape, generator=gen).to(z)
                z = w_z * z + w_x * x0 + scale * eps
            if return_all:
                zs.append(z)
        return torch.stack(zs, dim=0) if return_all else z

    def _sample_hip(self, z, n, noises, seed, verbose, use_graph, kwargs, return_all=False):
        coef = self.step_table(n).to(device=z.device, dtype=torch.float32).contiguous()
        cfg = self.w_cfg is not None and not self.training
        if cfg:
            assert "v_conditionings" in kwargs, "Need v_conditionings to mask out"
        return hip_graph_sampler(self.score_model, z, coef, noises, seed, verbose, use_graph, kwargs.get("s_conditioning"),
                                 list(kwargs.get("v_conditionings") or []), w_cfg=float(self.w_cfg) if cfg else None,
                                 mask_fn=self.cfg_mask, return_all=return_all)


def hip_graph_sampler(net, z, coef, noises, seed, verbose, use_graph, s_cond, v_conditionings, w_cfg=None, mask_fn=None,
                      return_all=False):
    """The multi-step sampling loop on the HIP backend, shared by the VDM ancestral sampler and the SFM Euler integrator: per step
    [conditioning-table row gather, UNet forward, fused update z <- ratio * (z - cs * net_out) + scale * noise, step counter + 1],
    captured once in a hipGraph and replayed; coef[n][4] = {ratio, cs, scale, network time} is read on the device at the row of the
    device-side step counter.  z is updated in place and returned; return_all: the stack [n, B, ...] of z after every step instead
    (frame vdm_model.py:429-442: ``return_all``), copied out between the replays of the same graph."""
    from . import hip_ops as ops
    from .unet_hip import hip_unet_apply
    dev = z.device
    n = coef.shape[0]
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    B = z.shape[0]
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
    noise_buf = torch.empty_like(z) if noises is not None else None

    # The conditioning of every step is known up front: ONE K6 launch embeds all n time values (one table row per step), one more
    # the vector conditionings; inside the step a single gather-add kernel builds the table (device-side step counter).
    W = net.table_width
    table_t = table_v = None
    cfg = w_cfg is not None
    with torch.no_grad():
        fl = net.flat.detach()
        if net.t_conditioning:
            table_t = ops.CondTable(net.cond_specs(coef[:, 3].contiguous(), None, fl, which="t"), n, W).forward(save=False)
        vs = [v.to(device=dev, dtype=torch.float32).expand(B, -1).contiguous() for v in v_conditionings]
        if cfg:                                            # guided + v-masked rows of one batch-doubled forward (VDM._cfg_pair)
            vs = [torch.cat([v, m], dim=0).contiguous() for v, m in zip(vs, mask_fn(vs))]
        R = 2 * B if cfg else B                            # rows the UNet sees
        if vs:
            table_v = ops.CondTable(net.cond_specs(None, vs, fl, which="v"), R, W).forward(save=False)
    if cfg and s_cond is not None:
        s_cond = s_cond.to(dev).expand(B, *s_cond.shape[1:])
        s_cond = torch.cat([s_cond, s_cond], dim=0).contiguous()
    table = torch.zeros(R, W, device=dev)
    zz = torch.empty(R, *z.shape[1:], device=dev) if cfg else z

    def one_step():
        if table_t is not None or table_v is not None:
            ops.cond_table_step(table_t, table_v, step, R, W, table)
        if cfg:
            zz[:B].copy_(z)
            zz[B:].copy_(z)
        eps_hat = hip_unet_apply(net, zz, s_cond, table=table).contiguous()
        if cfg:                                            # blend inside K9: the guided estimate is never materialised
            ops.ancestral_step(z, eps_hat[:B], noise_buf, coef, step, seed, eps_uncond=eps_hat[B:], w_cfg=w_cfg)
        else:
            ops.ancestral_step(z, eps_hat, noise_buf, coef, step, seed)
        ops.step_inc(step)

    graph = None
    if use_graph and n > 2:
        # warm-up on a side stream (packs weights, sizes the allocator), then capture one step
        side = torch.cuda.Stream(device=dev)
        z_keep = z.clone()
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if noise_buf is not None:
                noise_buf.copy_(noises[0].to(z))
            one_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        z.copy_(z_keep)
        step.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            one_step()
        z.copy_(z_keep)             # capture does not execute, but keep the state explicit
        step.zero_()
    zs = torch.empty((n,) + tuple(z.shape), dtype=z.dtype, device=dev) if return_all else None
    # supplied noise fields (tests / oracle comparisons; the product path draws them in-kernel): uploaded in blocks of <= 256 MB and copied
    # device-to-device per step - one pageable host-to-device copy between every two graph replays was the only thing the tests that
    # intermittently took the process down (round 4) did differently from the product's sampling loop
    blk, dev_block = 1, None
    if noise_buf is not None:
        blk = max(1, min(n, (256 << 20) // max(1, z.numel() * z.element_size())))
    for i in range(n):
        if noise_buf is not None:
            if i % blk == 0:
                dev_block = torch.stack([noises[k].to(dtype=z.dtype) for k in range(i, min(n, i + blk))]).to(dev)
            noise_buf.copy_(dev_block[i % blk])
        if graph is not None:
            graph.replay()
        else:
            one_step()
        if zs is not None:
            zs[i].copy_(z)
        if verbose and (i % 50 == 0 or i == n - 1):
            print(f"sampling: {i + 1}/{n}", flush=True)
    return zs if return_all else z


_TRAIN_GENS = {}


def train_generator(device):
    """The generator of the TRAINING step's own random draws (the stratification offset u0, the Philox seeds of the noise fields) on
    `device`: seeded once from torch.initial_seed() - identical on every rank after seed_everything(42) - and consumed by nothing
    else, so validation sampling on rank 0, user code or a rank-dependent number of draws elsewhere can never de-synchronise the
    ranks' u0 / seed sequence (the global generators are shared with all of those)."""
    key = str(torch.device(device))
    g = _TRAIN_GENS.get(key)
    if g is None:
        g = torch.Generator(device=device)
        g.manual_seed(torch.initial_seed() ^ 0x5EED)
        _TRAIN_GENS[key] = g
    return g


def reset_train_generators():
    """Forget the training generators: the next training step re-creates them from torch.initial_seed().  Call after
    torch.manual_seed / seed_everything when a run has to be reproduced inside one process (tests; entry.seed_everything does)."""
    _TRAIN_GENS.clear()


def noise_seed():
    """A fresh Philox seed for one noise field of the training step (host side; same sequence on every rank - the Philox stream id
    carries the rank)."""
    return int(torch.randint(0, 2 ** 62, (1,), generator=train_generator("cpu")).item())


def stratified_times(B, device, antithetic=True):
    """D10 antithetic sampling t_i = (u0 + i/B) mod 1, stratified over the GLOBAL batch under data parallelism: u0 comes from
    train_generator (same on all ranks) and rank r takes strata r*B .. r*B+B-1 of world*B - the variance of the t-sampling shrinks
    with the world size instead of every rank drawing the same B times."""
    rank, world = VDM._rank_world()
    g = train_generator(device)
    if antithetic:
        u0 = torch.rand(1, device=device, generator=g)
        i = torch.arange(B, device=device, dtype=torch.float32) + rank * B
        return torch.remainder(u0 + i / (world * B), 1.0)
    t = torch.rand(world * B, device=device, generator=g)
    return t[rank * B:(rank + 1) * B]


class LightVDM(nn.Module):
    """Stand-in for the LightningModule of the reference: same constructor / attributes / methods that the
    reference scripts touch; the fit loop lives in vdm4cdm_amd.trainer.Trainer."""

    def __init__(self, score_model, draw_figure=None, gamma_min=-13.3, gamma_max=13.3, noise_schedule="fixed_linear",
                 learning_rate=3.0e-4, **vdm_kwargs):
        super().__init__()
        self.model = VDM(score_model, noise_schedule=noise_schedule, gamma_min=gamma_min, gamma_max=gamma_max, **vdm_kwargs)
        self.draw_figure = draw_figure
        self.learning_rate = learning_rate
        self._logged = {}

    @property
    def device(self):
        return self.model.score_model.flat.device

    def log_dict(self, d):
        """Keeps the latest value per key WITHOUT reading it back (a float() here would synchronise host and GPU in every step);
        `logged` converts on access, i.e. when a logger actually wants the numbers."""
        self._logged.update({k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()})

    @property
    def logged(self):
        return {k: float(v) for k, v in self._logged.items()}

    @staticmethod
    def _unpack(batch):
        """Batch dict contract (/root/reference/trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:75-76)."""
        kw = {}
        if batch.get("conditioning") is not None:
            kw["s_conditioning"] = batch["conditioning"]
        if batch.get("conditioning_values") is not None:
            kw["v_conditionings"] = list(batch["conditioning_values"])
        return batch["x"], kw

    def _filter(self, kw):
        sm = self.model.score_model
        if not getattr(sm, "s_conditioning_channels", 1):
            kw.pop("s_conditioning", None)
        if not getattr(sm, "v_conditioning_dims", [1]):
            kw["v_conditionings"] = []
        return kw

    def training_step(self, batch, batch_idx=0):
        x, kw = self._unpack(batch)
        loss, metrics = self.model.get_loss(x, **self._filter(kw))
        self.log_dict({f"train/{k}": v for k, v in metrics.items()})
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        x, kw = self._unpack(batch)
        loss, metrics = self.model.get_loss(x, **self._filter(kw))
        self.log_dict({f"val/{k}": v for k, v in metrics.items()})
        return loss

    def configure_optimizers(self, capturable=False):
        """capturable: the optimizer state (step count) lives on the device, so that the step can be captured in a hipGraph
        (trainer.GraphedTrainStep)."""
        fused = all(p.is_cuda for p in self.parameters())                       # one fused kernel over the flat vector
        opt = torch.optim.AdamW(self.parameters(), lr=self.learning_rate, fused=fused, capturable=bool(capturable and fused))       # D11
        sm = self.model.score_model
        if hasattr(sm, "mark_weights_dirty"):      # fused steps do not bump Tensor._version: tell the HIP executor to re-pack
            def _after_step(*_):
                sm.mark_weights_dirty()
                if hasattr(sm, "repack_weights"):
                    sm.repack_weights()
            opt.register_step_post_hook(_after_step)
        return opt

    def draw_samples(self, batch_size, n_sampling_steps=250, verbose=False, return_all=False, **kwargs):
        return self.model.sample(batch_size=batch_size, n_sampling_steps=n_sampling_steps, device=self.device,
                                 verbose=verbose, return_all=return_all, **kwargs)

    # state dict: {"model.score_model.<name>": tensor, "model.gamma_b": ..., ...}
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = {} if destination is None else destination
        self.model.score_model.state_dict(destination=out, prefix=prefix + "model.score_model.", keep_vars=keep_vars)
        if self.model.noise_schedule == "learned_linear":
            out[prefix + "model.gamma_b"] = self.model.gamma_b.detach().clone()
            out[prefix + "model.gamma_w"] = self.model.gamma_w.detach().clone()
        return out

    def load_state_dict(self, state_dict, strict=True):
        pre = "model.score_model."
        sub = {k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}
        res = self.model.score_model.load_state_dict(sub, strict=strict)
        if self.model.noise_schedule == "learned_linear":
            with torch.no_grad():
                if "model.gamma_b" in state_dict:
                    self.model.gamma_b.copy_(state_dict["model.gamma_b"])
                if "model.gamma_w" in state_dict:
                    self.model.gamma_w.copy_(state_dict["model.gamma_w"])
        return res
