"""Oracle: VDM noise schedule, ELBO loss and ancestral sampler (fp32/fp64 scalars, CPU torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (``mltools.models.vdm_model``
is not in the reference tree).  Restates spec decisions D9-D12 of SURVEY.md section 8 and follows the
only in-tree fragments of the reference implementation:

* ``get_pred_noise``: ``score_model(zt, t=(gamma_t-gamma_min)/(gamma_max-gamma_min), **kwargs)``
  (notebook traceback ``vdm_model.py:318-324``)
* ``sample_zs_given_zt``: ``mean = alpha_s / alpha_t * (zt - c * sigma_t * pred_noise)``
  (``vdm_model.py:370-378``); DDNM return ``(w_z, w_x_0t, x_0t, scale)`` (/root/reference/src/utils.py:296-299)
* ``sample``: ``for i in range(n): z = sample_zs_given_zt(zt=z, t=steps[i], s=steps[i+1], **kwargs)``
  (``vdm_model.py:429-442``) with ``steps = linspace(1, 0, n+1)`` and ``z ~ randn((B, *score_model.shape))``
  (/root/reference/src/utils.py:286-287)
* ``sample_zt_given_zs(zs, t, s)`` (/root/reference/src/utils.py:294)
"""
import math

import torch

DATA_NOISE = 1.0e-3      # D10: reconstruction-likelihood std


class Schedule:
    """D9.  fixed_linear: gamma(t)=gmin+(gmax-gmin)t ; learned_linear: gamma(t)=b+|w|t."""

    def __init__(self, gamma_min=-13.3, gamma_max=13.3, kind="fixed_linear", b=None, w=None):
        self.gamma_min, self.gamma_max, self.kind = float(gamma_min), float(gamma_max), kind
        self.b = torch.tensor(self.gamma_min if b is None else b, dtype=torch.float64)
        self.w = torch.tensor(self.gamma_max - self.gamma_min if w is None else w, dtype=torch.float64)

    def gamma(self, t):
        t = torch.as_tensor(t, dtype=torch.float64)
        return self.b + self.w.abs() * t

    def dgamma_dt(self, t):
        return self.w.abs() * torch.ones_like(torch.as_tensor(t, dtype=torch.float64))

    @staticmethod
    def alpha(g):
        return torch.sqrt(torch.sigmoid(-g))

    @staticmethod
    def sigma(g):
        return torch.sqrt(torch.sigmoid(g))


def antithetic_times(u0, B):
    """D10: t_i = (u0 + i/B) mod 1."""
    return torch.remainder(float(u0) + torch.arange(B, dtype=torch.float64) / B, 1.0)


def vdm_loss(score_fn, sched, x, times, eps, eps0):
    """Continuous-time VDM ELBO in bits/dim (D10).

    score_fn(z_t, t_norm) -> eps_hat.  Returns dict(elbo, diffusion_loss, latent_loss,
    reconstruction_loss) (batch means) plus per-sample vectors under *_per_sample.
    """
    B = x.shape[0]
    red = tuple(range(1, x.dim()))
    bc = (B,) + (1,) * (x.dim() - 1)
    bpd = 1.0 / (x[0].numel() * math.log(2.0))
    g_t = sched.gamma(times)
    a_t, s_t = sched.alpha(g_t).to(x.dtype), sched.sigma(g_t).to(x.dtype)
    z_t = a_t.view(bc) * x + s_t.view(bc) * eps
    t_norm = ((g_t - sched.gamma_min) / (sched.gamma_max - sched.gamma_min)).to(x.dtype)
    eps_hat = score_fn(z_t, t_norm)
    diff = 0.5 * sched.dgamma_dt(times).to(x.dtype) * ((eps - eps_hat) ** 2).sum(red)
    g1 = sched.gamma(1.0)
    var1 = torch.sigmoid(g1)                       # fp64: var1 - log(var1) - 1 ~ 1e-12 cancels in fp32
    numel = x[0].numel()
    latent = (0.5 * (numel * (var1 - torch.log(var1) - 1.0) + (1.0 - var1) * (x.double() ** 2).sum(red))).to(x.dtype)
    g0 = sched.gamma(0.0)
    a0, s0 = sched.alpha(g0).to(x.dtype), sched.sigma(g0).to(x.dtype)
    z0_rescaled = (a0 * x + s0 * eps0) / a0
    logp = -0.5 * ((x - z0_rescaled) / DATA_NOISE) ** 2 - math.log(DATA_NOISE) - 0.5 * math.log(2 * math.pi)
    recons = -logp.sum(red)
    out = {
        "diffusion_loss_per_sample": diff * bpd,
        "latent_loss_per_sample": latent * bpd,
        "reconstruction_loss_per_sample": recons * bpd,
        "eps_hat": eps_hat,
        "z_t": z_t,
    }
    out["diffusion_loss"] = out["diffusion_loss_per_sample"].mean()
    out["latent_loss"] = out["latent_loss_per_sample"].mean()
    out["reconstruction_loss"] = out["reconstruction_loss_per_sample"].mean()
    out["elbo"] = out["diffusion_loss"] + out["latent_loss"] + out["reconstruction_loss"]
    return out


def cfg_score_fn(score_fn_v, v_conditionings, w_cfg):
    """Classifier-free guidance (notebook traceback ``vdm_model.py:318-327``: ``if self.w_cfg is None or self.training`` -> plain
    call; else ``assert "v_conditionings" in kwargs, "Need v_conditionings to mask out"``).  The blend itself is not in the
    reference tree; [INFERRED] the standard form  eps = (1 + w) * eps(v) - w * eps(masked v)  with masked v = zeros.
    score_fn_v(z, t_norm, v_list) -> eps_hat."""
    masked = [torch.zeros_like(v) for v in v_conditionings]

    def score(z, t_norm):
        return (1.0 + w_cfg) * score_fn_v(z, t_norm, v_conditionings) - w_cfg * score_fn_v(z, t_norm, masked)
    return score


def step_coeffs(sched, t, s):
    """Scalars of one ancestral step t -> s (D12).  c = -expm1(gamma_s-gamma_t)."""
    g_t, g_s = sched.gamma(t), sched.gamma(s)
    a_t, a_s = sched.alpha(g_t), sched.alpha(g_s)
    s_t, s_s = sched.sigma(g_t), sched.sigma(g_s)
    c = -torch.expm1(g_s - g_t)
    return dict(
        t_norm=(g_t - sched.gamma_min) / (sched.gamma_max - sched.gamma_min),
        alpha_t=a_t, alpha_s=a_s, sigma_t=s_t, sigma_s=s_s, c=c,
        ratio=a_s / a_t, scale=s_s * torch.sqrt(c),
    )


def sample_zs_given_zt(score_fn, sched, zt, t, s, noise, return_ddnm=False):
    k = step_coeffs(sched, t, s)
    B = zt.shape[0]
    t_norm = torch.full((B,), float(k["t_norm"]), dtype=zt.dtype)
    eps_hat = score_fn(zt, t_norm)
    f = lambda v: float(v)
    if return_ddnm:
        x0 = (zt - f(k["sigma_t"]) * eps_hat) / f(k["alpha_t"])
        return f(k["ratio"]) * (1.0 - f(k["c"])), f(k["alpha_s"]) * f(k["c"]), x0, f(k["scale"])
    mean = f(k["ratio"]) * (zt - f(k["c"]) * f(k["sigma_t"]) * eps_hat)
    return mean + f(k["scale"]) * noise


def sample_zt_given_zs(sched, zs, t, s, noise):
    g_t, g_s = sched.gamma(t), sched.gamma(s)
    a_ts = sched.alpha(g_t) / sched.alpha(g_s)
    var = torch.sigmoid(g_t) - a_ts ** 2 * torch.sigmoid(g_s)
    return float(a_ts) * zs + float(torch.sqrt(var)) * noise


def sample(score_fn, sched, z, n_sampling_steps, noises):
    """noises: sequence of n tensors (the eps' of each step), so the chain is reproducible."""
    steps = torch.linspace(1.0, 0.0, n_sampling_steps + 1).double()   # fp32 grid, as utils.py:286
    for i in range(n_sampling_steps):
        z = sample_zs_given_zt(score_fn, sched, z, steps[i], steps[i + 1], noises[i])
    return z
