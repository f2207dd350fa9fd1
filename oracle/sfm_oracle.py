"""Oracle: flow matching between paired fields (CPU torch, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: ``mltools.models.sfm_model`` is not in the reference tree and
``utils.get_model`` has no SFM branch (/root/reference/src/utils.py:472-473).  Anchored on the call sites only: constructor
``LightSFM(velocity_model=, draw_figure=, learning_rate=)`` and batch keys ``x0`` / ``x1`` / ``conditioning_values``
(/root/reference/trainSFM3D128_c_c_from_field_name_thick_lowbatch.py:71-72,112-127).  Spec D14 [INFERRED]: conditional flow
matching on the straight path (Lipman et al. 2023; I-CFM of Tong et al. 2023 for sigma > 0).
velocity_fn(x_t, t, x0) -> v.
"""
import torch


def sfm_loss(velocity_fn, x0, x1, times, eps=None, sigma=0.0):
    B = x1.shape[0]
    bc = (B,) + (1,) * (x1.dim() - 1)
    t = times.to(x1.dtype)
    xt = (1.0 - t).view(bc) * x0 + t.view(bc) * x1
    if sigma > 0.0:
        xt = xt + sigma * eps
    v = velocity_fn(xt, t, x0)
    return ((v - (x1 - x0)) ** 2).mean(), xt, v


def sfm_sample(velocity_fn, x0, n_sampling_steps):
    """Explicit Euler from x(0) = x0 to x(1) with n steps, network time t_i = i / n."""
    x = x0.clone()
    for i in range(n_sampling_steps):
        t = torch.full((x.shape[0],), i / n_sampling_steps, dtype=x.dtype)
        x = x + velocity_fn(x, t, x0) / n_sampling_steps
    return x
