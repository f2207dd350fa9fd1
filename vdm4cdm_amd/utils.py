"""Helpers the reference keeps in ``src/utils.py`` that sit on either side of the hot path.

* ``power`` / ``pk`` / ``get_ccs``: the acceptance metric (isotropic P(k), cross-correlation) -
  same estimator as /root/reference/src/utils.py:16-128, re-stated; runs on whatever device the field is on
  (rocFFT via ``torch.fft`` on the GPU).  Checked against golden vectors of the reference (tests/golden/pk_golden.npz).
* ``get_model`` / ``get_datamodule``: config -> model / data module with the reference's defaults
  (/root/reference/src/utils.py:401-475).
* ``get_ddnm_result``: the DDNM inpainting sampler (/root/reference/src/utils.py:277-304) on this package's VDM.
"""
import numpy as np
import torch

_SHELL_CACHE = {}


def _shells(size, device):
    """(kbin int64 [M], weights N int32 [M], |k| float32 [M]) for an rfftn grid of spatial `size`."""
    key = (tuple(size), str(device))
    hit = _SHELL_CACHE.get(key)
    if hit is not None:
        return hit
    rshape = tuple(size[:-1]) + (size[-1] // 2 + 1,)
    axes = []
    for ax, d in enumerate(rshape):
        j = torch.arange(d, dtype=torch.float32, device=device)
        if ax != len(rshape) - 1:
            j = torch.where(j > d // 2, j - d, j)            # signed frequencies on the full axes
        axes.append(j)
    grids = torch.meshgrid(*axes, indexing="ij")
    kmag = torch.sqrt(sum(g * g for g in grids)).flatten()
    w = torch.full(rshape, 2, dtype=torch.int32, device=device)  # Hermitian multiplicity of the half-spectrum
    w[..., 0] = 1
    if size[-1] % 2 == 0:
        w[..., -1] = 1
    out = (kmag.ceil().to(torch.int64), w.flatten(), kmag)
    _SHELL_CACHE[key] = out
    return out


def power(x, x2=None):
    """Shell-averaged (cross-)power of fields shaped (batch, channel, *spatial): mean over batch, sum over channels,
    integer-|k| shells by ceil, Hermitian-weighted, k=0 dropped, cut at the smallest Nyquist.  Returns (k, P, N)."""
    nd = x.dim() - 2
    size = tuple(x.shape[-nd:])
    kmax = min(size) // 2
    dims = tuple(range(-nd, 0))
    f1 = torch.fft.rfftn(x, s=size, dim=dims)
    f2 = f1 if x2 is None else torch.fft.rfftn(x2, s=size, dim=dims)
    spec = (f1 * f2.conj()).mean(dim=0).sum(dim=0).real.flatten()
    kbin, w, kmag = _shells(size, x.device)
    wf = w.to(spec.dtype)
    nb = int(kbin.max().item()) + 1
    ksum = torch.bincount(kbin, weights=kmag * wf, minlength=nb)
    psum = torch.bincount(kbin, weights=spec * wf, minlength=nb)
    nsum = torch.bincount(kbin, weights=wf, minlength=nb).round().to(torch.int32)
    sl = slice(1, 1 + kmax)
    n = nsum[sl]
    return ksum[sl] / n, psum[sl] / n, n


def pk(fields, fields2=None):
    """Per-sample spectra (summed over channels), stacked over the batch."""
    rows = [power(f[None], None if fields2 is None else fields2[i][None]) for i, f in enumerate(fields)]
    return tuple(torch.stack([r[j] for r in rows], dim=0) for j in range(3))


def get_ccs(fields1, fields2, full=False):
    """Cross-correlation coefficient P12 / sqrt(P11 P22), paired (default) or all pairs (full=True)."""
    ks, p11, _ = pk(fields1)
    p22 = pk(fields2)[1]
    if full:
        n = len(fields2)
        rows = [pk(f1[None].expand(n, *f1.shape), fields2)[1] for f1 in fields1]
        return ks, torch.stack(rows, dim=0) / torch.sqrt(p11[:, None] * p22[None, :])
    assert len(fields1) == len(fields2)
    return ks, pk(fields1, fields2)[1] / torch.sqrt(p11 * p22)


# ------------------------------------------------------------------------------------------------------------------
def get_datamodule(config):
    assert "data_params" in config, "data_params not in config"
    dp = config["data_params"]
    from . import data

    def return_func(fields, params):
        return {"conditioning": fields[0], "x": fields[1], "conditioning_values": [params]}

    return data.get_dataset(
        dataset_name=dp["dataset_name"], suite_name=dp.get("suite_name", "Astrid"), return_func=return_func,
        set_name=dp.get("set_name", "CV"), z_name=dp.get("z_name", "z_0.0"),
        channel_names=[config["in_field_name"], config["out_field_name"]], stage=dp.get("stage", "test"),
        batch_size=dp.get("batch_size", 1), cropsize=config["cropsize"], num_workers=8, mmap=False)


def get_model(config, backend="hip", precision=None, load_ckpt=True):
    """config dict (one entry of configs.yaml) -> LightVDM, defaults as in /root/reference/src/utils.py:434-462."""
    import os
    if config["type"] != "VDM":
        if config["type"] == "SFM":
            return None                                      # as in the reference (generate_3D.py refuses SFM)
        raise ValueError(f"Unknown model type {config['type']}")
    from . import networks, vdm_model
    n_values = config.get("conditioning_values", 6)
    cropsize = config.get("cropsize", 128)
    score_model = networks.CUNet(
        shape=(1, cropsize, cropsize, cropsize),
        chs=config.get("chs", [32, 64, 128, 256]),
        s_conditioning_channels=config.get("conditioning_channels", 1),
        v_conditioning_dims=[] if n_values == 0 else [n_values],
        t_conditioning=True, norm_groups=8, mid_attn=False, dropout_prob=0.1,
        conv_padding_mode="circular" if cropsize == 256 else "zeros", n_attention_heads=4,
        backend=backend, precision=precision or config.get("precision", "bf16"))
    vdm = vdm_model.LightVDM(score_model=score_model, draw_figure=None, gamma_max=13.3, learning_rate=3.0e-4)
    path = config.get("ckpt_path")
    if load_ckpt and path:
        if os.path.exists(path):
            vdm.load_state_dict(torch.load(path, map_location="cpu")["state_dict"])
        else:
            print(f"[vdm4cdm_amd] checkpoint {path} not found: using seeded random weights (no trained weights ship with this repo)")
    return vdm


def get_ddnm_result(vdm, y, A, AT, n_sampling_steps=250, l=10, return_all=False, verbose=0, **kwargs):
    """DDNM range/null-space sampler with time travel (length l), on this package's VDM."""
    if isinstance(l, int):
        l = np.full(n_sampling_steps, l)
    l = np.asarray(l)
    assert l.ndim == 1 and len(l) == n_sampling_steps and np.issubdtype(l.dtype, np.integer) and np.all(l >= 0), \
        "l must be a non-negative integer or an integer array of length n_sampling_steps"
    dev = vdm.device
    steps = torch.linspace(1.0, 0.0, n_sampling_steps + 1, device=dev)
    z = torch.randn((y.shape[0], *vdm.model.score_model.shape), device=dev)
    ATy = AT(y)
    xs = []
    x_r = None
    with torch.no_grad():
        for i in range(n_sampling_steps):
            L = int(min(l[i], i))
            z = vdm.model.sample_zt_given_zs(zs=z, t=steps[i - L], s=steps[i])          # travel back L steps
            for j in range(L, -1, -1):
                w_z, w_x, x0, scale = vdm.model.sample_zs_given_zt(zt=z, conditioning=None, t=steps[i - j], s=steps[i + 1 - j],
                                                                   return_ddnm=True, **kwargs)
                x_r = ATy + x0 - AT(A(x0))                                              # range-space correction
                z = w_z * z + w_x * x_r + scale * torch.randn_like(z)
            if return_all:
                xs.append(x_r)
            if verbose and i % 25 == 0:
                print(f"ddnm {i}/{n_sampling_steps}", flush=True)
    return torch.stack(xs, dim=0) if return_all else x_r
