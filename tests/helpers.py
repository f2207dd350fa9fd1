"""Shared test helpers: product <-> oracle parameter conversion, seeded synthetic inputs."""
import torch


def oracle_params(net, flat=None):
    """Product CUNet parameters -> oracle dict (torch conv layout [cout, cin, *k]; skip/skip2 concatenated)."""
    dim = net.dim
    out = {}
    views = {name: net.view(name, flat).detach().cpu().float() for name in net.spec.items}
    for name, v in views.items():
        if name.endswith("skip2.weight"):
            continue
        if name.endswith(".weight") and v.dim() == 3 and ("conv" in name or "skip" in name or ".down." in name or ".up." in name):
            if name.endswith("skip.weight") and name.replace("skip.weight", "skip2.weight") in views:
                v = torch.cat([v, views[name.replace("skip.weight", "skip2.weight")]], dim=2)
            taps, cout, cin = v.shape
            k = 3 if taps > 1 else 1
            v = v.view((k,) * dim + (cout, cin)).permute(dim, dim + 1, *range(dim)).contiguous()
        out[name] = v
    return out


def oracle_cfg(net):
    return {"chs": net.chs, "norm_groups": net.norm_groups, "padding_mode": net.conv_padding_mode}


def randomize(net, seed=0, zero_init_std=0.05):
    """Seeded init with the zero-init convs made non-trivial (so every layer shows up in the output)."""
    g = torch.Generator().manual_seed(seed)
    net.reset_parameters(generator=g, zero_init_std=zero_init_std)
    with torch.no_grad():
        for name in net.spec.items:
            if "norm" in name:
                v = net.view(name)
                v.add_(0.2 * torch.randn(v.shape, generator=g))
    return net


def grf(shape, seed, slope=-2.0):
    """Unit-variance Gaussian random field with P(k) ~ k^slope (SURVEY.md section 8d synthetic cube)."""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(shape, generator=g)
    dims = tuple(range(w.dim() - len(shape[2:]), w.dim())) if len(shape) > 3 else tuple(range(w.dim()))
    sp = shape[-3:] if len(shape) >= 3 else shape
    nd = len(sp)
    dims = tuple(range(w.dim() - nd, w.dim()))
    F = torch.fft.fftn(w, dim=dims)
    ks = torch.meshgrid(*[torch.fft.fftfreq(n) * n for n in sp], indexing="ij")
    k = torch.sqrt(sum(kk ** 2 for kk in ks))
    amp = torch.where(k > 0, k.clamp(min=1.0) ** (slope / 2.0), torch.zeros_like(k))
    x = torch.fft.ifftn(F * amp, dim=dims).real
    x = x - x.mean(dim=dims, keepdim=True)
    return (x / x.std(dim=dims, keepdim=True)).float()
