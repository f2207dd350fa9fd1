"""Synthetic CAMELS-like data module emitting the reference's batch-dict contract.

The reference's ``AstroDataModule`` (/root/reference/src/dataset/CAMELS_3D_dataset.py:76-198) reads 1000 x D^3 cubes
from a Harvard cluster path; that I/O layer is out of scope.  What the hot path consumes is
its batch dict, built by the scripts' ``return_func`` (/root/reference/trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:75-76):

    {"conditioning": (B,1,D,D,D), "x": (B,1,D,D,D), "conditioning_values": [(B,6)]}

and its ``norm_func`` / ``unnorm_func`` pair (CAMELS_3D_dataset.py:146-156): x = (log10(rho + alpha) - m) / s.
This module produces that dict from seeded synthetic fields: the target is a unit-variance Gaussian
random field with P(k) ~ k^-2 in normalised-log-density space (== a lognormal density cube), the conditioning
field is the standardised ``relu(g - 1)`` of the same field (sparse, stellar-mass-like), the six parameters are
uniform in the CAMELS ranges.
"""
import math

import torch

# field -> (mean, std, alpha) of log10(rho + alpha)   [/root/reference/src/dataset/normalizations_3d.json:2-5, alphas_3d.json:2-3]
FIELD_NORM = {"Mcdm": (10.019186, 0.552020, 1.0), "Mstar": (0.010429, 0.321929, 1.0)}
PARAM_LO = torch.tensor([0.1, 0.6, 0.25, 0.25, 0.5, 0.5])
PARAM_HI = torch.tensor([0.5, 1.0, 4.0, 4.0, 2.0, 2.0])


def gaussian_random_field(shape, generator=None, slope=-2.0, device="cpu"):
    """Unit-variance GRF with P(k) ~ k^slope over the trailing spatial dims of `shape` = (B, C, *spatial)."""
    nd = len(shape) - 2
    dims = tuple(range(2, 2 + nd))
    w = torch.randn(shape, generator=generator, device=device)
    Fw = torch.fft.fftn(w, dim=dims)
    ks = torch.meshgrid(*[torch.fft.fftfreq(n, device=device) * n for n in shape[2:]], indexing="ij")
    k = torch.sqrt(sum(kk ** 2 for kk in ks))
    amp = torch.where(k > 0, k.clamp(min=1.0) ** (slope / 2.0), torch.zeros_like(k))
    x = torch.fft.ifftn(Fw * amp, dim=dims).real
    x = x - x.mean(dim=dims, keepdim=True)
    return (x / x.std(dim=dims, keepdim=True)).float()


class SyntheticAstroDataModule:
    """Same surface as the reference DataModule as far as the hot path and the scripts touch it."""

    def __init__(self, cropsize=128, batch_size=2, dim=3, n_train=950, n_val=50, n_test=12, channel_names=("Mstar", "Mcdm"),
                 conditioning=True, n_params=6, seed=1000, device="cpu", return_func=None, pool=8):
        self.cropsize, self.batch_size, self.dim = cropsize, batch_size, dim
        self.n_train, self.n_val, self.n_test = n_train, n_val, n_test
        self.channel_names = list(channel_names)
        self.conditioning, self.n_params = conditioning, n_params
        self.seed, self.device = seed, device
        self.return_func = return_func
        self.pool = pool                      # number of distinct cached batches (generation is not the hot path)
        self._cache = {}

    # -- normalisation pair (CAMELS_3D_dataset.py:146-156) --------------------------------------------
    def _norm_consts(self, i_channel):
        return FIELD_NORM.get(self.channel_names[i_channel], (0.0, 1.0, 1.0))

    def norm_func(self, x, i_channel):
        m, s, a = self._norm_consts(i_channel)
        return (torch.log10(x + a) - m) / s

    def unnorm_func(self, x, i_channel):
        m, s, a = self._norm_consts(i_channel)
        return 10 ** (x * s + m) - a

    # -- batches -------------------------------------------------------------------------------------
    def _make_batch(self, seed, batch_size):
        g = torch.Generator().manual_seed(seed)
        shape = (batch_size, 1) + (self.cropsize,) * self.dim
        x = gaussian_random_field(shape, generator=g)
        params = PARAM_LO + (PARAM_HI - PARAM_LO) * torch.rand(batch_size, 6, generator=g)
        cond = None
        if self.conditioning:
            dims = tuple(range(2, 2 + self.dim))
            c = torch.relu(x - 1.0)
            c = c - c.mean(dim=dims, keepdim=True)
            cond = c / c.std(dim=dims, keepdim=True).clamp(min=1e-6)
        fields = [cond, x]
        if self.return_func is not None:
            batch = self.return_func(fields, params[:, :self.n_params])
        else:
            batch = {"conditioning": cond, "x": x, "conditioning_values": [params[:, :self.n_params]] if self.n_params else []}
        return batch

    def _loader(self, base_seed, n_items, batch_size, rank=0, world=1):
        n_batches = max(1, n_items // (batch_size * world))
        for b in range(n_batches):
            key = (base_seed, (b * world + rank) % self.pool, batch_size)
            if key not in self._cache:
                self._cache[key] = self._make_batch(base_seed + key[1], batch_size)
            batch = self._cache[key]
            yield {k: (self._to(v)) for k, v in batch.items()}

    def _to(self, v):
        if v is None:
            return None
        if isinstance(v, (list, tuple)):
            return [a.to(self.device, non_blocking=True) for a in v]
        return v.to(self.device, non_blocking=True)

    def train_dataloader(self, rank=0, world=1):
        return self._loader(self.seed + 7919 * rank, self.n_train, self.batch_size, rank, world)

    def val_dataloader(self):
        return self._loader(self.seed + 500000, self.n_val, self.batch_size)

    def test_dataloader(self):
        return self._loader(self.seed + 900000, self.n_test, self.batch_size)


def get_dataset(dataset_name="CMD_128", suite_name="Astrid", return_func=None, set_name="LH", z_name="z_0.0",
                channel_names=("Mstar", "Mcdm"), stage="fit", batch_size=2, cropsize=128, num_workers=0, mmap=False, **kw):
    """Signature of the reference's ``CAMELS_3D_dataset.get_dataset`` (/root/reference/src/dataset/CAMELS_3D_dataset.py:202-234).
    The CAMELS files are not available here (absolute cluster paths, data_source_3d.json); a seeded synthetic module
    with the same batch contract is returned instead."""
    n = {"LH": 1000, "CV": 27, "1P": 61}.get(set_name, 1000)
    if stage == "fit":
        n_train, n_val, n_test = int(0.95 * n), n - int(0.95 * n), 0
    else:
        n_train, n_val, n_test = 0, 0, max(n - 3, 1)          # CV sims 2/8/17 are excluded upstream
    return SyntheticAstroDataModule(cropsize=cropsize, batch_size=batch_size, dim=3, n_train=max(n_train, batch_size),
                                    n_val=max(n_val, batch_size), n_test=max(n_test, 1), channel_names=channel_names,
                                    return_func=return_func, **kw)
