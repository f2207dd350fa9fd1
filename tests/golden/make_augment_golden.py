"""Generate golden vectors for the data/augmentation path of the reference.

Runs ONLY in the build container (needs /root/reference).  Imports the reference's own
``src/dataset/augmentation.py`` and drives its ``Crop`` (periodic crop + random anchor shift), ``LogTransform``, ``Normalize``,
``Flip`` and ``Permutate`` classes in the order ``AstroDataset.__getitem__`` / ``AstroDataModule`` compose them
(/root/reference/src/dataset/CAMELS_3D_dataset.py:53-73,107-114: crop -> float32 tensor -> [log10(x + alpha), normalize] -> flip ->
permute).  ``torchvision`` (an un-vendored third-party dependency of that file, not installed here) is stubbed with the two symbols
the file touches: ``transforms.functional.normalize(img, mean, std)`` - torchvision's published definition for a float tensor is
``(img - mean) / std`` - and ``transforms.Resize`` (imported by the file, never called).  ``CAMELS_3D_dataset.py`` itself cannot be
imported (it opens absolute cluster paths at import time and needs ``lightning``): the index arithmetic of ``__getitem__``
(``divmod(idx, ncrops)``) and the CV-set exclusion are restated in oracle/augment_oracle.py and tested there.

The random choices (anchor shift, flip axes, permutation) are drawn by the reference code itself from torch's global generator
after ``torch.manual_seed(case seed)``; the fixture records what it drew together with the outputs.  Only data is written:
tests/golden/augment_golden.npz.

    python tests/golden/make_augment_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF_FILE = "/root/reference/src/dataset/augmentation.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "augment_golden.npz")

ALPHAS = [1.0, 1.0]                                       # alphas_3d.json: Mstar, Mcdm
MEANS = [0.010429391444558287, 10.019186475678042]        # normalizations_3d.json: Mstar_m, Mcdm_m
STDS = [0.3219291117577123, 0.5520203178284999]           # Mstar_s, Mcdm_s

CASES = [  # name, seed, fullsize S, crop D, n sims, sim index, icrop, ndim, train (shift / flip / permute on)
    ("s12_c8_a", 1, 12, 8, 3, 0, 0, 3, True),
    ("s12_c8_b", 2, 12, 8, 3, 2, 5, 3, True),
    ("s16_c16", 3, 16, 16, 2, 1, 0, 3, True),             # crop == full size: one anchor, shift wraps the whole box
    ("s24_c8", 4, 24, 8, 2, 1, 26, 3, True),              # last anchor (2, 2, 2) * 8
    ("s12_c6_test", 5, 12, 6, 2, 0, 3, 3, False),         # stage "test": no shift, no flip, no permutation
    ("s32_c16", 6, 32, 16, 2, 1, 7, 3, True),
    ("s20_c10_2d", 7, 20, 10, 3, 2, 1, 2, True),          # the 2D dataset uses the same classes with ndim = 2
]


def raw_fields(seed, n, S, ndim):
    """Two positive fields per simulation: a sparse stellar-mass-like one (exact zeros) and a lognormal matter-like one."""
    g = torch.Generator().manual_seed(1000 + seed)
    shape = (n, 1) + (S,) * ndim
    a = torch.randn(shape, generator=g)
    star = torch.relu(a - 0.5) * 3.0e1
    cdm = torch.exp(2.0 * torch.randn(shape, generator=g) + 22.0)
    return [star.numpy().astype(np.float32), cdm.numpy().astype(np.float32)]


def load_reference():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.normalize = lambda img, mean, std: (img - mean) / std
    tvt.functional = tvf
    tvt.Resize = object
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    spec = importlib.util.spec_from_file_location("ref_augmentation", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    aug = load_reference()
    out = {}
    for name, seed, S, D, n, sim, icrop, ndim, train in CASES:
        fields = raw_fields(seed, n, S, ndim)
        crop = aug.Crop(ndim, D, 0, fullsize=S, do_augshift=train)
        grid_anchor = crop.anchors[icrop].copy()
        torch.manual_seed(seed)
        sample = crop([f[sim].copy() for f in fields], icrop)                       # AstroDataset.__getitem__ : crop first
        anchor = crop.anchors[icrop].copy()                                        # (the reference shifts its anchor table in place)
        sample = [torch.from_numpy(np.ascontiguousarray(f)).to(torch.float32) for f in sample]
        sample = aug.Normalize(means=MEANS, stds=STDS)(aug.LogTransform(ALPHAS)(sample))
        flips = np.zeros(ndim, dtype=np.int64)
        perm = np.arange(ndim, dtype=np.int64)
        if train:
            fl, pm = aug.Flip(ndim=ndim), aug.Permutate(ndim=ndim)
            sample = pm(fl(sample))
            flips[fl.axes.numpy()] = 1
            perm = pm.axes.numpy().astype(np.int64)
        out[f"{name}/meta"] = np.array([seed, S, D, n, sim, icrop, ndim, int(train)], dtype=np.int64)
        out[f"{name}/ncrops"] = np.array([crop.ncrops], dtype=np.int64)
        out[f"{name}/grid_anchor"] = grid_anchor.astype(np.int64)
        out[f"{name}/anchor"] = anchor.astype(np.int64)
        out[f"{name}/flips"] = flips
        out[f"{name}/perm"] = perm
        for c, f in enumerate(sample):
            out[f"{name}/out{c}"] = f.contiguous().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("out1")})


if __name__ == "__main__":
    main()
