"""Oracle: the data / augmentation path of the reference in numpy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED: checked against tests/golden/augment_golden.npz, which
tests/golden/make_augment_golden.py produced by running the reference's own ``Crop`` / ``LogTransform`` / ``Normalize`` /
``Flip`` / ``Permutate`` classes (/root/reference/src/dataset/augmentation.py).

Restates, per sample (one simulation cube per channel, shape [1, S, S, S] or [1, S, S]):
* ``Crop.__call__``            augmentation.py:107-127  periodic window ``(anchor + arange(crop)) % fullsize`` per axis (pad = 0)
* ``Crop.__init__`` anchors    augmentation.py:97-103   ``mgrid[0:S:crop]`` flattened row-major -> ncrops anchors
* ``LogTransform`` / ``Normalize`` augmentation.py:8-41 ``(log10(x + alpha) - mean) / std`` in float32
* ``Flip.__call__``            augmentation.py:48-60    ``torch.flip(img, 1 + axes)``
* ``Permutate.__call__``       augmentation.py:69-80    ``img.permute([0] + (1 + perm))``
* ``AstroDataset.__getitem__`` CAMELS_3D_dataset.py:53-73  ``bidx, icrop = divmod(idx, ncrops)``; crop -> float32 -> transforms
* CV exclusion                 CAMELS_3D_dataset.py:112-117,124-129  simulations 2, 8, 17 of the CV set are dropped
* train/val split sizes        CAMELS_3D_dataset.py:134-137  ``int(len * 0.95)`` / remainder
"""
import numpy as np


def crop_anchors(fullsize, crop, ndim):
    """Row-major grid of window origins (augmentation.py:97-103)."""
    axes = [np.arange(0, fullsize, crop) for _ in range(ndim)]
    return np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, ndim)


def split_index(idx, ncrops):
    """Dataset index -> (simulation, crop) (CAMELS_3D_dataset.py:55)."""
    return divmod(int(idx), int(ncrops))


def cv_keep_mask(n):
    """CV set: simulations 2, 8 and 17 are excluded (CAMELS_3D_dataset.py:112-117)."""
    keep = np.ones(n, dtype=bool)
    keep[[i for i in (2, 8, 17) if i < n]] = False
    return keep


def split_sizes(n_items):
    """(train, valid) sizes of the "fit" stage (CAMELS_3D_dataset.py:134-136)."""
    n_train = int(n_items * 0.95)
    return n_train, n_items - n_train


def augment_sample(fields, anchor, crop, flips, perm, alphas, means, stds):
    """fields: list of [1, S, ...] raw arrays (one per channel) of ONE simulation.  Returns the list of float32 arrays the
    reference's pipeline hands to ``return_func``: periodic crop at `anchor`, log-normalise, flip the axes with flips[d] != 0,
    permute the spatial axes by `perm`."""
    ndim = len(anchor)
    out = []
    for f, alpha, mean, std in zip(fields, alphas, means, stds):
        S = f.shape[-1]
        ind = [slice(None)]
        for d in range(ndim):
            i = (int(anchor[d]) + np.arange(crop)) % S
            ind.append(i.reshape((-1,) + (1,) * (ndim - d - 1)))
        x = f[tuple(ind)].astype(np.float32)
        x = ((np.log10(x + np.float32(alpha)) - np.float32(mean)) / np.float32(std)).astype(np.float32)
        ax = [1 + d for d in range(ndim) if flips[d]]
        if ax:
            x = np.flip(x, ax)
        x = np.transpose(x, [0] + [1 + int(p) for p in perm])
        out.append(np.ascontiguousarray(x))
    return out
