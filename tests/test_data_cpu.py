"""CPU tests of the data / augmentation path: the numpy oracle against the fixtures produced by the
reference's own augmentation classes, and the host logic of the file-backed AstroDataModule."""
import importlib.util
import os
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mk():
    spec = importlib.util.spec_from_file_location("mk_aug", os.path.join(ROOT, "tests", "golden", "make_augment_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


MK = _mk()                                                   # CASES / raw_fields / constants only: never touches /root/reference here
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "augment_golden.npz"))


@pytest.mark.parametrize("case", MK.CASES, ids=[c[0] for c in MK.CASES])
def test_augment_oracle_matches_reference_golden(case):
    """oracle/augment_oracle.augment_sample vs the outputs of the reference's Crop -> LogTransform -> Normalize -> Flip -> Permutate
    (2D and 3D, shifted anchors that wrap, crop == full size, the un-augmented "test" stage): positions exact, values to 1 ulp."""
    from oracle import augment_oracle as ao
    name, seed, S, D, n, sim, icrop, ndim, train = case
    fields = MK.raw_fields(seed, n, S, ndim)
    anchors = ao.crop_anchors(S, D, ndim)
    assert len(anchors) == int(GOLD[f"{name}/ncrops"][0])
    assert np.array_equal(anchors[icrop], GOLD[f"{name}/grid_anchor"])
    out = ao.augment_sample([f[sim] for f in fields], GOLD[f"{name}/anchor"], D, GOLD[f"{name}/flips"], GOLD[f"{name}/perm"],
                            MK.ALPHAS, MK.MEANS, MK.STDS)
    for c, o in enumerate(out):
        ref = GOLD[f"{name}/out{c}"]
        assert o.shape == ref.shape and o.dtype == np.float32
        # one ulp of log10(rho + alpha) ~ 10 is 9.5e-7; the normalisation subtracts ~10 and divides by 0.55: 2 ulp -> 3.5e-6 absolute
        np.testing.assert_allclose(o, ref, rtol=0, atol=4e-6)
    if train:
        assert np.all(GOLD[f"{name}/anchor"] - GOLD[f"{name}/grid_anchor"] < D) and np.all(GOLD[f"{name}/anchor"] >= GOLD[f"{name}/grid_anchor"])


def test_augment_oracle_index_helpers():
    from oracle import augment_oracle as ao
    assert ao.split_index(17, 8) == (2, 1)
    keep = ao.cv_keep_mask(27)
    assert keep.sum() == 24 and not keep[[2, 8, 17]].any()
    assert ao.split_sizes(8000) == (7600, 400)
    assert ao.split_sizes(27) == (25, 2)


def _dataset(tmp_path, set_name="LH", n_sims=5, S=16):
    from vdm4cdm_amd import data
    return data.write_synthetic_camels(str(tmp_path), dataset_name="CMD_128", set_name=set_name, n_sims=n_sims, fullsize=S, seed=3)


def test_astro_datamodule_host_logic(tmp_path):
    """Reference constructor / get_dataset surface (CAMELS_3D_dataset.py:76-141,202-234) on a synthetic on-disk data set in the
    reference's file layout: crop grid, 95/5 split, divmod indexing, per-rank shards, CV exclusion, loud synthetic fallback."""
    from oracle import augment_oracle as ao
    from vdm4cdm_amd import data
    root = _dataset(tmp_path, n_sims=5, S=16)

    def return_func(fields, params):
        return {"conditioning": fields[0], "x": fields[1], "conditioning_values": [params]}

    dm = data.get_dataset(dataset_name="CMD_128", suite_name="Astrid", return_func=return_func, set_name="LH", z_name="z_0.0",
                          channel_names=["Mstar", "Mcdm"], stage="fit", batch_size=2, cropsize=8, num_workers=16, mmap=True,
                          data_root=root, seed=1)
    assert isinstance(dm, data.AstroDataModule) and dm.fullsize == 16 and dm.crop == 8
    assert np.array_equal(dm.anchors, ao.crop_anchors(16, 8, 3)) and dm.ncrops == 8 and dm.nsamples == 40
    assert (len(dm.train_idx), len(dm.valid_idx)) == ao.split_sizes(40)
    assert sorted(dm.train_idx + dm.valid_idx) == list(range(40))
    sim, anchor, flips, perm = dm.draw_sample(21, train=True)
    assert sim == ao.split_index(21, 8)[0]
    ga = dm.anchors[ao.split_index(21, 8)[1]]
    assert all(ga[d] <= anchor[d] < ga[d] + 8 for d in range(3)) and sorted(perm) == [0, 1, 2] and set(flips) <= {0, 1}
    assert dm.draw_sample(21, train=False) == (2, ga.tolist(), [0, 0, 0], [0, 1, 2])
    # normalisation pair round trip (CAMELS_3D_dataset.py:146-156)
    x = torch.rand(4, 4) * 3 - 1
    assert torch.allclose(dm.norm_func(dm.unnorm_func(x, 1), 1), x, atol=1e-4)
    # batches are built by a HIP kernel: no silent CPU path
    dm.device = "cpu"
    with pytest.raises(RuntimeError, match="HIP kernel"):
        next(iter(dm.test_dataloader() if dm.stage == "test" else dm.val_dataloader()))
    # CV: simulations 2, 8, 17 dropped from fields and parameters
    root_cv = _dataset(tmp_path / "cv", set_name="CV", n_sims=20, S=8)
    cv = data.get_dataset(dataset_name="CMD_128", set_name="CV", stage="test", batch_size=1, cropsize=8, data_root=root_cv)
    raw = np.load(data.field_path(root_cv, "CMD_128", "Astrid", "CV", "z_0.0", "Mcdm"))
    assert len(cv.fields[0]) == 17 and len(cv.params) == 17 and cv.nsamples == 17
    assert np.array_equal(np.asarray(cv.fields[1][2]), raw[3]) and np.array_equal(np.asarray(cv.fields[1][16]), raw[19])
    # no data root: the synthetic module, with a warning
    os.environ.pop(data.DATA_ROOT_ENV, None)
    with pytest.warns(UserWarning, match="SYNTHETIC"):
        syn = data.get_dataset(dataset_name="CMD_128", cropsize=16, batch_size=1)
    assert isinstance(syn, data.SyntheticAstroDataModule)


def test_astro_datamodule_rank_shards_are_disjoint_and_complete(tmp_path):
    """Data parallelism: rank r of `world` walks items r, r + world, ... of the same shuffled epoch (same seed on every rank)."""
    from vdm4cdm_amd import data
    root = _dataset(tmp_path, n_sims=6, S=8)
    seen = []
    for rank in range(2):
        dm = data.get_dataset(dataset_name="CMD_128", stage="fit", batch_size=2, cropsize=8, data_root=root, seed=5)
        picked = []
        real_draw = dm.draw_sample
        dm.draw_sample = lambda idx, train, picked=picked, real_draw=real_draw: picked.append(idx) or real_draw(idx, train)
        dm.make_batch = lambda samples: {}                    # (record instead of launching)
        epochs = []
        for _ in range(3):                                    # the shards must stay disjoint epoch after epoch
            del picked[:]
            for _ in dm.train_dataloader(rank, 2):
                pass
            epochs.append(list(picked))
        seen.append(epochs)
    for e in range(3):
        assert not set(seen[0][e]) & set(seen[1][e]), f"epoch {e}: ranks overlap"
        assert sorted(seen[0][e] + seen[1][e]) == sorted(dm.train_idx)
    assert seen[0][0] != seen[0][1], "epochs are reshuffled"
