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


def _record_epochs(dm, rank, world, n_epochs, loader="train"):
    """Item indices and augmentation tuples a rank draws per epoch (make_batch replaced by a recorder: no GPU)."""
    picked = []
    real_draw = dm.draw_sample
    dm.draw_sample = lambda idx, train, gen=None: picked.append((idx, real_draw(idx, train, gen))) or picked[-1][1]
    batches = []
    dm.make_batch = lambda samples: batches.append(len(samples)) or {}
    epochs = []
    for _ in range(n_epochs):
        del picked[:], batches[:]
        it = dm.train_dataloader(rank, world) if loader == "train" else dm.val_dataloader(rank, world)
        for _ in it:
            pass
        epochs.append((list(picked), list(batches)))
    dm.draw_sample = real_draw
    return epochs


@pytest.mark.parametrize("world,n_sims,bs", [(2, 8, 2), (3, 7, 2), (8, 6, 2)], ids=["w2_even", "w3_ragged", "w8_tiny"])
def test_astro_datamodule_rank_shards_are_equal_and_complete(tmp_path, world, n_sims, bs):
    """Data parallelism: every rank walks a strided shard of the SAME shuffled epoch (same seed on every rank), all
    shards have the same number of FULL batches also when len(train) % (world * batch) != 0 (wrap-around padding: DistributedSampler
    semantics - repeat, never drop), and together they cover every training item; a rank is never left without a batch."""
    from vdm4cdm_amd import data
    root = _dataset(tmp_path, n_sims=n_sims, S=8)
    seen = []
    for rank in range(world):
        dm = data.get_dataset(dataset_name="CMD_128", stage="fit", batch_size=bs, cropsize=8, data_root=root, seed=5)
        seen.append(_record_epochs(dm, rank, world, 3))
    n_train = len(dm.train_idx)
    per_rank = -(-n_train // (world * bs)) * bs
    for e in range(3):
        items = [[i for i, _ in seen[r][e][0]] for r in range(world)]
        assert all(len(it) == per_rank for it in items), f"epoch {e}: unequal shards {[len(it) for it in items]}"
        assert all(seen[r][e][1] == [bs] * (per_rank // bs) for r in range(world)), "every batch is full, same count on every rank"
        flat = [i for it in items for i in it]
        assert set(flat) == set(dm.train_idx), f"epoch {e}: items missing"
        if n_train % (world * bs) == 0:
            assert len(set(flat)) == len(flat), f"epoch {e}: ranks overlap without padding"
    if n_train > 2:
        assert any([i for i, _ in seen[0][0][0]] != [i for i, _ in seen[0][e][0]] for e in (1, 2)), "epochs are reshuffled"
    # world = 1: the reference's single-process loader, nothing padded
    dm = data.get_dataset(dataset_name="CMD_128", stage="fit", batch_size=bs, cropsize=8, data_root=root, seed=5)
    ep = _record_epochs(dm, 0, 1, 1)[0]
    assert sorted(i for i, _ in ep[0]) == sorted(dm.train_idx)


def test_validation_never_touches_the_training_augmentation_stream(tmp_path):
    """Regression (advisor, round 2): on rank r > 0 a validation pass used to re-seed the augmentation generator with rank 0's seed,
    and the next training epoch re-seeded it again with the initial per-rank seed - the (shift, flip, permutation) draws of every
    rank > 0 replayed with the validation period.  Now: train -> val -> train on rank 1 draws exactly what train -> train draws, the
    second epoch does not replay the first, and rank 1's stream differs from rank 0's."""
    from vdm4cdm_amd import data
    root = _dataset(tmp_path, n_sims=8, S=16)
    mk = lambda: data.get_dataset(dataset_name="CMD_128", stage="fit", batch_size=2, cropsize=8, data_root=root, seed=5)
    augs = lambda ep: [a[1:] for _, a in ep[0]]                # (anchor, flips, perm) per drawn sample
    a = mk()
    plain = _record_epochs(a, 1, 2, 2)
    b = mk()
    first = _record_epochs(b, 1, 2, 1)[0]
    _record_epochs(b, 1, 2, 1, loader="val")
    _record_epochs(b, 0, 2, 1, loader="val")                   # (also the default-rank call of a user script)
    second = _record_epochs(b, 1, 2, 1)[0]
    assert augs(first) == augs(plain[0]) and augs(second) == augs(plain[1]), "validation disturbed the training augmentation stream"
    assert augs(plain[0]) != augs(plain[1]), "epoch 2 replays the augmentation draws of epoch 1"
    r0 = _record_epochs(mk(), 0, 2, 1)[0]
    assert augs(r0) != augs(plain[0]), "ranks share one augmentation stream"
