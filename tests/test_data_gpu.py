"""GPU tests of the device data path (vdm_augment_batch through the C-ABI) against the fixtures of the reference's own augmentation
classes and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from test_data_cpu import GOLD, MK, _dataset

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# device log10f vs libm: one ulp of log10(rho + alpha) ~ 10 is 9.5e-7, and the normalisation subtracts ~10 and divides by 0.55
# (the device log10f is within ~3 ulp: 4 ulp -> 7e-6 absolute); positions are exact
TOL = dict(rtol=0, atol=7e-6)


@pytest.mark.parametrize("case", [c for c in MK.CASES if c[7] == 3], ids=[c[0] for c in MK.CASES if c[7] == 3])
def test_augment_kernel_matches_reference_golden(case):
    from vdm4cdm_amd import hip_ops as ops
    name, seed, S, D, n, sim, icrop, ndim, train = case
    fields = [torch.from_numpy(f[:, 0]).to(DEV) for f in MK.raw_fields(seed, n, S, ndim)]
    consts = list(zip(MK.ALPHAS, MK.MEANS, MK.STDS))
    out = ops.augment_batch(fields, consts, [(sim, GOLD[f"{name}/anchor"], GOLD[f"{name}/flips"], GOLD[f"{name}/perm"])], D)
    for c, o in enumerate(out):
        assert o.shape == (1, 1, D, D, D)
        np.testing.assert_allclose(o[0].cpu().numpy(), GOLD[f"{name}/out{c}"], **TOL)


def test_augment_kernel_matches_oracle_ragged_multi_sample():
    """Crop sizes that are not multiples of the 16^3 tile, every permutation x flip combination, anchors beyond the box, more samples
    than one launch carries (> 32)."""
    import itertools
    from oracle import augment_oracle as ao
    from vdm4cdm_amd import hip_ops as ops
    S, D, n = 40, 24, 3
    raw = MK.raw_fields(77, n, S, 3)
    fields = [torch.from_numpy(f[:, 0]).to(DEV) for f in raw]
    consts = list(zip(MK.ALPHAS, MK.MEANS, MK.STDS))
    g = np.random.default_rng(0)
    samples = []
    for perm in itertools.permutations(range(3)):
        for fl in itertools.product((0, 1), repeat=3):
            samples.append((int(g.integers(n)), g.integers(0, 2 * S, 3).tolist(), list(fl), list(perm)))
    assert len(samples) == 48
    out = ops.augment_batch(fields, consts, samples, D)
    for b, (sim, anchor, fl, perm) in enumerate(samples):
        ref = ao.augment_sample([f[sim] for f in raw], anchor, D, fl, perm, MK.ALPHAS, MK.MEANS, MK.STDS)
        for c in range(2):
            np.testing.assert_allclose(out[c][b].cpu().numpy(), ref[c], **TOL)
    for D2 in (7, 16, 33):                                    # ragged tiles on every side
        o = ops.augment_batch(fields, consts, samples[5:7], D2)
        for b, (sim, anchor, fl, perm) in enumerate(samples[5:7]):
            ref = ao.augment_sample([f[sim] for f in raw], anchor, D2, fl, perm, MK.ALPHAS, MK.MEANS, MK.STDS)
            np.testing.assert_allclose(o[1][b].cpu().numpy(), ref[1], **TOL)


def test_augment_full_size_properties():
    """BASELINE size (128^3 crops of a 128^3 box, batch 2): the augmentation is a bijection of the periodic box - undoing the flips,
    the permutation and the shift with torch ops on the device gives back the un-augmented normalised cube bit for bit."""
    from vdm4cdm_amd import hip_ops as ops
    S = D = 128
    g = torch.Generator().manual_seed(9)
    raw = torch.exp(torch.randn(2, S, S, S, generator=g) * 2 + 20).to(DEV)
    consts = [(1.0, 10.019186475678042, 0.5520203178284999)]
    plain = ops.augment_batch([raw], consts, [(0, [0, 0, 0], [0, 0, 0], [0, 1, 2]), (1, [0, 0, 0], [0, 0, 0], [0, 1, 2])], D)[0]
    anchor, fl, perm = [37, 120, 5], [1, 0, 1], [2, 0, 1]
    aug = ops.augment_batch([raw], consts, [(0, anchor, fl, perm), (1, anchor, fl, perm)], D)[0]
    inv = [perm.index(k) for k in range(3)]                   # out = flipped.permute(perm)  ->  flipped = out.permute(inv)
    und = aug[:, 0].permute(0, *[1 + i for i in inv])
    und = torch.flip(und, [1 + d for d in range(3) if fl[d]])
    und = torch.roll(und, shifts=anchor, dims=(1, 2, 3))
    assert torch.equal(und, plain[:, 0])
    ref = (torch.log10(raw + 1.0) - consts[0][1]) / consts[0][2]
    assert (plain[:, 0] - ref).abs().max().item() <= 7e-6


def test_astro_datamodule_batches_on_the_gpu(tmp_path):
    """get_dataset -> train / test loaders: the batch dict of the reference's return_func + collate_fn, built on the device, equals
    the oracle's per-sample pipeline for the choices the module drew."""
    from oracle import augment_oracle as ao
    from vdm4cdm_amd import data
    root = _dataset(tmp_path, n_sims=4, S=32)

    def return_func(fields, params):
        return {"conditioning": fields[0], "x": fields[1], "conditioning_values": [params]}

    dm = data.get_dataset(dataset_name="CMD_128", return_func=return_func, channel_names=["Mstar", "Mcdm"], stage="fit", batch_size=3,
                          cropsize=16, data_root=root, seed=2, device=DEV)
    drawn = []
    orig = dm.draw_sample
    dm.draw_sample = lambda idx, train, gen=None: drawn.append(orig(idx, train, gen)) or drawn[-1]
    batch = next(iter(dm.train_dataloader()))
    assert batch["x"].shape == (3, 1, 16, 16, 16) and batch["conditioning"].shape == (3, 1, 16, 16, 16) and batch["x"].is_cuda
    assert isinstance(batch["conditioning_values"], list) and batch["conditioning_values"][0].shape == (3, 6)
    raw = [np.asarray(f)[:, None] for f in dm.fields]
    for b, (sim, anchor, fl, perm) in enumerate(drawn):
        ref = ao.augment_sample([f[sim] for f in raw], anchor, 16, fl, perm, dm.alphas, dm.means, dm.stds)
        np.testing.assert_allclose(batch["conditioning"][b].cpu().numpy(), ref[0], **TOL)
        np.testing.assert_allclose(batch["x"][b].cpu().numpy(), ref[1], **TOL)
        np.testing.assert_allclose(batch["conditioning_values"][0][b].cpu().numpy(), dm.params[sim], rtol=1e-6)
    n_batches = 1 + sum(1 for _ in dm.train_dataloader())
    assert n_batches >= len(dm.train_idx) // 3
    # default return_func (CAMELS_3D_dataset.py:217-219): channels concatenated, params as a tensor
    dm2 = data.get_dataset(dataset_name="CMD_128", channel_names=["Mcdm"], stage="test", batch_size=2, cropsize=32, data_root=root, device=DEV)
    b2 = next(iter(dm2.test_dataloader()))
    assert b2["x"].shape == (2, 1, 32, 32, 32) and b2["conditioning"] is None and b2["conditioning_values"].shape == (2, 6)
    ref = ao.augment_sample([raw[1][0]], [0, 0, 0], 32, [0, 0, 0], [0, 1, 2], dm2.alphas, dm2.means, dm2.stds)
    np.testing.assert_allclose(b2["x"][0].cpu().numpy(), ref[0], **TOL)
