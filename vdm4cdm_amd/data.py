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

``AstroDataModule`` is the file-backed module with the reference's constructor and loader surface, re-designed for the GPU
(SURVEY.md section 8f rank 3): the raw ``.npy`` cube stacks are uploaded to HBM ONCE (the 128^3 LH set is 8 GB per field, the
256^3 set 67 GB - of 288 GB), and every batch is produced by ONE HIP launch (``vdm_augment_batch``: periodic crop at a shifted
anchor, log10 + normalisation, flips, axis permutation) instead of 16 CPU DataLoader workers.  Only the integer choices (which
simulation / crop, shift, flips, permutation) are drawn on the host.
"""
import math
import os
import warnings

import numpy as np
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

    def val_dataloader(self, rank=0, world=1):
        return self._loader(self.seed + 500000, self.n_val, self.batch_size, rank, world)

    def test_dataloader(self, rank=0, world=1):
        return self._loader(self.seed + 900000, self.n_test, self.batch_size, rank, world)


# field -> alpha of log10(rho + alpha)  [/root/reference/src/dataset/alphas_3d.json]; (mean, std) [normalizations_3d.json]
ALPHAS = {"Mcdm": 1.0, "Mstar": 1.0, "B": 1.0, "HI": 1.0, "Mgas": 1.0, "MgFe": 1.0, "ne": 1.0, "P": 1.0, "T": 1.0, "Z": 1.0,
          "Go7": 2.0, "Go8": 2.0, "Go9": 2.0}
NORMALIZATIONS = {"Mcdm": (10.019186475678042, 0.5520203178284999), "Mstar": (0.010429391444558287, 0.3219291117577123),
                  "Go7": (0.0, 1.0), "Go8": (0.0, 1.0), "Go9": (0.0, 1.0)}
DATA_ROOT_ENV = "VDM4CDM_DATA_ROOT"


def grid_size(dataset_name):
    """"CMD" -> 256, "CMD_128" -> 128, ... (the key scheme of /root/reference/src/dataset/data_source_3d.json)."""
    return 256 if dataset_name == "CMD" else int(dataset_name.split("_")[1])


def field_path(root, dataset_name, suite_name, set_name, z_name, channel_name):
    """File layout of the CAMELS grids below the cluster directory the reference reads
    (/root/reference/src/dataset/data_source_3d.json: .../Camels/3D_grids_new/Grids_Mcdm_Astrid_LH_256_z=0.0.npy,
    .../Camels/3D_grids_128/Grids_Mstar_Astrid_CV_128_z=0.0.npy), relative to `root`."""
    S = grid_size(dataset_name)
    sub = "3D_grids_new" if S == 256 else f"3D_grids_{S}"
    z = z_name.split("_", 1)[1]
    return os.path.join(root, sub, f"Grids_{channel_name}_{suite_name}_{set_name}_{S}_z={z}.npy")


def params_path(root, suite_name, set_name):
    """.../Camels/params_new/params_{set}_{suite}.txt (/root/reference/src/dataset/CAMELS_3D_dataset.py:123)."""
    return os.path.join(root, "params_new", f"params_{set_name}_{suite_name}.txt")


def default_return_func(fields, params):
    """/root/reference/src/dataset/CAMELS_3D_dataset.py:217-219."""
    return {"x": torch.cat(fields, dim=0), "conditioning": None, "conditioning_values": params}


class AstroDataModule:
    """File-backed CAMELS module: constructor, ``norm_func`` / ``unnorm_func``, ``collate_fn`` and the three loaders of the
    reference's ``AstroDataModule`` (/root/reference/src/dataset/CAMELS_3D_dataset.py:76-198), with the per-sample work of
    ``AstroDataset.__getitem__`` (:53-73) done by one HIP launch per batch on `device`.

    Differences by design: no worker processes (``num_workers`` is accepted and ignored), batches are born on the GPU, and the
    random anchor shift is drawn fresh around the grid anchor for every sample (the reference adds it IN PLACE to its anchor table,
    augmentation.py:113-117, so its anchors random-walk over the epochs; both are uniform periodic translations)."""

    def __init__(self, selection, channel_names, return_func, stage="fit", batch_size=1, do_crop=False, cropsize=256, ndim=3,
                 num_workers=1, mmap=True, data_root=None, device=None, seed=0):
        assert stage in ["fit", "test"], f"stage {stage} not recognized"
        assert ndim == 3, "the device data path covers the 3D grids (the 2D maps belong to the CPU plumbing config C1)"
        self.ndim, self.selection, self.channel_names = ndim, selection, list(channel_names)
        self.stage, self.batch_size, self.do_crop, self.cropsize = stage, batch_size, do_crop, cropsize
        self.num_workers, self.mmap = num_workers, mmap
        self.return_func = return_func if return_func is not None else default_return_func
        self.device = device
        self.alphas = [ALPHAS[c] for c in self.channel_names]
        self.means = [NORMALIZATIONS[c][0] for c in self.channel_names]
        self.stds = [NORMALIZATIONS[c][1] for c in self.channel_names]
        root = data_root or os.environ.get(DATA_ROOT_ENV)
        assert root, f"AstroDataModule needs the CAMELS directory (data_root= or ${DATA_ROOT_ENV})"
        sel = selection
        cv = sel["set_name"] == "CV"
        self.fields = []
        for c in self.channel_names:
            # always memory-mapped, whatever `mmap` says (the reference's scripts pass mmap=False and hold the whole set - 67 GB per field at
            # 256^3 - in host RAM, once per rank): here the cubes live in HBM after _resident() uploaded them slab by slab, the host only
            # ever touches one slab
            f = np.load(field_path(root, sel["dataset_name"], sel["suite_name"], sel["set_name"], sel["z_name"], c), mmap_mode="r")
            self.fields.append(f[_cv_keep(len(f))] if cv else f)
        self.params = np.atleast_2d(np.loadtxt(params_path(root, sel["suite_name"], sel["set_name"]))).astype(np.float32)
        if cv:
            self.params = self.params[_cv_keep(len(self.params))]
        self.fullsize = int(self.fields[0].shape[-1])
        for f in self.fields:
            assert f.ndim == 4 and f.shape[1:] == (self.fullsize,) * 3 and len(f) == len(self.fields[0]), "field shapes disagree"
        assert len(self.params) == len(self.fields[0]), f"len(params)={len(self.params)} != len(fields)={len(self.fields[0])}"
        self.crop = cropsize if do_crop else self.fullsize
        ax = np.arange(0, self.fullsize, self.crop)
        self.anchors = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1).reshape(-1, 3)      # augmentation.py:97-103
        self.ncrops = len(self.anchors)
        self.nsamples = len(self.fields[0]) * self.ncrops
        # two generators: the epoch shuffles must stay IDENTICAL on every rank (disjoint shards epoch after epoch), the augmentation
        # draws are per rank
        self._seed = int(seed)
        self._gen = torch.Generator().manual_seed(self._seed)                      # split + epoch shuffles (same on all ranks)
        # shifts / flips / permutations: one generator per (training rank), seeded ONCE at its first use and never again, and a
        # separate one for the validation / test loaders - a validation pass must not touch (let alone re-seed) the training stream
        self._aug_gens = {}
        self._aug_gen = self._aug_generator("train", 0)
        if stage == "fit":                                       # random_split(data, [95 %, 5 %])  (CAMELS_3D_dataset.py:134-137)
            order = torch.randperm(self.nsamples, generator=self._gen).tolist()
            n_train = int(self.nsamples * 0.95)
            self.train_idx, self.valid_idx = order[:n_train], order[n_train:]
        else:
            self.test_idx = list(range(self.nsamples))
        self._dev_fields = None

    # -- normalisation pair (CAMELS_3D_dataset.py:146-156) --------------------------------------------
    def unnorm_func(self, field, i_channel):
        return 10 ** (field * self.stds[i_channel] + self.means[i_channel]) - self.alphas[i_channel]

    def norm_func(self, field, i_channel):
        return (torch.log10(field + self.alphas[i_channel]) - self.means[i_channel]) / self.stds[i_channel]

    def collate_fn(self, batch):
        """CAMELS_3D_dataset.py:158-171: tensors are stacked, lists of tensors are stacked entry by entry, None stays None."""
        out, b0 = {}, batch[0]
        for key in b0.keys():
            if b0[key] is None:
                out[key] = None
            elif isinstance(b0[key], torch.Tensor):
                out[key] = torch.stack([b[key] for b in batch], dim=0)
            elif isinstance(b0[key], list):
                out[key] = [torch.stack([b[key][i] for b in batch], dim=0) for i in range(len(b0[key]))]
            else:
                raise ValueError(f"Type of {key} not recognized")
        return out

    # -- batches on the device ------------------------------------------------------------------------
    def _resident(self):
        if self._dev_fields is None:
            dev = torch.device(self.device if self.device is not None else "cuda")
            if dev.type != "cuda":
                raise RuntimeError("AstroDataModule builds its batches with a HIP kernel: set .device to a GPU (there is no CPU path)")
            self._dev_fields = []
            for f in self.fields:                             # upload in slabs of simulations: the (memory-mapped) 256^3 sets are
                d = torch.empty(f.shape, dtype=torch.float32, device=dev)      # 67 GB per field - never a second full copy on the host
                step = max(1, (1 << 30) // max(1, f[0].nbytes))
                for i in range(0, len(f), step):
                    d[i:i + step].copy_(torch.from_numpy(np.ascontiguousarray(f[i:i + step], dtype=np.float32)))
                self._dev_fields.append(d)
            self._dev_params = torch.from_numpy(self.params).to(dev)
        return self._dev_fields

    def _aug_generator(self, kind, rank):
        key = (kind, int(rank))
        g = self._aug_gens.get(key)
        if g is None:
            g = self._aug_gens[key] = torch.Generator().manual_seed(self._seed + 1 + 7919 * int(rank) + (0 if kind == "train" else 104729))
        return g

    def draw_sample(self, idx, train, gen=None):
        """(sim, anchor, flips, perm) of dataset item `idx`: bidx, icrop = divmod(idx, ncrops) (CAMELS_3D_dataset.py:55); in the
        "fit" stage the anchor is shifted by randint(crop) per axis, flips ~ randint(2), perm ~ randperm (augmentation.py:48-49,
        69, 113-117), all from `gen` (default: this module's rank-0 training generator)."""
        gen = self._aug_gen if gen is None else gen
        sim, icrop = divmod(int(idx), self.ncrops)
        anchor = self.anchors[icrop].copy()
        flips, perm = [0, 0, 0], [0, 1, 2]
        if train:
            anchor = anchor + torch.randint(self.crop, (3,), generator=gen).numpy()
            flips = torch.randint(2, (3,), generator=gen).tolist()
            perm = torch.randperm(3, generator=gen).tolist()
        return sim, anchor.tolist(), flips, perm

    def make_batch(self, samples):
        """samples: list of (sim, anchor, flips, perm) -> the collated batch dict (one HIP launch for all samples and channels)."""
        from . import hip_ops as ops
        fields = self._resident()
        consts = list(zip(self.alphas, self.means, self.stds))
        outs = ops.augment_batch(fields, consts, samples, self.crop)
        items = [self.return_func(fields=[o[b] for o in outs], params=self._dev_params[s[0]]) for b, s in enumerate(samples)]
        return self.collate_fn(items)

    def shard(self, idx, rank, world):
        """This rank's items of one epoch order.  world == 1: all of them (the reference's single-process DataLoader, a short last
        batch included).  world > 1: the order is padded by wrap-around to a multiple of world * batch_size and dealt out strided, so
        every rank sees the SAME number of FULL batches (torch's DistributedSampler rule: repeat, never drop) - ranks cannot drift
        across epoch boundaries, and the global time stratification (vdm_model.stratified_times) always sees equal local batches."""
        idx = list(idx)
        if world <= 1 or not idx:
            return idx
        unit = world * self.batch_size
        total = -(-len(idx) // unit) * unit
        idx = (idx * (total // len(idx) + 1))[:total]
        return idx[rank::world]

    def _loader(self, indices, train, shuffle, rank=0, world=1, kind="train"):
        gen = self._aug_generator(kind, rank)                     # (seeded once per (kind, rank); never re-seeded)
        idx = list(indices)
        if shuffle:
            idx = [idx[i] for i in torch.randperm(len(idx), generator=self._gen).tolist()]
        if kind == "eval" and world > 1:
            # validation / test: every item exactly once over the ranks (strided, NO wrap-around padding: a duplicated item would be
            # over-weighted in val_loss and emitted twice by a sharded test pass); ranks may differ by one item - Trainer.validate
            # weights its one all-reduce by the samples each rank really saw
            idx = idx[rank::world]
        else:
            idx = self.shard(idx, rank, world)                   # data parallelism: equal-length strided shards of the epoch
        for b0 in range(0, len(idx), self.batch_size):
            yield self.make_batch([self.draw_sample(i, train, gen) for i in idx[b0:b0 + self.batch_size]])

    def train_dataloader(self, rank=0, world=1):
        return self._loader(self.train_idx, True, True, rank, world, "train")

    def val_dataloader(self, rank=0, world=1):
        return self._loader(self.valid_idx, True, False, rank, world, "eval")          # (the reference's valid split shares the "fit" transforms)

    def test_dataloader(self, rank=0, world=1):
        return self._loader(self.test_idx, False, False, rank, world, "eval")


def _cv_keep(n):
    """CV set: simulations 2, 8 and 17 are excluded (/root/reference/src/dataset/CAMELS_3D_dataset.py:112-117)."""
    keep = np.ones(n, dtype=bool)
    keep[[i for i in (2, 8, 17) if i < n]] = False
    return keep


def write_synthetic_camels(root, dataset_name="CMD_128", suite_name="Astrid", set_name="LH", z_name="z_0.0",
                           channel_names=("Mstar", "Mcdm"), n_sims=4, fullsize=None, seed=0):
    """Writes a small synthetic data set in the reference's on-disk layout (raw, un-normalised densities + the parameter table),
    so that the file-backed module and the entry scripts can be exercised without the CAMELS files.  `fullsize` overrides the grid
    size the dataset name implies (tests)."""
    S = fullsize or grid_size(dataset_name)
    g = torch.Generator().manual_seed(seed)
    x = gaussian_random_field((n_sims, 1, S, S, S), generator=g)[:, 0]
    for c in channel_names:
        m, s = NORMALIZATIONS[c]
        if c == "Mstar":
            rho = torch.relu(10 ** (1.2 * (x - 1.0)) - 1.0)           # sparse, with exact zeros
        else:
            rho = 10 ** (x * s + m) - ALPHAS[c]
        p = field_path(root, dataset_name, suite_name, set_name, z_name, c)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        np.save(p, rho.clamp(min=0).numpy().astype(np.float32))
    params = (PARAM_LO + (PARAM_HI - PARAM_LO) * torch.rand(n_sims, 6, generator=g)).numpy()
    os.makedirs(os.path.dirname(params_path(root, suite_name, set_name)), exist_ok=True)
    np.savetxt(params_path(root, suite_name, set_name), params)
    return root


def get_dataset(dataset_name="CMD_128", suite_name="Astrid", return_func=None, set_name="LH", z_name="z_0.0",
                channel_names=("Mstar", "Mcdm"), stage="fit", batch_size=2, cropsize=128, ndim=3, num_workers=0, mmap=False,
                data_root=None, **kw):
    """Signature of the reference's ``CAMELS_3D_dataset.get_dataset`` (/root/reference/src/dataset/CAMELS_3D_dataset.py:202-234).
    With the CAMELS directory given (``data_root=`` or $VDM4CDM_DATA_ROOT, files laid out as on the reference's cluster) this is the
    file-backed ``AstroDataModule``; otherwise - LOUDLY - the seeded synthetic module with the same batch contract."""
    root = data_root or os.environ.get(DATA_ROOT_ENV)
    if root:
        selection = {"dataset_name": dataset_name, "suite_name": suite_name, "set_name": set_name, "z_name": z_name}
        return AstroDataModule(selection=selection, channel_names=list(channel_names), return_func=return_func, stage=stage,
                               batch_size=batch_size, do_crop=cropsize != 256, cropsize=cropsize, ndim=ndim,
                               num_workers=num_workers, mmap=mmap, data_root=root, **kw)
    warnings.warn(f"${DATA_ROOT_ENV} is not set: {dataset_name}/{suite_name}/{set_name} is replaced by SYNTHETIC lognormal fields "
                  "(same batch contract, not CAMELS data)", stacklevel=2)
    n = {"LH": 1000, "CV": 27, "1P": 61}.get(set_name, 1000)
    if stage == "fit":
        n_train, n_val, n_test = int(0.95 * n), n - int(0.95 * n), 0
    else:
        n_train, n_val, n_test = 0, 0, max(n - 3, 1)          # CV sims 2/8/17 are excluded upstream
    return SyntheticAstroDataModule(cropsize=cropsize, batch_size=batch_size, dim=3, n_train=max(n_train, batch_size),
                                    n_val=max(n_val, batch_size), n_test=max(n_test, 1), channel_names=channel_names,
                                    return_func=return_func, **kw)
