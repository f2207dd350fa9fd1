"""Entry points with the reference's command lines.

* ``trainVDM3D*_..._thick_lowbatch.py <field_in> <field_out> <cropsize>`` ->  ``train_vdm3d(variant, argv)``
  (hyper-parameters per variant: /root/reference/trainVDM3D{,128,160,192,224}_c_c_from_field_name_thick_lowbatch.py:57-73,
  trainVDM3D_c_uc_from_field_name_thick_lowbatch.py:57-73; trainer settings :38-49).
* ``train_uc_uc_from_field_name.py <field_name>`` (2D, learned-linear schedule, circular padding; BASELINE config C1 runs it on
  CPU PyTorch) -> ``train_uc_uc(argv)`` (/root/reference/train_uc_uc_from_field_name.py:50-120).
* ``trainSFM3D*_from_field_name*.py <field_in> <field_out> <cropsize>`` -> ``train_sfm3d(variant, argv)`` and the 2D
  ``trainSFM_c_uc_from_field_name.py <field_in> <field_out>`` (mid-level attention on) -> ``train_sfm_c_uc_2d(argv)``.
* ``generate_3D.py <model_name> <save_path> <runtype>`` -> ``generate_3d(argv)`` (/root/reference/generate_3D.py).
Environment knobs (build-side, all optional): VDM4CDM_MAX_STEPS, VDM4CDM_PRECISION (bf16|fp32), VDM4CDM_SAMPLING_STEPS,
VDM4CDM_LOG_DIR, VDM4CDM_CROPSIZE_2D / VDM4CDM_BATCH_2D (shrink the 2D plumbing config).
Multi-GPU: launch the same script under ``python -m torch.distributed.run --nproc-per-node N`` (one rank per GPU, RCCL).
"""
import argparse
import os
import sys

import numpy as np
import torch

# variant -> (dataset_name, chs, conditioning_values, val_check_interval, experiment-name pattern, slab thickness of the image)
VDM3D_VARIANTS = {
    "128": ("CMD_128", [32, 64, 128, 256], 6, 1000, "LH128_c_c_{i}_to_{o}_thick_lowbatch_{c}", 48),
    "160": ("CMD_160", [32, 64, 128, 256], 6, 5000, "LH160_c_c_{i}_to_{o}_thick_lowbatch_{c}", 48),
    "192": ("CMD_192", [32, 64, 128, 256], 6, 5000, "LH192_c_c_{i}_to_{o}_thick_lowbatch_{c}_re2", 48),
    "224": ("CMD_224", [16, 32, 64, 128], 6, 5000, "LH224_c_c_{i}_to_{o}_thick_lowbatch_{c}", 48),
    "256": ("CMD", [16, 32, 64, 128], 6, 5000, "LH_c_c_{i}_to_{o}_thick_lowbatch_{c}_re3", 32),
    "256_c_uc": ("CMD", [16, 32, 64, 128], 0, 5000, "LH_c_uc_{i}_to_{o}_thick_lowbatch_{c}_re2", 32),
}


def _seed_everything(seed=42):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    from .vdm_model import reset_train_generators
    reset_train_generators()                 # the training step's own generators restart from the new seed


def _figure_closure(dm, thickness, with_values):
    """draw_figure(batch, samples) as the scripts build it (images of a projected slab, P(k), cross-correlation)."""
    from . import figures, utils

    def to_np(t):
        return t.detach().cpu().numpy()

    def x_to_im(field):
        return to_np(dm.norm_func(dm.unnorm_func(field, 1)[0, :, :, :thickness].sum(-1), 1))

    def conditioning_to_im(field):
        return to_np(dm.norm_func(dm.unnorm_func(field, 0)[0, :, :, :thickness].sum(-1), 0))

    def pk_for_plot(field):
        ks, pks, _ = utils.pk(field[None, None] / field.sum())
        return to_np(ks[0]), to_np(pks[0])

    def cc_for_plot(f1, f2):
        ks, ccs = utils.get_ccs(f1[None, None] / f1.sum(), f2[None, None] / f2.sum(), full=False)
        return to_np(ks[0]), to_np(ccs[0])

    def draw_figure(batch, samples):
        return figures.draw_figure(
            batch, samples, x_to_im=x_to_im, conditioning_to_im=conditioning_to_im,
            conditioning_values_to_str=str if with_values else None,
            pk_func=lambda f, ic: pk_for_plot(dm.unnorm_func(f, ic)),
            cc_func=lambda f1, f2, ic: cc_for_plot(dm.unnorm_func(f1, ic), dm.unnorm_func(f2, ic)))

    return draw_figure


def train_vdm3d(variant, argv=None):
    from . import data, networks, vdm_model
    from .trainer import Trainer
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3:
        raise SystemExit("usage: <script> <field_in> <field_out> <cropsize>")
    field_in, field_out, cropsize = argv[0], argv[1], int(argv[2])
    dataset_name, chs, n_values, val_every, name_pat, thick = VDM3D_VARIANTS[variant]
    _seed_everything(42)
    batch_size = 2

    def return_func(fields, params):
        return {"conditioning": fields[0], "x": fields[1], "conditioning_values": [params] if n_values else None}

    dm = data.get_dataset(dataset_name=dataset_name, suite_name="Astrid", return_func=return_func, set_name="LH", z_name="z_0.0",
                          channel_names=[field_in, field_out], stage="fit", batch_size=batch_size, cropsize=cropsize,
                          num_workers=16, mmap=False)
    score_model = networks.CUNet(
        shape=(1, cropsize, cropsize, cropsize), chs=chs, s_conditioning_channels=1,
        v_conditioning_dims=[] if n_values == 0 else [n_values], t_conditioning=True, norm_groups=8, mid_attn=False,
        dropout_prob=0.1, conv_padding_mode="circular" if cropsize == 256 else "zeros", n_attention_heads=4,
        backend="hip", precision=os.environ.get("VDM4CDM_PRECISION", "bf16"))
    vdm = vdm_model.LightVDM(score_model=score_model, draw_figure=_figure_closure(dm, thick, n_values > 0), gamma_max=13.3,
                             learning_rate=3.0e-4)
    trainer = Trainer(max_steps=int(os.environ.get("VDM4CDM_MAX_STEPS", 1_000_000)), val_check_interval=val_every,
                      gradient_clip_val=0.5, every_n_train_steps=10_000,
                      default_root_dir=os.environ.get("VDM4CDM_LOG_DIR", "./data/logs/vdm4cdm-3D"),
                      experiment_name=name_pat.format(i=field_in, o=field_out, c=cropsize),
                      n_val_sampling_steps=int(os.environ.get("VDM4CDM_SAMPLING_STEPS", 250)))
    trainer.fit(model=vdm, datamodule=dm)
    return trainer


# variant -> (dataset_name, chs, conditioning_values, batch_size, experiment-name pattern, slab thickness of the image)
# [/root/reference/trainSFM3D{128,160,192,}_c_c_from_field_name_thick_lowbatch.py:36,59-68,74,89;
#  trainSFM3D_c_uc_from_field_name{,_thick,_thick_lowbatch}.py]
SFM3D_VARIANTS = {
    "128": ("CMD_128", [32, 64, 128, 256], 6, 4, "LH128_c_c_{i}_to_{o}_thick_lowbatch_{c}", 48),
    "160": ("CMD_160", [32, 64, 128, 256], 6, 4, "LH160_c_c_{i}_to_{o}_thick_lowbatch_{c}", 48),
    "192": ("CMD_192", [32, 64, 128, 256], 6, 2, "LH192_c_c_{i}_to_{o}_thick_lowbatch_{c}", 48),
    "256": ("CMD", [16, 32, 64, 128], 6, 2, "LH_c_c_{i}_to_{o}_thick_lowbatch_{c}", 32),
    "256_c_uc": ("CMD", [12, 36, 64, 128], 0, 2, "LH_c_uc_{i}_to_{o}", 32),
    "256_c_uc_thick": ("CMD", [16, 32, 64, 128], 0, 3, "LH_c_uc_{i}_to_{o}_thick", 32),
    "256_c_uc_thick_lowbatch": ("CMD", [16, 32, 64, 128], 0, 2, "LH_c_uc_{i}_to_{o}_thick_lowbatch_{c}", 32),
}


def _sfm_figure_closure(dm, thickness, with_values):
    """draw_figure_sfm(batch, samples) (/root/reference/src/utils.py:204-243): the same panels as the VDM figure with the source
    field x0 in the conditioning slot and x1 as the target."""
    vdm_fig = _figure_closure(dm, thickness, with_values)

    def draw_figure(batch, samples):
        return vdm_fig({"x": batch["x1"], "conditioning": batch["x0"], "conditioning_values": batch.get("conditioning_values")}, samples)

    return draw_figure


def train_sfm3d(variant, argv=None):
    """trainSFM3D*_from_field_name*.py <field_in> <field_out> <cropsize>: flow matching from field_in (x0) to field_out (x1)."""
    from . import data, networks, sfm_model
    from .trainer import Trainer
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3:
        raise SystemExit("usage: <script> <field_in> <field_out> <cropsize>")
    field_in, field_out, cropsize = argv[0], argv[1], int(argv[2])
    dataset_name, chs, n_values, batch_size, name_pat, thick = SFM3D_VARIANTS[variant]
    _seed_everything(42)

    def return_func(fields, params):
        return {"x0": fields[0], "x1": fields[1], "conditioning_values": [params] if n_values else None}

    dm = data.get_dataset(dataset_name=dataset_name, suite_name="Astrid", return_func=return_func, set_name="LH", z_name="z_0.0",
                          channel_names=[field_in, field_out], stage="fit", batch_size=batch_size, cropsize=cropsize,
                          num_workers=16, mmap=False)
    velocity_model = networks.CUNet(
        shape=(1, cropsize, cropsize, cropsize), chs=chs, s_conditioning_channels=1,
        v_conditioning_dims=[] if n_values == 0 else [n_values], t_conditioning=True, norm_groups=8, mid_attn=False,
        dropout_prob=0.1, conv_padding_mode="circular" if cropsize == 256 else "zeros", n_attention_heads=4,
        backend="hip", precision=os.environ.get("VDM4CDM_PRECISION", "bf16"))
    sfm = sfm_model.LightSFM(velocity_model=velocity_model, draw_figure=_sfm_figure_closure(dm, thick, n_values > 0), learning_rate=3.0e-4)
    trainer = Trainer(max_steps=int(os.environ.get("VDM4CDM_MAX_STEPS", 1_000_000)), val_check_interval=1000,
                      gradient_clip_val=0.5, every_n_train_steps=10_000,
                      default_root_dir=os.environ.get("VDM4CDM_LOG_DIR", "./data/logs/sfm4cdm-3D"),
                      experiment_name=name_pat.format(i=field_in, o=field_out, c=cropsize),
                      n_val_sampling_steps=int(os.environ.get("VDM4CDM_SAMPLING_STEPS", 100)))
    trainer.fit(model=sfm, datamodule=dm)
    return trainer


def train_sfm_c_uc_2d(argv=None):
    """trainSFM_c_uc_from_field_name.py <field_in> <field_out>: the 2D flow-matching script, the one place the reference switches
    the mid-level attention on (mid_attn=True, /root/reference/trainSFM_c_uc_from_field_name.py:58-67,104-119).  CPU plumbing like
    BASELINE config C1 (explicit backend='torch')."""
    from . import data, networks, sfm_model
    from .trainer import Trainer
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        raise SystemExit("usage: trainSFM_c_uc_from_field_name.py <field_in> <field_out>")
    field_in, field_out = argv
    _seed_everything(42)
    cropsize = int(os.environ.get("VDM4CDM_CROPSIZE_2D", 256))
    batch_size = int(os.environ.get("VDM4CDM_BATCH_2D", 12))
    dm = data.SyntheticAstroDataModule(cropsize=cropsize, batch_size=batch_size, dim=2, channel_names=[field_in, field_out], n_params=0,
                                       return_func=lambda fields, params: {"x0": fields[0], "x1": fields[1], "conditioning_values": None})
    velocity_model = networks.CUNet(shape=(1, cropsize, cropsize), chs=[48, 96, 192, 384], s_conditioning_channels=1,
                                    v_conditioning_dims=[], t_conditioning=True, norm_groups=8, mid_attn=True, dropout_prob=0.1,
                                    conv_padding_mode="circular", n_attention_heads=4, backend="torch")
    sfm = sfm_model.LightSFM(velocity_model=velocity_model, draw_figure=None)
    trainer = Trainer(max_steps=int(os.environ.get("VDM4CDM_MAX_STEPS", 1_000_000)), val_check_interval=1000, gradient_clip_val=0.5,
                      every_n_train_steps=10_000, default_root_dir=os.environ.get("VDM4CDM_LOG_DIR", "./data/logs/sfm4cdm-2D"),
                      experiment_name=f"LH_c_uc_{field_in}_to_{field_out}", device="cpu")
    trainer.fit(model=sfm, datamodule=dm)
    return trainer


def train_uc_uc(argv=None):
    """2D unconditional VDM.  Runs on CPU PyTorch via the explicit backend='torch' (BASELINE config C1)."""
    from . import data, networks, vdm_model
    from .trainer import Trainer
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit("usage: train_uc_uc_from_field_name.py <field_name>")
    field_name = argv[0]
    _seed_everything(42)
    cropsize = int(os.environ.get("VDM4CDM_CROPSIZE_2D", 256))
    batch_size = int(os.environ.get("VDM4CDM_BATCH_2D", 12))
    dm = data.SyntheticAstroDataModule(cropsize=cropsize, batch_size=batch_size, dim=2, channel_names=[field_name, field_name],
                                       conditioning=False, n_params=0,
                                       return_func=lambda fields, params: {"x": fields[1], "conditioning": None, "conditioning_values": None})
    score_model = networks.CUNet(shape=(1, cropsize, cropsize), chs=[48, 96, 192, 384], s_conditioning_channels=0,
                                 v_conditioning_dims=[], t_conditioning=True, norm_groups=8, dropout_prob=0.1,
                                 conv_padding_mode="circular", n_attention_heads=4, backend="torch")
    vdm = vdm_model.LightVDM(score_model=score_model, gamma_min=-13.3, gamma_max=13.3, noise_schedule="learned_linear",
                             draw_figure=None)
    trainer = Trainer(max_steps=int(os.environ.get("VDM4CDM_MAX_STEPS", 1_000_000)), val_check_interval=5000, gradient_clip_val=0.5,
                      every_n_train_steps=10_000, default_root_dir=os.environ.get("VDM4CDM_LOG_DIR", "./data/logs/vdm4cdm-2D"),
                      experiment_name=f"LH_uc_uc_{field_name}", device="cpu")
    trainer.fit(model=vdm, datamodule=dm)
    return trainer


GENERATE_WHITELIST = ["VDM_Go7_Mcdm_c_c_128", "VDM_Go8_Mcdm_c_c_128", "VDM_Go9_Mcdm_c_c_128", "VDM_Mstar_Mcdm_c_c_128",
                      "VDM_Mstar_Mcdm_c_c_160", "VDM_Mstar_Mcdm_c_c_192", "VDM_Mstar_Mcdm_c_c_224", "VDM_Mstar_Mcdm_c_c_256",
                      "VDM_Mstar_Mcdm_c_uc_256", "SFM_Mstar_Mcdm_c_c_128", "SFM_Mstar_Mcdm_c_c_256"]


def chain_seed(chain):
    """Seed of one sampling chain: a function of the GLOBAL chain id only, so a run gives the same cubes for any number of ranks."""
    return 1_000_003 * int(chain) + 17


def _sampling_setup(model_name, configs_path, set_name):
    """Shared by generate_3d / generate_3d_1p: process group (for the final barrier), device, model, data module."""
    import yaml
    from . import utils
    from .trainer import init_distributed
    backend = os.environ.get("VDM4CDM_BACKEND", "hip")
    use_cuda = torch.cuda.is_available() and backend == "hip"
    rank, local_rank, world = init_distributed("cuda" if use_cuda else "cpu")
    device = f"cuda:{local_rank}" if use_cuda else "cpu"
    if use_cuda:
        torch.cuda.set_device(local_rank)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    configs = yaml.safe_load(open(configs_path or os.path.join(root, "configs.yaml")))
    config = configs[model_name]
    model = utils.get_model(config, backend=backend)
    model.to(device)
    model.eval()
    config["data_params"].update(set_name=set_name, stage="test", batch_size=1)
    dm = utils.get_datamodule(config)
    dm.device = device
    return rank, world, device, config, model, dm


def _sample_repetitions(model, config, batch, device, rep, first_chain, rank, world, n_steps):
    """The `rep` independent chains of one conditioning cube that fall to this rank: [(repetition index, sample)]."""
    s_conditioning = batch["conditioning"].to(device)
    v_conditionings = [d.to(device) for d in batch["conditioning_values"]] if config.get("conditioning_values", 6) else []
    out = []
    for i in range(rep):
        chain = first_chain + i                              # global chain id: the unit that is dealt to the ranks
        if chain % world != rank:
            continue
        gen = model.draw_samples(batch_size=1, n_sampling_steps=n_steps, seed=chain_seed(chain), s_conditioning=s_conditioning,
                                 v_conditionings=v_conditionings, verbose=(rank == 0))
        out.append((i, gen.detach().cpu().numpy()))
    return out


def _save_or_shard(save_path, stem, parts, rank, world):
    if world == 1:
        np.save(os.path.join(save_path, f"{stem}.npy"), np.concatenate([g for _, g in parts], axis=0))
    elif parts:
        np.savez(os.path.join(save_path, f"{stem}_rank{rank}.npz"), gens=np.concatenate([g for _, g in parts], axis=0),
                 ids=np.array([i for i, _ in parts]))


def _merge_shards(save_path, stems, rep, rank, world):
    """Rank 0 assembles <stem>.npy exactly as a single process writes it (repetitions in order), then removes the shards."""
    if world == 1:
        return
    import torch.distributed as dist
    dist.barrier()
    if rank == 0:
        for stem in stems:
            parts = {}
            for r in range(world):
                f = os.path.join(save_path, f"{stem}_rank{r}.npz")
                if os.path.exists(f):
                    z = np.load(f)
                    for i, g in zip(z["ids"], z["gens"]):
                        parts[int(i)] = g
                    os.remove(f)
            assert sorted(parts) == list(range(rep)), f"{stem}: repetitions {sorted(parts)} of {rep} arrived"
            np.save(os.path.join(save_path, f"{stem}.npy"), np.stack([parts[i] for i in range(rep)], axis=0))
    dist.barrier()


def generate_3d(argv=None, configs_path=None):
    """Sampling entry point (/root/reference/generate_3D.py); (cube, repetition) chains are dealt round-robin to the ranks when
    launched under torchrun, rank 0 merges the shards into the reference's gen_{count}.npy files."""
    ap = argparse.ArgumentParser(description="Generate 3D CDM")
    ap.add_argument("model_name", type=str, help="Model name")
    ap.add_argument("save_path", type=str, help="Save path")
    ap.add_argument("runtype", type=str, help="Type of the generation")
    args = ap.parse_args(argv)
    assert args.model_name in GENERATE_WHITELIST, f"unknown model {args.model_name}"
    if "SFM" in args.model_name:
        raise NotImplementedError("This model is not implemented yet")
    assert args.runtype in ["CV_12_12", "CV_1_128"]
    os.makedirs(args.save_path, exist_ok=True)
    rank, world, device, config, model, dm = _sampling_setup(args.model_name, configs_path, "CV")
    n_steps = int(os.environ.get("VDM4CDM_SAMPLING_STEPS", 250))
    n_cubes, rep, sel = (12, 12, None) if args.runtype == "CV_12_12" else (1, 128, 2)
    rep = int(os.environ.get("VDM4CDM_REP", rep))
    count = 0
    for i_batch, batch in enumerate(dm.test_dataloader()):
        if sel is not None and i_batch != sel:
            continue
        parts = _sample_repetitions(model, config, batch, device, rep, count * rep, rank, world, n_steps)
        _save_or_shard(args.save_path, f"gen_{count}", parts, rank, world)
        count += 1
        if count == n_cubes:
            break
    _merge_shards(args.save_path, [f"gen_{c}" for c in range(count)], rep, rank, world)
    return count


ONE_P_GENS = [0, 4, 7, 23, 28]                          # indices into the 1P test set and their names (generate_3D_1P.py:44-45)
ONE_P_NAMES = ["fid", "Om_m2", "Om_p2", "ASN1_m3", "ASN1_p3"]


def generate_3d_1p(argv=None, configs_path=None):
    """``python generate_3D_1P.py <model_name> <save_path> <runtype>`` (runtype 1P_24 | 1P_128): `rep` samples for the fiducial and
    four one-parameter-varied conditioning cubes of the CAMELS 1P set -> <name>_<rep>.npy (/root/reference/generate_3D_1P.py:31-68)."""
    ap = argparse.ArgumentParser(description="Generate 3D CDM")
    ap.add_argument("model_name", type=str, help="Model name")
    ap.add_argument("save_path", type=str, help="Save path")
    ap.add_argument("runtype", type=str, help="Type of the generation")
    args = ap.parse_args(argv)
    assert args.model_name in GENERATE_WHITELIST + ["VDM_Mstar_Mcdm_c_c_256_comp"], f"unknown model {args.model_name}"
    if "SFM" in args.model_name:
        raise NotImplementedError("This model is not implemented yet")
    if args.runtype not in ["1P_24", "1P_128"]:
        raise NotImplementedError("This runtype is not implemented yet")
    os.makedirs(args.save_path, exist_ok=True)
    rank, world, device, config, model, dm = _sampling_setup(args.model_name, configs_path, "1P")
    n_steps = int(os.environ.get("VDM4CDM_SAMPLING_STEPS", 250))
    rep = int(os.environ.get("VDM4CDM_REP", 24 if args.runtype == "1P_24" else 128))
    stems = []
    for i_batch, batch in enumerate(dm.test_dataloader()):
        if i_batch not in ONE_P_GENS:
            continue
        k = ONE_P_GENS.index(i_batch)
        if rank == 0:
            print(ONE_P_NAMES[k], "params", batch["conditioning_values"], flush=True)
        parts = _sample_repetitions(model, config, batch, device, rep, k * rep, rank, world, n_steps)
        stems.append(f"{ONE_P_NAMES[k]}_{rep}")
        _save_or_shard(args.save_path, stems[-1], parts, rank, world)
    _merge_shards(args.save_path, stems, rep, rank, world)
    return stems


def train_uc_c(argv=None):
    """2D VDM conditioned on the six simulation parameters only (/root/reference/train_uc_c_from_field_name.py:50-122): same network
    as train_uc_uc plus one vector conditioning.  CPU plumbing through the explicit backend='torch'."""
    from . import data, networks, vdm_model
    from .trainer import Trainer
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit("usage: train_uc_c_from_field_name.py <field_name>")
    field_name = argv[0]
    _seed_everything(42)
    cropsize = int(os.environ.get("VDM4CDM_CROPSIZE_2D", 256))
    batch_size = int(os.environ.get("VDM4CDM_BATCH_2D", 12))
    dm = data.SyntheticAstroDataModule(cropsize=cropsize, batch_size=batch_size, dim=2, channel_names=[field_name, field_name],
                                       conditioning=False, n_params=6,
                                       return_func=lambda fields, params: {"x": fields[1], "conditioning": None, "conditioning_values": [params]})
    score_model = networks.CUNet(shape=(1, cropsize, cropsize), chs=[48, 96, 192, 384], s_conditioning_channels=0,
                                 v_conditioning_dims=[6], t_conditioning=True, norm_groups=8, dropout_prob=0.1,
                                 conv_padding_mode="circular", n_attention_heads=4, backend="torch")
    vdm = vdm_model.LightVDM(score_model=score_model, gamma_min=-13.3, gamma_max=13.3, noise_schedule="learned_linear",
                             draw_figure=None)
    trainer = Trainer(max_steps=int(os.environ.get("VDM4CDM_MAX_STEPS", 1_000_000)), val_check_interval=5000, gradient_clip_val=0.5,
                      every_n_train_steps=10_000, default_root_dir=os.environ.get("VDM4CDM_LOG_DIR", "./data/logs/vdm4cdm-2D"),
                      experiment_name=f"LH_uc_c_{field_name}", device="cpu")
    trainer.fit(model=vdm, datamodule=dm)
    return trainer
