"""CPU tests (no GPU): oracle vs the reference's golden vectors, host logic, C-ABI surface, C1 plumbing config."""
import importlib.util
import math
import os
import re

import numpy as np
import pytest
import torch

from helpers import grf, oracle_cfg, oracle_params, randomize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "pk_golden.npz"))


def _mk():
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_pk_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


MK = _mk()


# ------------------------------------------------------------------------------ P(k): oracle and product vs the reference
@pytest.mark.parametrize("case", MK.CASES, ids=[c[0] for c in MK.CASES])
def test_pk_oracle_matches_reference_golden(case):
    from oracle import pk_oracle
    name, seed, B, C, D, dim = case
    x = MK.make_field(seed, B, C, D, dim).numpy()
    y = MK.make_field(seed + 100, B, C, D, dim).numpy()
    k, p, n = pk_oracle.pk(x)
    assert np.array_equal(n, GOLD[f"{name}/N"])                       # integer mode counts: bit exact
    np.testing.assert_allclose(k, GOLD[f"{name}/k"], rtol=1e-5)
    np.testing.assert_allclose(p, GOLD[f"{name}/P"], rtol=2e-5)
    np.testing.assert_allclose(pk_oracle.pk(x, y)[1], GOLD[f"{name}/Pcross"], rtol=1e-4, atol=1e-5 * np.abs(GOLD[f"{name}/P"]).max())
    np.testing.assert_allclose(pk_oracle.get_ccs(x, y)[1], GOLD[f"{name}/cc"], atol=2e-6)
    np.testing.assert_allclose(pk_oracle.get_ccs(x, x)[1], GOLD[f"{name}/cc_self"], atol=2e-6)
    if dim == 2:
        np.testing.assert_allclose(pk_oracle.get_ccs(x, y, full=True)[1], GOLD[f"{name}/cc_full"], atol=2e-6)


@pytest.mark.parametrize("case", MK.CASES, ids=[c[0] for c in MK.CASES])
def test_pk_product_matches_reference_golden(case):
    from vdm4cdm_amd import utils
    name, seed, B, C, D, dim = case
    x = MK.make_field(seed, B, C, D, dim)
    y = MK.make_field(seed + 100, B, C, D, dim)
    k, p, n = utils.pk(x)
    assert n.dtype == torch.int32 and torch.equal(n, torch.from_numpy(GOLD[f"{name}/N"]))
    np.testing.assert_allclose(k.numpy(), GOLD[f"{name}/k"], rtol=1e-5)
    np.testing.assert_allclose(p.numpy(), GOLD[f"{name}/P"], rtol=2e-5)
    np.testing.assert_allclose(utils.get_ccs(x, y)[1].numpy(), GOLD[f"{name}/cc"], atol=2e-6)
    np.testing.assert_allclose(utils.get_ccs(x, x)[1].numpy(), np.ones_like(GOLD[f"{name}/cc_self"]), atol=2e-6)
    if dim == 2:
        np.testing.assert_allclose(utils.get_ccs(x, y, full=True)[1].numpy(), GOLD[f"{name}/cc_full"], atol=2e-6)


def test_pk_parseval_and_white_noise():
    """T7: white noise has flat E[P]; sum N*P equals sum |F|^2 over the kept shells."""
    from vdm4cdm_amd import utils
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 1, 32, 32, 32, generator=g)
    k, p, n = utils.pk(x)
    mean_p = p.mean(0)
    assert (mean_p / (32 ** 3) - 1).abs().max().item() < 0.35          # few modes in the lowest shells
    assert abs((mean_p[4:] / 32 ** 3).mean().item() - 1) < 0.02
    F = torch.fft.fftn(x[0, 0])
    kk = torch.sqrt(sum(g_ ** 2 for g_ in torch.meshgrid(*[torch.fft.fftfreq(32) * 32] * 3, indexing="ij")))
    kept = (kk.ceil() >= 1) & (kk.ceil() <= 16)
    assert (n[0] * p[0]).sum().item() == pytest.approx((F.abs() ** 2)[kept].sum().item(), rel=1e-4)


# ------------------------------------------------------------------------------ oracle self-consistency (T2, T3, T4)
def test_vdm_oracle_identities():
    from oracle import vdm_oracle as vo
    s = vo.Schedule(-13.3, 13.3)
    t = torch.linspace(0, 1, 21, dtype=torch.float64)
    g = s.gamma(t)
    assert torch.allclose(s.alpha(g) ** 2 + s.sigma(g) ** 2, torch.ones_like(g))
    assert g[0].item() == pytest.approx(-13.3) and g[-1].item() == pytest.approx(13.3)
    k = vo.step_coeffs(s, 0.6, 0.55)
    z, e = grf((1, 1, 8, 8, 8), 1, slope=0.0), grf((1, 1, 8, 8, 8), 2, slope=0.0)
    mean = float(k["ratio"]) * (z - float(k["c"]) * float(k["sigma_t"]) * e)
    x0 = (z - float(k["sigma_t"]) * e) / float(k["alpha_t"])
    ddnm = float(k["ratio"]) * (1 - float(k["c"])) * z + float(k["alpha_s"]) * float(k["c"]) * x0
    assert (mean - ddnm).abs().max().item() < 1e-5
    x = grf((2, 1, 8, 8, 8), 3)
    eps = grf((2, 1, 8, 8, 8), 4, slope=0.0)
    out = vo.vdm_loss(lambda zt, tn: eps, s, x, torch.tensor([0.2, 0.7], dtype=torch.float64), eps, eps)
    assert out["diffusion_loss"].item() == 0.0
    out0 = vo.vdm_loss(lambda zt, tn: eps, s, torch.zeros_like(x), torch.tensor([0.2, 0.7], dtype=torch.float64), eps, eps)
    var1 = 1 / (1 + math.exp(-13.3))
    assert out0["latent_loss"].item() == pytest.approx(0.5 * (var1 - math.log(var1) - 1) / math.log(2), rel=1e-3)
    assert torch.allclose(vo.antithetic_times(0.9, 4), torch.tensor([0.9, 0.15, 0.4, 0.65], dtype=torch.float64))


# ------------------------------------------------------------------------------ product host logic vs oracle (CPU)
@pytest.mark.parametrize("shape,chs,pm,sc,vd", [((1, 16, 16, 16), [8, 16, 32], "zeros", 1, [6]),
                                                  ((1, 16, 16), [16, 32, 48], "circular", 0, []),
                                                  ((1, 8, 8, 8), [8, 16], "circular", 1, [6, 3])])
def test_torch_backend_matches_oracle(shape, chs, pm, sc, vd):
    from oracle import unet_oracle
    from vdm4cdm_amd.networks import CUNet
    net = CUNet(shape=shape, chs=chs, s_conditioning_channels=sc, v_conditioning_dims=vd, t_conditioning=True, norm_groups=4,
                dropout_prob=0.1, conv_padding_mode=pm, backend="torch")
    randomize(net, 1).eval()
    B = 2
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(B, *shape, generator=g), torch.rand(B, generator=g)
    s = torch.randn(B, sc, *shape[1:], generator=g) if sc else None
    v = [torch.randn(B, d, generator=g) for d in vd]
    with torch.no_grad():
        y = net(x, t=t, s_conditioning=s, v_conditionings=v)
        ref = unet_oracle.cunet_forward(oracle_params(net), oracle_cfg(net), x, t, s, v)
    assert (y - ref).abs().max().item() < 1e-5


def test_cfg_pred_noise_and_sampler_on_torch_backend():
    """R10 on the explicit torch backend (C1 plumbing): `w_cfg` blends the given-v and masked-v estimates of one batch-doubled
    forward; off in training mode; the eager sampler follows the oracle's guided chain."""
    from oracle import unet_oracle, vdm_oracle
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import LightVDM
    shape = (1, 8, 8, 8)
    net = CUNet(shape=shape, chs=[8, 16], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=4, backend="torch")
    vdm = LightVDM(score_model=randomize(net, 5), gamma_max=13.3, w_cfg=1.5).eval()
    assert vdm.model.w_cfg == 1.5
    g = torch.Generator().manual_seed(1)
    B, n = 2, 6
    z1, s = torch.randn(B, *shape, generator=g), torch.randn(B, *shape, generator=g)
    v = [torch.rand(B, 6, generator=g)]
    noises = [torch.randn(B, *shape, generator=g) for _ in range(n)]
    P = oracle_params(net)
    score_v = lambda z, tn, vv: unet_oracle.cunet_forward(P, oracle_cfg(net), z, tn, s, vv)
    guided = vdm_oracle.cfg_score_fn(score_v, v, 1.5)
    with torch.no_grad():
        out = vdm.draw_samples(batch_size=B, n_sampling_steps=n, z=z1.clone(), noises=noises, s_conditioning=s, v_conditionings=v)
        ref = vdm_oracle.sample(guided, vdm_oracle.Schedule(-13.3, 13.3), z1.clone(), n, noises)
        assert (out - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-4
        g_t = vdm.model.gamma(torch.tensor(0.4))
        tn = torch.full((B,), 0.4)
        e = vdm.model.get_pred_noise(z1, g_t, s_conditioning=s, v_conditionings=v)
        assert (e - guided(z1, tn)).abs().max().item() < 1e-4
        vdm.train()
        e_tr = vdm.model.get_pred_noise(z1, g_t, s_conditioning=s, v_conditionings=v)
        net.eval()                                             # (dropout off for the comparison; vdm.model stays in training mode)
        e_tr = vdm.model.get_pred_noise(z1, g_t, s_conditioning=s, v_conditionings=v)
        assert vdm.model.training and (e_tr - score_v(z1, tn, v)).abs().max().item() < 1e-4
    with pytest.raises(AssertionError, match="mask out"):
        vdm.eval().model.get_pred_noise(z1, g_t, s_conditioning=s)


@pytest.mark.parametrize("shape", [(1, 16, 16), (1, 8, 8, 8)], ids=["2d", "3d"])
def test_mid_attention_torch_backend_matches_oracle(shape):
    """mid_attn=True (spec D13) on the explicit torch backend vs the oracle's attention block; the block is live."""
    from oracle import unet_oracle
    from vdm4cdm_amd.networks import CUNet
    net = CUNet(shape=shape, chs=[8, 16], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=4, mid_attn=True,
                n_attention_heads=4, backend="torch")
    randomize(net, 1).eval()
    g = torch.Generator().manual_seed(0)
    x, t, s, v = torch.randn(2, *shape, generator=g), torch.rand(2, generator=g), torch.randn(2, *shape, generator=g), [torch.randn(2, 6, generator=g)]
    with torch.no_grad():
        y = net(x, t=t, s_conditioning=s, v_conditionings=v)
        P = oracle_params(net)
        ref = unet_oracle.cunet_forward(P, oracle_cfg(net), x, t, s, v)
        ref_no = unet_oracle.cunet_forward({k: a for k, a in P.items() if not k.startswith("mid_attn")}, oracle_cfg(net), x, t, s, v)
    assert (y - ref).abs().max().item() < 1e-5
    assert (ref - ref_no).abs().max().item() > 1e-3


def test_sfm_torch_backend_matches_oracle_and_2d_script_runs(tmp_path):
    """SFM (spec D14) on the torch backend: loss and Euler sampler vs oracle/sfm_oracle.py; the 2D reference script
    trainSFM_c_uc_from_field_name.py (mid_attn=True) runs end to end on a shrunk CPU configuration."""
    from oracle import sfm_oracle, unet_oracle
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.sfm_model import LightSFM
    shape = (1, 8, 8, 8)
    net = CUNet(shape=shape, chs=[8, 16], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=4, backend="torch")
    sfm = LightSFM(velocity_model=randomize(net, 2), learning_rate=1e-3).eval()
    g = torch.Generator().manual_seed(4)
    B = 2
    x0, x1 = torch.randn(B, *shape, generator=g), torch.randn(B, *shape, generator=g)
    v = [torch.rand(B, 6, generator=g)]
    times = torch.tensor([0.2, 0.7])
    P = oracle_params(net)
    vel = lambda xt, t, x0_: unet_oracle.cunet_forward(P, oracle_cfg(net), xt, t, x0_, v)
    with torch.no_grad():
        loss, _ = sfm.model.get_loss(x0=x0, x1=x1, times=times, v_conditionings=v)
        ref, _, _ = sfm_oracle.sfm_loss(vel, x0, x1, times)
        assert loss.item() == pytest.approx(ref.item(), rel=1e-5)
        out = sfm.draw_samples(x0=x0, n_sampling_steps=5, v_conditionings=v)
        assert (out - sfm_oracle.sfm_sample(vel, x0, 5)).abs().max().item() < 1e-4
    batch = {"x0": x0, "x1": x1, "conditioning_values": v}
    sfm.train()
    l0 = sfm.training_step(batch)
    l0.backward()
    assert net.flat.grad is not None and net.flat.grad.abs().max().item() > 0
    sd = sfm.state_dict()
    assert all(k.startswith("model.velocity_model.") for k in sd)
    sfm.load_state_dict(sd)
    import subprocess
    import sys
    env = dict(os.environ, VDM4CDM_MAX_STEPS="2", VDM4CDM_LOG_DIR=str(tmp_path), VDM4CDM_CROPSIZE_2D="16", VDM4CDM_BATCH_2D="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "trainSFM_c_uc_from_field_name.py"), "Mstar", "Mcdm"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "train/loss" in open(tmp_path / "LH_c_uc_Mstar_to_Mcdm" / "metrics.jsonl").read()


def test_param_count_and_state_dict_roundtrip(tmp_path):
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import LightVDM
    net = CUNet(shape=(1, 128, 128, 128), chs=[32, 64, 128, 256], s_conditioning_channels=1, v_conditioning_dims=[6],
                norm_groups=8, dropout_prob=0.1, backend="hip")
    n = sum(math.prod(s) for _, s, _, _ in net.spec.items.values())
    assert 14.0e6 < n < 14.8e6                                     # ~14.2 M
    small = CUNet(shape=(1, 8, 8, 8), chs=[8, 16], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=4, backend="torch")
    vdm = LightVDM(score_model=randomize(small, 3), gamma_max=13.3)
    p = tmp_path / "m.ckpt"
    torch.save({"state_dict": vdm.state_dict()}, p)
    small2 = CUNet(shape=(1, 8, 8, 8), chs=[8, 16], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=4, backend="torch")
    vdm2 = LightVDM(score_model=small2, gamma_max=13.3)
    vdm2.load_state_dict(torch.load(p)["state_dict"])             # the reference's reload idiom (src/utils.py:468)
    assert torch.equal(small.flat, small2.flat)
    assert all(k.startswith("model.score_model.") for k in vdm.state_dict())


def test_hip_backend_refuses_cpu_and_2d():
    from vdm4cdm_amd.networks import CUNet
    net = CUNet(shape=(1, 8, 8, 8), chs=[8, 16], s_conditioning_channels=0, v_conditioning_dims=[], norm_groups=4, backend="hip")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 8, 8, 8), t=torch.zeros(1))
    net2 = CUNet(shape=(1, 8, 8), chs=[8, 16], norm_groups=4, backend="hip")
    with pytest.raises(NotImplementedError):
        net2(torch.zeros(1, 1, 8, 8), t=torch.zeros(1))
    with pytest.raises(AssertionError, match="n_attention_heads"):
        CUNet(shape=(1, 8, 8, 8), chs=[8, 18], mid_attn=True, n_attention_heads=4)


def test_c1_config_trains_on_cpu():
    """BASELINE config C1: 2D 64^2 unconditional VDM, batch 4, CPU PyTorch (explicit backend='torch'), learned-linear
    schedule and circular padding as in /root/reference/train_uc_uc_from_field_name.py:55-67,104-120."""
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import LightVDM
    from vdm4cdm_amd.trainer import clip_grad_norm_flat_
    torch.manual_seed(42)
    net = CUNet(shape=(1, 64, 64), chs=[16, 32, 48], s_conditioning_channels=0, v_conditioning_dims=[], t_conditioning=True,
                norm_groups=8, dropout_prob=0.1, conv_padding_mode="circular", n_attention_heads=4, backend="torch")
    vdm = LightVDM(score_model=net, draw_figure=None, gamma_min=-13.3, gamma_max=13.3, noise_schedule="learned_linear",
                   learning_rate=3e-4)
    opt = vdm.configure_optimizers()
    x = grf((4, 1, 64, 64), 11)
    losses = []
    for step in range(3):
        loss = vdm.training_step({"x": x, "conditioning": None, "conditioning_values": None}, step)
        opt.zero_grad()
        loss.backward()
        clip_grad_norm_flat_([p for p in vdm.parameters()], 0.5, use_hip=False)
        opt.step()
        losses.append(loss.item())
    assert all(math.isfinite(l) for l in losses)
    assert vdm.model.gamma_w.grad is not None
    out = vdm.draw_samples(batch_size=2, n_sampling_steps=3)
    assert out.shape == (2, 1, 64, 64)


def test_vdm_loss_torch_path_matches_oracle():
    from oracle import unet_oracle, vdm_oracle
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import VDM
    net = CUNet(shape=(1, 8, 8, 8), chs=[8, 16], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=4, backend="torch")
    randomize(net, 2).eval()
    m = VDM(net).eval()
    x, s = grf((2, 1, 8, 8, 8), 1), grf((2, 1, 8, 8, 8), 2)
    v = [torch.rand(2, 6, generator=torch.Generator().manual_seed(1))]
    times = torch.tensor([0.35, 0.85])
    eps, eps0 = grf(x.shape, 5, slope=0.0), grf(x.shape, 6, slope=0.0)
    with torch.no_grad():
        loss, met = m.get_loss(x, times=times, eps=eps, eps0=eps0, s_conditioning=s, v_conditionings=v)
    P = oracle_params(net)
    ref = vdm_oracle.vdm_loss(lambda z, tn: unet_oracle.cunet_forward(P, oracle_cfg(net), z, tn, s, v),
                              vdm_oracle.Schedule(-13.3, 13.3), x, times.double(), eps, eps0)
    for k in ("elbo", "diffusion_loss", "latent_loss", "reconstruction_loss"):
        assert met[k].item() == pytest.approx(ref[k].item(), rel=1e-4), k


def test_step_table_matches_oracle():
    from oracle import vdm_oracle
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import VDM
    m = VDM(CUNet(shape=(1, 8, 8, 8), chs=[8, 16], norm_groups=4, backend="torch"))
    n = 10
    tab = m.step_table(n)
    steps = torch.linspace(1.0, 0.0, n + 1).double()
    s = vdm_oracle.Schedule(-13.3, 13.3)
    for i in (0, 4, 9):
        k = vdm_oracle.step_coeffs(s, steps[i], steps[i + 1])
        ref = torch.stack([k["ratio"], k["c"] * k["sigma_t"], k["scale"], k["t_norm"]])
        assert torch.allclose(tab[i], ref, rtol=1e-12, atol=1e-15)


# ------------------------------------------------------------------------------ C-ABI surface
def test_cabi_exports_every_declared_symbol(hip_lib):
    from vdm4cdm_amd import _lib
    header = open(os.path.join(ROOT, "include", "vdm4cdm_hip.h")).read()
    declared = set(re.findall(r"\b(vdm_[a-z0-9_]+)\s*\(", header))
    declared -= {"vdm_status", "vdm_dtype"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(hip_lib, name), f"libvdm4cdm_hip.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), f"ctypes table out of sync: {declared ^ set(_lib.SIGNATURES)}"
    assert hip_lib.vdm_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define VDM_ABI_VERSION (\d+)", header).group(1))


def test_cabi_argument_errors_do_not_need_a_gpu(hip_lib):
    from vdm4cdm_amd._lib import ConvDesc
    d = ConvDesc(n=1, od=4, oh=4, ow=4, cin=32, cout=32, ksize=5, stride=1, upsample=0, pad_mode=0, dtype=0, out_f32=0)
    assert hip_lib.vdm_conv_packed_bytes(d, 0) == 0
    assert b"ksize" in hip_lib.vdm_last_error()
    d.ksize = 3
    assert hip_lib.vdm_conv_packed_bytes(d, 0) == 27 * 2 * 64 * 16 * 2     # 2 K-blocks (fp32: 16 ch) x 27 taps x NC=2 x 1 KiB
    assert hip_lib.vdm_conv_fwd(d, None, None, None, None, 0, None, None, None, None) == -1   # VDM_ERR_ARG, no launch
    # host-side planning only (4 tiles: since round 4 fp32 storage takes the small-grid tiles too - a 4^3 volume runs as four 1x4x16 tiles)
    assert hip_lib.vdm_conv_gn_tiles(d) == 4 and hip_lib.vdm_conv_kernel_variant(d, 0) == 0


def test_synthetic_datamodule_contract():
    from vdm4cdm_amd import data
    dm = data.get_dataset(dataset_name="CMD_128", channel_names=["Mstar", "Mcdm"], stage="fit", batch_size=2, cropsize=16,
                          return_func=lambda fields, params: {"conditioning": fields[0], "x": fields[1], "conditioning_values": [params]})
    b = next(iter(dm.train_dataloader()))
    assert b["x"].shape == (2, 1, 16, 16, 16) and b["conditioning"].shape == (2, 1, 16, 16, 16)
    assert isinstance(b["conditioning_values"], list) and b["conditioning_values"][0].shape == (2, 6)
    assert abs(b["x"].var().item() - 1) < 0.05
    rho = dm.unnorm_func(b["x"], 1)
    assert torch.allclose(dm.norm_func(rho, 1), b["x"], atol=1e-4)


def test_graft_entry_build_imports_and_checks_the_abi(hip_lib):
    """__graft_entry__.build() (run by the driver every round) must succeed on the CPU-only container."""
    import __graft_entry__ as g
    g.build()


# ------------------------------------------------------------------------------ DDNM: product loop vs the reference's own loop
from helpers import DD, replay_ddnm_case  # noqa: E402


@pytest.mark.parametrize("case", DD.CASES, ids=[c[0] for c in DD.CASES])
def test_ddnm_product_loop_matches_reference_golden(case):
    """utils.get_ddnm_result (product restatement, torch backend, CPU) against the output of the REFERENCE's own
    get_ddnm_result (/root/reference/src/utils.py:277-304) run on the oracle with the same noise stream."""
    x, gold, resid = replay_ddnm_case(case, "cpu", "torch")
    assert x.shape == gold.shape
    err = (x - gold).abs().max().item()
    assert err <= 1e-4 * gold.abs().max().item(), f"{case[0]}: {err} vs max|gold| {gold.abs().max().item()}"
    assert resid <= 1e-3


def test_load_state_dict_layouts():
    """Checkpoint tensors are accepted in this package's tap-major layout or in the PyTorch conv layout [cout, cin, k, k, k]
    (permuted, not reshaped); a concatenated skip weight is split; any other shape raises instead of being scrambled."""
    from vdm4cdm_amd.networks import CUNet
    mk = lambda: CUNet(shape=(1, 8, 8, 8), chs=[8, 16], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=4, backend="torch")
    a, b = randomize(mk(), 3), mk()
    sd = a.state_dict()
    torch_layout = {}
    for k, v in oracle_params(a).items():                 # oracle_params converts to [cout, cin, k, k, k] and concatenates skip/skip2
        torch_layout[k] = v
    b.load_state_dict(torch_layout)
    assert torch.equal(a.flat, b.flat)
    bad = dict(sd)
    bad["conv_in.weight"] = sd["conv_in.weight"].reshape(-1)           # right element count, wrong shape
    with pytest.raises(RuntimeError, match="expected"):
        mk().load_state_dict(bad)
    x = torch.randn(1, 1, 8, 8, 8, generator=torch.Generator().manual_seed(0))
    s = torch.randn(1, 1, 8, 8, 8, generator=torch.Generator().manual_seed(1))
    v = [torch.rand(1, 6, generator=torch.Generator().manual_seed(2))]
    with torch.no_grad():
        assert torch.equal(a.eval()(x, t=torch.tensor([0.3]), s_conditioning=s, v_conditionings=v),
                           b.eval()(x, t=torch.tensor([0.3]), s_conditioning=s, v_conditionings=v))


def test_host_side_under_asan_ubsan():
    """SURVEY section 5 (sanitizers): the HOST code of the C-ABI - descriptor validation incl. hostile values, pack plans, tile /
    workspace / scratch planners, kernel-variant selection, the argument-error paths of every launching entry point - runs under
    AddressSanitizer + UBSan (`make asan`: -Xarch_host -fsanitize=address,undefined; the device code is compiled but never run).
    CPU container only: GPU sanitizers are not available on the pool and the product never loads this build."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("sanitizer build is exercised on the CPU container only")
    csrc = os.path.join(ROOT, "vdm4cdm_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL)
    rt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "--print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    assert os.path.exists(rt), "ASan runtime of the ROCm clang not found"
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("VDM4CDM_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_asan_host_driver.py")], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "asan host driver ok" in r.stdout, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
