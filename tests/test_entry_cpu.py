"""Entry points and config registry (CPU)."""
import os
import subprocess
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_configs_yaml_schema_and_defaults():
    cfgs = yaml.safe_load(open(os.path.join(ROOT, "configs.yaml")))
    from vdm4cdm_amd import entry, utils
    assert set(entry.GENERATE_WHITELIST) == set(cfgs)           # generate_3D.py whitelist (reference generate_3D.py:10-14)
    for name, c in cfgs.items():
        assert {"type", "ckpt_path", "in_field_name", "out_field_name", "cropsize", "data_params"} <= set(c)
        assert "dataset_name" in c["data_params"]
    m = utils.get_model(cfgs["VDM_Mstar_Mcdm_c_c_224"], load_ckpt=False)
    sm = m.model.score_model
    assert sm.shape == (1, 224, 224, 224) and sm.chs == [16, 32, 64, 128] and sm.conv_padding_mode == "zeros"
    assert sm.norm_groups == 8 and sm.dropout_prob == 0.1 and sm.v_conditioning_dims == [6] and m.learning_rate == 3e-4
    m256 = utils.get_model(cfgs["VDM_Mstar_Mcdm_c_uc_256"], load_ckpt=False)
    assert m256.model.score_model.conv_padding_mode == "circular" and m256.model.score_model.v_conditioning_dims == []
    assert m.model.gamma_max == 13.3 and m.model.gamma_min == -13.3
    assert utils.get_model(cfgs["SFM_Mstar_Mcdm_c_c_128"]) is None
    dm = utils.get_datamodule(dict(cfgs["VDM_Mstar_Mcdm_c_c_128"], cropsize=16))
    b = next(iter(dm.test_dataloader()))
    assert b["x"].shape == (1, 1, 16, 16, 16) and len(b["conditioning_values"]) == 1


def test_train_uc_uc_entry_runs_on_cpu(tmp_path):
    """BASELINE config C1 through the reference-named script (shrunk to 32^2, 2 steps)."""
    env = dict(os.environ, VDM4CDM_MAX_STEPS="2", VDM4CDM_CROPSIZE_2D="32", VDM4CDM_BATCH_2D="4", VDM4CDM_LOG_DIR=str(tmp_path),
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_uc_uc_from_field_name.py"), "Mcdm"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(tmp_path / "LH_uc_uc_Mcdm" / "metrics.jsonl")


def test_entry_scripts_cli_errors():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "trainVDM3D128_c_c_from_field_name_thick_lowbatch.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "usage" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "generate_3D.py"), "SFM_Mstar_Mcdm_c_c_128", "/tmp/x", "CV_12_12"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "NotImplementedError" in r.stderr


def test_ddnm_sampler_api_on_torch_backend():
    """get_ddnm_result (reference src/utils.py:277-304) drives sample_zt_given_zs / sample_zs_given_zt(return_ddnm=True)."""
    import torch
    from helpers import randomize
    from vdm4cdm_amd import utils
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import LightVDM
    net = CUNet(shape=(1, 8, 8, 8), chs=[8, 16], s_conditioning_channels=0, v_conditioning_dims=[], norm_groups=4, backend="torch")
    vdm = LightVDM(score_model=randomize(net, 1, zero_init_std=0.01), gamma_max=13.3).eval()
    mask = torch.zeros(1, 1, 8, 8, 8)
    mask[..., :4] = 1
    y = torch.randn(1, 1, 8, 8, 8) * mask
    out = utils.get_ddnm_result(vdm, y, A=lambda x: x * mask, AT=lambda x: x * mask, n_sampling_steps=4, l=1)
    assert out.shape == (1, 1, 8, 8, 8) and torch.isfinite(out).all()
    assert torch.allclose(out * mask, y, atol=1e-5)            # the range-space part is pinned to the observation


# ------------------------------------------------------------------------------ calc_SS statistics vs the reference's own functions
import numpy as np  # noqa: E402

SS_GOLD = np.load(os.path.join(ROOT, "tests", "golden", "ss_golden.npz"))


def _ss_mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk_ss", os.path.join(ROOT, "tests", "golden", "make_ss_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


SSMK = _ss_mod()


def check_ss_case(case, device):
    """calc_ss.get_stats on `device` against the outputs of the reference's get_pk_3d / get_pk_2d / get_logpdf_3d / get_logpdf_2d
    (tests/golden/make_ss_golden.py): histogram counts bit-exact on the CPU; on a GPU the fp32 log10 may differ from the host
    libm's by one ulp, which can move an element that sits on a bin edge into the neighbouring bin - there the cumulative counts
    may differ by at most 1e-4 of the elements (and the totals must agree exactly).  P(k) within 1e-4, moments within 1e-5."""
    from vdm4cdm_amd import calc_ss
    name, seed, B, D = case
    f = SSMK.density(seed, B, D).to(device)
    st = calc_ss.get_stats(f, resol=D)
    assert not any("rwst" in k for k in st)
    for key in ("3d", "2d_half", "2d_quarter"):
        got, gold = st[f"{key}_logpdf"], SS_GOLD[f"{name}/{key}_logpdf"]
        if str(device) == "cpu":
            assert np.array_equal(got, gold), f"{name}/{key}_logpdf"
        else:
            assert np.array_equal(got.sum(1), gold.sum(1)), f"{name}/{key}_logpdf totals"
            slack = max(1, int(1e-4 * gold.sum(1).max()))
            assert np.abs(np.cumsum(got, 1) - np.cumsum(gold, 1)).max() <= slack, f"{name}/{key}_logpdf"
        assert got.sum() > 0
        np.testing.assert_allclose(st[f"{key}_pk"], SS_GOLD[f"{name}/{key}_pk"], rtol=1e-4)
        np.testing.assert_allclose([st[f"{key}_mean"], st[f"{key}_std"]], SS_GOLD[f"{name}/{key}_mean"], rtol=1e-5)


@pytest.mark.parametrize("case", SSMK.CASES, ids=[c[0] for c in SSMK.CASES])
def test_calc_ss_matches_reference_golden(case):
    check_ss_case(case, "cpu")


def test_sampling_scripts_and_calc_ss_end_to_end(tmp_path):
    """generate_3D.py CV_12_12 + generate_3D_1P.py 1P_24 (torch backend on CPU, shrunk registry entry) -> calc_SS.py -> summary.pth
    with the reference's key layout (stats / images per generated cube and per ground-truth cube)."""
    import torch
    cfgs = yaml.safe_load(open(os.path.join(ROOT, "configs.yaml")))
    name = "VDM_Mstar_Mcdm_c_c_128"
    cfgs[name].update(cropsize=8, res=8, chs=[8, 16], ckpt_path=str(tmp_path / "none.ckpt"))
    cfg_path = tmp_path / "configs.yaml"
    yaml.safe_dump(cfgs, open(cfg_path, "w"))
    gen_dir = tmp_path / "gen"
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, OMP_NUM_THREADS="2",
               VDM4CDM_BACKEND="torch", VDM4CDM_SAMPLING_STEPS="2", VDM4CDM_REP="2", VDM4CDM_GEN_DIR=str(gen_dir))
    for fn, runtype in (("generate_3d", "CV_12_12"), ("generate_3d_1p", "1P_24")):
        code = ("import sys; sys.path.insert(0, %r); from vdm4cdm_amd import entry; entry.%s([%r, %r, %r], configs_path=%r)"
                % (ROOT, fn, name, str(gen_dir / name / runtype), runtype, str(cfg_path)))
        r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
    assert sorted(f.name for f in (gen_dir / name / "1P_24").glob("*.npy")) == sorted(f"{n}_2.npy" for n in ["fid", "Om_m2", "Om_p2", "ASN1_m3", "ASN1_p3"])
    # calc_SS expects 12 repetitions per CV cube and 24 per 1P cube; the shrunk run has 2: statistics of what is there
    code = ("import sys; sys.path.insert(0, %r); from vdm4cdm_amd import calc_ss; calc_ss.main([%r], configs_path=%r)"
            % (ROOT, name, str(cfg_path)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, VDM4CDM_REP="2"), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    summ = torch.load(gen_dir / name / "summary.pth", weights_only=False)
    assert set(summ) == {"CV_12_12", "1P_24"}
    st = summ["CV_12_12"]["stats"]
    assert "Mcdm_GT_0" in st and "Mcdm_11_1" in st and st["Mcdm_0_0"]["3d_pk"].shape == (1, 4) and st["Mcdm_0_0"]["3d_logpdf"].shape == (1, 99)
    assert "half_Mcdm_3_1" in summ["CV_12_12"]["images"] and "quarter_cond_GT_5" in summ["CV_12_12"]["images"]
    assert "Mcdm_GT_fid" in summ["1P_24"]["stats"] and "Mcdm_ASN1_p3_1" in summ["1P_24"]["stats"]


def test_train_uc_c_entry_runs_on_cpu(tmp_path):
    env = dict(os.environ, VDM4CDM_MAX_STEPS="2", VDM4CDM_CROPSIZE_2D="32", VDM4CDM_BATCH_2D="4", VDM4CDM_LOG_DIR=str(tmp_path),
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_uc_c_from_field_name.py"), "Mcdm"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(tmp_path / "LH_uc_c_Mcdm" / "metrics.jsonl")


def test_bench_never_prints_a_line_for_fewer_gpus_than_asked():
    """`python bench.py --gpus 2` outside torchrun starts two rank processes itself; here (no GPU) they fail, and the launcher must
    return non-zero WITHOUT printing a JSON line - in particular never an n_gpus = 1 line for a 2-GPU request."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the self-spawn path is covered by tests/test_entry_gpu.py::test_bench_self_spawn_two_ranks_share_one_gpu")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank processes failed" in r.stderr
