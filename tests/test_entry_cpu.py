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
