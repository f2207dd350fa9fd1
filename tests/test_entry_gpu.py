"""GPU tests through the reference-named entry points and the P(k)-level acceptance check."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from helpers import DD, grf, oracle_cfg, oracle_params, randomize, replay_ddnm_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def test_train_script_runs_on_gpu(tmp_path):
    """python trainVDM3D128_c_c_from_field_name_thick_lowbatch.py Mstar Mcdm <cropsize> (HIP backend, 3 steps, 32^3 crop)."""
    env = dict(os.environ, VDM4CDM_MAX_STEPS="3", VDM4CDM_LOG_DIR=str(tmp_path), VDM4CDM_PRECISION="bf16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "trainVDM3D128_c_c_from_field_name_thick_lowbatch.py"), "Mstar", "Mcdm", "32"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    log = tmp_path / "LH128_c_c_Mstar_to_Mcdm_thick_lowbatch_32" / "metrics.jsonl"
    lines = open(log).read().strip().splitlines()
    assert len(lines) >= 1 and "train/elbo" in lines[0]


def test_sfm_train_script_runs_on_gpu_from_files(tmp_path):
    """python trainSFM3D128_c_c_from_field_name_thick_lowbatch.py Mstar Mcdm <cropsize> on a synthetic on-disk data set in the
    reference's file layout: file-backed AstroDataModule (device data path) -> LightSFM on the HIP backend, 3 steps."""
    from vdm4cdm_amd import data
    root = data.write_synthetic_camels(str(tmp_path / "camels"), dataset_name="CMD_128", n_sims=3, fullsize=32, seed=1)
    env = dict(os.environ, VDM4CDM_MAX_STEPS="3", VDM4CDM_LOG_DIR=str(tmp_path), VDM4CDM_PRECISION="bf16", VDM4CDM_DATA_ROOT=root)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "trainSFM3D128_c_c_from_field_name_thick_lowbatch.py"), "Mstar", "Mcdm", "16"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "SYNTHETIC" not in r.stderr
    lines = open(tmp_path / "LH128_c_c_Mstar_to_Mcdm_thick_lowbatch_16" / "metrics.jsonl").read().strip().splitlines()
    assert len(lines) >= 1 and "train/loss" in lines[0]


def test_generate_3d_runs_on_gpu(tmp_path):
    """python generate_3D.py <model> <dir> CV_12_12 on a shrunk registry entry (cropsize 16, 2 cubes x 2 reps, 5 steps)."""
    cfgs = yaml.safe_load(open(os.path.join(ROOT, "configs.yaml")))
    cfgs["VDM_Mstar_Mcdm_c_c_128"].update(cropsize=16, chs=[16, 32], ckpt_path=str(tmp_path / "none.ckpt"))
    cfg_path = tmp_path / "configs.yaml"
    yaml.safe_dump(cfgs, open(cfg_path, "w"))
    env = dict(os.environ, VDM4CDM_SAMPLING_STEPS="5", VDM4CDM_REP="2")
    code = ("import sys; sys.path.insert(0, %r); from vdm4cdm_amd.entry import generate_3d; "
            "generate_3d(['VDM_Mstar_Mcdm_c_c_128', %r, 'CV_12_12'], configs_path=%r)" % (ROOT, str(tmp_path / "out"), str(cfg_path)))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    g0 = np.load(tmp_path / "out" / "gen_0.npy")
    assert g0.shape == (2, 1, 16, 16, 16) and g0.dtype == np.float32 and np.isfinite(g0).all()
    assert len(list((tmp_path / "out").glob("gen_*.npy"))) == 12          # 12 conditioning cubes (reference generate_3D.py:49-68)
    assert not np.array_equal(g0[0], g0[1])                                # repetitions use different chain seeds


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 1e-2)], ids=["fp32", "bf16"])
def test_sample_power_spectrum_matches_oracle(precision, tol):
    """Acceptance metric of BASELINE.json (sampled fields' P(k) within 1 % per k-bin), evaluated HIP vs oracle since no
    trained weights exist: same weights, same z_1, same per-step noise, D=32, 20 steps; P(k) by the pinned estimator."""
    from oracle import unet_oracle, vdm_oracle
    from vdm4cdm_amd import utils
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import LightVDM
    D, n = 32, 20
    net = CUNet(shape=(1, D, D, D), chs=[16, 32, 64], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=8,
                dropout_prob=0.1, backend="hip", precision=precision)
    randomize(net, 4, zero_init_std=0.01)                 # near-identity denoiser: a tame (non-expansive) chain
    vdm = LightVDM(score_model=net, gamma_max=13.3).to(DEV).eval()
    s = grf((1, 1, D, D, D), 7)
    v = [torch.rand(1, 6, generator=torch.Generator().manual_seed(8))]
    z1 = grf((1, 1, D, D, D), 9, slope=0.0)
    noises = [grf((1, 1, D, D, D), 100 + i, slope=0.0) for i in range(n)]
    out = vdm.draw_samples(batch_size=1, n_sampling_steps=n, z=z1.clone(), noises=noises, s_conditioning=s.to(DEV),
                           v_conditionings=[v[0].to(DEV)]).cpu()
    P = oracle_params(net)
    ref = vdm_oracle.sample(lambda z, tn: unet_oracle.cunet_forward(P, oracle_cfg(net), z, tn, s, v),
                            vdm_oracle.Schedule(-13.3, 13.3), z1.clone(), n, noises)
    # P(k) of the sampled (normalised log-density) fields with the pinned estimator.  (calc_SS.get_pk_3d exponentiates first;
    # with untrained weights the samples are not physical log-densities and 10**x overflows fp32, so the raw fields are used.)
    assert torch.isfinite(out).all() and torch.isfinite(ref).all()
    k, pk_hip, _ = utils.pk(out)
    _, pk_ref, _ = utils.pk(ref)
    ratio = (pk_hip / pk_ref).numpy()
    assert np.abs(ratio - 1).max() < tol, f"P(k) ratio off by {np.abs(ratio - 1).max():.3e} ({precision})"


@pytest.mark.parametrize("case", DD.CASES, ids=[c[0] for c in DD.CASES])
def test_ddnm_hip_matches_reference_golden(case):
    """SURVEY 8f rank 1, pinned by the reference itself: utils.get_ddnm_result on the HIP backend (fp32 storage) against the output of
    the REFERENCE's own get_ddnm_result (/root/reference/src/utils.py:277-304) run on the CPU oracle with the same seeded noise
    stream (tests/golden/make_ddnm_golden.py -> ddnm_golden.npz).  Tolerance 2e-3 * max|gold| (the untrained chain is expansive:
    fp32 rounding differences between the MFMA and oneDNN accumulation orders are amplified ~1e3x along it)."""
    x, gold, resid = replay_ddnm_case(case, DEV, "hip", "fp32")
    assert x.shape == gold.shape and torch.isfinite(x).all()
    err = (x - gold).abs().max().item()
    assert err <= 2e-3 * gold.abs().max().item(), f"{case[0]}: {err} vs max|gold| {gold.abs().max().item()}"
    assert resid <= 1e-3


def test_ddnm_hip_bf16_stays_on_the_measurement():
    """bf16 storage: the chain is too expansive for an element-wise bound, but the DDNM range-space identity A x = y holds for any
    denoiser, and the result stays finite and close to the fp32 fixture in the cosine sense."""
    case = DD.CASES[1]
    x, gold, resid = replay_ddnm_case(case, DEV, "hip", "bf16")
    assert torch.isfinite(x).all() and resid <= 1e-3
    cos = torch.nn.functional.cosine_similarity(x.flatten(), gold.flatten(), dim=0).item()
    assert cos > 0.98, cos


def _pk_golden_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_pk_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


PKMK = _pk_golden_cases()


@pytest.mark.parametrize("case", PKMK.CASES, ids=[c[0] for c in PKMK.CASES])
def test_pk_on_device_matches_reference_golden(case):
    """R12: utils.pk / get_ccs on CUDA tensors (rocFFT + device bincount) against the golden vectors of the reference's own
    src/utils.py (tests/golden/pk_golden.npz): integer mode counts bit-exact, k and P within 2e-5 relative."""
    from vdm4cdm_amd import utils
    gold = np.load(os.path.join(ROOT, "tests", "golden", "pk_golden.npz"))
    name, seed, B, C, D, dim = case
    x = PKMK.make_field(seed, B, C, D, dim).to(DEV)
    y = PKMK.make_field(seed + 100, B, C, D, dim).to(DEV)
    k, p, n = utils.pk(x)
    assert k.is_cuda and n.dtype == torch.int32
    assert np.array_equal(n.cpu().numpy(), gold[f"{name}/N"])
    np.testing.assert_allclose(k.cpu().numpy(), gold[f"{name}/k"], rtol=1e-5)
    np.testing.assert_allclose(p.cpu().numpy(), gold[f"{name}/P"], rtol=2e-5)
    np.testing.assert_allclose(utils.pk(x, y)[1].cpu().numpy(), gold[f"{name}/Pcross"], rtol=1e-4, atol=1e-5 * np.abs(gold[f"{name}/P"]).max())
    np.testing.assert_allclose(utils.get_ccs(x, y)[1].cpu().numpy(), gold[f"{name}/cc"], atol=5e-6)
    if dim == 2:
        np.testing.assert_allclose(utils.get_ccs(x, y, full=True)[1].cpu().numpy(), gold[f"{name}/cc_full"], atol=5e-6)


def test_hip_training_step_under_rccl_world1(tmp_path):
    """HIP kernels and a process group in ONE process: Trainer.fit on the HIP backend inside an `nccl` (RCCL) group of world size 1,
    once with the bucketed all-reduce forced on (the N > 1 code path: 4 slices of the flat gradient averaged on the communication
    stream while the backward continues) and once without.  Averaging over one rank is the identity and the step is deterministic,
    so the two runs must end with bit-identical weights."""
    import socket
    outs = {}
    for forced in ("1", ""):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        d = tmp_path / f"f{forced or 0}"
        d.mkdir()
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VDM4CDM_FORCE_BUCKETS")}
        if forced:
            env["VDM4CDM_FORCE_BUCKETS"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_ddp_gpu_worker.py"), str(d), str(port)], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs[forced] = torch.load(d / "out.pt")
    assert outs["1"]["bucketed"] and not outs[""]["bucketed"]
    b = outs["1"]["bounds"]
    assert len(b) == 4 and b[-1][0] == 0 and all(b[i][0] == b[i + 1][1] for i in range(3))      # contiguous cover, back to front
    assert torch.isfinite(outs["1"]["flat"]).all()
    assert torch.equal(outs["1"]["flat"], outs[""]["flat"]), "bucketed all-reduce changed the result of a world-1 step"


def _spawn_ranks(cmd, world, port, tmp_path, extra_env=None, timeout=900):
    """world processes with the torchrun environment, all on GPU 0 over gloo (VDM4CDM_SHARE_GPU / VDM4CDM_DIST_BACKEND)."""
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VDM4CDM_SHARE_GPU="1", VDM4CDM_DIST_BACKEND="gloo", **(extra_env or {}))
        procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    return outs


@pytest.mark.parametrize("world,precision", [(2, "fp32"), (3, "bf16")], ids=["world2_fp32", "world3_bf16"])
def test_hip_training_two_ranks_share_one_gpu(tmp_path, world, precision, monkeypatch):
    """SURVEY 8e with HIP kernels AND a process group in the same processes (2 / 3 ranks on the one GPU of the box, gloo): weights
    are broadcast from rank 0, each rank draws its own data shard / noise (16 training cubes over 3 ranks: the wrap-around padding), the
    backward all-reduces the four gradient buckets on the communication stream, and after 3 steps all ranks hold bit-identical
    parameters.  world 3 runs bf16 storage: the skip convs folded into the GroupNorm passes and the fused tail write their gradients
    from the main stream under the buckets."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    monkeypatch.setenv("DDP_PRECISION", precision)
    _spawn_ranks([sys.executable, os.path.join(ROOT, "tests", "_ddp_gpu2_worker.py"), str(tmp_path)], world, port, tmp_path)
    outs = [torch.load(tmp_path / f"out{r}.pt") for r in range(world)]
    assert all(o["world"] == world and o["bucketed"] for o in outs)
    assert torch.isfinite(outs[0]["flat"]).all()
    for o in outs[1:]:
        assert torch.equal(outs[0]["flat"], o["flat"]), "ranks diverged"
    assert len({tuple(o["loss"]) for o in outs}) == world, "two ranks saw the same batch / noise"


def test_bench_two_ranks_share_one_gpu(tmp_path):
    """bench.py under the torchrun contract with N = 2 (rehearsal on one GPU): one JSON line from rank 0, aggregate voxels/s over both
    ranks, weak scaling."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    outs = _spawn_ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "4", "--warmup", "2",
                         "--no-cpu-baseline", "--sample-steps", "0"], 2, port, tmp_path)
    lines = [ln for so, _ in outs for ln in so.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly rank 0 prints the line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] == pytest.approx(2 * 2 * 32 ** 3 / (d["ms_per_step"] * 1e-3), rel=1e-6)


def test_bench_self_spawn_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2` WITHOUT a torchrun environment: bench.py starts the two rank processes itself (the parent never
    touches the GPU) and forwards rank 0's line, n_gpus = 2.  A mismatching WORLD_SIZE is refused instead of silently measured."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(VDM4CDM_SHARE_GPU="1", VDM4CDM_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tiny", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--sample-steps", "12"]
    r = subprocess.run(cmd + ["--gpus", "2"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] == pytest.approx(2 * 2 * 32 ** 3 / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["roofline"]["kernel"] and d["roofline"]["largest_by_total_time"]["kernel"]
    # every rank's own clock around the timed steps; `ms_per_step` is the slowest
    pr = d["ms_per_step_per_rank"]
    assert len(pr["all"]) == 2 and pr["max"] == pytest.approx(d["ms_per_step"], rel=1e-9) and pr["min"] <= pr["max"]
    # the sampling half of the metric on N GPUs: one chain per rank, different seeds -> different cubes, aggregate over the chains
    sm = d["sample"]
    assert "error" not in sm and sm["chains"] == 2 and sm["steps"] == 12 and sm["finite"]
    assert len(sm["seconds_per_rank"]) == 2 and sm["seconds_max"] == max(sm["seconds_per_rank"]) == sm["seconds"]
    assert sm["seeds"][0] != sm["seeds"][1] and sm["std_per_rank"][0] != sm["std_per_rank"][1], "the two ranks sampled the same chain"
    assert sm["aggregate_steps_per_s"] == pytest.approx(2 * 12 / sm["seconds_max"], rel=1e-9)
    # --gpus 4 inside a WORLD_SIZE=1 environment: refused, no line
    r = subprocess.run(cmd + ["--gpus", "4"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in r.stderr


def _ss_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("test_entry_cpu_mod", os.path.join(ROOT, "tests", "test_entry_cpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


ENTRY_CPU = _ss_cases()


@pytest.mark.parametrize("case", ENTRY_CPU.SSMK.CASES, ids=[c[0] for c in ENTRY_CPU.SSMK.CASES])
def test_calc_ss_on_device_matches_reference_golden(case):
    """calc_ss.get_stats on CUDA tensors (rocFFT P(k), device-side histograms) against the outputs of the reference's
    own calc_SS.py functions (tests/golden/ss_golden.npz): histogram counts (edge elements: see check_ss_case), P(k) within 1e-4."""
    ENTRY_CPU.check_ss_case(case, DEV)


@pytest.mark.parametrize("mode,precision", [("nocache", "bf16"), ("nocache", "fp32"), ("cached", "bf16")], ids=["nocache_bf16", "nocache_fp32", "cached_bf16"])
def test_streams_and_allocator_hazards(mode, precision):
    """Flush cross-stream memory hazards out deliberately (the round-3 use-after-free through the caching allocator surfaced 200 tests
    later, in rocFFT): one process runs [3 training steps (+ the graph-captured step) -> sampler -> rocFFT P(k) -> teardown] three times
    with identical seeds and must reproduce round 0 bit for bit - with PYTORCH_NO_CUDA_MEMORY_CACHING=1 (every free is a hipFree) and
    with the default allocator under empty_cache() between the phases (tests/_stream_hazard_worker.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(HAZARD_MODE=mode, HAZARD_PRECISION=precision)
    if mode == "nocache":
        env["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_stream_hazard_worker.py")], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "HAZARD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_fp32_exact_library_passes_the_fp32_parity_cases():
    """fp32 storage has two builds of the same C-ABI: the default computes each product as three bf16 MFMAs (hi*hi + hi*lo + lo*hi,
    csrc/common.h VDM_FP32_SPLIT), VDM4CDM_FP32_EXACT=1 loads libvdm4cdm_hip_fp32exact.so (v_mfma_f32_16x16x4_f32).  The in-process fp32
    tests cover the default; this child runs the conv / folded-GroupNorm / UNet fp32 cases against the exact library at the SAME
    tolerances, so the selectable build cannot rot."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["VDM4CDM_FP32_EXACT"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_kernels_gpu.py", "tests/test_unet_gpu.py", "-x", "-q", "-m", "gpu", "-k",
                        "(test_conv_grads and f32) or (test_gn_bwd_folded_into_dgrad and f32) or test_unet_forward_fp32 or test_unet_backward_fp32",
                        "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
