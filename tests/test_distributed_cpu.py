"""N>1 path on CPU: world_size-2 gloo run of the trainer (one all-reduce of the flat gradient per step)."""
import os
import socket
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_trainer_ddp_gloo_world2(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_ddp_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    a, b = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert a["steps"] == 2 and b["steps"] == 2
    assert torch.equal(a["flat"], b["flat"]), "ranks diverged: gradients were not all-reduced / weights not broadcast"
    assert torch.equal(a["gamma_w"], b["gamma_w"])
    # rank 0 wrote the checkpoint with the reference's top-level key
    ck = torch.load(tmp_path / "ddp" / "checkpoints" / "epoch=0-step=2.ckpt")
    assert "state_dict" in ck and any(k.startswith("model.score_model.") for k in ck["state_dict"])
    # the ranks saw different data (per-rank seeds), so the averaged gradient differs from either local one:
    assert a["history"][0]["loss"] != b["history"][0]["loss"]


def test_trainer_ddp_gloo_world3_epoch_boundaries_and_validation(tmp_path):
    """World 3 with a data set that does not divide (8 items, batch 2): 5 steps cross several epoch boundaries and two validation passes
    (sharded split, all-reduced loss, barrier after rank 0's part).  Every rank must take exactly the same number of steps, log the same
    validation loss and end with bit-identical parameters - a rank that ran one batch more or less per epoch would hang or diverge."""
    env = dict(os.environ, OMP_NUM_THREADS="2", DDP_MAX_STEPS="5", DDP_VAL_EVERY="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_ddp_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    outs = [torch.load(tmp_path / f"rank{k}.pt") for k in range(3)]
    assert [o["steps"] for o in outs] == [5, 5, 5]
    for o in outs[1:]:
        assert torch.equal(o["flat"], outs[0]["flat"]) and torch.equal(o["gamma_w"], outs[0]["gamma_w"]), "ranks diverged"
    vals = [[h["val_loss"] for h in o["history"] if "val_loss" in h] for o in outs]
    assert len(vals[0]) == 2 and vals[0] == vals[1] == vals[2], f"validation loss not shared by the ranks: {vals}"
    train = [[h["loss"] for h in o["history"] if "loss" in h] for o in outs]
    assert all(len(t) == 5 for t in train) and train[0] != train[1], "every rank logs 5 steps on its own shard"


def test_single_process_matches_itself(tmp_path):
    """Sanity: the same worker without torchrun (world 1) runs and is deterministic."""
    outs = []
    for i in range(2):
        d = tmp_path / f"r{i}"
        d.mkdir()
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_ddp_worker.py"), str(d)], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(torch.load(d / "rank0.pt")["flat"])
    assert torch.equal(outs[0], outs[1])


def _gen_cfg(tmp_path):
    import yaml
    cfgs = yaml.safe_load(open(os.path.join(ROOT, "configs.yaml")))
    cfgs["VDM_Mstar_Mcdm_c_c_128"].update(cropsize=8, chs=[8, 16], ckpt_path=str(tmp_path / "none.ckpt"))
    p = tmp_path / "configs.yaml"
    yaml.safe_dump(cfgs, open(p, "w"))
    return str(p)


def _run_generate(tmp_path, out, world):
    code = ("import sys; sys.path.insert(0, %r); from vdm4cdm_amd.entry import generate_3d; "
            "generate_3d(['VDM_Mstar_Mcdm_c_c_128', %r, 'CV_12_12'], configs_path=%r)" % (ROOT, str(out), _gen_cfg(tmp_path)))
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, OMP_NUM_THREADS="2",
               VDM4CDM_BACKEND="torch", VDM4CDM_SAMPLING_STEPS="3", VDM4CDM_REP="3")
    if world == 1:
        cmd = [sys.executable, "-c", code]
    else:
        script = tmp_path / "gen_worker.py"
        script.write_text(code)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_generate_3d_chain_sharding_gloo_world2(tmp_path):
    """generate_3D under torchrun (2 ranks, gloo): the (cube, repetition) chains are dealt round-robin, every chain is sampled exactly
    once, its seed depends on the global chain id only, and rank 0 assembles gen_{count}.npy - the merged files are bit-identical to
    a single-process run (disjoint + complete + seed = f(chain id)), and no shard files are left behind."""
    import numpy as np
    from vdm4cdm_amd.entry import chain_seed
    assert chain_seed(0) != chain_seed(1) and chain_seed(7) == chain_seed(7)
    one, two = tmp_path / "w1", tmp_path / "w2"
    _run_generate(tmp_path, one, 1)
    _run_generate(tmp_path, two, 2)
    files1 = sorted(f.name for f in one.glob("*"))
    files2 = sorted(f.name for f in two.glob("*"))
    assert files1 == [f"gen_{c}.npy" for c in sorted(range(12), key=str)] and files2 == files1, (files1, files2)
    for name in files1:
        a, b = np.load(one / name), np.load(two / name)
        assert a.shape == (3, 1, 8, 8, 8) and np.array_equal(a, b), name
    g0 = np.load(one / "gen_0.npy")
    assert not np.array_equal(g0[0], g0[1])


def test_ddp_gradient_equals_full_batch_gloo_world2(tmp_path):
    """Data parallelism is an identity, not just a consensus: 2 ranks x 2 samples with the gradient averaged over the ranks equal
    1 process x 4 samples on the same samples / times / noise (per-sample GroupNorm; loss = batch mean).  CPU twin (torch backend, gloo)
    of tests/test_train_step_gpu.py::test_two_ranks_times_two_samples_equal_one_rank_times_four."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, OMP_NUM_THREADS="2", RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DDP_DEVICE="cpu")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_equiv_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    o0, o1 = torch.load(tmp_path / "out0.pt"), torch.load(tmp_path / "out1.pt")
    assert torch.equal(o0["grad"], o1["grad"])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _ddp_equiv_worker as W
    vdm, net, batch = W.build("fp32", "cpu")
    loss = W.loss_of(vdm, batch, slice(0, 4))
    loss.backward()
    g1 = net.flat.grad.detach()
    gmax = g1.abs().max().item()
    assert (o0["grad"] - g1).abs().max().item() <= 1e-5 * gmax
    assert abs(float(loss) - 0.5 * (o0["loss"] + o1["loss"])) <= 1e-5 * abs(float(loss))
    assert (2.0 * o0["grad"] - g1).abs().max().item() > 1e-2 * gmax          # (a SUM without the division would be this far off)
