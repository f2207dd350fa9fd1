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
