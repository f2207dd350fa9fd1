"""Worker for tests/test_entry_gpu.py::test_hip_training_two_ranks_share_one_gpu: rank RANK of WORLD_SIZE trains a small 3D VDM on the
HIP backend through Trainer.fit; every rank uses GPU 0 (VDM4CDM_SHARE_GPU=1) and the gloo backend (RCCL refuses two ranks on one
device), so the N > 1 code path - per-rank data shards and noise streams, weight broadcast, gradient buckets all-reduced from inside
the backward on the communication stream, clip after the last bucket - runs with HIP kernels and a real process group in one process."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vdm4cdm_amd.data import SyntheticAstroDataModule  # noqa: E402
from vdm4cdm_amd.networks import CUNet  # noqa: E402
from vdm4cdm_amd.trainer import Trainer  # noqa: E402
from vdm4cdm_amd.vdm_model import LightVDM  # noqa: E402


def main():
    out = sys.argv[1]
    rank = int(os.environ["RANK"])
    torch.manual_seed(42)
    precision = os.environ.get("DDP_PRECISION", "fp32")     # bf16: the fused skip / tail kernels run under the gradient buckets too
    net = CUNet(shape=(1, 16, 16, 16), chs=[16, 32] if precision == "fp32" else [32, 64], s_conditioning_channels=1, v_conditioning_dims=[6],
                norm_groups=8, dropout_prob=0.1, backend="hip", precision=precision)
    # rank-dependent initial weights: the broadcast from rank 0 must make them identical
    net.reset_parameters(generator=torch.Generator().manual_seed(42 + rank), zero_init_std=0.02)
    vdm = LightVDM(score_model=net, gamma_max=13.3, learning_rate=1e-3)
    dm = SyntheticAstroDataModule(cropsize=16, batch_size=2, n_train=16, seed=5)
    tr = Trainer(max_steps=3, val_check_interval=0, gradient_clip_val=0.5, every_n_train_steps=0, default_root_dir=out,
                 experiment_name=f"rank{rank}", device="cuda", enable_progress=False, log_every_n_steps=1)
    tr.fit(vdm, dm)
    torch.cuda.synchronize()
    torch.save({"flat": net.flat.detach().cpu(), "bucketed": net._exec.buckets is not None, "world": tr.world,
                "loss": [h["loss"] for h in tr.history if "loss" in h]}, os.path.join(out, f"out{rank}.pt"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
