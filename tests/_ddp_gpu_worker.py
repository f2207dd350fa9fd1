"""Worker for tests/test_entry_gpu.py::test_hip_training_step_under_rccl_world1: 2 training steps of a small 3D VDM on the HIP backend
through Trainer.fit inside an initialised `nccl` (= RCCL) process group of world size 1.  With VDM4CDM_FORCE_BUCKETS=1 the backward
pass runs the bucketed all-reduce path (4 collectives on the communication stream, overlapped with the backward kernels)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vdm4cdm_amd.data import SyntheticAstroDataModule  # noqa: E402
from vdm4cdm_amd.networks import CUNet  # noqa: E402
from vdm4cdm_amd.trainer import Trainer  # noqa: E402
from vdm4cdm_amd.vdm_model import LightVDM  # noqa: E402


def main():
    out = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[2])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    torch.manual_seed(42)
    net = CUNet(shape=(1, 16, 16, 16), chs=[16, 32], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=8,
                dropout_prob=0.1, backend="hip", precision="bf16")
    net.reset_parameters(generator=torch.Generator().manual_seed(42), zero_init_std=0.02)
    vdm = LightVDM(score_model=net, gamma_max=13.3, learning_rate=1e-3)
    dm = SyntheticAstroDataModule(cropsize=16, batch_size=2, n_train=8, seed=5)
    tr = Trainer(max_steps=2, val_check_interval=0, gradient_clip_val=0.5, every_n_train_steps=0, default_root_dir=out,
                 experiment_name="rccl1", device="cuda", enable_progress=False, log_every_n_steps=1)
    tr.fit(vdm, dm)
    ex = net._exec
    torch.cuda.synchronize()
    torch.save({"flat": net.flat.detach().cpu(), "bucketed": ex.buckets is not None, "bounds": net.bucket_bounds(),
                "loss": [h["loss"] for h in tr.history if "loss" in h]}, os.path.join(out, "out.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
