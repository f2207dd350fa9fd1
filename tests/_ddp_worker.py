"""Worker for tests/test_distributed_cpu.py: 2 training steps of a tiny 2D VDM under torch.distributed (gloo)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vdm4cdm_amd.data import SyntheticAstroDataModule  # noqa: E402
from vdm4cdm_amd.networks import CUNet  # noqa: E402
from vdm4cdm_amd.trainer import Trainer, dist_env  # noqa: E402
from vdm4cdm_amd.vdm_model import LightVDM  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, _, world = dist_env()
    torch.manual_seed(100 + rank)                    # deliberately different init per rank: fit() must broadcast rank 0's
    net = CUNet(shape=(1, 16, 16), chs=[8, 16], s_conditioning_channels=0, v_conditioning_dims=[], norm_groups=4,
                dropout_prob=0.0, conv_padding_mode="circular", backend="torch")
    net.reset_parameters(zero_init_std=0.05)
    vdm = LightVDM(score_model=net, gamma_min=-13.3, gamma_max=13.3, noise_schedule="learned_linear")
    dm = SyntheticAstroDataModule(cropsize=16, batch_size=2, dim=2, conditioning=False, n_params=0, n_train=8,
                                  return_func=lambda f, p: {"x": f[1], "conditioning": None, "conditioning_values": None})
    max_steps = int(os.environ.get("DDP_MAX_STEPS", "2"))
    tr = Trainer(max_steps=max_steps, val_check_interval=int(os.environ.get("DDP_VAL_EVERY", "0")), gradient_clip_val=0.5,
                 every_n_train_steps=max_steps, default_root_dir=out_dir, experiment_name="ddp", device="cpu", enable_progress=False,
                 log_every_n_steps=1, limit_val_batches=2)
    tr.fit(vdm, dm)
    torch.save({"flat": net.flat.detach().clone(), "gamma_w": vdm.model.gamma_w.detach().clone(), "steps": tr.global_step,
                "history": tr.history}, os.path.join(out_dir, f"rank{rank}.pt"))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
