"""Worker / shared builder for the data-parallel EQUIVALENCE tests (tests/test_train_step_gpu.py::test_two_ranks_times_two_samples_...,
tests/test_distributed_cpu.py::test_ddp_gradient_equals_full_batch_gloo_world2): rank r of 2 computes the VDM loss on samples
[2r, 2r + 2) of ONE seeded 4-sample batch with supplied times / noise and lets the product average the gradient over the ranks
(HIP backend: the bucketed all-reduce inside the backward; torch backend: trainer.allreduce_mean_).  The test process then runs the
same 4 samples in one piece and compares.  DDP_DEVICE = cuda (HIP backend, every rank on GPU 0 over gloo) | cpu (torch backend)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(precision, device):
    from helpers import grf
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import LightVDM
    torch.manual_seed(42)
    hip = device != "cpu"
    D = 16
    net = CUNet(shape=(1, D, D, D), chs=[32, 64] if precision == "bf16" else [16, 32], s_conditioning_channels=1, v_conditioning_dims=[6],
                norm_groups=8, dropout_prob=0.0, backend="hip" if hip else "torch", precision=precision if hip else "fp32")
    net.reset_parameters(generator=torch.Generator().manual_seed(42), zero_init_std=0.05)
    vdm = LightVDM(score_model=net, gamma_max=13.3, learning_rate=1e-3).to(device).train()
    g = torch.Generator().manual_seed(11)
    batch = {"x": grf((4, 1, D, D, D), 3), "s": grf((4, 1, D, D, D), 4), "v": torch.rand(4, 6, generator=g),
             "times": torch.tensor([0.07, 0.41, 0.66, 0.93]), "eps": grf((4, 1, D, D, D), 50, slope=0.0), "eps0": grf((4, 1, D, D, D), 51, slope=0.0)}
    return vdm, net, {k: v.to(device) for k, v in batch.items()}


def loss_of(vdm, batch, sl):
    return vdm.model.get_loss(batch["x"][sl], times=batch["times"][sl], eps=batch["eps"][sl], eps0=batch["eps0"][sl],
                              s_conditioning=batch["s"][sl], v_conditionings=[batch["v"][sl]])[0]


def main():
    import torch.distributed as dist
    from vdm4cdm_amd.trainer import allreduce_mean_, init_distributed
    out = sys.argv[1]
    device = os.environ.get("DDP_DEVICE", "cuda")
    rank, local, world = init_distributed(device)
    assert world == 2
    dev = f"cuda:{local}" if device == "cuda" else "cpu"
    vdm, net, batch = build(os.environ.get("DDP_PRECISION", "fp32"), dev)
    net.enable_ddp(world)                                       # HIP backend: gradient buckets all-reduced inside the backward
    loss = loss_of(vdm, batch, slice(2 * rank, 2 * rank + 2))
    loss.backward()
    synced = getattr(net, "grad_synced", False)
    if not synced:                                              # what Trainer.fit does with a gradient the backward did not average
        allreduce_mean_(net.flat.grad, world)
    if device == "cuda":
        torch.cuda.synchronize()
    torch.save({"grad": net.flat.grad.detach().cpu(), "loss": float(loss), "bucketed": bool(synced)}, os.path.join(out, f"out{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
