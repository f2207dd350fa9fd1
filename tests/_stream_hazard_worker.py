"""Worker for tests/test_entry_gpu.py::test_streams_and_allocator_hazards: [training steps -> sampler -> rocFFT P(k) -> teardown] three
times in ONE process, every round with the same seeds, so every round must reproduce the first one BIT FOR BIT.  The product runs
three HIP streams (main, weight gradients, skip path) over buffers owned by torch's caching allocator; a buffer handed to a new owner
while a queued side-stream kernel still uses it shows up here as a changed number, a NaN or a fault in a later call.
HAZARD_MODE = nocache (PYTORCH_NO_CUDA_MEMORY_CACHING=1 in the environment: every free is a real hipFree - a stale use faults or reads
              unmapped memory; stream capture cannot allocate in this mode, so the sampler runs its eager loop)
            | cached  (default allocator, torch.cuda.empty_cache() between the phases: blocks move between streams as early as they can;
              hipGraph sampler and the graph-captured training step included)."""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one_round(mode, precision):
    import vdm4cdm_amd.unet_hip as uh
    from helpers import grf
    from vdm4cdm_amd import utils
    from vdm4cdm_amd import vdm_model as vm
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.trainer import GraphedTrainStep, clip_grad_norm_flat_
    dev = "cuda:0"
    cached = mode == "cached"
    flush = (lambda: (gc.collect(), torch.cuda.empty_cache())) if cached else (lambda: gc.collect())
    torch.manual_seed(7)
    uh._seed_counter[0] = 0
    vm.reset_train_generators()
    D = 32
    net = CUNet(shape=(1, D, D, D), chs=[32, 64, 128], s_conditioning_channels=1, v_conditioning_dims=[6], norm_groups=8, dropout_prob=0.1,
                backend="hip", precision=precision)
    net.reset_parameters(generator=torch.Generator().manual_seed(3), zero_init_std=0.05)
    vdm = vm.LightVDM(score_model=net, gamma_max=13.3, learning_rate=1e-3).to(dev).train()
    params = [p for p in vdm.parameters() if p.requires_grad]
    opt = vdm.configure_optimizers(capturable=cached)
    g = torch.Generator().manual_seed(5)
    batch = {"x": grf((2, 1, D, D, D), 3).to(dev), "conditioning": grf((2, 1, D, D, D), 4).to(dev),
             "conditioning_values": [torch.rand(2, 6, generator=g).to(dev)]}
    out = {}
    losses = []
    for _ in range(3):                                            # eager steps: three streams, deferred weight gradients, fused AdamW
        loss = vdm.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        clip_grad_norm_flat_(params, 0.5, use_hip=True, want_norm=False)
        opt.step()
        losses.append(loss.detach().clone())
        del loss
        flush()
    if cached:                                                    # the graph-captured step on top (its own side-stream forks under capture)
        gs = GraphedTrainStep(vdm, opt, params, 0.5, batch)
        for _ in range(3):
            losses.append(gs(batch).detach().clone())
        del gs
        flush()
    out["losses"] = torch.stack(losses).cpu()
    out["flat"] = net.flat.detach().cpu().clone()
    vdm.eval()
    z = vdm.draw_samples(batch_size=1, n_sampling_steps=6, seed=11, s_conditioning=batch["conditioning"][:1],
                         v_conditionings=[batch["conditioning_values"][0][:1]], use_graph=cached)
    flush()
    k, pk, n = utils.pk(z)                                        # rocFFT on the sampled cube
    out["z"], out["pk"] = z.cpu().clone(), pk.cpu().clone()
    torch.cuda.synchronize()
    del vdm, net, opt, params, batch, z, k, pk, n                 # teardown: executor, packed weights, workspaces, graphs
    flush()
    return out


def main():
    mode, precision = os.environ.get("HAZARD_MODE", "cached"), os.environ.get("HAZARD_PRECISION", "bf16")
    if mode == "nocache":
        assert os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1"
    rounds = [one_round(mode, precision) for _ in range(3)]
    for r in rounds:
        for k, v in r.items():
            assert torch.isfinite(v).all(), f"{k} is not finite"
    for i, r in enumerate(rounds[1:], 1):
        for k in r:
            if k == "pk":      # (torch's shell sums behind utils.power add with float atomics: equal to rounding, not bit for bit)
                assert torch.allclose(r[k], rounds[0][k], rtol=1e-5, atol=0.0), f"round {i}: pk differs from round 0 beyond rounding"
                continue
            assert torch.equal(r[k], rounds[0][k]), f"round {i}: {k} differs from round 0 (max|d| {(r[k] - rounds[0][k]).abs().max().item():.3e})"
    print(f"HAZARD_OK mode={mode} precision={precision} losses={rounds[0]['losses'].tolist()}")


if __name__ == "__main__":
    main()
