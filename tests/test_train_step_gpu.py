"""GPU parity tests of the BENCHMARKED training step (bench.py config C3: dropout 0.1, batch 2, clip 0.5, fused AdamW, data parallel
over ranks) against the CPU oracle (oracle/) - the holes the round-3 review listed:

* dropout ON at network level: the keep bits the HIP forward drew (one byte per 16-byte piece, written by gn_silu_fwd) are exported
  and handed to ``unet_oracle.cunet_forward(drop_masks=...)``; output and EVERY parameter gradient are compared (32^3 fp32 / bf16 and
  the C3 size 128^3, batch 2, bf16);
* one full optimizer step: K10 ``vdm_sumsq`` + ``vdm_clip_scale`` + fused AdamW against ``torch.nn.utils.clip_grad_norm_(0.5)`` +
  ``torch.optim.AdamW`` on the CPU, norms above and below the threshold, two steps (AdamW's first step is scale invariant);
* ``vdm_train_scalars`` with a device-side u0 for (rank, world) in {(0, 1), (1, 2), (5, 8)} against the oracle's antithetic grid;
* C2 (64^3, fp32) backward at its batch size 2;
* data-parallel equivalence: 2 ranks x 2 samples (gloo, one GPU) produce the gradient of 1 rank x 4 samples.

Reference call chain: /root/reference/trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:45 (gradient_clip_val=0.5), :64 (dropout_prob=0.1),
:131 (learning_rate=3e-4).
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

from helpers import grf, oracle_cfg, oracle_params
from test_unet_gpu import CFGS, DEV, _oracle_grads, _per_tensor_report, _product_grad_views, hip_forward, inputs, make_net, make_vdm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------ (a) dropout on, network level
def export_keep_masks(net, p):
    """The dropout keep bits of the LAST training-mode forward as oracle drop masks: {block prefix: [N, C, D, H, W] fp32 = keep / (1 - p)}.
    keep byte k of voxel v = the 16-byte piece k of the activated tensor a2 (EPL channels), bit j = channel k * EPL + j."""
    masks = {}
    for name, blk in net._exec.res.items():
        assert blk.saved is not None, f"{name}: no saved activations (forward did not run in training mode)"
        a2, mask = blk.saved[6], blk.saved[9]
        assert mask is not None, f"{name}: the forward wrote no keep bytes (fused GroupNorm backward off?)"
        n, C = a2.shape[0], a2.shape[-1]
        epl = 16 // a2.element_size()
        m = mask.view(n, -1, C // epl).to(torch.int32)                              # [N, V, pieces]
        bits = (m.unsqueeze(-1) >> torch.arange(epl, device=m.device, dtype=torch.int32)) & 1       # [N, V, pieces, EPL]
        keep = bits.reshape(n, *a2.shape[1:-1], C).permute(0, 4, 1, 2, 3).float()
        masks[name] = (keep / (1.0 - p)).cpu()
    return masks


def _dropout_case(net, B, seed, w_seed=77):
    """HIP training-mode forward + backward with dropout, masks exported in between; returns (y, gflat, masks, inputs, w)."""
    x, t, s, v = inputs(net, B, seed=seed)
    w = grf((B, 1) + net.shape[1:], w_seed) + 0.5
    net.zero_grad()
    y = hip_forward(net, x, t, s, v)
    masks = export_keep_masks(net, net.dropout_prob)
    (y * w.to(DEV)).sum().backward()
    return y.detach().cpu(), net.flat.grad.detach().cpu().clone(), masks, (x, t, s, v), w


def _oracle_grads_masked(net, x, t, s, v, w, masks):
    from oracle import unet_oracle
    p = {k: a.clone().requires_grad_(True) for k, a in oracle_params(net).items()}
    y = unet_oracle.cunet_forward(p, oracle_cfg(net), x, t, s, v, drop_masks=masks)
    (y * w).sum().backward()
    return y.detach(), {k: a.grad for k, a in p.items()}


def test_dropout_on_forward_backward_fp32_32cube():
    """fp32, 32^3, chs 32..256 (the C3 channel plan), batch 2, dropout 0.1: same masks on both sides -> the fp32 tolerances of
    test_unet_backward_fp32 (forward 2e-4, gradients 2e-3 of max|ref| per tensor)."""
    net = make_net(precision="fp32", dropout=0.1, **CFGS[2]).to(DEV).train()
    y, gflat, masks, (x, t, s, v), w = _dropout_case(net, 2, seed=3)
    frac = torch.cat([m.flatten() for m in masks.values()]).eq(0).float().mean().item()
    assert 0.08 < frac < 0.12, f"dropped fraction {frac}"
    yr, gref = _oracle_grads_masked(net, x, t, s, v, w, masks)
    assert (y - yr).abs().max().item() <= 2e-4 * yr.abs().max().item()
    got = _product_grad_views(net, gflat)
    bad = [(k, (got[k] - g).abs().max().item(), g.abs().max().item()) for k, g in gref.items()
           if g is not None and (got[k] - g).abs().max().item() > 2e-3 * max(g.abs().max().item(), 1e-8) + 1e-6]
    assert not bad, f"{len(bad)} tensors off: {bad[:8]}"
    # and the masks matter: the oracle WITHOUT them is far away
    from oracle import unet_oracle
    y0 = unet_oracle.cunet_forward(oracle_params(net), oracle_cfg(net), x, t, s, v)
    assert (y - y0).abs().max().item() > 50 * (y - yr).abs().max().item()


def test_dropout_on_forward_backward_bf16_32cube():
    net = make_net(precision="bf16", dropout=0.1, **CFGS[2]).to(DEV).train()
    y, gflat, masks, (x, t, s, v), w = _dropout_case(net, 2, seed=3)
    yr, gref = _oracle_grads_masked(net, x, t, s, v, w, masks)
    assert (y - yr).abs().max().item() <= 3e-2 * yr.abs().max().item()
    rows = _per_tensor_report(net, gflat, gref)
    bad = [(k, round(c, 5), round(r, 4)) for k, c, r, n in rows if c < 0.995]
    assert not bad, f"{len(bad)}/{len(rows)} gradient tensors off (name, cosine, max-rel): {bad[:10]}"


def test_c3_dropout_on_batch2_128_bf16():
    """The workload bench.py times: 128^3, chs 32..256, bf16 storage, batch 2, dropout 0.1 - forward and every parameter gradient.
    The HIP side runs the batch of 2 in one pass; the oracle takes the two samples one after the other (GroupNorm is per sample, the
    parameter gradient is the sum over samples: an exact identity) so that the CPU autograd graph stays at the batch-1 footprint.
    Tolerances of test_c3_backward_128_bf16: per-tensor cosine >= 0.999, max|d| <= 3e-2 max|ref|."""
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    net = make_net(D=128, chs=(32, 64, 128, 256), precision="bf16", dropout=0.1, seed=3).to(DEV).train()
    y, gflat, masks, (x, t, s, v), w = _dropout_case(net, 2, seed=5)
    gsum, ys = None, []
    for n in range(2):
        sl = slice(n, n + 1)
        yr, gref = _oracle_grads_masked(net, x[sl], t[sl], s[sl], [a[sl] for a in v], w[sl], {k: m[sl] for k, m in masks.items()})
        ys.append(yr)
        gsum = gref if gsum is None else {k: (gsum[k] + g if g is not None else None) for k, g in gref.items()}
        del gref
    yr = torch.cat(ys)
    assert (y - yr).abs().max().item() <= 3e-2 * yr.abs().max().item()
    rows = _per_tensor_report(net, gflat, gsum)
    bad = [(k, round(c, 5), round(r, 4)) for k, c, r, n in rows if c < 0.999 or r > 3e-2]
    assert not bad, f"{len(bad)}/{len(rows)} gradient tensors off (name, cosine, max-rel): {bad[:10]}"


# ------------------------------------------------------------------------------------------ (d) C2 at its batch size
def test_c2_backward_batch2_fp32():
    """BASELINE config C2 (64^3, chs 32..256, fp32 storage) at batch 2: all parameter gradients vs torch.autograd through the oracle."""
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    net = make_net(D=64, chs=(32, 64, 128, 256), precision="fp32", seed=9).to(DEV).train()
    x, t, s, v = inputs(net, 2, seed=13)
    w = grf((2, 1) + net.shape[1:], 77) + 0.5
    net.zero_grad()
    y = hip_forward(net, x, t, s, v)
    (y * w.to(DEV)).sum().backward()
    y, gflat = y.detach().cpu(), net.flat.grad.detach().cpu().clone()
    yr, gref = _oracle_grads(net, x, t, s, v, w)
    assert (y - yr).abs().max().item() <= 2e-4 * yr.abs().max().item()
    got = _product_grad_views(net, gflat)
    bad = [(k, (got[k] - g).abs().max().item(), g.abs().max().item()) for k, g in gref.items()
           if g is not None and (got[k] - g).abs().max().item() > 3e-3 * max(g.abs().max().item(), 1e-8) + 1e-6]
    assert not bad, f"{len(bad)} tensors off: {bad[:8]}"


# ------------------------------------------------------------------------------------------ (b) clip + AdamW
def _adamw_cpu(params0, grads, lr, clip, exact_norm=False):
    """torch reference on the CPU: clip_grad_norm_(clip) + AdamW(lr) (defaults of the product: betas (0.9, 0.999), eps 1e-8, wd 1e-2),
    one step per gradient in `grads`.  exact_norm: the clip coefficient min(1, clip / (norm + 1e-6)) from the fp64 norm instead of
    torch's fp32 CPU norm (which is itself only good to ~2e-5 over 10^6 terms) - the same formula, evaluated exactly.
    Returns (clipped gradients, parameter vectors after each step, norms)."""
    p = torch.nn.Parameter(params0.clone())
    opt = torch.optim.AdamW([p], lr=lr)
    clipped, after, norms = [], [], []
    for g in grads:
        p.grad = g.clone()
        if exact_norm:
            nrm = g.double().norm().item()
            p.grad.mul_(min(1.0, clip / (nrm + 1e-6)))
            norms.append(nrm)
        else:
            norms.append(float(torch.nn.utils.clip_grad_norm_([p], clip)))
        clipped.append(p.grad.clone())
        opt.step()
        after.append(p.detach().clone())
    return clipped, after, norms


@pytest.mark.parametrize("targets", [(0.2, 60.0), (35.0, 0.3)], ids=["below_then_above", "above_then_below"])
def test_clip_and_fused_adamw_match_torch(targets):
    """One optimizer step of the product = K10 sum of squares -> vdm_clip_scale (coefficient min(1, 0.5 / (norm + 1e-6)) derived on the
    device) -> fused AdamW over the flat vector.  Two consecutive steps whose gradient norms sit on different sides of the 0.5 threshold
    (AdamW's FIRST step is invariant to the gradient's scale: only the second one sees a wrong coefficient).
    (1) same gradients on both sides (the HIP gradient copied to the CPU): clipped gradient and updated parameters <= 1e-6 relative;
    (2) the oracle's gradients through torch's clip + AdamW: the update agrees to 2e-2 * lr wherever |g| is not noise."""
    from vdm4cdm_amd.trainer import clip_grad_norm_flat_
    lr, clip = 3.0e-4, 0.5
    net = make_net(precision="fp32", **CFGS[0]).to(DEV).train()
    vdm = make_vdm(net).to(DEV).train()
    params = [p for p in vdm.parameters() if p.requires_grad]
    assert len(params) == 1 and params[0] is net.flat
    opt = vdm.configure_optimizers()
    flat0 = net.flat.detach().cpu().clone()
    hip_grads, ora_grads, hip_clipped, hip_after, hip_norms = [], [], [], [], []
    for step, target in enumerate(targets):                      # gradient norms 0.2 / 60 (35 / 0.3): on either side of the 0.5 threshold
        x, t, s, v = inputs(net, 2, seed=3 + 10 * step)
        w = grf((2, 1) + net.shape[1:], 77 + step) + 0.5
        opt.zero_grad(set_to_none=True)                          # (the gradient is linear in w: probe its norm, then scale w)
        (hip_forward(net, x, t, s, v) * w.to(DEV)).sum().backward()
        w = w * (target / net.flat.grad.norm().item())
        # the oracle's gradient at the CURRENT product weights
        _, gref = _oracle_grads(net, x, t, s, v, w)
        ora_grads.append(gref)
        opt.zero_grad(set_to_none=True)
        y = hip_forward(net, x, t, s, v)
        (y * w.to(DEV)).sum().backward()
        hip_grads.append(net.flat.grad.detach().cpu().clone())
        nrm = clip_grad_norm_flat_(params, clip, use_hip=True, want_norm=True)
        hip_norms.append(float(nrm))
        hip_clipped.append(net.flat.grad.detach().cpu().clone())
        opt.step()
        hip_after.append(net.flat.detach().cpu().clone())
    # (1) identical gradients in, torch arithmetic on the CPU
    clipped, after, norms = _adamw_cpu(flat0, hip_grads, lr, clip, exact_norm=True)
    clipped_t, _, norms_t = _adamw_cpu(flat0, hip_grads, lr, clip)           # torch.nn.utils.clip_grad_norm_ itself (fp32 norm)
    assert (norms[0] > clip) != (norms[1] > clip), f"the two steps must straddle the threshold: norms {norms}"
    prev = flat0
    for k in range(2):
        assert hip_norms[k] == pytest.approx(norms[k], rel=1e-5) and hip_norms[k] == pytest.approx(norms_t[k], rel=1e-4)
        assert (hip_clipped[k] - clipped_t[k]).abs().max().item() <= 1e-4 * clipped_t[k].abs().max().item()
        gmax = clipped[k].abs().max().item()
        assert (hip_clipped[k] - clipped[k]).abs().max().item() <= 1e-6 * gmax, f"step {k}: clipped gradient"
        d_hip, d_ref = hip_after[k] - prev, after[k] - prev
        assert (d_hip - d_ref).abs().max().item() <= 1e-6 * max(after[k].abs().max().item(), 1.0) and \
            (d_hip - d_ref).abs().max().item() <= 2e-3 * lr, f"step {k}: parameter update"
        prev = hip_after[k]
    # (2) the oracle's gradients, brought into the product's flat layout by the product's own checkpoint loader (torch conv layout
    # [cout, cin, k, k, k] -> tap-major, concatenated skip weight -> skip / skip2)
    def flatten(gref):
        from vdm4cdm_amd.networks import CUNet
        shell = CUNet(shape=net.shape, chs=list(net.chs), s_conditioning_channels=net.s_conditioning_channels,
                      v_conditioning_dims=list(net.v_conditioning_dims), norm_groups=net.norm_groups, backend="torch")
        with torch.no_grad():
            shell.flat.zero_()
        shell.load_state_dict({k: g for k, g in gref.items() if g is not None}, strict=False)
        return shell.flat.detach().clone()
    og = [flatten(g) for g in ora_grads]
    for k in range(2):                                          # the HIP gradient is the oracle's (fp32 tolerance of test_unet_backward_fp32)
        assert (og[k] - hip_grads[k]).abs().max().item() <= 2e-3 * hip_grads[k].abs().max().item()
    _, after_o, norms_o = _adamw_cpu(flat0, og, lr, clip)
    assert norms_o[0] == pytest.approx(norms[0], rel=2e-3) and norms_o[1] == pytest.approx(norms[1], rel=2e-3)
    sig = (og[0].abs() > 1e-3 * og[0].abs().max()) & (og[1].abs() > 1e-3 * og[1].abs().max())
    assert sig.float().mean().item() > 0.2
    err = ((hip_after[1] - flat0) - (after_o[1] - flat0)).abs()[sig].max().item()
    assert err <= 2e-2 * lr, f"two optimizer steps on the oracle's gradients: update differs by {err / lr:.3g} lr"


# ------------------------------------------------------------------------------------------ (c) stratified time grid
@pytest.mark.parametrize("rank,world", [(0, 1), (1, 2), (5, 8)], ids=["r0w1", "r1w2", "r5w8"])
def test_train_scalars_stratified_grid_matches_oracle(rank, world):
    """vdm_train_scalars with a device-side u0: rank r takes strata r*B .. r*B+B-1 of the world*B strata of the GLOBAL batch,
    t_i = (u0 + (r*B + i) / (world*B)) mod 1 - the oracle's antithetic grid over world*B samples, sliced - and alpha / sigma / loss
    weight / normalised time of the fixed linear schedule from the oracle's fp64 formulas."""
    from oracle import vdm_oracle
    from vdm4cdm_amd import hip_ops as ops
    B, gmin, gmax = 2, -13.3, 13.3
    numel = 32 ** 3
    bpd = 1.0 / (numel * torch.log(torch.tensor(2.0, dtype=torch.float64)).item())
    sched = vdm_oracle.Schedule(gmin, gmax)
    for u in (0.0, 0.3183099, 0.93, 0.9999999):
        u0 = torch.tensor([u], dtype=torch.float32, device=DEV)
        sc = ops.train_scalars(B, DEV, rank, world, gmin, gmax, bpd / B, u0=u0).cpu().double()
        tref = vdm_oracle.antithetic_times(float(u0.item()), world * B)[rank * B:(rank + 1) * B]
        d = (sc[0] - tref).abs()
        d = torch.minimum(d, 1.0 - d)                           # (a stratum that lands exactly on the wrap is the same point of the circle)
        assert d.max().item() <= 2e-7, (u, sc[0], tref)
        t = sc[0]                                               # the schedule scalars at the kernel's own t (fp32 grid)
        g = sched.gamma(t)
        assert (sc[1] - sched.alpha(g)).abs().max().item() <= 2e-6 * max(sched.alpha(g).max().item(), 1e-3)
        assert (sc[2] - sched.sigma(g)).abs().max().item() <= 2e-6
        assert (sc[3] - sched.dgamma_dt(t) * bpd / B).abs().max().item() <= 1e-6 * (gmax - gmin) * bpd / B
        assert (sc[4] - (g - gmin) / (gmax - gmin)).abs().max().item() <= 2e-7
    # the strata of all ranks tile [0, 1): one sample per stratum of width 1 / (world * B)
    u0 = torch.tensor([0.4242], dtype=torch.float32, device=DEV)
    allt = torch.cat([ops.train_scalars(B, DEV, r, world, gmin, gmax, bpd / B, u0=u0)[0].cpu() for r in range(world)])
    strata = torch.floor(torch.remainder(allt.double() - 0.4242, 1.0) * world * B + 0.5).long() % (world * B)
    assert sorted(strata.tolist()) == list(range(world * B))


# ------------------------------------------------------------------------------------------ (e) data-parallel equivalence
def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_ranks_times_two_samples_equal_one_rank_times_four(tmp_path, precision):
    """per-sample GroupNorm makes data parallelism an exact identity up to summation order - 2 ranks x B = 2 with the
    gradient averaged over the ranks (the bucketed all-reduce inside the backward; gloo on the one GPU of the box) must equal
    1 rank x B = 4 on the same samples / times / noise.  Catches what "the ranks agree with each other" cannot: a SUM without the
    division, an AVG of already divided gradients, a wrong per-rank loss weight."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VDM4CDM_SHARE_GPU="1", VDM4CDM_DIST_BACKEND="gloo", DDP_PRECISION=precision, DDP_DEVICE="cuda")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_equiv_worker.py"), str(tmp_path)], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    o0, o1 = torch.load(tmp_path / "out0.pt"), torch.load(tmp_path / "out1.pt")
    assert o0["bucketed"] and o1["bucketed"] and torch.equal(o0["grad"], o1["grad"])
    # the single-process run on the full batch, here
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _ddp_equiv_worker as W
    vdm, net, batch = W.build(precision, "cuda")
    loss = W.loss_of(vdm, batch, slice(0, 4))
    net.flat.grad = None
    loss.backward()
    g1 = net.flat.grad.detach().cpu()
    gmax = g1.abs().max().item()
    tol = 1e-5 if precision == "fp32" else 2e-2                # bf16: the saved activations are the same, the tile sums are re-associated
    err = (o0["grad"] - g1).abs().max().item()
    assert err <= tol * gmax, f"2 x 2 vs 1 x 4: max|d| {err:.3e} vs max|g| {gmax:.3e}"
    assert float(loss) == pytest.approx(0.5 * (o0["loss"] + o1["loss"]), rel=1e-5)
    if precision == "fp32":                                    # the wrong pairings are far outside the tolerance
        assert (2.0 * o0["grad"] - g1).abs().max().item() > 100 * tol * gmax


# ------------------------------------------------------------------------------------------ fused head of the step (K7 + input packing, K8 with regenerated noise)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("D", [12, 20, 32], ids=["12cube", "20cube", "32cube"])
def test_diffuse_pack_equals_randn_diffuse_pack_input(dtype, D):
    """vdm_diffuse_pack (eps drawn in the kernel) == vdm_randn -> vdm_diffuse -> vdm_pack_input, bit for bit: z_t and conv_in's packed
    NDHWC input, ragged sizes (the packed pieces are re-dealt across the lanes of a wave: tails matter), with and without s_conditioning;
    with supplied eps it equals the three-kernel chain on that eps; alpha / sigma per sample against the oracle formula."""
    from vdm4cdm_amd import hip_ops as ops
    B = 3
    x = grf((B, 1, D, D, D), 5).to(DEV)
    s = grf((B, 1, D, D, D), 6).to(DEV)
    al, si = torch.tensor([0.9, 0.5, 0.1], device=DEV), torch.tensor([0.3, 0.8, 0.99], device=DEV)
    seed, sid = 123456789012345, 7
    eps = ops.randn(torch.empty_like(x), seed, sid)
    zref = ops.diffuse(x, eps, al, si)
    assert torch.allclose(zref.cpu(), (al.view(B, 1, 1, 1, 1) * x + si.view(B, 1, 1, 1, 1) * eps).cpu(), atol=1e-6)      # the oracle's z_t
    for sc in (s, None):
        pref = ops.pack_input(zref.reshape(B, D, D, D), None if sc is None else sc.reshape(B, D, D, D), dtype)
        z, packed = ops.diffuse_pack(x, sc, al, si, dtype, seed=seed, stream_id=sid, want_z=True)
        assert torch.equal(z, zref) and packed.shape == pref.shape and torch.equal(packed.view(torch.uint8), pref.view(torch.uint8))
        z2, p2 = ops.diffuse_pack(x, sc, al, si, dtype, eps=eps, want_z=True)
        assert torch.equal(z2, zref) and torch.equal(p2.view(torch.uint8), pref.view(torch.uint8))
        z3, p3 = ops.diffuse_pack(x, sc, al, si, dtype, seed=seed, stream_id=sid + 1, want_z=False)
        assert z3 is None and not torch.equal(p3.view(torch.uint8), pref.view(torch.uint8))


def test_loss_terms_regenerates_its_noise_fields():
    """vdm_loss_terms_rng with eps / eps0 = NULL regenerates the fields vdm_randn would have written: sums and d_eps_hat equal the
    supplied-field call bit for bit, and both agree with the formula."""
    from vdm4cdm_amd import hip_ops as ops
    B, per = 3, 20 ** 3
    x, eh = grf((B, 1, 20, 20, 20), 1).reshape(B, per).to(DEV), grf((B, 1, 20, 20, 20), 2).reshape(B, per).to(DEV)
    coef = torch.tensor([0.5, 1.5, 2.0], device=DEV)
    rng = ((111, 3), (222, 4))
    eps, e0 = ops.randn(torch.empty_like(x), *rng[0]), ops.randn(torch.empty_like(x), *rng[1])
    out = []
    for a, b, r in ((eps, e0, None), (None, None, rng), (eps, None, rng), (None, e0, rng)):
        sums, d = torch.zeros(B, 3, device=DEV), torch.empty(B, per, device=DEV)
        ops.loss_terms(x, a, eh, b, 0.01, coef, sums, d, rng=r)
        out.append((sums.cpu(), d.cpu()))
    for sm, d in out[1:]:
        assert torch.equal(sm, out[0][0]) and torch.equal(d, out[0][1])
    ref = torch.stack([((eh - eps) ** 2).sum(1), (x ** 2).sum(1), ((0.01 * e0) ** 2).sum(1)], 1).cpu()
    assert torch.allclose(out[0][0], ref, rtol=1e-5) and torch.allclose(out[0][1], (coef[:, None] * (eh - eps)).cpu(), atol=1e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_head_training_step_equals_unfused(precision, monkeypatch):
    """VDM.get_loss with the fused head (no noise field in memory, z_t and conv_in's input from one pass) against the unfused chain
    (randn x 2 -> diffuse -> pack_input -> ... -> loss_terms on the stored fields) from the same generator state: the same loss, ELBO
    parts and parameter gradient, bit for bit (dropout on: the masks depend only on the seed counter)."""
    import vdm4cdm_amd.unet_hip as uh
    net = make_net(precision=precision, dropout=0.1, **CFGS[0])
    vdm = make_vdm(net).to(DEV).train()
    x, _, s, v = inputs(net, 2)
    kw = dict(s_conditioning=s.to(DEV), v_conditionings=[a.to(DEV) for a in v])
    res = []
    for fused in (True, False):
        monkeypatch.setattr(vdm_model_mod_ref(), "FUSED_HEAD", fused)
        torch.manual_seed(9)
        uh._seed_counter[0] = 0
        vdm_model_mod_ref().reset_train_generators()
        net.flat.grad = None
        loss, metrics = vdm.model.get_loss(x.to(DEV), **kw)
        loss.backward()
        res.append((loss.detach().clone(), {k: m.clone() for k, m in metrics.items()}, net.flat.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1])
    assert torch.equal(res[0][2], res[1][2]) and torch.isfinite(res[0][2]).all() and res[0][2].abs().max().item() > 0


def vdm_model_mod_ref():
    import vdm4cdm_amd.vdm_model as m
    return m
