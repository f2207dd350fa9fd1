"""GPU parity tests: every HIP kernel (through the C-ABI) against the same torch.nn.functional primitive on
the CPU in fp32 (T1 of SURVEY.md section 8c).  Tolerances (stated per test):
  fp32 storage : |d| <= 1e-5 + 1e-5|ref| for GroupNorm / elementwise, <= 1e-4 * max|ref| for convolutions
  bf16 storage : inputs are rounded to bf16 first; outputs within 2^-7 * max|ref| (one bf16 rounding + fp32 accumulate)
"""
import math

import pytest
import torch
import torch.nn.functional as F


def _lib_mod():
    from vdm4cdm_amd import _lib
    return _lib

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]


def _ops():
    from vdm4cdm_amd import hip_ops
    return hip_ops


def rnd(shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * scale
    return x.to(dtype).float()          # value representable in `dtype`, held as fp32 on the CPU


def to_dev(x, dtype):
    return x.to(dtype).to(DEV).contiguous()


def conv_tol(dtype, ref):
    return (1e-4 if dtype == torch.float32 else 2.0 ** -7) * max(ref.abs().max().item(), 1e-6)


def ref_conv(x, w, bias, nbias, res, ks, stride, ups, circular):
    """x: [N,D,H,W,Cin] fp32 ; w: [T,Cout,Cin] ; returns NDHWC fp32."""
    xc = x.permute(0, 4, 1, 2, 3)
    if ups:
        xc = F.interpolate(xc, scale_factor=2, mode="nearest")
    cout, cin = w.shape[1], w.shape[2]
    wt = w.view(ks, ks, ks, cout, cin).permute(3, 4, 0, 1, 2)
    pad = ks // 2
    if pad and circular:
        y = F.conv3d(F.pad(xc, (pad,) * 6, mode="circular"), wt, bias, stride=stride)
    else:
        y = F.conv3d(xc, wt, bias, stride=stride, padding=pad)
    y = y.permute(0, 2, 3, 4, 1)
    if nbias is not None:
        y = y + nbias[:, None, None, None, :]
    if res is not None:
        y = y + res
    return y


CONV_CASES = [
    # name, N, (D,H,W) of the OUTPUT, cin, cout, ks, stride, ups, circular
    ("k3_32_32", 2, (8, 8, 16), 32, 32, 3, 1, 0, False),
    ("k3_32_32_ragged", 1, (6, 9, 20), 32, 32, 3, 1, 0, False),
    ("k3_64_32", 1, (4, 8, 16), 64, 32, 3, 1, 0, False),
    ("k3_32_64", 1, (4, 8, 16), 32, 64, 3, 1, 0, False),
    ("k3_128_256", 1, (4, 4, 16), 128, 256, 3, 1, 0, False),
    ("k3_cin2pad", 2, (8, 8, 16), 2, 32, 3, 1, 0, False),
    ("k3_cout1", 2, (8, 8, 16), 32, 1, 3, 1, 0, False),
    ("k3_mid_64_64", 2, (32, 32, 32), 64, 64, 3, 1, 0, False),      # 256 tiles of 2x8x16: the middle tile shape (bf16)
    ("k3_mid_32_128", 1, (16, 32, 32), 32, 128, 3, 1, 0, True),      # 128 tiles of 1x8x16 (2 chunks)
    ("k3_mid_128_64", 2, (16, 32, 32), 128, 64, 3, 1, 0, False),     # 4 K-blocks, 1x8x16 tiles, half-chunk workgroups
    ("k3_mid_64_128", 2, (32, 32, 32), 64, 128, 3, 1, 0, True),      # 2 K-blocks, 2x8x16 tiles, whole chunks
    ("k3_big_64_64", 2, (64, 64, 64), 64, 64, 3, 1, 0, False),       # 2 K-blocks, 4x8x16 tiles, 4 workgroups per CU
    ("k3_cin8_circ", 1, (6, 9, 20), 8, 16, 3, 1, 0, True),           # <= 8 input channels: tap-packed kernel (bf16)
    ("k3_cin3_ragged", 2, (5, 7, 18), 3, 32, 3, 1, 0, False),
    ("k3_circ", 1, (8, 8, 16), 32, 32, 3, 1, 0, True),
    ("k3_circ_small", 1, (2, 4, 6), 32, 32, 3, 1, 0, True),
    ("k3_s2", 2, (4, 4, 8), 32, 32, 3, 2, 0, False),
    ("k3_s2_circ", 1, (4, 6, 8), 64, 64, 3, 2, 0, True),
    ("k3_ups", 1, (8, 8, 16), 64, 32, 3, 1, 1, False),
    ("k3_ups_circ", 1, (4, 8, 12), 32, 16, 3, 1, 1, True),
    ("k3_ups_ragged", 2, (12, 20, 40), 64, 32, 3, 1, 1, False),
    ("k3_ups_128_64", 1, (8, 16, 32), 128, 64, 3, 1, 1, False),
    ("k3_s2_ragged", 1, (6, 10, 20), 32, 32, 3, 2, 0, False),
    ("k3_s2_128", 1, (4, 8, 16), 128, 128, 3, 2, 0, False),
    ("k1_64_32", 2, (8, 8, 16), 64, 32, 1, 1, 0, False),
    ("k1_32_128", 1, (4, 8, 12), 32, 128, 1, 1, 0, False),
    ("k3_16_16", 1, (8, 8, 16), 16, 16, 3, 1, 0, False),
    ("k3_48_96", 1, (4, 8, 16), 48, 96, 3, 1, 0, False),
    # deepest level: the K-split kernel (waves of a workgroup split the K-blocks; bf16, >= 4 K-blocks, 64-cout chunks, 1 x TY x 16 tiles)
    ("k3_deep_256_256", 1, (16, 16, 16), 256, 256, 3, 1, 0, False),    # 8 K-blocks (2 per wave), 1x4x16 tiles, the level-3 conv at batch 1
    ("k3_deep_160_128_circ", 2, (3, 6, 20), 160, 128, 3, 1, 0, True),  # 5 K-blocks (2,1,1,1 per wave), ragged tiles, circular
    ("k3_deep_256_64", 2, (8, 16, 16), 256, 64, 3, 1, 0, False),       # 1x8x16 tiles (8 rows per wave in the main loop)
]


def _conv_setup(case, dtype, seed=0):
    ops = _ops()
    name, N, (D, H, W), cin, cout, ks, stride, ups, circ = case
    if stride == 2:
        ishape = (N, 2 * D, 2 * H, 2 * W, cin)
    elif ups:
        ishape = (N, D // 2, H // 2, W // 2, cin)
    else:
        ishape = (N, D, H, W, cin)
    x = rnd(ishape, seed + 1, dtype)
    w = rnd((ks ** 3, cout, cin), seed + 2, dtype, scale=1.0 / math.sqrt(ks ** 3 * cin))
    conv = ops.Conv(cin, cout, ks, stride=stride, upsample=ups, circular=circ)
    conv.pack(w.to(DEV), dtype, need_dgrad=(stride == 1))
    cp = ops.cpad(cin, dtype)
    xd = torch.zeros(ishape[:-1] + (cp,), dtype=dtype, device=DEV)
    xd[..., :cin] = x.to(dtype).to(DEV)
    return ops, conv, x, w, xd


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd(case, dtype):
    ops, conv, x, w, xd = _conv_setup(case, dtype)
    name, N, (D, H, W), cin, cout, ks, stride, ups, circ = case
    bias = rnd((cout,), 5)
    nbias_full = rnd((N, cout + 7), 6)                     # strided view: row stride cout+7
    res = rnd((N, D, H, W, cout), 7, dtype)
    nb_dev = None if ups else nbias_full.to(DEV)[:, 3:3 + cout]         # (the up-sampling conv has no conditioning bias)
    out = conv.fwd(xd, bias.to(DEV), nb_dev, to_dev(res, dtype))
    ref = ref_conv(x, w, bias, None if ups else nbias_full[:, 3:3 + cout], res, ks, stride, ups, circ)
    err = (out.float().cpu() - ref).abs().max().item()
    assert out.shape == ref.shape
    assert err <= conv_tol(dtype, ref), f"{name}: max err {err} > {conv_tol(dtype, ref)}"
    # plain variant without epilogue terms
    out2 = conv.fwd(xd)
    ref2 = ref_conv(x, w, None, None, None, ks, stride, ups, circ)
    err2 = (out2.float().cpu() - ref2).abs().max().item()
    assert err2 <= conv_tol(dtype, ref2), f"{name} (no epilogue): max err {err2}"


def test_conv_out_f32():
    """conv_out writes fp32 eps_hat from bf16 activations (cout = 1)."""
    case = ("k3_cout1", 2, (8, 8, 16), 32, 1, 3, 1, 0, False)
    dtype = torch.bfloat16
    ops = _ops()
    x = rnd((2, 8, 8, 16, 32), 1, dtype)
    w = rnd((27, 1, 32), 2, dtype, scale=0.05)
    conv = ops.Conv(32, 1, 3, out_f32=True)
    conv.pack(w.to(DEV), dtype, need_dgrad=False)
    out = conv.fwd(to_dev(x, dtype), rnd((1,), 3).to(DEV))
    assert out.dtype == torch.float32
    ref = ref_conv(x, w, rnd((1,), 3), None, None, 3, 1, 0, False)
    assert (out.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-5


LARGE_CASES = [
    # large level-0 shapes (thousands of workgroups: several rounds per CU, uneven XCD split, volume boundaries of every kind)
    ("res_32_32", 2, (64, 64, 64), 32, 32, 3, 1, 0, False),
    ("res_uneven", 3, (52, 56, 64), 32, 32, 3, 1, 0, True),      # 1092 tiles: uneven split over XCDs and workgroups
    ("res_ragged_24_32", 3, (50, 60, 70), 24, 32, 3, 1, 0, False),
    ("res_circ", 2, (64, 56, 64), 32, 32, 3, 1, 0, True),
    ("res_cin2", 2, (64, 64, 64), 2, 32, 3, 1, 0, False),
    ("res_cout1", 2, (64, 64, 64), 32, 1, 3, 1, 0, False),
]


@pytest.mark.parametrize("case", LARGE_CASES, ids=[c[0] for c in LARGE_CASES])
def test_conv_large(case):
    """Level-0 sized grids: forward (all epilogue terms), input gradient and gradient accumulation."""
    dtype = torch.bfloat16
    ops, conv, x, w, xd = _conv_setup(case, dtype, seed=30)
    name, N, (D, H, W), cin, cout, ks, stride, ups, circ = case
    bias = rnd((cout,), 31)
    nbias = rnd((N, cout), 32)
    res = rnd((N, D, H, W, cout), 33, dtype)
    out = conv.fwd(xd, bias.to(DEV), nbias.to(DEV), to_dev(res, dtype))
    ref = ref_conv(x, w, bias, nbias, res, ks, stride, ups, circ)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= conv_tol(dtype, ref), f"{name}: max err {err} > {conv_tol(dtype, ref)}"
    out2 = conv.fwd(xd)                                   # deterministic: same bits on a second launch
    assert torch.equal(out2, conv.fwd(xd))
    dout = rnd((N, D, H, W, cout), 34, dtype)
    cpo = ops.cpad(cout, dtype)
    dd = torch.zeros((N, D, H, W, cpo), dtype=dtype, device=DEV)
    dd[..., :cout] = dout.to(dtype).to(DEV)
    dx = conv.dgrad(dd)
    xr = x.clone().requires_grad_(True)
    ref_conv(xr, w, None, None, None, ks, stride, ups, circ).backward(dout)
    err_x = (dx.float().cpu()[..., :cin] - xr.grad).abs().max().item()
    assert err_x <= conv_tol(dtype, xr.grad), f"{name}: dgrad err {err_x}"
    if cin == cout:                                       # gradient accumulation through the residual input of the dgrad
        acc_in = rnd((N, D, H, W, cin), 35, dtype)
        dx2 = conv.dgrad(dd, residual=to_dev(acc_in, dtype))
        err_a = (dx2.float().cpu() - (xr.grad + acc_in)).abs().max().item()
        assert err_a <= conv_tol(dtype, xr.grad + acc_in), f"{name}: dgrad+residual err {err_a}"


GN_FUSED_CASES = [c for c in CONV_CASES if c[0] in ("k3_deep_256_256", "k3_deep_160_128_circ", "k3_deep_256_64", "k3_32_32", "k3_32_32_ragged", "k3_64_32", "k3_128_256", "k3_cin2pad", "k3_cin8_circ", "k3_cin3_ragged", "k3_mid_64_64", "k3_mid_32_128", "k3_mid_128_64", "k3_mid_64_128", "k3_big_64_64", "k3_ups", "k3_ups_circ", "k3_ups_ragged", "k3_ups_128_64", "k3_circ_small",
                                                      "k3_s2", "k3_s2_ragged", "k1_64_32", "k3_16_16", "k3_48_96")] + LARGE_CASES[:2]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", GN_FUSED_CASES, ids=[c[0] for c in GN_FUSED_CASES])
def test_conv_fused_gn_stats(case, dtype):
    """GroupNorm statistics reduced in the conv epilogue (from the fp32 results) vs a separate gn_stats pass over the stored
    output, alone and as either half of a two-source GroupNorm."""
    ops, conv, x, w, xd = _conv_setup(case, dtype, seed=40)
    name, N, (D, H, W), cin, cout, ks, stride, ups, circ = case
    if cout % 8:
        pytest.skip("GroupNorm needs channels divisible by the group count")
    G = 8
    bias = rnd((cout,), 41)
    res = rnd((N, D, H, W, cout), 42, dtype)
    out = conv.fwd(xd, bias.to(DEV), None, to_dev(res, dtype), gn=True)
    assert out.gn_partials is not None and out.gn_partials.shape[0] == N and out.gn_partials.shape[2:] == (cout, 2)
    fused = ops.gn_stats(out, None, G)
    plain_t = out.clone()                                  # (no gn_partials attribute: full pass)
    plain = ops.gn_stats(plain_t, None, G)
    ref = out.float().cpu().reshape(N, -1, G, cout // G)
    ref = torch.stack([ref.sum(dim=(1, 3)), (ref * ref).sum(dim=(1, 3))], dim=-1)
    tol = 2e-5 * ref.abs().max().item() + 1e-4
    assert (plain.cpu() - ref).abs().max().item() <= tol
    # fused: moments of the fp32 results before the storage rounding (zero-mean errors of 2^-9 relative per element)
    if dtype == torch.float32:
        ftol = torch.full_like(ref, tol)
    else:                                                  # worst case: every element off by half a bf16 ulp in the same direction
        xa = out.float().cpu().reshape(N, -1, G, cout // G)
        ftol = torch.stack([2.0 ** -8 * xa.abs().sum(dim=(1, 3)), 2.0 ** -7 * (xa * xa).sum(dim=(1, 3))], dim=-1) + tol
    assert ((fused.cpu() - ref).abs() <= ftol).all(), f"{name}: fused stats differ from the stored tensor's"
    assert torch.equal(fused, ops.gn_stats(out, None, G))                      # deterministic
    other = to_dev(rnd((N, D, H, W, cout), 43, dtype), dtype)
    for a, b in ((out, other), (other, out)):
        two = ops.gn_stats(a, b, G)
        two_plain = ops.gn_stats(a.clone(), b.clone(), G)
        assert (two - two_plain).abs().max().item() <= ftol.max().item(), f"{name}: two-source stats"


GRAD_CASES = [c for c in CONV_CASES if c[0] in ("k3_32_32", "k3_32_32_ragged", "k3_64_32", "k3_32_64", "k3_128_256", "k3_cin2pad", "k3_cin8_circ", "k3_cin3_ragged", "k3_mid_64_64", "k3_mid_32_128", "k3_mid_128_64", "k3_mid_64_128", "k3_big_64_64",
                                                  "k3_cout1", "k3_circ", "k3_s2", "k3_s2_circ", "k3_ups", "k3_ups_circ", "k3_ups_ragged", "k3_ups_128_64",
                                                  "k3_s2_ragged", "k3_s2_128", "k1_64_32",
                                                  "k1_32_128", "k3_48_96")]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", GRAD_CASES, ids=[c[0] for c in GRAD_CASES])
def test_conv_grads(case, dtype):
    """dgrad and wgrad against torch.autograd of the reference conv."""
    ops, conv, x, w, xd = _conv_setup(case, dtype, seed=10)
    name, N, (D, H, W), cin, cout, ks, stride, ups, circ = case
    dout = rnd((N, D, H, W, cout), 11, dtype)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = ref_conv(xr, wr, None, None, None, ks, stride, ups, circ)
    ref.backward(dout)
    cpo = ops.cpad(cout, dtype)
    dd = torch.zeros((N, D, H, W, cpo), dtype=dtype, device=DEV)
    dd[..., :cout] = dout.to(dtype).to(DEV)
    # wgrad
    dw = torch.full((ks ** 3, cout, cin), float("nan"), device=DEV)
    db = torch.full((cout,), float("nan"), device=DEV) if ks == 3 else None
    conv.wgrad(xd, dd, dw, db)
    if db is not None:                       # fused bias gradient = column sums of dout
        bref = dout.reshape(-1, cout).sum(0)
        assert (db.cpu() - bref).abs().max().item() <= 1e-3 * max(bref.abs().max().item(), 1.0) + 1e-3, f"{name}: fused dbias"
    tol_w = (2e-4 if dtype == torch.float32 else 2.0 ** -7) * max(wr.grad.abs().max().item(), 1e-6)
    err_w = (dw.cpu() - wr.grad).abs().max().item()
    assert err_w <= tol_w, f"{name}: wgrad err {err_w} > {tol_w}"
    # accumulate flag
    conv.wgrad(xd, dd, dw, db, accumulate=True)
    assert (dw.cpu() - 2 * wr.grad).abs().max().item() <= 2 * tol_w
    if db is not None:
        assert (db.cpu() - 2 * dout.reshape(-1, cout).sum(0)).abs().max().item() <= 2e-3 * max(dout.reshape(-1, cout).sum(0).abs().max().item(), 1.0) + 2e-3
    # dgrad
    if stride == 2:
        conv.pack(w.to(DEV), dtype, need_dgrad=True)
    dx = conv.dgrad(dd)
    gref = xr.grad
    err_x = (dx.float().cpu()[..., :cin] - gref).abs().max().item()
    assert dx.shape[:-1] == gref.shape[:-1]
    assert err_x <= conv_tol(dtype, gref) * (4 if ups else 1), f"{name}: dgrad err {err_x}"


def test_dgrad_residual():
    ops, conv, x, w, xd = _conv_setup(("k3_32_32", 2, (8, 8, 16), 32, 32, 3, 1, 0, False), torch.float32, seed=20)
    dout = rnd((2, 8, 8, 16, 32), 21)
    res = rnd((2, 8, 8, 16, 32), 22)
    a = conv.dgrad(dout.to(DEV))
    b = conv.dgrad(dout.to(DEV), residual=res.to(DEV))
    assert (b.cpu() - a.cpu() - res).abs().max().item() <= 1e-5


# ------------------------------------------------------------------------------------------ GroupNorm + SiLU
GN_CASES = [  # N, voxels-shape, c1, c2, groups
    (2, (4, 6, 10), 32, 0, 8),
    (2, (4, 6, 10), 32, 32, 8),
    (1, (3, 5, 7), 64, 64, 8),
    (2, (4, 4, 4), 256, 0, 8),
    (1, (8, 8, 8), 16, 0, 8),      # group size 2 < piece width
    (1, (4, 4, 6), 48, 48, 8),
    (3, (2, 2, 2), 8, 0, 4),
]


def ew_tol(dtype, ref):
    return 1e-5 + 1e-5 * ref.abs().max().item() if dtype == torch.float32 else 2.0 ** -7 * max(ref.abs().max().item(), 1e-6)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", GN_CASES, ids=[f"gn{i}" for i in range(len(GN_CASES))])
def test_gn_silu_fwd_bwd(case, dtype):
    ops = _ops()
    N, sp, c1, c2, G = case
    C = c1 + c2
    x1 = (rnd((N,) + sp + (c1,), 1) * 1.5 + 0.3).to(dtype).float()
    x2 = rnd((N,) + sp + (c2,), 2, dtype) if c2 else None
    gamma = 1.0 + 0.3 * rnd((C,), 3)
    beta = 0.2 * rnd((C,), 4)
    dy = rnd((N,) + sp + (C,), 5, dtype)
    add1 = rnd((N,) + sp + (c1,), 6, dtype)
    add2 = rnd((N,) + sp + (c2,), 7, dtype) if c2 else None
    xc = (x1 if x2 is None else torch.cat([x1, x2], -1)).clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.silu(F.group_norm(xc.permute(0, 4, 1, 2, 3), G, gr, br, 1e-5)).permute(0, 2, 3, 4, 1)
    y_ref.backward(dy)

    d1, d2 = to_dev(x1, dtype), (to_dev(x2, dtype) if c2 else None)
    st = ops.gn_stats(d1, d2, G)
    # statistics
    V = math.prod(sp)
    xg = xc.detach().reshape(N, V, G, C // G)
    assert torch.allclose(st[..., 0].cpu(), xg.sum((1, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(st[..., 1].cpu(), (xg ** 2).sum((1, 3)), rtol=1e-4, atol=1e-3)
    y = ops.gn_silu_fwd(d1, d2, G, st, gamma.to(DEV), beta.to(DEV))
    err = (y.float().cpu() - y_ref.detach()).abs().max().item()
    assert err <= ew_tol(dtype, y_ref.detach()) * 4, f"gn fwd err {err}"

    dgam = torch.zeros(C, device=DEV)
    dbet = torch.zeros(C, device=DEV)
    cs = torch.zeros(N, C + 5, device=DEV)
    dx1, dx2 = ops.gn_silu_bwd(d1, d2, G, st, gamma.to(DEV), beta.to(DEV), to_dev(dy, dtype), dgam, dbet,
                               add1=to_dev(add1, dtype), add2=(to_dev(add2, dtype) if c2 else None), colsum=cs[:, 2:2 + C])
    gx = xc.grad
    ref1 = gx[..., :c1] + add1
    tol = ew_tol(dtype, ref1) * 8
    e1 = (dx1.float().cpu() - ref1).abs().max().item()
    assert e1 <= tol, f"gn bwd dx1 err {e1} > {tol}"
    full = ref1
    if c2:
        ref2 = gx[..., c1:] + add2
        e2 = (dx2.float().cpu() - ref2).abs().max().item()
        assert e2 <= tol, f"gn bwd dx2 err {e2}"
        full = torch.cat([ref1, ref2], -1)
    rt = 1e-3 if dtype == torch.float32 else 2e-2
    assert torch.allclose(dgam.cpu(), gr.grad, rtol=rt, atol=rt * gr.grad.abs().max().item())
    assert torch.allclose(dbet.cpu(), br.grad, rtol=rt, atol=rt * br.grad.abs().max().item())
    cref = full.reshape(N, V, C).sum(1)
    assert torch.allclose(cs[:, 2:2 + C].cpu(), cref, rtol=rt, atol=rt * cref.abs().max().item() + 1e-3)
    assert cs[:, :2].abs().max().item() == 0 and cs[:, 2 + C:].abs().max().item() == 0


def test_gn_dropout_consistency():
    """Dropout mask: fwd and bwd regenerate the same Philox keep-mask; keep rate ~ 1-p; kept values scaled 1/(1-p)."""
    ops = _ops()
    dtype = torch.float32
    N, sp, C, G, p = 2, (8, 8, 8), 32, 8, 0.1
    x = rnd((N,) + sp + (C,), 1)
    gamma, beta = torch.ones(C), torch.zeros(C)
    d = x.to(DEV)
    st = ops.gn_stats(d, None, G)
    y0 = ops.gn_silu_fwd(d, None, G, st, gamma.to(DEV), beta.to(DEV))
    y1 = ops.gn_silu_fwd(d, None, G, st, gamma.to(DEV), beta.to(DEV), p, 1234)
    y1b = ops.gn_silu_fwd(d, None, G, st, gamma.to(DEV), beta.to(DEV), p, 1234)
    y2 = ops.gn_silu_fwd(d, None, G, st, gamma.to(DEV), beta.to(DEV), p, 99)
    assert torch.equal(y1, y1b) and not torch.equal(y1, y2)
    kept = y1 != 0
    rate = kept.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    assert torch.allclose(y1[kept], y0[kept] / (1 - p), rtol=1e-6, atol=1e-7)
    # backward uses the same mask: dx == 0 contribution where dropped -> compare with autograd using the mask
    dy = rnd((N,) + sp + (C,), 5)
    mask = (kept.float() / (1 - p)).cpu()
    xr = x.clone().requires_grad_(True)
    yr = F.silu(F.group_norm(xr.permute(0, 4, 1, 2, 3), G, gamma, beta, 1e-5)).permute(0, 2, 3, 4, 1) * mask
    yr.backward(dy)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx, _ = ops.gn_silu_bwd(d, None, G, st, gamma.to(DEV), beta.to(DEV), dy.to(DEV), dg, db, dropout_p=p, seed=1234)
    assert (dx.cpu() - xr.grad).abs().max().item() <= 1e-4



# ------------------------------------------------------------------------------------------ fused GroupNorm backward
FOLD_CASES = [  # name, N, (D,H,W), producer cin, c1, c2 (GroupNorm over c1+c2 channels), conv cout, groups, dropout p, circular
    ("l0_32", 2, (8, 8, 16), 32, 32, 0, 32, 8, 0.1, False),          # NC=2 lanes (8 channels), dropout, 4x8x16 tiles
    ("l0_concat", 1, (6, 9, 20), 32, 32, 32, 32, 8, 0.0, False),      # two-source GroupNorm (64 ch, NC=4), ragged tiles
    ("deep_128", 2, (4, 4, 16), 64, 128, 0, 128, 8, 0.25, True),      # NC=4 / half-chunk workgroups on a small grid
    ("c16", 1, (8, 8, 16), 16, 16, 0, 32, 8, 0.1, False),             # NC=1 lanes (4 channels: half a bf16 piece)
    ("out_1", 2, (8, 8, 16), 32, 32, 0, 1, 8, 0.0, False),            # conv_out: 1 output channel -> tap-packed dgrad kernel
    ("concat_16", 1, (4, 8, 16), 16, 16, 16, 16, 8, 0.0, False),      # 32-channel concat of 16 + 16 (NC=2 lanes)
    ("deep_256", 1, (4, 8, 16), 64, 256, 0, 256, 8, 0.1, False),      # K-split kernel (8 K-blocks) with the folded epilogue
    ("deep_concat", 2, (8, 8, 16), 64, 64, 64, 192, 8, 0.0, False),   # K-split (6 K-blocks), two-source GroupNorm, 1x8x16 tiles
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", FOLD_CASES, ids=[c[0] for c in FOLD_CASES])
def test_gn_bwd_folded_into_dgrad(case, dtype):
    """Conv.dgrad_gn + gn_bwd_fused (GroupNorm+SiLU+dropout backward with the reduction folded into the dgrad epilogue, analytic
    column sums, no atomics) against torch.autograd of  x -> dropout(silu(group_norm(x))) -> conv3d  on the CPU in fp32.
    The GroupNorm input x is produced by a HIP conv with gn=True so that the forward partials / per-channel sums exist; the
    dropout mask of the reference is the one the forward kernel wrote (keep bytes)."""
    ops = _ops()
    name, N, sp, pcin, c1, c2, cout, G, p, circ = case
    C = c1 + c2
    D, H, W = sp
    epl = ops.epl(dtype)

    def produce(cx, seed):          # GroupNorm input tensor of cx channels, written by a conv epilogue (partials attached)
        cv = ops.Conv(pcin, cx, 3, circular=circ)
        cv.pack(rnd((27, cx, pcin), seed, dtype, scale=1.0 / math.sqrt(27 * pcin)).to(DEV), dtype, need_dgrad=False)
        x0 = to_dev(rnd((N, D, H, W, ops.cpad(pcin, dtype)), seed + 1, dtype), dtype)
        return cv.fwd(x0, (0.3 * rnd((cx,), seed + 2)).to(DEV), gn=True)
    x1 = produce(c1, 10)
    x2 = produce(c2, 20) if c2 else None
    gamma = 1.0 + 0.3 * rnd((C,), 3)
    beta = 0.2 * rnd((C,), 4)
    st = ops.gn_stats(x1, x2, G, chsum=True)
    assert st.chsum is not None and st.chsum.shape == (N, C)
    y = ops.gn_silu_fwd(x1, x2, G, st, gamma.to(DEV), beta.to(DEV), p, 777, want_mask=True)
    # keep mask bytes -> [N, V, C] multiplier
    V = D * H * W
    if p > 0:
        mb = y.keep_mask.cpu().to(torch.int32)
        assert mb.shape == (N, V, C // epl)
        bits = ((mb[..., None] >> torch.arange(epl)) & 1).reshape(N, V, C).float()
        assert abs(bits.mean().item() - (1 - p)) < 0.03
        yc, bsh = y.float().cpu(), bits.reshape(y.shape)
        assert (yc[bsh == 0] == 0).all(), "a dropped element must be stored as zero"
        assert ((yc != 0) & (bsh == 1)).float().sum().item() >= 0.999 * (bsh == 1).float().sum().item()
        mult = (bits / (1 - p)).reshape(N, D, H, W, C)
    else:
        assert y.keep_mask is None
        mult = torch.ones(N, D, H, W, C)
    # conv on y
    w = rnd((27, cout, C), 30, dtype, scale=1.0 / math.sqrt(27 * C))
    conv = ops.Conv(C, cout, 3, circular=circ)
    conv.pack(w.to(DEV), dtype, need_dgrad=True)
    dout = rnd((N, D, H, W, cout), 31, dtype)
    dd = torch.zeros((N, D, H, W, ops.cpad(cout, dtype)), dtype=dtype, device=DEV)
    dd[..., :cout] = dout.to(dtype).to(DEV)
    add1 = rnd((N, D, H, W, c1), 6, dtype)
    add2 = rnd((N, D, H, W, c2), 7, dtype) if c2 else None

    # reference
    xc = torch.cat([x1.float().cpu()] + ([x2.float().cpu()] if c2 else []), -1).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.silu(F.group_norm(xc.permute(0, 4, 1, 2, 3), G, gr, br, 1e-5)).permute(0, 2, 3, 4, 1) * mult
    ref_conv(yr, w, None, None, None, 3, 1, 0, circ).backward(dout)
    gx = xc.grad

    assert conv.gn_fold_ok(c1, c2, dtype)
    dyh = conv.dgrad_gn(dd, x1, x2, G, st, gamma.to(DEV), beta.to(DEV), keep_mask=y.keep_mask, dropout_p=p)
    assert dyh.shape == (N, D, H, W, C) and dyh.gnb_partials.shape[2:] == (C, 2)
    dgam = torch.full((C,), float("nan"), device=DEV)
    dbet = torch.full((C,), float("nan"), device=DEV)
    cs = torch.zeros(N, C + 5, device=DEV)
    dx1, dx2 = ops.gn_bwd_fused(x1, x2, G, st, gamma.to(DEV), dyh, dgam, dbet, add1=to_dev(add1, dtype),
                                add2=(to_dev(add2, dtype) if c2 else None), colsum=cs[:, 2:2 + C])
    scale = max(gx.abs().max().item(), 1e-6)
    tol = (2e-4 if dtype == torch.float32 else 2.0 ** -6) * scale + ew_tol(dtype, add1)
    e1 = (dx1.float().cpu() - (gx[..., :c1] + add1)).abs().max().item()
    assert e1 <= tol, f"{name}: dx1 err {e1} > {tol}"
    if c2:
        e2 = (dx2.float().cpu() - (gx[..., c1:] + add2)).abs().max().item()
        assert e2 <= tol, f"{name}: dx2 err {e2} > {tol}"
    rt = 2e-3 if dtype == torch.float32 else 3e-2
    assert torch.allclose(dgam.cpu(), gr.grad, rtol=rt, atol=rt * gr.grad.abs().max().item()), f"{name}: dgamma"
    assert torch.allclose(dbet.cpu(), br.grad, rtol=rt, atol=rt * br.grad.abs().max().item()), f"{name}: dbeta"
    cref = gx.reshape(N, V, C).sum(1)
    assert torch.allclose(cs[:, 2:2 + C].cpu(), cref, rtol=rt, atol=rt * cref.abs().max().item() + 1e-3), f"{name}: analytic colsum"
    assert cs[:, :2].abs().max().item() == 0 and cs[:, 2 + C:].abs().max().item() == 0
    # bit-reproducible: a second run gives identical bits
    dyh2 = conv.dgrad_gn(dd, x1, x2, G, st, gamma.to(DEV), beta.to(DEV), keep_mask=y.keep_mask, dropout_p=p)
    dgam2, dbet2 = torch.empty_like(dgam), torch.empty_like(dbet)
    cs2 = torch.zeros_like(cs)
    r1, r2 = ops.gn_bwd_fused(x1, x2, G, st, gamma.to(DEV), dyh2, dgam2, dbet2, add1=to_dev(add1, dtype),
                              add2=(to_dev(add2, dtype) if c2 else None), colsum=cs2[:, 2:2 + C],
                              dx1=(torch.empty_like(x1) if c2 else None))
    assert torch.equal(r1, dx1) and torch.equal(dgam, dgam2) and torch.equal(dbet, dbet2) and torch.equal(cs, cs2)


# ------------------------------------------------------------------------------------------ K6: conditioning table
@pytest.mark.parametrize("cfg", [dict(t=True, vd=(6,), chs=(32, 64, 128, 256), B=2), dict(t=True, vd=(), chs=(16, 32), B=3),
                                 dict(t=False, vd=(6, 3), chs=(16, 32, 64), B=1), dict(t=True, vd=(6, 3), chs=(16, 32), B=4),
                                 dict(t=True, vd=(6,), chs=(48, 96, 192, 384), B=2), dict(t=True, vd=(5,), chs=(64, 128), B=3)],
                         ids=["c3", "t_only", "v_only_two", "t_and_two_v", "chs48_width192", "chs64_width256"])
def test_cond_table_kernel(cfg):
    """K6 (sinusoid -> 2 x (Linear + GELU) -> projections of all blocks, one launch) against the oracle's
    sinusoidal_embedding / _mlp2 / per-block Linear on the CPU (fp32, 1e-5 relative), and its backward (2 launches) against
    torch.autograd through the same oracle ops, for every conditioning parameter and the conv1 biases."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import unet_oracle
    from helpers import oracle_params, randomize
    from vdm4cdm_amd.networks import CUNet
    ops = _ops()
    B = cfg["B"]
    net = CUNet(shape=(1, 16, 16, 16), chs=list(cfg["chs"]), s_conditioning_channels=0, v_conditioning_dims=list(cfg["vd"]),
                t_conditioning=cfg["t"], norm_groups=8, backend="hip", precision="fp32")
    randomize(net, 5)
    g = torch.Generator().manual_seed(1)
    t = torch.rand(B, generator=g)
    vs = [torch.randn(B, d, generator=g) for d in cfg["vd"]]
    # oracle: conds -> per-block projections, concatenated in block order
    P = {k: v.clone().requires_grad_(True) for k, v in oracle_params(net).items()}
    conds = []
    if cfg["t"]:
        conds.append(unet_oracle._mlp2(P, "t_embed", unet_oracle.sinusoidal_embedding(t)))
    for k, v in enumerate(vs):
        conds.append(unet_oracle._mlp2(P, f"v_embeds.{k}", v))
    ref = torch.cat([sum(F.linear(c, P[f"{b.name}.cond.{k}.weight"]) for k, c in enumerate(conds)) for b in net.blocks], dim=1)
    flat = net.flat.detach().to(DEV)
    specs = net.cond_specs(t.to(DEV), [v.to(DEV) for v in vs], flat)
    ct = ops.CondTable(specs, B, net.table_width)
    table = ct.forward(save=True)
    assert table.shape == ref.shape
    assert (table.cpu() - ref.detach()).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-6
    # backward
    dtab = torch.randn(ref.shape, generator=g)
    ref.backward(dtab)
    gflat = torch.full_like(flat, float("nan"))
    grads = [{k: sp[k] for k in ("w1", "b1", "w2", "b2", "wproj")} for sp in net.cond_specs(None, [None] * len(vs), gflat)]
    dpad = torch.zeros(B, net.table_width + 7, device=DEV)          # row stride != width
    dpad[:, :net.table_width] = dtab.to(DEV)
    ct.backward(dpad[:, :net.table_width], grads, dbias=net.conv1_bias_all(gflat))
    got = oracle_params(net, flat=gflat)
    names = [n for n in P if ("embed" in n or ".cond." in n)]
    assert names
    for n in names:
        r = P[n].grad
        err = (got[n] - r).abs().max().item()
        assert err <= 5e-5 * max(r.abs().max().item(), 1e-3) + 1e-6, f"{n}: {err}"      # (device sinf / erff vs libm, fp32 sums)
    assert torch.allclose(net.conv1_bias_all(gflat).cpu(), dtab.sum(0), rtol=1e-5, atol=1e-5)
    # the step gather of the sampler
    if cfg["t"]:
        tt = ops.CondTable(net.cond_specs(torch.linspace(0, 1, 5).to(DEV), None, flat, which="t"), 5, net.table_width).forward(save=False)
        step = torch.tensor([3], dtype=torch.int32, device=DEV)
        out = torch.empty(B, net.table_width, device=DEV)
        ops.cond_table_step(tt, table, step, B, net.table_width, out)
        assert torch.equal(out, tt[3][None] + table)


@pytest.mark.parametrize("rows", [250, 1000])
def test_cond_table_all_sampling_steps(rows):
    """C5: the sampler embeds the time values of ALL n steps with one K6 launch (rows = n_sampling_steps: 250 = the reference default,
    generate_3D.py:61; 1000 = BASELINE C5) and gathers row *step inside the captured graph.  Every row of the table against the
    oracle's sinusoidal_embedding / _mlp2 / per-block Linear on the fp32 time grid of VDM.step_table; then the device-side gather."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import unet_oracle
    from helpers import oracle_params, randomize
    from vdm4cdm_amd.networks import CUNet
    from vdm4cdm_amd.vdm_model import VDM
    ops = _ops()
    net = CUNet(shape=(1, 16, 16, 16), chs=[32, 64, 128, 256], s_conditioning_channels=0, v_conditioning_dims=[6], t_conditioning=True,
                norm_groups=8, backend="hip", precision="fp32")
    randomize(net, 6)
    coef = VDM(net).step_table(rows).float()
    t = coef[:, 3].contiguous()
    assert t.shape == (rows,) and t[0].item() == 1.0 and 0.0 < t[-1].item() < 2.0 / rows
    P = oracle_params(net)
    c = unet_oracle._mlp2(P, "t_embed", unet_oracle.sinusoidal_embedding(t))
    ref = torch.cat([F.linear(c, P[f"{b.name}.cond.0.weight"]) for b in net.blocks], dim=1)
    flat = net.flat.detach().to(DEV)
    tt = ops.CondTable(net.cond_specs(t.to(DEV), None, flat, which="t"), rows, net.table_width).forward(save=False)
    assert tt.shape == ref.shape == (rows, net.table_width)
    err = (tt.cpu() - ref).abs().max(dim=1).values
    assert err.max().item() <= 1e-5 * ref.abs().max().item() + 1e-6, f"row {int(err.argmax())}: {err.max().item()}"
    v = torch.rand(2, 6, generator=torch.Generator().manual_seed(2))
    tv = ops.CondTable(net.cond_specs(None, [v.to(DEV)], flat, which="v"), 2, net.table_width).forward(save=False)
    out = torch.empty(2, net.table_width, device=DEV)
    for s in (0, 1, rows // 2, rows - 1):
        ops.cond_table_step(tt, tv, torch.tensor([s], dtype=torch.int32, device=DEV), 2, net.table_width, out)
        assert torch.equal(out, tt[s][None] + tv)


def test_cond_table_rejects_unsupported_widths():
    """4 * chs[0] > 256 exceeds the K6 kernel's LDS vectors: CUNet(backend="hip") must say so at construction, not at the first forward."""
    from vdm4cdm_amd.networks import CUNet
    with pytest.raises(ValueError, match="chs\\[0\\]"):
        CUNet(shape=(1, 16, 16, 16), chs=[96, 192], t_conditioning=True, backend="hip")
    CUNet(shape=(1, 16, 16), chs=[96, 192], t_conditioning=True, backend="torch")          # the torch backend has no such limit


# ------------------------------------------------------------------------------------------ fused attention core
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("N,V,H,hd", [(2, 64, 4, 16), (1, 100, 2, 32), (1, 4096, 4, 64), (2, 216, 3, 96), (1, 36, 1, 128)],
                         ids=["v64_hd16", "v100_ragged_hd32", "v4096_hd64", "v216_hd96", "v36_hd128"])
def test_fused_attention_core(dtype, N, V, H, hd):
    """csrc/attention.hip (scores, online softmax, P V on the matrix cores; dK/dV and dQ passes) against softmax(q k^T / sqrt(hd)) v and
    its autograd gradients in fp32 on the CPU.  Ragged voxel counts (not a multiple of the 16 / 32-key tiles) exercise the masks.
    fp32 operands: exact MFMA, <= 2e-5 relative; bf16 operands: inputs are rounded to bf16 first (the reference sees the same values),
    the remaining difference is the bf16 rounding of P / dS inside the kernel: <= 2e-2 of max|ref|."""
    ops = _ops()
    C = H * hd
    g = torch.Generator().manual_seed(V + hd)
    qkv = (torch.randn(N, V, 3 * C, generator=g) * 0.7).to(dtype)
    dout = torch.randn(N, V, C, generator=g).to(dtype)
    scale = 1.0 / math.sqrt(hd)
    ref_in = qkv.float().clone().requires_grad_(True)
    q, k, v = (ref_in[:, :, i * C:(i + 1) * C].reshape(N, V, H, hd).permute(0, 2, 1, 3) for i in range(3))
    p = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * scale, dim=-1)
    ref = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(N, V, C)
    ref.backward(dout.float())
    dev = qkv.to(DEV)
    qh, qt = ops.attn_split_heads(dev, 0, 3, H)
    kh, kt = ops.attn_split_heads(dev, 1, 3, H)
    vh, vt = ops.attn_split_heads(dev, 2, 3, H)
    assert torch.equal(qh.cpu(), qkv[:, :, :C].reshape(N, V, H, hd).permute(0, 2, 1, 3)) and torch.equal(vt.cpu(), vh.cpu().transpose(-1, -2))
    out, lse = ops.attn_fwd(qh, kh, vt, scale, (N, V, C))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    err = (out.float().cpu() - ref.detach()).abs().max().item()
    assert err <= tol * ref.abs().max().item(), f"forward {err} vs {ref.abs().max().item()}"
    ref_lse = torch.logsumexp(torch.matmul(q, k.transpose(-1, -2)) * scale, dim=-1).detach()
    assert (lse.cpu() - ref_lse).abs().max().item() <= (1e-4 if dtype == torch.float32 else 5e-2)
    dqkv = ops.attn_bwd(qh, kh, vh, qt, kt, dout.to(DEV), out, lse, scale)
    gerr = (dqkv.float().cpu() - ref_in.grad).abs().max().item()
    assert gerr <= (2e-4 if dtype == torch.float32 else 3e-2) * ref_in.grad.abs().max().item(), f"backward {gerr} vs {ref_in.grad.abs().max().item()}"
    dqkv2 = ops.attn_bwd(qh, kh, vh, qt, kt, dout.to(DEV), out, lse, scale)
    assert torch.equal(dqkv, dqkv2), "attention backward is not bit-reproducible"


def test_fused_attention_rejects_unbuilt_head_widths():
    from vdm4cdm_amd.networks import CUNet
    with pytest.raises(ValueError, match="head width"):
        CUNet(shape=(1, 16, 16, 16), chs=[16, 40], mid_attn=True, n_attention_heads=1, backend="hip")
    with pytest.raises(_lib_mod().VdmError, match="multiple of 4"):
        _ops().attn_split_heads(torch.zeros(1, 6, 48, device=DEV), 0, 3, 1)


# ------------------------------------------------------------------------------------------ small ops
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_small_ops(dtype):
    ops = _ops()
    a, b = rnd((2, 4, 4, 4), 3), rnd((2, 4, 4, 4), 4)
    pk = ops.pack_input(a.to(DEV), b.to(DEV), dtype).float().cpu()
    assert pk.shape[-1] == ops.cpad(2, dtype)
    assert (pk[..., 0] - a.to(dtype).float()).abs().max() == 0 and (pk[..., 1] - b.to(dtype).float()).abs().max() == 0
    assert pk[..., 2:].abs().max() == 0
    pk1 = ops.pack_input(a.to(DEV), None, dtype).float().cpu()
    assert pk1[..., 1:].abs().max() == 0


def test_vdm_elementwise():
    ops = _ops()
    B, per = 3, 4 * 5 * 6 * 4
    x, eps, eh, e0 = rnd((B, per), 1), rnd((B, per), 2), rnd((B, per), 3), rnd((B, per), 4)
    al, si = torch.tensor([0.9, 0.5, 0.1]), torch.tensor([0.3, 0.8, 0.99])
    z = ops.diffuse(x.to(DEV), eps.to(DEV), al.to(DEV), si.to(DEV)).cpu()
    assert torch.allclose(z, al[:, None] * x + si[:, None] * eps, atol=1e-6)
    coef = torch.tensor([0.5, 1.5, 2.0])
    sums = torch.zeros(B, 3, device=DEV)
    d = torch.empty(B, per, device=DEV)
    ops.loss_terms(x.to(DEV), eps.to(DEV), eh.to(DEV), e0.to(DEV), 0.01, coef.to(DEV), sums, d)
    ref = torch.stack([((eh - eps) ** 2).sum(1), (x ** 2).sum(1), ((0.01 * e0) ** 2).sum(1)], 1)
    assert torch.allclose(sums.cpu(), ref, rtol=1e-5)
    assert torch.allclose(d.cpu(), coef[:, None] * (eh - eps), atol=1e-6)
    # ancestral update with supplied noise, scalars from the device table
    coefs = torch.tensor([[0.9, 0.1, 0.2, 0.7], [1.1, 0.3, 0.05, 0.4]])
    step = torch.tensor([1], dtype=torch.int32, device=DEV)
    zz = x.clone().to(DEV)
    ops.ancestral_step(zz, eh.to(DEV), e0.to(DEV), coefs.to(DEV), step, 0)
    assert torch.allclose(zz.cpu(), 1.1 * (x - 0.3 * eh) + 0.05 * e0, atol=1e-6)
    ops.step_inc(step)
    assert step.item() == 2
    s = torch.zeros(1, device=DEV)
    big = rnd((1_000_003,), 9)
    bd = torch.zeros(1_000_004, device=DEV)[:1_000_003]
    bd.copy_(big)
    ops.sumsq(bd, s)
    assert s.item() == pytest.approx((big.double() ** 2).sum().item(), rel=1e-5)


def test_philox_normals():
    """K9/K7 noise source: N(0,1) moments, stream independence, reproducibility."""
    ops = _ops()
    n = 1 << 22
    a = ops.randn(torch.empty(n, device=DEV), 42, 1)
    b = ops.randn(torch.empty(n, device=DEV), 42, 1)
    c = ops.randn(torch.empty(n, device=DEV), 42, 2)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.mean().item()) < 3e-3 and abs(a.var().item() - 1) < 5e-3
    assert abs((a ** 4).mean().item() - 3.0) < 0.05
    assert abs((a * c).mean().item()) < 3e-3
    assert abs((a[:-1] * a[1:]).mean().item()) < 3e-3
    # ancestral step with in-kernel noise: z' - ratio*(z - cs*eh) is N(0, scale^2)
    z = torch.zeros(n, device=DEV)
    coefs = torch.tensor([[1.0, 0.0, 2.0, 0.0]], device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.ancestral_step(z, torch.zeros(n, device=DEV), None, coefs, step, 7)
    assert abs(z.std().item() - 2.0) < 0.01


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_pack_many_equals_per_conv_packing(dtype):
    """The one-launch packing of a whole network (vdm_conv_pack_many) writes the same bytes as vdm_conv_pack_weights conv by conv,
    for every weight layout: generic (fwd / flipped-transposed dgrad), per-parity-class (up-sampling conv, stride-2 dgrad), tap-packed."""
    ops = _ops()
    shapes = [(32, 32, 3, 1, 0), (64, 32, 3, 1, 0), (48, 96, 3, 1, 0), (2, 32, 3, 1, 0), (32, 1, 3, 1, 0), (32, 32, 3, 2, 0),
              (64, 32, 3, 1, 1), (64, 32, 1, 1, 0), (128, 256, 3, 1, 0)]
    flat = torch.randn(sum(ks ** 3 * co * ci for ci, co, ks, _, _ in shapes), device=DEV)
    convs, ref, off = [], [], 0
    for ci, co, ks, st, up in shapes:
        n = ks ** 3 * co * ci
        w = flat[off:off + n].view(ks ** 3, co, ci)
        off += n
        a = ops.Conv(ci, co, ks, stride=st, upsample=up)
        b = ops.Conv(ci, co, ks, stride=st, upsample=up)
        b.pack(w, dtype, need_dgrad=True)
        convs.append((a, w))
        ref.append(b)
    plan = ops.PackPlan(convs, dtype, need_dgrad=True)
    for a, _ in convs:
        a.wf.fill_(0xAB)
        a.wd.fill_(0xAB)
    plan.run()
    for (a, _), b, shp in zip(convs, ref, shapes):
        assert torch.equal(a.wf, b.wf), f"fwd packing differs for {shp}"
        assert torch.equal(a.wd, b.wd), f"dgrad packing differs for {shp}"


def test_channel_sums():
    """Channel sums (bias gradients of the attention block's 1x1x1 projections) vs torch."""
    from vdm4cdm_amd import hip_ops as ops
    g = torch.Generator().manual_seed(3)
    for dtype in DTYPES:
        x = torch.randn(2, 5, 3, 7, 32, generator=g)
        xd = x.to(DEV).to(dtype)
        out = torch.empty(32, device=DEV)
        ops.channel_sums(xd, out)
        ref = xd.float().cpu().reshape(-1, 32).sum(0)
        assert (out.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-4


def test_gn_stats_from_many_tile_partials():
    """Many tiles per sample (level-0 sizes: more than one sweep of the reducing workgroup): statistics from the per-tile partials vs
    a float64 sum of the same partials; bit-reproducible; two-source form."""
    from vdm4cdm_amd import hip_ops as ops
    G = 8
    for N, shape, c in ((1, (64, 64, 128), 32), (2, (128, 64, 64), 32), (1, (64, 64, 64), 64)):
        conv = ops.Conv(c, c, 3)
        g = torch.Generator().manual_seed(N + c)
        conv.pack((torch.randn(27, c, c, generator=g) * 0.05).to(DEV), torch.bfloat16, need_dgrad=False)
        x = torch.randn(N, *shape, c, generator=g).to(DEV).to(torch.bfloat16)
        out = conv.fwd(x, gn=True)
        part = out.gn_partials                                 # [N, tiles, C, 2] fp32
        assert part.shape[1] > (256 // (c // G)) * 8, "more tiles than one sweep of the reducing workgroup"
        st = ops.gn_stats(out, None, G, chsum=True)
        ref = part.double().sum(dim=1).reshape(N, G, c // G, 2).sum(dim=2)
        assert ((st.double() - ref).abs() <= 2e-6 * ref.abs() + 1e-3).all()
        assert ((st.chsum.double() - part[..., 0].double().sum(dim=1)).abs() <= 2e-6 * part[..., 0].double().sum(dim=1).abs() + 1e-3).all()
        for _ in range(3):
            assert torch.equal(ops.gn_stats(out, None, G), st)
        two = ops.gn_stats(out, out, G)                        # two sources: two launches back to back;
        gs2 = 2 * c // G                                       # concat(out, out) in G groups of 2c / G channels
        ref2 = torch.cat([part, part], dim=2).double().sum(dim=1).reshape(N, G, gs2, 2).sum(dim=2)
        assert ((two.double() - ref2).abs() <= 2e-6 * ref2.abs() + 1e-3).all()


SKIP_CASES = [  # (N, spatial, c1, c2, cout, groups)
    (2, (8, 8, 16), 32, 32, 32, 8),      # level-0 up block of the 128^3 config
    (1, (5, 6, 7), 32, 0, 64, 8),        # level-1 down block; ragged voxel count (210: partial 16- and 32-voxel groups)
    (2, (4, 6, 6), 64, 0, 128, 8),       # level-2 down block (forward kernel only: the slab reduce outweighs the small tensor)
    (1, (6, 6, 6), 64, 64, 64, 8),       # level-1 up block (wide: fragments from LDS in the backward kernel)
    (1, (8, 8, 8), 16, 16, 16, 8),       # level 0 of the 256^3 config (chs 16..128): two sources inside one 32-channel K-step
    (3, (3, 5, 5), 16, 0, 32, 8),        # half-empty K-step
    (1, (4, 4, 5), 8, 0, 16, 4),         # one piece per voxel
    (1, (16, 16, 16), 32, 32, 64, 8),    # many chunks per wave
]


@pytest.mark.parametrize("case", SKIP_CASES, ids=[f"{c[2]}+{c[3]}to{c[4]}" for c in SKIP_CASES])
def test_gn_skip_fused_passes(case):
    """norm1 + the 1x1x1 skip conv in one pass (csrc/gn_skip.hip), forward and backward, against fp32 torch on the CPU:
    y = silu(gn(x)), s = W x + b;  dx = GroupNorm backward of dyh (the gradient at the GroupNorm output) + W^T dout,
    dW = dout^T x, dgamma, dbeta.  bf16 storage: outputs within 2^-7 of max|ref| (one rounding), weight gradients 2e-3."""
    ops = _ops()
    dtype = torch.bfloat16
    N, sp, c1, c2, cout, G = case
    C = c1 + c2
    V = math.prod(sp)
    fwd_ok, bwd_ok = ops.gn_skip_supported(c1, c2, cout, dtype)
    assert fwd_ok
    assert ops.gn_skip_supported(c1, c2, cout, torch.float32) == (False, False)
    x1 = (rnd((N,) + sp + (c1,), 11) * 1.5 + 0.3).to(dtype).float()
    x2 = rnd((N,) + sp + (c2,), 12, dtype) if c2 else None
    gamma = 1.0 + 0.3 * rnd((C,), 13)
    beta = 0.2 * rnd((C,), 14)
    w = rnd((cout, C), 15, scale=C ** -0.5)
    bias = 0.1 * rnd((cout,), 16)
    dyh = rnd((N,) + sp + (C,), 17, dtype)
    dout = rnd((N,) + sp + (cout,), 18, dtype)
    wb = w.to(dtype).float()                     # the kernel multiplies bf16-rounded weights
    xc = (x1 if x2 is None else torch.cat([x1, x2], -1)).clone().requires_grad_(True)
    gr, br, wr = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), wb.clone().requires_grad_(True)
    z = F.group_norm(xc.permute(0, 4, 1, 2, 3), G, gr, br, 1e-5).permute(0, 2, 3, 4, 1)
    y_ref = F.silu(z).detach()
    s_ref_t = xc @ wr.t() + bias
    ((z * dyh).sum() + (s_ref_t * dout).sum()).backward()

    d1, d2 = to_dev(x1, dtype), (to_dev(x2, dtype) if c2 else None)
    w1, w2 = w[:, :c1].contiguous().to(DEV), (w[:, c1:].contiguous().to(DEV) if c2 else None)
    st = ops.gn_stats(d1, d2, G)
    y, s = ops.gn_silu_skip_fwd(d1, d2, G, st, gamma.to(DEV), beta.to(DEV), w1, w2, bias.to(DEV))
    assert y.shape == (N,) + sp + (C,) and s.shape == (N,) + sp + (cout,)
    e = (y.float().cpu() - y_ref).abs().max().item()
    assert e <= ew_tol(dtype, y_ref) * 4, f"y err {e}"
    # the activation half is the plain kernel's arithmetic
    assert torch.equal(y, ops.gn_silu_fwd(d1, d2, G, st, gamma.to(DEV), beta.to(DEV)))
    s_ref = s_ref_t.detach()
    e = (s.float().cpu() - s_ref).abs().max().item()
    assert e <= conv_tol(dtype, s_ref), f"skip out err {e} > {conv_tol(dtype, s_ref)}"
    if not bwd_ok:
        assert (c1, c2, cout) == (64, 0, 128)
        return
    ddyh, ddout = to_dev(dyh, dtype), to_dev(dout, dtype)
    ddyh.gnb_partials = ops.channel_dot_sums(ddyh, d1, d2)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dw1, dw2 = torch.full((cout, c1), 7.0, device=DEV), (torch.full((cout, c2), 7.0, device=DEV) if c2 else None)
    dx1, dx2 = ops.gn_bwd_fused(d1, d2, G, st, gamma.to(DEV), ddyh, dgam, dbet, skip=(ddout, w1, w2, dw1, dw2))   # (c2 == 0: in place)
    gx = xc.grad
    tol = ew_tol(dtype, gx) * 8
    e1 = (dx1.float().cpu() - gx[..., :c1]).abs().max().item()
    assert e1 <= tol, f"dx1 err {e1} > {tol}"
    if c2:
        e2 = (dx2.float().cpu() - gx[..., c1:]).abs().max().item()
        assert e2 <= tol, f"dx2 err {e2} > {tol}"
    dw = dw1 if not c2 else torch.cat([dw1, dw2], 1)
    e = (dw.cpu() - wr.grad).abs().max().item()
    assert e <= 2e-3 * wr.grad.abs().max().item(), f"dW err {e} vs max {wr.grad.abs().max().item()}"
    assert torch.allclose(dgam.cpu(), gr.grad, rtol=2e-2, atol=2e-2 * gr.grad.abs().max().item())
    assert torch.allclose(dbet.cpu(), br.grad, rtol=2e-2, atol=2e-2 * br.grad.abs().max().item())
    # bit-reproducible (per-workgroup slabs folded in a fixed order)
    dw1b, dw2b = torch.zeros_like(dw1), (torch.zeros_like(dw2) if c2 else None)
    ddyh2 = to_dev(dyh, dtype)
    ddyh2.gnb_partials = ddyh.gnb_partials
    dx1b, _ = ops.gn_bwd_fused(d1, d2, G, st, gamma.to(DEV), ddyh2, dgam, dbet, skip=(ddout, w1, w2, dw1b, dw2b))
    assert torch.equal(dw1, dw1b) and torch.equal(dx1, dx1b)


GNP_CASES = [  # (N, spatial, cin, cout, circular, out_f32)
    (2, (8, 8, 16), 32, 32, False, False),       # level 0 (NC2, 4x8x16 tiles)
    (1, (5, 9, 20), 32, 32, False, False),       # ragged: partial tiles, zero padding inside the staged image
    (1, (6, 6, 6), 32, 32, True, False),         # circular: every halo voxel is a real one
    (1, (8, 8, 16), 64, 64, False, False),       # two K-blocks, NC4
    (1, (4, 4, 16), 128, 128, False, False),     # deep level: small-grid tiles / half-chunk workgroups
    (2, (8, 8, 16), 32, 1, False, True),         # conv_out: fp32 result
    (1, (8, 8, 16), 96, 32, False, False),       # 3 K-blocks, 12 channels per group
]


@pytest.mark.parametrize("case", GNP_CASES, ids=[f"{c[2]}to{c[3]}{'c' if c[4] else ''}{'f' if c[5] else ''}_{'x'.join(map(str, c[1]))}" for c in GNP_CASES])
def test_conv_fwd_with_groupnorm_prologue(case):
    """Inference: conv(silu(gn(x))) with GroupNorm + SiLU applied to the staged image inside the conv kernel (vdm_conv_fwd_gn)
    == gn_silu_fwd followed by the plain conv, BIT FOR BIT (same arithmetic on the same bf16 operands), and within the conv tolerance of
    the CPU fp32 reference."""
    ops = _ops()
    dtype = torch.bfloat16
    N, sp, cin, cout, circ, f32 = case
    G = 8
    x = (rnd((N,) + sp + (cin,), 21) * 1.3 + 0.2).to(dtype).float()
    gamma, beta = 1.0 + 0.3 * rnd((cin,), 22), 0.2 * rnd((cin,), 23)
    w = rnd((27, cout, cin), 24, scale=(27 * cin) ** -0.5)
    bias = 0.1 * rnd((cout,), 25)
    conv = ops.Conv(cin, cout, 3, circular=circ, out_f32=f32)
    conv.pack(w.to(DEV), dtype, need_dgrad=False)
    dx = to_dev(x, dtype)
    if not conv.gn_in_ok(dx):
        pytest.skip("this shape runs a kernel without the prologue")
    st = ops.gn_stats(dx, None, G)
    a = ops.gn_silu_fwd(dx, None, G, st, gamma.to(DEV), beta.to(DEV))
    ref_dev = conv.fwd(a, bias.to(DEV), gn=not f32)
    out = conv.fwd(dx, bias.to(DEV), gn=not f32, gn_in=(G, st, gamma.to(DEV), beta.to(DEV)))
    assert torch.equal(out, ref_dev), f"max diff {(out.float() - ref_dev.float()).abs().max().item()}"
    if not f32:
        assert torch.equal(out.gn_partials, ref_dev.gn_partials)
    y = F.silu(F.group_norm(x.permute(0, 4, 1, 2, 3), G, gamma, beta, 1e-5)).to(dtype).float()
    wt = w.to(dtype).float().reshape(3, 3, 3, cout, cin).permute(3, 4, 0, 1, 2)
    yp = F.pad(y, (1, 1, 1, 1, 1, 1), mode="circular") if circ else y
    ref = F.conv3d(yp, wt, bias, padding=0 if circ else 1).permute(0, 2, 3, 4, 1)
    e = (out.float().cpu()[..., :cout] - ref).abs().max().item()
    assert e <= 2 * conv_tol(dtype, ref), f"err {e} > {2 * conv_tol(dtype, ref)}"


THIN_CASES = [
    ("in_2_32", 2, (8, 8, 16), 2, 32, 3, 1, 0, False),
    ("in_2_32_ragged", 1, (5, 7, 40), 2, 32, 3, 1, 0, False),        # x chunks of 32 + 8 voxels, zero padding
    ("in_2_32_circ", 1, (4, 6, 20), 2, 32, 3, 1, 0, True),
    ("in_1_16", 2, (6, 6, 33), 1, 16, 3, 1, 0, False),
    ("in_2_64_circ", 1, (3, 5, 64), 2, 64, 3, 1, 0, True),
    ("out_32_1", 2, (8, 8, 16), 32, 1, 3, 1, 0, False),
    ("out_32_1_ragged_circ", 1, (5, 7, 40), 32, 1, 3, 1, 0, True),
    ("out_16_1", 1, (6, 6, 33), 16, 1, 3, 1, 0, False),
    ("out_64_1", 1, (4, 4, 70), 64, 1, 3, 1, 0, False),
    ("in_2_32_big", 1, (32, 32, 32), 2, 32, 3, 1, 0, False),         # many chunks per wave, every workgroup busy
]


@pytest.mark.parametrize("case", THIN_CASES, ids=[c[0] for c in THIN_CASES])
def test_wgrad_thin_side(case):
    """Weight gradient of conv_in / conv_out shaped convs (one side <= 2 channels: csrc/wgrad_thin.hip) against torch.autograd of the
    reference conv; bf16 storage: 2^-7 of max|dW|; bit-reproducible; the generic kernel (VDM4CDM_NO_THIN_WGRAD) agrees."""
    dtype = torch.bfloat16
    ops, conv, x, w, xd = _conv_setup(case, dtype, seed=30)
    name, N, (D, H, W), cin, cout, ks, stride, ups, circ = case
    dout = rnd((N, D, H, W, cout), 31, dtype)
    wr = w.clone().requires_grad_(True)
    ref_conv(x, wr, None, None, None, ks, stride, ups, circ).backward(dout)
    dd = torch.zeros((N, D, H, W, ops.cpad(cout, dtype)), dtype=dtype, device=DEV)
    dd[..., :cout] = dout.to(dtype).to(DEV)
    dw = torch.full((27, cout, cin), float("nan"), device=DEV)
    db = torch.full((cout,), float("nan"), device=DEV) if cout > 1 else None
    conv.wgrad(xd, dd, dw, db)
    tol = 2.0 ** -7 * max(wr.grad.abs().max().item(), 1e-6)
    err = (dw.cpu() - wr.grad).abs().max().item()
    assert err <= tol, f"{name}: wgrad err {err} > {tol}"
    if db is not None:
        bref = dout.reshape(-1, cout).sum(0)
        assert (db.cpu() - bref).abs().max().item() <= 1e-3 * max(bref.abs().max().item(), 1.0) + 1e-3
    dw2 = torch.zeros_like(dw)
    conv.wgrad(xd, dd, dw2, torch.zeros_like(db) if db is not None else None)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("case", [(2, (8, 8, 32), 32, 2, False, True), (1, (5, 7, 40), 32, 2, False, False), (1, (4, 6, 36), 16, 1, True, True),
                                  (3, (3, 4, 70), 64, 2, True, True)], ids=["c32", "c32_ragged_noadd", "c16_circ", "c64_circ"])
def test_last_groupnorm_backward_feeds_conv_in_wgrad(case):
    """vdm_gn_bwd_apply_wgrad_thin: norm1's apply pass of the first block + conv_in's weight / bias gradient in one pass == the apply
    pass followed by the weight-gradient kernel, BIT FOR BIT (dx is rounded to bf16 exactly like the tensor it replaces), and dgamma /
    dbeta equal."""
    ops = _ops()
    dtype = torch.bfloat16
    N, sp, C, cin, circ, with_add = case
    G = 8
    x = (rnd((N,) + sp + (C,), 41) * 1.2 + 0.1).to(dtype).float()
    dyh = rnd((N,) + sp + (C,), 42, dtype)
    add = rnd((N,) + sp + (C,), 43, dtype) if with_add else None
    xin = torch.zeros((N,) + sp + (8,))
    xin[..., :cin] = rnd((N,) + sp + (cin,), 44, dtype)
    gamma = 1.0 + 0.3 * rnd((C,), 45)
    conv_in = ops.Conv(cin, C, 3, circular=circ)
    dx_, ddyh, dxin = to_dev(x, dtype), to_dev(dyh, dtype), to_dev(xin, dtype)
    dadd = to_dev(add, dtype) if with_add else None
    assert ops.gn_tail_ok(conv_in, dx_)
    st = ops.gn_stats(dx_, None, G)
    part = ops.channel_dot_sums(ddyh, dx_, None)
    # reference: apply pass, then the weight gradient
    d1 = ddyh.clone()
    d1.gnb_partials = part
    dg0, db0 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dxr, _ = ops.gn_bwd_fused(dx_, None, G, st, gamma.to(DEV), d1, dg0, db0, add1=dadd)
    dw0, dbias0 = torch.zeros(27, C, cin, device=DEV), torch.zeros(C, device=DEV)
    conv_in.wgrad(dxin, dxr, dw0, dbias0)
    # fused
    d2 = ddyh.clone()
    d2.gnb_partials = part
    dg1, db1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dw1, dbias1 = torch.full((27, C, cin), 5.0, device=DEV), torch.full((C,), 5.0, device=DEV)
    out = ops.gn_bwd_fused(dx_, None, G, st, gamma.to(DEV), d2, dg1, db1, add1=dadd, tail=(conv_in, dxin, dw1, dbias1))
    assert out == (None, None)
    assert torch.equal(dw0, dw1) and torch.equal(dbias0, dbias1), f"max diff {(dw0 - dw1).abs().max().item()}"
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert dw1.abs().max().item() > 0


# ------------------------------------------------------------------------------------------ fused dgrad (+ folded GroupNorm backward) + wgrad
DGW_CASES = [  # name, N, (D,H,W), c1, c2 (GroupNorm input = conv input: c1 + c2 = 32), dropout p, circular, bias gradient
    ("l0_like", 2, (16, 64, 128), 32, 0, 0.1, False, True),           # the level-0 shape in small: whole tiles, dropout mask, bias
    ("ragged", 2, (13, 60, 120), 32, 0, 0.0, False, False),           # ragged in z (odd: half a step), y and x; no bias
    ("circular_concat", 3, (10, 44, 128), 16, 16, 0.1, True, True),   # two-source GroupNorm input (16 + 16), circular padding, 3 samples
]


@pytest.mark.parametrize("case", DGW_CASES, ids=[c[0] for c in DGW_CASES])
def test_fused_dgrad_wgrad_kernel(case):
    """Conv.dgrad_gn_wgrad (csrc/conv_dgw.hip: one launch stages dout once and produces the folded input gradient, the GroupNorm
    partial sums, the weight gradient and the bias gradient of a 32 -> 32 conv) against (a) torch.autograd of
    x -> dropout(silu(group_norm(x))) -> conv3d on the CPU and (b) the two separate kernels it replaces: dyh bit for bit (the same
    27-tap MFMA order per output), weight / bias gradients and GroupNorm sums to fp32 re-association."""
    ops = _ops()
    dtype = torch.bfloat16
    name, N, (D, H, W), c1, c2, p, circ, want_bias = case
    C, cout, G = c1 + c2, 32, 8

    def produce(cx, seed):
        cv = ops.Conv(8, cx, 3, circular=circ)
        cv.pack(rnd((27, cx, 8), seed, dtype, scale=1.0 / math.sqrt(27 * 8)).to(DEV), dtype, need_dgrad=False)
        return cv.fwd(to_dev(rnd((N, D, H, W, 8), seed + 1, dtype), dtype), (0.3 * rnd((cx,), seed + 2)).to(DEV), gn=True)
    x1 = produce(c1, 10)
    x2 = produce(c2, 20) if c2 else None
    gamma, beta = (1.0 + 0.3 * rnd((C,), 3)).to(DEV), (0.2 * rnd((C,), 4)).to(DEV)
    st = ops.gn_stats(x1, x2, G, chsum=True)
    y = ops.gn_silu_fwd(x1, x2, G, st, gamma, beta, p, 777, want_mask=True)          # the conv's saved input a
    w = rnd((27, cout, C), 30, dtype, scale=1.0 / math.sqrt(27 * C))
    conv = ops.Conv(C, cout, 3, circular=circ)
    conv.pack(w.to(DEV), dtype, need_dgrad=True)
    dd = to_dev(rnd((N, D, H, W, cout), 31, dtype), dtype)
    assert conv.dgw_ok(dd, c1, c2)
    # (b) the separate kernels
    dyh_s = conv.dgrad_gn(dd, x1, x2, G, st, gamma, beta, keep_mask=y.keep_mask, dropout_p=p)
    dw_s, db_s = torch.zeros(27, cout, C, device=DEV), torch.zeros(cout, device=DEV)
    conv.wgrad(y, dd, dw_s, db_s)
    # the fused launch
    dw_f = torch.full((27, cout, C), float("nan"), device=DEV)
    db_f = torch.full((cout,), float("nan"), device=DEV) if want_bias else None
    dyh_f = conv.dgrad_gn_wgrad(dd, y, x1, x2, G, st, gamma, beta, dw_f, db_f, keep_mask=y.keep_mask, dropout_p=p)
    assert torch.equal(dyh_f, dyh_s), f"{name}: folded input gradient differs from vdm_conv_dgrad_gn ({(dyh_f.float() - dyh_s.float()).abs().max().item():.3e})"
    wmax = dw_s.abs().max().item()
    assert (dw_f - dw_s).abs().max().item() <= 2e-5 * wmax, f"{name}: dW vs vdm_conv_wgrad {(dw_f - dw_s).abs().max().item():.3e} / {wmax:.3e}"
    if want_bias:
        assert (db_f - db_s).abs().max().item() <= 2e-5 * db_s.abs().max().item() + 1e-4
    s_f, s_s = dyh_f.gnb_partials.sum(1), dyh_s.gnb_partials.sum(1)              # per-(sample, channel) totals of the tile sums
    assert (s_f - s_s).abs().max().item() <= 1e-4 * s_s.abs().max().item()
    # (a) autograd on the CPU: weight gradient of the conv on the activated tensor the kernels saw
    yr = y.float().cpu().requires_grad_(False)
    wr = w.float().clone().requires_grad_(True)
    ref_conv(yr, wr, None, None, None, 3, 1, 0, circ).backward(dd.float().cpu())
    err = (dw_f.cpu() - wr.grad).abs().max().item()
    assert err <= 2e-3 * wr.grad.abs().max().item(), f"{name}: dW vs autograd {err:.3e} / {wr.grad.abs().max().item():.3e}"
    if want_bias:
        bref = dd.float().cpu().sum((0, 1, 2, 3))
        assert (db_f.cpu() - bref).abs().max().item() <= 1e-3 * bref.abs().max().item() + 1e-3
    # and the folded backward through gn_bwd_fused is the one of the separate path
    outs = []
    for dyh in (dyh_f, dyh_s):
        dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dy_ = dyh.clone() if x2 is None else dyh                 # (one source: the apply pass writes dx in place)
        dy_.gnb_partials = dyh.gnb_partials
        dx1, dx2 = ops.gn_bwd_fused(x1, x2, G, st, gamma, dy_, dgam, dbet)
        outs.append((dx1.float(), dgam, dbet))
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 2.0 ** -7 * outs[1][0].abs().max().item()
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-4 * outs[1][1].abs().max().item() + 1e-5
