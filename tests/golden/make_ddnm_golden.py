"""Generate golden vectors from the one sampler loop whose SOURCE is in the reference tree.

Runs ONLY in the build container (needs /root/reference).  Imports the reference's own ``src/utils.py`` (same one-symbol stub as
``make_pk_golden.py``) and runs its ``get_ddnm_result`` (/root/reference/src/utils.py:277-304: DDNM range/null-space sampler with
time travel) on a duck-typed ``vdm`` object backed by this repo's CPU oracle (``oracle/unet_oracle.py`` + ``oracle/vdm_oracle.py``).
``torch.randn`` / ``torch.randn_like`` are patched for the duration of the call so that every noise draw comes from ONE seeded CPU
generator in call order; the tests replay the same stream.  What this pins: the reference's loop structure (time-travel indices,
operator algebra ``ATy + x0 - AT(A(x0))``, the order of noise draws, the ``sample_zt_given_zs`` / ``sample_zs_given_zt(return_ddnm=True)``
contract, the ``conditioning=None`` kwarg and ``**kwargs`` pass-through) - NOT the network arithmetic (mltools is not in the tree;
see DESIGN.md section 0).  Only data is written (inputs by seed, outputs): tests/golden/ddnm_golden.npz.

    python tests/golden/make_ddnm_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "ddnm_golden.npz")

# (name, D, chs, net seed, B, n_sampling_steps, l, operator, conditional)
CASES = [
    ("mask_l0", 16, (16, 32), 4, 2, 6, 0, "mask", False),
    ("mask_l2", 16, (16, 32), 4, 2, 8, 2, "mask", False),
    ("mask_l2_cond", 16, (16, 32), 5, 1, 6, 2, "mask", True),
    ("pool_larr", 16, (16, 32), 6, 1, 7, [0, 1, 2, 3, 2, 1, 0], "pool", False),
]
NOISE_SEED = 20240601


class NoiseStream:
    """Every torch.randn / torch.randn_like inside the `with` block draws from one seeded CPU generator, in call order
    (fp32 CPU draw, then moved to the requested device): the same stream can be replayed next to any backend."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.calls = 0

    def _draw(self, shape, device=None):
        self.calls += 1
        x = torch.randn(tuple(shape), generator=self.g, dtype=torch.float32, device="cpu")
        return x if device is None else x.to(device)

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like
        stream = self

        def randn(*size, generator=None, device=None, dtype=None, **kw):
            if generator is not None:                      # an explicitly seeded draw is not part of the stream
                return stream._randn(*size, generator=generator, device=device, dtype=dtype, **kw)
            if len(size) == 1 and not isinstance(size[0], int):
                size = tuple(size[0])
            return stream._draw(size, device)

        def randn_like(t, **kw):
            return stream._draw(t.shape, t.device)

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def operators(kind, shape, device="cpu"):
    """(A, AT) with A AT = identity on the range: `mask` keeps half of the cube; `pool` is 2x average pooling along x with
    AT = its pseudo-inverse (nearest up-sampling)."""
    if kind == "mask":
        m = torch.zeros(shape, device=device)
        m[..., : shape[-1] // 2] = 1.0
        return (lambda x: x * m), (lambda x: x * m)
    if kind == "pool":
        A = lambda x: 0.5 * (x[..., 0::2] + x[..., 1::2])
        AT = lambda y: y.repeat_interleave(2, dim=-1)
        return A, AT
    raise ValueError(kind)


def case_inputs(case, gold=None):
    """Everything a test needs to replay a case: product CUNet with seeded weights (CPU), measurement y, kwargs.
    The fields y / s_conditioning / v_conditionings come from the fixture when `gold` (the loaded npz) is given: they are made with
    FFTs here, and FFT round-off differs between host CPUs (the weights are plain seeded uniform/normal draws and do reproduce)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    from helpers import grf, randomize
    from vdm4cdm_amd.networks import CUNet
    name, D, chs, seed, B, n, l, op, cond = case
    net = CUNet(shape=(1, D, D, D), chs=list(chs), s_conditioning_channels=1 if cond else 0, v_conditioning_dims=[6] if cond else [],
                norm_groups=8, backend="torch", precision="fp32")
    randomize(net, seed, zero_init_std=0.02)
    shape = (B, 1, D, D, D)
    if gold is not None:
        y = torch.from_numpy(gold[f"{name}/y"])
        kwargs = {}
        if cond:
            kwargs = {"s_conditioning": torch.from_numpy(gold[f"{name}/s_conditioning"]),
                      "v_conditionings": [torch.from_numpy(gold[f"{name}/v_conditioning"])]}
        return net, y, kwargs
    A, AT = operators(op, shape)
    y = A(grf(shape, 300 + seed))
    kwargs = {}
    if cond:
        kwargs = {"s_conditioning": grf(shape, 400 + seed),
                  "v_conditionings": [torch.rand(B, 6, generator=torch.Generator().manual_seed(500 + seed))]}
    return net, y, kwargs


class _OracleVDMModel:
    """`vdm.model` as the reference loop uses it (src/utils.py:287,294,296), backed by the oracle."""

    def __init__(self, net):
        from helpers import oracle_cfg, oracle_params
        from oracle import vdm_oracle
        self.P, self.cfg = oracle_params(net), oracle_cfg(net)
        self.score_model = types.SimpleNamespace(shape=net.shape)
        self.sched = vdm_oracle.Schedule(-13.3, 13.3)
        self.gamma_min, self.gamma_max, self.w_cfg = -13.3, 13.3, None

    def _score(self, kwargs):
        from oracle import unet_oracle
        s, v = kwargs.get("s_conditioning"), kwargs.get("v_conditionings") or ()
        return lambda z, tn: unet_oracle.cunet_forward(self.P, self.cfg, z, tn, s, v)

    def sample_zt_given_zs(self, zs, t, s):
        from oracle import vdm_oracle
        return vdm_oracle.sample_zt_given_zs(self.sched, zs, t, s, torch.randn_like(zs))

    def sample_zs_given_zt(self, zt, t, s, return_ddnm=False, conditioning=None, **kwargs):
        from oracle import vdm_oracle
        assert return_ddnm and conditioning is None
        return vdm_oracle.sample_zs_given_zt(self._score(kwargs), self.sched, zt, t, s, None, return_ddnm=True)


def import_reference_utils():
    stub = types.ModuleType("mltools")
    stub_ml = types.ModuleType("mltools.ml_utils")
    stub_ml.to_np = lambda t: t.detach().cpu().numpy()
    stub.ml_utils = stub_ml
    sys.modules["mltools"] = stub
    sys.modules["mltools.ml_utils"] = stub_ml
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")
    from src import utils as ref_utils
    return ref_utils


def main():
    ref_utils = import_reference_utils()
    out = {}
    for case in CASES:
        name, D, chs, seed, B, n, l, op, cond = case
        net, y, kwargs = case_inputs(case)
        vdm = types.SimpleNamespace(device=torch.device("cpu"), model=_OracleVDMModel(net))
        A, AT = operators(op, (B, 1, D, D, D))
        with NoiseStream(NOISE_SEED + seed) as ns:
            x_r = ref_utils.get_ddnm_result(vdm, y, A, AT, n_sampling_steps=n, l=l, **kwargs)
            x_all = None
        with NoiseStream(NOISE_SEED + seed):
            x_all = ref_utils.get_ddnm_result(vdm, y, A, AT, n_sampling_steps=n, l=l, return_all=True, **kwargs)
        assert torch.equal(x_all[-1], x_r)
        flat = net.flat.detach().double()
        out[f"{name}/x"] = x_r.numpy()
        out[f"{name}/x_all_absmax"] = x_all.abs().amax(dim=tuple(range(1, x_all.dim()))).numpy()
        out[f"{name}/weights_check"] = np.array([flat.sum().item(), (flat ** 2).sum().item()])
        out[f"{name}/y"] = y.numpy()
        if cond:
            out[f"{name}/s_conditioning"] = kwargs["s_conditioning"].numpy()
            out[f"{name}/v_conditioning"] = kwargs["v_conditionings"][0].numpy()
        out[f"{name}/noise_calls"] = np.array([ns.calls], dtype=np.int64)
        print(name, "x", tuple(x_r.shape), "max|x|", float(x_r.abs().max()), "noise draws", ns.calls)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
