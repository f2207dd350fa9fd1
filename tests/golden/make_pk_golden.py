"""Generate golden vectors for the in-tree power-spectrum estimators of the reference.

Runs ONLY in the build container (needs /root/reference).  Imports the reference's own
``src/utils.py`` (with a stub for its single missing import ``mltools.ml_utils.to_np``,
an un-vendored third-party symbol that ``power/pk/get_ccs`` never call) and records, for seeded
inputs, the outputs of ``utils.pk`` and ``utils.get_ccs``.  Only data (inputs by seed, outputs)
is written: tests/golden/pk_golden.npz.

    python tests/golden/make_pk_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pk_golden.npz")


def make_field(seed, B, C, D, dim):
    """Seeded lognormal-ish test field, float32, shape (B, C, D[,D],D)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, C) + (D,) * dim, generator=g)
    return torch.exp(0.5 * x).to(torch.float32)


CASES = [  # (name, seed, B, C, D, dim)
    ("d3_8", 11, 2, 1, 8, 3),
    ("d3_16", 12, 2, 1, 16, 3),
    ("d3_32", 13, 1, 1, 32, 3),
    ("d3_16_c2", 14, 2, 2, 16, 3),
    ("d2_16", 15, 3, 1, 16, 2),
    ("d2_32", 16, 2, 1, 32, 2),
]


def main():
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

    out = {}
    for name, seed, B, C, D, dim in CASES:
        x = make_field(seed, B, C, D, dim)
        y = make_field(seed + 100, B, C, D, dim)
        k, p, n = ref_utils.pk(x)
        out[f"{name}/meta"] = np.array([seed, B, C, D, dim], dtype=np.int64)
        out[f"{name}/k"] = k.numpy()
        out[f"{name}/P"] = p.numpy()
        out[f"{name}/N"] = n.numpy()
        _, px, _ = ref_utils.pk(x, y)
        out[f"{name}/Pcross"] = px.numpy()
        kc, cc = ref_utils.get_ccs(x, y, full=False)
        out[f"{name}/cc"] = cc.numpy()
        if dim == 2:   # the reference's full=True uses repeat(n,1,1,1): 2D fields only (utils.py:120)
            _, ccf = ref_utils.get_ccs(x, y, full=True)
            out[f"{name}/cc_full"] = ccf.numpy()
        _, cself = ref_utils.get_ccs(x, x)
        out[f"{name}/cc_self"] = cself.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("/P")})


if __name__ == "__main__":
    main()
