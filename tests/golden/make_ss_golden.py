"""Generate golden vectors for the summary statistics of the reference's ``calc_SS.py`` (P(k) of the normalised field in 3D and of
its slab projections, log-density histograms).

Runs ONLY in the build container (needs /root/reference).  ``calc_SS.py`` is a script: it is executed here with ``runpy`` inside a
scratch directory (a copy of the reference's ``configs.yaml``, an empty ``./data/ICML_v2/<model>/`` so that its main loop finds
nothing to process) with stubs for the two un-vendored ``mltools`` modules it imports at the top (``mltools.archive.LWT`` - wavelet
scattering, never called by the functions recorded here - and ``mltools.utils.cuda_tools.get_freer_device``).  The functions
``get_pk_3d`` / ``get_pk_2d`` / ``get_logpdf_3d`` / ``get_logpdf_2d`` (/root/reference/calc_SS.py:51-75) are then called - the
reference's own code - on seeded density fields.  Only data is written: tests/golden/ss_golden.npz (inputs and outputs).

    python tests/golden/make_ss_golden.py
"""
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "ss_golden.npz")
MODEL = "VDM_Mstar_Mcdm_c_c_128"      # res = 128 in configs.yaml -> half = 64, quarter = 32 (scaled to the case size below)

CASES = [("d32_b2", 21, 2, 32), ("d16_b3", 22, 3, 16)]   # name, seed, batch, D


def density(seed, B, D):
    """Seeded lognormal density cube with log10(rho + 1) spread over the histogram ranges of calc_SS (8.5 .. 15)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 1, D, D, D), generator=g)
    return (10.0 ** (10.5 + 0.9 * x)).to(torch.float32)


def load_reference_functions():
    lwt = types.ModuleType("mltools.archive.LWT")
    lwt.make_wavelets = lambda **kw: (None, [])
    lwt.WST_abs2 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("wavelet statistics are not part of this fixture"))
    lwt.get_rwst = lwt.WST_abs2
    cuda_tools = types.ModuleType("mltools.utils.cuda_tools")
    cuda_tools.get_freer_device = lambda: torch.device("cpu")
    ml_utils = types.ModuleType("mltools.ml_utils")
    ml_utils.to_np = lambda t: t.detach().cpu().numpy()
    for name, mod in (("mltools", types.ModuleType("mltools")), ("mltools.archive", types.ModuleType("mltools.archive")),
                      ("mltools.archive.LWT", lwt), ("mltools.utils", types.ModuleType("mltools.utils")),
                      ("mltools.utils.cuda_tools", cuda_tools), ("mltools.ml_utils", ml_utils)):
        sys.modules[name] = mod
    sys.modules["mltools.archive"].LWT = lwt
    sys.modules["mltools.utils"].cuda_tools = cuda_tools
    sys.modules["mltools"].ml_utils = ml_utils
    import matplotlib
    matplotlib.use("Agg")
    tmp = tempfile.mkdtemp(prefix="ss_golden_")
    shutil.copy(os.path.join(REF, "configs.yaml"), os.path.join(tmp, "configs.yaml"))
    os.makedirs(os.path.join(tmp, "data", "ICML_v2", MODEL))
    cwd, argv = os.getcwd(), sys.argv
    sys.path.insert(0, REF)
    try:
        os.chdir(tmp)
        sys.argv = ["calc_SS.py", MODEL]
        ns = runpy.run_path(os.path.join(REF, "calc_SS.py"), run_name="calc_SS_golden")
    finally:
        os.chdir(cwd)
        sys.argv = argv
        shutil.rmtree(tmp, ignore_errors=True)
    return ns


def main():
    ns = load_reference_functions()
    out = {}
    for name, seed, B, D in CASES:
        f = density(seed, B, D)
        half, quarter = D // 2, D // 4
        out[f"{name}/meta"] = np.array([seed, B, D], dtype=np.int64)
        out[f"{name}/3d_pk"] = ns["get_pk_3d"](f)
        out[f"{name}/3d_logpdf"] = ns["get_logpdf_3d"](f)
        out[f"{name}/3d_mean"] = np.array([f.mean().item(), f.std().item()])
        for tag, depth in (("half", half), ("quarter", quarter)):
            p = f[:, :, :depth].sum(2)                                   # calc_SS.py:84,91 with the case's own resolution
            out[f"{name}/2d_{tag}_pk"] = ns["get_pk_2d"](p)
            out[f"{name}/2d_{tag}_logpdf"] = ns["get_logpdf_2d"](p)
            out[f"{name}/2d_{tag}_mean"] = np.array([p.mean().item(), p.std().item()])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
