"""Summary statistics of generated cubes on the device - the consumer side of the sampling path.

Mirrors /root/reference/calc_SS.py: ``get_pk_3d`` / ``get_pk_2d`` (:67-75, P(k) of the field normalised to unit sum),
``get_logpdf_3d`` / ``get_logpdf_2d`` (:51-65, histograms of log10(rho + 1) over 99 bins), ``get_stats`` (:77-99: mean / std / P(k) /
log-PDF of the cube and of its half- and quarter-depth slab projections) and the driver loop (:107-243) that walks the generated
``gen_*.npy`` files of a model and writes ``summary.pth``.  Everything runs on the tensor's device (rocFFT through ``utils.pk``,
device-side bucketize + bincount for the histograms); results are moved to numpy only where the reference returns numpy.
The reduced wavelet-scattering statistics (``*_rwst``) need ``mltools.archive.LWT``, which is not in the reference tree: those keys
are omitted (out of scope).  Checked against golden vectors of the reference functions (tests/golden/ss_golden.npz).
"""
import os

import numpy as np
import torch

from . import utils

_EDGES = {}


def _edges(lo, hi, n, device):
    key = (lo, hi, n, str(device))
    if key not in _EDGES:                     # np.linspace in float64: exactly the bin edges np.histogram gets in the reference
        _EDGES[key] = torch.from_numpy(np.linspace(lo, hi, n)).to(device)
    return _EDGES[key]


def _log_hist(fields, lo, hi, n_edges):
    """np.histogram(log10(fields + 1), bins=np.linspace(lo, hi, n_edges)) per sample, on the device: int64 counts [B, n_edges-1].
    Bins are [e_i, e_{i+1}) and the last one is closed on the right, values outside [lo, hi] are dropped (numpy semantics)."""
    B = fields.shape[0]
    x = torch.log10(fields + 1).reshape(B, -1).to(torch.float64)          # (the reference takes log10 in fp32, then bins in fp64)
    e = _edges(lo, hi, n_edges, fields.device)
    nb = n_edges - 1
    idx = torch.bucketize(x, e, right=True) - 1
    idx = torch.where(x == e[-1], torch.full_like(idx, nb - 1), idx)
    valid = (idx >= 0) & (idx < nb)
    flat = (idx + nb * torch.arange(B, device=fields.device)[:, None])[valid]
    return torch.bincount(flat, minlength=B * nb).reshape(B, nb)


def get_logpdf_3d(fields):
    return _log_hist(fields, 8.5, 15.0, 100).cpu().numpy()


def get_logpdf_2d(fields):
    return _log_hist(fields, 10.5, 15.5, 100).cpu().numpy()


def get_pk_3d(fields):
    fields_u = fields / fields.sum((2, 3, 4), keepdim=True)
    return utils.pk(fields_u)[1].detach().cpu().numpy()


def get_pk_2d(fields):
    fields_u = fields / fields.sum((2, 3), keepdim=True)
    return utils.pk(fields_u)[1].detach().cpu().numpy()


def get_stats(fields, resol=None):
    """fields: un-normalised densities (B, 1, D, D, D) on any device.  resol: the `res` entry of configs.yaml (default: D)."""
    resol = fields.shape[-1] if resol is None else resol
    half, quarter = resol // 2, resol // 4
    stats = {"3d_mean": fields.mean().item(), "3d_std": fields.std().item(), "3d_pk": get_pk_3d(fields),
             "3d_logpdf": get_logpdf_3d(fields)}
    for tag, depth in (("half", half), ("quarter", quarter)):
        proj = fields[:, :, :depth].sum(2)
        stats[f"2d_{tag}_mean"] = proj.mean().item()
        stats[f"2d_{tag}_std"] = proj.std().item()
        stats[f"2d_{tag}_pk"] = get_pk_2d(proj)
        stats[f"2d_{tag}_logpdf"] = get_logpdf_2d(proj)
    return stats


def _images(store, prefix, x_unnorm, half, quarter):
    store[f"half_{prefix}"] = x_unnorm[:, :, :half].sum(2).detach().cpu().numpy()
    store[f"quarter_{prefix}"] = x_unnorm[:, :, :quarter].sum(2).detach().cpu().numpy()


def main(argv=None, configs_path=None):
    """``python calc_SS.py <model_name>``: statistics of every generated set found under <data_dir>/<model_name>/ -> summary.pth."""
    import argparse
    import yaml
    ap = argparse.ArgumentParser(description="Summary statistics of generated 3D CDM")
    ap.add_argument("model_name", type=str, help="Model name")
    args = ap.parse_args(argv)
    data_fol = os.path.join(os.environ.get("VDM4CDM_GEN_DIR", "./data/ICML_v2/"), args.model_name)
    assert os.path.exists(data_fol), f"{data_fol} does not exist"
    device = "cuda" if torch.cuda.is_available() else "cpu"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = yaml.safe_load(open(configs_path or os.path.join(root, "configs.yaml")))[args.model_name]
    resol = config.get("res", config.get("cropsize", 128))
    half, quarter = resol // 2, resol // 4
    summary = {}
    for key in ["CV_1_128", "CV_12_12", "1P_24", "1P_128"]:
        fol = os.path.join(data_fol, key)
        if not os.path.exists(fol):
            continue
        print("Processing", fol, flush=True)
        config["data_params"].update(set_name=key.split("_")[0], stage="test", batch_size=1)
        dm = utils.get_datamodule(config)
        ss, images = {}, {}

        def ground_truth(batch, tag):
            x = dm.unnorm_func(batch["x"].to(device), 1)
            c = dm.unnorm_func(batch["conditioning"].to(device), 0)
            ss[f"Mcdm_GT_{tag}"] = get_stats(x, resol)
            _images(images, f"Mcdm_GT_{tag}", x, half, quarter)
            _images(images, f"cond_GT_{tag}", c, half, quarter)

        def generated(data, tag_of):
            for j in range(data.shape[0]):
                x = dm.unnorm_func(torch.as_tensor(data[[j]]).to(device), 1)
                ss[f"Mcdm_{tag_of(j)}"] = get_stats(x, resol)
                _images(images, f"Mcdm_{tag_of(j)}", x, half, quarter)

        results = {"stats": ss, "images": images}
        if key == "CV_1_128":
            for i_batch, batch in enumerate(dm.test_dataloader()):
                if i_batch == 2:
                    ground_truth(batch, 0)
                    break
            data = np.load(os.path.join(fol, "gen_0.npy"))
            generated(data, lambda j: f"0_{j}")
            x_all = dm.unnorm_func(torch.as_tensor(data).to(device), 1)
            results["post_means"], results["post_stds"] = x_all.mean(0, keepdim=True), x_all.std(0, keepdim=True)
        elif key == "CV_12_12":
            for i_batch, batch in enumerate(dm.test_dataloader()):
                if i_batch == 12:
                    break
                ground_truth(batch, i_batch)
            for i in range(12):
                generated(np.load(os.path.join(fol, f"gen_{i}.npy")), lambda j, i=i: f"{i}_{j}")
        else:
            rep = int(os.environ.get("VDM4CDM_REP", 24 if key == "1P_24" else 128))      # (same override as the generate scripts)
            i_gens, names = [0, 4, 7, 23, 28], ["fid", "Om_m2", "Om_p2", "ASN1_m3", "ASN1_p3"]
            for i_batch, batch in enumerate(dm.test_dataloader()):
                if i_batch in i_gens:
                    ground_truth(batch, names[i_gens.index(i_batch)])
            for name in names:
                # reference quirk (calc_SS.py:232-236): 1P_128 is read as <name>.npy although generate_3D_1P.py writes <name>_<rep>.npy;
                # accept the reference's name first (files renamed by hand, as its authors must have) and fall back to what the
                # in-repo generate_3D_1P.py actually wrote, so that generate -> calc_SS runs without a manual rename
                cands = [os.path.join(fol, f"{name}_{rep}.npy")] if key == "1P_24" else \
                    [os.path.join(fol, f"{name}.npy"), os.path.join(fol, f"{name}_{rep}.npy")]
                f = next((c for c in cands if os.path.exists(c)), cands[0])
                assert os.path.exists(f), f"File {f} does not exist"
                generated(np.load(f), lambda j, name=name: f"{name}_{j}")
        summary[key] = results
    torch.save(summary, os.path.join(data_fol, "summary.pth"))
    return summary
