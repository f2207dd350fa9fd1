"""Validation figure (optional, matplotlib): conditioning / ground truth / sample images, histograms, P(k), cross-correlation.
Stand-in for the diagnostic ``utils.draw_figure`` of the reference (/root/reference/src/utils.py:131-202); not on the hot path."""
import numpy as np


def draw_figure(batch, samples, index=0, fontsize=16, x_to_im=None, conditioning_to_im=None, conditioning_values_to_str=None,
                pk_func=None, cc_func=None):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    x, cond, vals = batch["x"], batch.get("conditioning"), batch.get("conditioning_values")
    fig, axes = plt.subplots(2, 3, figsize=(20, 12))
    ax = axes.flat
    if cond is not None and conditioning_to_im is not None:
        ax[0].imshow(conditioning_to_im(cond[index]))
        ax[0].set_title("Conditioning", fontsize=fontsize)
    if x_to_im is not None:
        for a, t, title in ((ax[1], x, "GT Target"), (ax[2], samples, "Sampled Target")):
            a.imshow(x_to_im(t[index]))
            a.set_title(title, fontsize=fontsize)
    bins = np.linspace(-4, 4, 50)
    for name, t in (("GT", x), ("Sampled", samples), ("Conditioning", cond)):
        if t is None:
            continue
        for ic in range(t.shape[1]):
            ax[3].hist(t[index, ic].detach().float().cpu().numpy().ravel(), bins=bins, histtype="step", label=f"{name} Channel {ic}")
    ax[3].legend(fontsize=fontsize)
    if pk_func is not None:
        for name, t in (("GT", x), ("Sampled", samples), ("Conditioning", cond)):
            if t is None:
                continue
            for ic in range(t.shape[1]):
                ks, pks = pk_func(t[index, ic], ic)
                ax[4].plot(ks, pks, label=f"{name} Channel {ic}")
        ax[4].set(xscale="log", yscale="log", xlabel="k/k_grid", ylabel="Raw Pk", title="Powerspectra")
        ax[4].legend(fontsize=fontsize)
    if cc_func is not None:
        for ic in range(x.shape[1]):
            ks, ccs = cc_func(x[index, ic], samples[index, ic], ic)
            ax[5].plot(ks, ccs, label=f"CC GT-Sampled Channel {ic}")
        ax[5].set(xscale="log", xlabel="k", ylabel="CC", title="Cross Correlation")
        ax[5].legend(fontsize=fontsize)
    if conditioning_values_to_str is not None and vals is not None:
        ax[0].annotate(conditioning_values_to_str(vals[index] if not isinstance(vals, (list, tuple)) else vals[0][index]),
                       xy=(0, 0), xytext=(0.5, 0.5), textcoords="axes fraction", fontsize=fontsize, ha="center", va="center")
    return fig
