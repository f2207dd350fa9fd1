"""``sfm_model.SFM`` / ``sfm_model.LightSFM`` - flow matching between two fields on the same CUNet.

Mirrors the interface the reference uses for ``mltools.models.sfm_model``:
* construction ``LightSFM(velocity_model=, draw_figure=, learning_rate=)``
  (/root/reference/trainSFM3D128_c_c_from_field_name_thick_lowbatch.py:124-127, trainSFM_c_uc_from_field_name.py:116-119)
* batch dict ``{"x0": source field, "x1": target field, "conditioning_values": [params] | None}`` (:71-72) - the velocity network is
  the same ``CUNet(shape, chs, s_conditioning_channels=1, v_conditioning_dims, ...)`` as the VDM's score network (:112-123)
* validation figure ``draw_figure_sfm(batch, samples)`` reads ``batch["x1"]`` / ``batch["x0"]`` (/root/reference/src/utils.py:204-207)
The model source (``mltools.models.sfm_model``) is not in the reference tree and ``utils.get_model`` returns nothing for
``type: SFM`` (/root/reference/src/utils.py:472-473), so the arithmetic is spec D14 [INFERRED], the conditional flow matching of
Lipman et al. 2023 / Tong et al. 2023 (I-CFM) on the straight path between the paired fields:

    x_t = (1 - t) x0 + t x1 + sigma eps,   t ~ U(0, 1) (antithetic over the batch),   target velocity u = x1 - x0
    loss = mean over batch and elements of (v_theta(x_t, t; s_conditioning = x0, v_conditionings) - u)^2
    sample: x <- x + v_theta(x, t_i; x0, v) / n,  t_i = i / n, starting from x = x0       (explicit Euler, n steps)

On the HIP backend the interpolation / target are the K7 kernel (``vdm_diffuse``: a x + b y per sample), the loss and its
gradient the K8 kernel, and the Euler loop is the same captured hipGraph as the VDM sampler (``hip_graph_sampler``) with the
coefficient table {1, -1/n, 0, t_i}.
"""
import torch
import torch.nn as nn

from .vdm_model import _DiffusionLossFn, hip_graph_sampler, stratified_times


class SFM(nn.Module):
    def __init__(self, velocity_model, sigma=0.0, antithetic_time_sampling=True):
        super().__init__()
        self.velocity_model = velocity_model
        self.sigma = float(sigma)
        self.antithetic_time_sampling = antithetic_time_sampling

    @property
    def score_model(self):                                   # (the trainer and the DDP hook address the network under this name)
        return self.velocity_model

    def _hip(self, ref):
        return getattr(self.velocity_model, "backend", None) == "hip" and ref.is_cuda

    def velocity(self, xt, t, x0, v_conditionings=None):
        net = self.velocity_model
        kw = {}
        if getattr(net, "s_conditioning_channels", 1):
            kw["s_conditioning"] = x0
        return net(xt, t=t, v_conditionings=list(v_conditionings or []), **kw)

    def get_loss(self, x0, x1, times=None, eps=None, v_conditionings=None):
        B, numel = x1.shape[0], x1[0].numel()
        x0, x1 = x0.to(torch.float32).contiguous(), x1.to(torch.float32).contiguous()
        t = stratified_times(B, x1.device, self.antithetic_time_sampling) if times is None else times.to(x1.device, torch.float32)
        if self._hip(x1):
            from . import hip_ops as ops
            one = torch.ones(B, device=x1.device)
            xt = ops.diffuse(x1, x0, t.contiguous(), (1.0 - t).contiguous())            # t x1 + (1 - t) x0
            if self.sigma > 0.0:
                if eps is None:
                    from .vdm_model import VDM, noise_seed
                    eps = ops.randn(torch.empty_like(x1), noise_seed(), 2 * VDM._rank_world()[0] + 1)     # (per-rank Philox stream)
                xt = ops.diffuse(xt, eps.contiguous(), one, self.sigma * one)
            u = ops.diffuse(x1, x0, one, -one)                                           # target velocity x1 - x0
            v = self.velocity(xt, t, x0, v_conditionings)
            coef = torch.full((B,), 2.0 / (B * numel), device=x1.device)                 # loss = 0.5 sum_n coef_n S_n = mean sq. error
            sums = torch.zeros(B, 3, device=x1.device)
            loss = _DiffusionLossFn.apply(v, u, u, u, 0.0, coef, sums)
        else:
            bc = (B,) + (1,) * (x1.dim() - 1)
            xt = (1.0 - t).view(bc) * x0 + t.view(bc) * x1
            if self.sigma > 0.0:
                xt = xt + self.sigma * (torch.randn_like(x1) if eps is None else eps)
            v = self.velocity(xt, t, x0, v_conditionings)
            loss = ((v - (x1 - x0)) ** 2).mean()
        return loss, {"loss": loss.detach()}

    @staticmethod
    def step_table(n):
        """[n, 4] = {1, -dt, 0, t_i} for hip_graph_sampler: x <- 1 * (x - (-dt) v) + 0 * noise, network time t_i = i / n."""
        t = torch.arange(n, dtype=torch.float64) / n
        return torch.stack([torch.ones(n, dtype=torch.float64), torch.full((n,), -1.0 / n, dtype=torch.float64),
                            torch.zeros(n, dtype=torch.float64), t], dim=1)

    @torch.no_grad()
    def sample(self, x0, n_sampling_steps, v_conditionings=None, return_all=False, verbose=False, use_graph=True):
        x0 = x0.to(torch.float32).contiguous()
        x = x0.clone()
        if self._hip(x) and not return_all:
            coef = self.step_table(n_sampling_steps).to(device=x.device, dtype=torch.float32).contiguous()
            s_cond = x0 if getattr(self.velocity_model, "s_conditioning_channels", 1) else None
            return hip_graph_sampler(self.velocity_model, x, coef, None, 0, verbose, use_graph, s_cond, list(v_conditionings or []))
        xs = []
        for i in range(n_sampling_steps):
            t = torch.full((x.shape[0],), i / n_sampling_steps, device=x.device)
            x = x + self.velocity(x, t, x0, v_conditionings) / n_sampling_steps
            if return_all:
                xs.append(x)
        return torch.stack(xs, dim=0) if return_all else x


class LightSFM(nn.Module):
    """Stand-in for the reference's LightningModule ``sfm_model.LightSFM`` (same constructor / batch contract); the fit loop is
    vdm4cdm_amd.trainer.Trainer, as for LightVDM."""

    def __init__(self, velocity_model, draw_figure=None, learning_rate=3.0e-4, **sfm_kwargs):
        super().__init__()
        self.model = SFM(velocity_model, **sfm_kwargs)
        self.draw_figure = draw_figure
        self.learning_rate = learning_rate
        self._logged = {}

    @property
    def device(self):
        return self.model.velocity_model.flat.device

    def log_dict(self, d):
        self._logged.update({k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()})

    @property
    def logged(self):
        return {k: float(v) for k, v in self._logged.items()}

    @staticmethod
    def _unpack(batch):
        """(x1, kwargs of draw_samples) - batch contract /root/reference/trainSFM3D128_c_c_from_field_name_thick_lowbatch.py:71-72."""
        kw = {"x0": batch["x0"]}
        if batch.get("conditioning_values") is not None:
            kw["v_conditionings"] = list(batch["conditioning_values"])
        return batch["x1"], kw

    def _filter(self, kw):
        if not getattr(self.model.velocity_model, "v_conditioning_dims", [1]):
            kw["v_conditionings"] = []
        return kw

    def training_step(self, batch, batch_idx=0):
        x1, kw = self._unpack(batch)
        loss, metrics = self.model.get_loss(x1=x1, **self._filter(kw))
        self.log_dict({f"train/{k}": v for k, v in metrics.items()})
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        x1, kw = self._unpack(batch)
        loss, metrics = self.model.get_loss(x1=x1, **self._filter(kw))
        self.log_dict({f"val/{k}": v for k, v in metrics.items()})
        return loss

    def configure_optimizers(self):
        fused = all(p.is_cuda for p in self.parameters())
        opt = torch.optim.AdamW(self.parameters(), lr=self.learning_rate, fused=fused)
        vm = self.model.velocity_model
        if hasattr(vm, "mark_weights_dirty"):
            opt.register_step_post_hook(lambda *_: vm.mark_weights_dirty())
        return opt

    def draw_samples(self, x0=None, n_sampling_steps=100, batch_size=None, verbose=False, return_all=False, **kwargs):
        """Target-field samples for the source fields x0 (batch_size is implied by x0; accepted for the trainer's call)."""
        assert x0 is not None, "LightSFM.draw_samples needs the source field x0"
        return self.model.sample(x0.to(self.device), n_sampling_steps, v_conditionings=kwargs.get("v_conditionings"),
                                 return_all=return_all, verbose=verbose, use_graph=kwargs.get("use_graph", True))

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = {} if destination is None else destination
        self.model.velocity_model.state_dict(destination=out, prefix=prefix + "model.velocity_model.", keep_vars=keep_vars)
        return out

    def load_state_dict(self, state_dict, strict=True):
        pre = "model.velocity_model."
        return self.model.velocity_model.load_state_dict({k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}, strict=strict)
