"""``networks.CUNet`` - the conditional UNet score network of the VDM, MI355X-native.

Mirrors the interface the reference scripts use for ``mltools.networks.networks.CUNet``
(constructor kwargs: /root/reference/trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:116-127 and
/root/reference/src/utils.py:451-462; call ``score_model(zt, t=..., s_conditioning=..., v_conditionings=...)``:
notebook frames ``vdm_model.py:320-324`` / ``networks.py:259-265``; ``.shape``: /root/reference/src/utils.py:287).
The architecture is spec D1-D8 of SURVEY.md section 8 (the mltools source is not in the reference tree).

All parameters live in ONE flat fp32 vector (``self.flat``): the optimiser, gradient clipping and
the RCCL all-reduce each touch a single tensor.  Conv weights are stored tap-major
``[taps, cout, cin]`` (the layout the HIP pack kernel consumes).

Backends
* ``backend="hip"`` (default): 3D only; every activation-sized op is a hand-written gfx950 kernel
  behind the C-ABI (vdm4cdm_amd/unet_hip.py).  Raises if the library is missing or the input is
  not on a GPU - there is no silent fallback.
* ``backend="torch"``: explicit opt-in, plain PyTorch ops; exists for BASELINE config C1
  (2D 64^2 on CPU, "plumbing, no GPU") only.  Never selected automatically.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

T_EMB_DIM = 64
V_EMB_DIM = 64
GN_EPS = 1e-5


class _Spec:
    """name -> (offset, shape) inside the flat parameter vector (offsets 16-byte aligned)."""

    def __init__(self):
        self.items = {}       # name -> (offset, shape, init, fan_in)
        self.total = 0

    def add(self, name, shape, init, fan_in=1):
        n = 1
        for s in shape:
            n *= s
        self.items[name] = (self.total, tuple(shape), init, fan_in)
        self.total += (n + 3) // 4 * 4


class BlockInfo:
    def __init__(self, name, c1, c2, cout, level, table_off):
        self.name, self.c1, self.c2, self.cout, self.level, self.table_off = name, c1, c2, cout, level, table_off
        self.has_skip = (c1 + c2) != cout


class CUNet(nn.Module):
    def __init__(self, shape, chs=(32, 64, 128, 256), s_conditioning_channels=0, v_conditioning_dims=(),
                 t_conditioning=True, norm_groups=8, mid_attn=False, dropout_prob=0.0,
                 conv_padding_mode="zeros", n_attention_heads=4, backend="hip", precision="bf16"):
        super().__init__()
        assert conv_padding_mode in ("zeros", "circular")
        assert backend in ("hip", "torch")
        self.shape = tuple(shape)
        self.dim = len(self.shape) - 1
        self.in_channels = self.shape[0]
        self.chs = list(chs)
        self.s_conditioning_channels = int(s_conditioning_channels)
        self.v_conditioning_dims = list(v_conditioning_dims)
        self.t_conditioning = bool(t_conditioning)
        self.norm_groups = int(norm_groups)
        self.dropout_prob = float(dropout_prob)
        self.conv_padding_mode = conv_padding_mode
        self.n_attention_heads = int(n_attention_heads)
        self.mid_attn = bool(mid_attn)                # D13: self-attention between the two mid blocks
        assert not self.mid_attn or chs[-1] % self.n_attention_heads == 0, "chs[-1] must be divisible by n_attention_heads"
        if backend == "hip" and self.mid_attn and chs[-1] // self.n_attention_heads not in (16, 32, 64, 96, 128):
            raise ValueError(f"CUNet(backend='hip', mid_attn=True): head width chs[-1] / n_attention_heads = "
                             f"{chs[-1] // self.n_attention_heads} is not built by the fused attention kernels (16, 32, 64, 96, 128)")
        self.backend = backend
        self.precision = precision            # "bf16" | "fp32": activation storage of the HIP backend
        self.taps = 3 ** self.dim

        self.cond_dims = ([4 * self.chs[0]] if self.t_conditioning else []) + [V_EMB_DIM] * len(self.v_conditioning_dims)
        if backend == "hip" and max(self.cond_dims + [0] + self.v_conditioning_dims) > 256:
            raise ValueError(f"CUNet(backend='hip'): the conditioning kernel (K6) holds embeddings of at most 256 values; "
                             f"chs[0] = {self.chs[0]} gives a time embedding of width {4 * self.chs[0]} (use chs[0] <= 64 or backend='torch')")
        self.blocks = self._block_list()
        self.table_width = sum(b.cout for b in self.blocks)
        self.spec = self._build_spec()
        self.flat = nn.Parameter(torch.zeros(self.spec.total, dtype=torch.float32))
        self.reset_parameters()
        self._exec = None
        self.weights_epoch = 0            # see mark_weights_dirty()
        self.grad_synced = False          # set by the HIP backward when it already averaged flat.grad over the ranks (enable_ddp)

    # ------------------------------------------------------------------ structure
    def _block_list(self):
        chs, L = self.chs, len(self.chs)
        blocks, off = [], 0

        def add(name, c1, c2, cout, level):
            nonlocal off
            blocks.append(BlockInfo(name, c1, c2, cout, level, off))
            off += cout

        for i in range(L):
            add(f"downs.{i}.block", chs[max(i - 1, 0)], 0, chs[i], i)
        for j in range(2):
            add(f"mid.{j}", chs[-1], 0, chs[-1], L - 1)
        for i in reversed(range(L - 1)):
            add(f"ups.{i}.block", chs[i], chs[i], chs[i], i)
        return blocks

    def _build_spec(self):
        sp, T, chs, L = _Spec(), self.taps, self.chs, len(self.chs)
        # (1) conditioning projections of all blocks, contiguous per conditioning -> one [sum cout, dim] matrix
        for k, dk in enumerate(self.cond_dims):
            for b in self.blocks:
                sp.add(f"{b.name}.cond.{k}.weight", (b.cout, dk), "linear", dk)
        # (2) conv1 biases of all blocks, contiguous (their gradient is the column sum of the table gradient)
        for b in self.blocks:
            sp.add(f"{b.name}.conv1.bias", (b.cout,), "bias", T * (b.c1 + b.c2))
        # (3) the conditioning MLPs.  Everything the conditioning path reads lies in [0, head_end): the torch backend slices the flat
        # vector ONCE there (one autograd gradient for these parameters instead of one full-size tensor per parameter view).
        if self.t_conditioning:
            sp.add("t_embed.0.weight", (4 * chs[0], T_EMB_DIM), "linear", T_EMB_DIM)
            sp.add("t_embed.0.bias", (4 * chs[0],), "bias", T_EMB_DIM)
            sp.add("t_embed.2.weight", (4 * chs[0], 4 * chs[0]), "linear", 4 * chs[0])
            sp.add("t_embed.2.bias", (4 * chs[0],), "bias", 4 * chs[0])
        for k, dv in enumerate(self.v_conditioning_dims):
            sp.add(f"v_embeds.{k}.0.weight", (V_EMB_DIM, dv), "linear", dv)
            sp.add(f"v_embeds.{k}.0.bias", (V_EMB_DIM,), "bias", dv)
            sp.add(f"v_embeds.{k}.2.weight", (V_EMB_DIM, V_EMB_DIM), "linear", V_EMB_DIM)
            sp.add(f"v_embeds.{k}.2.bias", (V_EMB_DIM,), "bias", V_EMB_DIM)
        self.head_end = sp.total
        # (4) the layers in FORWARD order: the backward pass then completes the gradient vector from its end towards its start, so
        # contiguous slices of it can be all-reduced while the rest of the backward is still running (bucket_bounds)
        cin0 = self.in_channels + self.s_conditioning_channels
        sp.add("conv_in.weight", (T, chs[0], cin0), "conv", T * cin0)
        sp.add("conv_in.bias", (chs[0],), "bias", T * cin0)
        blocks = {b.name: b for b in self.blocks}

        def add_block(b):
            cin = b.c1 + b.c2
            sp.add(f"{b.name}.norm1.weight", (cin,), "ones")
            sp.add(f"{b.name}.norm1.bias", (cin,), "zeros")
            sp.add(f"{b.name}.conv1.weight", (T, b.cout, cin), "conv", T * cin)
            sp.add(f"{b.name}.norm2.weight", (b.cout,), "ones")
            sp.add(f"{b.name}.norm2.bias", (b.cout,), "zeros")
            sp.add(f"{b.name}.conv2.weight", (T, b.cout, b.cout), "zeros")          # D4: zero-init
            sp.add(f"{b.name}.conv2.bias", (b.cout,), "zeros")
            if b.has_skip:                                                           # 1^dim conv, split per source
                sp.add(f"{b.name}.skip.weight", (1, b.cout, b.c1), "conv", cin)
                if b.c2:
                    sp.add(f"{b.name}.skip2.weight", (1, b.cout, b.c2), "conv", cin)
                sp.add(f"{b.name}.skip.bias", (b.cout,), "bias", cin)

        for i in range(L):
            add_block(blocks[f"downs.{i}.block"])
            if i != L - 1:
                sp.add(f"downs.{i}.down.weight", (T, chs[i], chs[i]), "conv", T * chs[i])
                sp.add(f"downs.{i}.down.bias", (chs[i],), "bias", T * chs[i])
        for j in range(2):
            add_block(blocks[f"mid.{j}"])
            if j == 0 and self.mid_attn:               # D13 [INFERRED]: GroupNorm -> qkv (1^dim conv) -> attention -> proj (zero-init)
                C = chs[-1]
                sp.add("mid_attn.norm.weight", (C,), "ones")
                sp.add("mid_attn.norm.bias", (C,), "zeros")
                sp.add("mid_attn.qkv.weight", (1, 3 * C, C), "conv", C)
                sp.add("mid_attn.qkv.bias", (3 * C,), "bias", C)
                sp.add("mid_attn.proj.weight", (1, C, C), "zeros")
                sp.add("mid_attn.proj.bias", (C,), "zeros")
        for i in reversed(range(L - 1)):
            sp.add(f"ups.{i}.up.weight", (T, chs[i], chs[i + 1]), "conv", T * chs[i + 1])
            sp.add(f"ups.{i}.up.bias", (chs[i],), "bias", T * chs[i + 1])
            add_block(blocks[f"ups.{i}.block"])
        sp.add("norm_out.weight", (chs[0],), "ones")
        sp.add("norm_out.bias", (chs[0],), "zeros")
        sp.add("conv_out.weight", (T, self.in_channels, chs[0]), "zeros")            # D6: zero-init
        sp.add("conv_out.bias", (self.in_channels,), "zeros")
        return sp

    @torch.no_grad()
    def reset_parameters(self, generator=None, zero_init_std=None):
        """torch-default-like init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); zero-init convs per D4/D6.
        zero_init_std: if set, the zero-init convs get N(0, std) instead (benchmarks: SURVEY section 8d)."""
        self.flat.zero_()
        for name, (off, shape, init, fan_in) in self.spec.items.items():
            v = self.flat[off:off + math.prod(shape)]
            if init in ("conv", "linear", "bias"):
                bound = 1.0 / math.sqrt(fan_in)
                v.copy_((torch.rand(v.shape, generator=generator) * 2 - 1) * bound)
            elif init == "ones":
                v.fill_(1.0)
            elif init == "zeros" and zero_init_std and name.endswith("weight") and "norm" not in name:
                v.copy_(torch.randn(v.shape, generator=generator) * zero_init_std)

    # ------------------------------------------------------------------ parameter views / state dict
    def view(self, name, flat=None):
        off, shape, _, _ = self.spec.items[name]
        f = self.flat if flat is None else flat
        return f[off:off + math.prod(shape)].view(shape)

    def bucket_bounds(self):
        """Slices of the flat (gradient) vector in the order the backward pass completes them, with the layer after whose backward
        each is final: [(lo, hi, ready_after)], ready_after in {"ups", "mid", "downs.top", "end"}.  Data-parallel training
        all-reduces them one by one on a communication stream while the backward continues (unet_hip.HipUNet.backward)."""
        L = len(self.chs)
        off = lambda name: self.spec.items[name][0]
        total = self.spec.total
        if L < 2:
            return [(0, total, "end")]
        cuts = [(off(f"ups.{L - 2}.up.weight"), total, "ups"), (off("mid.0.norm1.weight"), off(f"ups.{L - 2}.up.weight"), "mid"),
                (off(f"downs.{L - 1}.block.norm1.weight"), off("mid.0.norm1.weight"), "downs.top"),
                (0, off(f"downs.{L - 1}.block.norm1.weight"), "end")]
        return [c for c in cuts if c[1] > c[0]]

    def enable_ddp(self, world, group=None):
        """Data-parallel training on the HIP backend: the backward pass all-reduces the flat gradient in buckets, overlapped with
        the remaining backward kernels (unet_hip.GradBuckets).  No-op on the torch backend (the trainer reduces p.grad itself)."""
        if self.backend != "hip":
            return
        if self._exec is None:
            from .unet_hip import HipUNet
            self._exec = HipUNet(self)
        self._exec.enable_ddp(world, group)

    def mark_weights_dirty(self):
        """Tell the HIP executor that the parameters changed through an op that does not bump Tensor._version (fused optimizers):
        the MFMA-packed weight copies are rebuilt at the next forward."""
        self.weights_epoch += 1

    def repack_weights(self):
        """Rebuild the MFMA-packed weight copies NOW (same dtype / forms as the last packing) instead of at the next forward: called
        from the optimizer post-step hook, the one launch then runs in the GPU-idle gap while the host prepares the next step."""
        ex = self._exec
        if self.backend == "hip" and ex is not None and ex._packed_key is not None and self.flat.is_cuda:
            ex.pack_weights(self.flat.detach(), ex._packed_key[3], ex._packed_key[4], overlap=True)

    def cond_matrix(self, k, flat=None):
        """[sum cout, dim_k] projection matrix of conditioning k for all blocks at once.  `flat` may be the flat vector or its head
        slice flat[:head_end]."""
        off = self.spec.items[f"{self.blocks[0].name}.cond.{k}.weight"][0]
        f = self.flat if flat is None else flat
        return f[off:off + self.table_width * self.cond_dims[k]].view(self.table_width, self.cond_dims[k])

    def conv1_bias_all(self, flat=None):
        off = self.spec.items[f"{self.blocks[0].name}.conv1.bias"][0]
        f = self.flat if flat is None else flat
        return f[off:off + self.table_width]

    def named_views(self):
        return {name: self.view(name) for name in self.spec.items}

    def state_dict(self, destination=None, prefix="", keep_vars=False):
        out = {} if destination is None else destination
        for name in self.spec.items:
            v = self.view(name)
            out[prefix + name] = v if keep_vars else v.detach().clone()
        return out

    def _convert_entry(self, name, value):
        """A checkpoint tensor -> this layout.  Exact shape, or for conv weights the PyTorch / Lightning layout
        [cout, cin, *k] (permuted to tap-major [taps, cout, cin]).  Anything else raises: equal element counts are not enough
        (a reshape would silently scramble a [cout, cin, 3, 3, 3] weight)."""
        want = self.view(name).shape
        v = value.detach().to(torch.float32)
        if tuple(v.shape) == tuple(want):
            return v
        if len(want) == 3 and v.dim() == 2 + self.dim and v.shape[0] == want[1] and v.shape[1] == want[2] \
                and math.prod(v.shape[2:]) == want[0]:
            return v.permute(*range(2, 2 + self.dim), 0, 1).reshape(want)
        raise RuntimeError(f"CUNet.load_state_dict: {name} has shape {tuple(v.shape)}, expected {tuple(want)}"
                           + (f" or (cout, cin, {'k, ' * self.dim}) = ({want[1]}, {want[2]}, ...)" if len(want) == 3 else ""))

    def load_state_dict(self, state_dict, strict=True):
        sd = dict(state_dict)
        # a concatenated 1^dim skip weight [cout, c1+c2, 1..] (the reference concatenates the skip tensor) -> skip / skip2
        for b in self.blocks:
            k1, k2 = f"{b.name}.skip.weight", f"{b.name}.skip2.weight"
            if b.has_skip and b.c2 and k1 in sd and k2 not in sd and sd[k1].dim() == 2 + self.dim and sd[k1].shape[1] == b.c1 + b.c2:
                w = sd[k1]
                sd[k1], sd[k2] = w[:, :b.c1].contiguous(), w[:, b.c1:].contiguous()
        missing = [k for k in self.spec.items if k not in sd]
        unexpected = [k for k in sd if k not in self.spec.items]
        if strict and (missing or unexpected):
            raise RuntimeError(f"CUNet.load_state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        with torch.no_grad():
            for name in self.spec.items:
                if name in sd:
                    self.view(name).copy_(self._convert_entry(name, sd[name]))
        self.mark_weights_dirty()
        return missing, unexpected

    # ------------------------------------------------------------------ conditioning (tiny; PyTorch)
    @staticmethod
    def sinusoidal_embedding(t, dim=T_EMB_DIM):
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = 1000.0 * t.to(torch.float32)[:, None] * freqs[None, :]
        return torch.cat([torch.sin(args), torch.cos(args)], dim=1)

    def _mlp2(self, prefix, x, head=None):
        x = F.gelu(F.linear(x, self.view(prefix + ".0.weight", head), self.view(prefix + ".0.bias", head)))
        return F.gelu(F.linear(x, self.view(prefix + ".2.weight", head), self.view(prefix + ".2.bias", head)))

    def cond_vectors(self, t, v_conditionings, head=None):
        conds = []
        if self.t_conditioning:
            assert t is not None, "t_conditioning=True needs t"
            conds.append(self._mlp2("t_embed", self.sinusoidal_embedding(t.reshape(-1)), head))
        vs = list(v_conditionings or [])
        assert len(vs) == len(self.v_conditioning_dims), "len(v_conditionings) != len(v_conditioning_dims)"
        for k, v in enumerate(vs):
            conds.append(self._mlp2(f"v_embeds.{k}", v.to(torch.float32), head))
        return conds

    def cond_table(self, conds, batch, head=None):
        """[B, sum cout]: sum_k cond_k @ W_k^T for every block (D4, additive injection)."""
        table = None
        for k, c in enumerate(conds):
            part = F.linear(c, self.cond_matrix(k, head))
            table = part if table is None else table + part
        if table is None:
            table = torch.zeros(batch, self.table_width, device=self.flat.device)
        return table

    def cond_specs(self, t, v_conditionings, flat=None, which="all"):
        """Descriptors of the conditioning MLPs for the K6 kernel (hip_ops.CondTable): views of `flat` (or of a gradient vector of
        the same layout).  which: "all" | "t" | "v".  t: [rows] fp32 or None; v_conditionings: list of [rows, d] fp32."""
        specs, k = [], 0
        if self.t_conditioning:
            if which in ("all", "t"):
                specs.append(dict(input=t, sinusoid=True, in_dim=T_EMB_DIM, w1=self.view("t_embed.0.weight", flat),
                                  b1=self.view("t_embed.0.bias", flat), w2=self.view("t_embed.2.weight", flat),
                                  b2=self.view("t_embed.2.bias", flat), wproj=self.cond_matrix(0, flat)))
            k = 1
        if which in ("all", "v"):
            for j, dv in enumerate(self.v_conditioning_dims):
                v = None if v_conditionings is None else v_conditionings[j]
                specs.append(dict(input=v, sinusoid=False, in_dim=dv, w1=self.view(f"v_embeds.{j}.0.weight", flat),
                                  b1=self.view(f"v_embeds.{j}.0.bias", flat), w2=self.view(f"v_embeds.{j}.2.weight", flat),
                                  b2=self.view(f"v_embeds.{j}.2.bias", flat), wproj=self.cond_matrix(k + j, flat)))
        return specs

    # ------------------------------------------------------------------ forward
    def forward(self, x, t=None, s_conditioning=None, v_conditionings=None, _packed_input=None, **ignored):
        """_packed_input (HIP backend, internal): conv_in's NDHWC input [B, D, H, W, cpad(2)] = {x, s_conditioning, 0...} already packed in
        the compute dtype by the caller (VDM.get_loss' fused head: vdm_diffuse_pack wrote it while it formed x = z_t)."""
        B = x.shape[0]
        if t is not None:
            t = torch.as_tensor(t, dtype=torch.float32, device=x.device).reshape(-1)
            if t.numel() == 1 and B > 1:
                t = t.expand(B)
        if self.backend == "torch":
            head = self.flat[:self.head_end]
            conds = self.cond_vectors(t, v_conditionings, head)
            table = self.cond_table(conds, B, head)
            return self._forward_torch(x, s_conditioning, table)
        if self.dim != 3 or self.in_channels != 1 or self.s_conditioning_channels > 1:
            raise NotImplementedError("the HIP backend covers the 3D, single-field configurations of the reference "
                                      "(shape=(1,D,D,D), s_conditioning_channels<=1); use backend='torch' for 2D plumbing")
        if not x.is_cuda:
            raise RuntimeError("CUNet(backend='hip') needs tensors on a GPU; there is no CPU fallback "
                               "(construct with backend='torch' for the CPU plumbing config)")
        from .unet_hip import hip_unet_apply
        if self.t_conditioning:
            assert t is not None, "t_conditioning=True needs t"
        vs = list(v_conditionings or [])
        assert len(vs) == len(self.v_conditioning_dims), "len(v_conditionings) != len(v_conditioning_dims)"
        return hip_unet_apply(self, x, s_conditioning, t=t, v_conditionings=vs, packed=_packed_input)

    # ------------------------------------------------------------------ explicit torch backend (C1 plumbing)
    def _conv_t(self, x, wname, bname, stride=1):
        w = self.view(wname)                                   # [taps, cout, cin]
        k = 3 if w.shape[0] > 1 else 1
        w = w.view((k,) * self.dim + w.shape[1:]).permute(self.dim, self.dim + 1, *range(self.dim))
        b = self.view(bname) if bname else None
        f = F.conv3d if self.dim == 3 else F.conv2d
        pad = k // 2
        if pad and self.conv_padding_mode == "circular":
            return f(F.pad(x, (pad, pad) * self.dim, mode="circular"), w, b, stride=stride)
        return f(x, w, b, stride=stride, padding=pad)

    def _block_t(self, b, x1, x2, table):
        x = x1 if x2 is None else torch.cat([x1, x2], dim=1)
        n = b.name
        h = F.silu(F.group_norm(x, self.norm_groups, self.view(n + ".norm1.weight"), self.view(n + ".norm1.bias"), GN_EPS))
        h = self._conv_t(h, n + ".conv1.weight", n + ".conv1.bias")
        nb = table[:, b.table_off:b.table_off + b.cout]
        h = h + nb.reshape(nb.shape + (1,) * self.dim)
        h = F.silu(F.group_norm(h, self.norm_groups, self.view(n + ".norm2.weight"), self.view(n + ".norm2.bias"), GN_EPS))
        h = F.dropout(h, self.dropout_prob, self.training)
        h = self._conv_t(h, n + ".conv2.weight", n + ".conv2.bias")
        if b.has_skip:
            s = self._conv_t(x1, n + ".skip.weight", n + ".skip.bias")
            if x2 is not None:
                s = s + self._conv_t(x2, n + ".skip2.weight", None)
            return s + h
        return x + h

    def _attn_t(self, x):
        """D13 on the torch backend: GroupNorm -> qkv -> softmax(q k^T / sqrt(hd)) v over all voxels -> proj -> + x."""
        N, C, H = x.shape[0], x.shape[1], self.n_attention_heads
        h = F.group_norm(x, self.norm_groups, self.view("mid_attn.norm.weight"), self.view("mid_attn.norm.bias"), GN_EPS)
        qkv = self._conv_t(h, "mid_attn.qkv.weight", "mid_attn.qkv.bias").reshape(N, 3, H, C // H, -1)
        q, k, v = (qkv[:, i].transpose(-1, -2) for i in range(3))                # [N, H, V, hd]
        a = F.scaled_dot_product_attention(q, k, v)                              # scale 1 / sqrt(hd)
        a = a.transpose(-1, -2).reshape(x.shape)
        return x + self._conv_t(a, "mid_attn.proj.weight", "mid_attn.proj.bias")

    def _forward_torch(self, x, s_conditioning, table):
        L = len(self.chs)
        blocks = {b.name: b for b in self.blocks}
        h = x if s_conditioning is None else torch.cat([x, s_conditioning], dim=1)
        h = self._conv_t(h, "conv_in.weight", "conv_in.bias")
        skips = []
        for i in range(L):
            h = self._block_t(blocks[f"downs.{i}.block"], h, None, table)
            if i != L - 1:
                skips.append(h)
                h = self._conv_t(h, f"downs.{i}.down.weight", f"downs.{i}.down.bias", stride=2)
        for j in range(2):
            h = self._block_t(blocks[f"mid.{j}"], h, None, table)
            if j == 0 and self.mid_attn:
                h = self._attn_t(h)
        for i in reversed(range(L - 1)):
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = self._conv_t(h, f"ups.{i}.up.weight", f"ups.{i}.up.bias")
            h = self._block_t(blocks[f"ups.{i}.block"], h, skips[i], table)
        h = F.silu(F.group_norm(h, self.norm_groups, self.view("norm_out.weight"), self.view("norm_out.bias"), GN_EPS))
        return self._conv_t(h, "conv_out.weight", "conv_out.bias")
