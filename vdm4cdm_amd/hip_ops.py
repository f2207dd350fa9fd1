"""Typed wrappers over the C-ABI: torch tensors in, kernels enqueued on torch's current HIP stream.

torch is used here only for device memory and streams (``Tensor.data_ptr()``,
``torch.cuda.current_stream()``); no torch compute op runs on activations in this module.
Activations are NDHWC tensors ``[N, D, H, W, C]`` (fp32 or bf16).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import CondMlp, ConvDesc, GnFold, PACK_DGRAD, PACK_FWD, VDM_BF16, VDM_F32, check

GN_EPS = 1e-5
# timing ablations (tools/ablate_step.sh): kernels named here are NOT launched (wrong results - only bench.py timing runs set this)
import os as _os
ABLATE = set(filter(None, _os.environ.get("VDM4CDM_ABLATE", "").split(",")))
if ABLATE or "VDM4CDM_ABLATE_REDUCE" in _os.environ:
    import sys as _sys
    print(f"\n*** vdm4cdm_amd.hip_ops: VDM4CDM_ABLATE={sorted(ABLATE)} / VDM4CDM_ABLATE_REDUCE set - kernels are SKIPPED and results are WRONG "
          "(timing ablations of tools/ablate_step.sh only; Trainer.fit refuses to run) ***\n", file=_sys.stderr, flush=True)


class KernelProfiler:
    """Optional per-launch HIP-event timing (bench.py): events are recorded on the launch stream around each
    C-ABI call; `summary()` aggregates per kernel key after a synchronize."""

    def __init__(self, tags=None):
        self.records = []          # (key, start, end, flops, bytes)
        self.tags = tags           # None: every launch; else only launches whose tag is in this set (event records cost time too)
        self.main_stream = None    # launches on any other stream are marked "side" (weight gradients, skip convs: off the critical path)

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, start, key, flops=0.0, nbytes=0.0, exec_flops=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if self.main_stream is None:
            self.main_stream = torch.cuda.default_stream().cuda_stream
        side = torch.cuda.current_stream().cuda_stream != self.main_stream
        self.records.append((key, start, e, flops, nbytes, flops if exec_flops is None else exec_flops, side))

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, s, e, fl, nb, xf, side in self.records:
            a = agg.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "exec_flops": 0.0, "side_launches": 0})
            a["launches"] += 1
            a["side_launches"] += 1 if side else 0
            a["ms"] += s.elapsed_time(e)
            a["flops"] += fl
            a["bytes"] += nb
            a["exec_flops"] += xf
        return agg


PROFILER = None        # set to a KernelProfiler by bench.py
# Device step counter (int32 tensor) mixed into every RNG seed inside the kernels (dropout masks, noise fields, time grid); set by
# vdm_model.GraphedTrainStep: a captured graph bakes the host seeds into its kernel arguments, the counter keeps the draws fresh.
SEED_STEP = None


def _pb(tag="other"):
    if PROFILER is None or (PROFILER.tags is not None and tag not in PROFILER.tags):
        return None
    return PROFILER.begin()


def _pe(ev, key, flops=0.0, nbytes=0.0, exec_flops=None):
    """flops: algorithmic (27 taps on the conv's own grid); exec_flops: what the kernel issues to the matrix cores (the
    per-parity-class kernels run 8 merged taps / 1-8 taps per class instead of 27)."""
    if ev is not None:
        PROFILER.end(ev, key, flops, nbytes, exec_flops)


def _nc_for(cout, dtype=None):
    return 1 if cout <= 16 else (2 if (cout <= 32 or dtype == torch.float32) else 4)


def _tname(dtype):
    return "float" if dtype == torch.float32 else "bf16_t"


def dt_id(dtype):
    if dtype == torch.float32:
        return VDM_F32
    if dtype == torch.bfloat16:
        return VDM_BF16
    raise ValueError(f"unsupported activation dtype {dtype}")


def epl(dtype):
    """elements per 16-byte piece"""
    return 4 if dtype == torch.float32 else 8


def cpad(c, dtype):
    e = epl(dtype)
    return (c + e - 1) // e * e


def _s():
    return torch.cuda.current_stream().cuda_stream


def _scratch(cache, key, numel, dtype, device, floor=0):
    """Per-(device, stream) scratch tensors are cached in module-level dicts - EXCEPT while the current stream is being captured into a
    hipGraph: an allocation made during capture comes out of that graph's private memory pool, the pool keeps the block for the replays
    whether or not the tensor object survives, and a module-level reference would pin memory of a pool whose graph is long gone (and
    hand it to later captures / eager launches on a stream that happens to reuse the handle).  Under capture: a fresh tensor, not cached."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(numel, floor), dtype=dtype, device=device)
    ws = cache.get(key)
    if ws is None or ws.numel() < numel:
        ws = cache[key] = torch.empty(max(numel, floor), dtype=dtype, device=device)
    return ws


# The side stream a launch is being issued on (set by unet_hip.SideStream.run around its body), else None.  Every tensor whose pointer
# goes to a kernel while it is set is recorded on that stream HERE - one choke point instead of a record_stream per call site: memory
# that was allocated on the main stream and is still referenced by a queued side-stream kernel (weight gradients run up to a whole
# backward pass behind the main stream) cannot be handed to a new owner by the caching allocator before that kernel has run.
SIDE_STREAM = None


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "HIP ops need device tensors"
    if SIDE_STREAM is not None:
        t.record_stream(SIDE_STREAM)
    return t.data_ptr()


def _contig(*ts):
    for t in ts:
        if t is not None:
            assert t.is_contiguous(), "HIP ops need contiguous tensors"


class Conv:
    """One convolution layer: descriptor factory + packed-weight buffers (fwd / dgrad order)."""

    _ws = {}          # (device, stream) -> wgrad workspace (uint8 tensor): launches on one stream are ordered, streams never share

    def __init__(self, cin, cout, ksize=3, stride=1, upsample=0, circular=False, out_f32=False):
        self.cin, self.cout, self.ksize, self.stride, self.upsample = cin, cout, ksize, stride, upsample
        self.circular, self.out_f32 = circular, out_f32
        self.wf = self.wd = None
        self._descs = {}

    def desc(self, n, od, oh, ow, dtype, stride=None):
        st = self.stride if stride is None else stride
        key = (n, od, oh, ow, dtype, st)
        d = self._descs.get(key)
        if d is None:
            d = ConvDesc(n=n, od=od, oh=oh, ow=ow, cin=self.cin, cout=self.cout, ksize=self.ksize, stride=st,
                         upsample=self.upsample, pad_mode=1 if self.circular else 0, dtype=dt_id(dtype),
                         out_f32=1 if (self.out_f32 and st == self.stride) else 0)
            self._descs[key] = d
        return d

    def pack(self, w_master, dtype, need_dgrad):
        """w_master: fp32 [taps, cout, cin] (a view into the flat parameter vector)."""
        L = _lib.lib()
        d = self.desc(1, 2, 2, 2, dtype)
        assert w_master.dtype == torch.float32 and w_master.is_contiguous()
        assert w_master.numel() == self.ksize ** 3 * self.cout * self.cin
        if self.wf is None or self.wf.device != w_master.device or self._packed_dtype != dtype:
            self.wf = torch.empty(L.vdm_conv_packed_bytes(d, PACK_FWD), dtype=torch.uint8, device=w_master.device)
            self.wd = None
            self._packed_dtype = dtype
        check(L.vdm_conv_pack_weights(d, PACK_FWD, _p(w_master), _p(self.wf), _s()), "vdm_conv_pack_weights(fwd)")
        if need_dgrad:
            if self.wd is None:
                self.wd = torch.empty(L.vdm_conv_packed_bytes(d, PACK_DGRAD), dtype=torch.uint8, device=w_master.device)
            check(L.vdm_conv_pack_weights(d, PACK_DGRAD, _p(w_master), _p(self.wd), _s()), "vdm_conv_pack_weights(dgrad)")

    def alloc_packed(self, device, dtype, need_dgrad):
        """Make sure the packed-weight buffers exist (pack_many fills them)."""
        L = _lib.lib()
        d = self.desc(1, 2, 2, 2, dtype)
        if self.wf is None or self.wf.device != device or self._packed_dtype != dtype:
            self.wf = torch.empty(L.vdm_conv_packed_bytes(d, PACK_FWD), dtype=torch.uint8, device=device)
            self.wd = None
            self._packed_dtype = dtype
        if need_dgrad and self.wd is None:
            self.wd = torch.empty(L.vdm_conv_packed_bytes(d, PACK_DGRAD), dtype=torch.uint8, device=device)

    def out_shape(self, x):
        n, d, h, w, _ = x.shape
        if self.stride == 2:
            return (n, d // 2, h // 2, w // 2, self.cout)
        if self.upsample:
            return (n, 2 * d, 2 * h, 2 * w, self.cout)
        return (n, d, h, w, self.cout)

    def gn_in_ok(self, x):
        """Can fwd(..., gn_in=...) apply GroupNorm + SiLU to the staged input (bf16, the generic 3x3x3 stride-1 kernel)?"""
        shp = self.out_shape(x)
        return x.dtype == torch.bfloat16 and bool(_lib.lib().vdm_conv_fwd_gn_supported(self.desc(shp[0], shp[1], shp[2], shp[3], x.dtype)))

    def fwd(self, x, bias=None, nbias=None, residual=None, out=None, gn=False, gn_in=None):
        """out = conv(x) + bias + nbias[n] + residual.  nbias: fp32 [N, >=cout] view (row stride honoured).
        gn_in = (groups, stats, gamma, beta) (inference, gn_in_ok): the conv input is silu(groupnorm(x)) and the kernel applies it to the
        staged image itself - no gn_silu_fwd pass, bit-identical result."""
        L = _lib.lib()
        _contig(x, bias, residual)
        assert x.shape[-1] == cpad(self.cin, x.dtype), f"conv input must have {cpad(self.cin, x.dtype)} channels, got {x.shape[-1]}"
        shp = self.out_shape(x)
        if out is None:
            out = torch.empty(shp, dtype=torch.float32 if self.out_f32 else x.dtype, device=x.device)
        assert tuple(out.shape) == shp and out.is_contiguous()
        nstride = 0
        if nbias is not None:
            assert nbias.dtype == torch.float32 and nbias.stride(1) == 1 and nbias.shape[0] == x.shape[0]
            nstride = nbias.stride(0)
        if residual is not None:
            assert residual.shape == out.shape and residual.dtype == x.dtype
        d = self.desc(shp[0], shp[1], shp[2], shp[3], x.dtype)
        part = None
        if gn:                    # GroupNorm statistics of `out`, reduced per tile by the epilogue; consumed by gn_stats(out, ...)
            tiles = L.vdm_conv_gn_tiles(d)
            if tiles > 0:
                part = torch.empty((shp[0], tiles, self.cout, 2), dtype=torch.float32, device=x.device)
        if "conv1" in ABLATE and self.ksize == 1:
            out.gn_partials = None
            return out
        ev = _pb("conv3" if self.ksize == 3 else "other")
        if gn_in is not None:
            G_, st_, gam_, bet_ = gn_in
            _contig(st_, gam_, bet_)
            check(L.vdm_conv_fwd_gn(d, _p(x), _p(self.wf), _p(bias), _p(nbias), nstride, _p(residual), _p(out), _p(part), _p(st_), _p(gam_),
                                    _p(bet_), int(G_), GN_EPS, _s()), "vdm_conv_fwd_gn")
        else:
            check(L.vdm_conv_fwd(d, _p(x), _p(self.wf), _p(bias), _p(nbias), nstride, _p(residual), _p(out), _p(part), _s()), "vdm_conv_fwd")
        out.gn_partials = part
        if ev is not None:
            nvox = shp[0] * shp[1] * shp[2] * shp[3]
            es = x.element_size()
            if self.ksize == 3 and self.upsample:
                key = f"conv_cls_kernel<{_tname(x.dtype)},NC{_nc_for(self.cout, x.dtype)},F>"
            elif L.vdm_conv_kernel_variant(d, 0) == 2:
                key = "conv_kpack_kernel"
            elif L.vdm_conv_kernel_variant(d, 0) == 3:
                key = f"conv_fwd_kernel<{_tname(x.dtype)},k3,s1,NC2,split>"
            elif L.vdm_conv_kernel_variant(d, 0) == 4:
                key = f"conv_ksplit_kernel<{_tname(x.dtype)},NC4>"
            else:
                key = f"conv_fwd_kernel<{_tname(x.dtype)},k{self.ksize},s{self.stride},NC{_nc_for(self.cout, x.dtype)}>"
            alg = 2.0 * nvox * self.ksize ** 3 * self.cin * self.cout
            _pe(ev, key, alg,
                x.numel() * es + out.numel() * out.element_size() + (residual.numel() * es if residual is not None else 0),
                alg * (8.0 / 27.0) if (self.ksize == 3 and self.upsample) else None)          # 8 merged taps per parity class
        return out

    def dgrad(self, dout, residual=None, out=None):
        """Gradient w.r.t. the conv input (stride 2: twice the dout dims; up-sampling conv: half)."""
        L = _lib.lib()
        _contig(dout, residual)
        n, od, oh, ow, c = dout.shape
        assert c == cpad(self.cout, dout.dtype)
        if self.stride == 2:
            ishape = (n, 2 * od, 2 * oh, 2 * ow, self.cin)
        elif self.upsample:
            ishape = (n, od // 2, oh // 2, ow // 2, self.cin)
        else:
            ishape = (n, od, oh, ow, self.cin)
        if out is None:
            out = torch.empty(ishape, dtype=dout.dtype, device=dout.device)
        assert tuple(out.shape) == ishape and (residual is None or tuple(residual.shape) == ishape)
        d = self.desc(n, od, oh, ow, dout.dtype)
        if "conv1" in ABLATE and self.ksize == 1:
            return out
        ev = _pb("conv3" if self.ksize == 3 else "other")
        check(L.vdm_conv_dgrad(d, _p(dout), _p(self.wd), _p(residual), _p(out), _s()), "vdm_conv_dgrad")
        if ev is not None:
            nvox = n * od * oh * ow
            es = dout.element_size()
            if self.ksize == 3 and (self.stride == 2 or self.upsample):
                key = f"conv_cls_kernel<{_tname(dout.dtype)},NC{_nc_for(self.cin, dout.dtype)},{'B' if self.upsample else 'F'}>"
            elif L.vdm_conv_kernel_variant(d, 1) == 2:
                key = "conv_kpack_kernel"
            elif L.vdm_conv_kernel_variant(d, 1) == 3:
                key = f"conv_fwd_kernel<{_tname(dout.dtype)},k3,s1,NC2,split>"
            elif L.vdm_conv_kernel_variant(d, 1) == 4:
                key = f"conv_ksplit_kernel<{_tname(dout.dtype)},NC4>"
            else:
                key = f"conv_fwd_kernel<{_tname(dout.dtype)},k{self.ksize},s1,NC{_nc_for(self.cin, dout.dtype)}>"
            alg = 2.0 * nvox * self.ksize ** 3 * self.cin * self.cout
            xf = None
            if self.ksize == 3 and self.upsample:
                xf = alg * (8.0 / 27.0)                       # 8 classes x 8 merged taps on the coarse grid (= 64/216 of 27 fine taps)
            elif self.ksize == 3 and self.stride == 2:
                xf = alg                                      # 27 taps in total over the 8 classes of the fine grid: nothing skipped
            _pe(ev, key, alg, dout.numel() * es + out.numel() * es + (residual.numel() * es if residual is not None else 0), xf)
        return out

    def gn_fold_ok(self, c1, c2, dtype):
        """Can dgrad_gn be used for a GroupNorm over concat(c1, c2) = this conv's input channels?  (a lane's 4*NC consecutive
        channels must not straddle the concat boundary; 3x3x3 stride-1 convs only)"""
        span = 4 * _nc_for(self.cin, dtype)
        return (self.ksize == 3 and self.stride == 1 and not self.upsample and c1 + c2 == self.cin and self.cin % span == 0
                and (c2 == 0 or c1 % span == 0) and self.cin % epl(dtype) == 0)

    def dgrad_gn(self, dout, x1, x2, groups, stats, gamma, beta, keep_mask=None, dropout_p=0.0, out=None):
        """Input gradient of a conv whose input was drop(silu(gn(concat(x1, x2)))), with the GroupNorm backward reduction folded
        into the epilogue.  Returns dyh = dL/dy * keep/(1-p) * silu'(yhat) with `.gnb_partials` ([N, tiles, cin, 2]: per-tile
        sums of dyh and dyh * x) for gn_bwd_finalize."""
        L = _lib.lib()
        _contig(dout, x1, x2, stats, gamma, beta, keep_mask)
        n, od, oh, ow, c = dout.shape
        assert c == cpad(self.cout, dout.dtype)
        c1, c2 = x1.shape[-1], (0 if x2 is None else x2.shape[-1])
        assert self.gn_fold_ok(c1, c2, dout.dtype), "dgrad_gn: unsupported channel split"
        assert tuple(x1.shape[:-1]) == (n, od, oh, ow) and x1.dtype == dout.dtype
        if out is None:
            out = torch.empty((n, od, oh, ow, self.cin), dtype=dout.dtype, device=dout.device)
        d = self.desc(n, od, oh, ow, dout.dtype)
        tiles = L.vdm_conv_dgrad_gn_tiles(d)
        part = torch.empty((n, tiles, self.cin, 2), dtype=torch.float32, device=dout.device)
        f = GnFold(x1=_p(x1), x2=_p(x2), c1=c1, c2=c2, groups=groups, stats=_p(stats), gamma=_p(gamma), beta=_p(beta), eps=GN_EPS,
                   inv_keep=1.0 / (1.0 - dropout_p) if keep_mask is not None else 1.0, keep_mask=_p(keep_mask), partials=_p(part))
        ev = _pb("conv3")
        check(L.vdm_conv_dgrad_gn(d, _p(dout), _p(self.wd), _p(out), C.byref(f), _s()), "vdm_conv_dgrad_gn")
        out.gnb_partials = part
        if ev is not None:
            es = dout.element_size()
            var = L.vdm_conv_kernel_variant(d, 1)
            key = "conv_kpack_kernel" if var == 2 else (f"conv_fwd_kernel<{_tname(dout.dtype)},k3,s1,NC2,split>" if var == 3 else
                                                        (f"conv_ksplit_kernel<{_tname(dout.dtype)},NC4>" if var == 4 else
                                                         f"conv_fwd_kernel<{_tname(dout.dtype)},k3,s1,NC{_nc_for(self.cin, dout.dtype)}>"))
            _pe(ev, key + "+gnb", 2.0 * n * od * oh * ow * 27 * self.cin * self.cout, dout.numel() * es + 2 * out.numel() * es)
        return out

    def dgw_ok(self, dout, c1, c2):
        """Can dgrad_gn_wgrad run this conv's two gradients as one launch (csrc/conv_dgw.hip: bf16, 32 -> 32 channels, large grids)?"""
        n, od, oh, ow, _ = dout.shape
        return (dout.dtype == torch.bfloat16 and self.gn_fold_ok(c1, c2, dout.dtype)
                and bool(_lib.lib().vdm_conv_dgw_supported(self.desc(n, od, oh, ow, dout.dtype))))

    def dgrad_gn_wgrad(self, dout, act, x1, x2, groups, stats, gamma, beta, dw, dbias=None, keep_mask=None, dropout_p=0.0):
        """dgrad_gn(dout, ...) and wgrad(act, dout, dw, dbias) in ONE launch that stages dout once.  Returns dyh (with `.gnb_partials`)."""
        L = _lib.lib()
        _contig(dout, act, x1, x2, stats, gamma, beta, keep_mask, dw, dbias)
        n, od, oh, ow, c = dout.shape
        c1, c2 = x1.shape[-1], (0 if x2 is None else x2.shape[-1])
        assert c == self.cout and act.shape == (n, od, oh, ow, self.cin) and act.dtype == dout.dtype
        d = self.desc(n, od, oh, ow, dout.dtype)
        out = torch.empty((n, od, oh, ow, self.cin), dtype=dout.dtype, device=dout.device)
        part = torch.empty((n, L.vdm_conv_dgw_tiles(d), self.cin, 2), dtype=torch.float32, device=dout.device)
        need = L.vdm_conv_dgw_workspace_bytes(d)
        ws = _scratch(Conv._ws, (dout.device, _s(), "dgw"), need, torch.uint8, dout.device)
        f = GnFold(x1=_p(x1), x2=_p(x2), c1=c1, c2=c2, groups=groups, stats=_p(stats), gamma=_p(gamma), beta=_p(beta), eps=GN_EPS,
                   inv_keep=1.0 / (1.0 - dropout_p) if keep_mask is not None else 1.0, keep_mask=_p(keep_mask), partials=_p(part))
        ev = _pb("conv3")
        check(L.vdm_conv_dgrad_gn_wgrad(d, _p(dout), _p(self.wd), _p(act), _p(out), C.byref(f), _p(dw), _p(dbias), 0, _p(ws), ws.numel(), _s()),
              "vdm_conv_dgrad_gn_wgrad")
        out.gnb_partials = part
        if ev is not None:
            es = dout.element_size()
            _pe(ev, f"conv_dgw_kernel<{_tname(dout.dtype)},k3,s1>(dgrad+gnb+wgrad)", 4.0 * n * od * oh * ow * 27 * self.cin * self.cout,
                (2 * dout.numel() + 3 * out.numel()) * es)
        return out

    def wgrad(self, x, dout, dw, dbias=None, accumulate=False):
        """dw (fp32 view [taps, cout, cin]) = sum_v dout[v] (x) x[v + tap];  dbias (optional, ksize 3): sum_v dout[v]."""
        L = _lib.lib()
        _contig(x, dout, dw)
        n, od, oh, ow, c = dout.shape
        assert c == cpad(self.cout, dout.dtype) and x.shape[-1] == cpad(self.cin, x.dtype)
        d = self.desc(n, od, oh, ow, x.dtype)
        need = L.vdm_conv_wgrad_workspace_bytes(d)
        ws = _scratch(Conv._ws, (x.device, _s()), need, torch.uint8, x.device, floor=32 << 20)
        if "wgrad" in ABLATE or ("wgrad1" in ABLATE and self.ksize == 1):
            return dw
        ev = _pb("wgrad" if self.ksize == 3 else "other")
        check(L.vdm_conv_wgrad(d, _p(x), _p(dout), _p(dw), _p(dbias), 1 if accumulate else 0, _p(ws), ws.numel(), _s()), "vdm_conv_wgrad")
        if ev is not None:
            es = x.element_size()
            alg = 2.0 * n * od * oh * ow * self.ksize ** 3 * self.cin * self.cout
            _pe(ev, f"conv_wgrad_kernel<{_tname(x.dtype)},k{self.ksize},s{self.stride},u{self.upsample}>+reduce",
                alg, x.numel() * es + dout.numel() * es, alg * (8.0 / 27.0) if (self.ksize == 3 and self.upsample) else None)
        return dw


def _nv(x):
    n = x.shape[0]
    return n, x.numel() // (n * x.shape[-1])


_gn_ws = {}


def gn_stats(x1, x2, groups, out=None, chsum=False):
    """stats[n][g] = (sum, sum of squares) of concat(x1, x2).  chsum=True (sources produced by Conv.fwd(gn=True) only): also the
    per-channel sums, attached as `out.chsum` ([N, C] fp32)."""
    L = _lib.lib()
    _contig(x1, x2)
    n, v = _nv(x1)
    if out is None:
        out = torch.empty((n, groups, 2), dtype=torch.float32, device=x1.device)
    ws = _scratch(_gn_ws, (x1.device, _s()), _lib.GN_STATS_WS_BYTES // 4, torch.float32, x1.device)
    c2 = 0 if x2 is None else x2.shape[-1]
    p1 = getattr(x1, "gn_partials", None)       # set by Conv.fwd(..., gn=True): that source needs no pass over the tensor
    p2 = getattr(x2, "gn_partials", None) if x2 is not None else None
    cs = None
    if chsum and p1 is not None and (x2 is None or p2 is not None):
        cs = torch.empty((n, x1.shape[-1] + c2), dtype=torch.float32, device=x1.device)
    ev = _pb()
    check(L.vdm_gn_stats(_p(x1), x1.shape[-1], _p(x2), c2, n, v, groups, dt_id(x1.dtype), _p(out), _p(ws),
                         _p(p1), 0 if p1 is None else p1.shape[1], _p(p2), 0 if p2 is None else p2.shape[1], _p(cs), _s()), "vdm_gn_stats")
    out.chsum = cs
    _pe(ev, "gn_stats", 0.0, (x1.numel() * x1.element_size() if p1 is None else 0)
        + (x2.numel() * x2.element_size() if (x2 is not None and p2 is None) else 0))
    return out


def gn_silu_fwd(x1, x2, groups, stats, gamma, beta, dropout_p=0.0, seed=0, out=None, want_mask=False, linear=False):
    """want_mask (with dropout_p > 0): the dropout keep bits are also written, one byte per 16-byte piece, attached as
    `out.keep_mask` (consumed by Conv.dgrad_gn)."""
    L = _lib.lib()
    _contig(x1, x2, gamma, beta, stats)
    n, v = _nv(x1)
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    if out is None:
        out = torch.empty(x1.shape[:-1] + (c1 + c2,), dtype=x1.dtype, device=x1.device)
    mask = None
    if want_mask and dropout_p > 0.0:
        mask = torch.empty((n, v, (c1 + c2) // epl(x1.dtype)), dtype=torch.uint8, device=x1.device)
    if "gn_fwd" in ABLATE:
        out.keep_mask = mask
        return out
    ev = _pb()
    check(L.vdm_gn_silu_fwd(_p(x1), c1, _p(x2), c2, n, v, groups, dt_id(x1.dtype), _p(stats), _p(gamma), _p(beta), GN_EPS,
                            float(dropout_p), int(seed), _p(out), _p(mask), int(bool(linear)), _p(SEED_STEP), _s()), "vdm_gn_silu_fwd")
    out.keep_mask = mask
    _pe(ev, "gn_silu_fwd", 0.0, 2.0 * out.numel() * out.element_size())
    return out


def gn_skip_supported(c1, c2, cout, dtype):
    """(forward, backward): whether the 1x1x1 skip conv (c1 + c2 -> cout) can ride along with norm1's GroupNorm passes
    (csrc/gn_skip.hip: bf16 storage, narrow layers)."""
    if dtype != torch.bfloat16:
        return False, False
    m = _lib.lib().vdm_gn_skip_supported(int(c1), int(c2), int(cout), dt_id(dtype))
    return bool(m & 1), bool(m & 2)


def gn_silu_skip_fwd(x1, x2, groups, stats, gamma, beta, w1, w2, bias):
    """(silu(gn(concat(x1, x2))), w1 x1 + w2 x2 + bias) in one pass over the block input: norm1 + the 1x1x1 skip conv of a ResNetBlock.
    w1 / w2: fp32 master weights [cout, c1(, 1, 1, 1)] / [cout, c2(, 1, 1, 1)] (w2 None without a second source)."""
    L = _lib.lib()
    _contig(x1, x2, gamma, beta, stats, w1, w2, bias)
    n, v = _nv(x1)
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    cout = w1.shape[0]
    y = torch.empty(x1.shape[:-1] + (c1 + c2,), dtype=x1.dtype, device=x1.device)
    sk = torch.empty(x1.shape[:-1] + (cout,), dtype=x1.dtype, device=x1.device)
    y.keep_mask = None
    if "gn_fwd" in ABLATE:
        return y, sk
    ev = _pb()
    check(L.vdm_gn_silu_skip_fwd(_p(x1), c1, _p(x2), c2, n, v, groups, dt_id(x1.dtype), _p(stats), _p(gamma), _p(beta), GN_EPS,
                                 _p(w1), _p(w2), _p(bias), cout, _p(y), _p(sk), _s()), "vdm_gn_silu_skip_fwd")
    _pe(ev, "gn_silu_fwd", 2.0 * n * v * (c1 + c2) * cout, (2.0 * y.numel() + sk.numel()) * y.element_size())
    return y, sk


_skip_ws = {}


def gn_silu_bwd(x1, x2, groups, stats, gamma, beta, dy, dgamma, dbeta, add1=None, add2=None, colsum=None,
                dropout_p=0.0, seed=0, dx1=None, dx2=None, linear=False):
    """GroupNorm(+SiLU+dropout) backward for a gradient `dy` that did NOT come out of Conv.dgrad_gn: dyh = dy * keep * silu'(yhat) as
    its own pass (vdm_gn_dyh), its per-sample channel totals (vdm_channel_dot_sums), then the fixed-order finalize + apply kernels of
    the folded path (gn_bwd_fused) - bit-reproducible, no float atomics.  Returns (dx1, dx2); writes dgamma / dbeta and, if given,
    colsum[n, c] = sum_v dx (bias / conditioning-table gradients)."""
    L = _lib.lib()
    _contig(x1, x2, dy, add1, add2, gamma, beta, dgamma, dbeta, stats)
    n, v = _nv(x1)
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    ev = _pb()
    dyh = dy if (dx1 is not None and dx1.data_ptr() == dy.data_ptr()) else torch.empty_like(dy)      # in place when the caller gave dy away
    check(L.vdm_gn_dyh(_p(x1), c1, _p(x2), c2, n, v, groups, dt_id(x1.dtype), _p(stats), _p(gamma), _p(beta), GN_EPS, float(dropout_p),
                       int(seed), _p(dy), _p(dyh), int(bool(linear)), _p(SEED_STEP), _s()), "vdm_gn_dyh")
    dyh.gnb_partials = channel_dot_sums(dyh, x1, x2)
    if colsum is not None and getattr(stats, "chsum", None) is None:      # the analytic column sums need sum_v x per channel
        cs = [channel_dot_sums(t, t)[:, 0, :, 0] for t in (x1, x2) if t is not None]
        stats.chsum = torch.cat(cs, dim=1).contiguous()
    _pe(ev, "gn_dyh+dot_sums", 0.0, 5.0 * dy.numel() * dy.element_size())
    out = gn_bwd_fused(x1, x2, groups, stats, gamma, dyh, dgamma, dbeta, add1=add1, add2=add2, colsum=colsum, dx1=dx1, dx2=dx2)
    if colsum is not None:                # colsum = sum_v dx INCLUDING the residual-path terms (the analytic form covers the GroupNorm part)
        if add1 is not None:
            colsum[:, :c1] += channel_dot_sums(add1, add1)[:, 0, :, 0]
        if add2 is not None:
            colsum[:, c1:c1 + c2] += channel_dot_sums(add2, add2)[:, 0, :, 0]
    return out


def gn_tail_ok(conv_in, x):
    """Can the last GroupNorm backward + conv_in's weight gradient run as one pass (gn_bwd_fused(..., tail=...))?  bf16, 16 / 32 / 64
    channels, <= 2 input channels of conv_in, batch <= 16."""
    return (x.dtype == torch.bfloat16 and x.shape[-1] in (16, 32, 64) and conv_in.ksize == 3 and conv_in.stride == 1 and not conv_in.upsample
            and conv_in.cin <= 2 and conv_in.cout == x.shape[-1] and x.shape[0] <= 16 and (not conv_in.circular or x.shape[3] >= 17)
            and "VDM4CDM_NO_THIN_WGRAD" not in _os.environ and _os.environ.get("VDM4CDM_FUSED_TAIL", "1") != "0")


def gn_bwd_fused(x1, x2, groups, stats, gamma, dyh, dgamma, dbeta, add1=None, add2=None, colsum=None, dx1=None, dx2=None, skip=None, tail=None):
    """Second half of the GroupNorm+SiLU backward after Conv.dgrad_gn (dyh carries `.gnb_partials`).  Writes dgamma / dbeta,
    colsum (optional [N, >=C] fp32 view, row stride honoured; needs stats.chsum) and returns (dx1, dx2).  No float atomics.
    tail = (conv_in, xin, dw, dbias) (gn_tail_ok): x1 is the output of conv_in - its gradient is not written but folded straight into
    conv_in's weight / bias gradient (returns (None, None)).
    skip = (dout, w1, w2, dw1, dw2): the block's 1x1x1 skip conv rides along (gn_skip_supported(...)[1]): W^T dout is added to dx and
    the skip weight gradients dw1 / dw2 are written in the same pass (instead of add1 / add2 from a separate dgrad conv)."""
    L = _lib.lib()
    _contig(x1, x2, dyh, add1, add2, gamma, dgamma, dbeta, stats)
    n, v = _nv(x1)
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    C_ = c1 + c2
    part = dyh.gnb_partials
    assert part is not None and part.shape[0] == n and part.shape[2] == C_ and dyh.shape[-1] == C_
    red = torch.empty((n, groups, 2), dtype=torch.float32, device=x1.device)
    chan = torch.empty((n, C_, 2), dtype=torch.float32, device=x1.device)
    cstride, chsum = 0, None
    if colsum is not None:
        assert colsum.dtype == torch.float32 and colsum.stride(1) == 1 and colsum.shape[0] == n
        cstride, chsum = colsum.stride(0), getattr(stats, "chsum", None)
        assert chsum is not None, "gn_bwd_fused: colsum needs the per-channel sums (gn_stats(..., chsum=True))"
    if dx1 is None and tail is None:
        dx1 = dyh if x2 is None else torch.empty_like(x1)
    if x2 is not None and dx2 is None:
        dx2 = torch.empty_like(x2)
    ev = _pb()
    check(L.vdm_gn_bwd_finalize(_p(part), part.shape[1], n, C_, groups, v, _p(stats), _p(gamma), GN_EPS, _p(chsum), _p(red), _p(chan),
                                _p(colsum), cstride, _s()), "vdm_gn_bwd_finalize")
    if tail is not None:           # the last GroupNorm of the backward pass: dx feeds only conv_in's weight gradient - never written
        conv_in, xin, dw, dbias = tail
        assert x2 is None and add2 is None and skip is None
        _contig(xin, dw, dbias, add1)
        n_, od, oh, ow, _ = x1.shape
        d = conv_in.desc(n_, od, oh, ow, x1.dtype)
        need = L.vdm_conv_wgrad_workspace_bytes(d)
        ws = _scratch(Conv._ws, (x1.device, _s()), need, torch.uint8, x1.device, floor=32 << 20)
        if "gn_apply" not in ABLATE and "wgrad" not in ABLATE:
            check(L.vdm_gn_bwd_apply_wgrad_thin(_p(x1), c1, n, od, oh, ow, groups, _p(stats), _p(gamma), GN_EPS, _p(dyh), _p(red), _p(chan), _p(add1),
                                                _p(xin), conv_in.cin, 1 if conv_in.circular else 0, _p(dgamma), _p(dbeta), _p(dw), _p(dbias), _p(ws),
                                                ws.numel(), _s()), "vdm_gn_bwd_apply_wgrad_thin")
        _pe(ev, "gn_bwd(finalize+apply)", 2.0 * n * v * 27 * conv_in.cin * c1, (2.0 + (1.0 if add1 is not None else 0.0)) * dyh.numel() * dyh.element_size())
        return None, None
    if skip is not None:
        assert add1 is None and add2 is None
        dout, w1, w2, dw1, dw2 = skip
        _contig(dout, w1, w2, dw1, dw2)
        cout = w1.shape[0]
        nws = L.vdm_gn_skip_ws_floats(c1, c2, cout, n, v)
        ws = _scratch(_skip_ws, (x1.device, _s()), nws, torch.float32, x1.device)
        if "gn_apply" not in ABLATE:
            check(L.vdm_gn_bwd_apply_skip(_p(x1), c1, _p(x2), c2, n, v, groups, dt_id(x1.dtype), _p(stats), _p(gamma), GN_EPS, _p(dyh), _p(red),
                                          _p(chan), _p(dout), _p(w1), _p(w2), cout, _p(dx1), _p(dx2), _p(dgamma), _p(dbeta), _p(dw1), _p(dw2),
                                          _p(ws), ws.numel(), _s()), "vdm_gn_bwd_apply_skip")
        _pe(ev, "gn_bwd(finalize+apply)", 4.0 * n * v * C_ * cout, (3.0 * dyh.numel() + dout.numel()) * dyh.element_size())
        return dx1, dx2
    if "gn_apply" not in ABLATE:
      check(L.vdm_gn_bwd_apply(_p(x1), c1, _p(x2), c2, n, v, groups, dt_id(x1.dtype), _p(stats), _p(gamma), GN_EPS, _p(dyh), _p(red), _p(chan),
                             _p(add1), _p(add2), _p(dx1), _p(dx2), _p(dgamma), _p(dbeta), _s()), "vdm_gn_bwd_apply")
    _pe(ev, "gn_bwd(finalize+apply)", 0.0, (3.0 + (1.0 if add1 is not None else 0.0)) * dyh.numel() * dyh.element_size())
    return dx1, dx2


def pack_input(a, b, dtype, out=None):
    """[N, D, H, W] fp32 (+ optional second channel) -> NDHWC [N, D, H, W, cpad(2)] of `dtype`, zero padded."""
    L = _lib.lib()
    _contig(a, b)
    assert a.dtype == torch.float32 and (b is None or (b.dtype == torch.float32 and b.shape == a.shape))
    cp = cpad(2, dtype)
    if out is None:
        out = torch.empty(tuple(a.shape) + (cp,), dtype=dtype, device=a.device)
    check(L.vdm_pack_input(_p(a), _p(b), a.numel(), cp, dt_id(dtype), _p(out), _s()), "vdm_pack_input")
    return out


class CondTable:
    """K6: the conditioning MLPs and the additive injection table of all blocks (vdm_cond_table_*).
    specs: list of dicts {input: tensor ([rows] for the sinusoidal t embedding, [rows, d] for a vector), sinusoid: bool, in_dim,
    w1, b1, w2, b2, wproj: fp32 device tensors (views into the flat parameter vector)}."""

    def __init__(self, specs, rows, width):
        self.specs, self.rows, self.width = specs, rows, width
        self.saved = None

    def _descs(self, grads=None):
        arr = (CondMlp * len(self.specs))()
        for k, sp in enumerate(self.specs):
            _contig(sp["input"], sp["w1"], sp["b1"], sp["w2"], sp["b2"], sp["wproj"])
            assert sp["input"].dtype == torch.float32 and sp["input"].shape[0] == self.rows
            d = arr[k]
            d.input, d.in_dim, d.dim, d.sinusoid = _p(sp["input"]), int(sp["in_dim"]), int(sp["w2"].shape[0]), 1 if sp["sinusoid"] else 0
            d.w1, d.b1, d.w2, d.b2, d.wproj = _p(sp["w1"]), _p(sp["b1"]), _p(sp["w2"]), _p(sp["b2"]), _p(sp["wproj"])
            assert tuple(sp["w1"].shape) == (d.dim, d.in_dim) and tuple(sp["wproj"].shape) == (self.width, d.dim)
            if grads is not None:
                g = grads[k]
                _contig(g["w1"], g["b1"], g["w2"], g["b2"], g["wproj"])
                d.dw1, d.db1, d.dw2, d.db2, d.dwproj = _p(g["w1"]), _p(g["b1"]), _p(g["w2"]), _p(g["b2"]), _p(g["wproj"])
        return arr

    def forward(self, save):
        L = _lib.lib()
        dev = self.specs[0]["w1"].device
        arr = self._descs()
        table = torch.empty((self.rows, self.width), dtype=torch.float32, device=dev)
        self.saved = torch.empty(L.vdm_cond_saved_floats(arr, len(self.specs), self.rows), dtype=torch.float32, device=dev) if save else None
        check(L.vdm_cond_table_fwd(arr, len(self.specs), self.rows, self.width, _p(table), _p(self.saved), _s()), "vdm_cond_table_fwd")
        return table

    def backward(self, dtable, grads, dbias=None):
        """grads: list of dicts {w1, b1, w2, b2, wproj} of fp32 views that receive the parameter gradients (plain stores)."""
        L = _lib.lib()
        assert self.saved is not None and dtable.dtype == torch.float32 and dtable.stride(1) == 1 and dtable.shape[0] == self.rows
        arr = self._descs(grads)
        scratch = torch.empty(L.vdm_cond_bwd_scratch_floats(arr, len(self.specs), self.rows, self.width), dtype=torch.float32, device=dtable.device)
        check(L.vdm_cond_table_bwd(arr, len(self.specs), self.rows, self.width, _p(dtable), dtable.stride(0), _p(self.saved), _p(scratch),
                                   _p(dbias), _s()), "vdm_cond_table_bwd")
        self.saved = None


def cond_table_step(table_t, table_v, step_ptr, rows, width, out):
    _contig(table_t, table_v, out)
    check(_lib.lib().vdm_cond_table_step(_p(table_t), _p(table_v), _p(step_ptr), rows, width, _p(out), _s()), "vdm_cond_table_step")
    return out


def augment_batch(fields, consts, samples, crop, out=None):
    """Data path on the device (vdm_augment_batch): fields = per-channel raw cube stacks [n_sims, S, S, S] fp32 on the GPU, consts =
    per-channel (alpha, mean, std), samples = list of (sim, anchor[3], flip[3], perm[3]).  Returns the per-channel batches
    [B, 1, crop, crop, crop] fp32: periodic crop at the anchor -> (log10(x + alpha) - mean) / std -> flips -> axis permutation."""
    L = _lib.lib()
    C_, B = len(fields), len(samples)
    S = fields[0].shape[-1]
    for f in fields:
        assert f.is_cuda and f.dtype == torch.float32 and f.is_contiguous() and f.dim() == 4 and tuple(f.shape[1:]) == (S, S, S), \
            "raw fields must be contiguous fp32 [n_sims, S, S, S] tensors on the GPU"
    nsim = min(f.shape[0] for f in fields)
    if out is None:
        out = [torch.empty(B, 1, crop, crop, crop, device=fields[0].device) for _ in range(C_)]
    ch = (_lib.AugmentChannel * C_)()
    for c, (f, (alpha, mean, std)) in enumerate(zip(fields, consts)):
        _contig(out[c])
        ch[c].field, ch[c].out, ch[c].alpha, ch[c].mean, ch[c].std = f.data_ptr(), out[c].data_ptr(), float(alpha), float(mean), float(std)
    tab = (_lib.AugmentSample * B)()
    for b, (sim, anchor, flip, perm) in enumerate(samples):
        assert 0 <= int(sim) < nsim, f"simulation index {sim} out of range (0..{nsim - 1})"
        tab[b].sim = int(sim)
        for d in range(3):
            tab[b].anchor[d], tab[b].flip[d], tab[b].perm[d] = int(anchor[d]), int(bool(flip[d])), int(perm[d])
    check(L.vdm_augment_batch(ch, C_, int(S), int(crop), tab, B, _s()), "vdm_augment_batch")
    return out


def channel_sums(x, out):
    """out[c] (fp32 view) = sum over all leading dims of x[..., c]."""
    _contig(x, out)
    c = x.shape[-1]
    check(_lib.lib().vdm_channel_sums(_p(x), x.numel() // c, c, dt_id(x.dtype), _p(out), _s()), "vdm_channel_sums")
    return out


def channel_dot_sums(a, b1, b2=None):
    """[N, 1, C, 2] fp32 = per sample and channel (sum a, sum a * b) over the voxels, b = concat(b1, b2): gnb_partials (one "tile" per
    sample) for gn_bwd_fused when the gradient a of a GroupNorm did not come out of Conv.dgrad_gn."""
    _contig(a, b1, b2)
    n, c = a.shape[0], a.shape[-1]
    c1 = b1.shape[-1]
    c2 = 0 if b2 is None else b2.shape[-1]
    assert c1 + c2 == c and b1.dtype == a.dtype and b1.shape[:-1] == a.shape[:-1]
    out = torch.empty((n, 1, c, 2), dtype=torch.float32, device=a.device)
    check(_lib.lib().vdm_channel_dot_sums(_p(a), _p(b1), c1, _p(b2), c2, n, a.numel() // (n * c), dt_id(a.dtype), _p(out), _s()),
          "vdm_channel_dot_sums")
    return out


def attn_split_heads(src, sub, nsub, heads, rowmajor=True, transposed=True):
    """src: [N, ..., nsub * C] activations (sub-tensor `sub` of nsub: the q / k / v of the qkv conv, or nsub = 1) -> (row-major
    [N, H, V, hd], transposed [N, H, hd, V]) in src's dtype (either None when not requested)."""
    _contig(src)
    N = src.shape[0]
    C = src.shape[-1] // nsub
    V = src.numel() // (N * nsub * C)
    hd = C // heads
    rm = torch.empty((N, heads, V, hd), dtype=src.dtype, device=src.device) if rowmajor else None
    tr = torch.empty((N, heads, hd, V), dtype=src.dtype, device=src.device) if transposed else None
    check(_lib.lib().vdm_attn_split_heads(_p(src), nsub * C, sub * C, N, V, heads, hd, dt_id(src.dtype), _p(rm), _p(tr), _s()), "vdm_attn_split_heads")
    return rm, tr


def attn_fwd(q, k, vt, scale, out_shape, want_lse=True):
    """q, k: [N, H, V, hd]; vt: [N, H, hd, V] -> out (out_shape = [N, ..., H*hd], the NDHWC tensor the projection conv reads), lse [N, H, V]."""
    _contig(q, k, vt)
    N, H, V, hd = q.shape
    out = torch.empty(out_shape, dtype=q.dtype, device=q.device)
    lse = torch.empty((N, H, V), dtype=torch.float32, device=q.device) if want_lse else None
    check(_lib.lib().vdm_attn_fwd(_p(q), _p(k), _p(vt), N, V, H, hd, dt_id(q.dtype), float(scale), _p(out), _p(lse), _s()), "vdm_attn_fwd")
    return out, lse


def attn_bwd(q, k, v, qt, kt, dout, out, lse, scale):
    """-> dqkv [N, ..., 3 * H*hd] (the gradient of the qkv conv's output).  dout / out: [N, ..., H*hd]."""
    _contig(q, k, v, qt, kt, dout, out, lse)
    N, H, V, hd = q.shape
    L = _lib.lib()
    doh, dot = attn_split_heads(dout, 0, 1, H)
    dsum = torch.empty((N, H, V), dtype=torch.float32, device=q.device)
    check(L.vdm_attn_rowdot(_p(dout), _p(out), N, V, H, hd, dt_id(q.dtype), _p(dsum), _s()), "vdm_attn_rowdot")
    dqkv = torch.empty(tuple(dout.shape[:-1]) + (3 * H * hd,), dtype=q.dtype, device=q.device)
    check(L.vdm_attn_bwd(_p(q), _p(k), _p(v), _p(qt), _p(kt), _p(doh), _p(dot), _p(lse), _p(dsum), N, V, H, hd, dt_id(q.dtype), float(scale),
                         _p(dqkv), _s()), "vdm_attn_bwd")
    return dqkv


def diffuse(x, eps, alpha, sigma, out=None):
    L = _lib.lib()
    _contig(x, eps, alpha, sigma)
    n = x.shape[0]
    if out is None:
        out = torch.empty_like(x)
    check(L.vdm_diffuse(_p(x), _p(eps), _p(alpha), _p(sigma), n, x.numel() // n, _p(out), _s()), "vdm_diffuse")
    return out


def diffuse_pack(x, s_cond, alpha, sigma, dtype, eps=None, seed=0, stream_id=0, want_z=False):
    """Fused head of the training step (vdm_diffuse_pack): z_t = alpha[n] x + sigma[n] eps -> (z_t fp32 or None, conv_in's NDHWC input
    [N, D, H, W, cpad(2)] = {z_t, s_cond, 0...} in `dtype`).  eps None: drawn inside the kernel from Philox(seed, stream_id) - the field
    randn(out, seed, stream_id) would have written."""
    L = _lib.lib()
    _contig(x, s_cond, eps, alpha, sigma)
    n = x.shape[0]
    per = x.numel() // n
    assert x.dtype == torch.float32 and per % 4 == 0 and (s_cond is None or (s_cond.shape == x.shape and s_cond.dtype == torch.float32))
    sp = tuple(x.shape[2:]) if x.dim() == 5 else tuple(x.shape[1:])
    packed = torch.empty((n,) + sp + (cpad(2, dtype),), dtype=dtype, device=x.device)
    z = torch.empty_like(x) if want_z else None
    check(L.vdm_diffuse_pack(_p(x), _p(s_cond), _p(eps), int(seed), int(stream_id), _p(SEED_STEP), _p(alpha), _p(sigma), n, per, dt_id(dtype),
                             _p(z), _p(packed), _s()), "vdm_diffuse_pack")
    return z, packed


_red_ws = {}


def _reduce_ws(device):
    """per-(device, stream) scratch of the fixed-order two-stage reductions (loss terms, gradient norm)"""
    return _scratch(_red_ws, (device, _s()), 2048 * 3, torch.float32, device)


def loss_terms(x, eps, eps_hat, eps0, sigma0_over_alpha0, coef, sums, d_eps_hat, rng=None):
    """K8.  rng = ((seed_eps, stream_eps), (seed_eps0, stream_eps0)): a noise field passed as None is regenerated inside the kernel from
    its Philox counters (vdm_loss_terms_rng) instead of read."""
    L = _lib.lib()
    _contig(x, eps, eps_hat, eps0, coef, sums, d_eps_hat)
    n = x.shape[0]
    per = x.numel() // n
    if per % 4 == 0 and (rng is not None or (eps is not None and eps0 is not None)):      # vector kernel (also for supplied fields: one
        (se, ie), (s0, i0) = rng if rng is not None else ((0, 0), (0, 0))                   # summation order for both forms)
        assert rng is not None or (eps is not None and eps0 is not None)
        check(L.vdm_loss_terms_rng(_p(x), _p(eps), int(se), int(ie), _p(eps_hat), _p(eps0), int(s0), int(i0), _p(SEED_STEP),
                                   float(sigma0_over_alpha0), _p(coef), n, per, _p(sums), _p(d_eps_hat), _p(_reduce_ws(x.device)), _s()),
              "vdm_loss_terms_rng")
        return
    assert eps is not None and eps0 is not None, "loss_terms: noise fields missing (and no Philox counters to regenerate them from)"
    check(L.vdm_loss_terms(_p(x), _p(eps), _p(eps_hat), _p(eps0), float(sigma0_over_alpha0), _p(coef), n, per,
                           _p(sums), _p(d_eps_hat), _p(_reduce_ws(x.device)), _s()), "vdm_loss_terms")


def ancestral_step(z, eps_hat, noise, coef, step_ptr, seed, eps_uncond=None, w_cfg=0.0):
    """K9.  eps_uncond given: classifier-free guidance, (1 + w_cfg) * eps_hat - w_cfg * eps_uncond blended inside the update."""
    L = _lib.lib()
    _contig(z, eps_hat, noise, coef, eps_uncond)
    if eps_uncond is None:
        check(L.vdm_ancestral_step(_p(z), _p(eps_hat), _p(noise), _p(coef), _p(step_ptr), int(seed), z.numel(), _s()),
              "vdm_ancestral_step")
    else:
        assert eps_uncond.shape == eps_hat.shape == z.shape and eps_uncond.dtype == torch.float32
        check(L.vdm_ancestral_step_cfg(_p(z), _p(eps_hat), _p(eps_uncond), float(w_cfg), _p(noise), _p(coef), _p(step_ptr), int(seed),
                                       z.numel(), _s()), "vdm_ancestral_step_cfg")


def randn(out, seed, stream_id=0):
    L = _lib.lib()
    _contig(out)
    assert out.dtype == torch.float32
    check(L.vdm_randn(_p(out), out.numel(), int(seed), int(stream_id), _p(SEED_STEP), _s()), "vdm_randn")
    return out


def step_inc(step_ptr):
    check(_lib.lib().vdm_step_inc(_p(step_ptr), _s()), "vdm_step_inc")


def sumsq(x, out):
    L = _lib.lib()
    _contig(x)
    assert x.dtype == torch.float32 and out.dtype == torch.float32
    check(L.vdm_sumsq(_p(x), x.numel(), _p(out), _p(_reduce_ws(x.device)), _s()), "vdm_sumsq")
    return out


def train_scalars(B, device, rank, world, gamma_min, gamma_max, bpd_over_B, u0=None, times=None, seed=0):
    """[5, B] fp32 = {t, alpha_t, sigma_t, coef, t_norm} (vdm_train_scalars); u0: device tensor with one uniform draw, or times [B];
    neither: u0 is drawn inside the kernel from (seed, SEED_STEP) - the graph-captured training step."""
    out = torch.empty((5, B), dtype=torch.float32, device=device)
    _contig(u0, times)
    check(_lib.lib().vdm_train_scalars(_p(u0), _p(times), B, rank, world, float(gamma_min), float(gamma_max), float(bpd_over_B), _p(out),
                                       int(seed), _p(SEED_STEP), _s()), "vdm_train_scalars")
    return out


def elbo_assemble(sums, coef, c_lat0, c_lat1, c_rec0, c_rec1):
    _contig(sums, coef)
    out = torch.empty(4, dtype=torch.float32, device=sums.device)
    check(_lib.lib().vdm_elbo_assemble(_p(sums), _p(coef), sums.shape[0], float(c_lat0), float(c_lat1), float(c_rec0), float(c_rec1), _p(out), _s()),
          "vdm_elbo_assemble")
    return out


def clip_scale_(x, sumsq_acc, max_norm):
    _contig(x)
    check(_lib.lib().vdm_clip_scale(_p(x), x.numel(), _p(sumsq_acc), float(max_norm), _s()), "vdm_clip_scale")
    return x


class PackPlan:
    """All (conv, form) weight packings of a network as one launch (vdm_conv_pack_many): the item and chunk tables are built
    once per (parameter storage, dtype, forms) and live on the device."""

    def __init__(self, convs_and_masters, dtype, need_dgrad):
        """convs_and_masters: [(Conv, fp32 master view [taps, cout, cin] inside the flat parameter vector)]."""
        import numpy as np
        L = _lib.lib()
        device = convs_and_masters[0][1].device
        items = []
        for conv, w in convs_and_masters:
            assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == conv.ksize ** 3 * conv.cout * conv.cin
            conv.alloc_packed(device, dtype, need_dgrad)
            d = conv.desc(1, 2, 2, 2, dtype)
            for mode, buf in ((PACK_FWD, conv.wf),) + (((PACK_DGRAD, conv.wd),) if need_dgrad else ()):
                it = _lib.PackItem()
                check(L.vdm_conv_pack_plan(d, mode, _p(w), _p(buf), C.byref(it)), "vdm_conv_pack_plan")
                items.append(it)
        chunks = []
        for i, it in enumerate(items):
            for first in range(0, it.elems, _lib.PACK_CHUNK):
                chunks.append((i, min(_lib.PACK_CHUNK, it.elems - first), first))
        item_arr = (_lib.PackItem * len(items))(*items)
        chunk_arr = (_lib.PackChunk * len(chunks))(*[_lib.PackChunk(*c) for c in chunks])
        self.items = torch.from_numpy(np.frombuffer(bytes(item_arr), dtype=np.uint8).copy()).to(device)
        self.chunks = torch.from_numpy(np.frombuffer(bytes(chunk_arr), dtype=np.uint8).copy()).to(device)
        self.nchunks = len(chunks)
        self.dtype = dtype

    def run(self):
        if "pack" in ABLATE and getattr(self, "_ran", False):
            return
        self._ran = True
        check(_lib.lib().vdm_conv_pack_many(_p(self.items), _p(self.chunks), self.nchunks, dt_id(self.dtype), _s()), "vdm_conv_pack_many")
