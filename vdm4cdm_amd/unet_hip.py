"""HIP execution of the CUNet graph: forward and hand-written backward over the C-ABI kernels.

The graph (D1-D8) is walked on the host; every activation-sized operation is one call
into libvdm4cdm_hip.so on torch's current stream (so a caller can capture the whole denoise step in
a hipGraph).  PyTorch only allocates buffers and runs the tiny conditioning MLPs (R5).

Tensors: activations NDHWC ``[N, D, H, W, C]`` in the compute dtype (bf16 or fp32); ``eps_hat`` fp32.
Backward returns the gradient of the flat parameter vector and of the conditioning table.
"""
import torch

import os

from . import hip_ops as ops
from .hip_ops import Conv

# GroupNorm statistics from the producing conv's epilogue (VDM4CDM_FUSED_GN=0: separate gn_stats passes, for A/B timing)
FUSED_GN = os.environ.get("VDM4CDM_FUSED_GN", "1") != "0"
# GroupNorm backward reduction folded into the producing dgrad conv's epilogue (VDM4CDM_FUSED_GNB=0: separate reduce pass with
# float atomics, for A/B timing).  Needs the forward partials (FUSED_GN) for the analytic conditioning-table gradient.
FUSED_GNB = FUSED_GN and os.environ.get("VDM4CDM_FUSED_GNB", "1") != "0"
GNB_MIN_K = int(os.environ.get("VDM4CDM_GNB_MIN_K", "0"))      # fold only into dgrad convs with at least this many reduction channels
# the 1x1x1 skip conv of a ResNetBlock rides along with norm1's GroupNorm passes where csrc/gn_skip.hip has a kernel (bf16, narrow layers)
FUSED_SKIP = os.environ.get("VDM4CDM_FUSED_SKIP", "1") != "0"
# input gradient (+ folded GroupNorm backward) and weight gradient of a 32 -> 32 conv as ONE launch that stages the gradient once
# (csrc/conv_dgw.hip; the level-0 convs of the 128^3 network).  Opt-in (VDM4CDM_FUSED_DGW=1) since the stand-alone weight-gradient kernel
# got its rolling z window and copy-free operand tuples: alone 0.55 ms fused against 0.345 + 0.213 ms separately, training step 12.23 fused
# against 12.02 ms separate (same box, alternating)
FUSED_DGW = os.environ.get("VDM4CDM_FUSED_DGW", "0") == "1"
# inference: norm2's GroupNorm + SiLU applied inside conv2 (to the staged halo image) instead of by a pass of its own.  Measured: the
# VALU work on the 2.1x halo costs the conv what the 5 TB/s pass cost - off by default
# "0" (default): nowhere; "deep": only at the levels of <= 32^3 voxels; "1": everywhere.  Round 4, same box, 300 sampling steps at 128^3:
# 2.42 / 2.43 / 2.46 ms per step for 0 / deep / 1 - the fold does not pay at any level (the level-3 convs run the K-split kernel, which
# has no prologue: "deep" reaches two launches per step)
GN_PROLOGUE = os.environ.get("VDM4CDM_GN_PROLOGUE", "0")
GN_PROLOGUE_MAX_VOXELS = 32 ** 3

class SideStream:
    """Weight-gradient kernels run on a side HIP stream: their results are needed only by the optimiser, so they overlap
    with the (HBM-bound) GroupNorm kernels and the dgrad chain on the main stream.  `run(fn, *tensors)` makes the side
    stream wait for everything issued so far on the main stream, runs fn on it and marks the tensors as in use there."""

    def __init__(self, device, enabled=True, with_second=True):
        self.enabled = enabled and torch.cuda.is_available()
        self.side = torch.cuda.Stream(device=device) if self.enabled else None
        # a second, independent side stream (the skip-path input gradients must not queue behind the weight gradients)
        self.second = SideStream(device, enabled and os.environ.get("VDM4CDM_SKIP_DGRAD_STREAM", "1") != "0", with_second=False) if with_second else None

    def run(self, fn, *tensors):
        if not self.enabled:
            return fn()
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        prev, ops.SIDE_STREAM = ops.SIDE_STREAM, self.side      # hip_ops._p: every tensor handed to a launch in here is recorded on the
        try:                                                    # side stream (the ONE place where main-stream memory meets a side queue)
            with torch.cuda.stream(self.side):
                fn()
        finally:
            ops.SIDE_STREAM = prev
        for t in tensors:
            if t is not None:
                t.record_stream(self.side)

    def join(self):
        if self.enabled:
            torch.cuda.current_stream().wait_stream(self.side)


class DeferredSide:
    """A SideStream whose run() only RECORDS the launch: the weight-gradient kernels of the level-0 up path are not started while the
    main stream is in its own level-0 kernels (every one of them fills the chip alone: co-running them is time-slicing, the folded
    dgrad convs there ran 1.7-2.5x longer than alone) but flushed onto the side stream when the main stream moves on to the deeper
    levels, whose small grids leave most CUs idle.  second / join / enabled are the real stream's (the skip-path input gradients feed
    the main chain and are never deferred)."""

    def __init__(self, ss):
        self.ss, self.pending = ss, []
        self.enabled, self.second, self.side = ss.enabled, ss.second, ss.side

    def run(self, fn, *tensors):
        self.pending.append((fn, tensors))

    def join(self):
        self.ss.join()

    def flush(self):
        """ONE fork of the side stream for all held-back launches.  (Flushing them with one wait_stream each - four event records at the
        same point of the main stream - was fine eagerly but produced a wrong hipGraph under stream capture: kernels issued on the main
        stream right after the flush saw stale inputs.  Found by comparing a replayed step with the eager one tensor by tensor.)"""
        pend, self.pending = self.pending, []
        if not pend:
            return

        def all_():
            for fn, _ in pend:
                fn()
        self.ss.run(all_, *[t for _, ts in pend for t in ts])


# weight gradients of the up path's first N levels are deferred until the main stream starts the next level (0: launch immediately)
DEFER_WGRAD_LEVELS = int(os.environ.get("VDM4CDM_DEFER_WGRAD_LEVELS", "1"))


class _Res:
    """ResNetBlock (D4) on HIP kernels; input may be two tensors (skip concat, never materialised)."""

    def __init__(self, net, info):
        self.i, self.net = info, net
        circ = net.conv_padding_mode == "circular"
        cin = info.c1 + info.c2
        self.conv1 = Conv(cin, info.cout, 3, circular=circ)
        self.conv2 = Conv(info.cout, info.cout, 3, circular=circ)
        self.skip1 = Conv(info.c1, info.cout, 1) if info.has_skip else None
        self.skip2 = Conv(info.c2, info.cout, 1) if (info.has_skip and info.c2) else None
        self.saved = None

    def convs(self):
        n = self.i.name
        out = [(self.conv1, n + ".conv1.weight"), (self.conv2, n + ".conv2.weight")]
        if self.skip1:
            out.append((self.skip1, n + ".skip.weight"))
        if self.skip2:
            out.append((self.skip2, n + ".skip2.weight"))
        return out

    def skip_rides(self, dtype):
        """(forward, backward): the skip conv is computed inside norm1's GroupNorm passes."""
        if self.skip1 is None or not FUSED_SKIP:
            return False, False
        return ops.gn_skip_supported(self.i.c1, self.i.c2, self.i.cout, dtype)

    def skip_weights(self, P):
        """The fp32 master weights of the two column blocks of the skip conv as [cout, c] matrices (views of the flat vector)."""
        i, n = self.i, self.i.name
        return P(n + ".skip.weight").view(i.cout, i.c1), (P(n + ".skip2.weight").view(i.cout, i.c2) if self.skip2 is not None else None)

    def fwd(self, P, x1, x2, table, save, p, seed, ss):
        i, G, n = self.i, self.net.norm_groups, self.i.name
        skip_out = []
        ride = self.skip_rides(x1.dtype)[0]      # norm1 and the skip conv read the same tensor: one pass (csrc/gn_skip.hip)
        if self.skip1 is not None and not ride:  # the 1x1 skip convs (HBM-bound) overlap with conv1 on the side stream
            def side_skip():
                s = self.skip1.fwd(x1, P(n + ".skip.bias"))
                if self.skip2 is not None:
                    s = self.skip2.fwd(x2, None, None, s)
                skip_out.append(s)
            ss.run(side_skip, x1, x2)
        st1 = ops.gn_stats(x1, x2, G)
        if ride:
            a1, s = ops.gn_silu_skip_fwd(x1, x2, G, st1, P(n + ".norm1.weight"), P(n + ".norm1.bias"), *self.skip_weights(P), P(n + ".skip.bias"))
        else:
            a1 = ops.gn_silu_fwd(x1, x2, G, st1, P(n + ".norm1.weight"), P(n + ".norm1.bias"))
        h = self.conv1.fwd(a1, P(n + ".conv1.bias"), table[:, i.table_off:i.table_off + i.cout], gn=FUSED_GN)
        st2 = ops.gn_stats(h, None, G, chsum=save and FUSED_GNB)
        inside = (GN_PROLOGUE != "0" and not save and p == 0.0 and (GN_PROLOGUE == "1" or h[0, ..., 0].numel() <= GN_PROLOGUE_MAX_VOXELS)
                  and self.conv2.gn_in_ok(h))
        a2 = None if inside else ops.gn_silu_fwd(h, None, G, st2, P(n + ".norm2.weight"), P(n + ".norm2.bias"), p, seed, want_mask=save and FUSED_GNB)
        if ride:
            pass
        elif self.skip1 is not None:
            ss.join()
            s = skip_out[0]
            if ss.enabled:
                s.record_stream(torch.cuda.current_stream())
        else:
            s = x1
        if inside:
            out = self.conv2.fwd(h, P(n + ".conv2.bias"), None, s, gn=FUSED_GN, gn_in=(G, st2, P(n + ".norm2.weight"), P(n + ".norm2.bias")))
        else:
            out = self.conv2.fwd(a2, P(n + ".conv2.bias"), None, s, gn=FUSED_GN)     # (every block output feeds a GroupNorm)
        if save:
            self.saved = (x1, x2, st1, a1, h, st2, a2, p, seed, a2.keep_mask)
        return out

    def bwd(self, P, GP, dout, dtable, ss, tail=None):
        """dout: gradient of the block output.  Returns (dx1, dx2).  Fills parameter grads via GP(name).
        ss: SideStream for the weight-gradient kernels.  tail = (conv_in, xin, dw, dbias) (first block of the network, when
        ops.gn_tail_ok): the block's input gradient is not materialised, conv_in's weight gradient comes out of norm1's apply pass."""
        i, G, n = self.i, self.net.norm_groups, self.i.name
        x1, x2, st1, a1, h, st2, a2, p, seed, mask2 = self.saved
        self.saved = None
        fused = (FUSED_GNB and getattr(st2, "chsum", None) is not None and (p == 0.0 or mask2 is not None) and i.cout >= GNB_MIN_K
                 and self.conv2.gn_fold_ok(i.cout, 0, h.dtype) and self.conv1.gn_fold_ok(i.c1, i.c2, h.dtype))

        skip_grads = []
        ride = fused and self.skip_rides(h.dtype)[1]      # skip dgrad + wgrad inside norm1's apply pass (csrc/gn_skip.hip)
        if self.skip1 is not None and not ride:  # input gradients of the 1x1 skip convs: independent of the main chain until norm1
            def side_skip_dgrad():
                skip_grads.append(self.skip1.dgrad(dout))
                skip_grads.append(self.skip2.dgrad(dout) if self.skip2 is not None else None)
            ss.second.run(side_skip_dgrad, dout)

        dgw2 = fused and FUSED_DGW and self.conv2.dgw_ok(dout, i.cout, 0)      # conv2: both gradients from one staging of dout
        tcols = dtable[:, i.table_off:i.table_off + i.cout]
        if dgw2:
            dyh2 = self.conv2.dgrad_gn_wgrad(dout, a2, h, None, G, st2, P(n + ".norm2.weight"), P(n + ".norm2.bias"), GP(n + ".conv2.weight"),
                                             GP(n + ".conv2.bias"), keep_mask=mask2, dropout_p=p)

        def side_conv2(a2=a2, dout=dout, x1=x1, x2=x2):      # conv2 (+ the 1x1 skip convs share dout); tensors bound now: may run deferred
            if not dgw2:
                self.conv2.wgrad(a2, dout, GP(n + ".conv2.weight"), GP(n + ".conv2.bias"))      # bias grad fused (column sums of dout)
            if self.skip1 is not None:
                GP(n + ".skip.bias").copy_(GP(n + ".conv2.bias"))          # same column sums of dout
                if not ride:
                    self.skip1.wgrad(x1, dout, GP(n + ".skip.weight"))
                    if self.skip2 is not None:
                        self.skip2.wgrad(x2, dout, GP(n + ".skip2.weight"))
        if not dgw2 or self.skip1 is not None:
            ss.run(side_conv2, a2, dout, x1, x2)
        if dgw2:
            dh, _ = ops.gn_bwd_fused(h, None, G, st2, P(n + ".norm2.weight"), dyh2, GP(n + ".norm2.weight"), GP(n + ".norm2.bias"),
                                     colsum=tcols)
        elif fused:      # dgrad epilogue: dyh = da2 * keep * silu'(.) + per-tile sums; then one finalize + one apply pass (no atomics)
            dyh2 = self.conv2.dgrad_gn(dout, h, None, G, st2, P(n + ".norm2.weight"), P(n + ".norm2.bias"), keep_mask=mask2, dropout_p=p)
            dh, _ = ops.gn_bwd_fused(h, None, G, st2, P(n + ".norm2.weight"), dyh2, GP(n + ".norm2.weight"), GP(n + ".norm2.bias"),
                                     colsum=tcols)
        else:
            da2 = self.conv2.dgrad(dout)
            dh, _ = ops.gn_silu_bwd(h, None, G, st2, P(n + ".norm2.weight"), P(n + ".norm2.bias"), da2,
                                    GP(n + ".norm2.weight"), GP(n + ".norm2.bias"), colsum=tcols, dropout_p=p, seed=seed, dx1=da2)
        del a2
        # conv1
        if fused and FUSED_DGW and self.conv1.dgw_ok(dh, i.c1, i.c2):
            dyh1 = self.conv1.dgrad_gn_wgrad(dh, a1, x1, x2, G, st1, P(n + ".norm1.weight"), P(n + ".norm1.bias"), GP(n + ".conv1.weight"))
        else:
            ss.run(lambda a1=a1, dh=dh: self.conv1.wgrad(a1, dh, GP(n + ".conv1.weight")), a1, dh)
            if fused:
                dyh1 = self.conv1.dgrad_gn(dh, x1, x2, G, st1, P(n + ".norm1.weight"), P(n + ".norm1.bias"))
            else:
                da1 = self.conv1.dgrad(dh)
        del a1, dh
        # skip path
        add1 = add2 = None
        if ride:
            return ops.gn_bwd_fused(x1, x2, G, st1, P(n + ".norm1.weight"), dyh1, GP(n + ".norm1.weight"), GP(n + ".norm1.bias"),
                                    skip=(dout,) + self.skip_weights(P) + self.skip_weights(GP))
        if self.skip1 is not None:
            ss.second.join()
            add1, add2 = skip_grads
            if ss.second.enabled:
                for t in (add1, add2):
                    if t is not None:
                        t.record_stream(torch.cuda.current_stream())
        else:
            add1 = dout
        if fused:
            return ops.gn_bwd_fused(x1, x2, G, st1, P(n + ".norm1.weight"), dyh1, GP(n + ".norm1.weight"), GP(n + ".norm1.bias"),
                                    add1=add1, add2=add2, tail=tail if (tail is not None and x2 is None and add2 is None) else None)
        return ops.gn_silu_bwd(x1, x2, G, st1, P(n + ".norm1.weight"), P(n + ".norm1.bias"), da1,
                               GP(n + ".norm1.weight"), GP(n + ".norm1.bias"), add1=add1, add2=add2)


class _Attn:
    """Self-attention between the two mid blocks (CUNet(mid_attn=True), spec D13): GroupNorm (no activation) -> 1x1x1 conv to q, k, v
    -> softmax(q k^T / sqrt(hd)) v over all voxels of the level -> 1x1x1 projection + residual.  Everything is a kernel of this library:
    GroupNorm, the two projections (MFMA convs) and the fused attention core (csrc/attention.hip: scores, online softmax and P V on the
    matrix cores, forward and backward, no [N, heads, V, V] tensor - 0.5 GB at 16^3 voxels before)."""

    def __init__(self, net):
        self.net = net
        C = net.chs[-1]
        self.C, self.H = C, net.n_attention_heads
        self.qkv = Conv(C, 3 * C, 1)
        self.proj = Conv(C, C, 1)
        self.saved = None

    def convs(self):
        return [(self.qkv, "mid_attn.qkv.weight"), (self.proj, "mid_attn.proj.weight")]

    def fwd(self, P, x, save):
        G, C, H = self.net.norm_groups, self.C, self.H
        st = ops.gn_stats(x, None, G)
        xn = ops.gn_silu_fwd(x, None, G, st, P("mid_attn.norm.weight"), P("mid_attn.norm.bias"), linear=True)
        qkv = self.qkv.fwd(xn, P("mid_attn.qkv.bias"))                                   # [N, d, h, w, 3C]
        scale = 1.0 / (C // H) ** 0.5
        q, qt = ops.attn_split_heads(qkv, 0, 3, H, transposed=save)
        k, kt = ops.attn_split_heads(qkv, 1, 3, H, transposed=save)
        v, vt = ops.attn_split_heads(qkv, 2, 3, H, rowmajor=save)
        a, lse = ops.attn_fwd(q, k, vt, scale, x.shape, want_lse=save)
        out = self.proj.fwd(a, P("mid_attn.proj.bias"), None, x, gn=FUSED_GN)          # + residual; feeds mid.1's GroupNorm
        if save:
            self.saved = (x, st, xn, q, k, v, qt, kt, a, lse)
        return out

    def bwd(self, P, GP, dout, ss):
        G, C, H = self.net.norm_groups, self.C, self.H
        x, st, xn, q, k, v, qt, kt, a, lse = self.saved
        self.saved = None
        scale = 1.0 / (C // H) ** 0.5
        ss.run(lambda: self.proj.wgrad(a, dout, GP("mid_attn.proj.weight")), a, dout)
        ops.channel_sums(dout, GP("mid_attn.proj.bias"))
        da = self.proj.dgrad(dout)                                                       # gradient of the attention output [N, ..., C]
        dqkv = ops.attn_bwd(q, k, v, qt, kt, da, a, lse, scale)                          # [N, ..., 3C]
        ss.run(lambda: self.qkv.wgrad(xn, dqkv, GP("mid_attn.qkv.weight")), xn, dqkv)
        ops.channel_sums(dqkv, GP("mid_attn.qkv.bias"))
        dxn = self.qkv.dgrad(dqkv)
        # plain GroupNorm backward in the fixed-order (bit-reproducible) form of the ResNetBlocks: dyh = dL/dy (no activation), its
        # per-channel (sum dyh, sum dyh * x) as one "tile" per sample, then the shared finalize + apply kernels (no float atomics)
        dxn.gnb_partials = ops.channel_dot_sums(dxn, x)
        dx, _ = ops.gn_bwd_fused(x, None, G, st, P("mid_attn.norm.weight"), dxn, GP("mid_attn.norm.weight"), GP("mid_attn.norm.bias"),
                                 add1=dout)
        return dx


class GradBuckets:
    """Data-parallel gradient averaging overlapped with the backward pass: the flat gradient vector is
    completed from its end towards its start (CUNet parameter order = forward order), so each finished slice (CUNet.bucket_bounds:
    up path | mid blocks | deepest down block | the rest) is all-reduced over RCCL on a communication stream while the remaining
    levels - the expensive level-0/1 dgrad and wgrad kernels - are still running.  `finish()` makes the main stream wait for all
    of them (before the global-norm clip, which must see the averaged gradient on every rank)."""

    def __init__(self, device, world, group=None):
        import torch.distributed as dist
        self.dist, self.world, self.group = dist, world, group
        self.comm = torch.cuda.Stream(device=device)
        self.works = []
        self.avg = dist.get_backend(group) == "nccl"          # RCCL averages in the collective; gloo sums (then one scale pass)
        self.issued = []

    def ready(self, gflat, lo, hi, side_streams):
        """The slice [lo, hi) of gflat is final once everything issued so far on the main stream and the side streams has run."""
        main = torch.cuda.current_stream()
        self.comm.wait_stream(main)
        for st in side_streams:
            if st is not None:
                self.comm.wait_stream(st)
        sl = gflat[lo:hi]
        with torch.cuda.stream(self.comm):
            op = self.dist.ReduceOp.AVG if self.avg else self.dist.ReduceOp.SUM
            self.works.append(self.dist.all_reduce(sl, op=op, group=self.group, async_op=True))
            if not self.avg:
                self.works[-1].wait()
                sl.div_(self.world)
        sl.record_stream(self.comm)
        self.issued.append((lo, hi))

    def finish(self):
        for w in self.works:
            w.wait()                                            # (stream-level wait for the collective, not a host block)
        torch.cuda.current_stream().wait_stream(self.comm)
        self.works = []
        done, self.issued = self.issued, []
        return done


class HipUNet:
    def __init__(self, net):
        self.net = net
        self.buckets = None               # GradBuckets when data-parallel training is on (enable_ddp)
        circ = net.conv_padding_mode == "circular"
        chs, L = net.chs, len(net.chs)
        self.cin0 = net.in_channels + net.s_conditioning_channels
        self.conv_in = Conv(self.cin0, chs[0], 3, circular=circ)
        self.conv_out = Conv(chs[0], net.in_channels, 3, circular=circ, out_f32=True)
        self.down = [Conv(chs[i], chs[i], 3, stride=2, circular=circ) for i in range(L - 1)]
        self.up = [Conv(chs[i + 1], chs[i], 3, upsample=1, circular=circ) for i in range(L - 1)]
        self.res = {b.name: _Res(net, b) for b in net.blocks}
        self.attn = _Attn(net) if net.mid_attn else None
        self._packed_key = None
        self._pack_plan = None
        self.saved = None
        self._ss = None

    def _all_convs(self):
        L = len(self.net.chs)
        out = [(self.conv_in, "conv_in.weight"), (self.conv_out, "conv_out.weight")]
        for i in range(L - 1):
            out.append((self.down[i], f"downs.{i}.down.weight"))
            out.append((self.up[i], f"ups.{i}.up.weight"))
        for r in self.res.values():
            out.extend(r.convs())
        if self.attn is not None:
            out.extend(self.attn.convs())
        return out

    def pack_weights(self, flat, dtype, need_dgrad, overlap=False):
        """Re-pack master fp32 weights into MFMA fragment order when the parameters changed (one launch on the current stream, ordered
        behind the optimizer step that changed them).  `overlap` is accepted and ignored: running the launch on a stream of its own next
        to the head of the next step bought 0.05 ms and cost a cross-stream use-after-free and a broken capture (rounds 3-4); deleted."""
        # flat._version alone is NOT enough: fused optimizers (torch._fused_adamw_) update the parameters without bumping the
        # version counter.  net.weights_epoch is bumped by every backward pass of this executor (an optimizer step follows) and by
        # CUNet.mark_weights_dirty() (the optimizer post-step hook LightVDM.configure_optimizers installs).
        key = (flat.data_ptr(), flat._version, self.net.weights_epoch, dtype, bool(need_dgrad))
        if self._packed_key == key:
            return
        if self._packed_key is not None and self._packed_key[:4] == key[:4] and not need_dgrad:
            return                                   # fwd buffers already current
        pkey = (flat.data_ptr(), dtype, bool(need_dgrad))
        if self._pack_plan is None or self._pack_plan[0] != pkey:         # one launch for all ~120 (conv, form) packings
            self._pack_plan = (pkey, ops.PackPlan([(conv, self.net.view(name, flat)) for conv, name in self._all_convs()], dtype, need_dgrad))
        self._pack_plan[1].run()
        self._packed_key = key

    def enable_ddp(self, world, group=None):
        """Average the gradient over `world` ranks inside backward(), bucket by bucket (see GradBuckets); CUNet.grad_synced tells
        the trainer that the flat gradient it receives is already averaged."""
        self.buckets = GradBuckets(self.net.flat.device, world, group) if world > 1 or os.environ.get("VDM4CDM_FORCE_BUCKETS") else None

    def _bucket_ready(self, gflat, after, ss):
        if self.buckets is None:
            return
        for lo, hi, when in self.net.bucket_bounds():
            if when == after:
                self.buckets.ready(gflat, lo, hi, [ss.side if ss.enabled else None, ss.second.side if (ss.second and ss.second.enabled) else None])

    def _side_stream(self, device):
        if self._ss is None or (self._ss.enabled and self._ss.side.device != device):
            self._ss = SideStream(device, enabled=os.environ.get("VDM4CDM_WGRAD_STREAM", "1") != "0")
        return self._ss

    # ---------------------------------------------------------------------------------------
    def forward(self, flat, table, z, s_cond, train, seed, dropout_p=None, packed=None):
        """z, s_cond: fp32 [N, D, H, W] (single channel).  Returns eps_hat fp32 [N, D, H, W].
        packed: conv_in's input {z, s_cond, 0...} already in NDHWC / compute dtype (the fused head of the training step wrote it).
        train: save the activations the backward pass needs (autograd is recording).  dropout_p: dropout probability of this
        call (None: net.dropout_prob in training mode, 0 in eval mode - the nn.Dropout / F.dropout(training=self.training)
        semantics of the reference stack, independent of whether autograd records)."""
        net = self.net
        p = (net.dropout_prob if net.training else 0.0) if dropout_p is None else float(dropout_p)
        dtype = torch.bfloat16 if net.precision == "bf16" else torch.float32
        self.pack_weights(flat, dtype, train)
        P = lambda name: net.view(name, flat)
        L = len(net.chs)
        ss = self._side_stream(flat.device)
        if packed is not None:
            assert packed.dtype == dtype and tuple(packed.shape) == tuple(z.shape) + (ops.cpad(2, dtype),) and packed.is_contiguous(), \
                "packed conv_in input does not match (z, compute dtype)"
            xin = packed
        else:
            xin = ops.pack_input(z, s_cond, dtype)
        h = self.conv_in.fwd(xin, P("conv_in.bias"), gn=FUSED_GN)
        skips = []
        for i in range(L):
            h = self.res[f"downs.{i}.block"].fwd(P, h, None, table, train, p, seed + 2 * i, ss)
            if i != L - 1:
                skips.append(h)
                h = self.down[i].fwd(h, P(f"downs.{i}.down.bias"), gn=FUSED_GN)
        for j in range(2):
            h = self.res[f"mid.{j}"].fwd(P, h, None, table, train, p, seed + 100 + j, ss)
            if j == 0 and self.attn is not None:
                h = self.attn.fwd(P, h, train)
        coarse = []
        for i in reversed(range(L - 1)):
            coarse.append(h)
            u = self.up[i].fwd(h, P(f"ups.{i}.up.bias"), gn=FUSED_GN)
            h = self.res[f"ups.{i}.block"].fwd(P, u, skips[i], table, train, p, seed + 200 + i, ss)
        st = ops.gn_stats(h, None, net.norm_groups)
        a = ops.gn_silu_fwd(h, None, net.norm_groups, st, P("norm_out.weight"), P("norm_out.bias"))
        eps = self.conv_out.fwd(a, P("conv_out.bias"))
        if train:
            self.saved = (flat, xin, skips, coarse[::-1], h, st, a, table.shape)
        return eps.view(z.shape)

    def backward(self, d_eps, cond=None):
        """d_eps: fp32 [N, D, H, W].  Returns (grad of flat, grad of conditioning table).  cond: the hip_ops.CondTable whose
        forward produced the table from this flat vector (its backward fills the conditioning parameters' gradients)."""
        net = self.net
        flat, xin, skips, coarse, h_last, st, a, tshape = self.saved
        self.saved = None
        L = len(net.chs)
        dtype = xin.dtype
        gflat = torch.zeros_like(flat)
        dtable = torch.zeros(tshape, dtype=torch.float32, device=flat.device)
        P = lambda name: net.view(name, flat)
        GP = lambda name: net.view(name, gflat)

        ss = self._side_stream(flat.device)
        ss.run(lambda: None, gflat)                   # (orders the side stream after the zero-fill of gflat)
        defer = DeferredSide(ss) if (DEFER_WGRAD_LEVELS > 0 and ss.enabled and L > 1) else None
        ss0 = defer if defer is not None else ss      # stream of the level-0 tail (conv_out) and up block
        dpad = ops.pack_input(d_eps.contiguous(), None, dtype)
        ss0.run(lambda a=a, dpad=dpad: self.conv_out.wgrad(a, dpad, GP("conv_out.weight")), a, dpad)
        GP("conv_out.bias").copy_(d_eps.sum().reshape(1))
        if FUSED_GNB and self.conv_out.gn_fold_ok(net.chs[0], 0, dtype):
            dyh = self.conv_out.dgrad_gn(dpad, h_last, None, net.norm_groups, st, P("norm_out.weight"), P("norm_out.bias"))
            dh, _ = ops.gn_bwd_fused(h_last, None, net.norm_groups, st, P("norm_out.weight"), dyh, GP("norm_out.weight"), GP("norm_out.bias"))
        else:
            da = self.conv_out.dgrad(dpad)
            dh, _ = ops.gn_silu_bwd(h_last, None, net.norm_groups, st, P("norm_out.weight"), P("norm_out.bias"), da,
                                    GP("norm_out.weight"), GP("norm_out.bias"), dx1=da)
        dskips = [None] * (L - 1)
        for i in range(L - 1):
            ssi = defer if (defer is not None and i < DEFER_WGRAD_LEVELS) else ss
            du, dskips[i] = self.res[f"ups.{i}.block"].bwd(P, GP, dh, dtable, ssi)
            ssi.run(lambda i=i, du=du: self.up[i].wgrad(coarse[i], du, GP(f"ups.{i}.up.weight"), GP(f"ups.{i}.up.bias")), coarse[i], du)
            dh = self.up[i].dgrad(du)              # gradient w.r.t. the coarse source (per-parity-class conv, no pooling pass)
            del du
            if defer is not None and i == min(DEFER_WGRAD_LEVELS, L - 1) - 1:
                defer.flush()                      # the main stream is entering the deeper levels: the held-back weight gradients start now
        self._bucket_ready(gflat, "ups", ss)       # norm_out / conv_out / every up block and up conv: final (RCCL starts on them)
        for j in reversed(range(2)):
            dh, _ = self.res[f"mid.{j}"].bwd(P, GP, dh, dtable, ss)
            if j == 1 and self.attn is not None:
                dh = self.attn.bwd(P, GP, dh, ss)
        self._bucket_ready(gflat, "mid", ss)
        for i in reversed(range(L)):
            if i == L - 2:
                self._bucket_ready(gflat, "downs.top", ss)
            if i != L - 1:
                ss.run(lambda i=i, dh=dh: self.down[i].wgrad(skips[i], dh, GP(f"downs.{i}.down.weight"), GP(f"downs.{i}.down.bias")),
                       skips[i], dh)
                dh = self.down[i].dgrad(dh, residual=dskips[i])      # per-parity-class conv: no zero-dilated intermediate
            tail = None
            if i == 0 and FUSED_GNB and self.res["downs.0.block"].skip1 is None and ops.gn_tail_ok(self.conv_in, self.res["downs.0.block"].saved[0]):
                tail = (self.conv_in, xin, GP("conv_in.weight"), GP("conv_in.bias"))      # the last apply pass + conv_in's weight gradient: one pass
            dh, _ = self.res[f"downs.{i}.block"].bwd(P, GP, dh, dtable, ss, tail=tail)
        if dh is not None:
            ss.run(lambda: self.conv_in.wgrad(xin, dh, GP("conv_in.weight"), GP("conv_in.bias")), xin, dh)
        # (the K6 backward only needs dtable - complete since the last block - and writes its own slice of gflat: it runs on the main
        # stream WHILE the side stream finishes the conv_in weight gradient, instead of behind the join)
        if cond is not None:      # K6 backward: conditioning MLPs + projections + the conv1 biases (column sums of dtable), 3 launches
            grads = [{k: sp[k] for k in ("w1", "b1", "w2", "b2", "wproj")} for sp in net.cond_specs(None, [None] * len(net.v_conditioning_dims), gflat)]
            cond.backward(dtable, grads, dbias=net.conv1_bias_all(gflat))
        else:                     # conv1 biases: column sums of the conditioning-table gradient (same additive broadcast)
            net.conv1_bias_all(gflat).copy_(dtable.sum(0))
        ss.join()
        self._bucket_ready(gflat, "end", ss)
        if self.buckets is not None:
            done = self.buckets.finish()
            assert sum(hi - lo for lo, hi in done) == gflat.numel(), "gradient buckets do not cover the parameter vector"
            net.grad_synced = True
        self.net.weights_epoch += 1                  # the caller is about to change the parameters: re-pack at the next forward
        return gflat, dtable


class _HipUNetFn(torch.autograd.Function):
    """eps_hat = CUNet(z, s_cond; t, v).  With `table` = None the conditioning table comes from the K6 kernel inside (and its
    gradient flows back into the flat parameter vector in backward); a caller-supplied table (the sampler's per-step rows) is used
    as is."""

    @staticmethod
    def forward(ctx, flat, table, z, s_cond, ex, train, seed, t, packed, *vs):
        ctx.ex, ctx.train, ctx.cond, ctx.table_grad, ctx.nvs = ex, train, None, False, len(vs)
        net = ex.net
        with torch.no_grad():
            fl = flat.detach()
            if table is None:
                specs = net.cond_specs(t, list(vs), fl)
                if specs:
                    cond = ops.CondTable(specs, z.shape[0], net.table_width)
                    tab = cond.forward(save=train)
                    ctx.cond = cond if train else None
                else:
                    tab = torch.zeros(z.shape[0], net.table_width, device=z.device)
            else:
                tab = table.detach().contiguous()
                ctx.table_grad = table.requires_grad
            return ex.forward(fl, tab, z, s_cond, train, seed, packed=packed)

    @staticmethod
    def backward(ctx, d_eps):
        if not ctx.train:
            raise RuntimeError("HIP CUNet: backward requested but the forward ran without saving activations")
        gflat, dtable = ctx.ex.backward(d_eps, ctx.cond)
        return (gflat, dtable if ctx.table_grad else None, None, None, None, None, None, None, None) + (None,) * ctx.nvs


_seed_counter = [0]


def _dist_rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def hip_unet_apply(net, x, s_conditioning, table=None, t=None, v_conditionings=None, packed=None):
    """x: [B, 1, D, H, W] fp32 on the GPU (NCDHW API; C == 1 so NDHWC is the same memory).
    table: optional precomputed conditioning table [B, table_width] (the sampler); otherwise t ([B]) / v_conditionings feed K6."""
    if net._exec is None:
        net._exec = HipUNet(net)
    B = x.shape[0]
    z = x.to(torch.float32).reshape(B, *x.shape[2:]).contiguous()
    s = None
    if net.s_conditioning_channels:
        assert s_conditioning is not None, "s_conditioning_channels=1 needs s_conditioning"
        s = s_conditioning.to(torch.float32).reshape(B, *x.shape[2:]).contiguous()
        if s.shape[0] != B:
            s = s.expand(B, *s.shape[1:]).contiguous()
    vs = []
    if table is None:
        if t is not None:
            t = t.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
            assert t.numel() == B
        for v in (v_conditionings or []):
            v = v.to(device=x.device, dtype=torch.float32).contiguous()
            if v.shape[0] != B:
                v = v.expand(B, *v.shape[1:]).contiguous()
            vs.append(v)
    # `train` here means "autograd records: keep the activations for backward"; the dropout probability follows net.training
    # (HipUNet.forward), exactly like nn.Dropout in the reference stack.
    train = torch.is_grad_enabled() and (net.flat.requires_grad or (table is not None and table.requires_grad))
    _seed_counter[0] += 1000
    # per-rank dropout masks under data parallelism: every rank calls seed_everything(42), so fold the rank in
    seed = (torch.initial_seed() + _seed_counter[0] + 0x9E3779B97F4A7C15 * _dist_rank()) & 0x7fffffffffffffff
    eps = _HipUNetFn.apply(net.flat, table, z, s, net._exec, train, seed, t, packed, *vs)
    return eps.view(x.shape)
