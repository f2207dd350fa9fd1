"""Drives every HOST-ONLY entry point of the C-ABI through the AddressSanitizer + UBSan build of the library's host code
(make -C vdm4cdm_amd/csrc asan -> libvdm4cdm_hip_asan.so: `hipcc -Xarch_host -fsanitize=address,undefined`: host code instrumented, never
loaded by the product, never run on the GPU box).  Started by tests/test_cpu.py::test_host_side_under_asan_ubsan with the ASan
runtime preloaded; any sanitizer report aborts the process (non-zero exit).

Covered: descriptor validation incl. hostile values, the planning helpers (packed sizes, pack plans, tile counts, kernel
variants, workspace / scratch sizes), the K6 size helpers, and the argument-error paths of the launching entry points (they
return VDM_ERR_ARG before any HIP call)."""
import ctypes as C
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VDM4CDM_LIB"] = os.path.join(ROOT, "vdm4cdm_amd", "libvdm4cdm_hip_asan.so")
from vdm4cdm_amd import _lib  # noqa: E402
from vdm4cdm_amd._lib import CondMlp, ConvDesc, GnFold, PackItem  # noqa: E402

L = _lib.lib()
assert L.vdm_abi_version() == _lib.ABI_VERSION
calls = 0


def desc(**kw):
    base = dict(n=2, od=16, oh=16, ow=16, cin=32, cout=32, ksize=3, stride=1, upsample=0, pad_mode=0, dtype=1, out_f32=0)
    base.update(kw)
    return ConvDesc(**base)


def plan_everything(d):
    """every host-only helper on one descriptor; the results only have to be consistent with each other"""
    global calls
    for mode in (0, 1):
        nbytes = L.vdm_conv_packed_bytes(d, mode)
        it = PackItem()
        buf = (C.c_char * 64)()
        st = L.vdm_conv_pack_plan(d, mode, C.addressof(buf), C.addressof(buf), C.byref(it))
        if nbytes:
            assert st == 0 and it.elems * (4 if d.dtype == 0 else 2) == nbytes, (nbytes, it.elems)
        else:
            assert st != 0 and L.vdm_last_error()
        L.vdm_conv_kernel_variant(d, mode)
        calls += 3
    t1, t2 = L.vdm_conv_gn_tiles(d), L.vdm_conv_dgrad_gn_tiles(d)
    ws = L.vdm_conv_wgrad_workspace_bytes(d)
    assert t1 >= 0 and t2 >= 0 and ws >= 0
    calls += 3


# 1. the shapes of the network family, every (ksize, stride, upsample, dtype, pad) the product uses
for D, (cin, cout), (ks, st, up), dt, pm, n in itertools.product(
        (2, 8, 16, 24, 128, 192, 256), ((1, 32), (2, 32), (32, 1), (16, 16), (32, 32), (64, 32), (96, 48), (256, 256), (128, 384)),
        ((3, 1, 0), (3, 2, 0), (3, 1, 1), (1, 1, 0)), (0, 1), (0, 1), (1, 2)):
    plan_everything(desc(n=n, od=D, oh=D, ow=D, cin=cin, cout=cout, ksize=ks, stride=st, upsample=up, dtype=dt, pad_mode=pm,
                         out_f32=1 if (cout == 1 and ks == 3 and st == 1 and not up) else 0))
# ragged / non-cubic grids
for dims in ((1, 1, 1), (3, 5, 7), (17, 4, 33), (1, 128, 2), (130, 6, 18)):
    plan_everything(desc(od=dims[0], oh=dims[1], ow=dims[2]))

# 2. hostile descriptors: every one must be REJECTED (0 bytes / error status), never crash or overflow
INT_MAX, INT_MIN = 2 ** 31 - 1, -2 ** 31
hostile = [dict(n=0), dict(n=-1), dict(od=0), dict(oh=-5), dict(ow=INT_MIN), dict(cin=0), dict(cout=-3), dict(ksize=0), dict(ksize=2),
           dict(ksize=5), dict(ksize=INT_MAX), dict(stride=0), dict(stride=3), dict(stride=2, ksize=1), dict(upsample=1, stride=2),
           dict(upsample=1, ksize=1), dict(upsample=1, od=3), dict(dtype=2), dict(dtype=-1), dict(pad_mode=7),
           dict(od=4095, oh=4095, ow=4095), dict(od=INT_MAX, oh=INT_MAX, ow=INT_MAX), dict(od=2048, oh=2048, ow=2048, cin=512),
           dict(od=4096), dict(stride=2, od=2048)]
for h in hostile:
    d = desc(**h)
    assert L.vdm_conv_packed_bytes(d, 0) == 0 and L.vdm_conv_packed_bytes(d, 1) == 0, h
    assert L.vdm_last_error(), h
    assert L.vdm_conv_gn_tiles(d) == 0 and L.vdm_conv_dgrad_gn_tiles(d) == 0 and L.vdm_conv_wgrad_workspace_bytes(d) == 0, h
    assert L.vdm_conv_kernel_variant(d, 0) == -1, h
    it = PackItem()
    assert L.vdm_conv_pack_plan(d, 0, 1, 1, C.byref(it)) != 0, h
    assert L.vdm_conv_fwd(d, 1, 1, None, None, 0, None, 1, None, None) != 0, h
    calls += 9
# large but legal channel counts / values close to the 32-bit in-sample index limit
plan_everything(desc(od=1024, oh=1024, ow=1024, cin=2, cout=2))
plan_everything(desc(cin=4096, cout=4096, od=4, oh=4, ow=4))
assert L.vdm_conv_packed_bytes(None, 0) == 0 and L.vdm_conv_gn_tiles(None) == 0 and L.vdm_conv_kernel_variant(None, 1) == -1

# 3. launching entry points: NULL / inconsistent arguments return before any HIP call
d = desc()
assert L.vdm_conv_fwd(d, None, None, None, None, 0, None, None, None, None) == -1
assert L.vdm_conv_dgrad(d, None, None, None, None, None) == -1
assert L.vdm_conv_wgrad(d, None, None, None, None, 0, None, 0, None) == -1
assert L.vdm_conv_pack_weights(d, 0, None, None, None) == -1 and L.vdm_conv_pack_weights(d, 9, 1, 1, None) == -1
assert L.vdm_conv_pack_many(None, None, 0, 1, None) == -1
f = GnFold()
assert L.vdm_conv_dgrad_gn(d, 1, 1, 1, C.byref(f), None) == -1                    # NULL pointers inside the fold
f = GnFold(x1=1, c1=16, c2=8, groups=8, stats=1, gamma=1, beta=1, partials=1, eps=1e-5, inv_keep=1.0)
assert L.vdm_conv_dgrad_gn(d, 1, 1, 1, C.byref(f), None) == -1                    # c1 + c2 != cin
assert L.vdm_conv_dgrad_gn(desc(stride=2), 1, 1, 1, C.byref(f), None) == -1
assert L.vdm_gn_stats(None, 32, None, 0, 2, 4096, 8, 1, None, None, None, 0, None, 0, None, None) == -1
assert L.vdm_gn_stats(1, 30, None, 0, 2, 4096, 8, 1, 1, 1, None, 0, None, 0, None, None) == -1      # channels not a multiple of a piece
assert L.vdm_gn_silu_fwd(1, 32, None, 0, 2, 4096, 8, 1, 1, 1, 1, 1e-5, 1.5, 0, 1, None, 0, None, None) == -1   # dropout_p out of range
assert L.vdm_gn_bwd_finalize(1, 0, 2, 32, 8, 4096, 1, 1, 1e-5, None, 1, 1, None, 0, None) == -1
assert L.vdm_gn_bwd_finalize(1, 4, 2, 32, 8, 4096, 1, 1, 1e-5, None, 1, 1, 1, 32, None) == -1            # colsum without chsum
assert L.vdm_pack_input(None, None, 10, 8, 1, None, None) == -1
assert L.vdm_diffuse(1, 1, 1, 1, 2, 7, 1, None) == -1                                 # per % 4 != 0
assert L.vdm_sumsq(3, 100, 1, 1, None) == -1                                          # misaligned
assert L.vdm_channel_sums(1, 4, 7, 1, 1, None) == -1 and L.vdm_attn_fwd(1, 1, 1, 1, 30, 2, 16, 1, 1.0, 1, None, None) == -1      # voxels % 4
assert L.vdm_augment_batch(None, 0, 16, 8, None, 0, None) == -1
calls += 20

# 4. K6 size helpers and descriptor checks
mlps = (CondMlp * 2)()
mlps[0].in_dim, mlps[0].dim = 64, 128
mlps[1].in_dim, mlps[1].dim = 6, 64
assert L.vdm_cond_saved_floats(mlps, 2, 3) == 3 * (64 + 3 * 128 + 6 + 3 * 64)
assert L.vdm_cond_saved_floats(None, 2, 3) == 0
nslab = (1312 + 31) // 32
assert L.vdm_cond_bwd_scratch_floats(mlps, 2, 2, 1312) == 2 * (2 + nslab) * (128 + 64)
assert L.vdm_cond_bwd_scratch_floats(mlps, 2, 0, 1312) == 0 and L.vdm_cond_bwd_scratch_floats(None, 2, 2, 1312) == 0
assert L.vdm_cond_table_fwd(mlps, 2, 2, 1312, 1, None, None) == -1                    # NULL pointers in the descriptors
assert L.vdm_cond_table_fwd(mlps, 9, 2, 1312, 1, None, None) == -1                    # more than 4 conditionings
for k in range(2):
    for fld in ("input", "w1", "b1", "w2", "b2", "wproj"):
        setattr(mlps[k], fld, 16)
mlps[0].dim = 1000
assert L.vdm_cond_table_fwd(mlps, 2, 2, 1312, 1, None, None) == -1 and b"out of range" in L.vdm_last_error()
mlps[0].dim = 128
mlps[0].w2 = 20                                                                       # rows of 128 floats but not 16-byte aligned
assert L.vdm_cond_table_fwd(mlps, 2, 2, 1312, 1, None, None) == -1 and b"aligned" in L.vdm_last_error()
mlps[0].w2 = 16
assert L.vdm_cond_table_bwd(mlps, 2, 2, 1312, 1, 1312, 1, 1, None, None) == -1       # NULL gradient pointers
assert L.vdm_cond_table_step(None, None, None, 2, 8, None, None) == -1
calls += 14
# 5. skip conv folded into the GroupNorm passes: shape table, workspace size, argument checks
for c1, c2, cout, want in [(32, 32, 32, 3), (32, 0, 64, 3), (64, 64, 64, 3), (64, 0, 128, 1), (128, 128, 128, 0), (16, 16, 16, 3), (8, 0, 16, 3),
                           (12, 0, 16, 0), (32, 0, 24, 0), (0, 0, 16, 0), (-8, 8, 16, 0), (32, -8, 16, 0), (2 ** 30, 2 ** 30, 16, 0)]:
    assert L.vdm_gn_skip_supported(c1, c2, cout, 1) == want, (c1, c2, cout)
    assert L.vdm_gn_skip_supported(c1, c2, cout, 0) == 0
    calls += 2
assert L.vdm_gn_skip_ws_floats(32, 32, 32, 2, 128 ** 3) == 512 * 32 * 64          # 256 workgroups per sample x [cout][64]
assert L.vdm_gn_skip_ws_floats(32, 0, 64, 1, 100) == 1 * 64 * 32                   # 4 chunks of 32 voxels: one workgroup
assert L.vdm_gn_skip_ws_floats(32, 0, 64, 0, 100) == 0 and L.vdm_gn_skip_ws_floats(32, 0, 64, 1, -5) == 0 and L.vdm_gn_skip_ws_floats(12, 0, 64, 1, 8) == 0
assert L.vdm_gn_silu_skip_fwd(16, 32, 16, 32, 2, 512, 8, 0, 16, 16, 16, 1e-5, 16, 16, 16, 32, 16, 16, None) == -1 and b"bf16" in L.vdm_last_error()
assert L.vdm_gn_silu_skip_fwd(16, 32, None, 32, 2, 512, 8, 1, 16, 16, 16, 1e-5, 16, 16, 16, 32, 16, 16, None) == -1     # x2 NULL with c2 > 0
assert L.vdm_gn_silu_skip_fwd(16, 128, 16, 128, 2, 512, 8, 1, 16, 16, 16, 1e-5, 16, 16, 16, 128, 16, 16, None) == -1 and b"not supported" in L.vdm_last_error()
assert L.vdm_gn_silu_skip_fwd(16, 32, 16, 32, 2, 512, 7, 1, 16, 16, 16, 1e-5, 16, 16, 16, 32, 16, 16, None) == -1       # 64 channels, 7 groups
assert L.vdm_gn_bwd_apply_skip(16, 128, 16, 128, 1, 512, 8, 1, 16, 16, 1e-5, 16, 16, 16, 16, 16, 16, 128, 16, 16, 16, 16, 16, 16, 16, 1 << 30,
                               None) == -1 and b"not supported" in L.vdm_last_error()                                   # too wide for the kernels
assert L.vdm_gn_bwd_apply_skip(16, 32, 16, 32, 1, 512, 8, 1, 16, 16, 1e-5, 16, 16, 16, 16, 16, 16, 32, 16, 16, 16, 16, 16, 16, 16, 10,
                               None) == -1 and b"workspace" in L.vdm_last_error()
assert L.vdm_gn_bwd_apply_wgrad_thin(16, 48, 2, 8, 8, 16, 8, 16, 16, 1e-5, 16, 16, 16, None, 16, 2, 0, 16, 16, 16, 16, 16, 1 << 30, None) == -1   # 48 channels
assert L.vdm_gn_bwd_apply_wgrad_thin(16, 32, 99, 8, 8, 16, 8, 16, 16, 1e-5, 16, 16, 16, None, 16, 2, 0, 16, 16, 16, 16, 16, 1 << 30, None) == -1  # batch > 16
assert L.vdm_gn_bwd_apply_wgrad_thin(16, 32, 2, 8, 8, 16, 8, 16, 16, 1e-5, 16, 16, 16, None, 16, 2, 1, 16, 16, 16, 16, 16, 1 << 30, None) == -1   # circular, 16 wide
assert L.vdm_gn_bwd_apply_wgrad_thin(16, 32, 2, 8, 8, 32, 8, 16, 16, 1e-5, 16, 16, 16, None, 16, 2, 0, 16, 16, 16, 16, 16, 64, None) == -1 and b"workspace" in L.vdm_last_error()
assert L.vdm_gn_bwd_apply_wgrad_thin(None, 32, 2, 8, 8, 32, 8, 16, 16, 1e-5, 16, 16, 16, None, 16, 2, 0, 16, 16, 16, 16, 16, 1 << 30, None) == -1
calls += 15
name = (C.c_char * 256)()
cus, lds = C.c_int(0), C.c_int(0)
L.vdm_device_info(0, C.byref(cus), C.byref(lds), name)                                 # no GPU here: must fail cleanly
print(f"asan host driver ok: {calls} calls")
