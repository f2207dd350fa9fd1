/* vdm4cdm_hip.h - C-ABI of libvdm4cdm_hip.so: the MI355X (gfx950) kernels of the vdm4cdm
 * variational-diffusion denoising hot path.
 *
 * The reference (cfpark00/vdm4cdm) has no FFI layer: its hot path is the Python API between its
 * scripts and the third-party `mltools` package, which bottoms out in ATen ops.  Each entry point
 * below replaces the ATen op family that one reference call reaches (SURVEY.md section 8a rows
 * R1-R12); the reference-side call that would bind it is cited per function as
 * [REF file:line] (files under /root/reference) or [NB ...] (notebook traceback fragment,
 * SURVEY.md section 3.2).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*.
 *  - the caller owns every buffer (they are torch tensors' data_ptr()s); the library never
 *    allocates device memory and never synchronises: all work is enqueued on `stream`
 *    (a hipStream_t passed as void*; NULL = the null stream).
 *  - activations are NDHWC: x[n][z][y][x][c], dtype VDM_F32 or VDM_BF16 (bf16 = storage only,
 *    arithmetic and accumulation are fp32).  Channel counts of conv inputs must be a multiple of
 *    16 bytes (4 fp32 / 8 bf16 elements); the host zero-pads (e.g. conv_in's 2 channels).
 *  - every function returns VDM_OK (0) or a negative vdm_status; vdm_last_error() returns a
 *    thread-local message.  No exceptions cross the boundary; HIP errors are translated.
 */
#ifndef VDM4CDM_HIP_H
#define VDM4CDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDM_ABI_VERSION 11

typedef enum { VDM_OK = 0, VDM_ERR_ARG = -1, VDM_ERR_HIP = -2, VDM_ERR_UNSUPPORTED = -3 } vdm_status;
typedef enum { VDM_F32 = 0, VDM_BF16 = 1 } vdm_dtype;
typedef enum { VDM_PAD_ZEROS = 0, VDM_PAD_CIRCULAR = 1 } vdm_pad_mode; /* [REF trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:125] conv_padding_mode */
typedef enum { VDM_PACK_FWD = 0, VDM_PACK_DGRAD = 1 } vdm_pack_mode;

/* One 3D convolution (k in {1,3}, stride in {1,2}, optional fused nearest x2 up-sampling of the
 * input).  od/oh/ow are the OUTPUT spatial dims.  Input dims: stride 1: the same; stride 2: 2x
 * (k=3, pad 1); upsample: the source tensor is (od/2, oh/2, ow/2) and is read at (z>>1,y>>1,x>>1). */
typedef struct {
    int32_t n, od, oh, ow;
    int32_t cin, cout;
    int32_t ksize;    /* 1 or 3 */
    int32_t stride;   /* 1 or 2 */
    int32_t upsample; /* 0 or 1 (only with stride 1, ksize 3) */
    int32_t pad_mode; /* vdm_pad_mode */
    int32_t dtype;    /* vdm_dtype of x / residual / out */
    int32_t out_f32;  /* 1: write `out` as fp32 regardless of dtype (conv_out -> eps_hat) */
} vdm_conv_desc;

const char* vdm_last_error(void);
int vdm_abi_version(void);
/* fills cu_count / lds_bytes_per_cu / gcn arch name (buffer of >= 64 bytes) for `device`. */
int vdm_device_info(int device, int* cu_count, int* lds_bytes, char* arch_name);

/* ---- K1/K3/K4/K5: convolution (implicit GEMM on MFMA) -------------------------------------
 * Replaces torch.nn.functional.conv3d as reached from ResNetBlock.net1/net2, ResNetDown and the
 * up path [NB blocks.py:129-132,166-170; networks.py:259-265].
 * Master weights are fp32 [taps][cout][cin] (tap = (dz*3+dy)*3+dx).  They are re-packed into the
 * MFMA fragment order once per optimiser step. */
size_t vdm_conv_packed_bytes(const vdm_conv_desc* d, int pack_mode);
int vdm_conv_pack_weights(const vdm_conv_desc* d, int pack_mode, const float* w_master, void* w_packed, void* stream);
/* All packings of a network in one launch (they are redone after every optimiser step).  vdm_conv_pack_plan fills one work item
 * per (conv, form) on the host; the caller cuts [0, item.elems) of every item into chunks of <= VDM_PACK_CHUNK elements, uploads the
 * item and chunk arrays once, and calls vdm_conv_pack_many on them whenever the master weights changed.  All items of one call
 * share the dtype. */
#define VDM_PACK_CHUNK 16384
typedef struct vdm_pack_item {
    const float* w_master;
    void* w_packed;
    int32_t taps, cout, cin, nc, nchunks, nkb, dgrad, variant, cls_kind, dtype;
    int64_t elems; /* packed elements of this item */
} vdm_pack_item;
typedef struct vdm_pack_chunk {
    int32_t item;  /* index into the item array */
    int32_t count; /* elements in this chunk */
    int64_t first; /* first packed element (inside the item) */
} vdm_pack_chunk;
int vdm_conv_pack_plan(const vdm_conv_desc* d, int pack_mode, const float* w_master, void* w_packed, vdm_pack_item* item);
int vdm_conv_pack_many(const vdm_pack_item* items_device, const vdm_pack_chunk* chunks_device, int nchunks, int dtype, void* stream);
/* out = conv(x, w) + bias[c] + nbias[n*nbias_stride + c] + residual   (bias, nbias, residual may be NULL).
 * nbias is the per-sample conditioning bias table (sum_k Linear_k(cond_k)); nbias_stride is the
 * element distance between samples (the table of all blocks is one [n][sum cout] matrix). */
/* gn_partials (may be NULL): the epilogue also reduces the GroupNorm statistics of the output it stores, per spatial tile:
 * gn_partials[n][tile][cout][2] = (sum, sum of squares) fp32, tile < vdm_conv_gn_tiles(d); feed them to vdm_gn_stats instead of
 * a separate pass over the tensor (the up-sampling conv has one slot per coarse tile and parity class). */
int vdm_conv_gn_tiles(const vdm_conv_desc* d);
int vdm_conv_fwd(const vdm_conv_desc* d, const void* x, const void* w_packed_fwd, const float* bias,
                 const float* nbias, int64_t nbias_stride, const void* residual, void* out, float* gn_partials, void* stream);
/* The same forward conv for an input that is y = silu(groupnorm(x)) of the raw tensor x [NB blocks.py:129-132: net1 / net2 =
 * Sequential(GroupNorm, SiLU, (Dropout,) Conv)] in INFERENCE (no dropout, nobody else needs y): GroupNorm + SiLU are applied to the
 * staged halo image in LDS - vdm_gn_silu_fwd's arithmetic, bit-identical results - instead of by a pass of their own (one read and
 * one write of the tensor per GroupNorm).  stats [n][groups][2] from vdm_gn_stats; gamma / beta [cin].  bf16 3x3x3 stride-1 convs
 * on the generic kernel only: ask vdm_conv_fwd_gn_supported (host only) first. */
int vdm_conv_fwd_gn_supported(const vdm_conv_desc* d);
int vdm_conv_fwd_gn(const vdm_conv_desc* d, const void* x, const void* w_packed_fwd, const float* bias, const float* nbias,
                    int64_t nbias_stride, const void* residual, void* out, float* gn_partials, const float* stats,
                    const float* gamma, const float* beta, int groups, float eps, void* stream);
/* dx = gradient w.r.t. the conv INPUT: the descriptor is the FORWARD conv's; dout has cout channels and the output
 * dims (od,oh,ow); dx gets cin channels and the input dims (stride 1: same; stride 2: 2x; up-sampling conv: the
 * coarse source grid od/2...).  Stride-2 and up-sampling convs run as per-parity-class convs (no dilated /
 * up-sampled intermediate). */
int vdm_conv_dgrad(const vdm_conv_desc* d, const void* dout, const void* w_packed_dgrad, const void* residual, void* dx,
                   void* stream); /* dx = dgrad (+ residual, same shape as dx, may be NULL) */
/* The same input gradient for a conv whose INPUT was y = dropout(silu(groupnorm(x))) [NB blocks.py:129-132: net1 / net2 =
 * Sequential(GroupNorm, SiLU, (Dropout,) Conv)], with the first half of that GroupNorm's backward folded into the epilogue:
 * instead of dL/dy the kernel stores dyh = dL/dy * keep/(1-p) * silu'(yhat) (same shape and dtype) and reduces, per spatial tile
 * and channel, partials[n][tile][cin][2] = (sum dyh, sum dyh * x), tile < vdm_conv_dgrad_gn_tiles(d) - one read of x instead of
 * a separate two-tensor reduction pass, in a fixed order (no float atomics).  vdm_gn_bwd_finalize + vdm_gn_bwd_apply finish the
 * GroupNorm backward.  Only for ksize 3, stride 1, no up-sampling (every conv behind a GroupNorm on the path). */
typedef struct vdm_gn_fold {
    const void* x1;          /* GroupNorm input, first c1 channels  [n][od][oh][ow][c1] */
    const void* x2;          /* ... remaining c2 channels (skip concat), or NULL */
    int32_t c1, c2, groups;  /* c1 + c2 == cin of the conv */
    const float* stats;      /* [n][groups][2] raw moments of x (vdm_gn_stats) */
    const float* gamma;      /* [cin] */
    const float* beta;       /* [cin] */
    float eps;
    float inv_keep;          /* 1 / (1 - p) when keep_mask is given */
    const uint8_t* keep_mask; /* dropout keep bits written by vdm_gn_silu_fwd, or NULL (no dropout) */
    float* partials;         /* out */
} vdm_gn_fold;
int vdm_conv_dgrad_gn_tiles(const vdm_conv_desc* d);
int vdm_conv_dgrad_gn(const vdm_conv_desc* d, const void* dout, const void* w_packed_dgrad, void* dyh, const vdm_gn_fold* fold,
                      void* stream);
/* dw[taps][cout][cin] (fp32) = sum over voxels; workspace holds per-workgroup partial slabs.
 * dbias (optional, ksize 3 only): dbias[cout] = sum over samples and voxels of dout (the conv bias gradient), computed
 * from the dOut tiles the kernel stages anyway.  accumulate != 0 adds to dw / dbias instead of overwriting. */
/* Which kernel family vdm_conv_fwd (dgrad = 0) / vdm_conv_dgrad (dgrad = 1) launches for this descriptor (profiling keys,
 * tests): the generic implicit-GEMM kernel or the per-parity-class kernel (up-sampling conv and its gradients, stride-2
 * dgrad).  < 0: bad descriptor. */
#define VDM_CONV_VARIANT_GENERIC 0
#define VDM_CONV_VARIANT_CLASS 1
#define VDM_CONV_VARIANT_SPLIT 3 /* generic kernel, half-chunk workgroups (64-cout chunks on a small grid) */
#define VDM_CONV_VARIANT_KSPLIT 4 /* deepest level: the waves of a workgroup split the K-blocks, one weight fetch per workgroup */
#define VDM_CONV_VARIANT_KPACK 2 /* Input gradient (folded GroupNorm backward, as vdm_conv_dgrad_gn) AND weight / bias gradient (as vdm_conv_wgrad) of ONE 3x3x3 stride-1
 * conv in one launch that stages dout once (ABI v11; csrc/conv_dgw.hip) [replaces the backward of net1 / net2 of a ResNetBlock,
 * NB blocks.py:129-132, for the convs vdm_conv_dgw_supported() accepts: bf16, 32 -> 32 channels, large grids - level 0 of the 128^3
 * network].  act = the conv's saved input (the activated tensor a = drop(silu(gn(x)))); the fold's `partials` buffer holds
 * vdm_conv_dgw_tiles(d) tiles (2 x 8 x 16 voxel steps); workspace >= vdm_conv_dgw_workspace_bytes(d).  Results equal the two separate
 * entries up to fp32 summation order (the weight gradient sums the same products tile by tile in another order). */
int vdm_conv_dgw_supported(const vdm_conv_desc* d);
int vdm_conv_dgw_tiles(const vdm_conv_desc* d);
size_t vdm_conv_dgw_workspace_bytes(const vdm_conv_desc* d);
int vdm_conv_dgrad_gn_wgrad(const vdm_conv_desc* d, const void* dout, const void* w_packed_dgrad, const void* act, void* dyh,
                            const vdm_gn_fold* fold, float* dw, float* dbias, int accumulate, void* workspace, size_t workspace_bytes,
                            void* stream);

/* <= 8 reduction channels (conv_in, conv_out's input gradient): 4 taps per MFMA K-step */
int vdm_conv_kernel_variant(const vdm_conv_desc* d, int dgrad);

size_t vdm_conv_wgrad_workspace_bytes(const vdm_conv_desc* d);
int vdm_conv_wgrad(const vdm_conv_desc* d, const void* x, const void* dout, float* dw, float* dbias, int accumulate,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- K2: GroupNorm + SiLU (+ dropout) -----------------------------------------------------
 * Replaces torch.group_norm / F.silu / F.dropout [NB normalization.py:273 frame under blocks.py:130].
 * The input may be the channel-concatenation of two tensors (skip connection), never materialised.
 * stats[n][g] = {sum, sumsq} in fp32 (raw moments; mean/rstd are derived where consumed).
 * workspace: VDM_GN_STATS_WS_BYTES of scratch for the per-workgroup partials (two-stage, fixed-order
 * reduction: the forward pass is bit-reproducible). */
#define VDM_GN_STATS_WS_BYTES (2048 * 2 * 64 * 4)
/* part1 / part2 (may be NULL): per-tile channel partials of that source written by vdm_conv_fwd (tilesK tiles per sample);
 * the source's groups are then summed from them and xK is not read (xK may be NULL). */
/* chsum (may be NULL; needs every source given as partials): chsum[n][c1+c2] = sum_v x[n][v][c], the per-channel sums. */
int vdm_gn_stats(const void* x1, int c1, const void* x2, int c2, int n, int64_t voxels, int groups,
                 int dtype, float* stats, float* workspace, const float* part1, int tiles1, const float* part2, int tiles2,
                 float* chsum, void* stream);
/* y[n][v][c1+c2] = dropout(silu(gn(concat(x1,x2)))) ; keep-mask from Philox(seed, element index).
 * linear != 0: no activation - the plain GroupNorm in front of the attention block's qkv projection (mid_attn=True).
 * keep_mask (may be NULL): with dropout_p > 0 the keep bits are also written, one byte per 16-byte piece of y
 * ([n][voxels][(c1+c2) / (4 fp32 | 8 bf16)], bit j = channel j of the piece), for vdm_conv_dgrad_gn.
 * seed_step (may be NULL; also vdm_gn_dyh, vdm_randn, vdm_train_scalars): DEVICE step counter mixed into the seed inside the kernel
 * (seed + *seed_step * 0x9E3779B97F4A7C15) - a hipGraph of the training step bakes the host seed into its kernel arguments; the
 * counter, bumped once per replay (vdm_step_inc), keeps dropout masks, noise fields and the time grid fresh. */
int vdm_gn_silu_fwd(const void* x1, int c1, const void* x2, int c2, int n, int64_t voxels, int groups,
                    int dtype, const float* stats, const float* gamma, const float* beta, float eps,
                    float dropout_p, uint64_t seed, void* y, uint8_t* keep_mask, int linear, const int32_t* seed_step, void* stream);
/* Backward of the above for a GroupNorm whose gradient did not come out of vdm_conv_dgrad_gn, first of three steps:
 * dyh = dy * keep/(1-p) * silu'(yhat) (linear != 0: dyh = dy); dyh may alias dy.  Then vdm_channel_dot_sums(dyh, x) gives the
 * per-sample (sum dyh, sum dyh * x) = the one-tile-per-sample `partials` of vdm_gn_bwd_finalize, and vdm_gn_bwd_apply writes dx:
 * the same fixed-order (bit-reproducible) arithmetic as the folded path; no float atomics. */
int vdm_gn_dyh(const void* x1, int c1, const void* x2, int c2, int n, int64_t voxels, int groups, int dtype, const float* stats,
               const float* gamma, const float* beta, float eps, float dropout_p, uint64_t seed, const void* dy, void* dyh, int linear,
               const int32_t* seed_step, void* stream);

/* Second half of the GroupNorm backward after vdm_conv_dgrad_gn (all in fixed summation order: bit-reproducible).
 * finalize: chan[n][c][2] = sum over tiles of the partials; red[n][g][2] = sum_c gamma_c chan[n][c];
 *           colsum (may be NULL; needs chsum of vdm_gn_stats): colsum[n*colsum_stride + c] = sum_v dx[n][v][c], analytically
 *           (the conditioning-table / conv-bias gradient of the ResNetBlock: h = conv1(..) + bias + sum_k Linear_k(cond_k)).
 * apply:    dx = rstd * (gamma * dyh - m1 - xhat * m2) (+ add1 / add2: residual-path gradients), m = red / (voxels * C/groups);
 *           dgamma[c] = sum_n chan[n][c][1], dbeta[c] = sum_n chan[n][c][0] (written, not accumulated).  dx1 may alias dyh. */
int vdm_gn_bwd_finalize(const float* partials, int tiles, int n, int c, int groups, int64_t voxels, const float* stats,
                        const float* gamma, float eps, const float* chsum, float* red, float* chan, float* colsum,
                        int64_t colsum_stride, void* stream);
int vdm_gn_bwd_apply(const void* x1, int c1, const void* x2, int c2, int n, int64_t voxels, int groups, int dtype,
                     const float* stats, const float* gamma, float eps, const void* dyh, const float* red, const float* chan,
                     const void* add1, const void* add2, void* dx1, void* dx2, float* dgamma, float* dbeta, void* stream);

/* ---- ResNetBlock skip path folded into the GroupNorm passes (bf16 storage) ------------------------------------------------
 * [NB blocks.py ResNetBlock: out = net2(net1(x) + cond) + skip(x), skip = Conv3d(cin, cout, 1) when cin != cout; call chain
 * trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:116-127.]  norm1 and the 1x1x1 skip conv read the same block input, and the
 * skip conv's input gradient is added to the result of norm1's backward: one pass each instead of a conv launch of their own.
 *   vdm_gn_skip_supported: bit 0 = the forward kernel, bit 1 = the backward kernel exist for (c1 + c2 -> cout) in `dtype`
 *     (host only; 0 for fp32 storage and for wide layers, which keep vdm_conv_fwd / vdm_conv_dgrad / vdm_conv_wgrad with ksize 1).
 *   vdm_gn_silu_skip_fwd: y = silu(gn(concat(x1, x2)))  [n][voxels][c1+c2]  (vdm_gn_silu_fwd without dropout) AND
 *     skip_out[n][voxels][cout] = w1 x1 + w2 x2 + bias, w1 [cout][c1], w2 [cout][c2] = the fp32 MASTER weights of the two column
 *     blocks of the skip conv (rounded to bf16 in the kernel like vdm_conv_pack_weights does), bias [cout] or NULL.
 *   vdm_gn_bwd_apply_skip: vdm_gn_bwd_apply with add = W^T dout computed in the pass (dout [n][voxels][cout] = the gradient of the
 *     block output), plus the skip weight gradients dw1 [cout][c1], dw2 [cout][c2] = sum_{n,v} dout x (written, not accumulated;
 *     per-workgroup slabs in `workspace` (vdm_gn_skip_ws_floats floats) + a fixed-order reduce: bit-reproducible).  The skip bias
 *     gradient is the column sum of dout, which vdm_conv_wgrad of the block's second conv already returns. */
/* The LAST GroupNorm backward of the network (norm1 of the first block) and the weight gradient of conv_in in one pass (bf16): the
 * tensor vdm_gn_bwd_apply would write there - the gradient at the output of conv_in - has no reader but that weight gradient, so it is
 * formed in registers (dx = rstd (gamma dyh - m1 - xhat m2) + add, rounded to bf16 like the tensor it replaces) and fed straight into
 * dw[27][c][thin_c] = sum_v dx[v] (x) thin_x[v + tap], dbias[c] = sum_v dx[v]; dgamma / dbeta as vdm_gn_bwd_apply.  x / dyh / add:
 * [n][od][oh][ow][c] with c in {16, 32, 64}; thin_x: the conv_in input [n][od][oh][ow][8] (thin_c <= 2 real channels); workspace:
 * vdm_conv_wgrad_workspace_bytes of conv_in.  Replaces a 3-tensor pass + a write and a read of dx on the exposed tail of the
 * backward pass. */
int vdm_gn_bwd_apply_wgrad_thin(const void* x, int c, int n, int od, int oh, int ow, int groups, const float* stats, const float* gamma,
                                float eps, const void* dyh, const float* red, const float* chan, const void* add, const void* thin_x,
                                int thin_c, int circular, float* dgamma, float* dbeta, float* dw, float* dbias, void* workspace,
                                size_t workspace_bytes, void* stream);
int vdm_gn_skip_supported(int c1, int c2, int cout, int dtype);
size_t vdm_gn_skip_ws_floats(int c1, int c2, int cout, int n, int64_t voxels);
int vdm_gn_silu_skip_fwd(const void* x1, int c1, const void* x2, int c2, int n, int64_t voxels, int groups, int dtype,
                         const float* stats, const float* gamma, const float* beta, float eps, const float* w1, const float* w2,
                         const float* bias, int cout, void* y, void* skip_out, void* stream);
int vdm_gn_bwd_apply_skip(const void* x1, int c1, const void* x2, int c2, int n, int64_t voxels, int groups, int dtype,
                          const float* stats, const float* gamma, float eps, const void* dyh, const float* red, const float* chan,
                          const void* dout, const float* w1, const float* w2, int cout, void* dx1, void* dx2, float* dgamma,
                          float* dbeta, float* dw1, float* dw2, float* workspace, size_t workspace_floats, void* stream);

/* ---- small tensor ops on the path ------------------------------------------------------------ */
/* out[n][v][cpad] <- channels {a[n][v], b[n][v] (b may be NULL)} zero-padded to cpad; a,b fp32. */
int vdm_pack_input(const float* a, const float* b, int64_t nvox, int cpad, int dtype, void* out, void* stream);

/* ---- K6: conditioning embeddings -> additive injection table of all ResNetBlocks [R5; D4/D7] -------------------
 * Replaces the ATen chain behind score_model(zt, t=(gamma_t-gamma_min)/(gamma_max-gamma_min), v_conditionings=[...])
 * [NB vdm_model.py:320-324, networks.py:259-265]: per conditioning  c = GELU(Linear2(GELU(Linear1(in))))  with
 * in = sinusoidal embedding of t (sinusoid != 0: `input` is t[rows]) or the raw vector (input[rows][in_dim]), and
 * table[row][w] = sum_k c_k[row] . wproj_k[w]   over the concatenated output channels w of all blocks.
 * `mlps` is a HOST array of n <= 4 descriptors holding DEVICE pointers.  saved: vdm_cond_saved_floats() floats kept for the
 * backward (may be NULL for inference).  bwd writes dw1/db1/dw2/db2/dwproj of every descriptor (plain stores) and, if dbias is
 * given, dbias[w] = sum_rows dtable[row][w] (the conv1 bias gradients); scratch: >= vdm_cond_bwd_scratch_floats() floats
 * (the per-row (dh1, dh2) vectors + the fixed-order slab partials of dtable . Wproj).  Widths: in_dim, dim <= 256; a weight matrix whose
 * row length is a multiple of 4 must be 16-byte aligned.  rows <= 65535 per call.
 * step: table[b] = table_t[*step_ptr] + table_v[b] (either may be NULL) - the per-step row gather of the captured sampler graph. */
typedef struct vdm_cond_mlp {
    const float* input;
    int32_t in_dim, dim, sinusoid, reserved;
    const float* w1; const float* b1; /* [dim][in_dim], [dim] */
    const float* w2; const float* b2; /* [dim][dim], [dim] */
    const float* wproj;               /* [width][dim] */
    float* dw1; float* db1; float* dw2; float* db2; float* dwproj; /* backward outputs (NULL in forward) */
} vdm_cond_mlp;
size_t vdm_cond_saved_floats(const vdm_cond_mlp* mlps, int n, int rows);
size_t vdm_cond_bwd_scratch_floats(const vdm_cond_mlp* mlps, int n, int rows, int width);
int vdm_cond_table_fwd(const vdm_cond_mlp* mlps, int n, int rows, int width, float* table, float* saved, void* stream);
int vdm_cond_table_bwd(const vdm_cond_mlp* mlps, int n, int rows, int width, const float* dtable, int64_t dtable_stride,
                       const float* saved, float* scratch, float* dbias, void* stream);
int vdm_cond_table_step(const float* table_t, const float* table_v, const int32_t* step_ptr, int rows, int width, float* out,
                        void* stream);

/* ---- K7/K8: VDM forward diffusion + ELBO pieces [R7; D9/D10] -----------------------------------
 * z_t = alpha[n] x + sigma[n] eps  (fp32, contiguous per sample of `per` elements). */
int vdm_diffuse(const float* x, const float* eps, const float* alpha, const float* sigma, int n, int64_t per,
                float* z_t, void* stream);
/* sums[n][3] += {sum (eps-eps_hat)^2, sum x^2, sum (x - z0r)^2} with z0r = x + (sigma0/alpha0) eps0;
 * d_eps_hat = coef[n] * (eps_hat - eps)   (coef folds bpd * gamma'(t) / B).  Caller zeroes sums. */
/* workspace: >= 2048 * 3 floats (per-workgroup partials, folded in a fixed order: the loss is bit-reproducible). */
int vdm_loss_terms(const float* x, const float* eps, const float* eps_hat, const float* eps0, float sigma0_over_alpha0,
                   const float* coef, int n, int64_t per, float* sums, float* d_eps_hat, float* workspace, void* stream);

/* Fused head of the training step (ABI v11) [replaces the randn_like -> alpha x + sigma eps -> cat([z_t, s_conditioning]) chain of
 * VDM.get_loss + CUNet.forward, NB vdm_model.py:309-327 / networks.py:259-265]: z_t = alpha[n] x + sigma[n] eps, written as fp32
 * (z_t, may be NULL) AND as conv_in's NDHWC input packed[n][voxel][16 B] = {z_t, s_cond (0 if NULL), 0 ...} in `dtype` - one pass over
 * x instead of vdm_randn + vdm_diffuse + vdm_pack_input.  eps == NULL: the noise is drawn inside the kernel from the Philox counters
 * (seed, stream_id, element group; seed_step as for vdm_randn) - exactly the field vdm_randn(seed, stream_id) would have written;
 * eps != NULL: supplied noise (parity tests).  per % 4 == 0. */
int vdm_diffuse_pack(const float* x, const float* s_cond, const float* eps, uint64_t seed, uint64_t stream_id, const int32_t* seed_step,
                     const float* alpha, const float* sigma, int n, int64_t per, int dtype, float* z_t, void* packed, void* stream);
/* vdm_loss_terms with eps and / or eps0 regenerated from their Philox counters when NULL (the fields vdm_randn(seed_eps, stream_eps) /
 * vdm_randn(seed_eps0, stream_eps0) would have written): the training step never materialises its noise fields.  per % 4 == 0;
 * workspace as vdm_loss_terms.  With both fields supplied the sums equal this kernel's sums for the regenerated fields bit for bit. */
int vdm_loss_terms_rng(const float* x, const float* eps, uint64_t seed_eps, uint64_t stream_eps, const float* eps_hat, const float* eps0,
                       uint64_t seed_eps0, uint64_t stream_eps0, const int32_t* seed_step, float sigma0_over_alpha0, const float* coef, int n,
                       int64_t per, float* sums, float* d_eps_hat, float* workspace, void* stream);

/* ---- data path: crop + log-normalise + flip + permute on the device -----------------------------
 * Replaces the per-sample CPU DataLoader work of [REF src/dataset/CAMELS_3D_dataset.py:53-73] (AstroDataset.__getitem__) and
 * [REF src/dataset/augmentation.py:8-127] (Crop, LogTransform, Normalize, Flip, Permutate):
 *   out[c][b][i0][i1][i2] = (log10(raw + alpha) - mean) / std  of  field[c][sim][(anchor_d + c_d) % fullsize],
 *   c_d = flip[d] ? crop-1-f_d : f_d,  f[perm[k]] = i_k   (crop -> log/normalise -> flip -> permute, the reference's order).
 * The two tables are HOST arrays (they travel in the kernel arguments); `field` / `out` inside them are device pointers
 * (field: [n_sims][fullsize]^3 fp32 raw cubes resident in HBM; out: [n_samples][crop]^3 fp32).  n_channels <= 4. */
typedef struct {
    const float* field;
    float* out;
    float alpha, mean, std; /* [REF src/dataset/alphas_3d.json, normalizations_3d.json] */
} vdm_augment_channel;
typedef struct {
    int32_t sim;       /* simulation index into `field` */
    int32_t anchor[3]; /* crop origin incl. the random shift (may exceed fullsize: wrapped) [REF augmentation.py:113-117] */
    int32_t flip[3];   /* != 0: flip this axis [REF augmentation.py:51-52] */
    int32_t perm[3];   /* axis permutation [REF augmentation.py:72] */
} vdm_augment_sample;
int vdm_augment_batch(const vdm_augment_channel* host_channels, int n_channels, int fullsize, int crop,
                      const vdm_augment_sample* host_samples, int n_samples, void* stream);

/* ---- attention block of the mid level (CUNet(mid_attn=True, n_attention_heads)) [REF trainSFM_c_uc_from_field_name.py:61,104-118;
 * NB blocks.py:169-170 `x = self.attention_blocks[i](x)`] ---------------------------------------------------------------------
 * Fused core  out_i = sum_j softmax_j(scale q_i . k_j) v_j  per sample and head on the matrix cores, forward and backward, without
 * the [N, heads, V, V] score tensor (csrc/attention.hip).  Operands in `dtype` (bf16 MFMA / exact fp32 MFMA), softmax and sums fp32.
 *   split_heads: src[n][v][src_stride] (channels src_offset + h*hd ...) -> rowmajor [n][h][v][hd] and / or transposed [n][h][hd][v]
 *                (either may be NULL).  The q, k, v of the block's 1x1x1 qkv conv ([n][v][3][C]: stride 3C, offsets 0, C, 2C).
 *   fwd:   q, k row-major, vt transposed -> out [n][v][heads*hd] (the layout the projection conv reads), lse [n][h][v] (may be NULL).
 *   rowdot: out[n][h][v] = sum_d a[n][v][h*hd+d] b[n][v][h*hd+d]   (dsum = rowsum(dOut * out) of the backward).
 *   bwd:   -> dqkv [n][v][3][heads*hd]; dQ in its own pass over the keys: no float atomics, bit-reproducible.
 * voxels % 4 == 0; head_dim in {16, 32, 64, 96, 128}. */
int vdm_attn_split_heads(const void* src, int64_t src_stride, int64_t src_offset, int n, int64_t voxels, int heads, int head_dim,
                         int dtype, void* rowmajor, void* transposed, void* stream);
int vdm_attn_fwd(const void* q, const void* k, const void* vt, int n, int64_t voxels, int heads, int head_dim, int dtype, float scale,
                 void* out, float* lse, void* stream);
int vdm_attn_rowdot(const void* a, const void* b, int n, int64_t voxels, int heads, int head_dim, int dtype, float* out, void* stream);
int vdm_attn_bwd(const void* q, const void* k, const void* v, const void* qt, const void* kt, const void* do_rowmajor,
                 const void* do_transposed, const float* lse, const float* dsum, int n, int64_t voxels, int heads, int head_dim,
                 int dtype, float scale, void* dqkv, void* stream);
/* out[c] = sum over rows of x[row][c] (x: [rows][c] in `dtype`): bias gradients of the block's two 1x1x1 projections. */
int vdm_channel_sums(const void* x, int64_t rows, int c, int dtype, float* out, void* stream);
/* out[n][c1+c2][2] = (sum_rows a, sum_rows a * b) per sample, a: [n][rows][c1+c2], b = concat(b1 [n][rows][c1], b2 [n][rows][c2] or
 * NULL): with a = dyh and b = x of a GroupNorm these are the "one tile per sample" partials vdm_gn_bwd_finalize / vdm_gn_bwd_apply take. */
int vdm_channel_dot_sums(const void* a, const void* b1, int c1, const void* b2, int c2, int n, int64_t rows_per_sample, int dtype,
                         float* out, void* stream);

/* ---- scalar glue of the training step [R7, R11; D9-D11] (replaces ~60 five-microsecond ATen launches on the critical path) --------
 * train_scalars: out[5][batch] = {t, alpha_t, sigma_t, coef = gamma'(t) bpd / batch, t_norm} of the fixed linear schedule; with u0
 *   (DEVICE pointer to one uniform draw) t_i = (u0 + (rank batch + i) / (world batch)) mod 1 - the antithetic time grid stratified over
 *   the global batch - else t_i = times[i]; with neither, u0 is drawn in the kernel from Philox(seed, *seed_step) (same on every rank).
 *   [REF trainVDM3D128...py:128-132 -> LightVDM.training_step; NB vdm_model.py:320-324]
 * elbo_assemble: out[4] = {elbo, diffusion, latent, reconstruction} in bits/dim from the sums of vdm_loss_terms:
 *   diffusion = 0.5 sum_n coef_n S0_n, latent = mean_n (c_lat0 + c_lat1 S1_n), reconstruction = mean_n (c_rec0 S2_n + c_rec1).
 * clip_scale: x *= min(1, max_norm / (sqrt(*sumsq) + 1e-6)) - gradient_clip_val [REF trainVDM3D128...py:45] on the device. */
int vdm_train_scalars(const float* u0, const float* times, int batch, int rank, int world, float gamma_min, float gamma_max,
                      float bpd_over_batch, float* out, uint64_t seed, const int32_t* seed_step, void* stream);
int vdm_elbo_assemble(const float* sums, const float* coef, int batch, float c_lat0, float c_lat1, float c_rec0, float c_rec1,
                      float* out, void* stream);
int vdm_clip_scale(float* x, int64_t n, const float* sumsq, float max_norm, void* stream);

/* ---- K9: ancestral update [NB vdm_model.py:370-378] ----------------------------------------
 * z <- ratio*(z - c_sigma_t*eps_hat) + scale*noise ; the four scalars are read from the DEVICE table
 * coef[step][4] = {ratio, c*sigma_t, scale, t_norm} at row *step_ptr (so one captured graph serves
 * every step).  noise==NULL: Philox normal from (seed, *step_ptr, element index). */
int vdm_ancestral_step(float* z, const float* eps_hat, const float* noise, const float* coef, const int32_t* step_ptr,
                       uint64_t seed, int64_t n, void* stream);
/* The same update under classifier-free guidance [NB vdm_model.py:318-327, the `w_cfg` branch of VDM.get_pred_noise]:
 * eps_hat = (1 + w_cfg) * eps_cond - w_cfg * eps_uncond, blended inside the update (eps_cond / eps_uncond are the two halves of
 * one batch-doubled UNet forward: given v_conditionings / masked v_conditionings). */
int vdm_ancestral_step_cfg(float* z, const float* eps_cond, const float* eps_uncond, float w_cfg, const float* noise,
                           const float* coef, const int32_t* step_ptr, uint64_t seed, int64_t n, void* stream);
/* standard-normal fill from Philox(seed, stream_id) (z_1 of the sampler; eps in training). */
int vdm_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, const int32_t* seed_step, void* stream);
/* *step_ptr += 1 (device-side step counter for the captured sampler graph). */
int vdm_step_inc(int32_t* step_ptr, void* stream);

/* ---- K10: global gradient norm [REF trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:45] -- */
/* out[0] += sum x^2 (caller zeroes).  workspace: >= 2048 floats (fixed-order fold: bit-reproducible). */
int vdm_sumsq(const float* x, int64_t n, float* out, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VDM4CDM_HIP_H */
