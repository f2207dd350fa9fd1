// gn_skip.hip - the 1x1x1 skip convolution of a ResNetBlock folded into the GroupNorm passes that read the same tensor
// (reference call chain: trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:116-127 -> [NB] blocks.py ResNetBlock:
//  out = conv2(drop(silu(norm2(conv1(silu(norm1(x))) + cond))) + skip(x),  skip = Conv3d(cin, cout, 1) when cin != cout).
//
// norm1 and the skip conv read the same raw block input x (two tensors on the up path: the concat is never materialised), and in
// the backward pass the GroupNorm apply pass reads x again while the skip conv's input gradient W^T dout is simply added to its
// result.  A 64 -> 32 channel 1x1x1 conv is 4 MFMAs per 16 voxels next to 3 KB of HBM traffic, so it rides along for free:
//
//   forward   y = silu(x A + B)            s = W x + b                      one read of x instead of two, no separate launch
//   backward  dx = P dyh + Q x + R + W^T dout,   dW = dout^T x              dout read once instead of 3x, no dxs round trip
//
// MFMA operand layouts (v_mfma_f32_16x16x32_bf16: A lane (row = lane&15, k = 8 (lane>>4) + j), B lane (col = lane&15, same k),
// D lane (col = lane&15, rows 4 (lane>>4) + e)):
//   * a lane (v = lane&15, q = lane>>4) streams the 16-byte piece q of K-step ks of voxel v - exactly the B operand of W x
//     (k = channel) and of W^T dout (k = cout); the result rows are permuted through the weight fragments so that a lane ends up
//     with consecutive channels of ITS voxel: the skip output goes out as 16-byte stores and the W^T dout tile lands on the lane
//     that holds the same 8 channels of dyh and x.
//   * dW = dout^T x contracts over voxels: both operands go through a per-wave LDS tile (32 voxels, row-major, written as the
//     pieces come in) and come back transposed by ds_read_b64_tr_b16 - the fetch of the weight-gradient convs (conv_wgrad.hip).
//     Per-workgroup slabs + a fixed-order reduce: no float atomics, bit-reproducible.
// bf16 storage only (the fp32 configurations keep the separate 1x1x1 conv kernels).
#include "common.h"

namespace vdm {

struct GnSkipArgs {
    const bf16_t* x1; const bf16_t* x2;
    int c1, c2, n, G, cout;
    int64_t V;
    const float* stats; const float* gamma; const float* beta;
    float eps;
    const float* w1; const float* w2; const float* bias;      // fp32 master weights [cout][c1], [cout][c2]; bias [cout]
    bf16_t* y; bf16_t* s;                                     // forward out
    const bf16_t* dyh; const bf16_t* dout;                    // backward in
    const float* red; const float* chan;
    bf16_t* dx1; bf16_t* dx2;
    float* dgamma; float* dbeta;
    float* slabs;                                             // [gridDim.y * gridDim.x][cout][32 KS]
};

__device__ __forceinline__ float skip_w(const GnSkipArgs& a, int co, int c) {
    if (co >= a.cout) return 0.f;
    if (c < a.c1) return a.w1[(size_t)co * a.c1 + c];
    if (c < a.c1 + a.c2) return a.w2[(size_t)co * a.c2 + (c - a.c1)];
    return 0.f;
}

__device__ __forceinline__ bf16x8 pack8(const float (&w)[8]) {
    const uint4 u = make_uint4(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]), pack_bf16x2(w[4], w[5]), pack_bf16x2(w[6], w[7]));
    return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ void gn_mean_rstd(const float* __restrict__ stats, int n, int G, int g, float cnt, float eps, float& mean, float& rstd) {
    const float sum = stats[((size_t)n * G + g) * 2], sq = stats[((size_t)n * G + g) * 2 + 1];
    mean = sum / cnt;
    rstd = rsqrtf(fmaxf(sq / cnt - mean * mean, 0.f) + eps);
}

// ---------------------------------------------------------------------------------------------
// forward: KS = K-steps of 32 input channels (C <= 32 KS), MT = 16-row tiles of cout (cout == 16 MT)
// ---------------------------------------------------------------------------------------------
template <int KS, int MT>
__global__ void __launch_bounds__(256) gn_silu_skip_fwd_kernel(const GnSkipArgs a) {
    constexpr int CW = 32 * KS, CO = 16 * MT;
    constexpr bool AB_LDS = KS >= 4;                        // wide inputs: the GroupNorm affines stay in LDS (registers go to the weights)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v16 = lane & 15, q = lane >> 4;
    const int C = a.c1 + a.c2, PPV = C / 8, P1 = a.c1 / 8, gs = C / a.G;
    const int n = blockIdx.y;
    const float cnt = (float)a.V * gs;

    // the workgroup rounds the master weights to bf16 and evaluates the per-channel affines once, through LDS (coalesced loads; a
    // per-lane gather of the fragments from global memory cost more than the streaming loop of a wave)
    __shared__ __attribute__((aligned(16))) uint16_t wl[CO * CW];
    __shared__ __attribute__((aligned(16))) float abA[CW], abB[CW];
    for (int i = threadIdx.x; i < CO * CW; i += 256) wl[i] = f32_to_bf16(skip_w(a, i / CW, i % CW));
    for (int c = threadIdx.x; c < CW; c += 256) {
        float A = 0.f, B = 0.f;
        if (c < C) {
            float mean, rstd;
            gn_mean_rstd(a.stats, n, a.G, c / gs, cnt, a.eps, mean, rstd);
            A = rstd * a.gamma[c];
            B = a.beta[c] - mean * A;
        }
        abA[c] = A;
        abB[c] = B;
    }
    __syncthreads();
    bf16x8 wf[MT][KS];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = (v16 >> 2) * (4 * MT) + m * 4 + (v16 & 3);       // D row 4g + e of tile m  <->  cout g * 4MT + 4m + e
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[m][ks] = *reinterpret_cast<const bf16x8*>(&wl[co * CW + ks * 32 + q * 8]);
    }
    float bia[4 * MT];
#pragma unroll
    for (int i = 0; i < 4 * MT; ++i) bia[i] = a.bias ? a.bias[q * 4 * MT + i] : 0.f;
    float A[AB_LDS ? 1 : KS][8], B[AB_LDS ? 1 : KS][8];
    if constexpr (!AB_LDS) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) { A[ks][j] = abA[ks * 32 + q * 8 + j]; B[ks][j] = abB[ks * 32 + q * 8 + j]; }
    }
    const bf16_t* x1 = a.x1 + (size_t)n * a.V * a.c1;
    const bf16_t* x2 = a.x2 ? a.x2 + (size_t)n * a.V * a.c2 : nullptr;
    bf16_t* y = a.y + (size_t)n * a.V * C;
    bf16_t* s = a.s + (size_t)n * a.V * a.cout;
    const int64_t ngroups = (a.V + 15) / 16, gstep = (int64_t)gridDim.x * 4;

    auto fetch = [&](uint4 (&raw)[KS], int64_t g) {
        const int64_t v = g * 16 + v16;
        const bool ok = g < ngroups && v < a.V;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int p = ks * 4 + q;
            raw[ks] = make_uint4(0u, 0u, 0u, 0u);
            if (ok && p < PPV)
                raw[ks] = p < P1 ? *reinterpret_cast<const uint4*>(x1 + v * a.c1 + p * 8) : *reinterpret_cast<const uint4*>(x2 + v * a.c2 + (p - P1) * 8);
        }
    };
    int64_t g = (int64_t)blockIdx.x * 4 + wave;
    uint4 raw[KS], nxt[KS];
    fetch(raw, g);
    for (; g < ngroups; g += gstep) {
        fetch(nxt, g + gstep);                              // the next group's loads fly behind this group's arithmetic
        const int64_t v = g * 16 + v16;
        const bool ok = v < a.V;
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m][ks], __builtin_bit_cast(bf16x8, raw[ks]), acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int p = ks * 4 + q;
            if (ok && p < PPV) {
                Piece<bf16_t> px;
                px.load(raw[ks]);
                if constexpr (AB_LDS) {
                    const float4 a0 = *reinterpret_cast<const float4*>(&abA[p * 8]), a1 = *reinterpret_cast<const float4*>(&abA[p * 8 + 4]);
                    const float4 b0 = *reinterpret_cast<const float4*>(&abB[p * 8]), b1 = *reinterpret_cast<const float4*>(&abB[p * 8 + 4]);
                    const float Aj[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, Bj[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) px.f[j] = silu_f(px.f[j] * Aj[j] + Bj[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) px.f[j] = silu_f(px.f[j] * A[ks][j] + B[ks][j]);
                }
                *reinterpret_cast<uint4*>(y + v * C + p * 8) = px.store();
            }
        }
        if (ok) {
            float o[4 * MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float e0 = acc[m][0], e1 = acc[m][1], e2 = acc[m][2], e3 = acc[m][3];
                o[m * 4 + 0] = e0 + bia[m * 4 + 0]; o[m * 4 + 1] = e1 + bia[m * 4 + 1];
                o[m * 4 + 2] = e2 + bia[m * 4 + 2]; o[m * 4 + 3] = e3 + bia[m * 4 + 3];
            }
            bf16_t* sp = s + v * a.cout + q * 4 * MT;
            if constexpr (MT == 1) {
                *reinterpret_cast<uint2*>(sp) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
            } else {
#pragma unroll
                for (int i = 0; i < MT / 2; ++i)
                    *reinterpret_cast<uint4*>(sp + i * 8) = make_uint4(pack_bf16x2(o[i * 8 + 0], o[i * 8 + 1]), pack_bf16x2(o[i * 8 + 2], o[i * 8 + 3]),
                                                                       pack_bf16x2(o[i * 8 + 4], o[i * 8 + 5]), pack_bf16x2(o[i * 8 + 6], o[i * 8 + 7]));
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) raw[ks] = nxt[ks];
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int KS, int MT> struct SkipBwdGeo {
    static constexpr int KD = (MT + 1) / 2;                 // K-steps of 32 couts for W^T dout
    static constexpr int CT = 2 * KS;                       // 16-channel tiles of the input channels
    static constexpr int PX = 64 * KS + 16;                 // LDS row pitch of the x tile (bytes; +16: spreads the rows over the banks)
    static constexpr int PD = 32 * MT + 16;                 // ... of the dout tile
    static constexpr int WAVE_BYTES = 32 * (PX + PD);
    static constexpr int FOLD_BYTES = 16 * MT * 32 * KS * 4;
    static constexpr int WT_BYTES = 32 * KS * 32 * KD * 2;  // prologue: W^T as bf16 [channel][cout]
    static constexpr int PRO_BYTES = WT_BYTES + 3 * 32 * KS * 4;
    static constexpr int LDS_BYTES = PRO_BYTES + (4 * WAVE_BYTES > FOLD_BYTES ? 4 * WAVE_BYTES : FOLD_BYTES);
};

// 16 columns (tile ct) x 32 rows of a row-major bf16 LDS tile, transposed: lane (g = lane>>4, i = lane&15) receives column i of the
// rows {4g..4g+3} and {16+4g..16+4g+3} - the k-slots of an MFMA operand (both operands of dW use the same row <-> k mapping).
__device__ __forceinline__ bf16x8 tr_tile(const char* tile, int pitch, int ct, int lane) {
    typedef __attribute__((address_space(3))) s16x4* lptr;
    const int g = lane >> 4, li = lane & 15, qp = li >> 2, p = li & 3;
    const char* ptr = tile + (4 * g + qp) * pitch + ct * 32 + p * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ptr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ptr + 16 * pitch));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
}

template <int KS, int MT>
__global__ void __launch_bounds__(256) gn_bwd_apply_skip_kernel(const GnSkipArgs a) {
    using Geo = SkipBwdGeo<KS, MT>;
    constexpr int KD = Geo::KD, CT = Geo::CT, PX = Geo::PX, PD = Geo::PD;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v16 = lane & 15, q = lane >> 4;
    const int C = a.c1 + a.c2, PPV = C / 8, P1 = a.c1 / 8, gs = C / a.G, PPD = a.cout / 8;
    const int n = blockIdx.y;
    const float cnt = (float)a.V * gs;
    if (blockIdx.x == 0 && blockIdx.y == 0) {               // GroupNorm parameter gradients: sum over the samples, fixed order
        for (int c = threadIdx.x; c < C; c += 256) {
            float db = 0.f, dg = 0.f;
            for (int k = 0; k < a.n; ++k) { db += a.chan[((size_t)k * C + c) * 2]; dg += a.chan[((size_t)k * C + c) * 2 + 1]; }
            a.dbeta[c] = db;
            a.dgamma[c] = dg;
        }
    }
    // prologue through LDS (coalesced, once per workgroup): W^T rounded to bf16 as [channel][cout] and the per-channel constants of
    // dx = dyh * P + x * Q + R   with  P = rstd gamma, Q = -rstd^2 m2, R = rstd (mean rstd m2 - m1)
    constexpr int CW = 32 * KS, KW = 32 * KD;
    uint16_t* wtl = reinterpret_cast<uint16_t*>(lds);                   // [CW][KW]
    float* pqr = reinterpret_cast<float*>(lds + Geo::WT_BYTES);        // [3][CW]
    for (int i = threadIdx.x; i < CW * KW; i += 256) wtl[i] = f32_to_bf16(skip_w(a, i % KW, i / KW));
    for (int c = threadIdx.x; c < CW; c += 256) {
        float P = 0.f, Q = 0.f, R = 0.f;
        if (c < C) {
            const int g = c / gs;
            float mean, rstd;
            gn_mean_rstd(a.stats, n, a.G, g, cnt, a.eps, mean, rstd);
            const float m1 = a.red[((size_t)n * a.G + g) * 2] / cnt, m2 = a.red[((size_t)n * a.G + g) * 2 + 1] / cnt;
            P = rstd * a.gamma[c];
            Q = -rstd * rstd * m2;
            R = rstd * (mean * rstd * m2 - m1);
        }
        pqr[c] = P; pqr[CW + c] = Q; pqr[2 * CW + c] = R;
    }
    __syncthreads();
    char* tx = lds + Geo::PRO_BYTES + wave * Geo::WAVE_BYTES;
    char* td = tx + 32 * PX;
    // W^T fragments: tile (cb, m2) row r <-> channel cb * 32 + (r >> 2) * 8 + m2 * 4 + (r & 3): D rows 4g + e of the two tiles are the
    // channels 8g + 4 m2 + e = the piece q = g of K-step cb that this lane streams.  Small shapes hold the fragments and the
    // constants in registers; wide ones (BIG) read them from LDS where they are used - their registers go to the dW accumulators.
    constexpr bool BIG = KS * MT > 8;
    constexpr int KSR = BIG ? 1 : KS;
    const int crow = (v16 >> 2) * 8 + (v16 & 3);
    bf16x8 wt[KSR][2][KD];
    float Pc[KSR][8], Qc[KSR][8], Rc[KSR][8];
    if constexpr (!BIG) {
#pragma unroll
        for (int cb = 0; cb < KS; ++cb)
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int kd = 0; kd < KD; ++kd) wt[cb][m2][kd] = *reinterpret_cast<const bf16x8*>(&wtl[(cb * 32 + crow + m2 * 4) * KW + kd * 32 + q * 8]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = ks * 32 + q * 8 + j;
                Pc[ks][j] = pqr[c]; Qc[ks][j] = pqr[CW + c]; Rc[ks][j] = pqr[2 * CW + c];
            }
    }
    f32x4 dw[MT][CT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dw[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bf16_t* x1 = a.x1 + (size_t)n * a.V * a.c1;
    const bf16_t* x2 = a.x2 ? a.x2 + (size_t)n * a.V * a.c2 : nullptr;
    const bf16_t* dyh = a.dyh + (size_t)n * a.V * C;
    const bf16_t* dout = a.dout + (size_t)n * a.V * a.cout;
    bf16_t* dx1 = a.dx1 + (size_t)n * a.V * a.c1;
    bf16_t* dx2 = a.dx2 ? a.dx2 + (size_t)n * a.V * a.c2 : nullptr;
    const int64_t nchunks = (a.V + 31) / 32;
    for (int64_t ch = (int64_t)blockIdx.x * 4 + wave; ch < nchunks; ch += (int64_t)gridDim.x * 4) {
        uint4 rx[2][KS], ry[2][KS], ro[2][KD];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t v = ch * 32 + h * 16 + v16;
            const bool ok = v < a.V;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int p = ks * 4 + q;
                rx[h][ks] = ry[h][ks] = make_uint4(0u, 0u, 0u, 0u);
                if (ok && p < PPV) {
                    rx[h][ks] = p < P1 ? *reinterpret_cast<const uint4*>(x1 + v * a.c1 + p * 8) : *reinterpret_cast<const uint4*>(x2 + v * a.c2 + (p - P1) * 8);
                    ry[h][ks] = *reinterpret_cast<const uint4*>(dyh + v * C + p * 8);
                }
            }
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                const int pd = kd * 4 + q;
                ro[h][kd] = (ok && pd < PPD) ? *reinterpret_cast<const uint4*>(dout + v * a.cout + pd * 8) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        // the tile for dW (zeros where there is no voxel / channel: they add nothing)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<uint4*>(tx + (h * 16 + v16) * PX + (ks * 4 + q) * 16) = rx[h][ks];
#pragma unroll
            for (int kd = 0; kd < KD; ++kd)
                if (kd * 4 + q < 2 * MT) *reinterpret_cast<uint4*>(td + (h * 16 + v16) * PD + (kd * 4 + q) * 16) = ro[h][kd];
        }
        // dx
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t v = ch * 32 + h * 16 + v16;
            const bool ok = v < a.V;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f}, t1 = t0;
#pragma unroll
                for (int kd = 0; kd < KD; ++kd) {
                    bf16x8 w0, w1;
                    if constexpr (BIG) {
                        w0 = *reinterpret_cast<const bf16x8*>(&wtl[(ks * 32 + crow) * KW + kd * 32 + q * 8]);
                        w1 = *reinterpret_cast<const bf16x8*>(&wtl[(ks * 32 + crow + 4) * KW + kd * 32 + q * 8]);
                    } else {
                        w0 = wt[ks][0][kd];
                        w1 = wt[ks][1][kd];
                    }
                    t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, __builtin_bit_cast(bf16x8, ro[h][kd]), t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, __builtin_bit_cast(bf16x8, ro[h][kd]), t1, 0, 0, 0);
                }
                const int p = ks * 4 + q;
                if (ok && p < PPV) {
                    Piece<bf16_t> px, pd;
                    px.load(rx[h][ks]);
                    pd.load(ry[h][ks]);
                    const float d0 = t0[0], d1 = t0[1], d2 = t0[2], d3 = t0[3], d4 = t1[0], d5 = t1[1], d6 = t1[2], d7 = t1[3];
                    const float ds[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
                    if constexpr (BIG) {
                        const float* c0 = pqr + p * 8;
#pragma unroll
                        for (int j = 0; j < 8; ++j) px.f[j] = fmaf(pd.f[j], c0[j], fmaf(px.f[j], c0[CW + j], c0[2 * CW + j])) + ds[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) px.f[j] = fmaf(pd.f[j], Pc[ks][j], fmaf(px.f[j], Qc[ks][j], Rc[ks][j])) + ds[j];
                    }
                    if (p < P1)
                        *reinterpret_cast<uint4*>(dx1 + v * a.c1 + p * 8) = px.store();
                    else
                        *reinterpret_cast<uint4*>(dx2 + v * a.c2 + (p - P1) * 8) = px.store();
                }
            }
        }
        // dW += dout^T x over the 32 voxels of the chunk (LDS operations of one wave execute in order; the fences keep the compiler
        // from moving the transposed reads across the stores of this / the next chunk)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8 bx[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bx[ct] = tr_tile(tx, PX, ct, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16x8 ad = tr_tile(td, PD, mt, lane);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dw[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, bx[ct], dw[mt][ct], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // workgroup slab: the four waves add their tiles in wave order through LDS (fixed order), then one coalesced write
    float* f = reinterpret_cast<float*>(lds + Geo::PRO_BYTES);
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float e0 = dw[mt][ct][0], e1 = dw[mt][ct][1], e2 = dw[mt][ct][2], e3 = dw[mt][ct][3];
                    const float e[4] = {e0, e1, e2, e3};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int idx = (mt * 16 + 4 * q + k) * CW + ct * 16 + v16;
                        f[idx] = w == 0 ? e[k] : f[idx] + e[k];
                    }
                }
        }
    }
    __syncthreads();
    float* slab = a.slabs + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (16 * MT * CW);
    for (int i = threadIdx.x; i < 16 * MT * CW; i += 256) slab[i] = f[i];
}

// dw1[co][c] / dw2[co][c - c1] = sum over the slabs, fixed order: a block owns 16 consecutive slab elements, thread (e = t & 15,
// l = t >> 4) adds the slabs l, l + 16, ..., the 16 partial sums are folded in lane order.
__global__ void __launch_bounds__(256) skip_dw_reduce_kernel(const float* __restrict__ slabs, int nslabs, int cout, int CW, int c1, int c2,
                                                            float* __restrict__ dw1, float* __restrict__ dw2) {
    const int e = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + e;
    const size_t per = (size_t)cout * CW;
    float s = 0.f;
    constexpr int U = 8;
    for (int b = l; b < nslabs; b += 16 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (b + 16 * u) < nslabs ? slabs[(size_t)(b + 16 * u) * per + idx] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
    }
    __shared__ float sm[256];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (l == 0) {
        float tot = 0.f;
        for (int k = 0; k < 16; ++k) tot += sm[k * 16 + e];
        const int co = idx / CW, c = idx % CW;
        if (c < c1) dw1[(size_t)co * c1 + c] = tot;
        else if (c < c1 + c2) dw2[(size_t)co * c2 + (c - c1)] = tot;
    }
}

static bool skip_shape(int c1, int c2, int cout, int& ks, int& mt) {
    if (c1 <= 0 || c2 < 0 || c1 > 512 || c2 > 512 || (c1 % 8) || (c2 % 8) || cout <= 0 || cout > 512 || (cout % 16)) return false;   // (bounds before sums)
    const int C = c1 + c2;
    ks = (C + 31) / 32;
    mt = cout / 16;
    return true;
}
static bool fwd_supported(int ks, int mt) {
    return (ks == 1 && (mt == 1 || mt == 2 || mt == 4)) || (ks == 2 && (mt == 2 || mt == 4 || mt == 8)) || (ks == 4 && mt == 4);
}
static bool bwd_supported(int ks, int mt) {
    return (ks == 1 && (mt == 1 || mt == 2 || mt == 4)) || (ks == 2 && (mt == 2 || mt == 4)) || (ks == 4 && mt == 4);
}

// wide shapes run one wave per SIMD (their dW accumulators fill the register file): one workgroup per CU, half the slabs to reduce
static int bwd_wgs_per_cu(int ks, int mt) { return ks * mt > 8 ? 1 : 2; }

static int skip_grid_x(int64_t units, int n, int per_cu) {      // workgroups per sample: 4 waves, one unit per wave and iteration
    int64_t want = (units + 3) / 4, cap = (256 * per_cu + n - 1) / n;
    if (want > cap) want = cap;
    return (int)(want < 1 ? 1 : want);
}

}  // namespace vdm

using namespace vdm;

extern "C" int vdm_gn_skip_supported(int c1, int c2, int cout, int dtype) {
    int ks, mt;
    if (dtype != VDM_BF16 || !skip_shape(c1, c2, cout, ks, mt)) return 0;
    return (fwd_supported(ks, mt) ? 1 : 0) | (bwd_supported(ks, mt) ? 2 : 0);
}

extern "C" size_t vdm_gn_skip_ws_floats(int c1, int c2, int cout, int n, int64_t voxels) {
    int ks, mt;
    if (!skip_shape(c1, c2, cout, ks, mt) || n <= 0 || voxels <= 0) return 0;
    return (size_t)skip_grid_x((voxels + 31) / 32, n, bwd_wgs_per_cu(ks, mt)) * n * cout * 32 * ks;
}

static int skip_common_check(int c1, int c2, int n, int64_t voxels, int groups, int cout, int dtype, const char* who) {
    VDM_REQUIRE(dtype == VDM_BF16, "%s: bf16 storage only (got dtype %d)", who, dtype);
    VDM_REQUIRE(n > 0 && n <= 65535 && voxels > 0 && groups > 0 && groups <= 64, "%s: bad n/voxels/groups", who);
    VDM_REQUIRE(c1 > 0 && c2 >= 0 && c1 <= 512 && c2 <= 512 && (c1 % 8) == 0 && (c2 % 8) == 0, "%s: channel counts must be multiples of 8 (got %d,%d)", who, c1, c2);
    const int C = c1 + c2;
    VDM_REQUIRE(C % groups == 0 && c1 % (C / groups) == 0, "%s: channels %d / groups %d (or a group straddles the concat boundary)", who, C, groups);
    VDM_REQUIRE(cout > 0 && cout % 16 == 0, "%s: cout must be a multiple of 16 (got %d)", who, cout);
    return VDM_OK;
}

#define SKIP_FWD_CASE(K, M)                                                                                                  \
    if (ks == K && mt == M) {                                                                                                \
        hipLaunchKernelGGL((gn_silu_skip_fwd_kernel<K, M>), grid, dim3(256), 0, s, a);                                       \
        launched = true;                                                                                                     \
    }

extern "C" int vdm_gn_silu_skip_fwd(const void* x1, int c1, const void* x2, int c2, int n, int64_t voxels, int groups, int dtype,
                                    const float* stats, const float* gamma, const float* beta, float eps, const float* w1, const float* w2,
                                    const float* bias, int cout, void* y, void* skip_out, void* stream) {
    int e = skip_common_check(c1, c2, n, voxels, groups, cout, dtype, "gn_silu_skip_fwd");
    if (e) return e;
    VDM_REQUIRE(x1 && stats && gamma && beta && w1 && y && skip_out && (c2 == 0 || (x2 && w2)), "gn_silu_skip_fwd: NULL pointer");
    int ks, mt;
    VDM_REQUIRE(skip_shape(c1, c2, cout, ks, mt) && fwd_supported(ks, mt), "gn_silu_skip_fwd: shape %d+%d -> %d not supported (vdm_gn_skip_supported)",
                c1, c2, cout);
    GnSkipArgs a{};
    a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.c1 = c1; a.c2 = c2; a.n = n; a.G = groups; a.cout = cout; a.V = voxels;
    a.stats = stats; a.gamma = gamma; a.beta = beta; a.eps = eps; a.w1 = w1; a.w2 = w2; a.bias = bias;
    a.y = (bf16_t*)y; a.s = (bf16_t*)skip_out;
    const dim3 grid(skip_grid_x((voxels + 15) / 16, n, 4), n);
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
    SKIP_FWD_CASE(1, 1) SKIP_FWD_CASE(1, 2) SKIP_FWD_CASE(1, 4) SKIP_FWD_CASE(2, 2) SKIP_FWD_CASE(2, 4) SKIP_FWD_CASE(2, 8) SKIP_FWD_CASE(4, 4)
    VDM_REQUIRE(launched, "gn_silu_skip_fwd: no kernel for KS=%d MT=%d", ks, mt);
    VDM_LAUNCH_CHECK("gn_silu_skip_fwd_kernel");
    return VDM_OK;
}

#define SKIP_BWD_CASE(K, M)                                                                                                  \
    if (ks == K && mt == M) {                                                                                                \
        constexpr int lds_bytes = SkipBwdGeo<K, M>::LDS_BYTES;                                                               \
        hipLaunchKernelGGL((gn_bwd_apply_skip_kernel<K, M>), grid, dim3(256), lds_bytes, s, a);                              \
        launched = true;                                                                                                     \
    }

extern "C" int vdm_gn_bwd_apply_skip(const void* x1, int c1, const void* x2, int c2, int n, int64_t voxels, int groups, int dtype,
                                     const float* stats, const float* gamma, float eps, const void* dyh, const float* red, const float* chan,
                                     const void* dout, const float* w1, const float* w2, int cout, void* dx1, void* dx2, float* dgamma,
                                     float* dbeta, float* dw1, float* dw2, float* workspace, size_t workspace_floats, void* stream) {
    int e = skip_common_check(c1, c2, n, voxels, groups, cout, dtype, "gn_bwd_apply_skip");
    if (e) return e;
    VDM_REQUIRE(x1 && stats && gamma && dyh && red && chan && dout && w1 && dx1 && dgamma && dbeta && dw1 && workspace &&
                    (c2 == 0 || (x2 && dx2 && w2 && dw2)), "gn_bwd_apply_skip: NULL pointer");
    int ks, mt;
    VDM_REQUIRE(skip_shape(c1, c2, cout, ks, mt) && bwd_supported(ks, mt), "gn_bwd_apply_skip: shape %d+%d -> %d not supported (vdm_gn_skip_supported)",
                c1, c2, cout);
    VDM_REQUIRE(workspace_floats >= vdm_gn_skip_ws_floats(c1, c2, cout, n, voxels), "gn_bwd_apply_skip: workspace too small (vdm_gn_skip_ws_floats)");
    GnSkipArgs a{};
    a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.c1 = c1; a.c2 = c2; a.n = n; a.G = groups; a.cout = cout; a.V = voxels;
    a.stats = stats; a.gamma = gamma; a.eps = eps; a.w1 = w1; a.w2 = w2;
    a.dyh = (const bf16_t*)dyh; a.dout = (const bf16_t*)dout; a.red = red; a.chan = chan;
    a.dx1 = (bf16_t*)dx1; a.dx2 = (bf16_t*)dx2; a.dgamma = dgamma; a.dbeta = dbeta; a.slabs = workspace;
    const dim3 grid(skip_grid_x((voxels + 31) / 32, n, bwd_wgs_per_cu(ks, mt)), n);
    hipStream_t s = (hipStream_t)stream;
    bool launched = false;
    SKIP_BWD_CASE(1, 1) SKIP_BWD_CASE(1, 2) SKIP_BWD_CASE(1, 4) SKIP_BWD_CASE(2, 2) SKIP_BWD_CASE(2, 4) SKIP_BWD_CASE(4, 4)
    VDM_REQUIRE(launched, "gn_bwd_apply_skip: no kernel for KS=%d MT=%d", ks, mt);
    VDM_LAUNCH_CHECK("gn_bwd_apply_skip_kernel");
    const int CW = 32 * ks;
    hipLaunchKernelGGL(skip_dw_reduce_kernel, dim3(cout * CW / 16), dim3(256), 0, s, (const float*)workspace, (int)(grid.x * grid.y), cout, CW, c1, c2,
                       dw1, dw2);
    VDM_LAUNCH_CHECK("skip_dw_reduce_kernel");
    return VDM_OK;
}
