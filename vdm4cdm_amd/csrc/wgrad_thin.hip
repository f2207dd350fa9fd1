// wgrad_thin.hip - weight gradient of the 3x3x3 stride-1 convs with ONE thin side: conv_in (2 -> chs[0] channels: the noisy field + the
// conditioning field) and conv_out (chs[0] -> 1) [reference: the first / last Conv3d of the score network, built in
// trainVDM3D128_c_c_from_field_name_thick_lowbatch.py:116-127 -> NB CUNet].  Both are
//     G[t][c][j] = sum_{n,v} D[n][v][c] * T[n][v + off(t)][j]        D: dense tensor (C = 16 / 32 / 64 channels), T: thin tensor (j < 2)
// conv_in:  D = dout, T = x:      dW[t][co][ci] = G[t][co][ci],  dbias[co] = sum_v D[v][co]
// conv_out: D = x,    T = dout:   dW[t][0][ci]  = sum_u x[u][ci] dout[u - off(t)] = G[26 - t][ci][0]
// The generic weight-gradient kernel runs them as 27 taps x (1 x 1) MFMA tiles of which 14 of 16 (15 of 16) rows multiply padding and
// stages a 64-byte-per-voxel halo for 4 real bytes: 0.19 ms for 0.3 GB of traffic.  Here the (tap, thin channel) pairs are the N
// dimension of the MFMA: per chunk of 32 x-consecutive voxels the nine (dz, dy) neighbour rows of T (36 voxels x 4 bytes each) go to LDS
// as flat bf16 rows L_r, and because consecutive dx are consecutive positions of a row the im2col matrix is a sliding window of them,
// M[v][8 r + c'] = L_r[2 v + c'] (c' = 2 dx + j < 6) - a transposed LDS read (ds_read_b64_tr_b16) of 16 columns at a time, no gather.
// The D chunk goes through LDS as in gn_skip.hip.  10 MFMAs per 32 voxels (C = 32) instead of 54, the dense tensor streamed once at
// the HBM rate.  Column 72 is a row of ones: its result column is sum_v D[v][c] (the bias gradient of conv_in).  Per-workgroup slabs,
// fixed-order reduce + scatter into the master layout: bit-reproducible.  bf16 storage.
#include "common.h"

namespace vdm {

struct ThinArgs {
    const bf16_t* dense;     // [N][Dz][Dy][Dx][C]
    const uint32_t* thin;    // [N][Dz][Dy][Dx]: channels 0..1 of the thin tensor, compacted (thin_compact_kernel)
    int N, Dz, Dy, Dx, C, circular;
    float* slabs;            // [gridDim.x][C][80]
    // GNA (the dense tensor is the RESULT of a GroupNorm backward apply pass that nobody else reads - the gradient at the output of
    // conv_in): dense = dyh * P + x * Q + R (+ add) is formed in the kernel instead of being written and read back
    const bf16_t* gx; const bf16_t* gdyh; const bf16_t* gadd;
    const float* gstats; const float* ggamma; const float* gred; const float* gchan;
    float* gdgamma; float* gdbeta;
    int gG; float geps;
    FastDiv fdx, fdy, fdz;   // divisions of the chunk index by the x chunks per row, Dy, Dz (scalar multiply-high: the index is wave-uniform)
};

constexpr int THIN_COLS = 80;                               // 9 rows x 8 + the ones row (72..79)
constexpr int THIN_ROWP = 160;                              // LDS pitch of one L row: 36 positions x 4 B, padded (bank spread)

__device__ __forceinline__ bf16x8 tr16(const char* p0, int second) {
    typedef __attribute__((address_space(3))) s16x4* lptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p0 + second));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
}

// channels 0..1 of T[voxel][8] as one 32-bit word per voxel: the row loads of the main kernel become contiguous (reading 4 of every
// 16 bytes in place cost it 3.5x the dense tensor's bytes through the vector L1)
__global__ void __launch_bounds__(256) thin_compact_kernel(const bf16_t* __restrict__ t, long long nvox, uint32_t* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvox; i += (long long)gridDim.x * 256)
        out[i] = *reinterpret_cast<const uint32_t*>(t + i * 8);
}

template <int MT, bool CIRC, bool GNA = false>              // MT: 16-channel tiles of the dense tensor; CIRC: circular padding
__global__ void __launch_bounds__(256) wgrad_thin_kernel(const ThinArgs a) {
    constexpr int C = 16 * MT, PD = C * 2 + 16;             // dense tile pitch (bytes)
    constexpr int LROWS = 10;                               // 9 neighbour rows + the ones row
    constexpr int WAVE_BYTES = 32 * PD + 2 * LROWS * THIN_ROWP;
    constexpr int FOLD_BYTES = C * THIN_COLS * 4;
    __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_BYTES > FOLD_BYTES ? 4 * WAVE_BYTES : FOLD_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v16 = lane & 15, q = lane >> 4;
    char* td = lds + wave * WAVE_BYTES;                     // dense chunk, row-major [32 voxels][C]
    char* la = td + 32 * PD;                                // L rows, copy A: entry p at 4 p
    char* lb = la + LROWS * THIN_ROWP;                      // copy B: entry p at 4 (p - 1)  (8-byte aligned reads for odd voxels)
    // the ones row (never rewritten): flat bf16 [1, 0, 1, 0, ...] -> M[v][72] = 1, M[v][73] = 0
    for (int p = lane; p < 40; p += 64) {
        *reinterpret_cast<uint32_t*>(la + 9 * THIN_ROWP + p * 4) = 0x00003f80u;
        *reinterpret_cast<uint32_t*>(lb + 9 * THIN_ROWP + p * 4) = 0x00003f80u;
    }
    f32x4 acc[MT][5];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // GNA: per (sample, channel) constants of dx = dyh * P + x * Q + R (gn_bwd_apply's), <= 16 samples
    __shared__ float pqr[GNA ? 16 * 16 * MT * 3 : 1];
    if constexpr (GNA) {
        const int gs = C / a.gG;
        const float cnt = (float)a.Dz * a.Dy * a.Dx * gs;
        for (int i = threadIdx.x; i < a.N * C; i += 256) {
            const int n = i / C, c = i % C, g = c / gs;
            const float sum = a.gstats[((size_t)n * a.gG + g) * 2], sq = a.gstats[((size_t)n * a.gG + g) * 2 + 1];
            const float mean = sum / cnt;
            const float rstd = rsqrtf(fmaxf(sq / cnt - mean * mean, 0.f) + a.geps);
            const float m1 = a.gred[((size_t)n * a.gG + g) * 2] / cnt, m2 = a.gred[((size_t)n * a.gG + g) * 2 + 1] / cnt;
            pqr[i * 3] = rstd * a.ggamma[c];
            pqr[i * 3 + 1] = -rstd * rstd * m2;
            pqr[i * 3 + 2] = rstd * (mean * rstd * m2 - m1);
        }
        if (blockIdx.x == 0) {                              // GroupNorm parameter gradients: sum over the samples, fixed order
            for (int c = threadIdx.x; c < C; c += 256) {
                float db = 0.f, dg = 0.f;
                for (int k = 0; k < a.N; ++k) { db += a.gchan[((size_t)k * C + c) * 2]; dg += a.gchan[((size_t)k * C + c) * 2 + 1]; }
                a.gdbeta[c] = db;
                a.gdgamma[c] = dg;
            }
        }
        __syncthreads();
    }

    const int xch = (a.Dx + 31) / 32;
    const long long nchunks = (long long)a.N * a.Dz * a.Dy * xch;
    // transposed-read lane constants: lane (g, q', p4) reads voxel 4g + q' (and +16), columns 4 p4 .. +3 of a 16-column tile
    const int g = lane >> 4, li = lane & 15, qp = li >> 2, p4 = li & 3;
    const int vtr = 4 * g + qp;
    const char* lcopy = (qp & 1) ? lb : la;
    const int boff = (p4 >> 1) * THIN_ROWP + 4 * (vtr - (qp & 1)) + (p4 & 1) * 8;     // + 2 nt * THIN_ROWP per tile
    const int doff = vtr * PD + p4 * 8;                                                // + 32 mt per tile

    // the loads of chunk i + 1 are issued before chunk i is consumed (register double buffer: 2 KB per wave is not enough in flight)
    // per-lane constants of the thin-row loads (entry idx = 64 k + lane -> neighbour row r9 = idx / 36, position p = idx % 36)
    int tdz[6], tdy[6], tpp[6];
    bool tlive[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int idx = k * 64 + lane;
        const int r9 = idx / 36;
        tdz[k] = r9 / 3 - 1; tdy[k] = r9 % 3 - 1; tpp[k] = idx % 36;
        tlive[k] = idx < 9 * 36 && tpp[k] < 34;
    }
    const unsigned uxch = (unsigned)xch, uDy = (unsigned)a.Dy, uDz = (unsigned)a.Dz;
    constexpr int KP = (MT + 1) / 2;                        // 16-byte pieces per lane and half chunk
    struct Regs {
        uint4 d[2][KP];                                     // the dense chunk (GNA: x)
        uint4 y[GNA ? 2 : 1][GNA ? KP : 1], ad[GNA ? 2 : 1][GNA ? KP : 1];      // GNA: dyh, residual-path gradient
        uint32_t t[6];
        int n, x0;
    };
    auto fetch = [&](unsigned ch, Regs& R) {                // (32-bit index arithmetic: the loop is issue-bound)
        const bool live = ch < (unsigned)nchunks;
        unsigned r = live ? ch : 0u;
        unsigned qd = fdiv(r, a.fdx);                      // (three float-reciprocal division sequences per chunk before: ~80 of its ~280 instructions)
        const int xc = (int)(r - qd * uxch); r = qd;
        qd = fdiv(r, a.fdy);
        const int y = (int)(r - qd * uDy); r = qd;
        qd = fdiv(r, a.fdz);
        const int z = (int)(r - qd * uDz);
        const int n = (int)qd;
        const int x0 = xc * 32;
        R.n = n;
        R.x0 = x0;
        // dense chunk: lane (v16, q) of half h -> voxel x0 + 16 h + v16, pieces q, q + 4, ...
        const size_t rowo = ((((size_t)n * a.Dz + z) * a.Dy + y) * a.Dx) * C;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int x = x0 + 16 * h + v16;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int pc = k * 4 + q;
                // (branch-free: a conditional load compiles to a branch with its own s_waitcnt - clamp the address, select the value)
                const bool okd = live && x < a.Dx && pc < 2 * MT;
                const size_t eo = rowo + (okd ? (size_t)x * C + pc * 8 : (size_t)0);
                const uint4 val = *reinterpret_cast<const uint4*>((GNA ? a.gx : a.dense) + eo);
                R.d[h][k] = make_uint4(okd ? val.x : 0u, okd ? val.y : 0u, okd ? val.z : 0u, okd ? val.w : 0u);
                if constexpr (GNA) {
                    const uint4 vy = *reinterpret_cast<const uint4*>(a.gdyh + eo);
                    R.y[h][k] = make_uint4(okd ? vy.x : 0u, okd ? vy.y : 0u, okd ? vy.z : 0u, okd ? vy.w : 0u);
                    const uint4 va = a.gadd ? *reinterpret_cast<const uint4*>(a.gadd + eo) : make_uint4(0u, 0u, 0u, 0u);
                    R.ad[h][k] = make_uint4(okd ? va.x : 0u, okd ? va.y : 0u, okd ? va.z : 0u, okd ? va.w : 0u);
                }
            }
        }
        // thin rows: entry (r9, p) = T[z + dz - 1][y + dy - 1][x0 - 1 + p][0..1], p < 36 (34 used)
        const uint32_t* tn = a.thin + (size_t)n * a.Dz * a.Dy * a.Dx;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int zz = z + tdz[k], yy = y + tdy[k], xx = x0 - 1 + tpp[k];
            bool ok = live && tlive[k];
            if constexpr (CIRC) {                           // (branch-free; x0 - 1 + p < Dx + 34 < 3 Dx: the host sends Dx < 17 elsewhere)
                zz += zz < 0 ? a.Dz : 0; zz -= zz >= a.Dz ? a.Dz : 0;
                yy += yy < 0 ? a.Dy : 0; yy -= yy >= a.Dy ? a.Dy : 0;
                xx += xx < 0 ? a.Dx : 0;
                xx -= xx >= a.Dx ? a.Dx : 0;
                xx -= xx >= a.Dx ? a.Dx : 0;
            } else {
                ok = ok && (unsigned)zz < (unsigned)a.Dz && (unsigned)yy < (unsigned)a.Dy && (unsigned)xx < (unsigned)a.Dx;
            }
            const uint32_t tv = tn[ok ? ((unsigned)zz * (unsigned)a.Dy + (unsigned)yy) * (unsigned)a.Dx + (unsigned)xx : 0u];
            R.t[k] = ok ? tv : 0u;
        }
    };
    const unsigned cstep = gridDim.x * 4u;
    Regs cur, nxt;
    unsigned ch = blockIdx.x * 4u + wave;
    fetch(ch, cur);
    for (; ch < (unsigned)nchunks; ch += cstep) {
        fetch(ch + cstep, nxt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the previous chunk's transposed reads are done)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < (MT + 1) / 2; ++k) {
                const int pc = k * 4 + q;
                uint4 val = cur.d[h][k];
                if constexpr (GNA) {                        // dx = dyh * P + x * Q + R (+ add), rounded to bf16 like the tensor it replaces
                    const int xv = cur.x0 + 16 * h + v16;
                    const bool inside = xv < a.Dx && pc < 2 * MT;
                    Piece<bf16_t> px, pd, pa;
                    px.load(val);
                    pd.load(cur.y[h][k]);
                    pa.load(cur.ad[h][k]);
                    const float* pq = pqr + ((size_t)cur.n * C + (pc < 2 * MT ? pc * 8 : 0)) * 3;
#pragma unroll
                    for (int j = 0; j < 8; ++j) px.f[j] = inside ? fmaf(pd.f[j], pq[j * 3], fmaf(px.f[j], pq[j * 3 + 1], pq[j * 3 + 2])) + pa.f[j] : 0.f;
                    val = px.store();
                }
                if (pc < 2 * MT) *reinterpret_cast<uint4*>(td + (16 * h + v16) * PD + pc * 16) = val;
            }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k * 64 + lane < 9 * 36) {
                const int ro = ((tdz[k] + 1) * 3 + tdy[k] + 1) * THIN_ROWP + tpp[k] * 4;
                *reinterpret_cast<uint32_t*>(la + ro) = cur.t[k];
                if (tpp[k] >= 1) *reinterpret_cast<uint32_t*>(lb + ro - 4) = cur.t[k];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8 bt[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) bt[t] = tr16(lcopy + boff + 2 * t * THIN_ROWP, 64);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bf16x8 ad = tr16(td + doff + 32 * m, 16 * PD);
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, bt[t], acc[m][t], 0, 0, 0);
        }
        cur = nxt;
    }
    // workgroup slab [C][80]: the waves add their tiles in wave order through LDS (fixed order)
    float* f = reinterpret_cast<float*>(lds);
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const float e0 = acc[m][t][0], e1 = acc[m][t][1], e2 = acc[m][t][2], e3 = acc[m][t][3];
                    const float e[4] = {e0, e1, e2, e3};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int idx = (m * 16 + 4 * q + k) * THIN_COLS + t * 16 + v16;
                        f[idx] = w == 0 ? e[k] : f[idx] + e[k];
                    }
                }
        }
    }
    __syncthreads();
    float* slab = a.slabs + (size_t)blockIdx.x * (C * THIN_COLS);
    for (int i = threadIdx.x; i < C * THIN_COLS; i += 256) slab[i] = f[i];
}

// G[c][col] = sum over the slabs (fixed order), scattered into the master layout.
//   mode 0 (conv_in):  dw[t][c][j] (cin = 2) = G[c][8 (t / 3) + 2 (t % 3) + j],  dbias[c] = G[c][72]
//   mode 1 (conv_out): dw[t][0][c]           = G[c][column of tap 26 - t, j = 0]
__global__ void __launch_bounds__(256) wgrad_thin_reduce_kernel(const float* __restrict__ slabs, int nslabs, int C, int mode, int cin,
                                                               float* __restrict__ dw, float* __restrict__ dbias) {
    const int e = threadIdx.x & 15, l = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + e;
    const size_t per = (size_t)C * THIN_COLS;
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
    if (l != 0) return;
    float tot = 0.f;
    for (int k = 0; k < 16; ++k) tot += sm[k * 16 + e];
    const int c = idx / THIN_COLS, col = idx % THIN_COLS;
    if (col == 72) {
        if (mode == 0 && dbias) dbias[c] = tot;
        return;
    }
    const int r9 = col >> 3, cp = col & 7;
    if (r9 >= 9 || cp >= 6) return;
    const int t = r9 * 3 + (cp >> 1), j = cp & 1;
    if (mode == 0) {
        if (j < cin) dw[((size_t)t * C + c) * cin + j] = tot;
    } else if (j == 0) {
        dw[(size_t)(26 - t) * C + c] = tot;
    }
}

// persistent workgroups (4 waves each, one chunk per wave in flight ahead of the one it consumes).  VDM4CDM_THIN_WGS: A/B
static int thin_grid(long long nchunks) {
    static const int cap = [] { const char* e = getenv("VDM4CDM_THIN_WGS"); const int v = e ? atoi(e) : 512; return v > 0 ? v : 512; }();
    long long want = (nchunks + 3) / 4;
    if (want > cap) want = cap;
    return (int)(want < 1 ? 1 : want);
}

// does the thin kernel cover this weight gradient?  returns the mode (0: thin input, 1: thin output) or -1
int wgrad_thin_mode(int dtype, int ksize, int stride, int upsample, int cin, int cout, bool want_bias, bool accumulate) {
    if (dtype != VDM_BF16 || ksize != 3 || stride != 1 || upsample || accumulate) return -1;
    if (cin >= 1 && cin <= 2 && (cout == 16 || cout == 32 || cout == 64)) return 0;
    if (cout == 1 && !want_bias && (cin == 16 || cin == 32 || cin == 64)) return 1;
    return -1;
}

static size_t thin_slab_bytes(int n, int od, int oh, int ow, int cdense) {
    const long long nchunks = (long long)n * od * oh * ((ow + 31) / 32);
    return (size_t)thin_grid(nchunks) * cdense * THIN_COLS * sizeof(float);
}

size_t wgrad_thin_workspace_bytes(int n, int od, int oh, int ow, int cdense) {      // slabs + the compacted thin tensor
    return thin_slab_bytes(n, od, oh, ow, cdense) + (size_t)n * od * oh * ow * sizeof(uint32_t);
}

int launch_wgrad_thin(int mode, const void* x, const void* dout, int n, int od, int oh, int ow, int cin, int cout, int circular, float* dw,
                      float* dbias, void* workspace, size_t workspace_bytes, hipStream_t s) {
    ThinArgs a{};
    a.dense = (const bf16_t*)(mode == 0 ? dout : x);
    const long long nvox = (long long)n * od * oh * ow;
    uint32_t* compact = reinterpret_cast<uint32_t*>((char*)workspace + thin_slab_bytes(n, od, oh, ow, mode == 0 ? cout : cin));
    a.thin = compact;
    a.N = n; a.Dz = od; a.Dy = oh; a.Dx = ow; a.C = mode == 0 ? cout : cin; a.circular = circular;
    a.slabs = (float*)workspace;
    if (workspace_bytes < wgrad_thin_workspace_bytes(n, od, oh, ow, a.C)) {
        set_error("conv_wgrad: workspace too small (vdm_conv_wgrad_workspace_bytes)");
        return VDM_ERR_ARG;
    }
    const long long nchunks = (long long)n * od * oh * ((ow + 31) / 32);
    if (nchunks + 4 * 4096 >= 0x7fffffffLL || nvox >= 0x7fffffffLL) { set_error("conv_wgrad: tensor too large for the thin-side kernel"); return VDM_ERR_ARG; }
    a.fdx = make_fastdiv((uint32_t)((ow + 31) / 32)); a.fdy = make_fastdiv((uint32_t)oh); a.fdz = make_fastdiv((uint32_t)od);
    const int grid = thin_grid(nchunks);
    const long long cb = (nvox + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(thin_compact_kernel, dim3((unsigned)(cb > 4096 ? 4096 : cb)), dim3(256), 0, s, (const bf16_t*)(mode == 0 ? x : dout), nvox, compact);
    if (circular) {
        switch (a.C) {
            case 16: hipLaunchKernelGGL((wgrad_thin_kernel<1, true>), dim3(grid), dim3(256), 0, s, a); break;
            case 32: hipLaunchKernelGGL((wgrad_thin_kernel<2, true>), dim3(grid), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((wgrad_thin_kernel<4, true>), dim3(grid), dim3(256), 0, s, a); break;
        }
    } else {
        switch (a.C) {
            case 16: hipLaunchKernelGGL((wgrad_thin_kernel<1, false>), dim3(grid), dim3(256), 0, s, a); break;
            case 32: hipLaunchKernelGGL((wgrad_thin_kernel<2, false>), dim3(grid), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((wgrad_thin_kernel<4, false>), dim3(grid), dim3(256), 0, s, a); break;
        }
    }
    VDM_LAUNCH_CHECK("wgrad_thin_kernel");
    hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(a.C * THIN_COLS / 16), dim3(256), 0, s, (const float*)workspace, grid, a.C, mode, cin, dw, dbias);
    VDM_LAUNCH_CHECK("wgrad_thin_reduce_kernel");
    return VDM_OK;
}

// conv_in's weight gradient with the GroupNorm backward apply pass that produces its dense operand folded in (see ThinArgs)
int launch_wgrad_thin_gna(const ThinArgs& g, const void* thin_x, int cin, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                          hipStream_t s) {
    ThinArgs a = g;
    if (workspace_bytes < wgrad_thin_workspace_bytes(a.N, a.Dz, a.Dy, a.Dx, a.C)) {
        set_error("gn_bwd_apply_wgrad_thin: workspace too small (vdm_conv_wgrad_workspace_bytes of conv_in)");
        return VDM_ERR_ARG;
    }
    const long long nvox = (long long)a.N * a.Dz * a.Dy * a.Dx;
    const long long nchunks = (long long)a.N * a.Dz * a.Dy * ((a.Dx + 31) / 32);
    if (nchunks + 4 * 4096 >= 0x7fffffffLL || nvox >= 0x7fffffffLL) { set_error("gn_bwd_apply_wgrad_thin: tensor too large"); return VDM_ERR_ARG; }
    a.fdx = make_fastdiv((uint32_t)((a.Dx + 31) / 32)); a.fdy = make_fastdiv((uint32_t)a.Dy); a.fdz = make_fastdiv((uint32_t)a.Dz);
    uint32_t* compact = reinterpret_cast<uint32_t*>((char*)workspace + thin_slab_bytes(a.N, a.Dz, a.Dy, a.Dx, a.C));
    a.slabs = (float*)workspace;
    a.thin = compact;
    const int grid = thin_grid(nchunks);
    const long long cb = (nvox + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(thin_compact_kernel, dim3((unsigned)(cb > 4096 ? 4096 : cb)), dim3(256), 0, s, (const bf16_t*)thin_x, nvox, compact);
    if (a.circular) {
        switch (a.C) {
            case 16: hipLaunchKernelGGL((wgrad_thin_kernel<1, true, true>), dim3(grid), dim3(256), 0, s, a); break;
            case 32: hipLaunchKernelGGL((wgrad_thin_kernel<2, true, true>), dim3(grid), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((wgrad_thin_kernel<4, true, true>), dim3(grid), dim3(256), 0, s, a); break;
        }
    } else {
        switch (a.C) {
            case 16: hipLaunchKernelGGL((wgrad_thin_kernel<1, false, true>), dim3(grid), dim3(256), 0, s, a); break;
            case 32: hipLaunchKernelGGL((wgrad_thin_kernel<2, false, true>), dim3(grid), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((wgrad_thin_kernel<4, false, true>), dim3(grid), dim3(256), 0, s, a); break;
        }
    }
    VDM_LAUNCH_CHECK("wgrad_thin_kernel(gna)");
    hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(a.C * THIN_COLS / 16), dim3(256), 0, s, (const float*)workspace, grid, a.C, 0, cin, dw, dbias);
    VDM_LAUNCH_CHECK("wgrad_thin_reduce_kernel");
    return VDM_OK;
}

}  // namespace vdm

using namespace vdm;

extern "C" int vdm_gn_bwd_apply_wgrad_thin(const void* x, int c, int n, int od, int oh, int ow, int groups, const float* stats,
                                           const float* gamma, float eps, const void* dyh, const float* red, const float* chan,
                                           const void* add, const void* thin_x, int thin_c, int circular, float* dgamma, float* dbeta,
                                           float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    VDM_REQUIRE(x && stats && gamma && dyh && red && chan && thin_x && dgamma && dbeta && dw && workspace, "gn_bwd_apply_wgrad_thin: NULL pointer");
    VDM_REQUIRE(c == 16 || c == 32 || c == 64, "gn_bwd_apply_wgrad_thin: 16, 32 or 64 dense channels (got %d)", c);
    VDM_REQUIRE(n > 0 && n <= 16 && od > 0 && oh > 0 && ow > 0 && od <= 4096 && oh <= 4096 && ow <= 4096, "gn_bwd_apply_wgrad_thin: bad n / dims");
    VDM_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0, "gn_bwd_apply_wgrad_thin: %d channels / %d groups", c, groups);
    VDM_REQUIRE(thin_c >= 1 && thin_c <= 2, "gn_bwd_apply_wgrad_thin: the thin side has 1 or 2 channels (got %d)", thin_c);
    VDM_REQUIRE(!circular || ow >= 17, "gn_bwd_apply_wgrad_thin: circular padding needs at least 17 voxels along x");
    ThinArgs a{};
    a.N = n; a.Dz = od; a.Dy = oh; a.Dx = ow; a.C = c; a.circular = circular;
    a.gx = (const bf16_t*)x; a.gdyh = (const bf16_t*)dyh; a.gadd = (const bf16_t*)add;
    a.gstats = stats; a.ggamma = gamma; a.gred = red; a.gchan = chan; a.gdgamma = dgamma; a.gdbeta = dbeta; a.gG = groups; a.geps = eps;
    return launch_wgrad_thin_gna(a, thin_x, thin_c, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream);
}
