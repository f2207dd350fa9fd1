// attention.hip - fused self-attention core of the mid-level attention block (CUNet(mid_attn=True, n_attention_heads); spec D13):
//     out[i] = sum_j softmax_j(scale * q_i . k_j) v_j      over all voxels j of the level, per sample and head,
// forward and backward, on the matrix cores, WITHOUT the [N, heads, V, V] score tensor (0.5 GB at 16^3 voxels, 4 heads, batch 2).
// Replaces the four batched library GEMMs + row-softmax kernels of rounds 1-2 (reference call sites: trainSFM_c_uc_from_field_name.py:61,
// 104-118 `mid_attn=True`; notebook frame blocks.py:169-170 `x = self.attention_blocks[i](x)`).
//
// Layout trick: no LDS and no transposed reads in the main loops.  An MFMA 16x16 result tile leaves lane (c = lane & 15, g = lane >> 4)
// with rows 4g .. 4g+3 of column c.  Computing the score tile TRANSPOSED - S^T[key][query] = K Q^T with the keys as rows - leaves a
// lane with 4 consecutive keys of ONE query: exactly the k-slots a lane must supply for the B operand of the next product
// (O^T[d][query] += V^T[d][keys] P^T[keys][query]), where any key order is allowed as long as the A operand uses the same one.  The A
// operand then is 4 (fp32) or 2 x 4 (bf16: two key tiles per K = 32) CONSECUTIVE keys of one row of V^T - a plain 8 / 16-byte global
// load from a head-major transposed copy.  So the kernels read q, k, v row-major [N][H][V][hd] and q^T, k^T, v^T, dO^T [N][H][hd][V]
// (written once by attn_split_heads_kernel; 4 MB each at 16^3 voxels - L2 resident), every wave runs independently (no barrier), and
// softmax statistics live per lane (lane <-> query).
// Arithmetic: operands in the activation dtype (bf16 -> v_mfma_f32_16x16x32_bf16, fp32 -> exact v_mfma_f32_16x16x4_f32), scores,
// softmax and all accumulators in fp32.  Deterministic: dQ comes from its own pass over the keys (no float atomics).
#include "common.h"

namespace vdm {

struct AttnArgs {
    const void* q; const void* k; const void* v;          // row-major [N][H][V][hd]
    const void* qt; const void* kt; const void* vt;       // transposed [N][H][hd][V]
    const void* doh; const void* dot;                     // dO row-major / transposed (backward)
    float* lse;                                           // [N][H][V]: log-sum-exp of the scaled scores
    const float* dsum;                                    // [N][H][V]: sum_d dO * O (backward)
    void* out;                                            // fwd: [N][V][H*hd];  bwd: dqkv [N][V][3][H*hd]
    int N, H, V;
    float scale;
};

template <typename T> __device__ __forceinline__ void attn_mma(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void attn_mma<bf16_t>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void attn_mma<float>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
}

// per-dtype geometry: one attn_mma consumes KB reduction slots, EPL per lane group g
template <typename T, int HD> struct AG {
    static constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    static constexpr int NF = (HD + KB - 1) / KB;           // fragments along the head dimension (zero padded past HD)
    static constexpr int KT = KB / 16;                      // 16-row tiles that fill the K of one product over keys / queries (bf16 2, fp32 1)
    static constexpr int ND = HD / 16;                      // 16-wide tiles of the head dimension
    static_assert(HD % 16 == 0, "head_dim must be a multiple of 16");
};

// row-major operand fragment: row `r` (clamped to a valid row), head-dim chunk f: elements [f*KB + g*EPL, +EPL); zero past HD
template <typename T, int HD>
__device__ __forceinline__ uint4 row_frag(const T* base, int r, int rmax, int f, int g) {
    using A = AG<T, HD>;
    const int e0 = f * A::KB + g * A::EPL;
    if (e0 >= HD) return make_uint4(0u, 0u, 0u, 0u);
    const int rr = r < rmax ? r : rmax - 1;
    return *reinterpret_cast<const uint4*>(base + (size_t)rr * HD + e0);
}

// transposed operand fragment: row d of X^T [hd][V], the lane's k-slots = 4 consecutive columns c0 + 4g .. of each of the KT tiles
// (bf16: tiles c0, c0 + 16 -> two 8-byte loads; fp32: one 16-byte load).  Columns are clamped (their partner in B is zero there).
template <typename T, int HD>
__device__ __forceinline__ uint4 col_frag(const T* base_t, int d, int V, int c0, int g) {
    const T* row = base_t + (size_t)d * V;
    if constexpr (sizeof(T) == 2) {
        int a0 = c0 + 4 * g, a1 = c0 + 16 + 4 * g;
        a0 = a0 + 4 <= V ? a0 : V - 4;
        a1 = a1 + 4 <= V ? a1 : V - 4;
        const uint2 lo = *reinterpret_cast<const uint2*>(row + a0), hi = *reinterpret_cast<const uint2*>(row + a1);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else {
        int a0 = c0 + 4 * g;
        a0 = a0 + 4 <= V ? a0 : V - 4;
        return *reinterpret_cast<const uint4*>(row + a0);
    }
}

// the lane's KT x 4 fp32 values (tile t, row 4g + c) -> B operand in the matching slot order
template <typename T, int KT>
__device__ __forceinline__ uint4 pack_slots(const f32x4 (&p)[KT]) {
    // (elements are copied to scalars first: __builtin_bit_cast applied directly to an ext-vector element expression `p[0][1]` was
    // compiled as element 0 for all four - seen in the ISA: the four fp32 MFMAs of a product all took the same B register)
    const float a0 = p[0][0], a1 = p[0][1], a2 = p[0][2], a3 = p[0][3];
    if constexpr (sizeof(T) == 2) {
        const float b0 = p[KT - 1][0], b1 = p[KT - 1][1], b2 = p[KT - 1][2], b3 = p[KT - 1][3];
        return make_uint4(pack_bf16x2(a0, a1), pack_bf16x2(a2, a3), pack_bf16x2(b0, b1), pack_bf16x2(b2, b3));
    } else {
        return make_uint4(__float_as_uint(a0), __float_as_uint(a1), __float_as_uint(a2), __float_as_uint(a3));
    }
}

// sum / max over the four lane groups g of one column c (lanes c, c+16, c+32, c+48)
__device__ __forceinline__ float groups_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float groups_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

template <typename T, int HD, int CH>
__device__ __forceinline__ void store_chunk(T* dst, const f32x4& v) {          // 4 consecutive channels
    if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    else *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------------------------
// forward: one wave = 16 queries; grid (ceil(V / 64), H, N)
// ---------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const AttnArgs a) {
    using A = AG<T, HD>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, n = blockIdx.z, V = a.V;
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    if (i0 >= V) return;                                     // (whole wave; no barriers in this kernel)
    const size_t hb = (size_t)(n * a.H + h);
    const T* Q = reinterpret_cast<const T*>(a.q) + hb * V * HD;
    const T* K = reinterpret_cast<const T*>(a.k) + hb * V * HD;
    const T* Vt = reinterpret_cast<const T*>(a.vt) + hb * HD * V;
    uint4 qf[A::NF];                                         // B operand of S^T = K Q^T: column = query i0 + c
#pragma unroll
    for (int f = 0; f < A::NF; ++f) qf[f] = row_frag<T, HD>(Q, i0 + c, V, f, g);
    f32x4 o[A::ND];
#pragma unroll
    for (int d = 0; d < A::ND; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -3.0e38f, l = 0.f;
    for (int j0 = 0; j0 < V; j0 += 16 * A::KT) {
        f32x4 s[A::KT];
#pragma unroll
        for (int t = 0; t < A::KT; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < A::NF; ++f) attn_mma<T>(s[t], row_frag<T, HD>(K, j0 + t * 16 + c, V, f, g), qf[f]);
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < A::KT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = j0 + t * 16 + 4 * g + e < V;
                s[t][e] = ok ? s[t][e] * a.scale : -3.0e38f;
                mx = fmaxf(mx, s[t][e]);
            }
        mx = groups_max(mx);
        const float mn = fmaxf(m, mx), alpha = __expf(m - mn);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < A::KT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[t][e] = __expf(s[t][e] - mn);               // (masked keys: exp(-3e38 - mn) = 0)
                rs += s[t][e];
            }
        l = l * alpha + groups_sum(rs);
        m = mn;
        const uint4 pf = pack_slots<T, A::KT>(s);
#pragma unroll
        for (int d = 0; d < A::ND; ++d) {
            o[d] *= alpha;
            attn_mma<T>(o[d], col_frag<T, HD>(Vt, d * 16 + c, V, j0, g), pf);
        }
    }
    const int i = i0 + c;
    if (i < V) {
        const float inv = 1.0f / l;
        T* out = reinterpret_cast<T*>(a.out) + ((size_t)n * V + i) * (a.H * HD) + h * HD;
#pragma unroll
        for (int d = 0; d < A::ND; ++d) store_chunk<T, HD, 0>(out + d * 16 + 4 * g, o[d] * inv);
        if (g == 0 && a.lse) a.lse[hb * V + i] = m + __logf(l);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, keys: one wave = 16 keys j; loops over the queries.  dV^T[d][j] = sum_i dO^T[d][i] P[i][j], dK^T[d][j] = sum_i Q^T[d][i] dS[i][j]
// with S[i][j] = Q K^T (queries as rows), P = exp(scale S - lse_i), dP = dO V^T, dS = P (dP - dsum_i) scale.
// ---------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ void __launch_bounds__(256) attn_bwd_kv_kernel(const AttnArgs a) {
    using A = AG<T, HD>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, n = blockIdx.z, V = a.V;
    const int j0 = (blockIdx.x * 4 + wave) * 16;
    if (j0 >= V) return;
    const size_t hb = (size_t)(n * a.H + h);
    const T* Q = reinterpret_cast<const T*>(a.q) + hb * V * HD;
    const T* K = reinterpret_cast<const T*>(a.k) + hb * V * HD;
    const T* Vr = reinterpret_cast<const T*>(a.v) + hb * V * HD;
    const T* dO = reinterpret_cast<const T*>(a.doh) + hb * V * HD;
    const T* Qt = reinterpret_cast<const T*>(a.qt) + hb * HD * V;
    const T* dOt = reinterpret_cast<const T*>(a.dot) + hb * HD * V;
    const float* lse = a.lse + hb * V;
    const float* dsum = a.dsum + hb * V;
    uint4 kf[A::NF], vf[A::NF];                              // B operands: column = key j0 + c
#pragma unroll
    for (int f = 0; f < A::NF; ++f) { kf[f] = row_frag<T, HD>(K, j0 + c, V, f, g); vf[f] = row_frag<T, HD>(Vr, j0 + c, V, f, g); }
    const bool key_ok = j0 + c < V;
    f32x4 dk[A::ND], dv[A::ND];
#pragma unroll
    for (int d = 0; d < A::ND; ++d) dk[d] = dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < V; i0 += 16 * A::KT) {
        f32x4 p[A::KT], ds[A::KT];
#pragma unroll
        for (int t = 0; t < A::KT; ++t) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < A::NF; ++f) {
                attn_mma<T>(s, row_frag<T, HD>(Q, i0 + t * 16 + c, V, f, g), kf[f]);       // rows = queries
                attn_mma<T>(dp, row_frag<T, HD>(dO, i0 + t * 16 + c, V, f, g), vf[f]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + t * 16 + 4 * g + e;
                const bool ok = key_ok && i < V;
                const int ic = i < V ? i : V - 1;
                const float pe = ok ? __expf(s[e] * a.scale - lse[ic]) : 0.f;
                p[t][e] = pe;
                ds[t][e] = pe * (dp[e] - dsum[ic]) * a.scale;
            }
        }
        const uint4 pf = pack_slots<T, A::KT>(p), dsf = pack_slots<T, A::KT>(ds);
#pragma unroll
        for (int d = 0; d < A::ND; ++d) {
            attn_mma<T>(dv[d], col_frag<T, HD>(dOt, d * 16 + c, V, i0, g), pf);
            attn_mma<T>(dk[d], col_frag<T, HD>(Qt, d * 16 + c, V, i0, g), dsf);
        }
    }
    if (key_ok) {
        const int C = a.H * HD;
        T* o = reinterpret_cast<T*>(a.out) + ((size_t)n * V + j0 + c) * (3 * C) + h * HD;
#pragma unroll
        for (int d = 0; d < A::ND; ++d) {
            store_chunk<T, HD, 0>(o + C + d * 16 + 4 * g, dk[d]);
            store_chunk<T, HD, 0>(o + 2 * C + d * 16 + 4 * g, dv[d]);
        }
    }
}

// backward, queries: one wave = 16 queries i; loops over the keys.  dQ^T[d][i] = sum_j K^T[d][j] dS[i][j] (scores transposed, as in the forward).
template <typename T, int HD>
__global__ void __launch_bounds__(256) attn_bwd_q_kernel(const AttnArgs a) {
    using A = AG<T, HD>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, n = blockIdx.z, V = a.V;
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    if (i0 >= V) return;
    const size_t hb = (size_t)(n * a.H + h);
    const T* Q = reinterpret_cast<const T*>(a.q) + hb * V * HD;
    const T* K = reinterpret_cast<const T*>(a.k) + hb * V * HD;
    const T* Vr = reinterpret_cast<const T*>(a.v) + hb * V * HD;
    const T* dO = reinterpret_cast<const T*>(a.doh) + hb * V * HD;
    const T* Kt = reinterpret_cast<const T*>(a.kt) + hb * HD * V;
    uint4 qf[A::NF], dof[A::NF];
#pragma unroll
    for (int f = 0; f < A::NF; ++f) { qf[f] = row_frag<T, HD>(Q, i0 + c, V, f, g); dof[f] = row_frag<T, HD>(dO, i0 + c, V, f, g); }
    const int ic = i0 + c < V ? i0 + c : V - 1;
    const float lse = a.lse[hb * V + ic], dsm = a.dsum[hb * V + ic];
    f32x4 dq[A::ND];
#pragma unroll
    for (int d = 0; d < A::ND; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < V; j0 += 16 * A::KT) {
        f32x4 ds[A::KT];
#pragma unroll
        for (int t = 0; t < A::KT; ++t) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < A::NF; ++f) {
                attn_mma<T>(s, row_frag<T, HD>(K, j0 + t * 16 + c, V, f, g), qf[f]);       // rows = keys
                attn_mma<T>(dp, row_frag<T, HD>(Vr, j0 + t * 16 + c, V, f, g), dof[f]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = j0 + t * 16 + 4 * g + e < V;
                const float pe = ok ? __expf(s[e] * a.scale - lse) : 0.f;
                ds[t][e] = pe * (dp[e] - dsm) * a.scale;
            }
        }
        const uint4 dsf = pack_slots<T, A::KT>(ds);
#pragma unroll
        for (int d = 0; d < A::ND; ++d) attn_mma<T>(dq[d], col_frag<T, HD>(Kt, d * 16 + c, V, j0, g), dsf);
    }
    if (i0 + c < V) {
        const int C = a.H * HD;
        T* o = reinterpret_cast<T*>(a.out) + ((size_t)n * V + i0 + c) * (3 * C) + h * HD;
#pragma unroll
        for (int d = 0; d < A::ND; ++d) store_chunk<T, HD, 0>(o + d * 16 + 4 * g, dq[d]);
    }
}

// ---------------------------------------------------------------------------------------------
// head split: src[n][v][src_stride] (channels src_off + h*hd ..) -> row-major [n][h][v][hd] and / or transposed [n][h][hd][v].
// One block = 64 voxels of one (n, h); the transposed copy goes through LDS so that both writes are coalesced.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) attn_split_heads_kernel(const T* __restrict__ src, int64_t src_stride, int64_t src_off, int V, int H, int hd,
                                                               T* __restrict__ rm, T* __restrict__ tr) {
    __shared__ T tile[64][129];
    const int h = blockIdx.y, n = blockIdx.z, v0 = blockIdx.x * 64;
    const size_t hb = (size_t)(n * H + h);
    for (int idx = threadIdx.x; idx < 64 * hd; idx += 256) {
        const int v = idx / hd, d = idx - v * hd;
        T val;
        if (v0 + v < V) {
            val = src[((size_t)n * V + v0 + v) * src_stride + src_off + h * hd + d];
            if (rm) rm[(hb * V + v0 + v) * hd + d] = val;
        } else {
            st_elem<T>(&val, 0.f);
        }
        tile[v][d] = val;
    }
    if (!tr) return;
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * hd; idx += 256) {
        const int d = idx >> 6, v = idx & 63;
        if (v0 + v < V) tr[(hb * hd + d) * V + v0 + v] = tile[v][d];
    }
}

// out[n][h][v] = sum_d a[n][v][h*hd + d] * b[n][v][h*hd + d]   (dsum = rowsum(dO * O) of the attention backward)
template <typename T>
__global__ void __launch_bounds__(256) attn_rowdot_kernel(const T* __restrict__ a, const T* __restrict__ b, int N, int V, int H, int hd,
                                                          float* __restrict__ out) {
    const int64_t total = (int64_t)N * V * H;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int h = (int)(t % H);
        const int64_t nv = t / H;
        const T* pa = a + nv * (H * hd) + h * hd;
        const T* pb = b + nv * (H * hd) + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(ld_elem<T>(pa + d), ld_elem<T>(pb + d), s);
        const int64_t n = nv / V, v = nv % V;
        out[((size_t)n * H + h) * V + v] = s;
    }
}

template <typename T, int HD>
static int launch_attn(int which, const AttnArgs& a, hipStream_t s) {
    const dim3 grid((a.V + 63) / 64, a.H, a.N);
    if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<T, HD>), grid, dim3(256), 0, s, a);
    else if (which == 1) hipLaunchKernelGGL((attn_bwd_kv_kernel<T, HD>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_bwd_q_kernel<T, HD>), grid, dim3(256), 0, s, a);
    VDM_LAUNCH_CHECK("attention kernel");
    return VDM_OK;
}

template <typename T>
static int launch_attn_hd(int which, const AttnArgs& a, int hd, hipStream_t s) {
    switch (hd) {
        case 16: return launch_attn<T, 16>(which, a, s);
        case 32: return launch_attn<T, 32>(which, a, s);
        case 64: return launch_attn<T, 64>(which, a, s);
        case 96: return launch_attn<T, 96>(which, a, s);
        case 128: return launch_attn<T, 128>(which, a, s);
    }
    set_error("attention: head_dim %d is not built (16, 32, 64, 96, 128)", hd);
    return VDM_ERR_UNSUPPORTED;
}

static int attn_check(int n, int64_t voxels, int heads, int head_dim, int dtype, const char* who) {
    VDM_REQUIRE(dtype == VDM_F32 || dtype == VDM_BF16, "%s: bad dtype %d", who, dtype);
    VDM_REQUIRE(n > 0 && heads > 0 && n <= 65535 && heads <= 65535, "%s: bad batch / heads", who);
    VDM_REQUIRE(voxels >= 4 && voxels % 4 == 0 && voxels < (1ll << 24), "%s: the voxel count must be a multiple of 4 below 2^24 (got %lld)", who,
                (long long)voxels);
    VDM_REQUIRE(head_dim == 16 || head_dim == 32 || head_dim == 64 || head_dim == 96 || head_dim == 128,
                "%s: head_dim %d is not built (16, 32, 64, 96, 128)", who, head_dim);
    return VDM_OK;
}

}  // namespace vdm

using namespace vdm;

extern "C" int vdm_attn_split_heads(const void* src, int64_t src_stride, int64_t src_offset, int n, int64_t voxels, int heads, int head_dim,
                                    int dtype, void* rowmajor, void* transposed, void* stream) {
    int e = attn_check(n, voxels, heads, head_dim, dtype, "attn_split_heads");
    if (e) return e;
    VDM_REQUIRE(src && (rowmajor || transposed) && src_stride >= (int64_t)heads * head_dim && src_offset >= 0, "attn_split_heads: bad arguments");
    const dim3 grid((unsigned)((voxels + 63) / 64), heads, n);
    if (dtype == VDM_F32)
        hipLaunchKernelGGL(attn_split_heads_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, src_stride, src_offset,
                           (int)voxels, heads, head_dim, (float*)rowmajor, (float*)transposed);
    else
        hipLaunchKernelGGL(attn_split_heads_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, src_stride, src_offset,
                           (int)voxels, heads, head_dim, (bf16_t*)rowmajor, (bf16_t*)transposed);
    VDM_LAUNCH_CHECK("attn_split_heads_kernel");
    return VDM_OK;
}

extern "C" int vdm_attn_rowdot(const void* a, const void* b, int n, int64_t voxels, int heads, int head_dim, int dtype, float* out, void* stream) {
    int e = attn_check(n, voxels, heads, head_dim, dtype, "attn_rowdot");
    if (e) return e;
    VDM_REQUIRE(a && b && out, "attn_rowdot: NULL pointer");
    const int64_t total = (int64_t)n * voxels * heads;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == VDM_F32)
        hipLaunchKernelGGL(attn_rowdot_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, n, (int)voxels,
                           heads, head_dim, out);
    else
        hipLaunchKernelGGL(attn_rowdot_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, n,
                           (int)voxels, heads, head_dim, out);
    VDM_LAUNCH_CHECK("attn_rowdot_kernel");
    return VDM_OK;
}

extern "C" int vdm_attn_fwd(const void* q, const void* k, const void* vt, int n, int64_t voxels, int heads, int head_dim, int dtype, float scale,
                            void* out, float* lse, void* stream) {
    int e = attn_check(n, voxels, heads, head_dim, dtype, "attn_fwd");
    if (e) return e;
    VDM_REQUIRE(q && k && vt && out, "attn_fwd: NULL pointer");
    AttnArgs a{};
    a.q = q; a.k = k; a.vt = vt; a.out = out; a.lse = lse; a.N = n; a.H = heads; a.V = (int)voxels; a.scale = scale;
    return dtype == VDM_F32 ? launch_attn_hd<float>(0, a, head_dim, (hipStream_t)stream) : launch_attn_hd<bf16_t>(0, a, head_dim, (hipStream_t)stream);
}

extern "C" int vdm_attn_bwd(const void* q, const void* k, const void* v, const void* qt, const void* kt, const void* do_rowmajor,
                            const void* do_transposed, const float* lse, const float* dsum, int n, int64_t voxels, int heads, int head_dim,
                            int dtype, float scale, void* dqkv, void* stream) {
    int e = attn_check(n, voxels, heads, head_dim, dtype, "attn_bwd");
    if (e) return e;
    VDM_REQUIRE(q && k && v && qt && kt && do_rowmajor && do_transposed && lse && dsum && dqkv, "attn_bwd: NULL pointer");
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.qt = qt; a.kt = kt; a.doh = do_rowmajor; a.dot = do_transposed; a.lse = const_cast<float*>(lse); a.dsum = dsum;
    a.out = dqkv; a.N = n; a.H = heads; a.V = (int)voxels; a.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    e = dtype == VDM_F32 ? launch_attn_hd<float>(1, a, head_dim, s) : launch_attn_hd<bf16_t>(1, a, head_dim, s);
    if (e) return e;
    return dtype == VDM_F32 ? launch_attn_hd<float>(2, a, head_dim, s) : launch_attn_hd<bf16_t>(2, a, head_dim, s);
}
