// cond.hip - K6: conditioning embeddings of the score network in one launch (R5 of SURVEY.md section 8a).
//
// Replaces the ATen chain behind  score_model(zt, t=(gamma_t - gamma_min) / (gamma_max - gamma_min), v_conditionings=...)
// [NB vdm_model.py:320-324 -> networks.py:259-265]: per conditioning k
//     c_k = GELU(Linear2(GELU(Linear1(in_k))))        in_t = [sin(1000 t f_i), cos(1000 t f_i)] (width 64), in_v = v (B, 6)
// and the additive injection table of ALL ResNetBlocks at once (spec D4/D7):
//     table[row][w] = sum_k sum_i c_k[row][i] * Wproj_k[w][i]        w over the concatenated output channels of the blocks.
// <1 MFLOP per row: latency only - one launch forward, three backward (all of ~70 ATen launches + 12 GEMMs before); the work of a row is
// spread over table slabs (forward: 64 columns per workgroup, backward: 32) so that no thread walks a long dependent chain.
// Deterministic (fixed summation orders, no atomics).  GELU is the exact erf form (torch.nn.functional.gelu default).
#include "common.h"

namespace vdm {

constexpr int COND_MAX = 4;          // conditionings per call
constexpr int COND_MAXDIM = 256;     // widest hidden layer / input (4 * chs[0] <= 256, i.e. chs[0] <= 64)
constexpr int COND_FSLAB = 64;       // table columns per forward workgroup
constexpr int COND_BSLAB = 32;       // table columns per backward stage-A1 workgroup

struct CondArgs {
    vdm_cond_mlp m[COND_MAX];
    int n, rows, width;
    float* table;                    // fwd out [rows][width]
    float* saved;                    // fwd out / bwd in: per mlp [rows][in_dim + 3 dim] = (input, h1, h2, c)
    const float* dtable;             // bwd in [rows][dtable_stride]
    long long dtable_stride;
    float* scratch;                  // bwd: per mlp [rows][2 dim] = (dh1, dh2), then the stage-A1 slab partials (part_base)
    float* dbias;                    // bwd out (optional) [width] = column sums of dtable
    int nslab;                       // bwd: slabs of COND_BSLAB table columns
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

__host__ __device__ __forceinline__ size_t saved_base(const CondArgs& a, int k) {      // float offset of mlp k inside `saved`
    size_t off = 0;
    for (int q = 0; q < k; ++q) off += (size_t)a.rows * (a.m[q].in_dim + 3 * a.m[q].dim);
    return off;
}
__host__ __device__ __forceinline__ size_t scratch_base(const CondArgs& a, int k) {
    size_t off = 0;
    for (int q = 0; q < k; ++q) off += (size_t)a.rows * 2 * a.m[q].dim;
    return off;
}
// stage-A1 partials: behind the (dh1, dh2) rows; per mlp [rows][nslab][dim]
__host__ __device__ __forceinline__ size_t part_base(const CondArgs& a, int k) {
    size_t off = scratch_base(a, a.n);
    for (int q = 0; q < k; ++q) off += (size_t)a.rows * a.nslab * a.m[q].dim;
    return off;
}

// These kernels are pure latency (< 1 MFLOP per row): what matters is how many dependent memory round trips a thread makes.
// dot_row: the `tpo` threads (sub = 0..tpo-1, adjacent lanes, tpo a power of two <= 4) of one output share the dot product of a
// weight row (global) with an LDS vector: 16-byte loads, 8 independent loads in flight per thread, xor-butterfly over the sub lanes
// (fixed order: deterministic).  Every thread of the group returns the total.
__device__ __forceinline__ float dot_row(const float* __restrict__ w, const float* xs, int len, int sub, int tpo) {
    float acc = 0.f;
    const int nq = (len & 3) == 0 ? len >> 2 : 0;            // rows of the weight matrices are 16-byte aligned iff len % 4 == 0
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (int q0 = sub; q0 < nq; q0 += 8 * tpo) {
        float4 wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u * tpo;
            wv[u] = q < nq ? w4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u * tpo;
            if (q < nq) {
                acc = fmaf(wv[u].x, xs[4 * q], acc); acc = fmaf(wv[u].y, xs[4 * q + 1], acc);
                acc = fmaf(wv[u].z, xs[4 * q + 2], acc); acc = fmaf(wv[u].w, xs[4 * q + 3], acc);
            }
        }
    }
    for (int i = 4 * nq + sub; i < len; i += tpo) acc = fmaf(w[i], xs[i], acc);
    for (int o = 1; o < tpo; o <<= 1) acc += __shfl_xor(acc, o, 64);
    return acc;
}

__device__ __forceinline__ int tpo_for(int dim) { return dim <= 64 ? 4 : (dim <= 128 ? 2 : 1); }

// forward: grid (table slabs of COND_FSLAB columns, rows).  Every workgroup recomputes the two-layer MLPs of its row (24 k FMAs
// per conditioning: cheaper than a second launch or a grid-wide dependency), then its slab of the table; slab 0 writes `saved`.
__global__ void __launch_bounds__(256) cond_table_fwd_kernel(const CondArgs a) {
    const int row = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
    __shared__ __attribute__((aligned(16))) float in_s[COND_MAXDIM], a1[COND_MAXDIM], c[COND_MAX][COND_MAXDIM];
    for (int k = 0; k < a.n; ++k) {
        const vdm_cond_mlp& m = a.m[k];
        float* sv = (a.saved && slab == 0) ? a.saved + saved_base(a, k) + (size_t)row * (m.in_dim + 3 * m.dim) : nullptr;
        if (tid < m.in_dim) {
            float v;
            if (m.sinusoid) {
                const int half = m.in_dim / 2, i = tid % half;
                const float f = expf(-9.210340371976184f * (float)i / (float)half);       // 10000^(-i/half)
                const float arg = 1000.0f * m.input[row] * f;
                v = tid < half ? sinf(arg) : cosf(arg);
            } else {
                v = m.input[(size_t)row * m.in_dim + tid];
            }
            in_s[tid] = v;
            if (sv) sv[tid] = v;
        }
        __syncthreads();
        const int tpo = tpo_for(m.dim), o = tid / tpo, sub = tid % tpo;
        if (o < m.dim) {
            const float h = m.b1[o] + dot_row(m.w1 + (size_t)o * m.in_dim, in_s, m.in_dim, sub, tpo);
            if (sub == 0) {
                if (sv) sv[m.in_dim + o] = h;
                a1[o] = gelu_f(h);
            }
        }
        __syncthreads();
        if (o < m.dim) {
            const float h = m.b2[o] + dot_row(m.w2 + (size_t)o * m.dim, a1, m.dim, sub, tpo);
            if (sub == 0) {
                const float cv = gelu_f(h);
                if (sv) { sv[m.in_dim + m.dim + o] = h; sv[m.in_dim + 2 * m.dim + o] = cv; }
                c[k][o] = cv;
            }
        }
        __syncthreads();
    }
    const int w = slab * COND_FSLAB + tid / 4, sub = tid & 3;      // four threads per table column
    if (w < a.width) {
        float acc = 0.f;
        for (int k = 0; k < a.n; ++k) acc += dot_row(a.m[k].wproj + (size_t)w * a.m[k].dim, c[k], a.m[k].dim, sub, 4);
        if (sub == 0) a.table[(size_t)row * a.width + w] = acc;
    }
}

// backward, stage A1: grid (slabs of COND_BSLAB table columns, rows): part[k][row][slab][i] = sum_{w in slab} dtable[row][w] Wproj_k[w][i].
// Thread (i, ws): column i of the projection (coalesced over i), the slab's rows w = ws, ws + nws, ...; fixed-order LDS fold over ws.
__global__ void __launch_bounds__(256) cond_table_bwd_slabs_kernel(const CondArgs a) {
    const int row = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
    __shared__ float dts[COND_BSLAB];
    __shared__ float red[256];
    const int w0 = slab * COND_BSLAB;
    if (tid < COND_BSLAB) dts[tid] = w0 + tid < a.width ? a.dtable[(size_t)row * a.dtable_stride + w0 + tid] : 0.f;
    __syncthreads();
    const int wn = min(COND_BSLAB, a.width - w0);
    for (int k = 0; k < a.n; ++k) {
        const vdm_cond_mlp& m = a.m[k];
        const int nws = tpo_for(m.dim), i = tid % (256 / nws), ws = tid / (256 / nws);      // 256 / nws >= dim
        float s = 0.f;
        if (i < m.dim) {
            const float* p = m.wproj + (size_t)w0 * m.dim + i;
            for (int j0 = ws; j0 < wn; j0 += 8 * nws) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * nws;
                    v[u] = j < wn ? p[(size_t)j * m.dim] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * nws;
                    if (j < wn) s = fmaf(dts[j], v[u], s);
                }
            }
        }
        red[tid] = s;
        __syncthreads();
        if (ws == 0 && i < m.dim) {
            float tot = 0.f;
            for (int q = 0; q < nws; ++q) tot += red[q * (256 / nws) + i];
            a.scratch[part_base(a, k) + ((size_t)row * a.nslab + slab) * m.dim + i] = tot;
        }
        __syncthreads();
    }
}

// backward, stage A2: one block per (row, mlp): fold the slab partials (fixed order) -> dh2 = dc * gelu'(h2), dh1 = (dh2 W2) * gelu'(h1)
__global__ void __launch_bounds__(256) cond_table_bwd_rows_kernel(const CondArgs a) {
    const int row = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
    __shared__ float dh2[COND_MAXDIM];
    __shared__ float red[256];
    const vdm_cond_mlp& m = a.m[k];
    const float* sv = a.saved + saved_base(a, k) + (size_t)row * (m.in_dim + 3 * m.dim);
    float* sc = a.scratch + scratch_base(a, k) + (size_t)row * 2 * m.dim;
    const int nws = tpo_for(m.dim), per = 256 / nws, i = tid % per, ws = tid / per;
    {   // dc[i] = sum over the slabs
        float s = 0.f;
        if (i < m.dim) {
            const float* p = a.scratch + part_base(a, k) + (size_t)row * a.nslab * m.dim + i;
            for (int j0 = ws; j0 < a.nslab; j0 += 8 * nws) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u * nws;
                    v[u] = j < a.nslab ? p[(size_t)j * m.dim] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
        }
        red[tid] = s;
        __syncthreads();
        if (ws == 0 && i < m.dim) {
            float tot = 0.f;
            for (int q = 0; q < nws; ++q) tot += red[q * per + i];
            const float d = tot * dgelu_f(sv[m.in_dim + m.dim + i]);
            dh2[i] = d;
            sc[m.dim + i] = d;
        }
        __syncthreads();
    }
    float s = 0.f;
    if (i < m.dim) {
        const float* p = m.w2 + i;
        for (int j0 = ws; j0 < m.dim; j0 += 8 * nws) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * nws;
                v[u] = j < m.dim ? p[(size_t)j * m.dim] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * nws;
                if (j < m.dim) s = fmaf(dh2[j], v[u], s);
            }
        }
    }
    red[tid] = s;
    __syncthreads();
    if (ws == 0 && i < m.dim) {
        float tot = 0.f;
        for (int q = 0; q < nws; ++q) tot += red[q * per + i];
        sc[i] = tot * dgelu_f(sv[m.in_dim + i]);
    }
}

// backward, stage B: every parameter-gradient element is a sum over the rows (fixed order).  Flat index over the segments
// [dw1, db1, dw2, db2, dwproj] of each mlp, then dbias[width].
__global__ void __launch_bounds__(256) cond_table_bwd_params_kernel(const CondArgs a, long long total) {
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        long long r = g;
        bool done = false;
        for (int k = 0; k < a.n && !done; ++k) {
            const vdm_cond_mlp& m = a.m[k];
            const size_t svs = m.in_dim + 3 * m.dim;
            const float* sv = a.saved + saved_base(a, k);
            const float* sc = a.scratch + scratch_base(a, k);
            const long long n1 = (long long)m.dim * m.in_dim, n2 = (long long)m.dim * m.dim, np = (long long)a.width * m.dim;
            if (r < n1) {                                   // dw1[j][i] = sum_r dh1[r][j] in[r][i]
                const int j = (int)(r / m.in_dim), i = (int)(r % m.in_dim);
                float s = 0.f;
                for (int q = 0; q < a.rows; ++q) s = fmaf(sc[(size_t)q * 2 * m.dim + j], sv[q * svs + i], s);
                m.dw1[r] = s; done = true; break;
            }
            r -= n1;
            if (r < m.dim) {                                // db1
                float s = 0.f;
                for (int q = 0; q < a.rows; ++q) s += sc[(size_t)q * 2 * m.dim + r];
                m.db1[r] = s; done = true; break;
            }
            r -= m.dim;
            if (r < n2) {                                   // dw2[j][i] = sum_r dh2[r][j] gelu(h1[r][i])
                const int j = (int)(r / m.dim), i = (int)(r % m.dim);
                float s = 0.f;
                for (int q = 0; q < a.rows; ++q) s = fmaf(sc[(size_t)q * 2 * m.dim + m.dim + j], gelu_f(sv[q * svs + m.in_dim + i]), s);
                m.dw2[r] = s; done = true; break;
            }
            r -= n2;
            if (r < m.dim) {                                // db2
                float s = 0.f;
                for (int q = 0; q < a.rows; ++q) s += sc[(size_t)q * 2 * m.dim + m.dim + r];
                m.db2[r] = s; done = true; break;
            }
            r -= m.dim;
            if (r < np) {                                   // dwproj[w][i] = sum_r dtable[r][w] c[r][i]
                const int w = (int)(r / m.dim), i = (int)(r % m.dim);
                float s = 0.f;
                for (int q = 0; q < a.rows; ++q) s = fmaf(a.dtable[(size_t)q * a.dtable_stride + w], sv[q * svs + m.in_dim + 2 * m.dim + i], s);
                m.dwproj[r] = s; done = true; break;
            }
            r -= np;
        }
        if (!done && a.dbias && r < a.width) {
            float s = 0.f;
            for (int q = 0; q < a.rows; ++q) s += a.dtable[(size_t)q * a.dtable_stride + r];
            a.dbias[r] = s;
        }
    }
}

// sampler: table[b][w] = table_t[*step][w] + table_v[b][w]   (row gather by the device-side step counter)
__global__ void __launch_bounds__(256) cond_table_step_kernel(const float* __restrict__ tt, const float* __restrict__ tv,
                                                             const int32_t* __restrict__ step, int rows, int width, float* __restrict__ out) {
    const int s = tt ? *step : 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * width; i += gridDim.x * 256) {
        const int w = i % width;
        out[i] = (tt ? tt[(size_t)s * width + w] : 0.f) + (tv ? tv[i] : 0.f);
    }
}

static int fill(CondArgs& a, const vdm_cond_mlp* mlps, int n, int rows, int width, bool bwd) {
    VDM_REQUIRE(mlps && n > 0 && n <= COND_MAX && rows > 0 && width > 0, "cond_table: bad arguments (n=%d rows=%d width=%d)", n, rows, width);
    for (int k = 0; k < n; ++k) {
        const vdm_cond_mlp& m = mlps[k];
        VDM_REQUIRE(m.input && m.w1 && m.b1 && m.w2 && m.b2 && m.wproj, "cond_table: NULL pointer in mlp %d", k);
        VDM_REQUIRE(m.in_dim > 0 && m.in_dim <= COND_MAXDIM && m.dim > 0 && m.dim <= COND_MAXDIM, "cond_table: widths of mlp %d out of range", k);
        VDM_REQUIRE(!m.sinusoid || m.in_dim % 2 == 0, "cond_table: the sinusoidal embedding width must be even");
        VDM_REQUIRE((m.in_dim % 4 != 0 || ((uintptr_t)m.w1 & 15) == 0) && (m.dim % 4 != 0 || ((((uintptr_t)m.w2) | ((uintptr_t)m.wproj)) & 15) == 0),
                    "cond_table: weight matrices of mlp %d with 16-byte rows must be 16-byte aligned", k);
        VDM_REQUIRE(!bwd || (m.dw1 && m.db1 && m.dw2 && m.db2 && m.dwproj), "cond_table_bwd: NULL gradient pointer in mlp %d", k);
        a.m[k] = m;
    }
    a.n = n; a.rows = rows; a.width = width;
    return VDM_OK;
}

}  // namespace vdm

using namespace vdm;

extern "C" size_t vdm_cond_saved_floats(const vdm_cond_mlp* mlps, int n, int rows) {
    size_t t = 0;
    for (int k = 0; mlps && k < n; ++k) t += (size_t)rows * (mlps[k].in_dim + 3 * mlps[k].dim);
    return t;
}

extern "C" size_t vdm_cond_bwd_scratch_floats(const vdm_cond_mlp* mlps, int n, int rows, int width) {
    if (!mlps || n <= 0 || rows <= 0 || width <= 0) return 0;
    const size_t nslab = (size_t)(width + COND_BSLAB - 1) / COND_BSLAB;
    size_t t = 0;
    for (int k = 0; k < n; ++k) t += (size_t)rows * (2 + nslab) * (size_t)(mlps[k].dim > 0 ? mlps[k].dim : 0);
    return t;
}

extern "C" int vdm_cond_table_fwd(const vdm_cond_mlp* mlps, int n, int rows, int width, float* table, float* saved, void* stream) {
    CondArgs a{};
    int e = fill(a, mlps, n, rows, width, false);
    if (e) return e;
    VDM_REQUIRE(table, "cond_table_fwd: NULL table");
    a.table = table; a.saved = saved;
    VDM_REQUIRE(rows <= 65535, "cond_table_fwd: at most 65535 rows per call (got %d)", rows);
    hipLaunchKernelGGL(cond_table_fwd_kernel, dim3((width + COND_FSLAB - 1) / COND_FSLAB, rows), dim3(256), 0, (hipStream_t)stream, a);
    VDM_LAUNCH_CHECK("cond_table_fwd_kernel");
    return VDM_OK;
}

extern "C" int vdm_cond_table_bwd(const vdm_cond_mlp* mlps, int n, int rows, int width, const float* dtable, int64_t dtable_stride,
                                  const float* saved, float* scratch, float* dbias, void* stream) {
    CondArgs a{};
    int e = fill(a, mlps, n, rows, width, true);
    if (e) return e;
    VDM_REQUIRE(dtable && saved && scratch && dtable_stride >= width, "cond_table_bwd: bad arguments");
    a.dtable = dtable; a.dtable_stride = dtable_stride; a.saved = const_cast<float*>(saved); a.scratch = scratch; a.dbias = dbias;
    long long total = dbias ? width : 0;
    for (int k = 0; k < n; ++k) {
        const vdm_cond_mlp& m = mlps[k];
        total += (long long)m.dim * m.in_dim + m.dim + (long long)m.dim * m.dim + m.dim + (long long)width * m.dim;
    }
    hipStream_t s = (hipStream_t)stream;
    VDM_REQUIRE(rows <= 65535, "cond_table_bwd: at most 65535 rows per call (got %d)", rows);
    a.nslab = (width + COND_BSLAB - 1) / COND_BSLAB;
    hipLaunchKernelGGL(cond_table_bwd_slabs_kernel, dim3(a.nslab, rows), dim3(256), 0, s, a);
    hipLaunchKernelGGL(cond_table_bwd_rows_kernel, dim3(rows, n), dim3(256), 0, s, a);
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cond_table_bwd_params_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, total);
    VDM_LAUNCH_CHECK("cond_table_bwd kernels");
    return VDM_OK;
}

extern "C" int vdm_cond_table_step(const float* table_t, const float* table_v, const int32_t* step_ptr, int rows, int width, float* out,
                                   void* stream) {
    VDM_REQUIRE(out && rows > 0 && width > 0 && (!table_t || step_ptr), "cond_table_step: bad arguments");
    const int blocks = (rows * width + 255) / 256;
    hipLaunchKernelGGL(cond_table_step_kernel, dim3(blocks < 64 ? blocks : 64), dim3(256), 0, (hipStream_t)stream, table_t, table_v, step_ptr,
                       rows, width, out);
    VDM_LAUNCH_CHECK("cond_table_step_kernel");
    return VDM_OK;
}
