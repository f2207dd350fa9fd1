// cond.hip - K6: conditioning embeddings of the score network in one launch (R5 of SURVEY.md section 8a).
//
// Replaces the ATen chain behind  score_model(zt, t=(gamma_t - gamma_min) / (gamma_max - gamma_min), v_conditionings=...)
// [NB vdm_model.py:320-324 -> networks.py:259-265]: per conditioning k
//     c_k = GELU(Linear2(GELU(Linear1(in_k))))        in_t = [sin(1000 t f_i), cos(1000 t f_i)] (width 64), in_v = v (B, 6)
// and the additive injection table of ALL ResNetBlocks at once (spec D4/D7):
//     table[row][w] = sum_k sum_i c_k[row][i] * Wproj_k[w][i]        w over the concatenated output channels of the blocks.
// <1 MFLOP per row: latency only - one launch forward, two backward (all of ~70 ATen launches + 12 GEMMs before).
// Deterministic (fixed summation orders, no atomics).  GELU is the exact erf form (torch.nn.functional.gelu default).
#include "common.h"

namespace vdm {

constexpr int COND_MAX = 4;          // conditionings per call
constexpr int COND_MAXDIM = 128;     // widest hidden layer / input

struct CondArgs {
    vdm_cond_mlp m[COND_MAX];
    int n, rows, width;
    float* table;                    // fwd out [rows][width]
    float* saved;                    // fwd out / bwd in: per mlp [rows][in_dim + 3 dim] = (input, h1, h2, c)
    const float* dtable;             // bwd in [rows][dtable_stride]
    long long dtable_stride;
    float* scratch;                  // bwd: per mlp [rows][2 dim] = (dh1, dh2)
    float* dbias;                    // bwd out (optional) [width] = column sums of dtable
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ size_t saved_base(const CondArgs& a, int k) {      // float offset of mlp k inside `saved`
    size_t off = 0;
    for (int q = 0; q < k; ++q) off += (size_t)a.rows * (a.m[q].in_dim + 3 * a.m[q].dim);
    return off;
}
__device__ __forceinline__ size_t scratch_base(const CondArgs& a, int k) {
    size_t off = 0;
    for (int q = 0; q < k; ++q) off += (size_t)a.rows * 2 * a.m[q].dim;
    return off;
}

// one block per row
__global__ void __launch_bounds__(256) cond_table_fwd_kernel(const CondArgs a) {
    const int row = blockIdx.x, tid = threadIdx.x;
    __shared__ float in_s[COND_MAXDIM], a1[COND_MAXDIM], c[COND_MAX][COND_MAXDIM];
    for (int k = 0; k < a.n; ++k) {
        const vdm_cond_mlp& m = a.m[k];
        float* sv = a.saved ? a.saved + saved_base(a, k) + (size_t)row * (m.in_dim + 3 * m.dim) : nullptr;
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
        if (tid < m.dim) {
            float h = m.b1[tid];
            const float* w = m.w1 + (size_t)tid * m.in_dim;
            for (int i = 0; i < m.in_dim; ++i) h = fmaf(w[i], in_s[i], h);
            if (sv) sv[m.in_dim + tid] = h;
            a1[tid] = gelu_f(h);
        }
        __syncthreads();
        if (tid < m.dim) {
            float h = m.b2[tid];
            const float* w = m.w2 + (size_t)tid * m.dim;
            for (int i = 0; i < m.dim; ++i) h = fmaf(w[i], a1[i], h);
            const float cv = gelu_f(h);
            if (sv) { sv[m.in_dim + m.dim + tid] = h; sv[m.in_dim + 2 * m.dim + tid] = cv; }
            c[k][tid] = cv;
        }
        __syncthreads();
    }
    for (int w = tid; w < a.width; w += 256) {
        float acc = 0.f;
        for (int k = 0; k < a.n; ++k) {
            const float* p = a.m[k].wproj + (size_t)w * a.m[k].dim;
            for (int i = 0; i < a.m[k].dim; ++i) acc = fmaf(c[k][i], p[i], acc);
        }
        a.table[(size_t)row * a.width + w] = acc;
    }
}

// backward, stage A: one block per row -> dh2 = (dtable Wproj) * gelu'(h2), dh1 = (dh2 W2) * gelu'(h1) into scratch
__global__ void __launch_bounds__(256) cond_table_bwd_rows_kernel(const CondArgs a) {
    const int row = blockIdx.x, tid = threadIdx.x;
    __shared__ float part[2][COND_MAXDIM], dh2[COND_MAXDIM];
    const float* dt = a.dtable + (size_t)row * a.dtable_stride;
    for (int k = 0; k < a.n; ++k) {
        const vdm_cond_mlp& m = a.m[k];
        const float* sv = a.saved + saved_base(a, k) + (size_t)row * (m.in_dim + 3 * m.dim);
        float* sc = a.scratch + scratch_base(a, k) + (size_t)row * 2 * m.dim;
        const int i = tid % COND_MAXDIM, half = tid / COND_MAXDIM;            // two halves of the w range per output i
        if (i < m.dim) {
            float s = 0.f;
            const int w0 = half * ((a.width + 1) / 2), w1 = half ? a.width : (a.width + 1) / 2;
            for (int w = w0; w < w1; ++w) s = fmaf(dt[w], m.wproj[(size_t)w * m.dim + i], s);
            part[half][i] = s;
        }
        __syncthreads();
        if (tid < m.dim) {
            const float d = (part[0][tid] + part[1][tid]) * dgelu_f(sv[m.in_dim + m.dim + tid]);
            dh2[tid] = d;
            sc[m.dim + tid] = d;
        }
        __syncthreads();
        if (tid < m.dim) {
            float s = 0.f;
            for (int j = 0; j < m.dim; ++j) s = fmaf(dh2[j], m.w2[(size_t)j * m.dim + tid], s);
            sc[tid] = s * dgelu_f(sv[m.in_dim + tid]);
        }
        __syncthreads();
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

extern "C" int vdm_cond_table_fwd(const vdm_cond_mlp* mlps, int n, int rows, int width, float* table, float* saved, void* stream) {
    CondArgs a{};
    int e = fill(a, mlps, n, rows, width, false);
    if (e) return e;
    VDM_REQUIRE(table, "cond_table_fwd: NULL table");
    a.table = table; a.saved = saved;
    hipLaunchKernelGGL(cond_table_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a);
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
    hipLaunchKernelGGL(cond_table_bwd_rows_kernel, dim3(rows), dim3(256), 0, s, a);
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
