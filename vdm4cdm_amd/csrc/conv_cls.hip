// conv_cls.hip - per-parity-class convolutions (see conv_common.h).
#include "conv_common.h"

namespace vdm {

// ---------------------------------------------------------------------------------------------
// "Class" convolutions: the convs that couple a coarse grid c and the 2x finer grid u = 2c + p, p in {0,1}^3.
// For a fixed parity class p only a subset of the 27 taps (possibly merged) touches a given coarse offset, so these
// convs are run per class on the COARSE index space with a short tap list (<= 8 entries) instead of 27 taps on the fine
// grid:
//   * nearest-x2 up-sampling conv, forward : out[2c+p] = sum_i Weff[p][i] . C[c + o_i]        (8 merged taps;  MODE_F)
//       per dim  p=0: {o=-1: w0, o=0: w1+w2}   p=1: {o=0: w0+w1, o=+1: w2}          -> 27/8 = 3.4x fewer FLOPs, exact
//   * its input gradient                  : dC[c]  = sum_p sum_i Weff[p][i]^T . dOut[2(c - o_i) + p]          (MODE_B)
//   * stride-2 conv, input gradient       : dIn[2c+p] = sum_i W[t_i]^T . dOut[c + o_i]   (1/2/4/8 taps per class; MODE_F)
//       per dim  p=0: {o=0: w1}                p=1: {o=+1: w0, o=0: w2}             -> no zero-dilated tensor, 1/8 of the FLOPs
// MODE_F: one workgroup = one class x one coarse tile, output scattered to the fine grid (stride 2).
// MODE_B: one workgroup = one coarse tile; loops over the 8 classes, re-staging the class sub-grid G_p[c] = dOut[2c+p]
//         (source stride 2) and accumulating in registers.
// The packed weights hold 64 (class, entry) slots per (chunk, K-block); pack_weights_cls_kernel sums the master taps of
// each slot's mask (and transposes for the gradient modes).
// ---------------------------------------------------------------------------------------------

// LDS-DMA staging of a halo tile whose logical voxel i maps to source voxel ss*i + so (class sub-grid when ss == 2).
template <typename T, typename G>
__device__ __forceinline__ void stage_halo_dma_sub(char* lds, const T* __restrict__ x, const ClsArgs& ca, int n, int oz0, int oy0,
                                                   int ox0, int kb, int ss, int soz, int soy, int sox, int wave, int lane) {
    stage_halo_dma_gen<T, G, 0>(lds, x, ca.c, n, oz0, oy0, ox0, kb, wave, lane, ss, soz, soy, sox, ca.sDz, ca.sDy, ca.sDx);
}

// tap-list MFMA loop (runtime offsets): weights one entry ahead, activations one entry ahead.
template <typename T, typename G, int NC, int NV>
__device__ __forceinline__ void taps_list(f32x4 (&acc)[NV][NC], const char* lds, const uint4* wk /* slot 0 of this class */,
                                          const int nent, const ClsEntry (&ent)[8], const int (&lanex)[3]) {
    constexpr int ROWB = G::HX * 64;
    auto lane_base = [&](int i) { return (ent[i].dx == 0 ? lanex[0] : (ent[i].dx == 1 ? lanex[1] : lanex[2])) + ent[i].lds_off; };
    constexpr int WPD = (NC <= 2) ? 3 : 1;                  // weight prefetch depth (entries); ring of WPD + 1 register sets
    uint4 wf[WPD + 1][NC];
    uint4 af[2][NV];
#pragma unroll
    for (int p = 0; p < WPD; ++p)
        if (p < nent) {
#pragma unroll
            for (int c = 0; c < NC; ++c) wf[p][c] = wk[(p * NC + c) * 64];
        }
    {
        const int b0 = lane_base(0);
#pragma unroll
        for (int v = 0; v < NV; ++v) af[0][v] = *reinterpret_cast<const uint4*>(lds + b0 + v * ROWB);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < nent) {
            if (i + WPD < nent) {
#pragma unroll
                for (int c = 0; c < NC; ++c) wf[(i + WPD) % (WPD + 1)][c] = wk[((i + WPD) * NC + c) * 64];
            }
            if (i + 1 < nent) {
                const int b1 = lane_base(i + 1 < 8 ? i + 1 : 7);
#pragma unroll
                for (int v = 0; v < NV; ++v) af[(i + 1) & 1][v] = *reinterpret_cast<const uint4*>(lds + b1 + v * ROWB);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int c = 0; c < NC; ++c) mma16<T>(acc[v][c], wf[i % (WPD + 1)][c], af[i & 1][v]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// epilogue with an output coordinate map  u = os * (tile voxel) + p   (os = 1: plain; os = 2: scatter into the fine grid)
template <typename T, typename G, int NC, int NV>
__device__ __forceinline__ void cls_epilogue(const f32x4 (&acc)[NV][NC], const ClsArgs& ca, int n, int chunk, int oz0, int oy0, int ox0,
                                             int os, int pz, int py, int px, int cwave, int lane, float* gn_sm = nullptr, int tile = 0) {
    constexpr int EPL = DT<T>::EPL;
    const ConvArgs& a = ca.c;
    const int lx = lane & 15, q = lane >> 4;
    const int cbase = chunk * NC * 16 + q * NC * 4;
    float gs[NC * 4], gq[NC * 4];                          // GroupNorm partials of the stored outputs (a.gnp; see conv_epilogue)
#pragma unroll
    for (int j = 0; j < NC * 4; ++j) gs[j] = gq[j] = 0.f;
    float badd[NC * 4];
#pragma unroll
    for (int j = 0; j < NC * 4; ++j) badd[j] = (a.bias && cbase + j < a.Cout) ? a.bias[cbase + j] : 0.f;
    const bool vec_ok = (a.Cout % (NC * 4) == 0) && (cbase + NC * 4 <= a.Cout);
    T* out = reinterpret_cast<T*>(a.out);
    const T* res = reinterpret_cast<const T*>(a.res);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int r = cwave * NV + v;
        const int cz = oz0 + r / G::TY, cy = oy0 + r % G::TY, cx = ox0 + lx;
        if (cz >= a.Dz || cy >= a.Dy || cx >= a.Dx) continue;
        const size_t vo = ((((size_t)n * ca.oDz + (os * cz + pz)) * ca.oDy + (os * cy + py)) * ca.oDx + (os * cx + px)) * a.Cout + cbase;
        float val[NC * 4];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) val[c * 4 + j] = acc[v][c][j] + badd[c * 4 + j];
        if (res) {
            if (vec_ok && NC * 4 >= EPL) {
#pragma unroll
                for (int i = 0; i < NC * 4 / EPL; ++i) {
                    Piece<T> pr;
                    pr.load(*reinterpret_cast<const uint4*>(res + vo + i * EPL));
#pragma unroll
                    for (int j = 0; j < EPL; ++j) val[i * EPL + j] += pr.f[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < NC * 4; ++j)
                    if (vec_ok || cbase + j < a.Cout) val[j] += ld_elem<T>(res + vo + j);
            }
        }
        if (a.gnp) {
#pragma unroll
            for (int j = 0; j < NC * 4; ++j)
                if (vec_ok || cbase + j < a.Cout) { gs[j] += val[j]; gq[j] += val[j] * val[j]; }
        }
        if (vec_ok && sizeof(T) == 2 && NC >= 2) {
            uint16_t* o16 = reinterpret_cast<uint16_t*>(out) + vo;
#pragma unroll
            for (int i = 0; i < NC / 2; ++i)
                *reinterpret_cast<uint4*>(o16 + i * 8) =
                    make_uint4(pack_bf16x2(val[i * 8], val[i * 8 + 1]), pack_bf16x2(val[i * 8 + 2], val[i * 8 + 3]),
                               pack_bf16x2(val[i * 8 + 4], val[i * 8 + 5]), pack_bf16x2(val[i * 8 + 6], val[i * 8 + 7]));
        } else if (vec_ok && sizeof(T) == 4) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + vo + c * 4) =
                    make_float4(val[c * 4], val[c * 4 + 1], val[c * 4 + 2], val[c * 4 + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < NC * 4; ++j)
                if (cbase + j < a.Cout) st_elem<T>(out + vo + j, val[j]);
        }
    }
    if (a.gnp)                                            // workgroup-uniform; one partial slot per (coarse tile, class)
        gn_partials_reduce<NC>(gs, gq, gn_sm, a.gnp + ((size_t)n * (a.ntz * a.nty * a.ntx * 8) + tile) * a.Cout * 2, chunk * NC * 16, a.Cout,
                               cwave, lane);
    (void)EPL;
}

// MODE: 0 = F (coarse -> fine, one class per workgroup), 1 = B (fine -> coarse, loops over the classes, accumulates).
// (A stage-once-for-all-classes variant of F was measured and was not faster: one workgroup per CU, cold restarts per class.)
template <typename T, int NC, int MODE>
__global__ void __launch_bounds__(256, 2) conv_cls_kernel(const ClsArgs ca) {
    using G = Geo<3, 1, 4, 8>;
    constexpr int NV = G::NV;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const ConvArgs& a = ca.c;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int b = xcd_remap(blockIdx.x, gridDim.x);
    int cls = 0;
    if (MODE == 0) { cls = b & 7; b >>= 3; }
    int tx, ty, tz, n, chunk;
    decode_tile(a, (uint32_t)b, tx, ty, tz, n, chunk);
    const int oz0 = tz * G::TZ, oy0 = ty * G::TY, ox0 = tx * 16;

    f32x4 acc[NV][NC];
    auto zero_acc = [&]() {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    int lanex[3];
    operand_lane_offsets<G, NV>(lanex, wave, lane);
    const T* x = reinterpret_cast<const T*>(a.x);
    const uint4* wbase = reinterpret_cast<const uint4*>(a.w) + (size_t)chunk * a.nkb * 64 * NC * 64 + lane;

    const int ncls = (MODE == 1) ? 8 : 1;
    bool first = true;
    for (int ci = 0; ci < ncls; ++ci) {
        const int cl = (MODE == 1) ? ci : cls;
        const int pz = (cl >> 2) & 1, py = (cl >> 1) & 1, px = cl & 1;
        for (int kb = 0; kb < a.nkb; ++kb) {
            if (!first) __syncthreads();
            first = false;
            if (MODE == 1)
                stage_halo_dma_sub<T, G>(lds, x, ca, n, oz0, oy0, ox0, kb, 2, pz, py, px, wave, lane);
            else
                stage_halo_dma_sub<T, G>(lds, x, ca, n, oz0, oy0, ox0, kb, 1, 0, 0, 0, wave, lane);
            __syncthreads();
            const uint4* wk = wbase + ((size_t)kb * 64 + cl * 8) * NC * 64;
            taps_list<T, G, NC, NV>(acc, lds, wk, ca.t.n[cl], ca.t.e[cl], lanex);
        }
    }
    if (MODE == 1)
        cls_epilogue<T, G, NC, NV>(acc, ca, n, chunk, oz0, oy0, ox0, 1, 0, 0, 0, wave, lane);
    else
        cls_epilogue<T, G, NC, NV>(acc, ca, n, chunk, oz0, oy0, ox0, 2, (cls >> 2) & 1, (cls >> 1) & 1, cls & 1, wave, lane,
                                   reinterpret_cast<float*>(lds + ((G::HVOX + 15) / 16) * 1024), ((tz * a.nty + ty) * a.ntx + tx) * 8 + cls);
}

template <typename T, int NC, int MODE>
static int launch_cls_cfg(const ClsArgs& ca0, hipStream_t s) {
    using G = Geo<3, 1, 4, 8>;
    ClsArgs ca = ca0;
    ConvArgs& a = ca.c;
    a.ntz = cdiv(a.Dz, G::TZ); a.nty = cdiv(a.Dy, G::TY); a.ntx = cdiv(a.Dx, 16);
    set_tile_divs(a);
    const size_t lds = (size_t)((G::HVOX + 15) / 16) * 1024 + GN_SCRATCH_BYTES;
    auto kern = conv_cls_kernel<T, NC, MODE>;
    static unsigned long long lds_done = 0;
    {
        int e = set_lds(kern, lds, lds_done);
        if (e) return e;
    }
    const long long nwg = (long long)a.N * a.ntz * a.nty * a.ntx * a.nchunks * (MODE == 0 ? 8 : 1);
    if (nwg > 0x7fffffffLL) { set_error("conv(class): grid too large"); return VDM_ERR_ARG; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, s, ca);
    VDM_LAUNCH_CHECK("conv_cls_kernel");
    return VDM_OK;
}

template <typename T>
static int launch_cls(const ClsArgs& ca, int nc, int mode_b, hipStream_t s) {
    if (mode_b) {
        switch (nc) {
            case 1: return launch_cls_cfg<T, 1, 1>(ca, s);
            case 2: return launch_cls_cfg<T, 2, 1>(ca, s);
            default: return launch_cls_cfg<T, 4, 1>(ca, s);
        }
    }
    switch (nc) {
        case 1: return launch_cls_cfg<T, 1, 0>(ca, s);
        case 2: return launch_cls_cfg<T, 2, 0>(ca, s);
        default: return launch_cls_cfg<T, 4, 0>(ca, s);
    }
}

// x: staged tensor (K channels), out: O channels.  cd/ch/cw: coarse dims.
int run_cls(const vdm_conv_desc* d, int kind, const void* x, const void* w, const float* bias, const void* res, void* out,
                   int cd, int ch, int cw, hipStream_t s, float* gn_partials) {
    const int dgrad = kind != CLS_UP_FWD;
    const Plan p = plan_of(d, dgrad);
    ClsArgs ca{};
    ClsMasks masks;
    build_cls(kind, ca.t, masks);
    ConvArgs& a = ca.c;
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.out = out; a.gnp = gn_partials;
    a.N = d->n; a.Dz = cd; a.Dy = ch; a.Dx = cw;
    a.Iz = cd; a.Iy = ch; a.Ix = cw;
    a.circular = d->pad_mode == VDM_PAD_CIRCULAR;
    a.Cin = p.K; a.CinStride = cpad(p.K, d->dtype); a.Cout = p.O;
    a.nchunks = p.nchunks; a.nkb = p.nkb;
    const int mode_b = kind == CLS_UP_DGRAD;
    ca.sDz = mode_b ? 2 * cd : cd; ca.sDy = mode_b ? 2 * ch : ch; ca.sDx = mode_b ? 2 * cw : cw;
    ca.oDz = mode_b ? cd : 2 * cd; ca.oDy = mode_b ? ch : 2 * ch; ca.oDx = mode_b ? cw : 2 * cw;
    if (d->dtype == VDM_F32) return launch_cls<float>(ca, p.nc, mode_b, s);
    return launch_cls<bf16_t>(ca, p.nc, mode_b, s);
}

}  // namespace vdm
