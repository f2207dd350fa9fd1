// conv_api.hip - weight packing kernels and the C-ABI entry points of the convolutions (see conv_common.h).
#include "conv_common.h"

namespace vdm {

// packed[chunk][kb][slot 0..63][ct][lane][EPL] = sum over the master taps in mask[slot] of W (transpose: W[t][k][o]).
template <typename T>
__global__ void pack_weights_cls_kernel(const float* __restrict__ w, T* __restrict__ p, int cout_m, int cin_m, int nc, int nchunks,
                                        int nkb, int transpose, const ClsMasks masks) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    const size_t total = (size_t)nchunks * nkb * 64 * nc * 64 * EPL;
    const int O = transpose ? cin_m : cout_m, K = transpose ? cout_m : cin_m;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int j = r % EPL; r /= EPL;
        const int lane = r % 64; r /= 64;
        const int ct = r % nc; r /= nc;
        const int slot = r % 64; r /= 64;
        const int kb = r % nkb; r /= nkb;
        const int chunk = (int)r;
        const int m = lane & 15, q = lane >> 4;
        const int o = chunk * nc * 16 + nc * 4 * (m >> 2) + 4 * ct + (m & 3);
        const int k = kb * KB + q * EPL + j;
        float v = 0.f;
        if (o < O && k < K) {
            const unsigned mask = masks.m[slot];
            for (int t = 0; t < 27; ++t)
                if ((mask >> t) & 1u) v += transpose ? w[((size_t)t * cout_m + k) * cin_m + o] : w[((size_t)t * cout_m + o) * cin_m + k];
        }
        st_packed_w<T>(p, i, v);
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing: master fp32 [taps][cout][cin] -> MFMA A-fragment order
//   packed[chunk][kb][tap][ct][lane][EPL]: lane (m = lane&15, q = lane>>4), element j:
//     out channel o = chunk*NC*16 + NC*4*(m>>2) + 4*ct + (m&3) ; reduction channel k = kb*KB + q*EPL + j
//   fwd  : W[tap][o][k]                       dgrad: W[flip(tap)][k][o]  (o indexes cin, k indexes cout)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ p, int taps, int cout_m, int cin_m,
                                    int nc, int nchunks, int nkb, int dgrad) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    const size_t total = (size_t)nchunks * nkb * taps * nc * 64 * EPL;
    const int O = dgrad ? cin_m : cout_m, K = dgrad ? cout_m : cin_m;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int j = r % EPL; r /= EPL;
        const int lane = r % 64; r /= 64;
        const int ct = r % nc; r /= nc;
        const int tap = r % taps; r /= taps;
        const int kb = r % nkb; r /= nkb;
        const int chunk = (int)r;
        const int m = lane & 15, q = lane >> 4;
        const int o = chunk * nc * 16 + nc * 4 * (m >> 2) + 4 * ct + (m & 3);
        const int k = kb * KB + q * EPL + j;
        float v = 0.f;
        if (o < O && k < K) {
            if (dgrad)
                v = w[((size_t)(taps - 1 - tap) * cout_m + k) * cin_m + o];
            else
                v = w[((size_t)tap * cout_m + o) * cin_m + k];
        }
        st_packed_w<T>(p, i, v);
    }
}

// packed weights of conv_kpack_kernel: [chunk][tap group g][cout tile][lane (m, q)][ci 0..7] = W[tap 4g+q][cout][ci]
__global__ void pack_weights_kpack_kernel(const float* __restrict__ w, bf16_t* __restrict__ p, int cout_m, int cin_m, int nc, int nchunks,
                                          int dgrad) {
    const size_t total = (size_t)nchunks * 7 * nc * 64 * 8;
    const int O = dgrad ? cin_m : cout_m, K = dgrad ? cout_m : cin_m;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int j = r % 8; r /= 8;
        const int lane = r % 64; r /= 64;
        const int ct = r % nc; r /= nc;
        const int g = r % 7; r /= 7;
        const int chunk = (int)r;
        const int m = lane & 15, q = lane >> 4;
        const int o = chunk * nc * 16 + nc * 4 * (m >> 2) + 4 * ct + (m & 3);
        const int tap = 4 * g + q;
        float v = 0.f;
        if (o < O && j < K && tap < 27) {
            if (dgrad)
                v = w[((size_t)(26 - tap) * cout_m + j) * cin_m + o];
            else
                v = w[((size_t)tap * cout_m + o) * cin_m + j];
        }
        st_elem<bf16_t>(p + i, v);
    }
}

}  // namespace vdm

using namespace vdm;

extern "C" size_t vdm_conv_packed_bytes(const vdm_conv_desc* d, int pack_mode) {
    if (validate(d) != VDM_OK) return 0;
    const Plan p = plan_of(d, pack_mode == VDM_PACK_DGRAD);
    if (uses_cls(d, pack_mode == VDM_PACK_DGRAD)) return (size_t)p.nchunks * p.nkb * 64 * p.nc * 64 * 16;      // 64 (class, entry) slots
    if (uses_kpack(d, pack_mode == VDM_PACK_DGRAD)) return (size_t)p.nchunks * 7 * p.nc * 64 * 16;                       // 7 groups of 4 taps
    return (size_t)p.nchunks * p.nkb * p.taps * p.nc * 64 * 16;
}

extern "C" int vdm_conv_pack_weights(const vdm_conv_desc* d, int pack_mode, const float* w_master, void* w_packed, void* stream) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(w_master && w_packed, "conv_pack_weights: NULL pointer");
    VDM_REQUIRE(pack_mode == VDM_PACK_FWD || pack_mode == VDM_PACK_DGRAD, "conv_pack_weights: bad mode %d", pack_mode);
    const int dg = pack_mode == VDM_PACK_DGRAD;
    const Plan p = plan_of(d, dg);
    const size_t elems = vdm_conv_packed_bytes(d, pack_mode) / (d->dtype == VDM_F32 ? 4 : 2);
    const unsigned grid = (unsigned)((elems + 255) / 256 < 2048 ? (elems + 255) / 256 : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (uses_cls(d, dg)) {
        ClsTable tab;
        ClsMasks masks;
        build_cls(cls_kind(d, dg), tab, masks);
        if (d->dtype == VDM_F32)
            hipLaunchKernelGGL(pack_weights_cls_kernel<float>, dim3(grid), dim3(256), 0, s, w_master, (float*)w_packed, d->cout, d->cin, p.nc,
                               p.nchunks, p.nkb, dg, masks);
        else
            hipLaunchKernelGGL(pack_weights_cls_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w_master, (bf16_t*)w_packed, d->cout, d->cin,
                               p.nc, p.nchunks, p.nkb, dg, masks);
        VDM_LAUNCH_CHECK("pack_weights_cls_kernel");
        return VDM_OK;
    }
    if (uses_kpack(d, dg)) {
        hipLaunchKernelGGL(pack_weights_kpack_kernel, dim3(grid), dim3(256), 0, s, w_master, (bf16_t*)w_packed, d->cout, d->cin, p.nc, p.nchunks, dg);
        VDM_LAUNCH_CHECK("pack_weights_kpack_kernel");
        return VDM_OK;
    }
    if (d->dtype == VDM_F32)
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid), dim3(256), 0, s, w_master, (float*)w_packed, p.taps, d->cout, d->cin,
                           p.nc, p.nchunks, p.nkb, dg);
    else
        hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w_master, (bf16_t*)w_packed, p.taps, d->cout,
                           d->cin, p.nc, p.nchunks, p.nkb, dg);
    VDM_LAUNCH_CHECK("pack_weights_kernel");
    return VDM_OK;
}

static void fwd_args(ConvArgs& a, const vdm_conv_desc* d) {
    const Plan p = plan_of(d, 0);
    fill_dims(a, d);
    a.Cin = d->cin; a.CinStride = cpad(d->cin, d->dtype); a.Cout = d->cout;
    a.nchunks = p.nchunks; a.nkb = p.nkb;
}

// ---- all weight packings of a network in ONE launch ---------------------------------------------------------------------
// The packed copies of every conv (forward and dgrad form) are rebuilt after each optimiser step: ~120 launches of a few
// microseconds each when done conv by conv.  vdm_conv_pack_plan() fills one work item per (conv, form) on the host; the caller
// concatenates them, cuts the concatenated element range into chunks of VDM_PACK_CHUNK elements that do not straddle items, uploads
// both tables once and calls vdm_conv_pack_many() per step.
__device__ ClsMasks g_cls_masks[3];

// value of packed element (row r = everything above the [lane][EPL] fragment, lane, j)
template <typename T>
__device__ __forceinline__ float pack_value(const vdm_pack_item& it, const float* __restrict__ w, size_t r, int lane, int j) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    const int dgrad = it.dgrad, cout_m = it.cout, cin_m = it.cin, nc = it.nc;
    const int O = dgrad ? cin_m : cout_m, K = dgrad ? cout_m : cin_m;
    const int ct = r % nc; r /= nc;
    const int m = lane & 15, q = lane >> 4;
    if (it.variant == VDM_CONV_VARIANT_KPACK) {
        const int g = r % 7; r /= 7;
        const int chunk = (int)r;
        const int o = chunk * nc * 16 + nc * 4 * (m >> 2) + 4 * ct + (m & 3);
        const int tap = 4 * g + q;
        if (!(o < O && j < K && tap < 27)) return 0.f;
        return dgrad ? w[((size_t)(26 - tap) * cout_m + j) * cin_m + o] : w[((size_t)tap * cout_m + o) * cin_m + j];
    }
    if (it.variant == VDM_CONV_VARIANT_CLASS) {
        const int slot = r % 64; r /= 64;
        const int kb = r % it.nkb; r /= it.nkb;
        const int chunk = (int)r;
        const int o = chunk * nc * 16 + nc * 4 * (m >> 2) + 4 * ct + (m & 3);
        const int k = kb * KB + q * EPL + j;
        float v = 0.f;
        if (o < O && k < K) {
            const unsigned mask = g_cls_masks[it.cls_kind].m[slot];
            for (int t = 0; t < 27; ++t)
                if ((mask >> t) & 1u) v += dgrad ? w[((size_t)t * cout_m + k) * cin_m + o] : w[((size_t)t * cout_m + o) * cin_m + k];
        }
        return v;
    }
    const int taps = it.taps;
    const int tap = r % taps; r /= taps;
    const int kb = r % it.nkb; r /= it.nkb;
    const int chunk = (int)r;
    const int o = chunk * nc * 16 + nc * 4 * (m >> 2) + 4 * ct + (m & 3);
    const int k = kb * KB + q * EPL + j;
    if (!(o < O && k < K)) return 0.f;
    return dgrad ? w[((size_t)(taps - 1 - tap) * cout_m + k) * cin_m + o] : w[((size_t)tap * cout_m + o) * cin_m + k];
}

// one block = one chunk (a whole number of [64 lanes][EPL] fragments): the slow coordinates (cout tile, tap, K-block, chunk) are
// block-uniform per fragment, only (lane, j) vary over the threads
template <typename T>
__global__ void __launch_bounds__(256) pack_many_kernel(const vdm_pack_item* __restrict__ items, const vdm_pack_chunk* __restrict__ chunks) {
    constexpr int EPL = DT<T>::EPL, FRAG = 64 * EPL;
    const vdm_pack_chunk c = chunks[blockIdx.x];
    const vdm_pack_item it = items[c.item];
    const float* w = it.w_master;
    T* out = reinterpret_cast<T*>(it.w_packed);
    const int j = threadIdx.x % EPL;
    for (long long f = c.first; f < c.first + c.count; f += FRAG) {          // (first and count are multiples of FRAG)
        const size_t r = (size_t)(f / FRAG);
#pragma unroll
        for (int u = 0; u < FRAG / 256; ++u) {
            const int e = threadIdx.x + u * 256;
            st_packed_w<T>(out, (size_t)(f + e), pack_value<T>(it, w, r, e / EPL, j));
        }
    }
}

extern "C" int vdm_conv_pack_plan(const vdm_conv_desc* d, int pack_mode, const float* w_master, void* w_packed, vdm_pack_item* item) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(item && w_master && w_packed, "conv_pack_plan: NULL pointer");
    VDM_REQUIRE(pack_mode == VDM_PACK_FWD || pack_mode == VDM_PACK_DGRAD, "conv_pack_plan: bad mode %d", pack_mode);
    const int dg = pack_mode == VDM_PACK_DGRAD;
    const Plan p = plan_of(d, dg);
    item->w_master = w_master; item->w_packed = w_packed;
    item->taps = p.taps; item->cout = d->cout; item->cin = d->cin; item->nc = p.nc; item->nchunks = p.nchunks; item->nkb = p.nkb;
    item->dgrad = dg;
    item->variant = uses_cls(d, dg) ? VDM_CONV_VARIANT_CLASS : (uses_kpack(d, dg) ? VDM_CONV_VARIANT_KPACK : VDM_CONV_VARIANT_GENERIC);
    item->cls_kind = item->variant == VDM_CONV_VARIANT_CLASS ? cls_kind(d, dg) : 0;
    item->dtype = d->dtype;
    item->elems = (long long)(vdm_conv_packed_bytes(d, pack_mode) / (d->dtype == VDM_F32 ? 4 : 2));
    return VDM_OK;
}

extern "C" int vdm_conv_pack_many(const vdm_pack_item* items_dev, const vdm_pack_chunk* chunks_dev, int nchunks, int dtype, void* stream) {
    VDM_REQUIRE(items_dev && chunks_dev && nchunks > 0, "conv_pack_many: empty work list");
    VDM_REQUIRE(dtype == VDM_F32 || dtype == VDM_BF16, "conv_pack_many: bad dtype %d", dtype);
    static unsigned long long masks_up = 0;                 // per device ordinal (the symbol lives in each device's copy of the module)
    const int dev = current_device();
    if (dev >= 64 || !((masks_up >> dev) & 1ull)) {
        ClsMasks h[3];
        ClsTable tab;
        for (int k = 0; k < 3; ++k) build_cls(k, tab, h[k]);
        int e = check_hip(hipMemcpyToSymbol(HIP_SYMBOL(g_cls_masks), h, sizeof(h)), "hipMemcpyToSymbol(g_cls_masks)");
        if (e) return e;
        if (dev < 64) masks_up |= 1ull << dev;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VDM_F32)
        hipLaunchKernelGGL(pack_many_kernel<float>, dim3(nchunks), dim3(256), 0, s, items_dev, chunks_dev);
    else
        hipLaunchKernelGGL(pack_many_kernel<bf16_t>, dim3(nchunks), dim3(256), 0, s, items_dev, chunks_dev);
    VDM_LAUNCH_CHECK("pack_many_kernel");
    return VDM_OK;
}

extern "C" int vdm_conv_gn_tiles(const vdm_conv_desc* d) {
    if (validate(d) != VDM_OK) return 0;
    if (uses_cls(d, 0)) return 8 * cdiv(d->od / 2, 4) * cdiv(d->oh / 2, 8) * cdiv(d->ow / 2, 16);     // up-sampling conv: (coarse tile, class)
    ConvArgs a{};
    fwd_args(a, d);
    int tz, ty;
    fwd_tile_shape(a, d->dtype, d->out_f32, d->ksize, d->stride, d->upsample, tz, ty);
    return cdiv(a.Dz, tz) * cdiv(a.Dy, ty) * cdiv(a.Dx, 16);
}

extern "C" int vdm_conv_fwd(const vdm_conv_desc* d, const void* x, const void* w_packed, const float* bias, const float* nbias,
                            int64_t nbias_stride, const void* residual, void* out, float* gn_partials, void* stream) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(x && w_packed && out, "conv_fwd: NULL pointer");
    if (uses_cls(d, 0)) {
        VDM_REQUIRE(!nbias && !d->out_f32, "conv_fwd: the up-sampling conv takes no per-sample bias / fp32 output");
        return run_cls(d, CLS_UP_FWD, x, w_packed, bias, residual, out, d->od / 2, d->oh / 2, d->ow / 2, (hipStream_t)stream, gn_partials);
    }
    const Plan p = plan_of(d, 0);
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.nbias = nbias; a.nbias_stride = nbias_stride; a.res = residual; a.out = out;
    a.gnp = gn_partials;
    fwd_args(a, d);
    return launch_fwd(a, d->dtype, d->out_f32, d->ksize, d->stride, d->upsample, p.nc, (hipStream_t)stream);
}

// host only: can vdm_conv_fwd_gn run this conv (the generic bf16 3x3x3 stride-1 kernel, not the tap-packed / K-split / class kernels)?
extern "C" int vdm_conv_fwd_gn_supported(const vdm_conv_desc* d) {
    if (validate(d) != VDM_OK) return 0;
    if (d->dtype != VDM_BF16 || d->ksize != 3 || d->stride != 1 || d->upsample || d->cin > 512 || d->cin % 8) return 0;
    return vdm_conv_kernel_variant(d, 0) == VDM_CONV_VARIANT_GENERIC || vdm_conv_kernel_variant(d, 0) == VDM_CONV_VARIANT_SPLIT;
}

extern "C" int vdm_conv_fwd_gn(const vdm_conv_desc* d, const void* x, const void* w_packed, const float* bias, const float* nbias,
                               int64_t nbias_stride, const void* residual, void* out, float* gn_partials, const float* stats,
                               const float* gamma, const float* beta, int groups, float eps, void* stream) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(x && w_packed && out && stats && gamma && beta, "conv_fwd_gn: NULL pointer");
    VDM_REQUIRE(vdm_conv_fwd_gn_supported(d), "conv_fwd_gn: this conv has no GroupNorm prologue (vdm_conv_fwd_gn_supported)");
    VDM_REQUIRE(groups > 0 && groups <= 64 && d->cin % groups == 0, "conv_fwd_gn: %d channels / %d groups", d->cin, groups);
    const Plan p = plan_of(d, 0);
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.nbias = nbias; a.nbias_stride = nbias_stride; a.res = residual; a.out = out;
    a.gnp = gn_partials;
    fwd_args(a, d);
    a.gstats = stats; a.ggamma = gamma; a.gbeta = beta; a.gG = groups; a.geps = eps;
    a.gcnt = (float)((double)d->od * d->oh * d->ow * (d->cin / groups));
    return launch_fwd_gnp(a, d->out_f32, p.nc, (hipStream_t)stream);
}

extern "C" int vdm_conv_dgrad(const vdm_conv_desc* d, const void* dout, const void* w_packed_dgrad, const void* residual, void* dx,
                              void* stream) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(dout && w_packed_dgrad && dx, "conv_dgrad: NULL pointer");
    if (uses_cls(d, 1)) {
        // up-sampling conv: dout is (od,oh,ow), dx is the coarse input (od/2,..);  stride-2 conv: dout is (od,oh,ow), dx is (2od,..)
        if (d->upsample) return run_cls(d, CLS_UP_DGRAD, dout, w_packed_dgrad, nullptr, residual, dx, d->od / 2, d->oh / 2, d->ow / 2, (hipStream_t)stream);
        return run_cls(d, CLS_S2_DGRAD, dout, w_packed_dgrad, nullptr, residual, dx, d->od, d->oh, d->ow, (hipStream_t)stream);
    }
    const Plan p = plan_of(d, 1);
    ConvArgs a{};
    a.x = dout; a.w = w_packed_dgrad; a.res = residual; a.out = dx;
    a.N = d->n; a.Dz = d->od; a.Dy = d->oh; a.Dx = d->ow;
    a.Iz = a.Sz = d->od; a.Iy = a.Sy = d->oh; a.Ix = a.Sx = d->ow;       // dgrad runs on the output grid
    a.circular = d->pad_mode == VDM_PAD_CIRCULAR;
    a.Cin = d->cout; a.CinStride = cpad(d->cout, d->dtype); a.Cout = d->cin;
    a.nchunks = p.nchunks; a.nkb = p.nkb;
    return launch_fwd(a, d->dtype, 0, d->ksize, 1, 0, p.nc, (hipStream_t)stream);
}

static void dgrad_args(ConvArgs& a, const vdm_conv_desc* d) {
    const Plan p = plan_of(d, 1);
    a.N = d->n; a.Dz = d->od; a.Dy = d->oh; a.Dx = d->ow;
    a.Iz = a.Sz = d->od; a.Iy = a.Sy = d->oh; a.Ix = a.Sx = d->ow;       // dgrad runs on the output grid
    a.circular = d->pad_mode == VDM_PAD_CIRCULAR;
    a.Cin = d->cout; a.CinStride = cpad(d->cout, d->dtype); a.Cout = d->cin;
    a.nchunks = p.nchunks; a.nkb = p.nkb;
}

extern "C" int vdm_conv_dgrad_gn_tiles(const vdm_conv_desc* d) {
    if (validate(d) != VDM_OK || d->ksize != 3 || d->stride != 1 || d->upsample) return 0;
    ConvArgs a{};
    dgrad_args(a, d);
    int tz, ty;
    fwd_tile_shape(a, d->dtype, 0, 3, 1, 0, tz, ty);
    return cdiv(a.Dz, tz) * cdiv(a.Dy, ty) * cdiv(a.Dx, 16);
}

extern "C" int vdm_conv_dgrad_gn(const vdm_conv_desc* d, const void* dout, const void* w_packed_dgrad, void* dyh, const vdm_gn_fold* f,
                                 void* stream) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(dout && w_packed_dgrad && dyh && f, "conv_dgrad_gn: NULL pointer");
    VDM_REQUIRE(d->ksize == 3 && d->stride == 1 && !d->upsample, "conv_dgrad_gn: only the 3x3x3 stride-1 conv feeds a GroupNorm backward");
    VDM_REQUIRE(f->x1 && f->stats && f->gamma && f->beta && f->partials && f->groups > 0, "conv_dgrad_gn: NULL pointer in the fold");
    const int C = d->cin;
    const Plan p = plan_of(d, 1);
    VDM_REQUIRE(f->c1 + f->c2 == C && (f->c2 == 0 || f->x2), "conv_dgrad_gn: c1 + c2 must equal the conv's input channels (%d)", C);
    VDM_REQUIRE(C % f->groups == 0 && C % (p.nc * 4) == 0 && C % epl_of(d->dtype) == 0, "conv_dgrad_gn: channel count %d not supported", C);
    VDM_REQUIRE(f->c2 == 0 || f->c1 % (p.nc * 4) == 0, "conv_dgrad_gn: a lane's %d channels would straddle the concat boundary", p.nc * 4);
    ConvArgs a{};
    a.x = dout; a.w = w_packed_dgrad; a.out = dyh; a.gnp = f->partials;
    dgrad_args(a, d);
    a.gx1 = f->x1; a.gx2 = f->x2; a.gc1 = f->c1; a.gc2 = f->c2; a.gG = f->groups;
    a.gstats = f->stats; a.ggamma = f->gamma; a.gbeta = f->beta; a.gmask = f->keep_mask;
    a.geps = f->eps; a.ginv_keep = f->keep_mask ? f->inv_keep : 1.0f;
    a.gcnt = (float)((double)d->od * d->oh * d->ow * (C / f->groups));
    return launch_fwd_gnb(a, d->dtype, p.nc, (hipStream_t)stream);
}

extern "C" int vdm_conv_kernel_variant(const vdm_conv_desc* d, int dgrad) {
    if (validate(d)) return -1;
    if (uses_cls(d, dgrad)) return VDM_CONV_VARIANT_CLASS;
    if (uses_kpack(d, dgrad)) return VDM_CONV_VARIANT_KPACK;
    if (d->dtype == VDM_BF16 && d->ksize == 3 && !d->upsample && (dgrad || (d->stride == 1 && !d->out_f32))) {
        const Plan p = plan_of(d, dgrad);
        if (p.nc == 4) {
            ConvArgs a{};
            a.N = d->n; a.Dz = d->od; a.Dy = d->oh; a.Dx = d->ow;         // (dgrad of a stride-1 conv runs on the output grid too)
            a.Cout = p.O; a.nchunks = p.nchunks; a.nkb = p.nkb;
            int tz, ty;
            small_grid_tile(a, tz, ty);
            if (d->stride == 1 && ksplit_tile(a, tz, ty)) return VDM_CONV_VARIANT_KSPLIT;
            if (uses_split(a, tz, ty)) return VDM_CONV_VARIANT_SPLIT;
        }
    }
    return VDM_CONV_VARIANT_GENERIC;
}

extern "C" size_t vdm_conv_wgrad_workspace_bytes(const vdm_conv_desc* d) {
    if (validate(d) != VDM_OK) return 0;
    const int CL = d->dtype == VDM_F32 ? 16 : 32;
    const int taps = d->ksize * d->ksize * d->ksize;
    const int cls = d->upsample ? 8 : 1;                   // up-sampling conv: 8 parity classes x 8 merged taps
    const int npairs = cls * cdiv(d->cout, CL) * cdiv(d->cin, CL);
    int P = wgrad_wgs() / npairs;
    if (P < 1) P = 1;
    const int slots = d->upsample ? 8 : (taps > 1 ? 1 : 4) * taps;
    size_t need = (size_t)npairs * P * slots * CL * CL * sizeof(float) + (size_t)cdiv(d->cout, CL) * cls * P * CL * sizeof(float);
    const int thin = wgrad_thin_mode(d->dtype, d->ksize, d->stride, d->upsample, d->cin, d->cout, false, false);
    if (thin >= 0) {
        const size_t t = wgrad_thin_workspace_bytes(d->n, d->od, d->oh, d->ow, thin == 0 ? d->cout : d->cin);
        need = need > t ? need : t;
    }
    return need;
}

extern "C" int vdm_conv_wgrad(const vdm_conv_desc* d, const void* x, const void* dout, float* dw, float* dbias, int accumulate,
                              void* workspace, size_t workspace_bytes, void* stream) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(x && dout && dw && workspace, "conv_wgrad: NULL pointer");
    {   // conv_in / conv_out: one side has <= 2 channels - the (tap, channel) pairs become the MFMA's N dimension (wgrad_thin.hip)
        static const bool off = getenv("VDM4CDM_NO_THIN_WGRAD") != nullptr;
        const int thin = off ? -1 : wgrad_thin_mode(d->dtype, d->ksize, d->stride, d->upsample, d->cin, d->cout, dbias != nullptr, accumulate != 0);
        if (thin >= 0 && !(thin == 1 && dbias) && !(d->pad_mode == VDM_PAD_CIRCULAR && d->ow < 17))
            return launch_wgrad_thin(thin, x, dout, d->n, d->od, d->oh, d->ow, d->cin, d->cout, d->pad_mode == VDM_PAD_CIRCULAR, dw, dbias, workspace,
                                     workspace_bytes, (hipStream_t)stream);
    }
    const int CL = d->dtype == VDM_F32 ? 16 : 32;
    WgradArgs w{};
    fill_dims(w.c, d);
    w.c.x = x;
    w.c.Cin = d->cin; w.c.CinStride = cpad(d->cin, d->dtype); w.c.Cout = d->cout;
    w.dout = dout; w.dout_stride = cpad(d->cout, d->dtype);
    w.slabs = (float*)workspace;
    w.ncb = cdiv(d->cout, CL); w.nkb = cdiv(d->cin, CL);
    hipStream_t s = (hipStream_t)stream;
    return launch_wgrad_any(w, dw, dbias, accumulate, d->cout, d->cin, d->ksize, d->stride, d->upsample, workspace_bytes, d->dtype, s);
}


#ifdef VDM_TIMELINE
namespace vdm { unsigned long long* g_timeline_stamps = nullptr; }
// diagnostic build only (make timeline): every later conv_fwd_kernel launch writes [workgroup][wave][8] stamps to `buf`
extern "C" int vdm_debug_set_stamps(void* buf) {
    vdm::g_timeline_stamps = (unsigned long long*)buf;
    return VDM_OK;
}
#endif
