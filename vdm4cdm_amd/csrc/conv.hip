// conv.hip - 3D convolution as implicit GEMM on the gfx950 matrix cores (K1/K3/K4/K5 of DESIGN.md).
//
// Replaces torch.nn.functional.conv3d as reached from mltools' ResNetBlock / ResNetDown / up path
// (reference call chain: SURVEY.md section 3.2; notebook frames blocks.py:129-132,166-170).
//
// Mapping (same for fp32 and bf16 storage; accumulation always fp32):
//   D[cout][voxel] += W[cout][k] * X[k][voxel],   k = (tap, cin)
//   MFMA 16x16x32 bf16 (or 16x16x4 f32): A = packed weights (global -> VGPR, fragment order),
//   B = activations read from an LDS halo tile, D: lane (v = lane&15, q = lane>>4) owns voxel v and
//   4*NC consecutive output channels -> 16-byte NDHWC stores.
//   One workgroup = 4 waves = TZ x TY x 16 output voxels x (NC*16) output channels.
//   LDS halo image: voxel-major, 64 B per halo voxel with the four 16-B pieces x-swizzled (see stage_halo_dma), filled by
//   LDS-DMA; the 64-lane ds_read_b128 of 16 x-consecutive voxels is bank-conflict-free.
// Kernels in this file: conv_fwd_kernel (generic fwd / dgrad, incl. half-chunk workgroups for small grids), conv_kpack_kernel
//   (<= 8 reduction channels), conv_cls_kernel (per-parity-class convs: up-sampling conv fwd / dgrad, stride-2 dgrad),
//   conv_wgrad_kernel (+ class mode for the up-sampling conv) with its slab reduce kernels, the weight packers
//   (per conv and pack_many_kernel), and the host-side planning behind the C-ABI entry points.
// wgrad: D[cout][cin] += dOut^T[cout][voxel] * X[voxel][cin]; both operands are voxel-major in LDS
//   and are read transposed with ds_read_b64_tr_b16 (bf16) / ds_read_b32 (fp32); workgroups are
//   persistent over spatial tiles and write partial slabs that a second kernel reduces
//   (deterministic, no float atomics).  Conv epilogues can also reduce the GroupNorm statistics of their output.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace vdm {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static constexpr int GN_SCRATCH_BYTES = 4 * 64 * 2 * 4;      // 4 waves x (NC <= 4) * 16 channels x (sum, sumsq) floats

// ---------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------
template <int KS_, int STRIDE_, int TZ_, int TY_>
struct Geo {
    static constexpr int KS = KS_, STRIDE = STRIDE_, TZ = TZ_, TY = TY_, TX = 16;
    static constexpr int PAD = KS / 2;
    static constexpr int TAPS = KS * KS * KS;
    static constexpr int HZ = (TZ - 1) * STRIDE + KS, HY = (TY - 1) * STRIDE + KS, HX = (TX - 1) * STRIDE + KS;
    static constexpr int HVOX = HZ * HY * HX;
    static constexpr int ROWS = TZ * TY;                 // 16-voxel MFMA columns-tiles per workgroup
    static constexpr int NV = ROWS / 4;                  // per wave
    static constexpr int OVOX = ROWS * 16;
    static_assert(ROWS % 4 == 0, "rows must split over 4 waves");
};

struct ConvArgs {
    const void* x;       // staged operand (input for fwd/wgrad, dOut for dgrad)
    const void* w;       // packed weights
    const float* bias;
    const float* nbias;
    long long nbias_stride;
    const void* res;
    void* out;
    int N, Dz, Dy, Dx;   // output spatial dims
    int Iz, Iy, Ix;      // logical input grid the taps index
    int Sz, Sy, Sx;      // source tensor dims (== I, or I/2 when up-sampling)
    int Cin, CinStride;  // reduction channels, channel stride of x
    int Cout;            // output channels (exact stride of out / res)
    int circular;
    int ntz, nty, ntx, nchunks, nkb;
    float* gnp;          // optional GroupNorm partials of the output: [N][ntz*nty*ntx][Cout][2] = (sum, sum of squares) per tile
};

__device__ __forceinline__ int wrap(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}

// 64 zero bytes: source of every padded / out-of-range piece of the LDS-DMA staging below.
__device__ uint4 g_zero_page[16];

// LDS image of the forward kernel: voxel-major, 64 B per halo voxel, the four 16-B pieces of a voxel stored at
// slot = piece ^ ((hx >> 1) & 3), hx = x position inside the halo row.  With this swizzle the MFMA operand read
// (ds_read_b128, 16 x-consecutive voxels x 4 k-chunks) is bank-conflict-free for every row alignment, and the
// (dz, dy, row) shifts stay compile-time ds_read offsets.
// Filled by LDS-DMA (global_load_lds_dwordx4): one wave-instruction = 16 voxels x 64 B = 1 KiB of LDS written
// linearly; the four lanes of a voxel fetch its (permuted) pieces, i.e. whole 64-B segments of the NDHWC row.
template <typename T, typename G, int UPS>
__device__ __forceinline__ void stage_halo_dma(char* lds, const T* __restrict__ x, const ConvArgs& a, int n,
                                               int oz0, int oy0, int ox0, int kb, int wave, int lane, int nwaves = 4) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NCHUNK = (G::HVOX + 15) / 16;
    const int iz0 = oz0 * G::STRIDE - G::PAD, iy0 = oy0 * G::STRIDE - G::PAD, ix0 = ox0 * G::STRIDE - G::PAD;
    const int k = lane >> 2, j = lane & 3;
    for (int c = wave; c < NCHUNK; c += nwaves) {
        const int hv = c * 16 + k;
        const int hx = hv % G::HX;
        const int t = hv / G::HX;
        const int hy = t % G::HY;
        const int hz = t / G::HY;
        const int pc = j ^ ((hx >> 1) & 3);
        int iz = iz0 + hz, iy = iy0 + hy, ix = ix0 + hx;
        const int ci = kb * KB + pc * EPL;
        bool ok = ci < a.Cin && hv < G::HVOX;
        if (a.circular) {
            iz = wrap(iz, a.Iz); iy = wrap(iy, a.Iy); ix = wrap(ix, a.Ix);
        } else {
            ok = ok && (unsigned)iz < (unsigned)a.Iz && (unsigned)iy < (unsigned)a.Iy && (unsigned)ix < (unsigned)a.Ix;
        }
        if (UPS) { iz >>= 1; iy >>= 1; ix >>= 1; }
        const size_t off = ((((size_t)n * a.Sz + iz) * a.Sy + iy) * a.Sx + ix) * a.CinStride + ci;
        const void* src = ok ? static_cast<const void*>(x + off) : static_cast<const void*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + c * 1024), 16, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA wrappers: acc[16 cout x 16 voxel] += A(16 cout x 64 B of k) * B(64 B of k x 16 voxel)
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<bf16_t>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
}

// XCD-aware, bijective block remap: blocks b and b+8 share an XCD (observed round-robin), so give
// each XCD a contiguous run of spatial tiles (their halos overlap -> hits in that XCD's L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel
// ---------------------------------------------------------------------------------------------
// Operand-read addressing (see stage_halo_dma): address = lanex[dx] + immediate((v, dz, dy) row shifts).
// lanex[dx] = wave's first row + this lane's voxel (hx = lx*S + dx) + swizzled k-chunk slot.
template <typename G, int NV>
__device__ __forceinline__ void operand_lane_offsets(int (&lanex)[G::KS], int cwave, int lane) {
    static_assert(G::TY % NV == 0, "a wave's rows must stay inside one z-slab");
    const int lx = lane & 15, q = lane >> 4;
    const int r0 = cwave * NV;
    const int wavebase = (((r0 / G::TY) * G::STRIDE) * G::HY + (r0 % G::TY) * G::STRIDE) * G::HX * 64;
#pragma unroll
    for (int dx = 0; dx < G::KS; ++dx) {
        const int hx = lx * G::STRIDE + dx;
        lanex[dx] = wavebase + hx * 64 + ((q * 16) ^ ((hx & 6) << 3));
    }
}

template <int NC> struct WPipe { static constexpr int WPD = (NC <= 2) ? 2 : 1; };   // weight prefetch depth (taps)

// bf16: first WPD taps' weights (issued before the staging barrier so their latency overlaps it)
template <int TAPS, int NC, int WPD, int NCW = NC>
__device__ __forceinline__ void taps_prefetch_weights(uint4 (&wf)[WPD + 1][NC], const uint4* wk) {
#pragma unroll
    for (int p = 0; p < WPD && p < TAPS; ++p)
#pragma unroll
        for (int c = 0; c < NC; ++c) wf[p][c] = wk[(p * NCW + c) * 64];
}

// bf16: explicit software pipeline over the fully unrolled taps.
//   weights (global, L2-resident)  : WPD taps ahead, ring of WPD+1 register sets
//   activations (LDS)              : one tap ahead, two register sets of NV fragments
// sched_barrier(0) pins [issue next operands] | [MFMAs of this tap] so the loads stay early.
template <typename T, typename G, int NC, int NV, int WPD, int NCW = NC>
__device__ __forceinline__ void taps_pipelined(f32x4 (&acc)[NV][NC], const char* lds, const uint4* wk,
                                               uint4 (&wf)[WPD + 1][NC], const int (&lanex)[G::KS]) {
    constexpr int TAPS = G::TAPS, KS = G::KS;
    constexpr int ROWB = G::STRIDE * G::HX * 64;            // byte shift between consecutive rows v of a wave
    uint4 af[2][NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) af[0][v] = *reinterpret_cast<const uint4*>(lds + lanex[0] + v * ROWB);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        if (tap + WPD < TAPS) {
#pragma unroll
            for (int c = 0; c < NC; ++c) wf[(tap + WPD) % (WPD + 1)][c] = wk[((tap + WPD) * NCW + c) * 64];
        }
        if (tap + 1 < TAPS) {
            const int t1 = tap + 1;
            const int dz = t1 / (KS * KS), dy = (t1 / KS) % KS, dx = t1 % KS;
            const int toff = (dz * G::HY + dy) * G::HX * 64;
#pragma unroll
            for (int v = 0; v < NV; ++v) af[t1 & 1][v] = *reinterpret_cast<const uint4*>(lds + lanex[dx] + v * ROWB + toff);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int c = 0; c < NC; ++c) mma16<T>(acc[v][c], wf[tap % (WPD + 1)][c], af[tap & 1][v]);
        // Interleave: the NC weight loads first, then one LDS read of the NEXT tap after every NC MFMAs of THIS tap
        // (masks: 0x8 MFMA, 0x20 VMEM read, 0x100 DS read).  Keeps the matrix pipe fed while the loads issue.
        __builtin_amdgcn_sched_group_barrier(0x20, NC, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            __builtin_amdgcn_sched_group_barrier(0x8, NC, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// fp32 (exact v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate): MFMA-bound.  Rolled tap loop (low register pressure), the
// next tap's weights are loaded before this tap's MFMAs, and the 4 k-steps of a fragment pair are issued STEP-MAJOR over the
// NV x NC independent accumulators (a dependent fp32 MFMA has 40 cycles of latency vs 32 of issue).
template <typename T, typename G, int NC, int NV>
__device__ __forceinline__ void taps_rolled(f32x4 (&acc)[NV][NC], const char* lds, const uint4* wk, const int (&lanex)[G::KS]) {
    constexpr int TAPS = G::TAPS, KS = G::KS;
    constexpr int ROWB = G::STRIDE * G::HX * 64;
    uint4 wn[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) wn[c] = wk[c * 64];
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
        const int toff = (dz * G::HY + dy) * G::HX * 64;
        const int lx0 = (dx == 0) ? lanex[0] : ((dx == 1) ? lanex[KS > 1 ? 1 : 0] : lanex[KS > 2 ? 2 : 0]);
        uint4 wf[NC], af[NV];
#pragma unroll
        for (int c = 0; c < NC; ++c) wf[c] = wn[c];
        const int tn = tap + 1 < TAPS ? tap + 1 : tap;
#pragma unroll
        for (int c = 0; c < NC; ++c) wn[c] = wk[(tn * NC + c) * 64];
#pragma unroll
        for (int v = 0; v < NV; ++v) af[v] = *reinterpret_cast<const uint4*>(lds + lx0 + v * ROWB + toff);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const uint32_t aw = st == 0 ? wf[c].x : (st == 1 ? wf[c].y : (st == 2 ? wf[c].z : wf[c].w));
                    const uint32_t bw = st == 0 ? af[v].x : (st == 1 ? af[v].y : (st == 2 ? af[v].z : af[v].w));
                    acc[v][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, aw), __builtin_bit_cast(float, bw), acc[v][c], 0, 0, 0);
                }
    }
}

// GroupNorm statistics fused into the producing conv: per-channel (sum, sum of squares) of the output values of one tile, taken
// from the fp32 results before they are rounded for storage (the rounding errors are zero-mean: the moments of the stored tensor
// differ by ~2^-9 / sqrt(#voxels) relative).  Lane sums over its rows -> xor-butterfly over the 16 voxel lanes -> the four waves fold through a small LDS
// scratch in a fixed order (deterministic) -> one partial per (sample, tile, channel).  vdm_gn_stats_from_partials sums the
// tiles and the channels of a group.  `sm` = NC*16*2*4 floats of LDS that no wave still reads as operand image.
// sum over the 16 lanes of a DPP row (= the 16 voxels of an MFMA column block), result in every lane: 4 x v_add_f32 with a
// rotated second operand (row_ror:8/4/2/1) - no LDS crossbar traffic
__device__ __forceinline__ float row16_sum(float v) {
#define VDM_ROR_ADD(n)                                                                                                          \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
    VDM_ROR_ADD(8);
    VDM_ROR_ADD(4);
    VDM_ROR_ADD(2);
    VDM_ROR_ADD(1);
#undef VDM_ROR_ADD
    return v;
}

template <int NC>
__device__ __forceinline__ void gn_partials_reduce(float (&gs)[NC * 4], float (&gq)[NC * 4], float* sm, float* dst /* [Cout][2] of this tile */,
                                                   int cout0, int Cout, int wave, int lane, int qstride = NC * 4) {
    const int lx = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < NC * 4; ++j) {
        gs[j] = row16_sum(gs[j]);
        gq[j] = row16_sum(gq[j]);
    }
    if (lx == 0) {
#pragma unroll
        for (int j = 0; j < NC * 4; ++j) {
            sm[((wave * NC * 16) + q * NC * 4 + j) * 2] = gs[j];
            sm[((wave * NC * 16) + q * NC * 4 + j) * 2 + 1] = gq[j];
        }
    }
    __syncthreads();
    const int t = wave * 64 + lane;
    if (t < NC * 16 * 2) {
        const int c = ((t >> 1) / (NC * 4)) * qstride + (t >> 1) % (NC * 4);      // lane group q owns NC*4 channels every qstride
        const float tot = (sm[t] + sm[NC * 16 * 2 + t]) + (sm[2 * NC * 16 * 2 + t] + sm[3 * NC * 16 * 2 + t]);
        if (cout0 + c < Cout) dst[(size_t)(cout0 + c) * 2 + (t & 1)] = tot;
    }
}

// epilogue: + bias + per-sample conditioning bias + residual, cast, 16-byte NDHWC stores
template <typename T, typename TO, typename G, int NC, int NV>
__device__ __forceinline__ void conv_epilogue(const f32x4 (&acc)[NV][NC], const ConvArgs& a, int n, int chunk, int oz0, int oy0,
                                              int ox0, int cwave, int lane, float* gn_sm = nullptr, int tile = 0, int cout0 = -1,
                                              int qstride = NC * 4) {
    constexpr int EPL = DT<T>::EPL;
    float gs[NC * 4], gq[NC * 4];
#pragma unroll
    for (int j = 0; j < NC * 4; ++j) gs[j] = gq[j] = 0.f;
    const int lx = lane & 15, q = lane >> 4;
    if (cout0 < 0) cout0 = chunk * NC * 16;             // (half-chunk kernels pass their own origin and lane-group stride)
    const int cbase = cout0 + q * qstride;              // first of this lane's NC*4 consecutive couts
    float badd[NC * 4];
#pragma unroll
    for (int j = 0; j < NC * 4; ++j) {
        float bv = 0.f;
        if (cbase + j < a.Cout) {
            if (a.bias) bv += a.bias[cbase + j];
            if (a.nbias) bv += a.nbias[(size_t)n * a.nbias_stride + cbase + j];
        }
        badd[j] = bv;
    }
    const bool vec_ok = (a.Cout % (NC * 4) == 0) && (cbase + NC * 4 <= a.Cout);
    TO* out = reinterpret_cast<TO*>(a.out);
    const T* res = reinterpret_cast<const T*>(a.res);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int r = cwave * NV + v;
        const int oz = oz0 + r / G::TY, oy = oy0 + r % G::TY, ox = ox0 + lx;
        if (oz >= a.Dz || oy >= a.Dy || ox >= a.Dx) continue;
        const size_t vo = ((((size_t)n * a.Dz + oz) * a.Dy + oy) * a.Dx + ox) * a.Cout + cbase;
        float val[NC * 4];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) val[c * 4 + j] = acc[v][c][j] + badd[c * 4 + j];
        if (a.gnp && !vec_ok) {                           // (scalar tail path: residual is added below, element by element)
#pragma unroll
            for (int j = 0; j < NC * 4; ++j)
                if (cbase + j < a.Cout) {
                    const float r = val[j] + (res ? ld_elem<T>(res + vo + j) : 0.f);
                    gs[j] += r; gq[j] += r * r;
                }
        }
        if (vec_ok) {
            if (res) {
                constexpr int RP = NC * 4 / EPL > 0 ? NC * 4 / EPL : 1;    // 16-B pieces (bf16 NC=1: half piece)
                if (NC * 4 >= EPL) {
#pragma unroll
                    for (int i = 0; i < RP; ++i) {
                        Piece<T> pr;
                        pr.load(*reinterpret_cast<const uint4*>(res + vo + i * EPL));
#pragma unroll
                        for (int j = 0; j < EPL; ++j) val[i * EPL + j] += pr.f[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NC * 4; ++j) val[j] += ld_elem<T>(res + vo + j);
                }
            }
            if (a.gnp) {
#pragma unroll
                for (int j = 0; j < NC * 4; ++j) { gs[j] += val[j]; gq[j] += val[j] * val[j]; }
            }
            if (sizeof(TO) == 4) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + vo + c * 4) =
                        make_float4(val[c * 4], val[c * 4 + 1], val[c * 4 + 2], val[c * 4 + 3]);
            } else {
                uint16_t* o16 = reinterpret_cast<uint16_t*>(out) + vo;
                if (NC == 1) {
                    *reinterpret_cast<uint2*>(o16) = make_uint2(pack_bf16x2(val[0], val[1]), pack_bf16x2(val[2], val[3]));
                } else {
#pragma unroll
                    for (int i = 0; i < NC / 2; ++i)
                        *reinterpret_cast<uint4*>(o16 + i * 8) =
                            make_uint4(pack_bf16x2(val[i * 8], val[i * 8 + 1]), pack_bf16x2(val[i * 8 + 2], val[i * 8 + 3]),
                                       pack_bf16x2(val[i * 8 + 4], val[i * 8 + 5]), pack_bf16x2(val[i * 8 + 6], val[i * 8 + 7]));
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NC * 4; ++j) {
                if (cbase + j < a.Cout) {
                    float o = val[j];
                    if (res) o += ld_elem<T>(res + vo + j);
                    st_elem<TO>(out + vo + j, o);
                }
            }
        }
    }
    if (a.gnp)                                            // workgroup-uniform
        gn_partials_reduce<NC>(gs, gq, gn_sm, a.gnp + ((size_t)n * (a.ntz * a.nty * a.ntx) + tile) * a.Cout * 2, cout0, a.Cout,
                               cwave, lane, qstride);
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel, one tile per workgroup (all variants; small grids)
// ---------------------------------------------------------------------------------------------
// SPLIT (NC == 2 only): the weights are packed for 64-cout chunks (4 tiles per tap) but a workgroup takes HALF a chunk (tiles 2h,
// 2h+1 = couts 16q + 8h .. +7 of every lane group q): twice the workgroups for the small grids of the deep levels, where a
// workgroup's K-blocks run strictly one after the other and only co-resident workgroups overlap staging with MFMAs.
template <typename T, typename TO, int KS, int STRIDE, int UPS, int NC, int TZ, int TY, bool SPLIT = false>
__global__ void __launch_bounds__(256, (TZ * TY <= 16 && NC <= 2) ? 3 : 2) conv_fwd_kernel(const ConvArgs a) {
    static_assert(!SPLIT || NC == 2, "half-chunk mode is the NC=2 kernel on NC=4 weights");
    constexpr int NCW = SPLIT ? 4 : NC;
    using G = Geo<KS, STRIDE, TZ, TY>;
    constexpr int NV = G::NV, TAPS = G::TAPS;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = b % a.ntx; b /= a.ntx;
    const int ty = b % a.nty; b /= a.nty;
    const int tz = b % a.ntz; b /= a.ntz;
    const int n = b % a.N;
    const int chunk = b / a.N;
    const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * 16;

    f32x4 acc[NV][NC];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int lanex[KS];
    operand_lane_offsets<G, NV>(lanex, wave, lane);

    const uint4* wbase = reinterpret_cast<const uint4*>(a.w) + (size_t)(SPLIT ? chunk >> 1 : chunk) * a.nkb * TAPS * NCW * 64 +
                         (SPLIT ? (chunk & 1) * 2 * 64 : 0) + lane;
    const T* x = reinterpret_cast<const T*>(a.x);

    for (int kb = 0; kb < a.nkb; ++kb) {
        if (kb) __syncthreads();
        stage_halo_dma<T, G, UPS>(lds, x, a, n, oz0, oy0, ox0, kb, wave, lane);
        const uint4* wk = wbase + (size_t)kb * TAPS * NCW * 64;
        if constexpr (sizeof(T) == 2) {
            constexpr int WPD = WPipe<NC>::WPD;
            uint4 wf[WPD + 1][NC];
            taps_prefetch_weights<TAPS, NC, WPD, NCW>(wf, wk);
            __syncthreads();
            taps_pipelined<T, G, NC, NV, WPD, NCW>(acc, lds, wk, wf, lanex);
        } else {
            __syncthreads();
            taps_rolled<T, G, NC, NV>(acc, lds, wk, lanex);
        }
    }
    constexpr int IMG = ((G::HVOX + 15) / 16) * 1024;     // the GN scratch sits behind the operand image
    if constexpr (SPLIT)
        conv_epilogue<T, TO, G, NC, NV>(acc, a, n, chunk, oz0, oy0, ox0, wave, lane, reinterpret_cast<float*>(lds + IMG),
                                        (tz * a.nty + ty) * a.ntx + tx, (chunk >> 1) * 64 + (chunk & 1) * 8, 16);
    else
        conv_epilogue<T, TO, G, NC, NV>(acc, a, n, chunk, oz0, oy0, ox0, wave, lane, reinterpret_cast<float*>(lds + IMG),
                                        (tz * a.nty + ty) * a.ntx + tx);
}

// ---------------------------------------------------------------------------------------------
// Tap-packed kernel for the convs with <= 8 input channels (conv_in: 2 -> 32; input gradient of conv_out: 1 -> 32), bf16.
// One 16-byte piece holds ALL channels of a voxel, so the 32-deep K of an MFMA is filled with 4 TAPS x 8 channels instead of
// one tap x 32 channels of which 24..31 are padding: 7 tap groups instead of 27 taps (3.9x fewer MFMAs), an LDS image of
// 16 B per halo voxel (17 KB), one LDS-DMA lane per voxel.  Lane (voxel lx, k-chunk q) reads the voxel shifted by tap 4g+q;
// the packed weights hold W[tap 4g+q][cout][ci] in the matching A-fragment slot (zero for tap >= 27, ci >= Cin).
// These convs are bound by writing / reading the 32-channel tensor (HBM), not by MFMA.
// ---------------------------------------------------------------------------------------------
template <typename TO, int NC>
__global__ void __launch_bounds__(256, 4) conv_kpack_kernel(const ConvArgs a) {
    using T = bf16_t;
    using G = Geo<3, 1, 4, 8>;
    constexpr int NV = G::NV, NG = (G::TAPS + 3) / 4;
    constexpr int NCH = (G::HVOX + 63) / 64;               // DMA chunks of 64 voxels (1 KiB)
    constexpr int ROWB = G::HX * 16;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = b % a.ntx; b /= a.ntx;
    const int ty = b % a.nty; b /= a.nty;
    const int tz = b % a.ntz; b /= a.ntz;
    const int n = b % a.N;
    const int chunk = b / a.N;
    const int oz0 = tz * G::TZ, oy0 = ty * G::TY, ox0 = tx * 16;

    // stage the halo: one lane per voxel
    const T* x = reinterpret_cast<const T*>(a.x);
    for (int c = wave; c < NCH; c += 4) {
        const int hv = c * 64 + lane;
        const int hx = hv % G::HX;
        const int t = hv / G::HX;
        const int hy = t % G::HY;
        const int hz = t / G::HY;
        int iz = oz0 - 1 + hz, iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        bool ok = hv < G::HVOX;
        if (a.circular) {
            iz = wrap(iz, a.Iz); iy = wrap(iy, a.Iy); ix = wrap(ix, a.Ix);
        } else {
            ok = ok && (unsigned)iz < (unsigned)a.Iz && (unsigned)iy < (unsigned)a.Iy && (unsigned)ix < (unsigned)a.Ix;
        }
        const size_t off = ((((size_t)n * a.Sz + iz) * a.Sy + iy) * a.Sx + ix) * a.CinStride;
        const void* src = ok ? static_cast<const void*>(x + off) : static_cast<const void*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + c * 1024), 16, 0, 0);
    }
    // weights of this cout chunk: NG x NC fragments
    uint4 wf[NG][NC];
    {
        const uint4* wk = reinterpret_cast<const uint4*>(a.w) + (size_t)chunk * NG * NC * 64 + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int c = 0; c < NC; ++c) wf[g][c] = wk[(g * NC + c) * 64];
    }
    // operand addresses: wave's first row + this lane's voxel + the shift of tap 4g + q
    const int lx = lane & 15, q = lane >> 4;
    const int r0 = wave * NV;
    const int base = (((r0 / G::TY) * G::HY + (r0 % G::TY)) * G::HX + lx) * 16;
    int goff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int tap = 4 * g + q;
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        goff[g] = tap < G::TAPS ? base + ((dz * G::HY + dy) * G::HX + dx) * 16 : base;      // (tap 27: zero weights)
    }
    f32x4 acc[NV][NC];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        uint4 af[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) af[v] = *reinterpret_cast<const uint4*>(lds + goff[g] + v * ROWB);
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int c = 0; c < NC; ++c) mma16<T>(acc[v][c], wf[g][c], af[v]);
    }
    constexpr int IMG = NCH * 1024;
    conv_epilogue<T, TO, G, NC, NV>(acc, a, n, chunk, oz0, oy0, ox0, wave, lane, reinterpret_cast<float*>(lds + IMG),
                                    (tz * a.nty + ty) * a.ntx + tx);
}

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
struct ClsEntry { int lds_off; int dx; };                   // halo offset ((dz*HY+dy)*HX)*64 bytes and dx in 0..2
struct ClsTable {
    int n[8];                                                // entries per class
    ClsEntry e[8][8];
};
struct ClsArgs {
    ConvArgs c;                                              // Dz/Dy/Dx = COARSE tile index space; out dims in oD*
    ClsTable t;
    int oDz, oDy, oDx;                                       // output tensor spatial dims (fine for MODE_F, coarse for MODE_B)
    int sDz, sDy, sDx;                                       // staged source tensor spatial dims
};

// LDS-DMA staging of a halo tile whose logical voxel i maps to source voxel ss*i + so (class sub-grid when ss == 2).
template <typename T, typename G>
__device__ __forceinline__ void stage_halo_dma_sub(char* lds, const T* __restrict__ x, const ClsArgs& ca, int n, int oz0, int oy0,
                                                   int ox0, int kb, int ss, int soz, int soy, int sox, int wave, int lane) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NCHUNK = (G::HVOX + 15) / 16;
    const ConvArgs& a = ca.c;
    const int iz0 = oz0 - G::PAD, iy0 = oy0 - G::PAD, ix0 = ox0 - G::PAD;
    const int k = lane >> 2, j = lane & 3;
    for (int c = wave; c < NCHUNK; c += 4) {
        const int hv = c * 16 + k;
        const int hx = hv % G::HX;
        const int t = hv / G::HX;
        const int hy = t % G::HY;
        const int hz = t / G::HY;
        const int pc = j ^ ((hx >> 1) & 3);
        int iz = iz0 + hz, iy = iy0 + hy, ix = ix0 + hx;
        const int ci = kb * KB + pc * EPL;
        bool ok = ci < a.Cin && hv < G::HVOX;
        if (a.circular) {
            iz = wrap(iz, a.Iz); iy = wrap(iy, a.Iy); ix = wrap(ix, a.Ix);
        } else {
            ok = ok && (unsigned)iz < (unsigned)a.Iz && (unsigned)iy < (unsigned)a.Iy && (unsigned)ix < (unsigned)a.Ix;
        }
        const size_t off = ((((size_t)n * ca.sDz + (ss * iz + soz)) * ca.sDy + (ss * iy + soy)) * ca.sDx + (ss * ix + sox)) * a.CinStride + ci;
        const void* src = ok ? static_cast<const void*>(x + off) : static_cast<const void*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + c * 1024), 16, 0, 0);
    }
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
    const int tx = b % a.ntx; b /= a.ntx;
    const int ty = b % a.nty; b /= a.nty;
    const int tz = b % a.ntz; b /= a.ntz;
    const int n = b % a.N;
    const int chunk = b / a.N;
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

// packed[chunk][kb][slot 0..63][ct][lane][EPL] = sum over the master taps in mask[slot] of W (transpose: W[t][k][o]).
struct ClsMasks { unsigned m[64]; };
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
        st_elem<T>(p + i, v);
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad kernel
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    ConvArgs c;          // c.x = input, c.Cin/CinStride = input channels, c.Cout = dOut channels
    const void* dout;
    int dout_stride;
    float* slabs;        // [pair][P][slot]... see wgrad_reduce
    float* bslabs;       // optional [cout block][P][CL] partial column sums of dOut (bias gradient), or NULL
    int P;               // persistent workgroups per (cout block, cin block) pair
    int ntiles;          // N * ntz * nty * ntx
    int ncb, nkb;        // cout blocks, cin blocks (64 B each)
};

template <typename T> struct WG;     // 16x16 tiles per 64-byte channel block
template <> struct WG<bf16_t> { static constexpr int NT = 2; };
template <> struct WG<float> { static constexpr int NT = 1; };

// dOut tile of a wgrad workgroup by LDS-DMA: OVOX voxels x 64 B (one cout block), same x-swizzled voxel-major
// image as stage_halo_dma (one chunk = one 16-voxel row).
template <typename T, typename G>
__device__ __forceinline__ void stage_dout_dma(char* lds, const T* __restrict__ g, const ConvArgs& a, int n, int oz0, int oy0,
                                               int ox0, int cb, int cstride, int wave, int lane) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    const int k = lane >> 2, j = lane & 3;
    const int pc = j ^ ((k >> 1) & 3);
    const int co = cb * KB + pc * EPL;
    for (int r = wave; r < G::ROWS; r += 4) {
        const int oz = oz0 + r / G::TY, oy = oy0 + r % G::TY, ox = ox0 + k;
        const bool ok = co < a.Cout && oz < a.Dz && oy < a.Dy && ox < a.Dx;
        const size_t off = ((((size_t)n * a.Dz + oz) * a.Dy + oy) * a.Dx + ox) * cstride + co;
        const void* src = ok ? static_cast<const void*>(g + off) : static_cast<const void*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + r * 1024), 16, 0, 0);
    }
}

// Transposed operand fetch from the x-swizzled voxel-major image: 16 channels (tile ct of the 64-B block) x the
// k-step's voxels.  address = LDS base + wave-uniform row/tap offset (uni) + per-lane offset(s) (computed once per tap).
//   bf16: NOFF = 1, two ds_read_b64_tr_b16 (rows r and r+1; the second row is a compile-time byte delta HI);
//         lane: g = lane>>4 (voxels 4g..4g+3), li = lane&15: voxel-in-quad q' = li>>2, column quad p = li&3.
//         Voxels x and x+4 of a 32-lane half use opposite piece pairs ((hx>>1)&3 differs by 2): conflict-free.
//   fp32: NOFF = 4, four ds_read_b32 (MFMA step s reads voxel x = 4*s + (lane>>4), channel lane&15).
// dOut tile of one parity class of the up-sampling conv: the tile's coarse voxels c map to the fine voxels 2c + p.
template <typename T, typename G>
__device__ __forceinline__ void stage_dout_dma_sub(char* lds, const T* __restrict__ g, const ConvArgs& a, int n, int oz0, int oy0,
                                                   int ox0, int cb, int cstride, int pz, int py, int px, int wave, int lane) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    const int k = lane >> 2, j = lane & 3;
    const int pc = j ^ ((k >> 1) & 3);
    const int co = cb * KB + pc * EPL;
    for (int r = wave; r < G::ROWS; r += 4) {
        const int oz = oz0 + r / G::TY, oy = oy0 + r % G::TY, ox = ox0 + k;
        const bool ok = co < a.Cout && oz < a.Dz && oy < a.Dy && ox < a.Dx;
        const size_t off = ((((size_t)n * (2 * a.Dz) + 2 * oz + pz) * (2 * a.Dy) + 2 * oy + py) * (2 * a.Dx) + 2 * ox + px) * cstride + co;
        const void* src = ok ? static_cast<const void*>(g + off) : static_cast<const void*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + r * 1024), 16, 0, 0);
    }
}

template <typename T> struct TrFetch;
template <> struct TrFetch<bf16_t> {
    static constexpr int NOFF = 1;
    static __device__ __forceinline__ void lane_off(int (&o)[1], int ct, int xs, int dx, int lane) {
        const int g = lane >> 4, li = lane & 15, qp = li >> 2, p = li & 3;
        const int hx = (4 * g + qp) * xs + dx;
        const int slot = (2 * ct + (p >> 1)) ^ ((hx >> 1) & 3);
        o[0] = hx * 64 + slot * 16 + (p & 1) * 8;
    }
    template <int HI>
    static __device__ __forceinline__ uint4 get(const char* lds, const int (&o)[1], int uni) {
        typedef __attribute__((address_space(3))) s16x4* lptr;
        const char* p = lds + uni + o[0];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p + HI));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    }
};
template <> struct TrFetch<float> {
    static constexpr int NOFF = 4;
    static __device__ __forceinline__ void lane_off(int (&o)[4], int ct, int xs, int dx, int lane) {
        (void)ct;
        const int m = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int hx = (4 * st + kq) * xs + dx;
            o[st] = hx * 64 + (((m >> 2) ^ ((hx >> 1) & 3)) * 16) + (m & 3) * 4;
        }
    }
    template <int HI>
    static __device__ __forceinline__ uint4 get(const char* lds, const int (&o)[4], int uni) {
        const char* p = lds + uni;
        uint4 r;
        r.x = *reinterpret_cast<const uint32_t*>(p + o[0]);
        r.y = *reinterpret_cast<const uint32_t*>(p + o[1]);
        r.z = *reinterpret_cast<const uint32_t*>(p + o[2]);
        r.w = *reinterpret_cast<const uint32_t*>(p + o[3]);
        return r;
    }
};

// NTA / NTB: 16-channel tiles of the 64-byte cout / cin block that hold real channels (conv_out has 1 output channel, conv_in 2
// input channels: half of the MFMAs and transposed reads of the block would multiply padding).
template <typename T, int KS, int STRIDE, int UPS, int TZ, int TY, int NTA = WG<T>::NT, int NTB = WG<T>::NT>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(const WgradArgs w) {
    using G = Geo<KS, STRIDE, TZ, TY>;
    using TF = TrFetch<T>;
    constexpr int NT = WG<T>::NT, NOFF = TF::NOFF;
    // UPS == 2: one parity class of the up-sampling conv on the COARSE grid (see the class convs above): 8 merged taps
    // e = (ez, ey, ex) in {0,1}^3 at halo offsets d = e + p, dOut = the class sub-grid dOut[2c + p]; the master-tap gradients are
    // recombined by wgrad_cls_reduce_kernel.  27/8 = 3.4x fewer MFMAs than 27 taps on the fine grid.
    constexpr bool CLS = (UPS == 2);
    constexpr int TAPS = CLS ? 8 : G::TAPS;
    constexpr int TPW = (TAPS + 3) / 4;                    // taps per wave (KS=3: 7; class mode: 2; KS=1: 1)
    constexpr int RSTEP = (sizeof(T) == 2) ? 2 : 1;        // rows consumed per k-step
    constexpr int IN_BYTES = ((G::HVOX + 15) / 16) * 1024;
    static_assert(sizeof(T) == 4 || (TY % 2) == 0, "bf16 k-step = two rows of the same z-slab");
    constexpr int HI_IN = STRIDE * G::HX * 64;             // byte delta to the second row of a bf16 k-step
    constexpr int HI_DO = 16 * 64;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* lds_in = lds;
    char* lds_do = lds + IN_BYTES;
    const ConvArgs& a = w.c;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int pair = blockIdx.x / w.P, pidx = blockIdx.x % w.P;
    const int cls = CLS ? pair / (w.ncb * w.nkb) : 0;      // parity class (pz, py, px)
    const int pr = CLS ? pair % (w.ncb * w.nkb) : pair;
    const int cb = pr / w.nkb, kb = pr % w.nkb;            // cout block, cin block
    const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;

    f32x4 acc[TPW][NTA][NTB];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < NTA; ++i)
#pragma unroll
            for (int j = 0; j < NTB; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // KS=3: wave owns taps wave, wave+4, ... over all rows.  KS=1: all waves own tap 0, rows split.
    int tap_u[TPW];                       // wave-uniform byte offset of the tap's (dz, dy) shift
    int lo_in[TPW][NTB][NOFF];             // per-lane offsets (depend on the tap's dx through the x-swizzle)
    int lo_do[NTA][NOFF];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        int tap = (TAPS > 1) ? wave + 4 * t : 0;
        if (tap >= TAPS) tap = TAPS - 1;                   // dummy (result discarded)
        const int dz = CLS ? ((tap >> 2) & 1) + pz : tap / (KS * KS), dy = CLS ? ((tap >> 1) & 1) + py : (tap / KS) % KS,
                  dx = CLS ? (tap & 1) + px : tap % KS;
        tap_u[t] = (dz * G::HY + dy) * G::HX * 64;
#pragma unroll
        for (int j = 0; j < NTB; ++j) TF::lane_off(lo_in[t][j], j, STRIDE, dx, lane);
    }
#pragma unroll
    for (int i = 0; i < NTA; ++i) TF::lane_off(lo_do[i], i, 1, 0, lane);
    const int row0 = (TAPS > 1) ? 0 : wave * RSTEP;
    const int rowinc = (TAPS > 1) ? RSTEP : 4 * RSTEP;

    const T* x = reinterpret_cast<const T*>(a.x);
    const T* g = reinterpret_cast<const T*>(w.dout);
    float bsum[DT<T>::EPL];
#pragma unroll
    for (int j = 0; j < DT<T>::EPL; ++j) bsum[j] = 0.f;
    auto in_base = [&](int r) { return (((r / TY) * STRIDE * G::HY + (r % TY) * STRIDE) * G::HX) * 64; };

    for (int tile = pidx; tile < w.ntiles; tile += w.P) {
        int b = tile;
        const int tx = b % a.ntx; b /= a.ntx;
        const int ty = b % a.nty; b /= a.nty;
        const int tz = b % a.ntz; b /= a.ntz;
        const int n = b;
        const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * 16;
        __syncthreads();                                   // every wave is done reading the previous tile
        stage_halo_dma<T, G, CLS ? 0 : UPS>(lds_in, x, a, n, oz0, oy0, ox0, kb, wave, lane);
        if constexpr (CLS)
            stage_dout_dma_sub<T, G>(lds_do, g, a, n, oz0, oy0, ox0, cb, w.dout_stride, pz, py, px, wave, lane);
        else
            stage_dout_dma<T, G>(lds_do, g, a, n, oz0, oy0, ox0, cb, w.dout_stride, wave, lane);
        __syncthreads();                                   // (drains the LDS-DMA: vmcnt(0) + barrier)
        // Software pipeline: while the MFMAs of tap t run, the transposed fragments of tap t+1 (or of the next
        // row's tap 0 and its dOut fragments) are already in flight; sched_barrier(0) pins that order.
        uint4 af[NTA], afn[NTA], bfA[NTB], bfB[NTB];
#pragma unroll
        for (int i = 0; i < NTA; ++i) af[i] = TF::template get<HI_DO>(lds_do, lo_do[i], row0 * 1024);
#pragma unroll
        for (int j = 0; j < NTB; ++j) bfA[j] = TF::template get<HI_IN>(lds_in, lo_in[0][j], in_base(row0) + tap_u[0]);
        for (int r = row0; r < G::ROWS; r += rowinc) {
            const int rn = (r + rowinc < G::ROWS) ? r + rowinc : r;      // clamp: the last prefetch is harmless
            const int i0 = in_base(r), in0 = in_base(rn);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                uint4 (&cur)[NTB] = (t & 1) ? bfB : bfA;
                uint4 (&nxt)[NTB] = (t & 1) ? bfA : bfB;
                if (t + 1 < TPW) {
#pragma unroll
                    for (int j = 0; j < NTB; ++j) nxt[j] = TF::template get<HI_IN>(lds_in, lo_in[t + 1 < TPW ? t + 1 : 0][j], i0 + tap_u[t + 1 < TPW ? t + 1 : 0]);
                } else {
#pragma unroll
                    for (int i = 0; i < NTA; ++i) afn[i] = TF::template get<HI_DO>(lds_do, lo_do[i], rn * 1024);
#pragma unroll
                    for (int j = 0; j < NTB; ++j) nxt[j] = TF::template get<HI_IN>(lds_in, lo_in[0][j], in0 + tap_u[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NTA; ++i)
#pragma unroll
                    for (int j = 0; j < NTB; ++j) mma16<T>(acc[t][i][j], af[i], cur[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NTA; ++i) af[i] = afn[i];
            if (TPW & 1) {
#pragma unroll
                for (int j = 0; j < NTB; ++j) bfA[j] = bfB[j];
            }
        }
        // bias gradient: column sums of this dOut tile (already in LDS), by the workgroups with cin block 0; every wave
        // takes a quarter of the rows.  Lane l sums the 16-B slot (l & 3) of voxels x = l >> 2: with the x-swizzle that
        // is always the same channel piece, so the sums stay in EPL registers until the kernel ends.
        if (TAPS > 1 && w.bslabs != nullptr && kb == 0) {
            const int vx = lane >> 2, sl = lane & 3;
#pragma unroll
            for (int r = wave; r < G::ROWS; r += 4) {
                Piece<T> pz;
                pz.load(*reinterpret_cast<const uint4*>(lds_do + r * 1024 + vx * 64 + sl * 16));
#pragma unroll
                for (int j = 0; j < DT<T>::EPL; ++j) bsum[j] += pz.f[j];
            }
        }
    }

    constexpr int CL = NT * 16;                               // channels per 64-B block
    if (TAPS > 1 && w.bslabs != nullptr && kb == 0) {          // workgroup-uniform condition
        float* shb = reinterpret_cast<float*>(lds);           // all tiles are done: the LDS image is free
        __syncthreads();
        if (tid < CL) shb[tid] = 0.f;
        __syncthreads();
        {
            const int piece = (lane & 3) ^ (((lane >> 2) >> 1) & 3);
#pragma unroll
            for (int j = 0; j < DT<T>::EPL; ++j) atomicAdd(&shb[piece * DT<T>::EPL + j], bsum[j]);
        }
        __syncthreads();
        if (tid < CL) w.bslabs[((size_t)cb * (CLS ? 8 : 1) * w.P + cls * w.P + pidx) * CL + tid] = shb[tid];
    }
    // ---- write this wave's partial tiles: slab[tap][co_local][ci_local] ------------------------
    constexpr int SLAB = TAPS * CL * CL;
    const int nslab_per_wg = (TAPS > 1) ? 1 : 4;
    float* slab = w.slabs + ((size_t)(pair * w.P + pidx) * nslab_per_wg + ((TAPS > 1) ? 0 : wave)) * SLAB;
    const int gq = lane >> 4, col = lane & 15;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = (TAPS > 1) ? wave + 4 * t : 0;
        if (tap >= TAPS) continue;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    slab[(tap * CL + i * 16 + gq * 4 + rg) * CL + j * 16 + col] = (i < NTA && j < NTB) ? acc[t][i < NTA ? i : 0][j < NTB ? j : 0][rg] : 0.f;
    }
}

// dw[tap][co][ci] (+)= sum over slabs.  Block = 64 outputs x 4 slab groups; each thread sums its slabs with 8
// independent loads in flight, then the 4 groups are combined through LDS in a fixed order (deterministic).
constexpr int WRED_OUT = 64, WRED_GRP = 4;                  // outputs per block x slab groups (256 threads); 32 x 8 is not faster
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int taps,
                                                          int cout, int cin, int ncb, int nkb, int CL, int nslabs, int accumulate) {
    const int total = taps * cout * cin;
    const int o = threadIdx.x % WRED_OUT, sg = threadIdx.x / WRED_OUT;
    const int i = blockIdx.x * WRED_OUT + o;
    __shared__ float part[WRED_GRP][WRED_OUT];
    float sum = 0.f;
    if (i < total) {
        const int ci = i % cin, co = (i / cin) % cout, tap = i / (cin * cout);
        const int pair = (co / CL) * nkb + ci / CL;
        const size_t slab_elems = (size_t)taps * CL * CL;
        const float* s = slabs + (size_t)pair * nslabs * slab_elems + ((size_t)tap * CL + co % CL) * CL + ci % CL;
        int k = sg;
        for (; k + 7 * WRED_GRP < nslabs; k += 8 * WRED_GRP) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s[(size_t)(k + WRED_GRP * u) * slab_elems];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; k < nslabs; k += WRED_GRP) sum += s[(size_t)k * slab_elems];
    }
    part[sg][o] = sum;
    __syncthreads();
    if (sg == 0 && i < total) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < WRED_GRP; ++g) tot += part[g][o];
        dw[i] = accumulate ? dw[i] + tot : tot;
    }
}
__global__ void __launch_bounds__(256) wgrad_bias_reduce_kernel(const float* __restrict__ bslabs, float* __restrict__ dbias, int cout,
                                                               int CL, int P, int accumulate) {
    const int c = threadIdx.x & 15, gp = threadIdx.x >> 4;
    const int co = blockIdx.x * 16 + c;
    __shared__ float part[16][17];
    float sum = 0.f;
    if (co < cout) {
        const float* s = bslabs + (size_t)(co / CL) * P * CL + co % CL;
        for (int p = gp; p < P; p += 16) sum += s[(size_t)p * CL];
    }
    part[gp][c] = sum;
    __syncthreads();
    if (gp == 0 && co < cout) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += part[k][c];
        dbias[co] = accumulate ? dbias[co] + tot : tot;
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing: master fp32 [taps][cout][cin] -> MFMA A-fragment order
//   packed[chunk][kb][tap][ct][lane][EPL]: lane (m = lane&15, q = lane>>4), element j:
//     out channel o = chunk*NC*16 + NC*4*(m>>2) + 4*ct + (m&3) ; reduction channel k = kb*KB + q*EPL + j
//   fwd  : W[tap][o][k]                       dgrad: W[flip(tap)][k][o]  (o indexes cin, k indexes cout)
// ---------------------------------------------------------------------------------------------
// Master-tap gradients of the up-sampling conv from the class slabs: dW[t] = sum over the (class p, entry e) whose merged tap
// contains t - per dimension t=0: (p0,e0),(p1,e0); t=1: (p0,e1),(p1,e0); t=2: (p0,e1),(p1,e1), i.e. p = b, e = (t + 1 - b) / 2 for
// b in {0,1} - and over the P persistent workgroups of each; fixed order (deterministic).
__global__ void __launch_bounds__(256) wgrad_cls_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int cout, int cin,
                                                              int ncb, int nkb, int CL, int P, int accumulate) {
    const int total = 27 * cout * cin;
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    __shared__ float part[4][64];
    float sum = 0.f;
    if (i < total) {
        const int ci = i % cin, co = (i / cin) % cout, tap = i / (cin * cout);
        const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
        const size_t slab_elems = (size_t)8 * CL * CL;
        for (int b = 0; b < 8; ++b) {
            const int bz = (b >> 2) & 1, by = (b >> 1) & 1, bx = b & 1;
            const int cls = b, e = (((tz + 1 - bz) >> 1) << 2) | (((ty + 1 - by) >> 1) << 1) | ((tx + 1 - bx) >> 1);
            const int pair = (cls * ncb + co / CL) * nkb + ci / CL;
            const float* s = slabs + (size_t)pair * P * slab_elems + ((size_t)e * CL + co % CL) * CL + ci % CL;
            for (int k = sg; k < P; k += 4) sum += s[(size_t)k * slab_elems];
        }
    }
    part[sg][o] = sum;
    __syncthreads();
    if (sg == 0 && i < total) {
        const float tot = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
        dw[i] = accumulate ? dw[i] + tot : tot;
    }
}

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
        st_elem<T>(p + i, v);
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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// output-channel tiles (of 16) per workgroup.  fp32 is MFMA-bound (weight reuse is irrelevant) and its deep-level grids are
// small: cap at 2 so that twice as many workgroups exist.
static int nc_for(int cout, int dtype) { return cout <= 16 ? 1 : ((cout <= 32 || dtype == VDM_F32) ? 2 : 4); }
static int epl_of(int dtype) { return dtype == VDM_F32 ? 4 : 8; }
static int kb_of(int dtype) { return dtype == VDM_F32 ? 16 : 32; }
static int cpad(int c, int dtype) { const int e = epl_of(dtype); return (c + e - 1) / e * e; }

static int validate(const vdm_conv_desc* d) {
    VDM_REQUIRE(d != nullptr, "conv desc is NULL");
    VDM_REQUIRE(d->n > 0 && d->od > 0 && d->oh > 0 && d->ow > 0, "conv: bad output dims %d %d %d %d", d->n, d->od, d->oh, d->ow);
    VDM_REQUIRE(d->cin > 0 && d->cout > 0, "conv: bad channels cin=%d cout=%d", d->cin, d->cout);
    VDM_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv: ksize must be 1 or 3 (got %d)", d->ksize);
    VDM_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride must be 1 or 2 (got %d)", d->stride);
    VDM_REQUIRE(!(d->stride == 2 && d->ksize != 3), "conv: stride 2 needs ksize 3");
    VDM_REQUIRE(!(d->upsample && (d->stride != 1 || d->ksize != 3)), "conv: upsample needs stride 1, ksize 3");
    VDM_REQUIRE(!(d->upsample && ((d->od | d->oh | d->ow) & 1)), "conv: upsample needs even output dims");
    VDM_REQUIRE(d->dtype == VDM_F32 || d->dtype == VDM_BF16, "conv: bad dtype %d", d->dtype);
    VDM_REQUIRE(d->pad_mode == VDM_PAD_ZEROS || d->pad_mode == VDM_PAD_CIRCULAR, "conv: bad pad_mode %d", d->pad_mode);
    return VDM_OK;
}

struct Plan {           // derived launch parameters of one conv in one direction
    int taps, nc, nchunks, nkb, O, K;
};
static Plan plan_of(const vdm_conv_desc* d, int dgrad) {
    Plan p;
    p.taps = d->ksize * d->ksize * d->ksize;
    p.O = dgrad ? d->cin : d->cout;
    p.K = dgrad ? d->cout : d->cin;
    p.nc = nc_for(p.O, d->dtype);
    p.nchunks = cdiv(p.O, p.nc * 16);
    p.nkb = cdiv(p.K, kb_of(d->dtype));
    return p;
}

static int cu_count();

static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}

// The dynamic-LDS limit of a kernel is a per-device attribute: `done_mask` (one static per kernel instantiation) has one bit per
// device ordinal, so a process that drives several GPUs raises the limit on each of them.
template <typename K>
static int set_lds(K kernel, size_t bytes, unsigned long long& done_mask) {
    const int dev = current_device();
    if (dev < 64 && ((done_mask >> dev) & 1ull)) return VDM_OK;
    int e = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                      "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (e) return e;
    if (dev < 64) done_mask |= 1ull << dev;
    return VDM_OK;
}

template <typename T, typename TO, int KS, int STRIDE, int UPS, int NC, int TZ, int TY, bool SPLIT = false>
static int launch_fwd_cfg(const ConvArgs& a0, hipStream_t s) {
    using G = Geo<KS, STRIDE, TZ, TY>;
    ConvArgs a = a0;
    if (SPLIT) a.nchunks *= 2;
    a.ntz = cdiv(a.Dz, TZ); a.nty = cdiv(a.Dy, TY); a.ntx = cdiv(a.Dx, 16);
    const size_t lds = (size_t)((G::HVOX + 15) / 16) * 1024 + GN_SCRATCH_BYTES;
    auto kern = conv_fwd_kernel<T, TO, KS, STRIDE, UPS, NC, TZ, TY, SPLIT>;
    static unsigned long long lds_done = 0;
    {
        int e = set_lds(kern, lds, lds_done);
        if (e) return e;
    }
    const long long nwg = (long long)a.N * a.ntz * a.nty * a.ntx * a.nchunks;
    if (nwg > 0x7fffffffLL) { set_error("conv: grid too large"); return VDM_ERR_ARG; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, s, a);
    VDM_LAUNCH_CHECK("conv_fwd_kernel");
    return VDM_OK;
}

static int cu_count() {                                   // of the current device (cached per device ordinal)
    static int cached[64] = {0};
    const int dev = current_device();
    int n = dev < 64 ? cached[dev] : 0;
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        if (dev < 64) cached[dev] = n;
    }
    return n;
}

// Spatial tile of the bf16 3x3x3 stride-1 kernel.  Large grids: 4x8x16 (least halo traffic).  Small grids (deep UNet levels, and
// everything at batch 1): smaller tiles so that more workgroups exist - a workgroup runs its K-blocks strictly one after the other
// (stage, barrier, 27 taps), so a deep-level conv is bound by that serial chain unless co-resident workgroups overlap it:
// 2x8x16 (46 KB LDS, 3 per CU) below 2 workgroups per CU, 1x8x16 / 1x4x16 when even those leave CUs empty.
static void small_grid_tile(const ConvArgs& a, int& tz, int& ty) {
    const long long per_sample = (long long)a.nchunks * a.N * cdiv(a.Dx, 16);
    const long long tiles48 = per_sample * cdiv(a.Dz, 4) * cdiv(a.Dy, 8), tiles28 = per_sample * cdiv(a.Dz, 2) * cdiv(a.Dy, 8);
    tz = 4; ty = 8;
    if (tiles48 >= 2LL * cu_count()) return;
    tz = 2;
    if (tiles28 >= cu_count()) return;
    tz = 1;
    if (2 * tiles28 < cu_count()) ty = 4;
}

// NC=4 conv with fewer than two workgroups per CU even on the small tiles: half-chunk workgroups (the NC=2 kernel on the same packed
// weights; bf16 3x3x3 stride 1, not the up-sampling conv)
static bool uses_split(const ConvArgs& a, int tz, int ty) {
    const long long wgs = (long long)a.nchunks * a.N * cdiv(a.Dz, tz) * cdiv(a.Dy, ty) * cdiv(a.Dx, 16);
    return tz <= 2 && wgs < 2LL * cu_count() && a.Cout % 64 == 0 && getenv("VDM4CDM_NO_SPLIT") == nullptr;
}

template <typename T, typename TO, int KS, int STRIDE, int UPS, int NC>
static int launch_fwd_geo(const ConvArgs& a, hipStream_t s) {
    if constexpr (STRIDE == 2)
        return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 2, 4>(a, s);
    else if constexpr (KS == 3 && sizeof(T) == 2) {
        int tz, ty;
        small_grid_tile(a, tz, ty);
        if constexpr (NC == 4 && UPS == 0 && sizeof(TO) == 2) {
            if (uses_split(a, tz, ty)) {
                if (tz == 1) return ty == 4 ? launch_fwd_cfg<T, TO, KS, STRIDE, UPS, 2, 1, 4, true>(a, s) : launch_fwd_cfg<T, TO, KS, STRIDE, UPS, 2, 1, 8, true>(a, s);
                return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, 2, 2, 8, true>(a, s);
            }
        }
        if (tz == 1) return ty == 4 ? launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 1, 4>(a, s) : launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 1, 8>(a, s);
        if (tz == 2) return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 2, 8>(a, s);
        return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 4, 8>(a, s);
    } else
        return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 4, 8>(a, s);
}

template <typename T, typename TO, int KS, int STRIDE, int UPS>
static int launch_fwd_nc(const ConvArgs& a, int nc, hipStream_t s) {
    switch (nc) {
        case 1: return launch_fwd_geo<T, TO, KS, STRIDE, UPS, 1>(a, s);
        case 2: return launch_fwd_geo<T, TO, KS, STRIDE, UPS, 2>(a, s);
        default: return launch_fwd_geo<T, TO, KS, STRIDE, UPS, 4>(a, s);
    }
}

template <typename T, typename TO>
static int launch_fwd_variant(const ConvArgs& a, int ks, int stride, int ups, int nc, hipStream_t s) {
    if (ks == 1) return launch_fwd_nc<T, TO, 1, 1, 0>(a, nc, s);
    if (stride == 2) return launch_fwd_nc<T, TO, 3, 2, 0>(a, nc, s);
    if (ups) return launch_fwd_nc<T, TO, 3, 1, 1>(a, nc, s);
    return launch_fwd_nc<T, TO, 3, 1, 0>(a, nc, s);
}

// tap-packed kernel: bf16 3x3x3 stride-1 convs whose reduction has <= 8 channels (one 16-byte piece per voxel)
static bool uses_kpack(int dtype, int ks, int stride, int ups, int K, int O, int out_f32) {
    return dtype == VDM_BF16 && ks == 3 && stride == 1 && !ups && K <= 8 && O <= 32 && !out_f32 && getenv("VDM4CDM_NO_KPACK") == nullptr;
}
static bool uses_kpack(const vdm_conv_desc* d, int dgrad) {
    return uses_kpack(d->dtype, d->ksize, d->stride, d->upsample, dgrad ? d->cout : d->cin, dgrad ? d->cin : d->cout, dgrad ? 0 : d->out_f32);
}

template <int NC>
static int launch_kpack(const ConvArgs& a0, hipStream_t s) {
    using G = Geo<3, 1, 4, 8>;
    ConvArgs a = a0;
    a.ntz = cdiv(a.Dz, G::TZ); a.nty = cdiv(a.Dy, G::TY); a.ntx = cdiv(a.Dx, 16);
    const size_t lds = (size_t)((G::HVOX + 63) / 64) * 1024 + GN_SCRATCH_BYTES;
    const long long nwg = (long long)a.N * a.ntz * a.nty * a.ntx * a.nchunks;
    if (nwg > 0x7fffffffLL) { set_error("conv: grid too large"); return VDM_ERR_ARG; }
    hipLaunchKernelGGL((conv_kpack_kernel<bf16_t, NC>), dim3((unsigned)nwg), dim3(256), lds, s, a);
    VDM_LAUNCH_CHECK("conv_kpack_kernel");
    return VDM_OK;
}

static int launch_fwd(const ConvArgs& a, int dtype, int out_f32, int ks, int stride, int ups, int nc, hipStream_t s) {
    if (uses_kpack(dtype, ks, stride, ups, a.Cin, a.Cout, out_f32)) return nc == 1 ? launch_kpack<1>(a, s) : launch_kpack<2>(a, s);
    if (dtype == VDM_F32) return launch_fwd_variant<float, float>(a, ks, stride, ups, nc, s);
    if (out_f32) {
        if (!(ks == 3 && stride == 1 && !ups && nc == 1)) {
            set_error("conv: out_f32 with bf16 input is only built for ksize 3, stride 1, cout <= 16");
            return VDM_ERR_UNSUPPORTED;
        }
        return launch_fwd_geo<bf16_t, float, 3, 1, 0, 1>(a, s);
    }
    return launch_fwd_variant<bf16_t, bf16_t>(a, ks, stride, ups, nc, s);
}

template <typename T, int KS, int STRIDE, int UPS, int TZ, int TY, int NTA = WG<T>::NT, int NTB = WG<T>::NT>
static int launch_wgrad_cfg(WgradArgs w, float* dw, float* dbias, int accumulate, int cout, int cin, size_t ws_bytes, hipStream_t s) {
    using G = Geo<KS, STRIDE, TZ, TY>;
    constexpr int CL = WG<T>::NT * 16;
    ConvArgs& a = w.c;
    a.ntz = cdiv(a.Dz, TZ); a.nty = cdiv(a.Dy, TY); a.ntx = cdiv(a.Dx, 16);
    w.ntiles = a.N * a.ntz * a.nty * a.ntx;
    const int npairs = w.ncb * w.nkb;
    int P = 512 / npairs;                     // persistent: ~2 workgroups per CU over all (cout, cin) block pairs
    if (P < 1) P = 1;
    if (P > w.ntiles) P = w.ntiles;
    w.P = P;
    const int per_wg = (G::TAPS > 1) ? 1 : 4;
    const size_t slab_bytes = (size_t)npairs * P * per_wg * G::TAPS * CL * CL * sizeof(float);
    const size_t need = slab_bytes + (size_t)w.ncb * P * CL * sizeof(float);
    if (need > ws_bytes) { set_error("conv_wgrad: workspace too small (%zu < %zu)", ws_bytes, need); return VDM_ERR_ARG; }
    if (dbias != nullptr && G::TAPS == 1) { set_error("conv_wgrad: fused bias gradient is only built for ksize 3"); return VDM_ERR_UNSUPPORTED; }
    w.bslabs = dbias ? reinterpret_cast<float*>(reinterpret_cast<char*>(w.slabs) + slab_bytes) : nullptr;
    const size_t lds = (size_t)((G::HVOX + 15) / 16) * 1024 + (size_t)G::OVOX * 64;
    auto kern = conv_wgrad_kernel<T, KS, STRIDE, UPS, TZ, TY, NTA, NTB>;
    static unsigned long long lds_done = 0;
    {
        int e = set_lds(kern, lds, lds_done);
        if (e) return e;
    }
    hipLaunchKernelGGL(kern, dim3(npairs * P), dim3(256), lds, s, w);
    VDM_LAUNCH_CHECK("conv_wgrad_kernel");
    const int total = G::TAPS * cout * cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, WRED_OUT)), dim3(256), 0, s,
                       (const float*)w.slabs, dw, G::TAPS, cout, cin, w.ncb, w.nkb, CL, P * per_wg, accumulate);
    VDM_LAUNCH_CHECK("wgrad_reduce_kernel");
    if (dbias) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3(cdiv(cout, 16)), dim3(256), 0, s, (const float*)w.bslabs, dbias, cout, CL, P, accumulate);
        VDM_LAUNCH_CHECK("wgrad_bias_reduce_kernel");
    }
    return VDM_OK;
}

// up-sampling conv: 8 parity classes x 8 merged taps on the coarse grid
template <typename T, int TZ, int TY, int WGS>
static int launch_wgrad_cls(WgradArgs w, float* dw, float* dbias, int accumulate, int cout, int cin, size_t ws_bytes, hipStream_t s) {
    using G = Geo<3, 1, TZ, TY>;
    constexpr int CL = WG<T>::NT * 16;
    ConvArgs& a = w.c;
    a.Dz /= 2; a.Dy /= 2; a.Dx /= 2;                       // everything runs on the coarse grid
    a.Iz = a.Sz = a.Dz; a.Iy = a.Sy = a.Dy; a.Ix = a.Sx = a.Dx;
    a.ntz = cdiv(a.Dz, G::TZ); a.nty = cdiv(a.Dy, G::TY); a.ntx = cdiv(a.Dx, 16);
    w.ntiles = a.N * a.ntz * a.nty * a.ntx;
    const int npairs = 8 * w.ncb * w.nkb;
    int P = WGS / npairs;
    if (P < 1) P = 1;
    if (P > w.ntiles) P = w.ntiles;
    w.P = P;
    const size_t slab_bytes = (size_t)npairs * P * 8 * CL * CL * sizeof(float);
    const size_t need = slab_bytes + (size_t)w.ncb * 8 * P * CL * sizeof(float);
    if (need > ws_bytes) { set_error("conv_wgrad: workspace too small (%zu < %zu)", ws_bytes, need); return VDM_ERR_ARG; }
    w.bslabs = dbias ? reinterpret_cast<float*>(reinterpret_cast<char*>(w.slabs) + slab_bytes) : nullptr;
    const size_t lds = (size_t)((G::HVOX + 15) / 16) * 1024 + (size_t)G::OVOX * 64;
    auto kern = conv_wgrad_kernel<T, 3, 1, 2, TZ, TY>;
    static unsigned long long lds_done = 0;
    {
        int e = set_lds(kern, lds, lds_done);
        if (e) return e;
    }
    hipLaunchKernelGGL(kern, dim3(npairs * P), dim3(256), lds, s, w);
    VDM_LAUNCH_CHECK("conv_wgrad_kernel(class)");
    hipLaunchKernelGGL(wgrad_cls_reduce_kernel, dim3(cdiv(27 * cout * cin, 64)), dim3(256), 0, s, (const float*)w.slabs, dw, cout, cin, w.ncb,
                       w.nkb, CL, P, accumulate);
    VDM_LAUNCH_CHECK("wgrad_cls_reduce_kernel");
    if (dbias) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3(cdiv(cout, 16)), dim3(256), 0, s, (const float*)w.bslabs, dbias, cout, CL, 8 * P, accumulate);
        VDM_LAUNCH_CHECK("wgrad_bias_reduce_kernel");
    }
    return VDM_OK;
}

template <typename T>
static int launch_wgrad(const WgradArgs& w, float* dw, float* db, int acc, int cout, int cin, int ks, int stride, int ups, size_t ws,
                        hipStream_t s) {
    if (ks == 1) return launch_wgrad_cfg<T, 1, 1, 0, 4, 8>(w, dw, db, acc, cout, cin, ws, s);
    if (stride == 2) return launch_wgrad_cfg<T, 3, 2, 0, 2, 4>(w, dw, db, acc, cout, cin, ws, s);
    if (ups) return launch_wgrad_cls<T, 2, 8, 512>(w, dw, db, acc, cout, cin, ws, s);      // (2x4x16 tiles with 1024 workgroups: same time)
    if constexpr (sizeof(T) == 2) {                          // 64-byte blocks with a single real 16-channel tile
        if (cin <= 16) return launch_wgrad_cfg<T, 3, 1, 0, 2, 8, 2, 1>(w, dw, db, acc, cout, cin, ws, s);
        if (cout <= 16) return launch_wgrad_cfg<T, 3, 1, 0, 2, 8, 1, 2>(w, dw, db, acc, cout, cin, ws, s);
    }
    return launch_wgrad_cfg<T, 3, 1, 0, 2, 8>(w, dw, db, acc, cout, cin, ws, s);
}

// ---- class-conv tables -----------------------------------------------------------------------
enum ClsKind { CLS_UP_FWD = 0, CLS_UP_DGRAD = 1, CLS_S2_DGRAD = 2 };

// per-dimension entries of parity p: (coarse offset o, set of master taps merged into the entry)
static int cls_dim_entries(int kind, int p, int o[2], unsigned tapset[2]) {
    if (kind == CLS_S2_DGRAD) {
        if (p == 0) { o[0] = 0; tapset[0] = 1u << 1; return 1; }
        o[0] = +1; tapset[0] = 1u << 0; o[1] = 0; tapset[1] = 1u << 2; return 2;
    }
    if (p == 0) { o[0] = -1; tapset[0] = 1u << 0; o[1] = 0; tapset[1] = (1u << 1) | (1u << 2); return 2; }
    o[0] = 0; tapset[0] = (1u << 0) | (1u << 1); o[1] = +1; tapset[1] = 1u << 2; return 2;
}

static void build_cls(int kind, ClsTable& tab, ClsMasks& masks) {
    using G = Geo<3, 1, 4, 8>;
    for (int i = 0; i < 64; ++i) masks.m[i] = 0;
    for (int cl = 0; cl < 8; ++cl) {
        const int p[3] = {(cl >> 2) & 1, (cl >> 1) & 1, cl & 1};
        int o[3][2];
        unsigned ts[3][2];
        int cnt[3];
        for (int d = 0; d < 3; ++d) cnt[d] = cls_dim_entries(kind, p[d], o[d], ts[d]);
        int ne = 0;
        for (int iz = 0; iz < cnt[0]; ++iz)
            for (int iy = 0; iy < cnt[1]; ++iy)
                for (int ix = 0; ix < cnt[2]; ++ix) {
                    // halo coordinate read by this entry: forward-type kernels read c + o (halo origin -1 -> o + 1);
                    // the up-conv input gradient reads the class sub-grid at c - o (-> 1 - o)
                    const int hz = kind == CLS_UP_DGRAD ? 1 - o[0][iz] : o[0][iz] + 1;
                    const int hy = kind == CLS_UP_DGRAD ? 1 - o[1][iy] : o[1][iy] + 1;
                    const int hx = kind == CLS_UP_DGRAD ? 1 - o[2][ix] : o[2][ix] + 1;
                    tab.e[cl][ne].lds_off = (hz * G::HY + hy) * G::HX * 64;
                    tab.e[cl][ne].dx = hx;
                    unsigned m = 0;
                    for (int tz = 0; tz < 3; ++tz)
                        for (int ty = 0; ty < 3; ++ty)
                            for (int tx = 0; tx < 3; ++tx)
                                if (((ts[0][iz] >> tz) & 1) && ((ts[1][iy] >> ty) & 1) && ((ts[2][ix] >> tx) & 1)) m |= 1u << ((tz * 3 + ty) * 3 + tx);
                    masks.m[cl * 8 + ne] = m;
                    ++ne;
                }
        tab.n[cl] = ne;
        for (int i = ne; i < 8; ++i) tab.e[cl][i] = tab.e[cl][0];
    }
}

static bool uses_cls(const vdm_conv_desc* d, int dgrad) { return d->ksize == 3 && (d->upsample || (dgrad && d->stride == 2)); }
static int cls_kind(const vdm_conv_desc* d, int dgrad) { return d->upsample ? (dgrad ? CLS_UP_DGRAD : CLS_UP_FWD) : CLS_S2_DGRAD; }

template <typename T, int NC, int MODE>
static int launch_cls_cfg(const ClsArgs& ca0, hipStream_t s) {
    using G = Geo<3, 1, 4, 8>;
    ClsArgs ca = ca0;
    ConvArgs& a = ca.c;
    a.ntz = cdiv(a.Dz, G::TZ); a.nty = cdiv(a.Dy, G::TY); a.ntx = cdiv(a.Dx, 16);
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
static int run_cls(const vdm_conv_desc* d, int kind, const void* x, const void* w, const float* bias, const void* res, void* out,
                   int cd, int ch, int cw, hipStream_t s, float* gn_partials = nullptr) {
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

static void fill_dims(ConvArgs& a, const vdm_conv_desc* d) {
    a.N = d->n; a.Dz = d->od; a.Dy = d->oh; a.Dx = d->ow;
    a.Iz = d->od * d->stride; a.Iy = d->oh * d->stride; a.Ix = d->ow * d->stride;
    a.Sz = d->upsample ? a.Iz / 2 : a.Iz; a.Sy = d->upsample ? a.Iy / 2 : a.Iy; a.Sx = d->upsample ? a.Ix / 2 : a.Ix;
    a.circular = d->pad_mode == VDM_PAD_CIRCULAR;
}

// spatial tile (TZ, TY; TX = 16) that vdm_conv_fwd will use for this conv - mirrors launch_fwd / launch_fwd_geo
static void fwd_tile_shape(const ConvArgs& a, int dtype, int out_f32, int ks, int stride, int ups, int& tz, int& ty) {
    tz = 4; ty = 8;
    if (stride == 2) { tz = 2; ty = 4; return; }
    if (uses_kpack(dtype, ks, stride, ups, a.Cin, a.Cout, out_f32)) return;
    if (ks == 3 && dtype == VDM_BF16) small_grid_tile(a, tz, ty);
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
            st_elem<T>(out + f + e, pack_value<T>(it, w, r, e / EPL, j));
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

extern "C" int vdm_conv_kernel_variant(const vdm_conv_desc* d, int dgrad) {
    if (validate(d)) return -1;
    if (uses_cls(d, dgrad)) return VDM_CONV_VARIANT_CLASS;
    if (uses_kpack(d, dgrad)) return VDM_CONV_VARIANT_KPACK;
    if (d->dtype == VDM_BF16 && d->ksize == 3 && !d->upsample && (dgrad || (d->stride == 1 && !d->out_f32))) {
        const Plan p = plan_of(d, dgrad);
        if (p.nc == 4) {
            ConvArgs a{};
            a.N = d->n; a.Dz = d->od; a.Dy = d->oh; a.Dx = d->ow;         // (dgrad of a stride-1 conv runs on the output grid too)
            a.Cout = p.O; a.nchunks = p.nchunks;
            int tz, ty;
            small_grid_tile(a, tz, ty);
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
    int P = 512 / npairs;
    if (P < 1) P = 1;
    const int slots = d->upsample ? 8 : (taps > 1 ? 1 : 4) * taps;
    return (size_t)npairs * P * slots * CL * CL * sizeof(float) + (size_t)cdiv(d->cout, CL) * cls * P * CL * sizeof(float);
}

extern "C" int vdm_conv_wgrad(const vdm_conv_desc* d, const void* x, const void* dout, float* dw, float* dbias, int accumulate,
                              void* workspace, size_t workspace_bytes, void* stream) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(x && dout && dw && workspace, "conv_wgrad: NULL pointer");
    const int CL = d->dtype == VDM_F32 ? 16 : 32;
    WgradArgs w{};
    fill_dims(w.c, d);
    w.c.x = x;
    w.c.Cin = d->cin; w.c.CinStride = cpad(d->cin, d->dtype); w.c.Cout = d->cout;
    w.dout = dout; w.dout_stride = cpad(d->cout, d->dtype);
    w.slabs = (float*)workspace;
    w.ncb = cdiv(d->cout, CL); w.nkb = cdiv(d->cin, CL);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == VDM_F32)
        return launch_wgrad<float>(w, dw, dbias, accumulate, d->cout, d->cin, d->ksize, d->stride, d->upsample, workspace_bytes, s);
    return launch_wgrad<bf16_t>(w, dw, dbias, accumulate, d->cout, d->cin, d->ksize, d->stride, d->upsample, workspace_bytes, s);
}
