// conv_common.h - shared pieces of the convolution kernels (conv_fwd.hip, conv_cls.hip, conv_wgrad.hip, conv_api.hip).
//
// 3D convolution as implicit GEMM on the gfx950 matrix cores (K1/K3/K4/K5 of DESIGN.md).
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
// Translation units: conv_fwd.hip (generic fwd / dgrad kernel incl. half-chunk workgroups, tap-packed kernel for <= 8 reduction
//   channels), conv_cls.hip (per-parity-class convs: up-sampling conv fwd / dgrad, stride-2 dgrad), conv_wgrad.hip (weight
//   gradients + slab reduce kernels), conv_api.hip (weight packers and the C-ABI entry points).
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace vdm {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static constexpr int GNP_TABLE_BYTES = 512 * 2 * 4;          // GroupNorm prologue: (A, B) of up to 512 input channels
static constexpr int GN_SCRATCH_BYTES = 8 * 64 * 2 * 4;      // <= 8 waves x (NC <= 4) * 16 channels x (sum, sumsq) floats

// ---------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------
template <int KS_, int STRIDE_, int TZ_, int TY_, int NW_ = 4>
struct Geo {
    static constexpr int KS = KS_, STRIDE = STRIDE_, TZ = TZ_, TY = TY_, TX = 16;
    static constexpr int NW = NW_;                       // waves per workgroup (the rows of a tile are split over them)
    static constexpr int PAD = KS / 2;
    static constexpr int TAPS = KS * KS * KS;
    static constexpr int HZ = (TZ - 1) * STRIDE + KS, HY = (TY - 1) * STRIDE + KS, HX = (TX - 1) * STRIDE + KS;
    static constexpr int HVOX = HZ * HY * HX;
    static constexpr int ROWS = TZ * TY;                 // 16-voxel MFMA columns-tiles per workgroup
    static constexpr int NV = ROWS / NW;                 // per wave
    static constexpr int OVOX = ROWS * 16;
    static_assert(ROWS % NW == 0, "rows must split over the waves");
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
    int nseg, zsteps;                // rolling-z kernel: segments per tile column, z steps (tiles) per segment
    FastDiv fdx, fdy, fdz, fdn;      // divisions by ntx, nty, ntz, N (set_tile_divs)
    float* gnp;          // optional GroupNorm partials of the output: [N][ntz*nty*ntx][Cout][2] = (sum, sum of squares) per tile
    // GroupNorm backward folded into a dgrad epilogue (conv_epilogue_gnb): the conv result is dL/dy of y = drop(silu(gn(x)));
    // the epilogue turns it into dyh = dL/dy * keep * silu'(yhat), stores THAT, and reduces per tile and channel
    // (sum dyh, sum dyh*x) into gnp.  x = concat(gx1 [c1 ch], gx2 [c2 ch]) is the GroupNorm input.
    const void* gx1;
    const void* gx2;
    const float* gstats;             // [N][G][2] raw moments of x
    const float* ggamma;
    const float* gbeta;
    const unsigned char* gmask;      // optional dropout keep bits, one byte per 16-byte piece of y: [N][V][C/EPL]
    int gc1, gc2, gG;
    float geps, gcnt, ginv_keep;     // gcnt = voxels * channels per group
#ifdef VDM_TIMELINE
    unsigned long long* stamps;      // diagnostic build only (tools/conv_timeline.py): [workgroup][wave][8] s_memrealtime stamps + HW ids
#endif
};

#ifdef VDM_TIMELINE
#define VDM_STAMP(k)                                                                       \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        tl_t[k] = __builtin_amdgcn_s_memrealtime();                                        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
extern unsigned long long* g_timeline_stamps;            // set by vdm_debug_set_stamps (conv_api.hip)
#else
#define VDM_STAMP(k) do { } while (0)
#endif

static inline void set_tile_divs(ConvArgs& a) {
    a.fdx = make_fastdiv((uint32_t)a.ntx); a.fdy = make_fastdiv((uint32_t)a.nty); a.fdz = make_fastdiv((uint32_t)a.ntz);
    a.fdn = make_fastdiv((uint32_t)a.N);
}
// linear tile id -> (tx, ty, tz, n, rest) with b = (((rest * N + n) * ntz + tz) * nty + ty) * ntx + tx; scalar arithmetic
__device__ __forceinline__ void decode_tile(const ConvArgs& a, uint32_t b, int& tx, int& ty, int& tz, int& n, int& rest) {
    uint32_t q = fdiv(b, a.fdx);
    tx = (int)(b - q * (uint32_t)a.ntx); b = q;
    q = fdiv(b, a.fdy);
    ty = (int)(b - q * (uint32_t)a.nty); b = q;
    q = fdiv(b, a.fdz);
    tz = (int)(b - q * (uint32_t)a.ntz); b = q;
    q = fdiv(b, a.fdn);
    n = (int)(b - q * (uint32_t)a.N);
    rest = (int)q;
}

__device__ __forceinline__ int wrap(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}

// 64 zero bytes: source of every padded / out-of-range piece of the LDS-DMA staging below.
// (one copy per translation unit: no relocatable device code needed)
static __device__ uint4 g_zero_page[16];

// LDS image of the forward kernel: voxel-major, 64 B per halo voxel, the four 16-B pieces of a voxel stored at
// slot = piece ^ ((hx >> 1) & 3), hx = x position inside the halo row.  With this swizzle the MFMA operand read
// (ds_read_b128, 16 x-consecutive voxels x 4 k-chunks) is bank-conflict-free for every row alignment, and the
// (dz, dy, row) shifts stay compile-time ds_read offsets.
// Filled by LDS-DMA (global_load_lds_dwordx4): one wave-instruction = 16 voxels x 64 B = 1 KiB of LDS written
// linearly; the four lanes of a voxel fetch its (permuted) pieces, i.e. whole 64-B segments of the NDHWC row.
//
// The address arithmetic of the ~17 chunks a wave stages is on the critical path of every tile (tools/conv_timeline.py: a
// workgroup used to spend 5 of its 17 us between kernel entry and the last DMA issue): the halo coordinate (hz, hy, hx) of a
// lane advances by a CONSTANT from chunk to chunk (64 halo voxels), so it is carried incrementally (adds + two conditional
// carries, no division), and the voxel offset inside the sample is 32-bit with 24-bit multiplies (full-rate v_mul_u32_u24;
// 32-bit integer multiplies are quarter rate).  The host checks that a sample's elements fit 32 bits.
// ss / so*: logical voxel i maps to source voxel ss * i + so (class sub-grid of the per-parity-class kernels when ss == 2);
// sD*: source tensor dims.
// NWAVES: waves that share the chunks (4: all waves of a workgroup; 1: a wave stages a private image - conv_ksplit_kernel).
template <typename T, typename G, int UPS, int NWAVES = 4>
__device__ __forceinline__ void stage_halo_dma_chunks(char* lds, const T* __restrict__ x, const ConvArgs& a, int n, int oz0, int oy0,
                                                      int ox0, int kb, int wave, int lane, int ss, int soz, int soy, int sox, int sDz,
                                                      int sDy, int sDx) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NCHUNK = (G::HVOX + 15) / 16;
    constexpr int STEP = 16 * NWAVES;                       // halo voxels between two chunks of one wave
    constexpr int DX = STEP % G::HX, DY = (STEP / G::HX) % G::HY, DZ = STEP / (G::HX * G::HY);
    const int iz0 = oz0 * G::STRIDE - G::PAD, iy0 = oy0 * G::STRIDE - G::PAD, ix0 = ox0 * G::STRIDE - G::PAD;
    const int k = lane >> 2, j = lane & 3;
    const int hv0 = wave * 16 + k;                          // first chunk of this wave (constant divisors: once per call)
    int hx = hv0 % G::HX, hy = (hv0 / G::HX) % G::HY, hz = hv0 / (G::HX * G::HY);
    const T* xn = x + (size_t)n * ((size_t)sDz * sDy * sDx * a.CinStride);
    const bool fastwrap = a.Iz >= G::HZ && a.Iy >= G::HY && a.Ix >= G::HX;      // one conditional add / subtract wraps
    for (int c = wave; c < NCHUNK; c += NWAVES) {
        const int pc = j ^ ((hx >> 1) & 3);
        const int ci = kb * KB + pc * EPL;
        int iz = iz0 + hz, iy = iy0 + hy, ix = ix0 + hx;
        bool ok = ci < a.Cin && hz < G::HZ;                 // (hz < HZ <=> the chunk's tail is inside the halo)
        if (a.circular) {
            if (fastwrap) {
                iz += iz < 0 ? a.Iz : 0; iz -= iz >= a.Iz ? a.Iz : 0;
                iy += iy < 0 ? a.Iy : 0; iy -= iy >= a.Iy ? a.Iy : 0;
                ix += ix < 0 ? a.Ix : 0; ix -= ix >= a.Ix ? a.Ix : 0;
            } else {
                iz = wrap(iz, a.Iz); iy = wrap(iy, a.Iy); ix = wrap(ix, a.Ix);
            }
        } else {
            ok = ok && (unsigned)iz < (unsigned)a.Iz && (unsigned)iy < (unsigned)a.Iy && (unsigned)ix < (unsigned)a.Ix;
        }
        if (UPS) { iz >>= 1; iy >>= 1; ix >>= 1; }
        const unsigned sz = (unsigned)(ss * iz + soz) & 0xffffffu, sy = (unsigned)(ss * iy + soy) & 0xffffffu,
                       sx = (unsigned)(ss * ix + sox) & 0xffffffu;      // (masked: garbage of !ok lanes stays a legal u24 operand)
        const unsigned row = __umul24(sz, (unsigned)sDy) + sy;
        const unsigned vox = __umul24(row, (unsigned)sDx) + sx;
        const unsigned eoff = vox * (unsigned)a.CinStride + (unsigned)ci;
        const void* src = ok ? static_cast<const void*>(xn + eoff) : static_cast<const void*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + c * 1024), 16, 0, 0);
        hx += DX; hy += DY; hz += DZ;
        if (hx >= G::HX) { hx -= G::HX; hy += 1; }
        if (hy >= G::HY) { hy -= G::HY; hz += 1; }
    }
}


// ---------------------------------------------------------------------------------------------
// Row-wise staging (round 4).  Counters (profiles/r04_pmc_mfma_util.json): the level-0 conv issued 2 070 vector instructions per wave
// and tile next to its 432 MFMAs, and about half of them were the address arithmetic of the chunk walk above - it re-derives
// (hz, hy, hx), the wrap / bounds tests, two 24-bit multiplies and a 64-bit pointer select per lane for every one of a wave's 17
// chunks, because a 16-voxel chunk of the linear halo image straddles two halo rows.  A SIMD issues one vector instruction per 4
// cycles from a wave and an MFMA holds the issue port for 8 of its 16: the staging arithmetic of one workgroup costs the co-resident
// workgroup's tap loop about as many issue slots as the taps themselves (why "fewer staged chunks" bought 8-11 % and more waves nothing).
// Same LDS image, another walk: a halo ROW (hz, hy) is wave-uniform, so its validity, its wrap and its source address are SCALAR
// arithmetic (the scalar unit is otherwise idle); per lane only the x offset of its voxel / piece remains, and that is the same for
// every row - computed once per tile.  Per row: NSEG LDS-DMAs of 16 voxels (16 B per lane, hx = 16 s ... 16 s + 15) and, for the
// HX - 16 NSEG leftover voxels at the end of the row (hx = 16, 17 of an 18-voxel row), ONE 4-byte-per-lane LDS-DMA with 16 lanes per
// voxel active (the leftover voxels of a row are contiguous in the image; masked lanes write nothing).
// ---------------------------------------------------------------------------------------------
#ifndef VDM_ROWSTAGE
#define VDM_ROWSTAGE 1
#endif
// leftover voxels of a row: 0 = one 4-byte-per-lane LDS-DMA (16 lanes per voxel), 1 = one 16-byte-per-lane LDS-DMA (4 lanes per voxel)
#ifndef VDM_TAIL16
#define VDM_TAIL16 0
#endif
// per-lane x part (once per tile / per persistent workgroup) + one halo row per call
template <typename T, typename G, int UPS>
struct RowStager {
    static constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB, SH = DT<T>::SHIFT;
    static constexpr int HX = G::HX, HY = G::HY, HZ = G::HZ, PAD = G::PAD;
    static constexpr int NSEG = HX / 16, NTAIL = HX - 16 * NSEG;
    static_assert(NTAIL <= 4, "the leftover voxels of a row go out as one 256-byte LDS-DMA");
    unsigned xoff[NSEG], toff;
    bool okx[NSEG], tok;
    int lane;
    const char* xn;
    int iz0, iy0, ss, soz, soy, sDy, sDx;
    // byte offset of the lane's 16 B (segments) / 4 B (tail) inside a source row, or "take zeros"
    __device__ __forceinline__ RowStager(const T* __restrict__ x, const ConvArgs& a, int n, int oz0, int oy0, int ox0, int kb, int lane_, int ss_,
                                         int soz_, int soy_, int sox, int sDz, int sDy_, int sDx_)
        : lane(lane_), ss(ss_), soz(soz_), soy(soy_), sDy(sDy_), sDx(sDx_) {
        iz0 = oz0 * G::STRIDE - PAD; iy0 = oy0 * G::STRIDE - PAD;
        const int ix0 = ox0 * G::STRIDE - PAD;
        xn = reinterpret_cast<const char*>(x + (size_t)n * ((size_t)sDz * sDy * sDx * a.CinStride));
        auto xpart = [&](int hx, int slot, unsigned& off) -> bool {
            const int pc = slot ^ ((hx >> 1) & 3);                            // source piece of this LDS slot (x-swizzle)
            const int ci = kb * KB + pc * EPL;
            int ix = ix0 + hx;
            bool ok = ci < a.Cin;
            if (a.circular) ix = wrap(ix, a.Ix);
            else ok = ok && (unsigned)ix < (unsigned)a.Ix;
            if (UPS) ix >>= 1;
            const unsigned sx = (unsigned)(ss * ix + sox) & 0xffffffu;
            off = (__umul24(sx, (unsigned)a.CinStride) + (unsigned)ci) << SH;
            return ok;
        };
        toff = 0; tok = false;
#pragma unroll
        for (int sgm = 0; sgm < NSEG; ++sgm) okx[sgm] = xpart(16 * sgm + (lane >> 2), lane & 3, xoff[sgm]);
        if constexpr (NTAIL > 0) {
#if VDM_TAIL16
            tok = xpart(16 * NSEG + ((lane >> 2) & 3), lane & 3, toff) && lane < 4 * NTAIL;      // 16 B per lane, 4 lanes per leftover voxel
#else
            const int d = lane & 15;                                          // dword of the tail voxel lane >> 4
            tok = xpart(16 * NSEG + (lane >> 4), d >> 2, toff) && lane < 16 * NTAIL;
            toff += (unsigned)(d & 3) * 4u;
#endif
        }
    }
    // halo row (hz, hy) of the tile (wave-uniform) -> LDS row `lrow` (HX * 64 bytes): NSEG (+1) LDS-DMAs, scalar address arithmetic
    __device__ __forceinline__ void row(char* lrow, const ConvArgs& a, int hz, int hy) const {
        int iz = iz0 + hz, iy = iy0 + hy;
        bool okrow = true;
        if (a.circular) {
            iz = wrap(iz, a.Iz); iy = wrap(iy, a.Iy);
        } else {
            okrow = (unsigned)iz < (unsigned)a.Iz && (unsigned)iy < (unsigned)a.Iy;
        }
        if (UPS) { iz >>= 1; iy >>= 1; }
        const unsigned sz = okrow ? (unsigned)(ss * iz + soz) : 0u, sy = okrow ? (unsigned)(ss * iy + soy) : 0u;
        const size_t rowel = (size_t)((sz * (unsigned)sDy + sy) * (unsigned)sDx) * (unsigned)a.CinStride;      // (a sample's elements fit 32 bits: host check)
        const char* rowp = xn + (rowel << SH);
        const char* zp = reinterpret_cast<const char*>(g_zero_page);
#pragma unroll
        for (int sgm = 0; sgm < NSEG; ++sgm) {
            const char* src = (okrow && okx[sgm]) ? rowp + xoff[sgm] : zp;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lrow + sgm * 1024), 16, 0, 0);
        }
#ifndef VDM_EXP_NOTAIL                                                        // (timing experiment: rows without their leftover voxels)
        if constexpr (NTAIL > 0) {
#if VDM_TAIL16
            if (lane < 4 * NTAIL) {
                const char* src = (okrow && tok) ? rowp + toff : zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lrow + NSEG * 1024), 16, 0, 0);
            }
#else
            if (lane < 16 * NTAIL) {
                const char* src = (okrow && tok) ? rowp + toff : zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lrow + NSEG * 1024), 4, 0, 0);
            }
#endif
        }
#endif
    }
};

template <typename T, typename G, int UPS, int NWAVES = 4>
__device__ __forceinline__ void stage_halo_dma_rows(char* lds, const T* __restrict__ x, const ConvArgs& a, int n, int oz0, int oy0,
                                                    int ox0, int kb, int wave, int lane, int ss, int soz, int soy, int sox, int sDz,
                                                    int sDy, int sDx) {
    const RowStager<T, G, UPS> st(x, a, n, oz0, oy0, ox0, kb, lane, ss, soz, soy, sox, sDz, sDy, sDx);
    constexpr int HY = G::HY, NROW = G::HZ * G::HY;
    for (int r = wave; r < NROW; r += NWAVES) st.row(lds + r * (G::HX * 64), a, r / HY, r % HY);
}

template <typename T, typename G, int UPS, int NWAVES = 4>
__device__ __forceinline__ void stage_halo_dma_gen(char* lds, const T* __restrict__ x, const ConvArgs& a, int n, int oz0, int oy0,
                                                   int ox0, int kb, int wave, int lane, int ss, int soz, int soy, int sox, int sDz,
                                                   int sDy, int sDx) {
#if VDM_ROWSTAGE
    stage_halo_dma_rows<T, G, UPS, NWAVES>(lds, x, a, n, oz0, oy0, ox0, kb, wave, lane, ss, soz, soy, sox, sDz, sDy, sDx);
#else
    stage_halo_dma_chunks<T, G, UPS, NWAVES>(lds, x, a, n, oz0, oy0, ox0, kb, wave, lane, ss, soz, soy, sox, sDz, sDy, sDx);
#endif
}

// GroupNorm + SiLU applied to the staged halo image in place (inference: the conv input is silu(gn(x)) of a tensor nobody else
// needs activated - the separate gn_silu_fwd pass, one read + one write of the tensor, disappears).  Every wave transforms exactly
// the chunks it staged itself (same chunk / lane walk as stage_halo_dma_gen), so its own `s_waitcnt vmcnt(0)` is all the
// synchronisation needed before the workgroup barrier in front of the taps.  Voxels outside the volume came from the zero page and
// stay zero (the conv pads the ACTIVATED tensor).  tab: LDS, [0, Cin) = A = rstd * gamma, [Cin, 2 Cin) = B = beta - mean * A of
// this sample.  The arithmetic is gn_silu_fwd's: the conv sees bit-identical operands.
template <typename T, typename G, int NWAVES = 4>
__device__ __forceinline__ void gn_prologue_inplace(char* lds, const float* tab, const ConvArgs& a, int oz0, int oy0, int ox0, int kb,
                                                    int wave, int lane) {
    static_assert(G::STRIDE == 1, "prologue: stride-1 convs only");
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB;
    constexpr int NCHUNK = (G::HVOX + 15) / 16;
    constexpr int STEP = 16 * NWAVES;
    constexpr int DX = STEP % G::HX, DY = (STEP / G::HX) % G::HY, DZ = STEP / (G::HX * G::HY);
    const int iz0 = oz0 - G::PAD, iy0 = oy0 - G::PAD, ix0 = ox0 - G::PAD;
    const int k = lane >> 2, j = lane & 3;
    const int hv0 = wave * 16 + k;
    int hx = hv0 % G::HX, hy = (hv0 / G::HX) % G::HY, hz = hv0 / (G::HX * G::HY);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's LDS-DMA chunks have landed
    for (int c = wave; c < NCHUNK; c += NWAVES) {
        const int pc = j ^ ((hx >> 1) & 3);
        const int ci = kb * KB + pc * EPL;
        const int iz = iz0 + hz, iy = iy0 + hy, ix = ix0 + hx;
        bool ok = ci < a.Cin && hz < G::HZ;
        if (!a.circular) ok = ok && (unsigned)iz < (unsigned)a.Iz && (unsigned)iy < (unsigned)a.Iy && (unsigned)ix < (unsigned)a.Ix;
        if (ok) {
            uint4* q = reinterpret_cast<uint4*>(lds + c * 1024 + lane * 16);
            Piece<T> px;
            px.load(*q);
#pragma unroll
            for (int e = 0; e < EPL; ++e) px.f[e] = silu_f(px.f[e] * tab[ci + e] + tab[a.Cin + ci + e]);
            *q = px.store();
        }
        hx += DX; hy += DY; hz += DZ;
        if (hx >= G::HX) { hx -= G::HX; hy += 1; }
        if (hy >= G::HY) { hy -= G::HY; hz += 1; }
    }
}

// the per-channel affines of sample n for gn_prologue_inplace (all threads of the workgroup; a barrier must follow)
__device__ __forceinline__ void gn_prologue_table(float* tab, const ConvArgs& a, int n, int tid, int nthreads) {
    const int gs = a.Cin / a.gG;
    for (int c = tid; c < a.Cin; c += nthreads) {
        const float sum = a.gstats[((size_t)n * a.gG + c / gs) * 2], sq = a.gstats[((size_t)n * a.gG + c / gs) * 2 + 1];
        const float mean = sum / a.gcnt;
        const float rstd = rsqrtf(fmaxf(sq / a.gcnt - mean * mean, 0.f) + a.geps);
        const float A = rstd * a.ggamma[c];
        tab[c] = A;
        tab[a.Cin + c] = a.gbeta[c] - mean * A;
    }
}

template <typename T, typename G, int UPS>
__device__ __forceinline__ void stage_halo_dma(char* lds, const T* __restrict__ x, const ConvArgs& a, int n,
                                               int oz0, int oy0, int ox0, int kb, int wave, int lane) {
    stage_halo_dma_gen<T, G, UPS, G::NW>(lds, x, a, n, oz0, oy0, ox0, kb, wave, lane, 1, 0, 0, 0, a.Sz, a.Sy, a.Sx);
}

// ---------------------------------------------------------------------------------------------
// MFMA wrappers: acc[16 cout x 16 voxel] += A(16 cout x 64 B of k) * B(64 B of k x 16 voxel)
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<bf16_t>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
#if VDM_FP32_SPLIT
// both operands in split form (hi01, hi23, lo01, lo23): three bf16 MFMAs, small terms first
__device__ __forceinline__ void mma16_ss(f32x4& acc, const uint4& a, const uint4& b) {
    const s16x4 ah = __builtin_bit_cast(s16x4, make_uint2(a.x, a.y)), al = __builtin_bit_cast(s16x4, make_uint2(a.z, a.w));
    const s16x4 bh = __builtin_bit_cast(s16x4, make_uint2(b.x, b.y)), bl = __builtin_bit_cast(s16x4, make_uint2(b.z, b.w));
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
}
// A = packed weights (already split), B = raw fp32 piece from LDS
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const uint4& a, const uint4& b) { mma16_ss(acc, a, split_frag(b)); }
#else
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
}
#endif
// both operands are raw activation / gradient fragments (weight gradients)
template <typename T> __device__ __forceinline__ void mma16_act(f32x4& acc, const uint4& a, const uint4& b) { mma16<T>(acc, a, b); }
#if VDM_FP32_SPLIT
template <> __device__ __forceinline__ void mma16_act<float>(f32x4& acc, const uint4& a, const uint4& b) { mma16_ss(acc, split_frag(a), split_frag(b)); }
#endif

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

// Weight prefetch depth (taps ahead; ring of WPD + 1 register sets).  The packed weights are L1/L2 hits, but a CU's vector L1
// returns hits IN ORDER behind the HBM misses of every other wave of the CU (tools/tcp_order_probe.hip: a hit takes 156 cycles
// on an idle CU and ~1000-1300 while another wave keeps 16 LDS-DMA pieces in flight - which is what the co-resident workgroup
// does while it stages its halo).  -DVDM_WPD_NC2 / -DVDM_WPD_NC4 override for experiments (make variant).
#ifndef VDM_WPD_NC2
#define VDM_WPD_NC2 2
#endif
#ifndef VDM_WPD_NC4
#define VDM_WPD_NC4 1
#endif
template <int NC> struct WPipe { static constexpr int WPD = (NC <= 2) ? VDM_WPD_NC2 : VDM_WPD_NC4; };

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

// bf16, 3x3x3 stride 1: the same pipeline with the activation rows kept in registers across the three dy taps.
// A wave's NV rows are y-consecutive inside one z-slab, so for a fixed (dz, dx) the taps dy = 0, 1, 2 of row v read halo rows
// v, v+1, v+2: NV + 2 distinct rows instead of 3 * NV operand reads (10 instead of 24 at NV = 8).  The LDS pipe moves 128 B per
// clock per CU, a 64-lane ds_read_b128 takes 8 of them, and at NC = 2 the plain order needs exactly as many LDS clocks (8 reads x
// 4 waves x 8) as MFMA clocks (16 MFMAs x 16) per tap - before the LDS-DMA writes of the co-resident workgroup; this order cuts
// the operand reads 2.4x.  Execution order: group g = (dz, dx), then dy; a row register dies after its dy = 2 use, and the same
// row of the NEXT group is loaded right there (ring of NV + 2 fragments: 40 VGPRs instead of the 64 of two full sets).
#ifndef VDM_ROWREUSE
#define VDM_ROWREUSE 1
#endif
__host__ __device__ constexpr int rr_tap(int e) { return ((e / 3) / 3) * 9 + (e % 3) * 3 + ((e / 3) % 3); }     // execution slot -> tap (dz, dy, dx)

template <int NC, int WPD, int NCW = NC>
__device__ __forceinline__ void rr_prefetch_weights(uint4 (&wf)[WPD + 1][NC], const uint4* wk) {
#pragma unroll
    for (int p = 0; p < WPD; ++p)
#pragma unroll
        for (int c = 0; c < NC; ++c) wf[p][c] = wk[(rr_tap(p) * NCW + c) * 64];
}

// zoff: byte offsets of the three z slices (dz = 0, 1, 2) of the wave's output slab inside the image - the compile-time constants
// {0, HSLAB, 2 HSLAB} of a tile image (taps_rowreuse below), or the wave-uniform ring slots of the rolling-z kernel.
template <typename T, typename G, int NC, int NV, int WPD, int NCW = NC>
__device__ __forceinline__ void taps_rowreuse_z(f32x4 (&acc)[NV][NC], const char* lds, const uint4* wk, uint4 (&wf)[WPD + 1][NC],
                                                const int (&lanex)[3], const int (&zoff)[3]) {
    static_assert(G::KS == 3 && G::STRIDE == 1, "3x3x3 stride-1 geometry");
    constexpr int NR = NV + 2, HROW = G::HX * 64;
    uint4 rows[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) rows[r] = *reinterpret_cast<const uint4*>(lds + zoff[0] + lanex[0] + r * HROW);
#pragma unroll
    for (int e = 0; e < 27; ++e) {
        const int g = e / 3, dy = e % 3;
        if (e + WPD < 27) {
#pragma unroll
            for (int c = 0; c < NC; ++c) wf[(e + WPD) % (WPD + 1)][c] = wk[(rr_tap(e + WPD) * NCW + c) * 64];
        }
        if (dy < 2 || g == 8) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int c = 0; c < NC; ++c) mma16<T>(acc[v][c], wf[e % (WPD + 1)][c], rows[v + dy]);
            __builtin_amdgcn_sched_group_barrier(0x20, NC, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, NV * NC, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {                                           // last use of the rows: refill them for group g + 1 as they die
            const int g1 = g + 1;
            const char* nb = lds + lanex[g1 % 3] + zoff[g1 / 3];
            __builtin_amdgcn_sched_barrier(0);
            rows[0] = *reinterpret_cast<const uint4*>(nb);                     // (rows 0 and 1 died with the dy = 1 tap)
            rows[1] = *reinterpret_cast<const uint4*>(nb + HROW);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int c = 0; c < NC; ++c) mma16<T>(acc[v][c], wf[e % (WPD + 1)][c], rows[v + 2]);
                __builtin_amdgcn_sched_barrier(0);
                rows[v + 2] = *reinterpret_cast<const uint4*>(nb + (v + 2) * HROW);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <typename T, typename G, int NC, int NV, int WPD, int NCW = NC>
__device__ __forceinline__ void taps_rowreuse(f32x4 (&acc)[NV][NC], const char* lds, const uint4* wk, uint4 (&wf)[WPD + 1][NC],
                                              const int (&lanex)[3]) {
    constexpr int HSLAB = G::HY * G::HX * 64;
    const int zoff[3] = {0, HSLAB, 2 * HSLAB};
    taps_rowreuse_z<T, G, NC, NV, WPD, NCW>(acc, lds, wk, wf, lanex, zoff);
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
#if VDM_FP32_SPLIT
        uint4 bs[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) bs[v] = split_frag(af[v]);      // once per (tap, row): reused by the NC output tiles
#pragma unroll
        for (int m = 0; m < 3; ++m)                                   // term-major over the NV x NC independent accumulators
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const s16x4 ah = __builtin_bit_cast(s16x4, make_uint2(wf[c].x, wf[c].y)), al = __builtin_bit_cast(s16x4, make_uint2(wf[c].z, wf[c].w));
                    const s16x4 bh = __builtin_bit_cast(s16x4, make_uint2(bs[v].x, bs[v].y)), bl = __builtin_bit_cast(s16x4, make_uint2(bs[v].z, bs[v].w));
                    acc[v][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(m == 1 ? al : ah, m == 0 ? bl : bh, acc[v][c], 0, 0, 0);
                }
#else
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
#endif
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

// the same for K values at once with the DPP operand folded into the add (v_add_f32_dpp): the builtin form above compiles to a
// v_mov_b32_dpp + v_add_f32 pair per step (128 vector instructions for the 16 sums of an NC = 2 epilogue instead of 64).  One asm
// statement per rotation step over all K values: a value is read by its next step K instructions later (the 2 wait states a DPP read
// needs after a vector write of the same register are covered inside the statement; hipcc pads nothing inside asm).
template <int K>
__device__ __forceinline__ void row16_sum_block(float (&v)[K]) {
    static_assert(K % 4 == 0 && K >= 4, "blocks of four values per statement");
#define VDM_DPP4(ctl)                                                                                                            \
    for (int i = 0; i < K; i += 4)                                                                                               \
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " ctl " row_mask:0xf bank_mask:0xf\n\t"                               \
                     "v_add_f32_dpp %1, %1, %1 " ctl " row_mask:0xf bank_mask:0xf\n\t"                                         \
                     "v_add_f32_dpp %2, %2, %2 " ctl " row_mask:0xf bank_mask:0xf\n\t"                                         \
                     "v_add_f32_dpp %3, %3, %3 " ctl " row_mask:0xf bank_mask:0xf\n\ts_nop 0"                                  \
                     : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]))
#pragma unroll
    VDM_DPP4("row_ror:8");
#pragma unroll
    VDM_DPP4("row_ror:4");
#pragma unroll
    VDM_DPP4("row_ror:2");
#pragma unroll
    VDM_DPP4("row_ror:1");
#undef VDM_DPP4
}

template <int NC, int NW = 4>
__device__ __forceinline__ void gn_partials_reduce(float (&gs)[NC * 4], float (&gq)[NC * 4], float* sm, float* dst /* [Cout][2] of this tile */,
                                                   int cout0, int Cout, int wave, int lane, int qstride = NC * 4) {
    const int lx = lane & 15, q = lane >> 4;
    row16_sum_block<NC * 4>(gs);
    row16_sum_block<NC * 4>(gq);
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
        float tot = (sm[t] + sm[NC * 16 * 2 + t]) + (sm[2 * NC * 16 * 2 + t] + sm[3 * NC * 16 * 2 + t]);
        if constexpr (NW == 8)
            tot += (sm[4 * NC * 16 * 2 + t] + sm[5 * NC * 16 * 2 + t]) + (sm[6 * NC * 16 * 2 + t] + sm[7 * NC * 16 * 2 + t]);
        static_assert(NW == 4 || NW == 8, "4 or 8 waves fold their tile sums");
        if (cout0 + c < Cout) dst[(size_t)(cout0 + c) * 2 + (t & 1)] = tot;
    }
}

// Output voxels of a lane: row v of wave `cwave` is (oz, oyw + v, ox); its linear voxel index inside the sample is
// vox0 + v * Dx (a wave's NV rows share one z-slab: TY % NV == 0), all in 32 bits (no 64-bit multiply chains per row).
template <typename G, int NV>
struct RowMap {
    unsigned vox0;
    int oyw;
    bool zx_ok;
    __device__ __forceinline__ RowMap(const ConvArgs& a, int oz0, int oy0, int ox0, int cwave, int lane) {
        static_assert(G::TY % NV == 0, "a wave's rows must stay inside one z-slab");
        const int r0 = cwave * NV;
        const int oz = oz0 + r0 / G::TY, ox = ox0 + (lane & 15);
        oyw = oy0 + r0 % G::TY;
        zx_ok = oz < a.Dz && ox < a.Dx;
        vox0 = __umul24(__umul24((unsigned)oz, (unsigned)a.Dy) + (unsigned)oyw, (unsigned)a.Dx) + (unsigned)ox;
    }
    __device__ __forceinline__ bool ok(const ConvArgs& a, int v) const { return zx_ok && oyw + v < a.Dy; }
    __device__ __forceinline__ unsigned vox(const ConvArgs& a, int v) const { return vox0 + (unsigned)(v * a.Dx); }
};

// per-lane additive terms of the epilogue (bias + per-sample conditioning bias): loaded BEFORE the tap loop by the kernels, so
// that their latency is not exposed at the start of the epilogue
template <int NC>
__device__ __forceinline__ void load_badd(float (&badd)[NC * 4], const ConvArgs& a, int n, int cbase) {
    // BRANCH-FREE on purpose.  Written as `if (c < Cout) { if (bias) bv += bias[c]; if (nbias) bv += nbias[c]; }` the compiler emitted
    // 2 * NC * 4 loads each behind its own branch and each followed by s_waitcnt vmcnt(0): 16 dependent memory round trips (~2 us)
    // at the top of every workgroup, in front of the staging DMA.  Here every load is unconditional (a missing pointer reads the
    // zero page, a channel past Cout reads the last valid one - its value is never stored), so all of them are in flight at once
    // and the first wait is the epilogue's.
    const float* zp = reinterpret_cast<const float*>(g_zero_page);
    const float* pb = a.bias ? a.bias : zp;
    const float* pn = a.nbias ? a.nbias + (size_t)n * a.nbias_stride : zp;
    const int last = a.Cout - 1;
#pragma unroll
    for (int j = 0; j < NC * 4; ++j) {
        const int c = min(cbase + j, last);
        badd[j] = pb[a.bias ? c : 0] + pn[a.nbias ? c : 0];
    }
}

// epilogue: + bias + per-sample conditioning bias + residual, cast, 16-byte NDHWC stores
template <typename T, typename TO, typename G, int NC, int NV>
__device__ __forceinline__ void conv_epilogue(const f32x4 (&acc)[NV][NC], const ConvArgs& a, const float (&badd)[NC * 4], int n, int oz0,
                                              int oy0, int ox0, int cwave, int lane, float* gn_sm, int tile, int cout0, int qstride) {
    constexpr int EPL = DT<T>::EPL;
    float gs[NC * 4], gq[NC * 4];
#pragma unroll
    for (int j = 0; j < NC * 4; ++j) gs[j] = gq[j] = 0.f;
    const int q = lane >> 4;
    const int cbase = cout0 + q * qstride;              // first of this lane's NC*4 consecutive couts
    const bool vec_ok = (a.Cout % (NC * 4) == 0) && (cbase + NC * 4 <= a.Cout);
    const size_t sample = (size_t)n * ((size_t)a.Dz * a.Dy * a.Dx * a.Cout);
    TO* out = reinterpret_cast<TO*>(a.out) + sample;
    const T* res = a.res ? reinterpret_cast<const T*>(a.res) + sample : nullptr;
    const RowMap<G, NV> rm(a, oz0, oy0, ox0, cwave, lane);
    // Fast path (round 4): interior tile, whole channel chunk, bf16 output - which is every workgroup of the 128^3 network.  The
    // general loop below tests row / lane / channel validity per row and element (the compiler keeps both the vector and the scalar
    // tail path per row: ~900 vector instructions and ~500 scalar branches per wave); here the validity is ONE scalar test, the rows
    // are straight-line code and a row's address is the wave-uniform row stride added to a per-lane base.
    if constexpr (sizeof(TO) == 2 && sizeof(T) == 2 && NC >= 2) {
        const bool fast = oz0 + G::TZ <= a.Dz && oy0 + G::TY <= a.Dy && ox0 + 16 <= a.Dx && a.Cout % (NC * 4) == 0 &&
                          cout0 + 3 * qstride + NC * 4 <= a.Cout;
        if (fast) {
            const unsigned e0 = rm.vox0 * (unsigned)a.Cout + (unsigned)cbase, rs = (unsigned)(a.Dx * a.Cout);
            uint16_t* o16 = reinterpret_cast<uint16_t*>(out) + e0;
            const uint16_t* r16 = res ? reinterpret_cast<const uint16_t*>(res) + e0 : nullptr;
            const bool want = a.gnp != nullptr;
            auto row = [&](int v, bool with_res) {
                float val[NC * 4];
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) val[c * 4 + j] = acc[v][c][j] + badd[c * 4 + j];
                if (with_res) {
#pragma unroll
                    for (int i = 0; i < NC / 2; ++i) {
                        Piece<bf16_t> pr;
                        pr.load(*reinterpret_cast<const uint4*>(r16 + (size_t)v * rs + i * 8));
#pragma unroll
                        for (int j = 0; j < 8; ++j) val[i * 8 + j] += pr.f[j];
                    }
                }
                if (want) {
#pragma unroll
                    for (int j = 0; j < NC * 4; ++j) { gs[j] += val[j]; gq[j] = fmaf(val[j], val[j], gq[j]); }
                }
#pragma unroll
                for (int i = 0; i < NC / 2; ++i)
                    *reinterpret_cast<uint4*>(o16 + (size_t)v * rs + i * 8) =
                        make_uint4(pack_bf16x2(val[i * 8], val[i * 8 + 1]), pack_bf16x2(val[i * 8 + 2], val[i * 8 + 3]),
                                   pack_bf16x2(val[i * 8 + 4], val[i * 8 + 5]), pack_bf16x2(val[i * 8 + 6], val[i * 8 + 7]));
            };
            if (r16) {
#pragma unroll
                for (int v = 0; v < NV; ++v) row(v, true);
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) row(v, false);
            }
            if (want)
                gn_partials_reduce<NC, G::NW>(gs, gq, gn_sm, a.gnp + ((size_t)n * (a.ntz * a.nty * a.ntx) + tile) * a.Cout * 2, cout0, a.Cout,
                                              cwave, lane, qstride);
            return;
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (!rm.ok(a, v)) continue;
        const unsigned vo = rm.vox(a, v) * (unsigned)a.Cout + (unsigned)cbase;
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
        gn_partials_reduce<NC, G::NW>(gs, gq, gn_sm, a.gnp + ((size_t)n * (a.ntz * a.nty * a.ntx) + tile) * a.Cout * 2, cout0, a.Cout,
                                      cwave, lane, qstride);
}

// ---------------------------------------------------------------------------------------------
// dgrad epilogue with the GroupNorm+SiLU(+dropout) backward reduction folded in (see ConvArgs::gx1).
// Per output element: xhat = (x - mean) * rstd, yhat = xhat * gamma + beta, s = sigmoid(yhat),
//   dyh = acc * keep/(1-p) * s * (1 + yhat * (1 - s));   S1[c] += dyh;  S2[c] += dyh * xhat;   store dyh.
// The per-(tile, channel) sums go through the same fixed-order fold as the forward statistics (gn_partials_reduce):
// the backward pass stays bit-reproducible and needs no float atomics.  vdm_gn_bwd_finalize sums the tiles.
// A lane's NC*4 consecutive channels are processed in sub-chunks of <= 8 (one bf16 piece): the per-channel constants of a
// sub-chunk (32 registers) are loaded once and reused over the NV rows.
// ---------------------------------------------------------------------------------------------
template <typename T, int SUB>
__device__ __forceinline__ void st_sub(T* p, const float (&f)[SUB]) {
    constexpr int EPL = DT<T>::EPL;
    if constexpr (SUB >= EPL) {
#pragma unroll
        for (int i = 0; i < SUB / EPL; ++i) {
            Piece<T> pc;
#pragma unroll
            for (int j = 0; j < EPL; ++j) pc.f[j] = f[i * EPL + j];
            *reinterpret_cast<uint4*>(p + i * EPL) = pc.store();
        }
    } else {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
    }
}

// Everything the folded epilogue reads from memory, as registers: the raw x pieces of the lane's NV rows, the dropout keep bytes
// and the per-channel affine of -yhat log2(e) = x * A + B (yhat = (x - mean) rstd gamma + beta).  gnb_issue() runs BEFORE the tap loop
// when the registers allow it (NC <= 2: the loads then retire behind the staging barrier, their latency is never exposed) or right
// after it (NC = 4: the operand registers of the tap loop are free by then).
// The second sum is taken against the RAW x (S2raw = sum dyh * x); vdm_gn_bwd_finalize converts the tile totals,
// sum dyh * xhat = rstd * (S2raw - mean * S1), so the inner loop needs neither xhat nor the per-group constants.
#ifndef VDM_GNB_PACKED
#define VDM_GNB_PACKED 1
#endif
#ifndef VDM_GNB_TABLE
#define VDM_GNB_TABLE 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T, int NC, int NV>
struct GnbRegs {
    static constexpr int EPL = DT<T>::EPL;
    static constexpr int CH = NC * 4;                       // channels per lane
    static constexpr int SUB = CH >= 8 ? 8 : 4;             // channels per sub-chunk (<= one bf16 piece)
    static constexpr int NSUB = CH / SUB;
    static constexpr int XQ = (SUB * (int)sizeof(T) + 15) / 16;   // 16-byte words per sub-chunk load (bf16: 1, fp32 SUB=8: 2)
    static constexpr int NBT = CH >= EPL ? CH / EPL : 1;    // keep-mask bytes (= 16-byte pieces of y) per lane and row (<= 4)
    uint4 x[NV][NSUB][XQ];
    uint32_t mb[NV];                                        // the NBT keep bytes of the row, byte k = piece k of the lane
    float A[CH], B[CH];
};

template <typename T, int NC, int NV>
struct GnbLane {                                            // where this lane's channels live
    const T* xsrc; int xc, xoff, cbase; bool lane_ok;
    __device__ __forceinline__ GnbLane(const ConvArgs& a, int n, int lane, int cout0, int qstride) {
        const int q = lane >> 4;
        const int cbase0 = cout0 + q * qstride;
        lane_ok = cbase0 + NC * 4 <= a.Cout;
        cbase = lane_ok ? cbase0 : 0;                       // (idle lanes read valid memory, their results are discarded)
        const bool first = cbase < a.gc1;
        xc = first ? a.gc1 : a.gc2;
        xoff = first ? cbase : cbase - a.gc1;
        xsrc = reinterpret_cast<const T*>(first ? a.gx1 : a.gx2) + (size_t)n * ((size_t)a.Dz * a.Dy * a.Dx * xc);     // this sample
    }
};

template <typename T, typename G, int NC, int NV>
__device__ __forceinline__ void gnb_issue_row(GnbRegs<T, NC, NV>& r, const GnbLane<T, NC, NV>& L, const RowMap<G, NV>& rm, const ConvArgs& a,
                                              int n, int v) {
    using R = GnbRegs<T, NC, NV>;
    constexpr int EPL = R::EPL, SUB = R::SUB, NSUB = R::NSUB;
    const int PPV = a.Cout / EPL;
    const unsigned vox = rm.ok(a, v) ? rm.vox(a, v) : 0u;   // (rows outside the tensor read voxel 0: valid memory, discarded)
#pragma unroll
    for (int sc = 0; sc < NSUB; ++sc) {
        const T* px = L.xsrc + vox * L.xc + L.xoff + sc * SUB;
        if constexpr (SUB * sizeof(T) >= 16) {
#pragma unroll
            for (int k = 0; k < R::XQ; ++k) r.x[v][sc][k] = reinterpret_cast<const uint4*>(px)[k];
        } else {
            const uint2 u = *reinterpret_cast<const uint2*>(px);
            r.x[v][sc][0] = make_uint4(u.x, u.y, 0u, 0u);
        }
    }
    uint32_t m = 0xffffffffu;
    if (a.gmask) {
        const unsigned char* pm = a.gmask + (size_t)n * ((size_t)a.Dz * a.Dy * a.Dx * PPV) + vox * (unsigned)PPV + L.cbase / EPL;
        m = 0;
#pragma unroll
        for (int k = 0; k < R::NBT; ++k) m |= (uint32_t)pm[k] << (8 * k);
    }
    r.mb[v] = m;
}

template <typename T, int NC, int NV>
__device__ __forceinline__ void gnb_issue_consts(GnbRegs<T, NC, NV>& r, const GnbLane<T, NC, NV>& L, const ConvArgs& a, int n) {
    const int gs = a.Cout / a.gG;
#pragma unroll
    for (int j = 0; j < NC * 4; ++j) {
        const int c = L.cbase + j;
        const int g = c / gs;
        const float sum = a.gstats[((size_t)n * a.gG + g) * 2], sq = a.gstats[((size_t)n * a.gG + g) * 2 + 1];
        const float mean = sum / a.gcnt;
        const float var = fmaxf(sq / a.gcnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + a.geps);
        const float A = rstd * a.ggamma[c];
        r.A[j] = -1.44269504089f * A;                       // yl = -yhat * log2(e) = x * A' + B'  (sigmoid = 1 / (1 + 2^yl))
        r.B[j] = -1.44269504089f * (a.gbeta[c] - mean * A);
    }
}

// The same constants ONCE per workgroup: thread t < 64 evaluates channel cout0 + t into an LDS table (the lanes of a workgroup cover at most
// 64 channels: index q * qstride + j), the lanes read their NC * 4 pairs back behind the workgroup's next barrier.  gnb_issue_consts costs
// every lane ~40 instructions per channel (two exact divisions, the group index by integer division, rsqrt): 640 per lane at NC = 4 - as
// many as the element-wise epilogue itself.  (Same arithmetic: bit-identical constants.)
constexpr int GNB_TABLE_BYTES = 64 * 2 * 4;
__device__ __forceinline__ void gnb_consts_table(float* tab, const ConvArgs& a, int n, int cout0, int tid) {
    if (tid < 64) {
        const int c = cout0 + tid;
        float A2 = 0.f, B2 = 0.f;
        if (c < a.Cout) {
            const int gs = a.Cout / a.gG;
            const int g = c / gs;
            const float sum = a.gstats[((size_t)n * a.gG + g) * 2], sq = a.gstats[((size_t)n * a.gG + g) * 2 + 1];
            const float mean = sum / a.gcnt;
            const float var = fmaxf(sq / a.gcnt - mean * mean, 0.f);
            const float rstd = rsqrtf(var + a.geps);
            const float A = rstd * a.ggamma[c];
            A2 = -1.44269504089f * A;
            B2 = -1.44269504089f * (a.gbeta[c] - mean * A);
        }
        tab[2 * tid] = A2;
        tab[2 * tid + 1] = B2;
    }
}
template <typename T, int NC, int NV>
__device__ __forceinline__ void gnb_consts_read(GnbRegs<T, NC, NV>& r, const float* tab, int lane, int qstride) {
    const float* t = tab + 2 * ((lane >> 4) * qstride);
#pragma unroll
    for (int j = 0; j < NC * 4; j += 2) {
        const float4 v = *reinterpret_cast<const float4*>(t + 2 * j);
        r.A[j] = v.x; r.B[j] = v.y; r.A[j + 1] = v.z; r.B[j + 1] = v.w;
    }
}

// everything up front (NC <= 2, before the tap loop); ctab != nullptr: the constants come from the workgroup's table (gnb_consts_read, later)
template <typename T, typename G, int NC, int NV, bool TABLED = false>
__device__ __forceinline__ void gnb_issue(GnbRegs<T, NC, NV>& r, const ConvArgs& a, int n, int oz0, int oy0, int ox0, int cwave, int lane,
                                          int cout0, int qstride) {
    const GnbLane<T, NC, NV> L(a, n, lane, cout0, qstride);
    const RowMap<G, NV> rm(a, oz0, oy0, ox0, cwave, lane);
#pragma unroll
    for (int v = 0; v < NV; ++v) gnb_issue_row<T, G, NC, NV>(r, L, rm, a, n, v);
    if constexpr (!TABLED) gnb_issue_consts<T, NC, NV>(r, L, a, n);
}

template <typename T, int SUB, int XQ>
__device__ __forceinline__ void raw_unpack(const uint4 (&q)[XQ], float (&f)[SUB]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int k = 0; k < XQ; ++k) {
            f[4 * k] = __builtin_bit_cast(float, q[k].x); f[4 * k + 1] = __builtin_bit_cast(float, q[k].y);
            f[4 * k + 2] = __builtin_bit_cast(float, q[k].z); f[4 * k + 3] = __builtin_bit_cast(float, q[k].w);
        }
    } else {
        const uint32_t w[4] = {q[0].x, q[0].y, q[0].z, q[0].w};
#pragma unroll
        for (int i = 0; i < SUB / 2; ++i) {
            f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
            f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
        }
    }
}

// PREFETCHED: gnb_issue() already ran (before the tap loop).  Otherwise the rows are fetched here through a rolling window of PD
// rows (the lane's accumulators die row by row, which makes room for the x pieces still in flight).
template <typename T, typename G, int NC, int NV, bool PREFETCHED, bool TABLED = false>
__device__ __forceinline__ void conv_epilogue_gnb(const f32x4 (&acc)[NV][NC], const ConvArgs& a, GnbRegs<T, NC, NV>& r, int n,
                                                  int oz0, int oy0, int ox0, int cwave, int lane, float* gn_sm, int tile, int cout0,
                                                  int qstride, const float* ctab = nullptr) {
    using R = GnbRegs<T, NC, NV>;
    constexpr int EPL = R::EPL, CH = R::CH, SUB = R::SUB, NSUB = R::NSUB;
    constexpr int PD = NV < 3 ? NV : 3;                    // (4 rows in flight spill the NC = 4, 4x8x16 kernel)
    const int q = lane >> 4;
    const int cbase = cout0 + q * qstride;
    const int C = a.Cout;
    const GnbLane<T, NC, NV> L(a, n, lane, cout0, qstride);
    const RowMap<G, NV> rm(a, oz0, oy0, ox0, cwave, lane);
    if constexpr (!PREFETCHED) {
#pragma unroll
        for (int v = 0; v < PD; ++v) gnb_issue_row<T, G, NC, NV>(r, L, rm, a, n, v);
        if constexpr (!TABLED) gnb_issue_consts<T, NC, NV>(r, L, a, n);
    }
    if constexpr (TABLED) gnb_consts_read<T, NC, NV>(r, ctab, lane, qstride);
    float gsum[CH], gsq[CH];                               // S1 = sum dyh, S2raw = sum dyh * x
#pragma unroll
    for (int j = 0; j < CH; ++j) gsum[j] = gsq[j] = 0.f;
    f32x2 gs2[CH / 2], gq2[CH / 2];                        // (the same sums as channel pairs: bf16 path)
#pragma unroll
    for (int j = 0; j < CH / 2; ++j) gs2[j] = gq2[j] = f32x2{0.f, 0.f};
    const bool lane_ok = cbase + CH <= C;                  // (host: C % (NC*4) == 0 and no lane straddles c1)
    T* out = reinterpret_cast<T*>(a.out) + (size_t)n * ((size_t)a.Dz * a.Dy * a.Dx * C);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if constexpr (!PREFETCHED) {
            if (v + PD < NV) gnb_issue_row<T, G, NC, NV>(r, L, rm, a, n, v + PD);
        }
        if (!rm.ok(a, v) || !lane_ok) continue;
        const unsigned vox = rm.vox(a, v);
        const uint32_t mb = r.mb[v];
        if constexpr (sizeof(T) == 2 && VDM_GNB_PACKED) {
            // bf16 storage: explicit channel PAIRS (2i, 2i + 1) - one dword of x, two adjacent accumulator registers, one dword of dyh - so that
            // every arithmetic step is one packed instruction on registers that are already adjacent (the auto-vectorised scalar form paired
            // channels (0, 2), (1, 3) of the unpacked words and re-paired them with ~6 v_mov per pair: 5.2 k instructions per tile at NC = 4,
            // as many issue cycles as the 864 MFMAs); keep factor as sign-extended bit AND 1/(1-p): 2 instead of 3 instructions
            const uint32_t ikb = __builtin_bit_cast(uint32_t, a.ginv_keep);
            const uint32_t mbs = CH >= EPL ? mb : (mb >> ((unsigned)cbase % EPL));
#pragma unroll
            for (int sc = 0; sc < NSUB; ++sc) {
                const uint32_t w[4] = {r.x[v][sc][0].x, r.x[v][sc][0].y, r.x[v][sc][0].z, r.x[v][sc][0].w};
                uint32_t ow[SUB / 2];
#pragma unroll
                for (int i = 0; i < SUB / 2; ++i) {
                    const int jj = sc * SUB + 2 * i;        // channels jj, jj + 1 of the lane: accumulator (jj / 4, jj % 4), (jj / 4, jj % 4 + 1)
                    const f32x2 xv = {__builtin_bit_cast(float, w[i] << 16), __builtin_bit_cast(float, w[i] & 0xffff0000u)};
                    const f32x2 A2 = {r.A[jj], r.A[jj + 1]}, B2 = {r.B[jj], r.B[jj + 1]};
                    const f32x2 yl = xv * A2 + B2;
                    const f32x2 e1 = f32x2{__builtin_amdgcn_exp2f(yl.x), __builtin_amdgcn_exp2f(yl.y)} + 1.0f;
                    const f32x2 sg = {__builtin_amdgcn_rcpf(e1.x), __builtin_amdgcn_rcpf(e1.y)};
                    const f32x2 ds = sg * ((yl * -0.69314718056f) * (1.0f - sg) + 1.0f);      // silu'(y) = s (1 + y (1 - s))
                    const int bit = CH >= EPL ? (jj / EPL) * 8 + jj % EPL : jj;
                    const f32x2 keep = {__builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_sbfe((int)mbs, bit, 1) & ikb),
                                        __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_sbfe((int)mbs, bit + 1, 1) & ikb)};
                    const f32x2 ac = {acc[v][jj >> 2][jj & 3], acc[v][jj >> 2][(jj & 3) + 1]};
                    const f32x2 dd = ac * (ds * keep);
                    gs2[jj / 2] += dd;
                    gq2[jj / 2] = dd * xv + gq2[jj / 2];
                    ow[i] = pack_bf16x2(dd.x, dd.y);
                }
                T* po = out + (vox * (unsigned)C + (unsigned)(cbase + sc * SUB));
                if constexpr (SUB == 8) *reinterpret_cast<uint4*>(po) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                else *reinterpret_cast<uint2*>(po) = make_uint2(ow[0], ow[1]);
            }
            continue;
        }
#pragma unroll
        for (int sc = 0; sc < NSUB; ++sc) {
            float xv[SUB], d[SUB];
            raw_unpack<T, SUB, R::XQ>(r.x[v][sc], xv);
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                const int jj = sc * SUB + j;               // channel inside the lane: accumulator (jj / 4, jj % 4)
                const float yl = fmaf(xv[j], r.A[jj], r.B[jj]);
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(yl));     // v_exp_f32 + v_rcp_f32 (1 ulp each)
                const float ds = sg * fmaf(-0.69314718056f * yl, 1.0f - sg, 1.0f);             // silu'(y) = s (1 + y (1 - s))
                const int bit = CH >= EPL ? (jj / EPL) * 8 + jj % EPL : (cbase + jj) % EPL;
                const float keep = ((mb >> bit) & 1u) ? a.ginv_keep : 0.f;
                const float dd = acc[v][jj >> 2][jj & 3] * (ds * keep);
                d[j] = dd;
                gsum[jj] += dd;
                gsq[jj] = fmaf(dd, xv[j], gsq[jj]);
            }
            st_sub<T, SUB>(out + (vox * (unsigned)C + (unsigned)(cbase + sc * SUB)), d);
        }
    }
    if constexpr (sizeof(T) == 2 && VDM_GNB_PACKED) {
#pragma unroll
        for (int j = 0; j < CH / 2; ++j) {
            gsum[2 * j] = gs2[j].x; gsum[2 * j + 1] = gs2[j].y;
            gsq[2 * j] = gq2[j].x; gsq[2 * j + 1] = gq2[j].y;
        }
    }
    gn_partials_reduce<NC, G::NW>(gsum, gsq, gn_sm, a.gnp + ((size_t)n * (a.ntz * a.nty * a.ntx) + tile) * a.Cout * 2, cout0, a.Cout, cwave,
                                  lane, qstride);
}

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
struct ClsMasks { unsigned m[64]; };

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

// ---------------------------------------------------------------------------------------------
// weight-gradient operand staging / transposed fetch (conv_wgrad.hip, conv_dgw.hip)
// ---------------------------------------------------------------------------------------------
// (row-wise like stage_halo_dma_rows: the row (oz, oy) is wave-uniform - scalar validity and address; per lane only the x part,
// computed once per tile.)  sub = 1: the class sub-grid dOut[2c + p] of the up-sampling conv's parity class (pz, py, px).
template <typename T, typename G>
__device__ __forceinline__ void stage_dout_dma_gen(char* lds, const T* __restrict__ g, const ConvArgs& a, int n, int oz0, int oy0,
                                                   int ox0, int cb, int cstride, int sub, int pz, int py, int px, int wave, int lane) {
    constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB, SH = DT<T>::SHIFT;
    const int k = lane >> 2, j = lane & 3;
    const int pc = j ^ ((k >> 1) & 3);
    const int co = cb * KB + pc * EPL;
    const int ox = ox0 + k;
    const int m = sub ? 2 : 1;                                               // fine-grid voxels per tile voxel and dimension
    const bool okx = co < a.Cout && ox < a.Dx;
    const unsigned xoff = (__umul24((unsigned)(m * ox + px) & 0xffffffu, (unsigned)cstride) + (unsigned)co) << SH;
    const char* gn = reinterpret_cast<const char*>(g) + (((size_t)n * (m * a.Dz) * (m * a.Dy) * (m * a.Dx) * cstride) << SH);
    const char* zp = reinterpret_cast<const char*>(g_zero_page);
    for (int r = wave; r < G::ROWS; r += 4) {
        const int oz = oz0 + r / G::TY, oy = oy0 + r % G::TY;
        const bool okrow = oz < a.Dz && oy < a.Dy;
        const unsigned fz = okrow ? (unsigned)(m * oz + pz) : 0u, fy = okrow ? (unsigned)(m * oy + py) : 0u;
        const size_t rowel = (size_t)((fz * (unsigned)(m * a.Dy) + fy) * (unsigned)(m * a.Dx)) * (unsigned)cstride;
        const char* src = (okrow && okx) ? gn + (rowel << SH) + xoff : zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + r * 1024), 16, 0, 0);
    }
}

// one dOut row per call (a persistent walk issues them between MFMA groups): per-lane x part once per column
template <typename T, typename G>
struct DoutRowStager {
    static constexpr int EPL = DT<T>::EPL, KB = DT<T>::KB, SH = DT<T>::SHIFT;
    unsigned xoff;
    bool okx;
    const char* gn;
    int cstride;
    __device__ __forceinline__ DoutRowStager(const T* __restrict__ g, const ConvArgs& a, int n, int ox0, int cb, int cstride_, int lane) : cstride(cstride_) {
        const int k = lane >> 2, j = lane & 3;
        const int pc = j ^ ((k >> 1) & 3);
        const int co = cb * KB + pc * EPL;
        const int ox = ox0 + k;
        okx = co < a.Cout && ox < a.Dx;
        xoff = (__umul24((unsigned)ox & 0xffffffu, (unsigned)cstride) + (unsigned)co) << SH;
        gn = reinterpret_cast<const char*>(g) + (((size_t)n * a.Dz * a.Dy * a.Dx * cstride) << SH);
    }
    // tile row r (of G::ROWS) of the tile at (oz0, oy0) -> lds + r * 1024
    __device__ __forceinline__ void row(char* lds, const ConvArgs& a, int oz0, int oy0, int r) const {
        const int oz = oz0 + r / G::TY, oy = oy0 + r % G::TY;
        const bool okrow = oz < a.Dz && oy < a.Dy;
        const unsigned fz = okrow ? (unsigned)oz : 0u, fy = okrow ? (unsigned)oy : 0u;
        const size_t rowel = (size_t)((fz * (unsigned)a.Dy + fy) * (unsigned)a.Dx) * (unsigned)cstride;
        const char* src = (okrow && okx) ? gn + (rowel << SH) + xoff : reinterpret_cast<const char*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + r * 1024), 16, 0, 0);
    }
};

template <typename T, typename G>
__device__ __forceinline__ void stage_dout_dma(char* lds, const T* __restrict__ g, const ConvArgs& a, int n, int oz0, int oy0,
                                               int ox0, int cb, int cstride, int wave, int lane) {
    stage_dout_dma_gen<T, G>(lds, g, a, n, oz0, oy0, ox0, cb, cstride, 0, 0, 0, 0, wave, lane);
}

// dOut tile of one parity class of the up-sampling conv: the tile's coarse voxels c map to the fine voxels 2c + p.
template <typename T, typename G>
__device__ __forceinline__ void stage_dout_dma_sub(char* lds, const T* __restrict__ g, const ConvArgs& a, int n, int oz0, int oy0,
                                                   int ox0, int cb, int cstride, int pz, int py, int px, int wave, int lane) {
    stage_dout_dma_gen<T, G>(lds, g, a, n, oz0, oy0, ox0, cb, cstride, 1, pz, py, px, wave, lane);
}

// Transposed operand fetch from the x-swizzled voxel-major image: 16 channels (tile ct of the 64-B block) x the
// k-step's voxels.  address = LDS base + wave-uniform row/tap offset (uni) + per-lane offset(s) (computed once per tap).
//   bf16: NOFF = 1, two ds_read_b64_tr_b16 (rows r and r+1; the second row is a compile-time byte delta HI);
//         lane: g = lane>>4 (voxels 4g..4g+3), li = lane&15: voxel-in-quad q' = li>>2, column quad p = li&3.
//         Voxels x and x+4 of a 32-lane half use opposite piece pairs ((hx>>1)&3 differs by 2): conflict-free.
//   fp32: NOFF = 4, four ds_read_b32 (MFMA step s reads voxel x = 4*s + (lane>>4), channel lane&15).
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


// ---------------------------------------------------------------------------------------------
// host side (shared planning helpers; the launchers live next to their kernels)
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
    // the kernels index voxels and elements INSIDE one sample with 32-bit arithmetic (24-bit multiplies on the coordinates).
    // Order matters: bound every factor first, so that the products below cannot overflow for hostile descriptors (UBSan finding).
    VDM_REQUIRE((long long)(d->od > d->oh ? (d->od > d->ow ? d->od : d->ow) : (d->oh > d->ow ? d->oh : d->ow)) * d->stride < (1 << 12),
                "conv: spatial extent too large");
    VDM_REQUIRE(d->cin <= (1 << 16) && d->cout <= (1 << 16) && d->n <= (1 << 20), "conv: channel count / batch out of range");
    const long long fine = (long long)d->od * d->oh * d->ow * (d->stride == 2 ? 8 : 1);           // < 2^36
    const long long cmax = cpad(d->cin > d->cout ? d->cin : d->cout, d->dtype);                   // <= 2^16
    VDM_REQUIRE(fine * cmax < (1LL << 32), "conv: %lld voxels x %lld channels per sample exceed the 32-bit in-sample index", fine, cmax);
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
    if (const char* f = getenv("VDM4CDM_FORCE_TZ")) {      // experiments: force the z extent of the tile (4 | 2 | 1)
        tz = atoi(f);
        if (tz == 1 || tz == 2 || tz == 4) return;
        tz = 4;
    }
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

// K-split kernel of the deepest levels (conv_fwd.hip, conv_ksplit_kernel): bf16 3x3x3 stride-1 convs with >= 4 K-blocks and 64-cout
// chunks whose grid is small enough for small_grid_tile() to leave the 4x8x16 tile.  Takes the (tz, ty) small_grid_tile chose and
// returns the tile the K-split kernel uses (1 x ty x 16; a 2x8x16 choice becomes 1x8x16).  VDM4CDM_KSPLIT: 0 off, 1 (default) only
// where the generic choice was a 1 x ty x 16 tile (level 3 of the 128^3 network), 2 also the 2x8x16 grids (level 2: measured SLOWER
// there - 0.078 vs 0.054 ms at batch 2 - one workgroup per CU loses more than the saved weight traffic gains).
static bool ksplit_tile(const ConvArgs& a, int& tz, int& ty) {
    static const int level = getenv("VDM4CDM_KSPLIT") ? atoi(getenv("VDM4CDM_KSPLIT")) : 1;
    if (level <= 0 || a.nkb < 4 || a.Cout % 64 != 0 || tz > 2 || (tz == 2 && level < 2)) return false;
    if (tz == 2) { tz = 1; ty = 8; }
    return true;
}

// stride-2 conv: output tile tz x 4 x 16.  2x4x16 needs a 5x9x33-voxel image (95 KB: ONE workgroup per CU, nothing overlaps its
// staging); 1x4x16 needs 3x9x33 voxels (57 KB: two per CU).  VDM4CDM_S2_TZ selects (experiments).
static int s2_tile_z() {
    static const int v = getenv("VDM4CDM_S2_TZ") ? atoi(getenv("VDM4CDM_S2_TZ")) : 2;
    return v == 1 ? 1 : 2;
}

// persistent workgroups of a weight-gradient launch over all (cout, cin) block pairs (~2 per CU); VDM4CDM_WGRAD_WGS: experiments
static int wgrad_wgs() {
    static const int v = getenv("VDM4CDM_WGRAD_WGS") ? atoi(getenv("VDM4CDM_WGRAD_WGS")) : 512;
    return v > 0 ? v : 512;
}

static bool uses_kpack(int dtype, int ks, int stride, int ups, int K, int O, int out_f32) {
    return dtype == VDM_BF16 && ks == 3 && stride == 1 && !ups && K <= 8 && O <= 32 && !out_f32 && getenv("VDM4CDM_NO_KPACK") == nullptr;
}
static bool uses_kpack(const vdm_conv_desc* d, int dgrad) {
    return uses_kpack(d->dtype, d->ksize, d->stride, d->upsample, dgrad ? d->cout : d->cin, dgrad ? d->cin : d->cout, dgrad ? 0 : d->out_f32);
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

static void fill_dims(ConvArgs& a, const vdm_conv_desc* d) {
    a.N = d->n; a.Dz = d->od; a.Dy = d->oh; a.Dx = d->ow;
    a.Iz = d->od * d->stride; a.Iy = d->oh * d->stride; a.Ix = d->ow * d->stride;
    a.Sz = d->upsample ? a.Iz / 2 : a.Iz; a.Sy = d->upsample ? a.Iy / 2 : a.Iy; a.Sx = d->upsample ? a.Ix / 2 : a.Ix;
    a.circular = d->pad_mode == VDM_PAD_CIRCULAR;
}

// spatial tile (TZ, TY; TX = 16) that vdm_conv_fwd will use for this conv - mirrors launch_fwd / launch_fwd_geo
static void fwd_tile_shape(const ConvArgs& a, int dtype, int out_f32, int ks, int stride, int ups, int& tz, int& ty) {
    tz = 4; ty = 8;
    if (stride == 2) { tz = s2_tile_z(); ty = 4; return; }
    if (uses_kpack(dtype, ks, stride, ups, a.Cin, a.Cout, out_f32)) return;
    if (ks == 3) {
        small_grid_tile(a, tz, ty);
        if (dtype == VDM_BF16 && stride == 1 && !ups && !out_f32) ksplit_tile(a, tz, ty);
    }
}

// launchers (defined in conv_fwd.hip / conv_cls.hip / conv_wgrad.hip)
int launch_fwd(const ConvArgs& a, int dtype, int out_f32, int ks, int stride, int ups, int nc, hipStream_t s);
int launch_fwd_gnb(const ConvArgs& a, int dtype, int nc, hipStream_t s);
int launch_fwd_gnp(const ConvArgs& a, int out_f32, int nc, hipStream_t s);
// wgrad_thin.hip: weight gradient of conv_in / conv_out (one thin side)
int wgrad_thin_mode(int dtype, int ksize, int stride, int upsample, int cin, int cout, bool want_bias, bool accumulate);
size_t wgrad_thin_workspace_bytes(int n, int od, int oh, int ow, int cdense);
int launch_wgrad_thin(int mode, const void* x, const void* dout, int n, int od, int oh, int ow, int cin, int cout, int circular, float* dw,
                      float* dbias, void* workspace, size_t workspace_bytes, hipStream_t s);
int run_cls(const vdm_conv_desc* d, int kind, const void* x, const void* w, const float* bias, const void* res, void* out,
            int cd, int ch, int cw, hipStream_t s, float* gn_partials = nullptr);
// conv_wgrad.hip: fixed-order fold of the P per-workgroup slabs [P][27][32][32] (+ [P][32] bias partials) of conv_dgw.hip
int launch_dgw_reduce(const float* slabs, const float* bslabs, float* dw, float* dbias, int P, int accumulate, hipStream_t s);
int launch_wgrad_any(const WgradArgs& w, float* dw, float* db, int acc, int cout, int cin, int ks, int stride, int ups, size_t ws,
                     int dtype, hipStream_t s);

}  // namespace vdm
