// vdm4cdm_amd: common device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes.  All activation tensors are NDHWC ("voxel-major": [n][z][y][x][c]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vdm4cdm_hip.h"

namespace vdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct bf16_t {            // storage-only bf16 (arithmetic is always fp32)
    uint16_t v;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {      // RNE, NaN-preserving (plain cast)
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}
// (Round 4: the one-instruction form `__builtin_convertvector(f32x2 -> bf16x2)` = v_cvt_pk_bf16_f32 saves an sdwa-or per pair, but with it
//  the training step stopped being bit-reproducible run to run (62 gradient tensors of the down path at 128^3 differ by ~1e-4 relative in
//  3 of 5 repetitions; same sources with this scalar form: 0 of 5 at 128^3 and 192^3, `tools/repro_bits.py`).  Cause not established - a
//  vector with one undefined lane bit-cast to a dword is undefined as a whole (pieces with padding channels), or a hazard the compiler
//  misses between the quarter-rate instructions that feed the conversion and the new instruction.  VDM_PACK_CVT=1 re-enables the vector
//  form for experiments.)
#ifndef VDM_PACK_CVT
#define VDM_PACK_CVT 0
#endif
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#if VDM_PACK_CVT
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{lo, hi}, bf16x2_));
#else
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
#endif
}

// Per-dtype constants.  A "piece" is 16 bytes of consecutive channels of one voxel
// (EPL elements); a "K-block" is 4 pieces = 64 bytes = KB channels.
template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int EPL = 4, KB = 16, ID = VDM_F32;
    static constexpr int SHIFT = 2;                  // log2(sizeof)
};
template <> struct DT<bf16_t> {
    static constexpr int EPL = 8, KB = 32, ID = VDM_BF16;
    static constexpr int SHIFT = 1;
};

// ---- 16-byte piece <-> fp32 lanes -------------------------------------------------------
template <typename T> struct Piece;                   // EPL floats
template <> struct Piece<float> {
    float f[4];
    __device__ __forceinline__ void load(const uint4& u) {
        f[0] = __builtin_bit_cast(float, u.x); f[1] = __builtin_bit_cast(float, u.y);
        f[2] = __builtin_bit_cast(float, u.z); f[3] = __builtin_bit_cast(float, u.w);
    }
    __device__ __forceinline__ uint4 store() const {
        return make_uint4(__builtin_bit_cast(uint32_t, f[0]), __builtin_bit_cast(uint32_t, f[1]),
                          __builtin_bit_cast(uint32_t, f[2]), __builtin_bit_cast(uint32_t, f[3]));
    }
};
template <> struct Piece<bf16_t> {
    float f[8];
    __device__ __forceinline__ void load(const uint4& u) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
            f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ uint4 store() const {
        return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                          pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
};

template <typename T> __device__ __forceinline__ float ld_elem(const T* p);
template <> __device__ __forceinline__ float ld_elem<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_elem<bf16_t>(const bf16_t* p) { return bf16_to_f32(p->v); }
template <typename T> __device__ __forceinline__ void st_elem(T* p, float v);
template <> __device__ __forceinline__ void st_elem<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_elem<bf16_t>(bf16_t* p, float v) { p->v = f32_to_bf16(v); }

// ---- fp32 storage on the bf16 matrix pipe (round 4) ------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 MFMA rate (exact fp32: 64 FLOP/clk/SIMD).  Default for fp32 storage since round 4:
// every fp32 operand is split into two bf16 halves, x = hi + lo (hi = bf16(x), lo = bf16(x - hi): |x - hi - lo| <= 2^-17 |x|), and a
// product is three bf16 MFMAs with fp32 accumulation, a b ~= hi_a lo_b + lo_a hi_b + hi_a hi_b - relative error 2^-16 per product (the
// dropped lo lo term), tighter than the TF32 (2^-11) path the reference's GPU convolutions take.  v_mfma_f32_16x16x16_bf16 takes the four
// consecutive channels of an fp32 piece as its four k-values per lane: the LDS image, the staging and the packed-weight SIZE are the
// fp32 ones (a packed weight fragment is 16 bytes per lane either way: 4 floats, or 4 hi + 4 lo halves).
// -DVDM_FP32_SPLIT=0 builds the exact variant (libvdm4cdm_hip_fp32exact.so, selected by VDM4CDM_FP32_EXACT=1).
#ifndef VDM_FP32_SPLIT
#define VDM_FP32_SPLIT 1
#endif
// (hi01, hi23, lo01, lo23) of the four floats of a piece
__device__ __forceinline__ uint4 split_frag(const uint4& raw) {
    const float x0 = __builtin_bit_cast(float, raw.x), x1 = __builtin_bit_cast(float, raw.y), x2 = __builtin_bit_cast(float, raw.z),
                x3 = __builtin_bit_cast(float, raw.w);
    const uint32_t h01 = pack_bf16x2(x0, x1), h23 = pack_bf16x2(x2, x3);
    const float r0 = x0 - __builtin_bit_cast(float, h01 << 16), r1 = x1 - __builtin_bit_cast(float, h01 & 0xffff0000u),
                r2 = x2 - __builtin_bit_cast(float, h23 << 16), r3 = x3 - __builtin_bit_cast(float, h23 & 0xffff0000u);
    return make_uint4(h01, h23, pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
}
// element i of a packed-weight buffer (fragments of [64 lanes][EPL]): fp32 split mode stores (hi, lo) halves instead of the float
template <typename T> __device__ __forceinline__ void st_packed_w(T* p, size_t i, float v) { st_elem<T>(p + i, v); }
template <> __device__ __forceinline__ void st_packed_w<float>(float* p, size_t i, float v) {
#if VDM_FP32_SPLIT
    uint16_t* q = reinterpret_cast<uint16_t*>(p) + (i >> 2) * 8;
    const int j = (int)(i & 3);
    const uint16_t hi = f32_to_bf16(v);
    q[j] = hi;
    q[4 + j] = f32_to_bf16(v - bf16_to_f32(hi));
#else
    p[i] = v;
#endif
}

// ---- wave / block reductions (wave64) -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// (v_rcp_f32 instead of the correctly rounded division: the activation is stored as bf16; every forward path uses this one form)
__device__ __forceinline__ float silu_f(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }
__device__ __forceinline__ float sigmoid_f(float y) { return 1.0f / (1.0f + __expf(-y)); }

// ---- Philox4x32-10 counter RNG (dropout masks, sampler noise) ---------------------------------
struct Philox {
    static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
        const uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32);
        lo = (uint32_t)p;
    }
    // counter = (c0,c1,c2,c3), key = (k0,k1) -> 4 x u32
    __host__ __device__ static inline void gen(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint32_t h0, l0, h1, l1;
            mulhilo(M0, c0, h0, l0);
            mulhilo(M1, c2, h1, l1);
            const uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += W0; k1 += W1;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};
// Division of a block / tile index by a launch constant as multiply-high + shifts (Granlund-Montgomery round-up form, exact for all
// 32-bit n): on wave-uniform operands it is three SCALAR instructions - `b % a.ntx; b /= a.ntx` on runtime divisors compiled to a
// float-reciprocal sequence on the vector unit (~20 vector instructions per division, eight divisions per workgroup).
struct FastDiv { uint32_t m, s1, s2; };
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                          // ceil(log2 d)
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    const uint32_t t = __umulhi(n, f.m);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

__host__ __device__ inline float u32_to_unit(uint32_t u) {          // (0,1]
    return ((float)(u >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// ---- host-side error plumbing ------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define VDM_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            vdm::set_error(__VA_ARGS__);            \
            return VDM_ERR_ARG;                     \
        }                                           \
    } while (0)

#define VDM_LAUNCH_CHECK(what)                                   \
    do {                                                         \
        int _e = vdm::check_hip(hipGetLastError(), what);        \
        if (_e) return _e;                                       \
    } while (0)

}  // namespace vdm
