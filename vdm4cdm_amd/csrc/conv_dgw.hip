// conv_dgw.hip - input gradient (with the folded GroupNorm backward) AND weight gradient of one 32 -> 32 channel 3x3x3 stride-1 conv in
// ONE kernel that stages the gradient dy once (round 4; SURVEY.md section 7 step 4, DESIGN.md section 3 "K3'").
//
// Replaces, for y = conv(a) with a = drop(silu(gn(x))) [NB blocks.py:129-132: net1 / net2 of a ResNetBlock, backward pass], the pair
//   vdm_conv_dgrad_gn(dy)  : da = sum_t W[t]^T dy[. + 1 - t]  -> dyh, GroupNorm partial sums     (stages the dy halo)
//   vdm_conv_wgrad(a, dy)  : dW[t] = sum_v dy[v] (x) a[v + t - 1], db = sum_v dy[v]              (stages the a halo AND dy)
// The weight gradient is taken with dy as the SHIFTED operand: dW[t][co][ci] = sum_u a[u][ci] dy[u + 1 - t][co] - then both gradients
// read the same dy halo and the activation is needed without a halo.  A persistent workgroup walks up a column of 2 x 8 x 16 voxel
// steps (rolling z window as conv_roll_kernel: ring of 4 dy slices, 2 new ones per step) and stages per step 23 KB of dy and a 16 KB
// activation tile for 216 + 224 MFMAs per wave - the two separate kernels stage 46 KB + 124 KB for the same 512 voxels.
// One workgroup of EIGHT waves per CU, two roles: waves 0-3 run the input gradient (27 taps on four rows each, then the folded
// GroupNorm epilogue), waves 4-7 the weight gradient (their 7 taps over all 16 rows; 112 accumulator registers) and all the staging -
// a SIMD hosts one wave of each role, so the epilogue's vector work and stores of one run under the MFMAs of the other.  (A first
// version ran both loops one after the other in four waves with one wave per SIMD: 0.646 ms against 0.356 + 0.266 ms for the two
// separate kernels at level 0 - nothing overlaps anything with a single wave per SIMD.)  Three workgroup barriers per step: data
// landed / operands read (ring slots free) / the one inside the epilogue's GroupNorm fold.
#include <type_traits>

#include "conv_common.h"

namespace vdm {

struct DgwArgs {
    ConvArgs c;           // dgrad form: c.x = dy, c.w = dgrad-packed weights, c.out = dyh, fold fields; c.Cin = conv cout, c.Cout = conv cin
    const void* act;      // a [N, D, H, W, C]: the conv's saved input (activated tensor)
    float* slabs;         // [P][27][32][32] partial weight gradients, master layout [tap][cout][cin]
    float* bslabs;        // [P][32] partial bias gradients or NULL
};

#ifdef VDM_DGW_STAMPS      // diagnostic build (make variant NAME=dgwst DEFS=-DVDM_DGW_STAMPS): per-role phase times, summed over the steps of a workgroup
#define DGW_T(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define DGW_ACC(k, t1, t0) tl_acc[k] += (t1) - (t0)
#else
#define DGW_T(var) do { } while (0)
#define DGW_ACC(k, t1, t0) do { } while (0)
#endif

template <bool GNB>
__global__ void __launch_bounds__(512, 2) conv_dgw_kernel(const DgwArgs w) {
#ifdef VDM_DGW_STAMPS
    unsigned long long tl_acc[4] = {0, 0, 0, 0};
#endif
    using T = bf16_t;
    using G = Geo<3, 1, 2, 8>;
    using TF = TrFetch<T>;
    constexpr int NC = 2, NV = G::NV, R = 5, SLICE = G::HY * G::HX * 64, HROW = G::HX * 64, ATILE = G::ROWS * 1024;
    constexpr int TPW = 7;
    static_assert(NV == 4 && G::ROWS == 16, "two z slabs of eight rows per step, four rows per wave");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring = lds;
    char* abuf = lds + R * SLICE;
    float* gn_sm = reinterpret_cast<float*>(lds + R * SLICE + 2 * ATILE);
    const ConvArgs& a = w.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, rw = wave & 3;               // waves 0-3: input gradient, waves 4-7: weight gradient (+ all staging)

    uint32_t b = (uint32_t)xcd_remap(blockIdx.x, gridDim.x);
    const int pidx = (int)b;                                 // slab of this workgroup
    uint32_t q = fdiv(b, a.fdx);
    const int tx = (int)(b - q * (uint32_t)a.ntx); b = q;
    q = fdiv(b, a.fdy);
    const int ty = (int)(b - q * (uint32_t)a.nty); b = q;
    q = fdiv(b, a.fdz);                                      // (divides by nseg)
    const int seg = (int)(b - q * (uint32_t)a.nseg);
    const int n = (int)q;
    const int zs0 = seg * a.zsteps, zs1 = min(a.ntz, zs0 + a.zsteps);
    const int oy0 = ty * G::TY, ox0 = tx * 16;

    const T* dy = reinterpret_cast<const T*>(a.x);
    const T* act = reinterpret_cast<const T*>(w.act);
    const RowStager<T, G, 0> st(dy, a, n, 0, oy0, ox0, 0, lane, 1, 0, 0, 0, a.Sz, a.Sy, a.Sx);
    // dy slices by position p = iz + 1 - 2 zs0 in the walk, slot p mod 5 (four slices in use by a step + one free slot: the first of
    // the next step's two new slices is staged at the top of a step, the second once position p0 is free); rows dealt to `nw` waves
    auto stage_dy = [&](int p0, int cnt, int w0, int nw) {
        for (int r = w0; r < cnt * G::HY; r += nw) {
            const int sl = r / G::HY, hy = r % G::HY, p = p0 + sl;
            st.row(ring + (p % R) * SLICE + hy * HROW, a, 2 * zs0 + p, hy);
        }
    };
    auto stage_act = [&](int s) {                            // the 2 x 8 x 16 activation tile of step s (no halo); weight-gradient waves
        stage_dout_dma<T, G>(abuf + (s & 1) * ATILE, act, a, n, s * G::TZ, oy0, ox0, 0, a.Cout, rw, lane);
    };
    stage_dy(0, 4, wave, 8);
    if (role == 1) stage_act(zs0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // B1 of the first step

    if (role == 0) {
        // =================================================================== input gradient: 27 taps on the wave's four rows
        constexpr int WPD = 6;                               // (four rows per wave: a tap is 8 MFMAs = 128 cycles; 2 taps ahead is less than an L2 hit)
        int lanex[3];
        operand_lane_offsets<G, NV>(lanex, rw & 1, lane);    // rows (rw & 1) * 4 ... of slab rw >> 1 (the slab is a ring slot)
        const uint4* wk = reinterpret_cast<const uint4*>(a.w) + lane;
        const int slabw = rw >> 1;
        // what the folded epilogue reads from memory (x rows, keep bytes, per-channel constants) is fetched one step ahead: issued
        // behind the previous epilogue's stores, landed by the time this step's taps are through
        GnbRegs<T, NC, GNB ? NV : 1> gr;
        if constexpr (GNB) gnb_issue<T, G, NC, NV>(gr, a, n, zs0 * G::TZ, oy0, ox0, rw, lane, 0, NC * 4);
        for (int s = zs0; s < zs1; ++s) {
            const int oz0 = s * G::TZ, p0 = 2 * (s - zs0);
            DGW_T(t0);
            const uint4* wks = wk;
            asm volatile("" : "+v"(wks));                    // (keeps the 54 weight-fragment addresses inside the step loop)
            uint4 wf[WPD + 1][NC];
            rr_prefetch_weights<NC, WPD, NC>(wf, wks);
            f32x4 acc[NV][NC];
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int zoff[3] = {((p0 + slabw) % R) * SLICE, ((p0 + slabw + 1) % R) * SLICE, ((p0 + slabw + 2) % R) * SLICE};
            taps_rowreuse_z<T, G, NC, NV, WPD, NC>(acc, ring, wks, wf, lanex, zoff);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DGW_T(t1);
            __builtin_amdgcn_s_barrier();                    // B2: every wave has read its operands of this step
            DGW_T(t2);
            const int tile = (s * a.nty + ty) * a.ntx + tx;
            if constexpr (GNB) {                             // (one workgroup barrier inside: B3)
                conv_epilogue_gnb<T, G, NC, NV, true>(acc, a, gr, n, oz0, oy0, ox0, rw, lane, gn_sm, tile, 0, NC * 4);
                if (s + 1 < zs1) gnb_issue<T, G, NC, NV>(gr, a, n, oz0 + G::TZ, oy0, ox0, rw, lane, 0, NC * 4);
            } else {
                float badd[NC * 4];
#pragma unroll
                for (int j = 0; j < NC * 4; ++j) badd[j] = 0.f;
                conv_epilogue<T, T, G, NC, NV>(acc, a, badd, n, oz0, oy0, ox0, rw, lane, gn_sm, tile, 0, NC * 4);
                __builtin_amdgcn_s_barrier();                // B3
            }
            DGW_T(t3);
            __builtin_amdgcn_s_barrier();                    // B1 of the next step (the staging waves have waited for their DMA)
            DGW_T(t4);
            DGW_ACC(0, t1, t0); DGW_ACC(1, t2, t1); DGW_ACC(2, t3, t2); DGW_ACC(3, t4, t3);      // taps | wait B2 | epilogue | wait B1
        }
        __builtin_amdgcn_s_barrier();                        // (the one barrier of the other role's tail: bias fold through LDS)
#ifdef VDM_DGW_STAMPS
        if (lane == 0 && rw == 0) for (int k = 0; k < 4; ++k) w.slabs[(size_t)gridDim.x * (27 * 32 * 32 + 32) + (size_t)pidx * 8 + k] = (float)tl_acc[k];
#endif
        return;
    }

    // ======================================================================= weight gradient (+ bias gradient, + all staging)
    int lo_in[TPW][2][1], lo_do[2][1], tap_dz[TPW], tap_dy[TPW];      // (ring and tile addresses are relative to `lds`)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        int tap = rw + 4 * t;
        if (tap >= 27) tap = 26;                             // dummy (wave 3's eighth slot never exists: 27 = 7 + 7 + 7 + 6)
        tap_dz[t] = tap / 9; tap_dy[t] = (tap / 3) % 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) TF::lane_off(lo_in[t][j], j, 1, tap % 3, lane);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) TF::lane_off(lo_do[i], i, 1, 0, lane);
    const int ntap = rw == 3 ? 6 : 7;
    f32x4 accw[TPW][2][2];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) accw[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

    for (int s = zs0; s < zs1; ++s) {
        const int oz0 = s * G::TZ, p0 = 2 * (s - zs0);
        DGW_T(u0);
        if (s + 1 < zs1) {
            stage_dy(p0 + 4, 1, rw, 4);                      // first new slice of the next step: into the free slot
            stage_act(s + 1);                                // (the other tile buffer was last read one step ago)
        }
        // per-lane bases of the transposed fetches for this step (the ring slots of a tap's z slice are wave-uniform but not compile-time):
        // vb[slab of the k-step][tap][tile j], va[tile i]; everything else of an address is a compile-time ds_read offset
        int vb[2][TPW][2], va[2];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                vb[0][t][j] = lo_in[t][j][0] + ((p0 + tap_dz[t]) % R) * SLICE + tap_dy[t] * HROW;
                vb[1][t][j] = lo_in[t][j][0] + ((p0 + 1 + tap_dz[t]) % R) * SLICE + tap_dy[t] * HROW;
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) va[i] = lo_do[i][0] + (int)(R * SLICE) + (s & 1) * ATILE;
        auto fetch = [&](int vbase, auto hi_tag, int imm) -> uint4 {       // two transposed 8-byte reads: rows r and r + 1 (byte delta HI)
            constexpr int HI = decltype(hi_tag)::value;
            typedef __attribute__((address_space(3))) s16x4* lptr;
            const char* pp = lds + vbase + imm;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(pp));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(pp + HI));
            const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
            return make_uint4(l2.x, l2.y, h2.x, h2.y);
        };
        using HiA = std::integral_constant<int, 1024>;
        using HiB = std::integral_constant<int, HROW>;
        {
            uint4 af[2], afn[2], bfA[2], bfB[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = fetch(va[i], HiA{}, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfA[j] = fetch(vb[0][0][j], HiB{}, 0);
#pragma unroll
            for (int r = 0; r < G::ROWS; r += 2) {
                if (r == G::TY) {                            // slab 0 done: nobody reads position p0 any more (the other role is through its taps)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();            // B2
                    if (s + 1 < zs1) stage_dy(p0 + 5, 1, rw, 4);      // second new slice of the next step: into the slot of position p0
                }
                const int sl = r / G::TY, ry = r % G::TY;
                const int rn = r + 2 < G::ROWS ? r + 2 : r;
                const int sln = rn / G::TY, ryn = rn % G::TY;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    uint4 (&cur)[2] = (t & 1) ? bfB : bfA;
                    uint4 (&nxt)[2] = (t & 1) ? bfA : bfB;
                    if (t + 1 < TPW) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) nxt[j] = fetch(vb[sl][t + 1][j], HiB{}, ry * HROW);
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i) afn[i] = fetch(va[i], HiA{}, rn * 1024);
#pragma unroll
                        for (int j = 0; j < 2; ++j) nxt[j] = fetch(vb[sln][0][j], HiB{}, ryn * HROW);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // a-tile rows are the A operand (16 channels ci x 32 voxels), the shifted dy rows the B operand (32 voxels x 16 co)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) mma16<T>(accw[t][i][j], af[i], cur[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = afn[i];
#pragma unroll
                for (int j = 0; j < 2; ++j) bfA[j] = bfB[j];  // (TPW odd: the last prefetch went to bfB)
            }
        }
        // ---- bias gradient: column sums of the step's own dy voxels (ring rows hy = 1..8, hx = 1..16 of positions p0+1, p0+2); a
        // lane reads the 16-byte slot (lane & 3) of voxel lane >> 2: always the same channel piece (x-swizzle), 8 running sums
        if (w.bslabs != nullptr) {
            const int vx = lane >> 2, sl4 = lane & 3;
            const bool xin = ox0 + vx < a.Dx;                 // (circular padding wraps the halo: voxels past the volume are not zeros)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = rw * 4 + rr;
                const char* src = ring + ((p0 + 1 + r / G::TY) % R) * SLICE + (1 + r % G::TY) * HROW + (vx + 1) * 64 + sl4 * 16;
                Piece<T> pz;
                pz.load(*reinterpret_cast<const uint4*>(src));
                const bool rin = xin && oz0 + r / G::TY < a.Dz && oy0 + r % G::TY < a.Dy;
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[j] += rin ? pz.f[j] : 0.f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DGW_T(u1);
        __builtin_amdgcn_s_barrier();                        // B3 (the barrier inside the other role's epilogue)
        DGW_T(u2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next step's dy slices and activation tile have landed
        DGW_T(u3);
        __builtin_amdgcn_s_barrier();                        // B1 of the next step
        DGW_T(u4);
        DGW_ACC(0, u1, u0); DGW_ACC(1, u2, u1); DGW_ACC(2, u3, u2); DGW_ACC(3, u4, u3);      // both halves (incl. wait B2) | wait B3 | DMA wait | wait B1
    }
#ifdef VDM_DGW_STAMPS
    if (lane == 0 && rw == 0) for (int k = 0; k < 4; ++k) w.slabs[(size_t)gridDim.x * (27 * 32 * 32 + 32) + (size_t)pidx * 8 + 4 + k] = (float)tl_acc[k];
#endif

    // ---- partial bias gradient of this workgroup: fixed-order fold through LDS
    {
        float* shb = reinterpret_cast<float*>(lds);
        const int t4 = rw * 64 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) shb[t4 * 8 + j] = bsum[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (matched by the input-gradient waves' last barrier)
        if (w.bslabs != nullptr && t4 < 32) {
            const int piece = t4 / 8, j = t4 % 8;
            float tot = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int vx = 0; vx < 16; ++vx) {            // the lane of voxel vx that holds this channel piece (hx = vx + 1)
                    const int ln = vx * 4 + (piece ^ (((vx + 1) >> 1) & 3));
                    tot += shb[(wv * 64 + ln) * 8 + j];
                }
            w.bslabs[(size_t)pidx * 32 + t4] = tot;
        }
    }
    // ---- partial weight gradient: accw[t][i][j][rg] = G[tap'][ci = 16 i + 4 gq + rg][co = 16 j + col] with tap' the mirrored tap
    float* slab = w.slabs + (size_t)pidx * (27 * 32 * 32);
    const int gq = lane >> 4, col = lane & 15;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tapm = rw + 4 * t;
        if (t < ntap) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        slab[((26 - tapm) * 32 + j * 16 + col) * 32 + i * 16 + gq * 4 + rg] = accw[t][i][j][rg];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool dgw_supported(const vdm_conv_desc* d) {
    static const bool off = getenv("VDM4CDM_NO_DGW") != nullptr;
    return !off && d->dtype == VDM_BF16 && d->ksize == 3 && d->stride == 1 && !d->upsample && d->cin == 32 && d->cout == 32 && d->od >= 8 &&
           (long long)d->n * cdiv(d->oh, 8) * cdiv(d->ow, 16) >= cu_count() / 2;
}

struct DgwPlan { int ntz, nty, ntx, nseg, zsteps, P; };
static DgwPlan dgw_plan(const vdm_conv_desc* d) {
    DgwPlan p;
    p.ntz = cdiv(d->od, 2); p.nty = cdiv(d->oh, 8); p.ntx = cdiv(d->ow, 16);
    const long long ncols = (long long)d->n * p.nty * p.ntx;
    int nseg = (int)((cu_count() + ncols - 1) / ncols);                 // one persistent workgroup per CU
    if (nseg > p.ntz / 4) nseg = p.ntz / 4;
    if (nseg < 1) nseg = 1;
    p.zsteps = cdiv(p.ntz, nseg);
    p.nseg = cdiv(p.ntz, p.zsteps);
    p.P = (int)(ncols * p.nseg);
    return p;
}

}  // namespace vdm

using namespace vdm;

extern "C" int vdm_conv_dgw_supported(const vdm_conv_desc* d) { return validate(d) == VDM_OK && dgw_supported(d) ? 1 : 0; }

extern "C" int vdm_conv_dgw_tiles(const vdm_conv_desc* d) {
    if (validate(d) != VDM_OK || !dgw_supported(d)) return 0;
    const DgwPlan p = dgw_plan(d);
    return p.ntz * p.nty * p.ntx;
}

extern "C" size_t vdm_conv_dgw_workspace_bytes(const vdm_conv_desc* d) {
    if (validate(d) != VDM_OK || !dgw_supported(d)) return 0;
    const DgwPlan p = dgw_plan(d);
    return (size_t)p.P * (27 * 32 * 32 + 32 + 8) * sizeof(float);      // (+ 8 per workgroup: phase times of the diagnostic build)
}

extern "C" int vdm_conv_dgrad_gn_wgrad(const vdm_conv_desc* d, const void* dout, const void* w_packed_dgrad, const void* act, void* dyh,
                                       const vdm_gn_fold* f, float* dw, float* dbias, int accumulate, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    int e = validate(d);
    if (e) return e;
    VDM_REQUIRE(dout && w_packed_dgrad && act && dyh && f && dw && workspace, "conv_dgrad_gn_wgrad: NULL pointer");
    VDM_REQUIRE(dgw_supported(d), "conv_dgrad_gn_wgrad: only bf16 3x3x3 stride-1 convs with 32 -> 32 channels on large grids (vdm_conv_dgw_supported)");
    VDM_REQUIRE(f->x1 && f->stats && f->gamma && f->beta && f->partials && f->groups > 0, "conv_dgrad_gn_wgrad: NULL pointer in the fold");
    VDM_REQUIRE(f->c1 + f->c2 == 32 && (f->c2 == 0 || f->x2) && 32 % f->groups == 0 && (f->c2 == 0 || f->c1 % 8 == 0),
                "conv_dgrad_gn_wgrad: the GroupNorm input must have the conv's 32 input channels (c1 %d + c2 %d)", f->c1, f->c2);
    const DgwPlan p = dgw_plan(d);
    VDM_REQUIRE(workspace_bytes >= (size_t)p.P * (27 * 32 * 32 + 32) * sizeof(float), "conv_dgrad_gn_wgrad: workspace too small");
    DgwArgs w{};
    ConvArgs& a = w.c;
    a.x = dout; a.w = w_packed_dgrad; a.out = dyh; a.gnp = f->partials;
    a.N = d->n; a.Dz = d->od; a.Dy = d->oh; a.Dx = d->ow;
    a.Iz = a.Sz = d->od; a.Iy = a.Sy = d->oh; a.Ix = a.Sx = d->ow;
    a.circular = d->pad_mode == VDM_PAD_CIRCULAR;
    a.Cin = 32; a.CinStride = 32; a.Cout = 32; a.nchunks = 1; a.nkb = 1;
    a.gx1 = f->x1; a.gx2 = f->x2; a.gc1 = f->c1; a.gc2 = f->c2; a.gG = f->groups;
    a.gstats = f->stats; a.ggamma = f->gamma; a.gbeta = f->beta; a.gmask = f->keep_mask;
    a.geps = f->eps; a.ginv_keep = f->keep_mask ? f->inv_keep : 1.0f;
    a.gcnt = (float)((double)d->od * d->oh * d->ow * (32 / f->groups));
    a.ntz = p.ntz; a.nty = p.nty; a.ntx = p.ntx; a.nseg = p.nseg; a.zsteps = p.zsteps;
    a.fdx = make_fastdiv((uint32_t)a.ntx); a.fdy = make_fastdiv((uint32_t)a.nty); a.fdz = make_fastdiv((uint32_t)a.nseg);
    a.fdn = make_fastdiv((uint32_t)a.N);
    w.act = act;
    w.slabs = (float*)workspace;
    w.bslabs = dbias ? w.slabs + (size_t)p.P * 27 * 32 * 32 : nullptr;
    using G = Geo<3, 1, 2, 8>;
    const size_t lds = (size_t)5 * G::HY * G::HX * 64 + 2 * (size_t)G::ROWS * 1024 + GN_SCRATCH_BYTES;
    auto kern = conv_dgw_kernel<true>;
    static unsigned long long lds_done = 0;
    e = set_lds(kern, lds, lds_done);
    if (e) return e;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(kern, dim3((unsigned)p.P), dim3(512), lds, s, w);
    VDM_LAUNCH_CHECK("conv_dgw_kernel");
    return launch_dgw_reduce(w.slabs, w.bslabs, dw, dbias, p.P, accumulate, s);
}
