// conv_fwd.hip - generic forward / dgrad convolution kernel and the tap-packed kernel (see conv_common.h).
#include "conv_common.h"

namespace vdm {

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel, one tile per workgroup (all variants; small grids)
// ---------------------------------------------------------------------------------------------
// SPLIT (NC == 2 only): the weights are packed for 64-cout chunks (4 tiles per tap) but a workgroup takes HALF a chunk (tiles 2h,
// 2h+1 = couts 16q + 8h .. +7 of every lane group q): twice the workgroups for the small grids of the deep levels, where a
// workgroup's K-blocks run strictly one after the other and only co-resident workgroups overlap staging with MFMAs.
// GNB: the GroupNorm+SiLU backward reduction is folded into the epilogue (dgrad launches that feed a GroupNorm: conv_epilogue_gnb).
// GNP: the conv input is silu(gn(x)) of the raw tensor x (inference): GroupNorm + SiLU are applied to the staged image in LDS
// (gn_prologue_inplace) instead of by a pass of their own.
// NW = 8 (large bf16 3x3x3 stride-1 grids): the SAME tile and LDS image shared by eight waves of NV = 4 rows - two workgroups per CU
// are then four waves per SIMD (<= 128 registers each) instead of two: twice the waves to issue the staging DMA, to cover the
// barrier-to-barrier phases of the co-resident workgroup and to drain the epilogue, at unchanged staged bytes per voxel.
template <typename T, typename TO, int KS, int STRIDE, int UPS, int NC, int TZ, int TY, bool SPLIT = false, bool GNB = false, bool GNP = false, int NW = 4>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 4 : ((TZ * TY <= 16 && NC <= 2) ? 3 : 2)) conv_fwd_kernel(const ConvArgs a) {
    static_assert(!SPLIT || NC == 2, "half-chunk mode is the NC=2 kernel on NC=4 weights");
    static_assert(!(GNB && GNP), "the prologue belongs to forward convs, the folded backward to dgrad convs");
    constexpr int NCW = SPLIT ? 4 : NC;
    using G = Geo<KS, STRIDE, TZ, TY, NW>;
    constexpr int NV = G::NV, TAPS = G::TAPS;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef VDM_TIMELINE
    unsigned long long tl_t[7] = {0, 0, 0, 0, 0, 0, 0};
    VDM_STAMP(0);
#endif

#ifdef VDM_STAGGER
    // experiment: the second workgroup a CU receives at launch (wave slot 1 of its SIMDs) starts half a tile period late, so that the
    // two co-resident workgroups run in anti-phase (one stages / stores while the other is in its tap loop)
    if (blockIdx.x < 2u * 256u && (__builtin_amdgcn_s_getreg((4 << 11) | 4) & 1)) {      // HW_REG_HW_ID bits [3:0] = wave slot
        for (int i = 0; i < VDM_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    int tx, ty, tz, n, chunk;
    decode_tile(a, (uint32_t)xcd_remap(blockIdx.x, gridDim.x), tx, ty, tz, n, chunk);
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
    VDM_STAMP(5);
    // folded GroupNorm backward: what the epilogue reads from memory is fetched up front when the registers allow it (NC <= 2):
    // the loads retire behind the first staging barrier instead of stalling the epilogue
    const int e_cout0 = SPLIT ? (chunk >> 1) * 64 + (chunk & 1) * 8 : chunk * NC * 16, e_qstride = SPLIT ? 16 : NC * 4;
    GnbRegs<T, NC, GNB ? NV : 1> gr;
    constexpr bool GNB_PRE = GNB && NC <= 2 && NW == 4;      // (eight-wave workgroups: 128 registers per wave, the rolling window instead)
    float* gnb_tab = reinterpret_cast<float*>(lds + ((G::HVOX + 15) / 16) * 1024 + GN_SCRATCH_BYTES);      // (behind the GN scratch; read after the first barrier)
    if constexpr (GNB && VDM_GNB_TABLE) gnb_consts_table(gnb_tab, a, n, e_cout0, tid);
    if constexpr (GNB_PRE) gnb_issue<T, G, NC, NV, VDM_GNB_TABLE != 0>(gr, a, n, oz0, oy0, ox0, wave, lane, e_cout0, e_qstride);
    float badd[NC * 4];                                      // bias + conditioning bias of the lane's channels (latency hidden behind the taps)
    if constexpr (!GNB) load_badd<NC>(badd, a, n, e_cout0 + (lane >> 4) * e_qstride);

    for (int kb = 0; kb < a.nkb; ++kb) {
        if (kb) __syncthreads();
        if constexpr (GNP)      // (the in-place GroupNorm prologue re-walks the chunks a wave staged: the chunk walk)
            stage_halo_dma_chunks<T, G, UPS, NW>(lds, x, a, n, oz0, oy0, ox0, kb, wave, lane, 1, 0, 0, 0, a.Sz, a.Sy, a.Sx);
        else
            stage_halo_dma<T, G, UPS>(lds, x, a, n, oz0, oy0, ox0, kb, wave, lane);
        if (kb == 0) VDM_STAMP(6);
        constexpr int IMG_ = ((G::HVOX + 15) / 16) * 1024;
        float* gn_tab = reinterpret_cast<float*>(lds + IMG_ + GN_SCRATCH_BYTES);
        if constexpr (GNP) {
            if (kb == 0) {                                  // (behind the first DMA issue: the table is built while the halo is in flight)
                gn_prologue_table(gn_tab, a, n, tid, 64 * NW);
                __syncthreads();
            }
        }
        const uint4* wk = wbase + (size_t)kb * TAPS * NCW * 64;
        if constexpr (sizeof(T) == 2) {
            constexpr int WPD = WPipe<NC>::WPD;
            uint4 wf[WPD + 1][NC];
            constexpr bool RR = VDM_ROWREUSE && KS == 3 && STRIDE == 1;     // activation rows reused across the dy taps
            if constexpr (RR) rr_prefetch_weights<NC, WPD, NCW>(wf, wk);
            else taps_prefetch_weights<TAPS, NC, WPD, NCW>(wf, wk);
            if (kb == 0) VDM_STAMP(1);
            if constexpr (GNP) gn_prologue_inplace<T, G, NW>(lds, gn_tab, a, oz0, oy0, ox0, kb, wave, lane);
            __syncthreads();
            if (kb == 0) VDM_STAMP(2);
#ifdef VDM_TAP_PRIO
            __builtin_amdgcn_s_setprio(VDM_TAP_PRIO);      // experiment: the wave in its tap loop wins instruction arbitration against the co-resident wave
#endif
            if constexpr (RR) taps_rowreuse<T, G, NC, NV, WPD, NCW>(acc, lds, wk, wf, lanex);
            else taps_pipelined<T, G, NC, NV, WPD, NCW>(acc, lds, wk, wf, lanex);
#ifdef VDM_TAP_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        } else {
            if constexpr (GNP) gn_prologue_inplace<T, G, NW>(lds, gn_tab, a, oz0, oy0, ox0, kb, wave, lane);
            __syncthreads();
            taps_rolled<T, G, NC, NV>(acc, lds, wk, lanex);
        }
    }
    VDM_STAMP(3);
    constexpr int IMG = ((G::HVOX + 15) / 16) * 1024;     // the GN scratch sits behind the operand image
    float* gn_sm = reinterpret_cast<float*>(lds + IMG);
    const int tile = (tz * a.nty + ty) * a.ntx + tx;
    if constexpr (GNB) {
        static_assert(sizeof(T) == sizeof(TO), "the folded GroupNorm backward stores dyh in the activation dtype");
        conv_epilogue_gnb<T, G, NC, NV, GNB_PRE, VDM_GNB_TABLE != 0>(acc, a, gr, n, oz0, oy0, ox0, wave, lane, gn_sm, tile, e_cout0, e_qstride, gnb_tab);
    } else
        conv_epilogue<T, TO, G, NC, NV>(acc, a, badd, n, oz0, oy0, ox0, wave, lane, gn_sm, tile, e_cout0, e_qstride);
#ifdef VDM_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the stores of the epilogue have left the wave's queue)
    VDM_STAMP(4);
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int k = 0; k < 7; ++k) o[k] = tl_t[k];
        o[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Rolling-z kernel (round 4) for the single-K-block bf16 3x3x3 stride-1 convs on large grids (level 0: 32 reduction channels).
// A tile stages 6 z-slices of the halo for 4 output slabs and its z neighbour stages 2 of them again (halo factor 2.1 = 6/4 * 10/8 *
// 18/16; beyond the 4 MB L2 of an XCD the re-read comes from the Infinity Cache / HBM).  Here a PERSISTENT workgroup walks up a
// column of tiles (fixed y, x; z ascending): the LDS image is a ring of 6 slice slots, a step stages only the 4 NEW slices (the two
// top slices of the previous step are the two bottom ones of this step) - 46 KB instead of 69.6 KB per tile, 34 % fewer LDS-DMA
// instructions, and the y / x halos are shared through the L2 with the neighbour columns that walk up in step.  Wave w owns output
// slab w of the step: its three input slices are ring slots (wave-uniform, taps_rowreuse_z).  The staging of step s + 1 is issued
// right behind the tap loop of step s (the slots it overwrites are free once every wave has left the taps) and lands under the
// epilogue's stores.
// ---------------------------------------------------------------------------------------------
template <typename T, int NC, bool GNB>
__global__ void __launch_bounds__(256, 2) conv_roll_kernel(const ConvArgs a) {
    static_assert(sizeof(T) == 2, "bf16 storage");
    using G = Geo<3, 1, 4, 8>;
    constexpr int NV = G::NV, TAPS = G::TAPS, NCW = NC, R = 6;
    constexpr int SLICE = G::HY * G::HX * 64;
    static_assert(NV == G::TY, "a wave owns one z slab of the step");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (tx, ty, segment of the column, n, chunk)
    uint32_t b = (uint32_t)xcd_remap(blockIdx.x, gridDim.x);
    uint32_t q = fdiv(b, a.fdx);
    const int tx = (int)(b - q * (uint32_t)a.ntx); b = q;
    q = fdiv(b, a.fdy);
    const int ty = (int)(b - q * (uint32_t)a.nty); b = q;
    q = fdiv(b, a.fdz);                                     // (fdz divides by nseg here)
    const int seg = (int)(b - q * (uint32_t)a.nseg); b = q;
    q = fdiv(b, a.fdn);
    const int n = (int)(b - q * (uint32_t)a.N);
    const int chunk = (int)q;
    const int zs0 = seg * a.zsteps, zs1 = min(a.ntz, zs0 + a.zsteps);
    const int oy0 = ty * G::TY, ox0 = tx * 16;

    int lanex[3];
    operand_lane_offsets<G, NV>(lanex, 0, lane);            // (no slab offset: the slab is a ring slot)
    const uint4* wk = reinterpret_cast<const uint4*>(a.w) + (size_t)chunk * TAPS * NCW * 64 + lane;
    const T* x = reinterpret_cast<const T*>(a.x);
    const int e_cout0 = chunk * NC * 16, e_qstride = NC * 4;
    float badd[NC * 4];
    if constexpr (!GNB) load_badd<NC>(badd, a, n, e_cout0 + (lane >> 4) * e_qstride);
    // the per-lane x part of the staging is the same for every step of the column
    const RowStager<T, G, 0> st(x, a, n, 0, oy0, ox0, 0, lane, 1, 0, 0, 0, a.Sz, a.Sy, a.Sx);
    float* gn_sm = reinterpret_cast<float*>(lds + R * SLICE);
    float* gnb_tab = reinterpret_cast<float*>(lds + R * SLICE + GN_SCRATCH_BYTES);
    if constexpr (GNB && VDM_GNB_TABLE) gnb_consts_table(gnb_tab, a, n, e_cout0, tid);     // (one sample, one chunk per column: once; read behind the first barrier)
    constexpr int WPD = WPipe<NC>::WPD;

    // slices are addressed by their position p = iz + 1 - 4 zs0 >= 0 in the column walk; slot = p mod 6
    auto stage = [&](int p0, int cnt) {                      // positions p0 .. p0 + cnt - 1
        for (int r = wave; r < cnt * G::HY; r += 4) {
            const int sl = r / G::HY, hy = r % G::HY;
            const int p = p0 + sl;
            st.row(lds + (p % R) * SLICE + hy * (G::HX * 64), a, 4 * zs0 + p, hy);      // (RowStager: iz = iz0 + hz with iz0 = -1)
        }
    };
    stage(0, R);
    for (int s = zs0; s < zs1; ++s) {
        const int oz0 = s * G::TZ, p0 = 4 * (s - zs0);      // this step reads positions p0 .. p0 + 5
        f32x4 acc[NV][NC];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        GnbRegs<T, NC, GNB ? NV : 1> gr;                   // (filled by the epilogue through its rolling window: the loop-carried state of
                                                           //  the walk leaves no room to prefetch all rows in front of the taps)
        // (LICM would hoist the 54 weight-fragment addresses of the unrolled tap loop out of the step loop: 108 registers and spills;
        //  an opaque copy of the pointer per step keeps them inside)
        const uint4* wks = wk;
        asm volatile("" : "+v"(wks));
        uint4 wf[WPD + 1][NC];
        rr_prefetch_weights<NC, WPD, NCW>(wf, wks);
        __syncthreads();                                    // (vmcnt(0) + barrier: the slices of this step have landed)
        const int zoff[3] = {((p0 + wave) % R) * SLICE, ((p0 + wave + 1) % R) * SLICE, ((p0 + wave + 2) % R) * SLICE};
        taps_rowreuse_z<T, G, NC, NV, WPD, NCW>(acc, lds, wks, wf, lanex, zoff);
        if (s + 1 < zs1) {                                  // next step's four new slices, behind this step's taps
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // every wave has read its operands: positions p0 .. p0 + 3 are free
            stage(p0 + R, 4);
        }
        const int tile = (s * a.nty + ty) * a.ntx + tx;
        if constexpr (GNB)
            conv_epilogue_gnb<T, G, NC, NV, false, VDM_GNB_TABLE != 0>(acc, a, gr, n, oz0, oy0, ox0, wave, lane, gn_sm, tile, e_cout0, e_qstride, gnb_tab);
        else
            conv_epilogue<T, T, G, NC, NV>(acc, a, badd, n, oz0, oy0, ox0, wave, lane, gn_sm, tile, e_cout0, e_qstride);
    }
}

// ---------------------------------------------------------------------------------------------
// K-split kernel for the deepest level (bf16, 3x3x3, stride 1, >= 4 K-blocks, 64-cout chunks, 1 x TY x 16 tiles) - conv_ksplit_kernel.
//
// What the generic kernel does there (tools/conv_timeline.py, 16^3 voxels x 256 channels, batch 1): 512 workgroups of 64 voxels x 32
// couts live 41 us each for 0.44 us of MFMA work per K-block - 8 K-block rounds (stage, barrier, 27 taps of 2 MFMAs), and every one
// of the four waves fetches the SAME 55 KB of weights per round: 3.9 MB through the vector L1 of a CU per launch = 92 GB/s of its
// 134 GB/s fill rate, for a conv with 3.5 MB of weights and 2 MB of activations.
// Here the four waves of a workgroup split the K-blocks instead of the voxel rows: wave w stages K-blocks w, w+4, ... into its
// PRIVATE LDS image (no workgroup barrier in the main loop: four independent pipelines hide each other's DMA latency), accumulates
// the whole tile (all TY rows x NC cout tiles) over them - so a weight fragment is fetched ONCE per workgroup - and the partial
// accumulators are summed across the waves through LDS in a fixed order.  The epilogue is the generic one (wave w owns rows
// w*TY/4 ...), incl. the folded GroupNorm backward.  One workgroup per CU (4 x 21 KB images at TY = 4).
// ---------------------------------------------------------------------------------------------
template <typename T, typename TO, int NC, int TY, bool GNB>
__global__ void __launch_bounds__(256, 1) conv_ksplit_kernel(const ConvArgs a) {
    static_assert(sizeof(T) == 2 && sizeof(TO) == 2 && (TY == 4 || TY == 8), "bf16, 1 x 4 x 16 or 1 x 8 x 16 tiles");
    using G = Geo<3, 1, 1, TY>;
    constexpr int NVT = G::ROWS;                          // rows a wave accumulates (the whole tile)
    constexpr int NVE = G::NV;                            // rows a wave owns in the epilogue
    constexpr int TAPS = G::TAPS, NCW = 4, WPD = NC == 4 ? 4 : 6;
    constexpr int IMG = ((G::HVOX + 15) / 16) * 1024;
    static_assert(NVT * NC * 1024 <= IMG, "the reduction buffer of a wave must fit its image");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tx, ty, tz, n, chunk;
    decode_tile(a, (uint32_t)xcd_remap(blockIdx.x, gridDim.x), tx, ty, tz, n, chunk);
    const int oz0 = tz, oy0 = ty * TY, ox0 = tx * 16;
    const int cout0 = chunk * NC * 16, qstride = NC * 4;

    f32x4 acc[NVT][NC];
#pragma unroll
    for (int v = 0; v < NVT; ++v)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[v][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int lanex[3];
    operand_lane_offsets<G, NVT>(lanex, 0, lane);         // every wave walks all rows of the tile
    const uint4* wbase = reinterpret_cast<const uint4*>(a.w) + (size_t)chunk * a.nkb * TAPS * NCW * 64 + lane;
    const T* x = reinterpret_cast<const T*>(a.x);
    // folded GroupNorm backward: everything the epilogue reads is fetched up front also at NC = 4 (one wave per SIMD: 512 registers),
    // with a single workgroup per CU nothing else would hide those loads
    GnbRegs<T, NC, GNB ? NVE : 1> gr;
    if constexpr (GNB) gnb_issue<T, G, NC, NVE>(gr, a, n, oz0, oy0, ox0, wave, lane, cout0, qstride);
    float badd[NC * 4];
    if constexpr (!GNB) load_badd<NC>(badd, a, n, cout0 + (lane >> 4) * qstride);

    char* image = lds + wave * IMG;
    for (int kb = wave; kb < a.nkb; kb += 4) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // (the previous round's operand reads have returned)
        stage_halo_dma_gen<T, G, 0, 1>(image, x, a, n, oz0, oy0, ox0, kb, 0, lane, 1, 0, 0, 0, a.Sz, a.Sy, a.Sx);
        const uint4* wk = wbase + (size_t)kb * TAPS * NCW * 64;
        uint4 wf[WPD + 1][NC];
        rr_prefetch_weights<NC, WPD, NCW>(wf, wk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // this wave's own DMA has landed: no barrier needed
        taps_rowreuse<T, G, NC, NVT, WPD, NCW>(acc, image, wk, wf, lanex);
    }
    // sum the partial accumulators of the four waves (fixed order: deterministic); wave w keeps rows w * NVE ... for the epilogue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                                            // every wave is done with its image
    f32x4* red = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int v = 0; v < NVT; ++v)
#pragma unroll
        for (int c = 0; c < NC; ++c) red[(size_t)wave * (IMG / 16) + (v * NC + c) * 64 + lane] = acc[v][c];
    __syncthreads();
    f32x4 tot[NVE][NC];
#pragma unroll
    for (int e = 0; e < NVE; ++e)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int off = ((wave * NVE + e) * NC + c) * 64 + lane;
            const f32x4 p0 = red[off], p1 = red[(IMG / 16) + off], p2 = red[2 * (IMG / 16) + off], p3 = red[3 * (IMG / 16) + off];
            tot[e][c] = (p0 + p1) + (p2 + p3);
        }
    float* gn_sm = reinterpret_cast<float*>(lds + 4 * IMG);
    const int tile = (tz * a.nty + ty) * a.ntx + tx;
    if constexpr (GNB)
        conv_epilogue_gnb<T, G, NC, NVE, true>(tot, a, gr, n, oz0, oy0, ox0, wave, lane, gn_sm, tile, cout0, qstride);
    else
        conv_epilogue<T, TO, G, NC, NVE>(tot, a, badd, n, oz0, oy0, ox0, wave, lane, gn_sm, tile, cout0, qstride);
}

// ---------------------------------------------------------------------------------------------
// Tap-packed kernel for the convs with <= 8 input channels (conv_in: 2 -> 32; input gradient of conv_out: 1 -> 32), bf16.
// One 16-byte piece holds ALL channels of a voxel, so the 32-deep K of an MFMA is filled with 4 TAPS x 8 channels instead of
// one tap x 32 channels of which 24..31 are padding: 7 tap groups instead of 27 taps (3.9x fewer MFMAs), an LDS image of
// 16 B per halo voxel (17 KB), one LDS-DMA lane per voxel.  Lane (voxel lx, k-chunk q) reads the voxel shifted by tap 4g+q;
// the packed weights hold W[tap 4g+q][cout][ci] in the matching A-fragment slot (zero for tap >= 27, ci >= Cin).
// These convs are bound by writing / reading the 32-channel tensor (HBM), not by MFMA.
// ---------------------------------------------------------------------------------------------
template <typename TO, int NC, bool GNB = false>
__global__ void __launch_bounds__(256, GNB ? 2 : 4) conv_kpack_kernel(const ConvArgs a) {
    using T = bf16_t;
    using G = Geo<3, 1, 4, 8>;
    constexpr int NV = G::NV, NG = (G::TAPS + 3) / 4;
    constexpr int NCH = (G::HVOX + 63) / 64;               // DMA chunks of 64 voxels (1 KiB)
    constexpr int ROWB = G::HX * 16;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tx, ty, tz, n, chunk;
    decode_tile(a, (uint32_t)xcd_remap(blockIdx.x, gridDim.x), tx, ty, tz, n, chunk);
    const int oz0 = tz * G::TZ, oy0 = ty * G::TY, ox0 = tx * 16;

    GnbRegs<T, NC, GNB ? NV : 1> gr;
    if constexpr (GNB) gnb_issue<T, G, NC, NV>(gr, a, n, oz0, oy0, ox0, wave, lane, chunk * NC * 16, NC * 4);
    float badd[NC * 4];
    if constexpr (!GNB) load_badd<NC>(badd, a, n, chunk * NC * 16 + (lane >> 4) * NC * 4);
    // stage the halo: one lane per voxel
    const T* x = reinterpret_cast<const T*>(a.x);
    {   // (same incremental / 32-bit address arithmetic as stage_halo_dma_gen; 256 halo voxels between two chunks of a wave)
        constexpr int DXs = 256 % G::HX, DYs = (256 / G::HX) % G::HY, DZs = 256 / (G::HX * G::HY);
        const int hv0 = wave * 64 + lane;
        int hx = hv0 % G::HX, hy = (hv0 / G::HX) % G::HY, hz = hv0 / (G::HX * G::HY);
        const T* xn = x + (size_t)n * ((size_t)a.Sz * a.Sy * a.Sx * a.CinStride);
        const bool fastwrap = a.Iz >= G::HZ && a.Iy >= G::HY && a.Ix >= G::HX;
        for (int c = wave; c < NCH; c += 4) {
            int iz = oz0 - 1 + hz, iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            bool ok = hz < G::HZ;
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
            const unsigned row = __umul24((unsigned)iz & 0xffffffu, (unsigned)a.Sy) + ((unsigned)iy & 0xffffffu);
            const unsigned vox = __umul24(row, (unsigned)a.Sx) + ((unsigned)ix & 0xffffffu);
            const void* src = ok ? static_cast<const void*>(xn + vox * (unsigned)a.CinStride) : static_cast<const void*>(g_zero_page);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + c * 1024), 16, 0, 0);
            hx += DXs; hy += DYs; hz += DZs;
            if (hx >= G::HX) { hx -= G::HX; hy += 1; }
            if (hy >= G::HY) { hy -= G::HY; hz += 1; }
        }
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
    if constexpr (GNB)
        conv_epilogue_gnb<T, G, NC, NV, true>(acc, a, gr, n, oz0, oy0, ox0, wave, lane, reinterpret_cast<float*>(lds + IMG),
                                              (tz * a.nty + ty) * a.ntx + tx, chunk * NC * 16, NC * 4);
    else
        conv_epilogue<T, TO, G, NC, NV>(acc, a, badd, n, oz0, oy0, ox0, wave, lane, reinterpret_cast<float*>(lds + IMG),
                                        (tz * a.nty + ty) * a.ntx + tx, chunk * NC * 16, NC * 4);
}

template <typename T, typename TO, int KS, int STRIDE, int UPS, int NC, int TZ, int TY, bool SPLIT = false, bool GNB = false, bool GNP = false, int NW = 4>
static int launch_fwd_cfg(const ConvArgs& a0, hipStream_t s) {
    using G = Geo<KS, STRIDE, TZ, TY, NW>;
    ConvArgs a = a0;
    if (SPLIT) a.nchunks *= 2;
    a.ntz = cdiv(a.Dz, TZ); a.nty = cdiv(a.Dy, TY); a.ntx = cdiv(a.Dx, 16);
    set_tile_divs(a);
    static const int lds_pad = getenv("VDM4CDM_LDS_PAD") ? atoi(getenv("VDM4CDM_LDS_PAD")) : 0;      // experiments: fewer workgroups per CU
    const size_t lds = (size_t)((G::HVOX + 15) / 16) * 1024 + GN_SCRATCH_BYTES + (GNP ? GNP_TABLE_BYTES : 0) + (GNB ? GNB_TABLE_BYTES : 0) + (size_t)(lds_pad > 0 ? lds_pad : 0);
    auto kern = conv_fwd_kernel<T, TO, KS, STRIDE, UPS, NC, TZ, TY, SPLIT, GNB, GNP, NW>;
    static unsigned long long lds_done = 0;
    {
        int e = set_lds(kern, lds, lds_done);
        if (e) return e;
    }
    const long long nwg = (long long)a.N * a.ntz * a.nty * a.ntx * a.nchunks;
    if (nwg > 0x7fffffffLL) { set_error("conv: grid too large"); return VDM_ERR_ARG; }
#ifdef VDM_TIMELINE
    a.stamps = g_timeline_stamps;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * NW), lds, s, a);
    VDM_LAUNCH_CHECK("conv_fwd_kernel");
    return VDM_OK;
}

// VDM4CDM_WG8: bit mask of the NC = 2 kernel classes that run eight-wave workgroups on the 4x8x16 tile - 1: plain, 2: folded GroupNorm backward
static bool wg8_enabled(int nc, bool gnb) {
    static const int mask = getenv("VDM4CDM_WG8") ? atoi(getenv("VDM4CDM_WG8")) : 0;
    return nc == 2 && ((mask >> (gnb ? 1 : 0)) & 1);
}

// rolling-z kernel: bf16, one K-block, 3x3x3 stride 1, bf16 output, 4x8x16 steps; VDM4CDM_ROLL=0 switches it off (A/B)
static bool roll_enabled() {
    static const bool on = getenv("VDM4CDM_ROLL") ? atoi(getenv("VDM4CDM_ROLL")) != 0 : true;
    return on;
}
template <typename T, int NC, bool GNB>
static int launch_roll(const ConvArgs& a0, hipStream_t s) {
    using G = Geo<3, 1, 4, 8>;
    ConvArgs a = a0;
    a.ntz = cdiv(a.Dz, G::TZ); a.nty = cdiv(a.Dy, G::TY); a.ntx = cdiv(a.Dx, 16);
    const long long ncols = (long long)a.N * a.nty * a.ntx * a.nchunks;
    int nseg = (int)((2LL * cu_count() + ncols - 1) / ncols);          // ~2 persistent workgroups per CU
    if (nseg > a.ntz / 2) nseg = a.ntz / 2;                             // >= 2 steps per segment, or the walk saves nothing
    if (nseg < 1) nseg = 1;
    a.zsteps = cdiv(a.ntz, nseg);
    a.nseg = cdiv(a.ntz, a.zsteps);
    a.fdx = make_fastdiv((uint32_t)a.ntx); a.fdy = make_fastdiv((uint32_t)a.nty); a.fdz = make_fastdiv((uint32_t)a.nseg);
    a.fdn = make_fastdiv((uint32_t)a.N);
    const size_t lds = (size_t)6 * G::HY * G::HX * 64 + GN_SCRATCH_BYTES + (GNB ? GNB_TABLE_BYTES : 0);
    auto kern = conv_roll_kernel<T, NC, GNB>;
    static unsigned long long lds_done = 0;
    {
        int e = set_lds(kern, lds, lds_done);
        if (e) return e;
    }
    const long long nwg = ncols * a.nseg;
    if (nwg > 0x7fffffffLL) { set_error("conv: grid too large"); return VDM_ERR_ARG; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, s, a);
    VDM_LAUNCH_CHECK("conv_roll_kernel");
    return VDM_OK;
}

template <typename T, typename TO, int NC, int TY, bool GNB>
static int launch_ksplit(const ConvArgs& a0, hipStream_t s) {
    using G = Geo<3, 1, 1, TY>;
    ConvArgs a = a0;
    a.ntz = a.Dz; a.nty = cdiv(a.Dy, TY); a.ntx = cdiv(a.Dx, 16);
    set_tile_divs(a);
    const size_t lds = 4 * (size_t)((G::HVOX + 15) / 16) * 1024 + GN_SCRATCH_BYTES;
    auto kern = conv_ksplit_kernel<T, TO, NC, TY, GNB>;
    static unsigned long long lds_done = 0;
    {
        int e = set_lds(kern, lds, lds_done);
        if (e) return e;
    }
    const long long nwg = (long long)a.N * a.ntz * a.nty * a.ntx * a.nchunks;
    if (nwg > 0x7fffffffLL) { set_error("conv: grid too large"); return VDM_ERR_ARG; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, s, a);
    VDM_LAUNCH_CHECK("conv_ksplit_kernel");
    return VDM_OK;
}

template <typename T, typename TO, int KS, int STRIDE, int UPS, int NC, bool GNB = false, bool GNP = false>
static int launch_fwd_geo(const ConvArgs& a, hipStream_t s) {
    if constexpr (STRIDE == 2)
        return s2_tile_z() == 1 ? launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 1, 4>(a, s) : launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 2, 4>(a, s);
    else if constexpr (KS == 3) {      // (fp32 storage too since round 4: on the bf16 pipe it is no longer MFMA-bound, small grids need small tiles)
        int tz, ty;
        small_grid_tile(a, tz, ty);
        if constexpr (NC == 4 && UPS == 0 && sizeof(TO) == 2 && sizeof(T) == 2) {
            if (ksplit_tile(a, tz, ty)) {
                if constexpr (GNP) { set_error("conv_fwd_gn: the K-split kernel has no GroupNorm prologue (vdm_conv_fwd_gn_supported)"); return VDM_ERR_UNSUPPORTED; }
                else return ty == 4 ? launch_ksplit<T, TO, 4, 4, GNB>(a, s) : launch_ksplit<T, TO, 4, 8, GNB>(a, s);
            }
            if (uses_split(a, tz, ty)) {
                if (tz == 1) return ty == 4 ? launch_fwd_cfg<T, TO, KS, STRIDE, UPS, 2, 1, 4, true, GNB, GNP>(a, s) : launch_fwd_cfg<T, TO, KS, STRIDE, UPS, 2, 1, 8, true, GNB, GNP>(a, s);
                return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, 2, 2, 8, true, GNB, GNP>(a, s);
            }
        }
        if (tz == 1) return ty == 4 ? launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 1, 4, false, GNB, GNP>(a, s) : launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 1, 8, false, GNB, GNP>(a, s);
        if (tz == 2) return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 2, 8, false, GNB, GNP>(a, s);
        if constexpr (STRIDE == 1 && UPS == 0 && sizeof(TO) == 2 && sizeof(T) == 2 && NC == 2 && !GNP) {      // one K-block, large grid: persistent walk up z
            if (a.nkb == 1 && cdiv(a.Dz, 4) >= 4 && roll_enabled()) return launch_roll<T, NC, GNB>(a, s);
        }
        if constexpr (STRIDE == 1 && UPS == 0 && sizeof(TO) == 2 && sizeof(T) == 2 && NC == 2 && !GNP) {      // large grids: eight waves share the 4x8x16 tile
            // (NC = 4 needs > 128 registers per wave: its accumulators alone are 64)
            if (wg8_enabled(NC, GNB)) return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 4, 8, false, GNB, GNP, 8>(a, s);
        }
        return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 4, 8, false, GNB, GNP>(a, s);
    } else
        return launch_fwd_cfg<T, TO, KS, STRIDE, UPS, NC, 4, 8, false, GNB, GNP>(a, s);
}

template <typename T, typename TO, int KS, int STRIDE, int UPS, bool GNB = false, bool GNP = false>
static int launch_fwd_nc(const ConvArgs& a, int nc, hipStream_t s) {
    switch (nc) {
        case 1: return launch_fwd_geo<T, TO, KS, STRIDE, UPS, 1, GNB, GNP>(a, s);
        case 2: return launch_fwd_geo<T, TO, KS, STRIDE, UPS, 2, GNB, GNP>(a, s);
        default: return launch_fwd_geo<T, TO, KS, STRIDE, UPS, 4, GNB, GNP>(a, s);
    }
}

template <typename T, typename TO>
static int launch_fwd_variant(const ConvArgs& a, int ks, int stride, int ups, int nc, hipStream_t s) {
    if (ks == 1) return launch_fwd_nc<T, TO, 1, 1, 0>(a, nc, s);
    if (stride == 2) return launch_fwd_nc<T, TO, 3, 2, 0>(a, nc, s);
    if (ups) return launch_fwd_nc<T, TO, 3, 1, 1>(a, nc, s);
    return launch_fwd_nc<T, TO, 3, 1, 0>(a, nc, s);
}


template <int NC, bool GNB = false>
static int launch_kpack(const ConvArgs& a0, hipStream_t s) {
    using G = Geo<3, 1, 4, 8>;
    ConvArgs a = a0;
    a.ntz = cdiv(a.Dz, G::TZ); a.nty = cdiv(a.Dy, G::TY); a.ntx = cdiv(a.Dx, 16);
    set_tile_divs(a);
    const size_t lds = (size_t)((G::HVOX + 63) / 64) * 1024 + GN_SCRATCH_BYTES;
    const long long nwg = (long long)a.N * a.ntz * a.nty * a.ntx * a.nchunks;
    if (nwg > 0x7fffffffLL) { set_error("conv: grid too large"); return VDM_ERR_ARG; }
    hipLaunchKernelGGL((conv_kpack_kernel<bf16_t, NC, GNB>), dim3((unsigned)nwg), dim3(256), lds, s, a);
    VDM_LAUNCH_CHECK("conv_kpack_kernel");
    return VDM_OK;
}

int launch_fwd(const ConvArgs& a, int dtype, int out_f32, int ks, int stride, int ups, int nc, hipStream_t s) {
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

// dgrad of a 3x3x3 stride-1 conv with the GroupNorm+SiLU backward reduction folded into the epilogue (ConvArgs::gx1 ... set)
int launch_fwd_gnb(const ConvArgs& a, int dtype, int nc, hipStream_t s) {
    if (uses_kpack(dtype, 3, 1, 0, a.Cin, a.Cout, 0)) return nc == 1 ? launch_kpack<1, true>(a, s) : launch_kpack<2, true>(a, s);
    if (dtype == VDM_F32) return launch_fwd_nc<float, float, 3, 1, 0, true>(a, nc, s);
    return launch_fwd_nc<bf16_t, bf16_t, 3, 1, 0, true>(a, nc, s);
}

// forward 3x3x3 stride-1 conv of y = silu(gn(x)) with GroupNorm + SiLU applied to the staged image (ConvArgs::gstats ... set; bf16)
int launch_fwd_gnp(const ConvArgs& a, int out_f32, int nc, hipStream_t s) {
    if (out_f32) return launch_fwd_geo<bf16_t, float, 3, 1, 0, 1, false, true>(a, s);
    return launch_fwd_nc<bf16_t, bf16_t, 3, 1, 0, false, true>(a, nc, s);
}

}  // namespace vdm
