// conv_wgrad.hip - weight-gradient kernels and their slab reduce kernels (see conv_common.h).
#include "conv_common.h"

namespace vdm {

// diagnostic build (make timeline; tools/wgrad_phases.py): per wave the s_memrealtime ticks (100 MHz) of each phase, summed over its tiles
#ifdef VDM_TIMELINE
#define WG_TL_DECL unsigned long long tl_acc[6] = {0, 0, 0, 0, 0, 0}, tl_last = __builtin_amdgcn_s_memrealtime(); const unsigned long long tl_first = tl_last
#define WG_TL(k)                                                          \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
        tl_acc[k] += now_ - tl_last;                                      \
        tl_last = now_;                                                   \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#define WG_TL_WRITE(a_)                                                                                     \
    do {                                                                                                    \
        if ((a_).stamps && lane == 0) {                                                                     \
            unsigned long long* o_ = (a_).stamps + ((size_t)blockIdx.x * 4 + wave) * 8;                      \
            for (int k_ = 0; k_ < 6; ++k_) o_[k_] = tl_acc[k_];                                             \
            o_[6] = tl_first;                                                                               \
            o_[7] = __builtin_amdgcn_s_memrealtime();                                                       \
        }                                                                                                   \
    } while (0)
#else
#define WG_TL_DECL do { } while (0)
#define WG_TL(k) do { } while (0)
#define WG_TL_WRITE(a_) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// wgrad kernel
// ---------------------------------------------------------------------------------------------

// dOut tile of a wgrad workgroup by LDS-DMA: OVOX voxels x 64 B (one cout block), same x-swizzled voxel-major
// image as stage_halo_dma (one chunk = one 16-voxel row).
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
    WG_TL_DECL;

    for (int tile = pidx; tile < w.ntiles; tile += w.P) {
        int tx, ty, tz, n, rest_;
        decode_tile(a, (uint32_t)tile, tx, ty, tz, n, rest_);      // (tile < ntiles: rest_ == 0)
        const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * 16;
        __syncthreads();                                   // every wave is done reading the previous tile
        WG_TL(0);
        stage_halo_dma<T, G, CLS ? 0 : UPS>(lds_in, x, a, n, oz0, oy0, ox0, kb, wave, lane);
        if constexpr (CLS)
            stage_dout_dma_sub<T, G>(lds_do, g, a, n, oz0, oy0, ox0, cb, w.dout_stride, pz, py, px, wave, lane);
        else
            stage_dout_dma<T, G>(lds_do, g, a, n, oz0, oy0, ox0, cb, w.dout_stride, wave, lane);
        WG_TL(1);
        __syncthreads();                                   // (drains the LDS-DMA: vmcnt(0) + barrier)
        WG_TL(2);
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
                    for (int j = 0; j < NTB; ++j) mma16_act<T>(acc[t][i][j], af[i], cur[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NTA; ++i) af[i] = afn[i];
            if (TPW & 1) {
#pragma unroll
                for (int j = 0; j < NTB; ++j) bfA[j] = bfB[j];
            }
        }
        WG_TL(3);
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
        WG_TL(4);
    }

    constexpr int CL = NT * 16;                               // channels per 64-B block
    if (TAPS > 1 && w.bslabs != nullptr && kb == 0) {          // workgroup-uniform condition
        float* shb = reinterpret_cast<float*>(lds);           // all tiles are done: the LDS image is free
        constexpr int EPLc = DT<T>::EPL;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPLc; ++j) shb[tid * EPLc + j] = bsum[j];
        __syncthreads();
        if (tid < CL) {                                       // fixed-order fold (no LDS atomics: the bias gradient is bit-reproducible)
            const int piece = tid / EPLc, j = tid % EPLc;
            float tot = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int vx = 0; vx < 16; ++vx) {             // the lane of voxel vx that holds this channel piece
                    const int ln = vx * 4 + (piece ^ ((vx >> 1) & 3));
                    tot += shb[(wv * 64 + ln) * EPLc + j];
                }
            w.bslabs[((size_t)cb * (CLS ? 8 : 1) * w.P + cls * w.P + pidx) * CL + tid] = tot;
        }
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
    WG_TL(5);
    WG_TL_WRITE(a);
}

// ---------------------------------------------------------------------------------------------
// wgrad kernel, halo rows of the shifted operand resident in registers (round 4; bf16, 3^3, stride 1, full 32 x 32 channel blocks)
// ---------------------------------------------------------------------------------------------
// conv_wgrad_kernel splits the 27 taps over its waves: per k-step (two 16-voxel rows) a wave reads 2 dOut fragments and, for each of its
// 7 taps, 2 fragments of the shifted input = 32 transposed LDS reads for 28 MFMAs, and the same halo row is read again for every (dy, k-step)
// that touches it.  Here a wave owns ONE 16 x 16 tile of the (cout, cin) block and all 27 taps (108 accumulator registers).  With the
// fragment's K order chosen as (row R: 16 voxels | row R + 1: 16 voxels) one ds_read_b64_tr_b16 is one whole halo row, so for a halo slab hz
// and a shift dx the TY + 2 rows are read ONCE (10 reads) and serve every (dz, dy, k-step) combination: tap (dz, dy, dx) of output slab
// oz = hz - dz multiplies the dOut fragment of rows (2j, 2j + 1) with the register pair (row 2j + dy, row 2j + dy + 1).  The dOut fragments
// of the whole tile stay in registers (TZ * TY / 2 fragments).  Per 2 x 8 x 16 tile and wave: 120 + 16 reads for 216 MFMAs (0.63 reads per MFMA
// instead of 1.14).  Same slab layout as conv_wgrad_kernel, same reduce kernels; the summation order per output differs (halo-slab-major).
template <typename T, int TZ, int TY>
__global__ void __launch_bounds__(256, 2) conv_wgrad_rows_kernel(const WgradArgs w) {
    using G = Geo<3, 1, TZ, TY>;
    using TF = TrFetch<T>;
    static_assert(sizeof(T) == 2 && (TY % 2) == 0, "bf16 only: a fragment = two rows of the same z-slab");
    constexpr int IN_BYTES = ((G::HVOX + 15) / 16) * 1024;
    constexpr int NJ = TY / 2, NR = TY + 2, NG = G::HZ * 3, CL = 32;
    typedef __attribute__((address_space(3))) s16x4* lptr;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* lds_in = lds;
    char* lds_do = lds + IN_BYTES;
    const ConvArgs& a = w.c;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ia = wave >> 1, jb = wave & 1;               // this wave's cout tile / cin tile of the 32 x 32 block
    const int pair = blockIdx.x / w.P, pidx = blockIdx.x % w.P;
    const int cb = pair / w.nkb, kb = pair % w.nkb;

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int lo_a[1], lo_b[3][1];
    TF::lane_off(lo_a, ia, 1, 0, lane);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) TF::lane_off(lo_b[dx], jb, 1, dx, lane);

    const T* x = reinterpret_cast<const T*>(a.x);
    const T* g = reinterpret_cast<const T*>(w.dout);
    float bsum[DT<T>::EPL];
#pragma unroll
    for (int j = 0; j < DT<T>::EPL; ++j) bsum[j] = 0.f;

    auto rows_of = [&](uint2 (&dst)[NR], int grp) {         // the TY + 2 halo rows of (halo slab, dx) = grp
        const int hz = grp / 3, dx = grp % 3;
#pragma unroll
        for (int r = 0; r < NR; ++r)
            dst[r] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds_in + (hz * G::HY + r) * G::HX * 64 + lo_b[dx][0])));
    };

    WG_TL_DECL;
    for (int tile = pidx; tile < w.ntiles; tile += w.P) {
        int tx, ty, tz, n, rest_;
        decode_tile(a, (uint32_t)tile, tx, ty, tz, n, rest_);
        const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * 16;
        __syncthreads();                                   // every wave is done reading the previous tile
        WG_TL(0);
        stage_halo_dma<T, G, 0>(lds_in, x, a, n, oz0, oy0, ox0, kb, wave, lane);
        stage_dout_dma<T, G>(lds_do, g, a, n, oz0, oy0, ox0, cb, w.dout_stride, wave, lane);
        WG_TL(1);
        __syncthreads();                                   // (drains the LDS-DMA: vmcnt(0) + barrier)
        WG_TL(2);
        uint4 af[TZ][NJ];
#pragma unroll
        for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
            for (int j = 0; j < NJ; ++j) af[oz][j] = TF::template get<1024>(lds_do, lo_a, (oz * TY + 2 * j) * 1024);
        uint2 rowsA[NR], rowsB[NR];
        rows_of(rowsA, 0);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            uint2 (&cur)[NR] = (grp & 1) ? rowsB : rowsA;
            uint2 (&nxt)[NR] = (grp & 1) ? rowsA : rowsB;
            if (grp + 1 < NG) rows_of(nxt, grp + 1);        // in flight behind this group's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            const int hz = grp / 3, dx = grp % 3;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int oz = hz - dz;
                    if (oz < 0 || oz >= TZ) continue;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const uint2 r0 = cur[2 * j + dy], r1 = cur[2 * j + dy + 1];
                        mma16_act<T>(acc[(dz * 3 + dy) * 3 + dx], af[oz][j], make_uint4(r0.x, r0.y, r1.x, r1.y));
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        WG_TL(3);
        // bias gradient: column sums of this dOut tile (see conv_wgrad_kernel)
        if (w.bslabs != nullptr && kb == 0) {
            const int vx = lane >> 2, sl = lane & 3;
#pragma unroll
            for (int r = wave; r < G::ROWS; r += 4) {
                Piece<T> pz;
                pz.load(*reinterpret_cast<const uint4*>(lds_do + r * 1024 + vx * 64 + sl * 16));
#pragma unroll
                for (int j = 0; j < DT<T>::EPL; ++j) bsum[j] += pz.f[j];
            }
        }
        WG_TL(4);
    }

    if (w.bslabs != nullptr && kb == 0) {                  // workgroup-uniform condition
        float* shb = reinterpret_cast<float*>(lds);
        constexpr int EPLc = DT<T>::EPL;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPLc; ++j) shb[tid * EPLc + j] = bsum[j];
        __syncthreads();
        if (tid < CL) {
            const int piece = tid / EPLc, j = tid % EPLc;
            float tot = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int vx = 0; vx < 16; ++vx) {
                    const int ln = vx * 4 + (piece ^ ((vx >> 1) & 3));
                    tot += shb[(wv * 64 + ln) * EPLc + j];
                }
            w.bslabs[((size_t)cb * w.P + pidx) * CL + tid] = tot;
        }
    }
    // ---- this wave's 16 x 16 tile of every tap: slab[tap][co_local][ci_local] ---------------------
    float* slab = w.slabs + (size_t)(pair * w.P + pidx) * (27 * CL * CL);
    const int gq = lane >> 4, col = lane & 15;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) slab[(t * CL + ia * 16 + gq * 4 + rg) * CL + jb * 16 + col] = acc[t][rg];
    WG_TL(5);
    WG_TL_WRITE(a);
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
// Few slabs per output (the deep levels: 8 - 32 persistent workgroups per channel-block pair, but millions of outputs): the grouped
// form above launches 27 648 blocks of two loads per thread - 1.8 TB/s.  One thread per output walks all its slabs (8 loads in flight,
// fixed order), 256 outputs per block.
__global__ void __launch_bounds__(256) wgrad_reduce_direct_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int taps,
                                                                 int cout, int cin, int ncb, int nkb, int CL, int nslabs, int accumulate) {
    const int total = taps * cout * cin;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ci = i % cin, co = (i / cin) % cout, tap = i / (cin * cout);
    const int pair = (co / CL) * nkb + ci / CL;
    const size_t slab_elems = (size_t)taps * CL * CL;
    const float* s = slabs + (size_t)pair * nslabs * slab_elems + ((size_t)tap * CL + co % CL) * CL + ci % CL;
    float sum = 0.f;
    int k = 0;
    for (; k + 7 < nslabs; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s[(size_t)(k + u) * slab_elems];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; k < nslabs; ++k) sum += s[(size_t)k * slab_elems];
    dw[i] = accumulate ? dw[i] + sum : sum;
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

// timing ablation (tools/ablate_step.sh): the slab reduce is NOT launched - weight gradients are garbage.  Says so, loudly, once.
static bool ablate_reduce() {
    static const bool on = [] {
        const bool v = getenv("VDM4CDM_ABLATE_REDUCE") != nullptr;
        if (v) fprintf(stderr, "\n*** libvdm4cdm_hip: VDM4CDM_ABLATE_REDUCE is set - weight-gradient slab reduces are SKIPPED, gradients are WRONG "
                               "(timing ablation only; unset it for any real run) ***\n\n");
        return v;
    }();
    return on;
}

// VDM4CDM_WGRAD_ROWS=0: the tap-split kernel also for the full bf16 3^3 stride-1 blocks (A/B switch)
static bool wgrad_rows_enabled() {
    static const bool on = [] { const char* e = getenv("VDM4CDM_WGRAD_ROWS"); return e == nullptr || atoi(e) != 0; }();
    return on;
}

template <typename T, int KS, int STRIDE, int UPS, int TZ, int TY, int NTA = WG<T>::NT, int NTB = WG<T>::NT>
static int launch_wgrad_cfg(WgradArgs w, float* dw, float* dbias, int accumulate, int cout, int cin, size_t ws_bytes, hipStream_t s) {
    using G = Geo<KS, STRIDE, TZ, TY>;
    constexpr int CL = WG<T>::NT * 16;
    ConvArgs& a = w.c;
    a.ntz = cdiv(a.Dz, TZ); a.nty = cdiv(a.Dy, TY); a.ntx = cdiv(a.Dx, 16);
    set_tile_divs(a);
    w.ntiles = a.N * a.ntz * a.nty * a.ntx;
#ifdef VDM_TIMELINE
    a.stamps = g_timeline_stamps;
#endif
    const int npairs = w.ncb * w.nkb;
    int P = wgrad_wgs() / npairs;             // persistent: ~2 workgroups per CU over all (cout, cin) block pairs
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
    bool launched = false;
    if constexpr (sizeof(T) == 2 && KS == 3 && STRIDE == 1 && UPS == 0 && NTA == 2 && NTB == 2) {
        if (wgrad_rows_enabled()) {
            auto kern = conv_wgrad_rows_kernel<T, TZ, TY>;
            static unsigned long long lds_done_rows = 0;
            int e = set_lds(kern, lds, lds_done_rows);
            if (e) return e;
            hipLaunchKernelGGL(kern, dim3(npairs * P), dim3(256), lds, s, w);
            launched = true;
        }
    }
    if (!launched) {
        auto kern = conv_wgrad_kernel<T, KS, STRIDE, UPS, TZ, TY, NTA, NTB>;
        static unsigned long long lds_done = 0;
        int e = set_lds(kern, lds, lds_done);
        if (e) return e;
        hipLaunchKernelGGL(kern, dim3(npairs * P), dim3(256), lds, s, w);
    }
    VDM_LAUNCH_CHECK("conv_wgrad_kernel");
    if (ablate_reduce()) return VDM_OK;
    const int total = G::TAPS * cout * cin;
    static const bool grouped_only = getenv("VDM4CDM_GROUPED_REDUCE") != nullptr;      // (same result, other launch shape)
    if (P * per_wg <= 32 && !grouped_only)
        hipLaunchKernelGGL(wgrad_reduce_direct_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s,
                           (const float*)w.slabs, dw, G::TAPS, cout, cin, w.ncb, w.nkb, CL, P * per_wg, accumulate);
    else
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
#ifdef VDM_TIMELINE
    a.stamps = g_timeline_stamps;
#endif
    a.Iz = a.Sz = a.Dz; a.Iy = a.Sy = a.Dy; a.Ix = a.Sx = a.Dx;
    a.ntz = cdiv(a.Dz, G::TZ); a.nty = cdiv(a.Dy, G::TY); a.ntx = cdiv(a.Dx, 16);
    set_tile_divs(a);
    w.ntiles = a.N * a.ntz * a.nty * a.ntx;
    const int npairs = 8 * w.ncb * w.nkb;
    int P = (WGS * wgrad_wgs() / 512) / npairs;
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
    if (ablate_reduce()) return VDM_OK;
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

int launch_dgw_reduce(const float* slabs, const float* bslabs, float* dw, float* dbias, int P, int accumulate, hipStream_t s) {
    const int total = 27 * 32 * 32;
    if (ablate_reduce()) return VDM_OK;
    if (P <= 32)
        hipLaunchKernelGGL(wgrad_reduce_direct_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, slabs, dw, 27, 32, 32, 1, 1, 32, P, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, WRED_OUT)), dim3(256), 0, s, slabs, dw, 27, 32, 32, 1, 1, 32, P, accumulate);
    VDM_LAUNCH_CHECK("wgrad_reduce_kernel(dgw)");
    if (dbias) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3(2), dim3(256), 0, s, bslabs, dbias, 32, 32, P, accumulate);
        VDM_LAUNCH_CHECK("wgrad_bias_reduce_kernel(dgw)");
    }
    return VDM_OK;
}

int launch_wgrad_any(const WgradArgs& w, float* dw, float* db, int acc, int cout, int cin, int ks, int stride, int ups, size_t ws,
                     int dtype, hipStream_t s) {
    if (dtype == VDM_F32) return launch_wgrad<float>(w, dw, db, acc, cout, cin, ks, stride, ups, ws, s);
    return launch_wgrad<bf16_t>(w, dw, db, acc, cout, cin, ks, stride, ups, ws, s);
}

}  // namespace vdm
