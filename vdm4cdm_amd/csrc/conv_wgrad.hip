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
// ROLL (rolling z window): a persistent workgroup walks up a column of tiles (fixed n, y, x; z ascending) and the input image is a ring of
// HZ = TZ + 2 slice slots: a step stages only its TZ NEW slices (the top two of the previous step are the bottom two of this one) - 39 KB and
// 56 LDS-DMA instructions per 2 x 8 x 16 step instead of 62 KB and 96 (the phase stamps of tools/wgrad_phases.py: issuing the DMAs of a tile
// takes as long as its 216 MFMAs per wave, and a leftover-voxel DMA of 128 B costs a wave as much as a 1-KiB one).  Halo slab hz of step r
// lives in slot (TZ r + hz) mod HZ: with TZ = 2 the mapping has period two, so the step body exists twice (PAR = r & 1), all LDS addresses
// still immediates.
template <int V> struct IntC { static constexpr int value = V; };
// Register tuples.  An MFMA operand is four CONSECUTIVE registers = two rows.  Pairing the output rows as (2j, 2j + 1) for every dy makes the
// B operand of dy = 1 the rows (2j + 1, 2j + 2) - overlapping the tuples of dy = 0 / 2, and the compiler assembles overlapping tuples with
// copies (measured: ~1.5 v_mov per MFMA, issued right in front of the MFMA that reads them; one wave per SIMD then needs 2.4 us for its
// 216 MFMAs instead of 1.65).  So dy = 1 pairs the OUTPUT rows as (2m + 1, 2m + 2), m = 0 .. TY/2 - 2: their halo rows (2m + 2, 2m + 3) are
// again an even pair, and the two left-over output rows (0, TY - 1) form one more k-step whose halo rows (1, TY) get a tuple of their own
// (read a second time).  All B tuples are then the disjoint even pairs P_p = rows (2p, 2p + 1), p = 0 .. TY/2, used by (dy 0, j = p),
// (dy 2, j = p - 1), (dy 1, m = p - 1) - and a pair is refilled for the next group right behind its last MFMA (no double buffer).  The dOut
// fragments exist in the three pairings (TZ x (TY/2 + TY/2 - 1 + 1) fragments, resident).  Per tile and wave: 144 + 32 transposed reads for
// 216 MFMAs (0.81 per MFMA; conv_wgrad_kernel: 1.14).
template <typename T, int TZ, int TY, int PAR>
__device__ __forceinline__ void wgrad_rows_step(f32x4 (&acc)[27], f32x4& accb, bool bias, const char* lds_in, const char* lds_do,
                                                const int (&lo_a)[1], const int (&lo_b)[3][1]) {
    using G = Geo<3, 1, TZ, TY>;
    using TF = TrFetch<T>;
    constexpr int NJ = TY / 2, NP = NJ + 1, NG = G::HZ * 3, R = G::HZ, ROWB = G::HX * 64;
    auto pair_of = [&](int grp, int p) {                    // halo rows (2p, 2p + 1) of (halo slab, dx) = grp
        const int hz = grp / 3, dx = grp % 3;
        return TF::template get<ROWB>(lds_in, lo_b[dx], (((TZ * PAR + hz) % R) * G::HY + 2 * p) * ROWB);
    };
    // (the wrap rows are also halves of P_0 and P_NJ: through laundered copies of the lane offsets the compiler cannot merge the reads -
    //  it would replace the second read by register copies into the wrap tuple)
    int lo_w[3][1];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        lo_w[dx][0] = lo_b[dx][0];
        asm volatile("" : "+v"(lo_w[dx][0]));
    }
    auto wrap_of = [&](int grp) {                           // halo rows (1, TY)
        const int hz = grp / 3, dx = grp % 3;
        return TF::template get<(TY - 1) * ROWB>(lds_in, lo_w[dx], (((TZ * PAR + hz) % R) * G::HY + 1) * ROWB);
    };
    uint4 ae[TZ][NJ], ao[TZ][NJ - 1], aw[TZ];              // dOut rows (2j, 2j + 1) | (2m + 1, 2m + 2) | (0, TY - 1)
#pragma unroll
    for (int oz = 0; oz < TZ; ++oz) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) ae[oz][j] = TF::template get<1024>(lds_do, lo_a, (oz * TY + 2 * j) * 1024);
#pragma unroll
        for (int m = 0; m < NJ - 1; ++m) ao[oz][m] = TF::template get<1024>(lds_do, lo_a, (oz * TY + 2 * m + 1) * 1024);
        aw[oz] = TF::template get<(TY - 1) * 1024>(lds_do, lo_a, oz * TY * 1024);
    }
    uint4 bp[NP], bw;
#pragma unroll
    for (int p = 0; p < NP; ++p) bp[p] = pair_of(0, p);
    bw = wrap_of(0);
    if (bias) {          // bias gradient = column sums of the dOut tile: dOut^T x ones on the matrix pipe (every result column holds the sum)
        const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);      // bf16 1.0 pairs
#pragma unroll
        for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
            for (int j = 0; j < NJ; ++j) mma16_act<T>(accb, ae[oz][j], ones);
    }
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        const int hz = grp / 3, dx = grp % 3;
#pragma unroll
        for (int p = 0; p <= NP; ++p) {                     // p == NP: the wrap k-step
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                const int oz = hz - dz;
                if (oz < 0 || oz >= TZ) continue;
                const int t = dz * 9 + dx;                  // tap index = (dz * 3 + dy) * 3 + dx
                if (p == NP) {
                    mma16_act<T>(acc[t + 3], aw[oz], bw);
                } else {
                    if (p < NJ) mma16_act<T>(acc[t], ae[oz][p], bp[p]);                         // dy = 0, j = p
                    if (p >= 1 && p <= NJ - 1) mma16_act<T>(acc[t + 3], ao[oz][p - 1], bp[p]);  // dy = 1, m = p - 1
                    if (p >= 1) mma16_act<T>(acc[t + 6], ae[oz][p - 1], bp[p]);                 // dy = 2, j = p - 1
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (grp + 1 < NG) {                             // the tuple is dead: the next group's rows take its place
                if (p == NP) bw = wrap_of(grp + 1);
                else bp[p] = pair_of(grp + 1, p);
            }
        }
    }
}

template <typename T, int TZ, int TY, bool ROLL>
__global__ void __launch_bounds__(256, 2) conv_wgrad_rows_kernel(const WgradArgs w) {
    using G = Geo<3, 1, TZ, TY>;
    using TF = TrFetch<T>;
    static_assert(sizeof(T) == 2 && (TY % 2) == 0, "bf16 only: a fragment = two rows of the same z-slab");
    static_assert(!ROLL || TZ == 2, "the ring mapping has period two for TZ = 2");
    constexpr int IN_BYTES = ((G::HVOX + 15) / 16) * 1024;
    constexpr int CL = 32, R = G::HZ;
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
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};                 // bias gradient of this wave's 16 couts (waves with cin tile 0 of cin block 0)
    const bool bias = w.bslabs != nullptr && kb == 0 && jb == 0;
    WG_TL_DECL;
    if constexpr (ROLL) {
        // w.ntiles = column segments (n, ty, tx, seg); a.fdz divides by a.nseg
        for (int cs = pidx; cs < w.ntiles; cs += w.P) {
            uint32_t b = (uint32_t)cs;
            uint32_t q = fdiv(b, a.fdx);
            const int tx = (int)(b - q * (uint32_t)a.ntx); b = q;
            q = fdiv(b, a.fdy);
            const int ty = (int)(b - q * (uint32_t)a.nty); b = q;
            q = fdiv(b, a.fdz);
            const int seg = (int)(b - q * (uint32_t)a.nseg);
            const int n = (int)q;
            const int zs0 = seg * a.zsteps, zs1 = min(a.ntz, zs0 + a.zsteps);
            const int oy0 = ty * TY, ox0 = tx * 16;
            const RowStager<T, G, 0> st(x, a, n, 0, oy0, ox0, kb, lane, 1, 0, 0, 0, a.Sz, a.Sy, a.Sx);      // (iz0 = -1; x part once per column)
            // slices are addressed by their position p = iz + 1 - TZ zs0 >= 0 in the walk; slot = p mod R
            auto step = [&](int s, auto parc) {
                const int rel = s - zs0;
                const int p0 = rel == 0 ? 0 : TZ * rel + (R - TZ), cnt = rel == 0 ? R : TZ;
                __syncthreads();                           // every wave is done reading the previous step
                WG_TL(0);
                for (int r = wave; r < cnt * G::HY; r += 4) {
                    const int sl = r / G::HY, hy = r % G::HY, p = p0 + sl;
                    st.row(lds_in + ((p % R) * G::HY + hy) * (G::HX * 64), a, TZ * zs0 + p, hy);
                }
                stage_dout_dma<T, G>(lds_do, g, a, n, s * TZ, oy0, ox0, cb, w.dout_stride, wave, lane);
                WG_TL(1);
                __syncthreads();                           // (drains the LDS-DMA: vmcnt(0) + barrier)
                WG_TL(2);
                wgrad_rows_step<T, TZ, TY, decltype(parc)::value>(acc, accb, bias, lds_in, lds_do, lo_a, lo_b);
                WG_TL(3);
            };
            // (two steps per iteration: the slot mapping has period two.  Staging step s + 1 from inside step s - its new slices go to slots
            //  that are free after the first half of the groups - was built and measured: the DMAs issued between the MFMA groups block the
            //  wave for ~0.28 us each, the step got longer: 6.7 us against 1.7 + 0.6 + 3.3 us)
            for (int s = zs0; s < zs1; s += 2) {
                step(s, IntC<0>());
                if (s + 1 < zs1) step(s + 1, IntC<1>());
            }
        }
    } else {
        for (int tile = pidx; tile < w.ntiles; tile += w.P) {
            int tx, ty, tz, n, rest_;
            decode_tile(a, (uint32_t)tile, tx, ty, tz, n, rest_);
            const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * 16;
            __syncthreads();                               // every wave is done reading the previous tile
            WG_TL(0);
            stage_halo_dma<T, G, 0>(lds_in, x, a, n, oz0, oy0, ox0, kb, wave, lane);
            stage_dout_dma<T, G>(lds_do, g, a, n, oz0, oy0, ox0, cb, w.dout_stride, wave, lane);
            WG_TL(1);
            __syncthreads();                               // (drains the LDS-DMA: vmcnt(0) + barrier)
            WG_TL(2);
            wgrad_rows_step<T, TZ, TY, 0>(acc, accb, bias, lds_in, lds_do, lo_a, lo_b);
            WG_TL(3);
        }
    }

    if (bias && (lane & 15) == 0) {                         // (result column 0: rows = couts ia * 16 + 4 (lane >> 4) ...)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) w.bslabs[((size_t)cb * w.P + pidx) * CL + ia * 16 + (lane >> 4) * 4 + rg] = accb[rg];
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
            // rolling z window (VDM4CDM_WGRAD_ROLL=0: off): the persistent workgroups walk column segments instead of scattered tiles
            static const bool roll_on = [] { const char* e = getenv("VDM4CDM_WGRAD_ROLL"); return e == nullptr || atoi(e) != 0; }();
            const int ncols = a.N * a.nty * a.ntx;
            int nseg = roll_on && TZ == 2 ? P / ncols : 0;               // P = the workgroups this pair may use (slab space is sized for it)
            if (nseg > a.ntz / 2) nseg = a.ntz / 2;                      // >= 2 steps per segment, or the walk saves nothing
            if (nseg >= 1 || (roll_on && TZ == 2 && a.ntz >= 4)) {
                if (nseg < 1) nseg = 1;                                  // more columns than workgroups: a workgroup walks several
                a.zsteps = cdiv(a.ntz, nseg);
                a.nseg = cdiv(a.ntz, a.zsteps);
                a.fdz = make_fastdiv((uint32_t)a.nseg);
                w.ntiles = ncols * a.nseg;
                if (P > w.ntiles) P = w.ntiles;
                w.P = P;
                auto kern = conv_wgrad_rows_kernel<T, TZ, TY, (TZ == 2)>;
                static unsigned long long lds_done_roll = 0;
                int e = set_lds(kern, lds, lds_done_roll);
                if (e) return e;
                hipLaunchKernelGGL(kern, dim3(npairs * P), dim3(256), lds, s, w);
            } else {
                auto kern = conv_wgrad_rows_kernel<T, TZ, TY, false>;
                static unsigned long long lds_done_rows = 0;
                int e = set_lds(kern, lds, lds_done_rows);
                if (e) return e;
                hipLaunchKernelGGL(kern, dim3(npairs * P), dim3(256), lds, s, w);
            }
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
    if (stride == 2) {
        // 1x4x16 output tiles (57-KB halo, two workgroups per CU; default since round 4: alone 0.100 -> 0.091 / 0.060 -> 0.048 / 0.038 ->
        // 0.034 ms at levels 0 / 1 / 2, step unchanged).  VDM4CDM_S2W_TZ=2: 2x4x16 tiles (95-KB halo, -17 % staged bytes, one workgroup per
        // CU - and none next to a main-stream kernel that holds 2 x 70 KB of the CU's LDS)
        static const int tz = [] { const char* e = getenv("VDM4CDM_S2W_TZ"); return e ? atoi(e) : 1; }();
        if (tz == 1) return launch_wgrad_cfg<T, 3, 2, 0, 1, 4>(w, dw, db, acc, cout, cin, ws, s);
        return launch_wgrad_cfg<T, 3, 2, 0, 2, 4>(w, dw, db, acc, cout, cin, ws, s);
    }
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
