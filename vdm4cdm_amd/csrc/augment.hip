// augment.hip - the data path on the device: one launch turns the raw simulation cubes resident in
// HBM into a training batch - periodic crop at a shifted anchor, log10(x + alpha), normalisation, axis flips, axis permutation.
//
// Replaces, per sample and channel, the CPU DataLoader work of the reference:
//   Crop.__call__      /root/reference/src/dataset/augmentation.py:107-127   (anchor + arange(crop)) % fullsize per axis
//   LogTransform       augmentation.py:8-21        log10(img + alpha)
//   Normalize          augmentation.py:23-41       (img - mean) / std
//   Flip               augmentation.py:43-60       torch.flip(img, 1 + axes)
//   Permutate          augmentation.py:63-80       img.permute([0] + (1 + perm))
// composed in the order of AstroDataset.__getitem__ (CAMELS_3D_dataset.py:53-73): crop -> log/normalise -> flip -> permute, i.e.
//   out[i0, i1, i2] = g(raw[sim][(a_d + c_d) % S]),  c_d = flip_d ? D-1-f_d : f_d,  f[perm[k]] = i_k.
//
// Bound: HBM (4 B read + 4 B written per voxel and channel).  A permutation makes the source walk strided in the output's x, so a
// workgroup moves a 16^3 tile through LDS: it reads the tile along the SOURCE x (64-byte runs of the raw cube), applies g, scatters
// into LDS at the output position (row pitch 17: conflict-free for every permutation) and writes the tile along the OUTPUT x.
#include "common.h"

namespace vdm {

constexpr int AUG_MAX_SAMPLES = 32, AUG_MAX_CHANNELS = 4, AUG_T = 16;

struct AugArgs {
    vdm_augment_channel ch[AUG_MAX_CHANNELS];
    vdm_augment_sample s[AUG_MAX_SAMPLES];
    int S, D, nt;
};

__device__ __forceinline__ int pick3(int k, int v0, int v1, int v2) { return k == 0 ? v0 : (k == 1 ? v1 : v2); }

__global__ void __launch_bounds__(256) augment_kernel(const AugArgs a) {
    __shared__ float tile[AUG_T * AUG_T * (AUG_T + 1)];
    const int b = blockIdx.y, c = blockIdx.z;
    const vdm_augment_sample& s = a.s[b];
    const vdm_augment_channel& ch = a.ch[c];
    const int S = a.S, D = a.D;
    int t = blockIdx.x;
    const int I2 = (t % a.nt) * AUG_T; t /= a.nt;
    const int I1 = (t % a.nt) * AUG_T;
    const int I0 = (t / a.nt) * AUG_T;
    const int E0 = min(AUG_T, D - I0), E1 = min(AUG_T, D - I1), E2 = min(AUG_T, D - I2);      // tile extents in output space
    const int p0 = s.perm[0], p1 = s.perm[1], p2 = s.perm[2];
    // flip space f: f[perm[k]] = i_k  ->  origin / extent of the tile along flip-space axis d
    int F[3], EF[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        F[d] = p0 == d ? I0 : (p1 == d ? I1 : I2);
        EF[d] = p0 == d ? E0 : (p1 == d ? E1 : E2);
    }
    // crop space c_d = flip_d ? D-1-f_d : f_d: the tile starts at C_d and is walked ascending (l_d), f_local = flip ? EF-1-l : l
    int C[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) C[d] = s.flip[d] ? D - F[d] - EF[d] : F[d];
    const float* __restrict__ src = ch.field + (size_t)s.sim * S * S * S;
    const float alpha = ch.alpha, mean = ch.mean, stdv = ch.std;
    for (int e = threadIdx.x; e < AUG_T * AUG_T * AUG_T; e += 256) {
        const int l2 = e & 15, l1 = (e >> 4) & 15, l0 = e >> 8;
        if (l0 >= EF[0] || l1 >= EF[1] || l2 >= EF[2]) continue;
        const int z = (s.anchor[0] + C[0] + l0) % S, y = (s.anchor[1] + C[1] + l1) % S, x = (s.anchor[2] + C[2] + l2) % S;
        const float v = src[((size_t)z * S + y) * S + x];
        const float g = (log10f(v + alpha) - mean) / stdv;
        const int f0 = s.flip[0] ? EF[0] - 1 - l0 : l0, f1 = s.flip[1] ? EF[1] - 1 - l1 : l1, f2 = s.flip[2] ? EF[2] - 1 - l2 : l2;
        const int i0 = pick3(p0, f0, f1, f2), i1 = pick3(p1, f0, f1, f2), i2 = pick3(p2, f0, f1, f2);
        tile[(i0 * AUG_T + i1) * (AUG_T + 1) + i2] = g;
    }
    __syncthreads();
    float* __restrict__ out = ch.out + (size_t)b * D * D * D;
    for (int e = threadIdx.x; e < AUG_T * AUG_T * AUG_T; e += 256) {
        const int i2 = e & 15, i1 = (e >> 4) & 15, i0 = e >> 8;
        if (i0 >= E0 || i1 >= E1 || i2 >= E2) continue;
        out[((size_t)(I0 + i0) * D + (I1 + i1)) * D + (I2 + i2)] = tile[(i0 * AUG_T + i1) * (AUG_T + 1) + i2];
    }
}

}  // namespace vdm

using namespace vdm;

extern "C" int vdm_augment_batch(const vdm_augment_channel* host_channels, int n_channels, int fullsize, int crop,
                                 const vdm_augment_sample* host_samples, int n_samples, void* stream) {
    VDM_REQUIRE(host_channels && host_samples, "augment_batch: NULL table");
    VDM_REQUIRE(n_channels > 0 && n_channels <= AUG_MAX_CHANNELS, "augment_batch: 1..%d channels (got %d)", AUG_MAX_CHANNELS, n_channels);
    VDM_REQUIRE(n_samples > 0, "augment_batch: no samples");
    VDM_REQUIRE(fullsize > 0 && crop > 0 && crop <= fullsize && fullsize <= 1024, "augment_batch: need 0 < crop (%d) <= fullsize (%d) <= 1024",
                crop, fullsize);
    for (int c = 0; c < n_channels; ++c) {
        VDM_REQUIRE(host_channels[c].field && host_channels[c].out, "augment_batch: NULL pointer in channel %d", c);
        VDM_REQUIRE(host_channels[c].std != 0.f, "augment_batch: channel %d has std = 0", c);
    }
    for (int b = 0; b < n_samples; ++b) {
        const vdm_augment_sample& s = host_samples[b];
        VDM_REQUIRE(s.sim >= 0, "augment_batch: sample %d: negative simulation index", b);
        int seen = 0;
        for (int d = 0; d < 3; ++d) {
            VDM_REQUIRE(s.anchor[d] >= 0, "augment_batch: sample %d: negative anchor", b);
            VDM_REQUIRE(s.perm[d] >= 0 && s.perm[d] < 3, "augment_batch: sample %d: perm[%d] = %d", b, d, s.perm[d]);
            seen |= 1 << s.perm[d];
        }
        VDM_REQUIRE(seen == 7, "augment_batch: sample %d: perm is not a permutation of (0, 1, 2)", b);
    }
    AugArgs a;
    for (int c = 0; c < n_channels; ++c) a.ch[c] = host_channels[c];
    a.S = fullsize; a.D = crop; a.nt = (crop + AUG_T - 1) / AUG_T;
    for (int b0 = 0; b0 < n_samples; b0 += AUG_MAX_SAMPLES) {               // (the sample table travels in the kernel arguments)
        const int nb = n_samples - b0 < AUG_MAX_SAMPLES ? n_samples - b0 : AUG_MAX_SAMPLES;
        for (int b = 0; b < nb; ++b) a.s[b] = host_samples[b0 + b];
        AugArgs l = a;
        for (int c = 0; c < n_channels; ++c) l.ch[c].out = a.ch[c].out + (size_t)b0 * crop * crop * crop;
        hipLaunchKernelGGL(augment_kernel, dim3(a.nt * a.nt * a.nt, nb, n_channels), dim3(256), 0, (hipStream_t)stream, l);
        VDM_LAUNCH_CHECK("augment_kernel");
    }
    return VDM_OK;
}
