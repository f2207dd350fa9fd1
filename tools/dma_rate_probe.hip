// dma_rate_probe.hip - how fast can ONE CU pull HBM data into LDS with LDS-DMA (global_load_lds_dwordx4), as a function of the
// number of issuing waves and of the pieces each keeps in flight?  Every CU streams a disjoint part of a 4 GiB buffer (all misses).
//     hipcc -O3 --offload-arch=gfx950 tools/dma_rate_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
// Output: GB/s per CU and chip-wide.  Context: a conv workgroup stages a 69.6 KB halo tile with 4 waves x 17 pieces.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DEPTH>
__global__ void __launch_bounds__(512) stream_kernel(const uint4* __restrict__ buf, size_t per_wave_uint4, int pieces, int rowmode) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    const uint4* p = buf + ((size_t)blockIdx.x * nw + wave) * per_wave_uint4;
    char* dst = lds + wave * 16 * 1024;
    // rowmode 0: each piece = 1 KiB contiguous, pieces back to back; 1: 16 segments of 64 B, 4 KiB apart;
    //         2: 1 KiB contiguous pieces 8 KiB apart (the y rows of a 128^3 x 32-channel bf16 tensor: what a conv halo tile reads);
    //         3: as 2 but every piece starts 64 B before a 1 KiB boundary (halo origin x0 - 1);  4: as 3 with the 4 lanes of a voxel
    //         fetching its 16-B pieces in the x-swizzled order of the conv image
    size_t lane_off = (size_t)lane, step = 64;
    if (rowmode == 1) { lane_off = (size_t)(lane >> 2) * 256 + (lane & 3); step = 16 * 256; }
    if (rowmode >= 2) step = 512;                           // 8 KiB in uint4
    if (rowmode >= 3) lane_off += 60;                       // + 1 KiB - 64 B
    if (rowmode == 4) lane_off = (size_t)(lane & ~3) + ((lane & 3) ^ (((lane >> 2) >> 1) & 3)) + 60;
    for (int i = 0; i < pieces; i += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)(i + k) * step + lane_off),
                                             (__attribute__((address_space(3))) void*)(dst + (k & 15) * 1024), 16, 0, 0);
        if (DEPTH >= 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DEPTH>
static void run(const uint4* buf, size_t bytes, int ncu, int waves, int rowmode) {
    const size_t per_wave = bytes / 16 / ((size_t)ncu * waves);
    const size_t stepu = rowmode == 0 ? 64 : (rowmode == 1 ? 16 * 256 : 512);
    const int pieces = (int)((per_wave / stepu - 2) / DEPTH * DEPTH);
    const int use = pieces > 4096 ? 4096 : pieces;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(ncu), dim3(64 * waves), waves * 16 * 1024, 0, buf, per_wave, use, rowmode);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double gb = (double)ncu * waves * use * 1024.0 / 1e9;
    const char* names[5] = {"linear     ", "seg64B@4K  ", "rows@8K    ", "rows@8K-64B", "rows-64B+swz"};
    printf("  %s waves/CU %d  pieces in flight/wave %2d : %7.1f GB/s per CU  %6.2f TB/s chip   (%.0f us)\n", names[rowmode], waves, DEPTH,
           gb / (best * 1e-3) / ncu, gb / (best * 1e-3) / 1e3, best * 1e3);
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    uint4* buf;
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 1, bytes));
    for (int rowmode = 0; rowmode < 5; ++rowmode)
        for (int waves : {1, 4, 8}) {
            run<8>(buf, bytes, ncu, waves, rowmode);
            run<16>(buf, bytes, ncu, waves, rowmode);
        }
    return 0;
}
