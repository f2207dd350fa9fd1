// Calibration: attainable MFMA rate (v_mfma_f32_16x16x32_bf16, register operands only) and shader clock under that load.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, unsigned long long* clk) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)(float)(threadIdx.x & 1); }
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NACC>
void run(int wg_per_cu, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flop = (double)cus * wg_per_cu * 4 * iters * NACC * 16384.0;
    printf("NACC=%2d waves/SIMD=%d : %.3f ms  %.0f TFLOP/s   shader clock %.0f MHz (cycles %llu / realtime ticks %llu @100MHz), cycles per MFMA per wave %.2f\n",
           NACC, wg_per_cu, ms, flop / ms / 1e9, (double)h[0] / ((double)h[1] / 100.0), h[0], h[1], (double)h[0] / ((double)iters * NACC));
}

int main() {
    run<16>(1, 20000);
    run<16>(2, 20000);
    run<4>(1, 80000);
    run<8>(1, 40000);
    run<16>(1, 200000);
    return 0;
}
