// tcp_order_probe.hip - does a CU's vector L1 (TCP) return an L2-HIT load of one wave while HBM-MISS loads of another wave are in
// flight, or do hits queue behind the misses (in-order return across waves)?
//
// One workgroup per CU, 2 waves.  Wave 0 ("streamer") issues back-to-back LDS-DMA loads (global_load_lds_dwordx4, 1 KiB each)
// that sweep a buffer far larger than the Infinity Cache (every load misses to HBM) - exactly what a conv workgroup does while it
// stages a halo tile.  Wave 1 ("prober") repeatedly loads the same 4 KiB table (L1/L2-resident: what the tap loop does for its
// packed weights) and measures the round-trip latency of each load with s_memtime.  Run with the streamer off and on:
//     hipcc -O3 --offload-arch=gfx950 tools/tcp_order_probe.hip -o /tmp/tcp_probe && /tmp/tcp_probe
// If the hit latency jumps from the L2-hit figure (~0.1-0.25 us) to the HBM figure (~1+ us) with the streamer on, hits wait behind
// the misses of the OTHER wave: a conv's weight loads stall whenever the co-resident workgroup stages.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int PROBES = 512;

__global__ void __launch_bounds__(128) probe_kernel(const uint4* __restrict__ stream_buf, size_t stream_uint4, const uint4* __restrict__ table,
                                                   int stream_on, int stream_depth, unsigned* __restrict__ lat, unsigned long long* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave == 0) {
        if (!stream_on) return;
        // stream: each iteration issues `stream_depth` DMA pieces, then waits for all but the last few (keeps the queue full)
        size_t pos = ((size_t)blockIdx.x * 7919u * 1024u) % (stream_uint4 - 64 * 64);
        for (int it = 0; it < 200000; ++it) {
            for (int k = 0; k < stream_depth; ++k) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream_buf + pos + lane),
                                                 (__attribute__((address_space(3))) void*)(lds + (k & 15) * 1024), 16, 0, 0);
                pos += 64 * 257;                             // (stride: a new DRAM page almost every piece)
                if (pos >= stream_uint4 - 64) pos -= stream_uint4 - 64;
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (*(volatile int*)&done) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // prober
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int w = 0; w < 8; ++w) {                            // warm the table
        const uint4 v = table[lane + 64 * (w & 3)];
        acc.x ^= v.x;
    }
    __builtin_amdgcn_s_sleep(100);
    for (int p = 0; p < PROBES; ++p) {
        unsigned long long t0, t1;
        uint4 v;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(table + lane + 64 * (p & 3)) : "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        acc.x ^= v.x; acc.y ^= v.y;
        if (lane == 0) lat[(size_t)blockIdx.x * PROBES + p] = (unsigned)(t1 - t0);
        __builtin_amdgcn_s_sleep(20);
    }
    if (lane == 0) *(volatile int*)&done = 1;
    if (acc.x == 0x12345678u) sink[0] = acc.y;
}

int main() {
    const size_t stream_bytes = (size_t)2 << 30;             // 2 GiB >> 256 MiB Infinity Cache
    uint4 *buf, *table;
    unsigned* lat;
    unsigned long long* sink;
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    CHECK(hipMalloc(&buf, stream_bytes));
    CHECK(hipMemset(buf, 1, stream_bytes));
    CHECK(hipMalloc(&table, 4096));
    CHECK(hipMemset(table, 2, 4096));
    CHECK(hipMalloc(&lat, sizeof(unsigned) * ncu * PROBES));
    CHECK(hipMalloc(&sink, 8));
    std::vector<unsigned> h(ncu * PROBES);
    for (int depth : {0, 4, 16}) {
        const int on = depth > 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe_kernel, dim3(ncu), dim3(128), 16 * 1024, 0, buf, stream_bytes / 16, table, on, on ? depth : 1, lat, sink);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), lat, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost));
        std::vector<unsigned> v;
        for (int b = 0; b < ncu; ++b)
            for (int p = 32; p < PROBES; ++p) v.push_back(h[(size_t)b * PROBES + p]);
        std::sort(v.begin(), v.end());
        printf("streamer %s (pieces per burst %2d): L2-hit load latency of the OTHER wave  p10 %5u  p50 %5u  p90 %5u  p99 %5u  cycles (s_memtime)\n",
               on ? "ON " : "off", depth, v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100]);
    }
    return 0;
}
