// LDS atomic throughput on one CU: 16 waves issue N ds_add_u32 each with a given address pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, int RTN>
__global__ __launch_bounds__(1024) void k(uint64_t *out, uint32_t iters, uint32_t seed) {
    extern __shared__ uint32_t t[];
    const uint32_t words = 28672;   // 112 KB
    for (uint32_t i = threadIdx.x; i < words; i += 1024) t[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t x = seed + threadIdx.x * 2654435761u, acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 5u + 12345u + u;                                        // cheap per-lane pseudo-random walk
            uint32_t a;
            if (MODE == 0) a = ((x >> 3) & 511) * 32 + (lane & 31);          // bank-private, random row
            else if (MODE == 1) a = (x >> 3) & 8191;                        // random over 8192 dwords
            else if (MODE == 2) a = lane + 64 * u;                           // perfectly linear
            else if (MODE == 3) a = ((x >> 3) & 1023) * 16 + (lane & 15);    // 16 replicas
            else if (MODE == 5) a = ((x >> 3) & 63) * 5 * 0 + ((x >> 3) & 63);     // 64 hot dwords: same-address collisions
            else a = ((x >> 12) & 7) * 32 + (lane & 31);                     // bank-private, 8 hot rows
            if (RTN) acc += atomicAdd(&t[a], 1u); else atomicAdd(&t[a], 1u);
        }
    }
    __syncthreads();
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678) out[1000] = acc;
}
template <int MODE, int RTN> void run(const char *name, int blocks) {
    uint64_t *d; hipMalloc(&d, 8 * 2048);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE, RTN>), hipFuncAttributeMaxDynamicSharedMemorySize, 114688);
    const uint32_t iters = 2000;
    hipLaunchKernelGGL((k<MODE, RTN>), dim3(blocks), dim3(1024), 114688, 0, d, iters, 12345u);
    hipDeviceSynchronize();
    uint64_t h[1]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    // s_memtime-like counter: report raw counts per wave-instruction per CU (16 waves x iters x 8 instructions)
    printf("%-44s rtn=%d: %.2f counter ticks per wave-atomic (CU-wide)\n", name, RTN, (double)h[0] / (16.0 * iters * 8));
    hipFree(d);
}
int main() {
    run<2, 0>("linear", 1); run<0, 0>("bank-private random row", 1); run<4, 0>("bank-private 8 hot rows", 1);
    run<1, 0>("random over 8820 dwords", 1); run<3, 0>("16 replicas", 1);
    run<5, 0>("64 hot dwords", 1); run<0, 1>("bank-private random row", 1); run<1, 1>("random over 8820 dwords", 1);
    return 0;
}
