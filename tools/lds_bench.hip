// What same-address LDS traffic costs on gfx950: cycles per wave instruction for ds_or_b64 / ds_add_u32 / ds_write_b32 / ds_read_b32 / ds_write_b64 / ds_or_b32
// when the 64 lanes address G distinct entries (G = 64: all different ... G = 1: all the same), one wave on a CU.
// Development aid for the model kernel's mask / count updates (qvz_coder_gpu.hip):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bench tools/lds_bench.hip && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
template <int OP>
__global__ __launch_bounds__(64) void k(uint64_t *out, uint32_t groups, uint32_t stride_words) {
    __shared__ unsigned long long m[2048];
    uint32_t *w = reinterpret_cast<uint32_t *>(m);
    for (int i = threadIdx.x; i < 2048; i += 64) m[i] = 0;
    __syncthreads();
    const uint32_t g = threadIdx.x % groups;                     // lanes with the same g address the same entry
    const uint32_t a32 = g * stride_words, a64 = g * stride_words;
    unsigned long long acc = 0;
    uint32_t acc2 = 0;
    const unsigned long long bit = 1ull << threadIdx.x;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; ++it) {
#pragma unroll
        for (int i = 0; i < REP; ++i) {
            if (OP == 0) __hip_atomic_fetch_or(&m[a64], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == 1) atomicAdd(&w[a32], 8u);
            if (OP == 2) asm volatile("ds_write_b32 %0, %1" : : "v"(a32 * 4), "v"(g) : "memory");
            if (OP == 3) { uint32_t v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a32 * 4) : "memory"); acc2 ^= v; }
            if (OP == 4) asm volatile("ds_write_b64 %0, %1" : : "v"(a64 * 8), "v"(bit) : "memory");
            if (OP == 5) __hip_atomic_fetch_or(&w[a32], (uint32_t)bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 0x1234567 || acc2 == 0x7654321) out[1] = acc + m[5];
}

int main() {
    uint64_t *d, h[2];
    hipMalloc(&d, 16);
    const char *names[] = {"ds_or_b64 (atomic)", "ds_add_u32 (atomic)", "ds_write_b32", "ds_read_b32", "ds_write_b64", "ds_or_b32 (atomic)"};
    for (int op = 0; op < 6; ++op)
        for (uint32_t stride : {1u, 3u}) {
            printf("%-20s entry stride %u words:", names[op], stride * (op == 0 || op == 4 ? 2 : 1));
            for (uint32_t groups : {64u, 32u, 16u, 8u, 4u, 2u, 1u}) {
                for (int rep = 0; rep < 2; ++rep) {
                    switch (op) {
                        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, groups, stride); break;
                        case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, groups, stride); break;
                        case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, groups, stride); break;
                        case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, groups, stride); break;
                        case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, groups, stride); break;
                        default: hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, d, groups, stride); break;
                    }
                    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                }
                printf("  G=%-2u %5.1f", groups, (double)h[0] / (16.0 * REP));
            }
            printf("   cycles per instruction\n");
        }
    return 0;
}
