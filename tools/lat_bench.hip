// Dependent-chain latency of the instructions on the arithmetic coder's critical path (one wave, gfx950).
// Development aid: hipcc --offload-arch=gfx950 -O3 -o /tmp/lat_bench tools/lat_bench.hip && /tmp/lat_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 256
#define CHAIN(name, decl, body)                                                        \
    __global__ void name(uint64_t *out, uint32_t seed, double dseed) {                 \
        decl;                                                                          \
        uint64_t t0 = __builtin_readcyclecounter();                                    \
        for (int it = 0; it < 64; ++it) {                                              \
            _Pragma("unroll") for (int i = 0; i < REP; ++i) { body; }                  \
        }                                                                              \
        uint64_t t1 = __builtin_readcyclecounter();                                    \
        if (threadIdx.x == 0) { out[0] = t1 - t0; }                                    \
        sink(out);                                                                     \
    }

#define SINK32(v) __device__ __forceinline__ void sink_(uint64_t *o, uint32_t v) { if (v == 0x12345) o[1] = v; }

CHAIN(k_add_u32, uint32_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(seed)))
CHAIN(k_lshl_b32, uint32_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(seed & 1)))
CHAIN(k_ffbh, uint32_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_ffbh_u32 %0, %0" : "+v"(x)))
CHAIN(k_bitop3, uint32_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0xf3" : "+v"(x) : "v"(seed)))
CHAIN(k_cmp_cnd, uint32_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_cmp_ne_u32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(x) : "v"(seed) : "vcc"))
CHAIN(k_lshl_b64, uint64_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(x) : "v"(seed & 1)))
CHAIN(k_fma_f64, double x = dseed; auto sink = [&](uint64_t *o) { if (x == 12345.0) o[1] = 1; },
      asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(dseed)))
CHAIN(k_add_f64, double x = dseed; auto sink = [&](uint64_t *o) { if (x == 12345.0) o[1] = 1; },
      asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(dseed)))
CHAIN(k_mul_f64, double x = dseed; auto sink = [&](uint64_t *o) { if (x == 12345.0) o[1] = 1; },
      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(dseed)))
CHAIN(k_cvt_rt, uint32_t x = seed; double d; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_cvt_f64_u32 %1, %0\n\tv_cvt_u32_f64 %0, %1" : "+v"(x), "=&v"(d)))
CHAIN(k_fma_f32, float x = (float)dseed; auto sink = [&](uint64_t *o) { if (x == 12345.0f) o[1] = 1; },
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"((float)dseed)))
CHAIN(k_mul_lo, uint32_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(seed)))
CHAIN(k_mul_hi, uint32_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(seed)))
CHAIN(k_rcp_f64, double x = dseed; auto sink = [&](uint64_t *o) { if (x == 12345.0) o[1] = 1; },
      asm volatile("v_rcp_f64 %0, %0" : "+v"(x)))
CHAIN(k_mad_u64, uint64_t x = seed; auto sink = [&](uint64_t *o) { if (x == 12345) o[1] = x; },
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x) : "v"(seed) : "vcc"))
CHAIN(k_indep_add, uint32_t x = seed; uint32_t y = seed + 1; uint32_t z = seed + 2; uint32_t w = seed + 3; auto sink = [&](uint64_t *o) { if (x + y + z + w == 12345) o[1] = x; },
      asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(seed)))
CHAIN(k_interval, uint32_t l = seed; uint32_t nu = seed * 3; double a = dseed * 1e-9; double c = dseed * 2e-9; auto sink = [&](uint64_t *o) { if (l + nu == 12345) o[1] = l; },
      {
          const double rd = 4294967296.0 - (double)(l + nu);
          const uint32_t ql = (uint32_t)__builtin_fma(rd, a, 5.9604644775390625e-08);
          const uint32_t qc = (uint32_t)__builtin_fma(rd, c, 0.99999994039535522461);
          l += ql << 10; nu += qc << 10;
          const uint32_t k = (uint32_t)__builtin_clz(~(l ^ nu) | 1024);
          l <<= k; nu <<= k;
          const uint32_t j = (uint32_t)__builtin_clz((~(l & nu) << 1) | 1024);
          l = (l << j) & 0x7FFFFFFFu; nu = (nu << j) & 0x7FFFFFFFu;
          asm volatile("" : "+v"(l), "+v"(nu));
      })
__global__ void k_lds(uint64_t *out, uint32_t seed) {
    __shared__ uint32_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = (i * 7 + seed) & 1023;
    __syncthreads();
    uint32_t x = threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64 * REP; ++it) x = buf[x];
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (x == 12345) out[1] = x;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    uint64_t *d, h[2];
    printf("start\n");
    hipMalloc(&d, 16);
#define RUN(k, per, ...)                                                                             \
    do {                                                                                             \
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, __VA_ARGS__);                              \
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, __VA_ARGS__);                              \
        hipDeviceSynchronize();                                                                      \
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);                                                  \
        printf("%-14s %7.2f cycles per step (%d instr)\n", #k, (double)h[0] / (64.0 * REP), per);    \
    } while (0)
    RUN(k_add_u32, 1, 3u, 1.5);
    RUN(k_indep_add, 4, 3u, 1.5);
    RUN(k_lshl_b32, 1, 3u, 1.5);
    RUN(k_ffbh, 1, 3u, 1.5);
    RUN(k_bitop3, 1, 3u, 1.5);
    RUN(k_cmp_cnd, 3, 3u, 1.5);
    RUN(k_lshl_b64, 1, 3u, 1.5);
    RUN(k_mad_u64, 1, 3u, 1.5);
    RUN(k_mul_lo, 1, 3u, 1.5);
    RUN(k_mul_hi, 1, 3u, 1.5);
    RUN(k_fma_f32, 1, 3u, 1.0000001);
    RUN(k_fma_f64, 1, 3u, 1.0000001);
    RUN(k_add_f64, 1, 3u, 1.0000001);
    RUN(k_mul_f64, 1, 3u, 1.0000001);
    RUN(k_cvt_rt, 2, 3u, 1.5);
    RUN(k_rcp_f64, 1, 3u, 1.5);
    RUN(k_interval, 24, 3u, 1.5);
    // partially filled waves: does a wave with 32 or 16 live lanes issue dependent instructions faster?
#define RUNW(k, per, w, ...)                                                                         \
    do {                                                                                             \
        hipLaunchKernelGGL(k, dim3(1), dim3(w), 0, 0, d, __VA_ARGS__);                               \
        hipLaunchKernelGGL(k, dim3(1), dim3(w), 0, 0, d, __VA_ARGS__);                               \
        hipDeviceSynchronize();                                                                      \
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);                                                  \
        printf("%-14s %2d lanes %7.2f cycles per step (%d instr)\n", #k, w, (double)h[0] / (64.0 * REP), per); \
    } while (0)
    RUNW(k_add_u32, 1, 32, 3u, 1.5);
    RUNW(k_add_u32, 1, 16, 3u, 1.5);
    RUNW(k_fma_f64, 1, 32, 3u, 1.0000001);
    RUNW(k_fma_f64, 1, 16, 3u, 1.0000001);
    RUNW(k_interval, 24, 32, 3u, 1.5);
    RUNW(k_interval, 24, 16, 3u, 1.5);
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d, 3u);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-14s %7.2f cycles per dependent ds_read_b32\n", "k_lds", (double)h[0] / (64.0 * REP));
    return 0;
}
