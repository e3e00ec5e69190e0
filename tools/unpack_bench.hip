// Throughput of the coder loaders' record unpacking (qvz_coder_gpu.hip, unpack_records4) on one wave and on two waves that share a
// SIMD (five waves in a workgroup: the fifth lands on the first one's SIMD), gfx950. Development aid:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/unpack_bench tools/unpack_bench.hip && /tmp/unpack_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int NEWTON, bool F32SEED>
__device__ __forceinline__ void unpack4(const uint4 &a, const uint4 &b, double2 *out) {
    const uint32_t lo[4] = {a.x, a.z, b.x, b.z}, hi[4] = {a.y, a.w, b.y, b.w};
    double nd[4], rn[4], e[4], cd[4], qd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) nd[i] = (double)(hi[i] & 0xFFFFFC00u);
#pragma unroll
    for (int i = 0; i < 4; ++i) rn[i] = F32SEED ? (double)__builtin_amdgcn_rcpf((float)(hi[i] & 0xFFFFFC00u)) : __builtin_amdgcn_rcp(nd[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) cd[i] = (double)(lo[i] & 0x1FFFFFu);
#pragma unroll
    for (int i = 0; i < 4; ++i) qd[i] = (double)(__builtin_amdgcn_alignbit(hi[i], lo[i], 21) & 0x1FFFFFu);
#pragma unroll
    for (int k = 0; k < NEWTON; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = __builtin_fma(-nd[i], rn[i], 1.0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rn[i] = __builtin_fma(e[i], rn[i], rn[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = make_double2(cd[i] * rn[i], qd[i] * rn[i]);
}

template <int NEWTON, bool F32SEED>
__global__ void k_bench(uint64_t *out, uint4 seed, int active_hw_waves_mask) {
    const int wave = threadIdx.x >> 6;
    if (!((active_hw_waves_mask >> wave) & 1)) return;
    uint4 a = seed, b = seed;
    a.x += threadIdx.x; b.z += threadIdx.x * 3;
    double acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 4096; ++it) {
        double2 r[4];
        unpack4<NEWTON, F32SEED>(a, b, r);
        acc += r[0].x + r[1].y + r[2].x + r[3].y;
        a.x += 0x200000; a.y += 1024; b.x += 3; b.y += 2048; a.w += 4096; b.w += 1024; a.z += 5; b.z += 7;
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (acc == 1.2345) out[15] = 1;
    uint32_t hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((threadIdx.x & 63) == 0) out[8 + wave] = (hwid >> 4) & 3;
}

// accuracy of the refined reciprocal over every n the model can put into a record: max relative error of rn against 1 / (n * 2^10)
template <int NEWTON>
__global__ void k_acc(double *worst, uint32_t nmax) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x + 1;
    if (n > nmax) return;
    const double nd = (double)(n << 10);
    double rn = __builtin_amdgcn_rcp(nd);
    for (int k = 0; k < NEWTON; ++k) { const double e = __builtin_fma(-nd, rn, 1.0); rn = __builtin_fma(e, rn, rn); }
    // residual 1 - nd * rn evaluated exactly by fma: |residual| = relative error of rn
    const double res = __builtin_fabs(__builtin_fma(-nd, rn, 1.0));
    // atomic max on the bit pattern (positive doubles order like integers)
    atomicMax(reinterpret_cast<unsigned long long *>(worst), (unsigned long long)__double_as_longlong(res));
}

int main() {
    uint64_t *d, h[16];
    (void)hipMalloc(&d, sizeof(h));
    const uint4 seed = make_uint4(0x00345678u, 0x00ABC400u, 0x00111111u, 0x00DEF800u);
    auto run = [&](const char *name, auto kern, int threads, int mask) {
        (void)hipMemset(d, 0, sizeof(h));
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, seed, mask);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, seed, mask);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-46s", name);
        for (int w = 0; w < 5; ++w) if ((mask >> w) & 1) printf("  wave %d (SIMD %llu): %.1f cycles per 4 records", w, (unsigned long long)h[8 + w], h[w] / 4096.0);
        printf("\n");
    };
    run("two Newton steps, one wave", k_bench<2, false>, 64, 1);
    run("two Newton steps, waves 0 and 4 of five", k_bench<2, false>, 320, 0x11);
    run("one Newton step, one wave", k_bench<1, false>, 64, 1);
    run("one Newton step, waves 0 and 4 of five", k_bench<1, false>, 320, 0x11);
    run("f32 seed + two Newton steps, one wave", k_bench<2, true>, 64, 1);
    run("f32 seed + two Newton, waves 0 and 4 of five", k_bench<2, true>, 320, 0x11);
    double *dw, hw;
    (void)hipMalloc(&dw, 8);
    const uint32_t nmax = (1u << 19) + 1024;
    for (int nw = 0; nw < 3; ++nw) {
        (void)hipMemset(dw, 0, 8);
        if (nw == 0) hipLaunchKernelGGL(k_acc<0>, dim3((nmax + 255) / 256), dim3(256), 0, 0, dw, nmax);
        if (nw == 1) hipLaunchKernelGGL(k_acc<1>, dim3((nmax + 255) / 256), dim3(256), 0, 0, dw, nmax);
        if (nw == 2) hipLaunchKernelGGL(k_acc<2>, dim3((nmax + 255) / 256), dim3(256), 0, 0, dw, nmax);
        (void)hipMemcpy(&hw, dw, 8, hipMemcpyDeviceToHost);
        printf("v_rcp_f64 + %d Newton step(s): worst relative error over n = 1 .. %u: %.3g = 2^%.1f\n", nw, nmax, hw, log2(hw > 0 ? hw : 1e-300));
    }
    return 0;
}
