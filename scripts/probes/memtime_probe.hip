// What does s_memtime count on gfx950?  (round 4: the phase probes of round 3 read "1.58 G ticks per second" off it while
// sysfs reports sclk 2.35 GHz under the same load.)  One wave runs a fixed chain of dependent v_fma_f32 and a block of
// MFMAs between two stamps of BOTH counters - s_memtime and s_memrealtime (the constant 100 MHz reference clock) - so
//     ticks of s_memtime per second = d(memtime) / d(memrealtime) x 100 MHz
// independent of any host timer; the same measured idle, and beside a load kernel that keeps every CU's matrix pipe busy
// (second stream).  hipcc --offload-arch=gfx950 -O3 memtime_probe.hip -o memtime_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void stamp_kernel(unsigned long long* out, float* sink, int n_fma, int n_mfma) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 u, v;
    for (int e = 0; e < 8; ++e) { u[e] = (__bf16)(0.5f + threadIdx.x); v[e] = (__bf16)0.25f; }
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n_fma; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));       // dependent chain
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n_mfma; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, acc, 0, 0, 0);  // dependent chain
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    asm volatile("" : "+v"(s));
    const unsigned long long m2 = __builtin_amdgcn_s_memtime();
    const unsigned long long r2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = m1 - m0; out[1] = r1 - r0; out[2] = m2 - m1; out[3] = r2 - r1;
    }
    sink[threadIdx.x] = a + s;
}

// load: every CU, 8 waves, MFMA chains for ~`iters` x 64 MFMAs
__global__ __launch_bounds__(512) void load_kernel(float* sink, int iters) {
    floatx16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
    bf16x8 u, v;
    for (int e = 0; e < 8; ++e) { u[e] = (__bf16)(0.001f * threadIdx.x); v[e] = (__bf16)0.5f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    unsigned long long* out;
    float *sink, *sink2;
    hipMalloc(&out, 64);
    hipMalloc(&sink, 4096);
    hipMalloc(&sink2, 256 * 512 * 4);
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    const int n_fma = 200000, n_mfma = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 4; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            if (mode == 1) hipLaunchKernelGGL(load_kernel, dim3(255), dim3(512), 0, s2, sink2, 40000);     // ~10+ ms of MFMAs
            hipEventRecord(e0, s1);
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s1, out, sink, n_fma, n_mfma);
            hipEventRecord(e1, s1);
            hipDeviceSynchronize();
            unsigned long long h[4];
            hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double f1 = (double)h[0] / (double)h[1] * 100.0, f2 = (double)h[2] / (double)h[3] * 100.0;
            printf("%s rep %d: fma chain: %llu memtime ticks / %llu realtime ticks -> s_memtime runs at %.1f MHz, %.2f memtime ticks "
                   "(%.2f ns) per dependent v_fma_f32 | mfma chain: s_memtime at %.1f MHz, %.2f ticks (%.2f ns) per dependent "
                   "32x32x16 MFMA | kernel %.3f ms by events\n",
                   mode ? "beside a chip-wide MFMA load" : "idle chip", rep, h[0], h[1], f1, (double)h[0] / n_fma,
                   (double)h[1] * 10.0 / n_fma, f2, (double)h[2] / n_mfma, (double)h[3] * 10.0 / n_mfma, ms);
        }
    }
    return 0;
}
