// Do a pure-MFMA wave and a pure-VALU wave on the SAME SIMD run concurrently on gfx950?  One 512-thread workgroup per CU:
// waves 0-3 issue N independent-accumulator 32x32x16 bf16 MFMAs, waves 4-7 issue M independent VALU instructions (8 chains);
// role masks select which half works (the other half exits at once).  Each wave reports s_memtime ticks and its SIMD id
// (HW_REG_HW_ID), so that the pairs that really shared a SIMD can be told from the ones that did not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int VOP, int SWAP, int PRIO>
__global__ __launch_bounds__(512, 2) void probe(float* out, unsigned long long* ticks, unsigned* simd, int roles, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool is_m = SWAP ? wave >= 4 : wave < 4;       // SWAP: the MFMA waves are the YOUNGER half of the workgroup
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0) simd[blockIdx.x * 8 + wave] = (hw >> 4) & 3;
    if (!((roles >> (is_m ? 0 : 1)) & 1)) { if (lane == 0) ticks[blockIdx.x * 8 + wave] = 0; return; }
    float res = 0.f;
    unsigned long long t0, t1;
    if (is_m) {
        floatx16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        bf16x8 u, v;
        for (int e = 0; e < 8; ++e) { u[e] = (__bf16)(0.001f * lane); v[e] = (__bf16)0.5f; }
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n_mfma / 16; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a3, 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < 16; ++r) res += a0[r] + a1[r] + a2[r] + a3[r];
    } else {
        float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
        const float c = 1.0001f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n_valu / 64; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (VOP == 0)
                    asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                                 "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
                else
                    asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                                 "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        res = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VOP, int SWAP = 0, int PRIO = 0>
void run(const char* vname, int n_mfma, int n_valu) {
    float* out; unsigned long long* ticks; unsigned* simd;
    const int nb = 256;
    (void)hipMalloc(&out, nb * 512 * 4); (void)hipMalloc(&ticks, nb * 8 * 8); (void)hipMalloc(&simd, nb * 8 * 4);
    for (int roles : {1, 2, 3}) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<VOP, SWAP, PRIO>), dim3(nb), dim3(512), 0, 0, out, ticks, simd, roles, n_mfma, n_valu);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(nb * 8); std::vector<unsigned> sd(nb * 8);
        (void)hipMemcpy(h.data(), ticks, nb * 8 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(sd.data(), simd, nb * 8 * 4, hipMemcpyDeviceToHost);
        double tm = 0, tv = 0; int same = 0;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) {
            tm += (double)h[b * 8 + (SWAP ? 4 : 0) + w]; tv += (double)h[b * 8 + (SWAP ? 0 : 4) + w];
            same += sd[b * 8 + w] == sd[b * 8 + 4 + w];
        }
        printf("%-38s roles %s: %6.2f ticks per MFMA (MFMA waves), %5.2f ticks per VALU instruction (VALU waves); wave w and w + 4 on the "
               "same SIMD in %d of %d pairs\n", vname, roles == 1 ? "MFMA only " : roles == 2 ? "VALU only " : "both      ",
               tm / (nb * 4) / n_mfma, tv / (nb * 4) / n_valu, same, nb * 4);
    }
    (void)hipFree(out); (void)hipFree(ticks); (void)hipFree(simd);
}

int main() {
    // equal solo durations: 4096 MFMAs x 32 cycles = 131k cycles; VALU at ~4-5 cycles each: 28672 instructions
    run<0>("v_fma_f32", 4096, 28672);
    run<1>("v_xor_b32", 4096, 28672);
    run<0, 1>("fma, MFMA waves younger", 4096, 28672);
    run<0, 1, 3>("fma, MFMA waves younger, s_setprio 3", 4096, 28672);
    run<0, 0, 3>("fma, MFMA waves older, s_setprio 3", 4096, 28672);
    return 0;
}
