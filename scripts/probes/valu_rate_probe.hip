// Issue cost of integer VALU instructions on gfx950 (s_memtime around 512 independent instructions per wave, 1 and 2 waves
// per SIMD): which of the dropout hash's operations are slow?  hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void probe(unsigned* out, unsigned long long* ticks, unsigned seed) {
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a1 * 3u;
    unsigned a4 = a0 + 1, a5 = a1 + 2, a6 = a2 + 3, a7 = a3 + 4;
    const unsigned c = 0x7feb352du;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 8; ++it) {
        if (OP == 0) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 1) { REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 2) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 3) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 4) { REP8(asm volatile("v_lshrrev_b32 %0, 15, %0\n v_lshrrev_b32 %1, 15, %1\n v_lshrrev_b32 %2, 15, %2\n v_lshrrev_b32 %3, 15, %3\n v_lshrrev_b32 %4, 15, %4\n v_lshrrev_b32 %5, 15, %5\n v_lshrrev_b32 %6, 15, %6\n v_lshrrev_b32 %7, 15, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 5) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 6) { REP8(asm volatile("v_dot2c_f32_bf16 %0, %1, %8\n v_dot2c_f32_bf16 %1, %2, %8\n v_dot2c_f32_bf16 %2, %3, %8\n v_dot2c_f32_bf16 %3, %4, %8\n v_dot2c_f32_bf16 %4, %5, %8\n v_dot2c_f32_bf16 %5, %6, %8\n v_dot2c_f32_bf16 %6, %7, %8\n v_dot2c_f32_bf16 %7, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name) {
    for (int threads : {256, 512}) {       // 1 and 2 waves per SIMD (one workgroup per CU: 256 workgroups)
        unsigned* out;
        unsigned long long* ticks;
        const int nb = 256, nw = nb * threads / 64;
        hipMalloc(&out, nb * threads * 4);
        hipMalloc(&ticks, nw * 8);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<OP>, dim3(nb), dim3(threads), 0, 0, out, ticks, 1u + i);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), ticks, nw * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        printf("%-18s %d waves/SIMD: %6.2f ticks per instruction per wave (512 instructions)\n", name, threads / 256, s / nw / 512.0);
        hipFree(out);
        hipFree(ticks);
    }
}

int main() {
    run<1>("v_xor_b32");
    run<4>("v_lshrrev_b32");
    run<0>("v_mul_lo_u32");
    run<2>("v_mul_u32_u24");
    run<3>("v_mad_u32_u24");
    run<5>("v_fma_f32");
    run<6>("v_dot2c_f32_bf16");
    return 0;
}
