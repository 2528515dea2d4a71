// How fast can ONE CU pull an L2-resident stream?  (round 4: the group-stage layer kernels stream 1 MiB of weights per 32-row
// workgroup and take 31 us; the latent chain 640 KiB in 26 us; the weight-gradient GEMM ~37 GB/s per CU through LDS-DMA.)
// Each 512-thread workgroup reads the SAME `bytes_per_wg` of a buffer (so every XCD's L2 holds it after the first touch) as
// 16-byte loads per lane with D loads in flight per wave, like the kernels' weight streams; workgroups: 16 / 128 / 256 (one per CU).
// Reports GB/s per CU.  hipcc --offload-arch=gfx950 -O3 cu_ingest_probe.hip -o cu_ingest_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void stream_kernel(const u32x4* __restrict__ buf, unsigned* out, int frags_per_wave, int passes) {
    // wave w of the workgroup streams fragments w, w + 8, ... (1 KiB each: 64 lanes x 16 B) of the shared region
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int p = 0; p < passes; ++p) {
        const u32x4* base = buf + (size_t)wave * 64 + lane;
        for (int f = 0; f + D <= frags_per_wave; f += D) {
            u32x4 v[D];
#pragma unroll
            for (int d = 0; d < D; ++d) v[d] = __builtin_nontemporal_load(base + (size_t)(f + d) * 8 * 64) ;
#pragma unroll
            for (int d = 0; d < D; ++d) acc ^= v[d];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// the same stream by LDS-DMA (global_load_lds_dwordx4, 4 KiB per dma group per wave) from a 256-thread (4-wave) workgroup with G groups
// (= 4 G KiB) in flight per wave, which is how the bf16 GEMM and the fused kernels feed their operand images
__device__ __forceinline__ void dma4(const void* src, uint32_t lds) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}

template <int G, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const char* __restrict__ buf, unsigned* out, int blocks_per_wave, int passes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((void __attribute__((address_space(3)))*)lds);
    const uint32_t my_dst = __builtin_amdgcn_readfirstlane(lds0 + wave * (G * 4096));
    for (int p = 0; p < passes; ++p)
        for (int b = 0; b < blocks_per_wave; ++b) {
            dma4(buf + ((size_t)b * WAVES + wave) * 4096 + lane * 16, my_dst + (uint32_t)(b % G) * 4096);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (G - 1)) : "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = ((const unsigned*)lds)[threadIdx.x];
}

template <int G, int WAVES>
void run_dma(int nwg, size_t bytes_per_wg) {
    char* buf; unsigned* out;
    (void)hipMalloc(&buf, bytes_per_wg);
    (void)hipMemset(buf, 1, bytes_per_wg);
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int blocks_per_wave = (int)(bytes_per_wg / 4096 / WAVES);
    const int passes = 4;
    const size_t smem = (size_t)G * WAVES * 4096;
    (void)hipFuncSetAttribute((const void*)dma_kernel<G, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((dma_kernel<G, WAVES>), dim3(nwg), dim3(WAVES * 64), smem, 0, buf, out, blocks_per_wave, passes);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((dma_kernel<G, WAVES>), dim3(nwg), dim3(WAVES * 64), smem, 0, buf, out, blocks_per_wave, passes);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double gb = (double)blocks_per_wave * WAVES * 4096 * passes / 1e9;
    printf("LDS-DMA: %3d workgroups of %d waves, %4zu KiB per workgroup x %d passes, %3d KiB in flight per CU: %7.1f us -> %6.1f GB/s per CU, "
           "%5.2f TB/s chip-wide\n", nwg, WAVES, bytes_per_wg / 1024, passes, (int)(smem / 1024), ms * 1e3, gb / (ms * 1e-3), gb * nwg / (ms * 1e-3) / 1e3);
    (void)hipFree(buf); (void)hipFree(out);
}

template <int D>
void run(int nwg, size_t bytes_per_wg) {
    u32x4* buf; unsigned* out;
    (void)hipMalloc(&buf, bytes_per_wg);
    (void)hipMemset(buf, 1, bytes_per_wg);
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int frags_per_wave = (int)(bytes_per_wg / 1024 / 8);
    const int passes = 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_kernel<D>, dim3(nwg), dim3(512), 0, 0, buf, out, frags_per_wave, passes);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(stream_kernel<D>, dim3(nwg), dim3(512), 0, 0, buf, out, frags_per_wave, passes);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double gb = (double)(frags_per_wave / D * D) * 8 * 1024 * passes / 1e9;
    printf("%3d workgroups, %4zu KiB per workgroup x %d passes, %2d loads in flight per wave (%3d KiB per CU): %7.1f us -> %6.1f GB/s per CU, "
           "%5.2f TB/s chip-wide\n", nwg, bytes_per_wg / 1024, passes, D, D * 8, ms * 1e3, gb / (ms * 1e-3), gb * nwg / (ms * 1e-3) / 1e3);
    (void)hipFree(buf); (void)hipFree(out);
}

int main() {
    for (int nwg : {16, 128, 256}) {
        run<4>(nwg, 1 << 20);
        run<12>(nwg, 1 << 20);
        run<32>(nwg, 1 << 20);
    }
    for (int nwg : {16, 256}) {
        run_dma<2, 4>(nwg, 1 << 20);
        run_dma<6, 4>(nwg, 1 << 20);
        run_dma<8, 4>(nwg, 1 << 20);
        run_dma<4, 8>(nwg, 1 << 20);
    }
    run<12>(128, 128 << 10);
    run<12>(128, 8 << 20);
    return 0;
}
