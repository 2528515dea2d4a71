// Round 6: how fast is the one-round "burst" in which every workgroup of a token-stationary launch reads (or writes) its 256 rows of
// 512 bytes at the same time - with the access pattern of the kernels' epilogues (lane = token row: 16-byte pieces, two lanes per
// row, 64 different 128-byte lines per wave instruction) against fully coalesced wave instructions (1 KiB contiguous = 2 rows)?
// 248 workgroups of 512 threads (one per CU, lock-step), 63,488 rows, buffers rotated (cold).  ffn_bwd_dx's epilogue measured 9.5-10.6 us
// per 32 MB read burst and 10 us per 32 MB store burst = 3.2 TB/s (scripts/ffn_bwd_dx_probe.py).
// hipcc --offload-arch=gfx950 -O3 row_burst_probe.hip -o row_burst_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: row per lane pair (the epilogues' pattern); 1: coalesced (wave instruction = 1 KiB contiguous)
template <int MODE, bool STORE>
__global__ __launch_bounds__(512) void burst_kernel(u32x4* __restrict__ buf, unsigned* out, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tok = lane & 31, half = lane >> 5;
    const int row0 = blockIdx.x * 256 + wave * 32;
    u32x4 v[16];
    char* base = reinterpret_cast<char*>(buf);
    if (!STORE) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            size_t off;
            if (MODE == 0) off = (size_t)min(row0 + tok, rows - 1) * 512 + (size_t)((i >> 1) * 64 + half * 32 + (i & 1) * 16);
            else off = (size_t)min(row0 + 2 * i + (lane >> 5), rows - 1) * 512 + (size_t)(lane & 31) * 16;
            v[i] = *reinterpret_cast<const u32x4*>(base + off);
        }
        u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= v[i];
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;      // (never true: keeps the loads)
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            size_t off;
            if (MODE == 0) off = (size_t)min(row0 + tok, rows - 1) * 512 + (size_t)((i >> 1) * 64 + half * 32 + (i & 1) * 16);
            else off = (size_t)min(row0 + 2 * i + (lane >> 5), rows - 1) * 512 + (size_t)(lane & 31) * 16;
            const u32x4 w = {(unsigned)off, (unsigned)i, (unsigned)lane, 7u};
            *reinterpret_cast<u32x4*>(base + off) = w;
        }
    }
}

template <int MODE, bool STORE>
static float run(std::vector<u32x4*>& bufs, unsigned* out, int rows, int reps) {
    const int nb = (rows + 255) / 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((burst_kernel<MODE, STORE>), dim3(nb), dim3(512), 0, 0, bufs[i % bufs.size()], out, rows);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((burst_kernel<MODE, STORE>), dim3(nb), dim3(512), 0, 0, bufs[i % bufs.size()], out, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int rows = 63488, nbuf = 12;      // 12 x 32.5 MB: more than the 256 MB last-level cache
    std::vector<u32x4*> bufs(nbuf);
    for (auto& b : bufs) { hipMalloc(&b, (size_t)rows * 512); hipMemset(b, 1, (size_t)rows * 512); }
    unsigned* out;
    hipMalloc(&out, 4096);
    const double mb = rows * 512.0 / 1e6;
    for (int rep = 0; rep < 2; ++rep) {
        float t;
        t = run<0, false>(bufs, out, rows, 36); printf("read,  row per lane pair (epilogue pattern): %6.1f us  %5.2f TB/s\n", t, mb / t);
        t = run<1, false>(bufs, out, rows, 36); printf("read,  coalesced wave instructions:          %6.1f us  %5.2f TB/s\n", t, mb / t);
        t = run<0, true>(bufs, out, rows, 36);  printf("store, row per lane pair (epilogue pattern): %6.1f us  %5.2f TB/s\n", t, mb / t);
        t = run<1, true>(bufs, out, rows, 36);  printf("store, coalesced wave instructions:          %6.1f us  %5.2f TB/s\n", t, mb / t);
    }
    return 0;
}
