// How do the MFMA and VALU work of ffn_fwd's chunk loop share a SIMD?  (round 4.)  A stripped copy of the loop's three stages on
// static LDS contents - G1: 16 dependent 32x32x16 MFMAs with the 4-deep A-fragment ring of ds_read_b128, E1: the bias / ReLU /
// dropout / pack VALU block (repeated NE times), G2: 16 MFMAs into 8 accumulators - no DMA, no global memory.  One workgroup
// per CU, 4 or 8 waves (1 or 2 per SIMD); s_memtime per wave around `reps` chunks.  Variants:
//   barrier 0 none | 1 one s_barrier per chunk at the top for every wave | 2 the kernel's offset (waves 0-3 before G1, 4-7 before G2)
//   fine    0 the three stages one after the other | 1 E1 of the NEXT chunk cut into 16 slices, one behind each MFMA of G2
//           | 2 slices behind the MFMAs of G1 and G2 (32 slices)
// hipcc --offload-arch=gfx950 -O3 mfma_issue_probe.hip -o mfma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
union Frag8 { bf16x8 v; uint4 u; };

__device__ __forceinline__ uint32_t pk(float lo, float hi) {
    const f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// one E1 unit of work on values v[0..7] (bias + relu + dropout word + pack), ~ (16 + 2 * 6 + 16 * 3 + 8) / 2 VALU
__device__ __forceinline__ uint4 e1_half(const floatx16& hid, int base, const float* sb, uint32_t hh, int ks2, uint32_t thresh, float scale) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(hid[base + e] + sb[e], 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t w = hh + (4 * ks2 + i + 1u) * 0x9e3779b9u;
        w ^= w >> 16; w *= 0x7feb352du; w ^= w >> 15;
        v[2 * i] *= (w & 0xffffu) < thresh ? 0.f : scale;
        v[2 * i + 1] *= (w >> 16) < thresh ? 0.f : scale;
    }
    return make_uint4(pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7]));
}

template <int BARRIER, int FINE, int NE, int PRIO>
__global__ __launch_bounds__(512, 2) void probe(float* out, unsigned long long* ticks, int reps, uint32_t seed) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];       // 2 chunk slots of 32 KiB + 2 KiB bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    for (int i = tid; i < 16384 + 512; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 7);   // small bf16 values
    __syncthreads();
    const float* sb1 = reinterpret_cast<const float*>(smem + 65536);
    bf16x8 xf[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { Frag8 f; f.u = make_uint4(0x3c003c00u + lane + k, 0x3c003c01u, 0x3c003c02u, 0x3c003c03u); xf[k] = f.v; }
    floatx16 yacc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[t][r] = 0.f;
    floatx16 hid = {}, hacc;       // hacc: G1's accumulator; hid: the finished tile E1 reads (FINE == 2: copied at the end of G1, 16 v_mov)
    bf16x8 hf[2];
    uint4 ring[4];
    const char* lbase = smem + lane * 16;
    auto ld = [&](const char* p) -> uint4 { return *reinterpret_cast<const uint4*>(p); };
    const uint32_t thresh = 6554u;
    const float scale = 1.f / 0.9f;
    uint32_t hh = hash32(seed ^ tid);
    auto g2frag = [](int p) -> int { return 2 * (p & 7) + (p >> 3); };
    uint4 e1o[2];
    auto e1_slice = [&](int s, int of) {        // slice s of `of`: the work of E1 cut evenly (NE repetitions each)
        // 16 accumulator values = 2 halves; cut into `of` slices over (NE x 2) half-units
#pragma unroll
        for (int u = 0; u < NE * 2; ++u) {
            if ((u * of) / (NE * 2) == s) {
                const int ks2 = u & 1;
                const uint4 r = e1_half(hid, 8 * ks2, sb1 + 8 * (u & 3), hh + u, ks2, thresh, scale);
                if (u < 2) e1o[ks2] = r;
                else { e1o[ks2].x ^= r.x; e1o[ks2].y ^= r.y; e1o[ks2].z ^= r.z; e1o[ks2].w ^= r.w; }
            }
        }
    };
    // E1 one accumulator value at a time (FINE == 3: unit i behind MFMA i of G2): ~9 VALU per unit
    float pend = 0.f;
    uint32_t wcur = 0;
    uint32_t e1w[8];
    auto e1_unit = [&](int i) {
        float v = fmaxf(hid[i] + sb1[i], 0.f);
        if ((i & 1) == 0) {
            uint32_t w = hh + ((i >> 1) + 1u) * 0x9e3779b9u;
            w ^= w >> 16; w *= 0x7feb352du; w ^= w >> 15;
            wcur = w;
            pend = v * ((w & 0xffffu) < thresh ? 0.f : scale);
        } else {
            v *= (wcur >> 16) < thresh ? 0.f : scale;
            e1w[i >> 1] = pk(pend, v);
        }
    };
    auto e1_units_commit = [&]() {
        e1o[0] = make_uint4(e1w[0], e1w[1], e1w[2], e1w[3]);
        e1o[1] = make_uint4(e1w[4], e1w[5], e1w[6], e1w[7]);
    };
    auto e1_commit = [&]() { Frag8 a, b; a.u = e1o[0]; b.u = e1o[1]; hf[0] = a.v; hf[1] = b.v; hh = hh * 0x9e3779b1u + 1u; };
    {
        Frag8 a; a.u = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); hf[0] = a.v; hf[1] = a.v;
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[i] = ld(lbase + i * 1024);
        e1o[0] = a.u; e1o[1] = a.u;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < reps; ++c) {
        const char* sc = lbase + (c & 1) * 32768;
        const char* sn = lbase + ((c + 1) & 1) * 32768;
        if (BARRIER == 1 || (BARRIER == 2 && !late)) __builtin_amdgcn_s_barrier();
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        // G1
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Frag8 a; a.u = ring[ks & 3];
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, xf[ks], hacc, 0, 0, 0);
            if (ks + 4 < 16) ring[ks & 3] = ld(sc + (ks + 4) * 1024);
            else ring[ks & 3] = ld(sc + (16 + g2frag(ks + 4 - 16)) * 1024);
            if (FINE == 2) e1_slice(ks, 32);        // (works on the PREVIOUS chunk's hid image: timing only)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FINE == 2) e1_commit();
        hid = hacc;
        if (FINE == 0) {
            if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int s = 0; s < 2; ++s) e1_slice(s, 2);
            e1_commit();
            if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        }
        if (BARRIER == 2 && late) __builtin_amdgcn_s_barrier();
        // G2
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            Frag8 a; a.u = ring[n & 3];
            yacc[n & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, hf[n >> 3], yacc[n & 7], 0, 0, 0);
            if (n + 4 < 16) ring[n & 3] = ld(sc + (16 + g2frag(n + 4)) * 1024);
            else ring[n & 3] = ld(sn + (n + 4 - 16) * 1024);
            if (FINE == 1) e1_slice(n, 16);
            if (FINE == 3) e1_unit(n);
            if (FINE == 2) e1_slice(16 + n, 32);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FINE == 1) e1_commit();
        if (FINE == 3) { e1_units_commit(); e1_commit(); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += yacc[t][r];
    out[blockIdx.x * blockDim.x + tid] = s + hid[0];
    if (lane == 0) ticks[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int BARRIER, int FINE, int NE, int PRIO = 0>
void run(const char* name) {
    const int reps = 64;
    for (int threads : {256, 512}) {
        float* out;
        unsigned long long* ticks;
        const int nb = 256, nw = nb * threads / 64;
        (void)hipMalloc(&out, nb * threads * 4);
        (void)hipMalloc(&ticks, nw * 8);
        (void)hipFuncSetAttribute((const void*)probe<BARRIER, FINE, NE, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 67584);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<BARRIER, FINE, NE, PRIO>), dim3(nb), dim3(threads), 67584, 0, out, ticks, reps, 1u + i);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<BARRIER, FINE, NE, PRIO>), dim3(nb), dim3(threads), 67584, 0, out, ticks, reps, 7u);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nw);
        (void)hipMemcpy(h.data(), ticks, nw * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[nw / 2] / reps;
        printf("%-58s %d wave(s)/SIMD: %7.0f ticks per chunk per wave (median; min %6.0f max %6.0f) = %5.1f cycles per MFMA of the SIMD | "
               "launch %.1f us -> %.0f MHz\n", name, threads / 256, med, (double)h[0] / reps, (double)h[nw - 1] / reps,
               med / (32.0 * (threads / 256)), ms * 1e3, (double)h[nw / 2] / (ms * 1e3));
        (void)hipFree(out); (void)hipFree(ticks);
    }
}

int main() {
    run<0, 0, 0>("MFMA + ring reads only (no E1), no barrier");
    run<0, 0, 1>("G1 E1 G2 coarse, no barrier");
    run<1, 0, 1>("G1 E1 G2 coarse, lockstep barrier");
    run<2, 0, 1>("G1 E1 G2 coarse, the kernel's offset barrier");
    run<0, 0, 1, 1>("G1 E1 G2 coarse, no barrier, s_setprio 1 in G1 / G2, 0 in E1");
    run<0, 0, 1, 3>("G1 E1 G2 coarse, no barrier, s_setprio 3 in G1 / G2, 0 in E1");
    run<2, 0, 1, 3>("G1 E1 G2 coarse, offset barrier, s_setprio 3 in G1 / G2, 0 in E1");
    run<1, 0, 1, 3>("G1 E1 G2 coarse, lockstep barrier, s_setprio 3 in G1 / G2, 0 in E1");
    run<0, 3, 1>("E1 one value behind each MFMA of G2 (16 units of ~9 VALU), no barrier");
    run<2, 3, 1>("E1 one value behind each MFMA of G2, offset barrier");
    run<0, 3, 1, 3>("E1 one value behind each MFMA of G2, no barrier, s_setprio 3 throughout");
    run<0, 0, 2>("G1 2xE1 G2 coarse, no barrier");
    run<0, 1, 1>("E1 in 16 slices behind G2's MFMAs, no barrier");
    run<0, 2, 1>("E1 in 32 slices behind G1's and G2's MFMAs, no barrier");
    run<2, 2, 1>("E1 in 32 slices, the kernel's offset barrier");
    run<0, 2, 2>("2xE1 in 32 slices, no barrier");
    return 0;
}
