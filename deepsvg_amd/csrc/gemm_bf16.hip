// bf16 GEMM (fp32 accumulate) on v_mfma_f32_32x32x16_bf16 for gfx950.
//   block 256 threads = 4 waves (2x2), block tile 128x128x64, wave tile 64x64 = 2x2 MFMA tiles.
//   Operands are staged global -> registers (16 B/lane) -> LDS.  Two LDS images:
//     k-contiguous operand  : [mn][k], row stride 72 el (144 B)  -> ds_read_b128 fragments,
//                             conflict-free because 36 dwords * i is distinct mod 64 for i mod 16
//     mn-contiguous operand : [k][mn], row stride 160 el (320 B) -> ds_read_b64_tr_b16 (hardware
//                             transpose) fragments; 4 rows * 64 B fold onto all 64 banks
//   so the weight-gradient GEMM (both operands token-major, reduction over tokens) needs no
//   transposed copies in HBM.
#include "dsvg_common.h"
#include "../../include/dsvg.h"
#include "gemm_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int TBM = 128, TBN = 128, TBK = 64;
constexpr int LDK = 72;
constexpr int LDM = 160;

__device__ __forceinline__ uint4 drop_chunk8(uint4 v, const DropCtx& dc, uint64_t e) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo = __uint_as_float(w[i] << 16) * drop_mult(dc, e + 2 * i);
        float hi = __uint_as_float(w[i] & 0xffff0000u) * drop_mult(dc, e + 2 * i + 1);
        w[i] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// keep the first `nvalid` (1..7) bf16 elements of an 8-element chunk, zero the rest (K tail)
__device__ __forceinline__ uint4 mask_tail8(uint4 v, int nvalid) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (2 * i >= nvalid) w[i] = 0u;
        else if (2 * i + 1 >= nvalid) w[i] &= 0x0000ffffu;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

union Frag8 {
    bf16x8 v;
    shortx4 h[2];
    uint4 u;
};

template <bool AKC, bool BKC, bool ADROP>
__global__ __launch_bounds__(256) void gemm_bf16_mfma_kernel(dsvg_gemm_desc p, int tiles_n, int nwg_mn,
                                                             int k_chunk, float* part) {
    constexpr int A_ELEMS = AKC ? TBM * LDK : TBK * LDM;
    constexpr int B_ELEMS = BKC ? TBN * LDK : TBK * LDM;
    __shared__ __attribute__((aligned(16))) bf16_t smem[A_ELEMS + B_ELEMS];
    bf16_t* As = smem;
    bf16_t* Bs = smem + A_ELEMS;

    const int bid = blockIdx.x;
    const int q8 = nwg_mn / 8, r8 = nwg_mn % 8;
    const int xcd = bid % 8, local = bid / 8;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    const int tile_m = wgid / tiles_n, tile_n = wgid % tiles_n;
    const int m0 = tile_m * TBM, n0 = tile_n * TBN;
    const int kz = blockIdx.y;
    const int k_begin = kz * k_chunk;
    const int k_end = min(p.K, k_begin + k_chunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;
    const DropCtx adc = drop_make(p.a_drop_p, p.seed, p.a_drop_site);

    uint4 ra[4], rb[4];
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

    auto load_a = [&](int k0) {
        if (AKC) {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gm = m0 + r + 32 * j, gk = k0 + 8 * c;
                uint4 v = zero4;
                if (gm < p.M && gk < k_end) {
                    v = *reinterpret_cast<const uint4*>(A + (size_t)gm * p.lda + gk);
                    if (gk + 8 > k_end) v = mask_tail8(v, k_end - gk);
                    if (ADROP && adc.on) v = drop_chunk8(v, adc, (uint64_t)gm * p.a_drop_ld + gk);
                }
                ra[j] = v;
            }
        } else {
            const int c = tid & 15, r = tid >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gk = k0 + r + 16 * j, gm = m0 + 8 * c;
                uint4 v = zero4;
                if (gk < k_end && gm < p.M) {
                    v = *reinterpret_cast<const uint4*>(A + (size_t)gk * p.lda + gm);
                    if (ADROP && adc.on) v = drop_chunk8(v, adc, (uint64_t)gk * p.a_drop_ld + gm);
                }
                ra[j] = v;
            }
        }
    };
    auto load_b = [&](int k0) {
        if (BKC) {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gn = n0 + r + 32 * j, gk = k0 + 8 * c;
                uint4 v = zero4;
                if (gn < p.N && gk < k_end) {
                    v = *reinterpret_cast<const uint4*>(B + (size_t)gn * p.ldb + gk);
                    if (gk + 8 > k_end) v = mask_tail8(v, k_end - gk);
                }
                rb[j] = v;
            }
        } else {
            const int c = tid & 15, r = tid >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gk = k0 + r + 16 * j, gn = n0 + 8 * c;
                rb[j] = (gk < k_end && gn < p.N) ? *reinterpret_cast<const uint4*>(B + (size_t)gk * p.ldb + gn) : zero4;
            }
        }
    };
    auto store_lds = [&]() {
        if (AKC) {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(&As[(r + 32 * j) * LDK + 8 * c]) = ra[j];
        } else {
            const int c = tid & 15, r = tid >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(&As[(r + 16 * j) * LDM + 8 * c]) = ra[j];
        }
        if (BKC) {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(&Bs[(r + 32 * j) * LDK + 8 * c]) = rb[j];
        } else {
            const int c = tid & 15, r = tid >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(&Bs[(r + 16 * j) * LDM + 8 * c]) = rb[j];
        }
    };

    // fragment fetch for MFMA k-step kk (16 k values), 32-row sub-tile starting at `base`
    auto frag_kc = [&](const bf16_t* S, int base, int kk) -> bf16x8 {
        Frag8 f;
        f.u = *reinterpret_cast<const uint4*>(&S[(base + (lane & 31)) * LDK + 16 * kk + 8 * (lane >> 5)]);
        return f.v;
    };
    auto frag_tr = [&](const bf16_t* S, int base, int kk) -> bf16x8 {
        const int g = lane >> 4, q = lane & 15;
        const int krow = 16 * kk + 8 * (g >> 1) + (q >> 2);
        const int col = base + 16 * (g & 1) + 4 * (q & 3);
        Frag8 f;
        f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (shortx4 __attribute__((address_space(3)))*)(&S[krow * LDM + col]));
        f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (shortx4 __attribute__((address_space(3)))*)(&S[(krow + 4) * LDM + col]));
        return f.v;
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (k_begin < k_end) {
        load_a(k_begin);
        load_b(k_begin);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += TBK) {
        store_lds();
        __syncthreads();
        if (k0 + TBK < k_end) {
            load_a(k0 + TBK);
            load_b(k0 + TBK);
        }
#pragma unroll
        for (int kk = 0; kk < TBK / 16; ++kk) {
            bf16x8 a0, a1, b0, b1;
            if (AKC) { a0 = frag_kc(As, wm * 64, kk); a1 = frag_kc(As, wm * 64 + 32, kk); }
            else     { a0 = frag_tr(As, wm * 64, kk); a1 = frag_tr(As, wm * 64 + 32, kk); }
            if (BKC) { b0 = frag_kc(Bs, wn * 64, kk); b1 = frag_kc(Bs, wn * 64 + 32, kk); }
            else     { b0 = frag_tr(Bs, wn * 64, kk); b1 = frag_tr(Bs, wn * 64 + 32, kk); }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    const DropCtx dc = drop_make(p.drop_p, p.seed, p.drop_site);
    float* my_part = part ? part + (size_t)kz * p.M * p.N : nullptr;
    if (my_part) {   // split-K partial: raw fp32 accumulators, 128-byte row segments per half wave
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
                    if (m < p.M && n < p.N) my_part[(size_t)m * p.N + n] = acc[i][j][r];
                }
        return;
    }
    // Row-per-lane-group epilogue: the MFMA C layout gives every lane ONE column (2-byte stores); instead each
    // wave transposes its 32x64 half tile through a private fp32 LDS slab (the A/B images are dead after the
    // last barrier) and every lane finishes 8 consecutive columns of a row: 16-byte residual/gate loads and
    // 16-byte C stores (8 rows x 128 B per wave instruction).
    constexpr int SLD = 68;
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * SLD);
    const int chunk = lane & 7, rsub = lane >> 3;
    const int nb = n0 + wn * 64 + chunk * 8;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (p.bias && nb + e < p.N) ? p.bias[nb + e] : 0.f;
    const bool vec_c = p.c_f32 ? (!(p.ldc & 3) && !((uintptr_t)p.C & 15)) : (!(p.ldc & 7) && !((uintptr_t)p.C & 15));
    const bool vec_res = !p.res || (!(p.ldres & 7) && !((uintptr_t)p.res & 15));
    const bool vec_gate = !p.gate || (!(p.ldgate & 7) && !((uintptr_t)p.gate & 15));
    const bool fast = vec_c && vec_res && vec_gate && (nb + 8 <= p.N);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + j * 32 + (lane & 31)] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rl = rsub + 8 * q;
            const int m = m0 + wm * 64 + i * 32 + rl;
            const float4 lo = *reinterpret_cast<const float4*>(&slab[rl * SLD + chunk * 8]);
            const float4 hi = *reinterpret_cast<const float4*>(&slab[rl * SLD + chunk * 8 + 4]);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (m >= p.M || nb >= p.N) continue;
            if (!fast) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (nb + e < p.N) gemm_epilogue<bf16_t>(p, dc, m, nb + e, v[e]);
                continue;
            }
            float rv[8];
            if (p.res) {
                const uint4 t = *reinterpret_cast<const uint4*>((const bf16_t*)p.res + (size_t)m * p.ldres + nb);
                const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rv[2 * e] = __uint_as_float(w[e] << 16);
                    rv[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] += bias8[e];
                if (p.res && p.res_pre) v[e] += rv[e];
                if (p.act == 1) v[e] = fmaxf(v[e], 0.f);
            }
            if (p.gate) {
                const uint4 t = *reinterpret_cast<const uint4*>((const bf16_t*)p.gate + (size_t)m * p.ldgate + nb);
                const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = __uint_as_float(w[e] << 16) > 0.f ? v[2 * e] * p.gate_scale : 0.f;
                    v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u) > 0.f ? v[2 * e + 1] * p.gate_scale : 0.f;
                }
            }
            if (dc.on) {
                const uint64_t eid = (uint64_t)m * p.N + nb;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= drop_mult(dc, eid + e);
            }
            if (p.res && !p.res_pre) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            if (p.c_f32) {
                float4* c = reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + nb);
                if (p.accumulate) {
                    const float4 c0 = c[0], c1 = c[1];
                    v[0] += c0.x; v[1] += c0.y; v[2] += c0.z; v[3] += c0.w;
                    v[4] += c1.x; v[5] += c1.y; v[6] += c1.z; v[7] += c1.w;
                }
                c[0] = make_float4(v[0], v[1], v[2], v[3]);
                c[1] = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                uint4* c = reinterpret_cast<uint4*>((bf16_t*)p.C + (size_t)m * p.ldc + nb);
                if (p.accumulate) {
                    const uint4 t = *c;
                    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(w[e] << 16);
                        v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                    }
                }
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (uint32_t)f2bf(v[2 * e]) | ((uint32_t)f2bf(v[2 * e + 1]) << 16);
                *c = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

template <typename T>
__global__ void gemm_naive_kernel(dsvg_gemm_desc p, int k_begin, int k_end, float* part);  // gemm.hip

// raw ds_read_b64_tr_b16 probe: lane l reads from byte offset off[l] of a 4 KiB LDS image filled with
// img[i] = i (16-bit).  Used by the test-suite to pin the hardware transpose semantics.
__global__ void trread_probe_kernel(const int* off, short* out) {
    __shared__ __attribute__((aligned(16))) short L[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) L[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    shortx4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (shortx4 __attribute__((address_space(3)))*)((char*)L + off[l]));
    out[l * 4 + 0] = v[0]; out[l * 4 + 1] = v[1]; out[l * 4 + 2] = v[2]; out[l * 4 + 3] = v[3];
}
extern "C" int dsvg_probe_trread(const int* off, short* out, void* stream) {
    hipLaunchKernelGGL(trread_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, off, out);
    DSVG_LAUNCH_CHECK("trread_probe");
    return 0;
}

int dsvg_gemm_bf16_launch(const dsvg_gemm_desc& d, int k_chunk, float* part, hipStream_t st) {
    const int Kp = (d.K + 7) / 8 * 8;   // k-contiguous operands are read in 8-element chunks (tail masked)
    const bool aligned = !(d.lda & 7) && !(d.ldb & 7) && !((uintptr_t)d.A & 15) && !((uintptr_t)d.B & 15) &&
                         (!d.a_kc || d.lda >= Kp) && (!d.b_kc || d.ldb >= Kp);
    if (!aligned) {
        dsvg_set_error("gemm(bf16): operands must be 16-byte aligned with lda/ldb/K multiples of 8 "
                       "(lda=%lld ldb=%lld K=%d)", (long long)d.lda, (long long)d.ldb, d.K);
        return -1;
    }
    // mn-contiguous operands are read in 8-element chunks: the row must be padded to a multiple of 8
    if (!d.a_kc && d.lda < ((d.M + 7) / 8) * 8) { dsvg_set_error("gemm(bf16): lda too small for TN operand"); return -1; }
    if (!d.b_kc && d.ldb < ((d.N + 7) / 8) * 8) { dsvg_set_error("gemm(bf16): ldb too small for NN operand"); return -1; }
    const int tiles_m = dsvg_cdiv(d.M, TBM), tiles_n = dsvg_cdiv(d.N, TBN);
    const int nwg = tiles_m * tiles_n;
    const int nsplit = (d.K + k_chunk - 1) / k_chunk;
    dim3 grid(nwg, nsplit);
    // the dropout-replay prologue is a compile-time variant so the common kernels carry no RNG code
    const bool adrop = d.a_drop_p > 0.f;
#define DSVG_LAUNCH_BF16(AK, BK)                                                                                   \
    do {                                                                                                           \
        if (adrop) hipLaunchKernelGGL((gemm_bf16_mfma_kernel<AK, BK, true>), grid, dim3(256), 0, st, d, tiles_n,   \
                                      nwg, k_chunk, part);                                                         \
        else hipLaunchKernelGGL((gemm_bf16_mfma_kernel<AK, BK, false>), grid, dim3(256), 0, st, d, tiles_n, nwg,   \
                                k_chunk, part);                                                                    \
    } while (0)
    if (d.a_kc && d.b_kc) DSVG_LAUNCH_BF16(true, true);
    else if (d.a_kc && !d.b_kc) DSVG_LAUNCH_BF16(true, false);
    else if (!d.a_kc && d.b_kc) DSVG_LAUNCH_BF16(false, true);
    else DSVG_LAUNCH_BF16(false, false);
#undef DSVG_LAUNCH_BF16
    DSVG_LAUNCH_CHECK("gemm_bf16_mfma");
    return 0;
}
