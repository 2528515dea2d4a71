// bf16 GEMM (fp32 accumulate) on v_mfma_f32_32x32x16_bf16 for gfx950.
//   block 256 threads = 4 waves (2x2), block tile 128x128x64, wave tile 64x64 = 2x2 MFMA tiles.
//   Operands are staged global -> registers (16 B/lane) -> LDS.  Two LDS images:
//     k-contiguous operand  : [mn][k], row stride 72 el (144 B)  -> ds_read_b128 fragments,
//                             conflict-free because 36 dwords * i is distinct mod 64 for i mod 16
//     mn-contiguous operand : [k][mn], row stride 160 el (320 B) -> ds_read_b64_tr_b16 (hardware
//                             transpose) fragments; 4 rows * 64 B fold onto all 64 banks
//   so the weight-gradient GEMM (both operands token-major, reduction over tokens) needs no
//   transposed copies in HBM.
//   Epilogue: each wave transposes its accumulators through a private fp32 LDS slab and finishes 8
//   consecutive columns per lane (16-byte residual/gate loads and C stores).  The epilogue kinds the model
//   uses are compile-time variants (EPI_*), so the hot kernels carry no dead branches or RNG code.
#include "gemm_bf16.h"
#include "gemm_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int TBM = 128, TBN = 128, TBK = 64;
constexpr int LDK = 72;
constexpr int LDM = 160;

__device__ __forceinline__ void unpack8(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(w[e] << 16);
        v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]), f2bf_pk(v[4], v[5]), f2bf_pk(v[6], v[7]));
}

// dropout replay on an 8-element chunk of operand A whose first element has id e
__device__ __forceinline__ uint4 drop_chunk8(uint4 t, const DropCtx& dc, uint64_t e) {
    float v[8], m[8];
    unpack8(t, v);
    if ((e & 7ull) == 0) {
        drop_mult8(dc, e, m);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = drop_mult(dc, e + i);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= m[i];
    return pack8(v);
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

template <bool AKC, bool BKC, bool ADROP, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_mfma_kernel(dsvg_gemm_desc p, int tiles_n, int nwg_mn,
                                                             int k_chunk, float* part, float* rs_part, int mode) {
    constexpr int A_ELEMS = AKC ? TBM * LDK : TBK * LDM;
    constexpr int B_ELEMS = BKC ? TBN * LDK : TBK * LDM;
    __shared__ __attribute__((aligned(16))) bf16_t smem[A_ELEMS + B_ELEMS];
    bf16_t* As = smem;
    bf16_t* Bs = smem + A_ELEMS;

    // Tile schedule (XCD-aware, bijective; workgroup b runs on XCD b % 8):
    //   mode 0  one tile per workgroup, 2-D grid: x = tile (each XCD owns a contiguous chunk of the tile list, so
    //           the column tiles of one row tile share that XCD's L2), y = K slice
    //   mode 1  split-K as a 1-D grid of nwg_mn * nsplit workgroups (nsplit % 8 == 0): all output tiles of one K
    //           slice run back to back on ONE XCD: each operand tile is fetched from HBM once, re-used from L2
    //   mode 2  persistent: gridDim.x (<= 4 per CU) resident workgroups walk their XCD's chunk of the tile list,
    //           which removes the launch/drain gaps of 10 us workgroups (the K loop is only 4-8 steps long)
    const int bid = blockIdx.x;
    const int xcd = bid % 8, local = bid / 8;
    int wg_base, wl_first, wl_end, wl_stride, kz;
    if (mode == 1) {
        wg_base = 0; wl_first = local % nwg_mn; wl_end = wl_first + 1; wl_stride = 1;
        kz = (local / nwg_mn) * 8 + xcd;
    } else {
        const int q8 = nwg_mn / 8, r8 = nwg_mn % 8;
        wg_base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        wl_first = local; wl_end = q8 + (xcd < r8 ? 1 : 0);
        wl_stride = mode == 2 ? ((int)gridDim.x + 7) / 8 : (wl_end > 0 ? wl_end : 1);
        kz = blockIdx.y;
    }
    for (int wl = wl_first; wl < wl_end; wl += wl_stride) {
    const int wgid = wg_base + wl;
    const int tile_m = wgid / tiles_n, tile_n = wgid % tiles_n;
    const int m0 = tile_m * TBM, n0 = tile_n * TBN;
    const int k_begin = kz * k_chunk;
    const int k_end = min(p.K, k_begin + k_chunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;
    const DropCtx adc = drop_make(ADROP ? p.a_drop_p : 0.f, p.seed, p.a_drop_site);

    // per-thread staging geometry, hoisted out of the K loop
    //   k-contiguous operand : thread -> (row = r + 32 j, 8-element k chunk c)      r = tid/8,  c = tid%8
    //   mn-contiguous operand: thread -> (k row = r + 16 j, 8-element mn chunk c)   r = tid/16, c = tid%16
    const int ca = AKC ? (tid & 7) : (tid & 15), rwa = AKC ? (tid >> 3) : (tid >> 4);
    const int cb = BKC ? (tid & 7) : (tid & 15), rwb = BKC ? (tid >> 3) : (tid >> 4);
    const bf16_t* pa[4];
    const bf16_t* pb[4];
    bool va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (AKC) { const int gm = m0 + rwa + 32 * j; va[j] = gm < p.M; pa[j] = A + (size_t)gm * p.lda + 8 * ca; }
        else     { const int gm = m0 + 8 * ca;       va[j] = gm < p.M; pa[j] = A + (size_t)(rwa + 16 * j) * p.lda + gm; }
        if (BKC) { const int gn = n0 + rwb + 32 * j; vb[j] = gn < p.N; pb[j] = B + (size_t)gn * p.ldb + 8 * cb; }
        else     { const int gn = n0 + 8 * cb;       vb[j] = gn < p.N; pb[j] = B + (size_t)(rwb + 16 * j) * p.ldb + gn; }
    }

    uint4 ra[4], rb[4];
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 v = zero4;
            if (AKC) {
                const int gk = k0 + 8 * ca;
                if (va[j] && gk < k_end) {
                    v = *reinterpret_cast<const uint4*>(pa[j] + k0);
                    if (gk + 8 > k_end) v = mask_tail8(v, k_end - gk);
                    if (ADROP) v = drop_chunk8(v, adc, (uint64_t)(m0 + rwa + 32 * j) * p.a_drop_ld + gk);
                }
            } else {
                const int gk = k0 + rwa + 16 * j;
                if (va[j] && gk < k_end) {
                    v = *reinterpret_cast<const uint4*>(pa[j] + (size_t)k0 * p.lda);
                    if (ADROP) v = drop_chunk8(v, adc, (uint64_t)gk * p.a_drop_ld + m0 + 8 * ca);
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint4 v = zero4;
            if (BKC) {
                const int gk = k0 + 8 * cb;
                if (vb[j] && gk < k_end) {
                    v = *reinterpret_cast<const uint4*>(pb[j] + k0);
                    if (gk + 8 > k_end) v = mask_tail8(v, k_end - gk);
                }
            } else {
                if (vb[j] && k0 + rwb + 16 * j < k_end) v = *reinterpret_cast<const uint4*>(pb[j] + (size_t)k0 * p.ldb);
            }
            rb[j] = v;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (AKC) *reinterpret_cast<uint4*>(&As[(rwa + 32 * j) * LDK + 8 * ca]) = ra[j];
            else     *reinterpret_cast<uint4*>(&As[(rwa + 16 * j) * LDM + 8 * ca]) = ra[j];
            if (BKC) *reinterpret_cast<uint4*>(&Bs[(rwb + 32 * j) * LDK + 8 * cb]) = rb[j];
            else     *reinterpret_cast<uint4*>(&Bs[(rwb + 16 * j) * LDM + 8 * cb]) = rb[j];
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

    // optional row sums of A (bias gradient inside the weight-gradient GEMM): one extra MFMA against an all-ones
    // B fragment per A fragment, only in the column-tile-0 workgroups and only in the waves with wn == 0
    const bool do_rs = (EPI == EPI_PARTIAL) && rs_part != nullptr && tile_n == 0 && wn == 0;
    floatx16 accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
    Frag8 ones;
    ones.u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);

    if (k_begin < k_end) load_tiles(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += TBK) {
        store_lds();
        __syncthreads();
        if (k0 + TBK < k_end) load_tiles(k0 + TBK);   // register prefetch overlaps the MFMAs below
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
            if (EPI == EPI_PARTIAL && do_rs) {
                accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, ones.v, accb[0], 0, 0, 0);
                accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ones.v, accb[1], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if (EPI == EPI_PARTIAL || (EPI == EPI_GENERIC && part)) {   // split-K partial: raw fp32 accumulators
        const size_t slice = dsvg_splitk_slice(p.M, p.N, rs_part != nullptr);   // [M*N partial | M row sums] per K slice
        float* my_part = part + (size_t)kz * slice;
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
        if (EPI == EPI_PARTIAL && do_rs && (lane & 31) == 0) {   // every column of accb holds the same row sums
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < p.M) rs_part[(size_t)kz * slice + m] = accb[i][r];
                }
        }
        return;
    }

    // ---- epilogue through a per-wave LDS slab (the A/B images are dead after the last barrier) --------------
    constexpr bool GEN = EPI == EPI_GENERIC;
    const bool has_res = GEN ? (p.res != nullptr) : (EPI == EPI_BIAS_RES_DROP);
    const bool has_gate = GEN ? (p.gate != nullptr) : (EPI == EPI_GATE);
    const bool relu = GEN ? (p.act == 1) : (EPI == EPI_BIAS_RELU_DROP);
    const bool res_pre = GEN ? (p.res_pre != 0) : false;
    const bool may_drop = GEN || EPI == EPI_BIAS_RES_DROP || EPI == EPI_BIAS_RELU_DROP;
    const DropCtx dc = drop_make(may_drop ? p.drop_p : 0.f, p.seed, p.drop_site);

    constexpr int SLD = 68;
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * SLD);
    const int chunk = lane & 7, rsub = lane >> 3;
    const int nb = n0 + wn * 64 + chunk * 8;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (EPI != EPI_GATE && p.bias && nb + e < p.N) ? p.bias[nb + e] : 0.f;
    bool fast = true;
    if (GEN) {
        const bool vec_c = p.c_f32 ? (!(p.ldc & 3) && !((uintptr_t)p.C & 15)) : (!(p.ldc & 7) && !((uintptr_t)p.C & 15));
        const bool vec_res = !p.res || (!(p.ldres & 7) && !((uintptr_t)p.res & 15));
        const bool vec_gate = !p.gate || (!(p.ldgate & 7) && !((uintptr_t)p.gate & 15));
        fast = vec_c && vec_res && vec_gate && (nb + 8 <= p.N);
    }
    const bool drop_aligned = !(p.N & 7);
    // NB: the two 32-row halves are handled by one lambda invoked with acc[0][*] and acc[1][*] explicitly: a loop
    // over i whose (large) body the compiler declines to unroll would index `acc` dynamically and push the
    // accumulators to scratch memory for the whole kernel (guide rule 20; measured 4x slowdown).
    auto half_tile = [&](const floatx16& c0, const floatx16& c1, const int i) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            slab[row * SLD + (lane & 31)] = c0[r];
            slab[row * SLD + 32 + (lane & 31)] = c1[r];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rl = rsub + 8 * q;
            const int m = m0 + wm * 64 + i * 32 + rl;
            const float4 lo = *reinterpret_cast<const float4*>(&slab[rl * SLD + chunk * 8]);
            const float4 hi = *reinterpret_cast<const float4*>(&slab[rl * SLD + chunk * 8 + 4]);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (m >= p.M || nb >= p.N) continue;
            if (GEN && !fast) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (nb + e < p.N) gemm_epilogue<bf16_t>(p, dc, m, nb + e, v[e]);
                continue;
            }
            float rv[8];
            if (has_res) unpack8(*reinterpret_cast<const uint4*>((const bf16_t*)p.res + (size_t)m * p.ldres + nb), rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] += bias8[e];
                if (has_res && res_pre) v[e] += rv[e];
                if (relu) v[e] = fmaxf(v[e], 0.f);
            }
            if (has_gate) {
                float gv[8];
                unpack8(*reinterpret_cast<const uint4*>((const bf16_t*)p.gate + (size_t)m * p.ldgate + nb), gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gv[e] > 0.f ? v[e] * p.gate_scale : 0.f;
            }
            if (may_drop && dc.on) {
                const uint64_t eid = (uint64_t)m * p.N + nb;
                float dm[8];
                if (drop_aligned) {
                    drop_mult8(dc, eid, dm);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dm[e] = drop_mult(dc, eid + e);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dm[e];
            }
            if (has_res && !res_pre) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            if (GEN && p.c_f32) {
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
                if (GEN && p.accumulate) {
                    float cv[8];
                    unpack8(*c, cv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += cv[e];
                }
                *c = pack8(v);
            }
        }
    };
    half_tile(acc[0][0], acc[0][1], 0);
    half_tile(acc[1][0], acc[1][1], 1);
    if (wl + wl_stride < wl_end) __syncthreads();   // the slab aliases the next tile's A/B images
    }   // persistent tile loop
}

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

template <bool AKC, bool BKC, bool ADROP, int EPI>
static void launch_variant(const dsvg_gemm_desc& d, dim3 grid, int tiles_n, int nwg, int k_chunk, float* part,
                           float* rs_part, int mode, hipStream_t st) {
    hipLaunchKernelGGL((gemm_bf16_mfma_kernel<AKC, BKC, ADROP, EPI>), grid, dim3(256), 0, st, d, tiles_n, nwg, k_chunk,
                       part, rs_part, mode);
}

int dsvg_gemm_bf16_launch(const dsvg_gemm_desc& d, int k_chunk, int nsplit, float* part, float* rs_part,
                          hipStream_t st, int* part_is_bf16) {
    *part_is_bf16 = 0;
    const int Kp = (d.K + 7) / 8 * 8;   // k-contiguous operands are read in 8-element chunks (tail masked)
    const bool aligned = !(d.lda & 7) && !(d.ldb & 7) && !((uintptr_t)d.A & 15) && !((uintptr_t)d.B & 15) &&
                         (!d.a_kc || d.lda >= Kp) && (!d.b_kc || d.ldb >= Kp);
    if (!aligned) {
        dsvg_set_error("gemm(bf16): operands must be 16-byte aligned with lda/ldb multiples of 8 "
                       "(lda=%lld ldb=%lld K=%d)", (long long)d.lda, (long long)d.ldb, d.K);
        return -1;
    }
    // mn-contiguous operands are read in 8-element chunks: the row must be padded to a multiple of 8
    if (!d.a_kc && d.lda < ((d.M + 7) / 8) * 8) { dsvg_set_error("gemm(bf16): lda too small for TN operand"); return -1; }
    if (!d.b_kc && d.ldb < ((d.N + 7) / 8) * 8) { dsvg_set_error("gemm(bf16): ldb too small for NN operand"); return -1; }
    const int tiles_m = dsvg_cdiv(d.M, TBM), tiles_n = dsvg_cdiv(d.N, TBN);
    const int nwg = tiles_m * tiles_n;
    // schedule (see the kernel): split-K -> mode 1 (or 0), everything else -> persistent (mode 2, <= 4 blocks per CU)
    static const bool force_2d = getenv("DSVG_SPLITK_2D") != nullptr;     // debugging knobs
    // persistent scheduling measured 5-12 % SLOWER than one tile per workgroup on the T x {256..768} x {256,512}
    // shapes (the 4-8 step K loop is issue-bound, not launch-bound), so it is opt-in
    static const bool no_persist = getenv("DSVG_GEMM_PERSIST") == nullptr;
    dim3 grid(nwg, nsplit);
    int mode = 0;
    if (nsplit > 1) {
        if ((nsplit % 8) == 0 && !force_2d) { grid = dim3(nwg * nsplit, 1); mode = 1; }
    } else if (!no_persist && nwg > 1024) {
        grid = dim3(1024, 1);
        mode = 2;
    }
    const bool adrop = d.a_drop_p > 0.f;
#define DSVG_V(AK, BK, AD, EP) launch_variant<AK, BK, AD, EP>(d, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, st)
    if (part) {     // split-K slices: dedicated variants that contain no epilogue code at all
        if (dsvg_gemm_bf16_glds_try(d, EPI_PARTIAL, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, st, part_is_bf16)) {
            DSVG_LAUNCH_CHECK("gemm_bf16_glds(split-k)");
            return 0;
        }
        if (d.a_kc && d.b_kc) { if (adrop) DSVG_V(true, true, true, EPI_PARTIAL); else DSVG_V(true, true, false, EPI_PARTIAL); }
        else if (d.a_kc && !d.b_kc) { if (adrop) DSVG_V(true, false, true, EPI_PARTIAL); else DSVG_V(true, false, false, EPI_PARTIAL); }
        else if (!d.a_kc && d.b_kc) { if (adrop) DSVG_V(false, true, true, EPI_PARTIAL); else DSVG_V(false, true, false, EPI_PARTIAL); }
        else { if (adrop) DSVG_V(false, false, true, EPI_PARTIAL); else DSVG_V(false, false, false, EPI_PARTIAL); }
        DSVG_LAUNCH_CHECK("gemm_bf16_mfma(split-k)");
        return 0;
    }

    // pick the compile-time epilogue variant when the call matches one exactly and everything is 16-byte aligned
    // (the LDS-DMA kernel also takes N % 8 != 0: it finishes the last partial 8-column chunk element-wise)
    auto classify = [&](bool ok) -> int {
        if (!ok) return EPI_GENERIC;
        if (!d.res && !d.gate && d.act == 0 && d.drop_p <= 0.f) return EPI_BIAS;
        if (d.res && !d.gate && d.act == 0) return EPI_BIAS_RES_DROP;
        if (!d.res && !d.gate && d.act == 1) return EPI_BIAS_RELU_DROP;
        if (d.gate && !d.res && !d.bias && d.act == 0 && d.drop_p <= 0.f) return EPI_GATE;
        return EPI_GENERIC;
    };
    // (res_pre - the residual inside the dropout, the embedding's `drop(fcn(...) + command + position)` - is a run-time
    // switch of the LDS-DMA kernel's residual epilogue; this file's compile-time variants have the residual outside)
    const bool vec_dma = !part && !d.c_f32 && !d.accumulate && !(d.ldc & 7) &&
                         !((uintptr_t)d.C & 15) && (!d.res || (!(d.ldres & 7) && !((uintptr_t)d.res & 15))) &&
                         (!d.gate || (!(d.ldgate & 7) && !((uintptr_t)d.gate & 15)));
    const bool vec_base = vec_dma && !d.res_pre;
    const int epi_dma = classify(vec_dma);
    const int epi = classify(vec_base && !(d.N & 7));
    if (epi_dma != EPI_GENERIC && dsvg_gemm_bf16_glds_try(d, epi_dma, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, st, nullptr)) {
        DSVG_LAUNCH_CHECK("gemm_bf16_glds");
        return 0;
    }
    if (d.a_kc && d.b_kc) {                 // forward layers
        if (adrop) DSVG_V(true, true, true, EPI_GENERIC);
        else if (epi == EPI_BIAS) DSVG_V(true, true, false, EPI_BIAS);
        else if (epi == EPI_BIAS_RES_DROP) DSVG_V(true, true, false, EPI_BIAS_RES_DROP);
        else if (epi == EPI_BIAS_RELU_DROP) DSVG_V(true, true, false, EPI_BIAS_RELU_DROP);
        else DSVG_V(true, true, false, EPI_GENERIC);
    } else if (d.a_kc && !d.b_kc) {         // input gradients (B = weight read [k][n])
        if (adrop) {
            if (epi == EPI_GATE) DSVG_V(true, false, true, EPI_GATE);
            else if (epi == EPI_BIAS) DSVG_V(true, false, true, EPI_BIAS);
            else DSVG_V(true, false, true, EPI_GENERIC);
        } else {
            if (epi == EPI_GATE) DSVG_V(true, false, false, EPI_GATE);
            else if (epi == EPI_BIAS) DSVG_V(true, false, false, EPI_BIAS);
            else DSVG_V(true, false, false, EPI_GENERIC);
        }
    } else if (!d.a_kc && !d.b_kc) {        // weight gradients (both operands token-major)
        if (adrop) DSVG_V(false, false, true, EPI_GENERIC);
        else DSVG_V(false, false, false, EPI_GENERIC);
    } else {
        if (adrop) DSVG_V(false, true, true, EPI_GENERIC);
        else DSVG_V(false, true, false, EPI_GENERIC);
    }
#undef DSVG_V
    DSVG_LAUNCH_CHECK("gemm_bf16_mfma");
    return 0;
}
