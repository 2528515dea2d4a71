// Weight-stationary bf16 GEMM (fp32 accumulate) for the token-side GEMMs of the hot path:
//     C[M, N] = epilogue( X[M, K] . W^T )      M = tokens (tens of thousands), K in {256, 512}, N in {256..2827}
// forward linears (W stored [N][K]) and input-gradient GEMMs (W stored [K][N]).
//
// Why: with 128x128 tiles every workgroup streams BOTH operand tiles through L2 -> LDS (LDS-DMA tops out near
// 25 GB/s per CU, MI355X_MICROARCH "ldsdma-fill"), i.e. M*N*K*2*(1/128 + 1/128) bytes per GEMM - 537 MB for the
// 131072 x 512 x 256 FFN GEMM whose HBM traffic is 201 MB; the tiled kernel (gemm_bf16_glds.hip) measures ~5 TB/s on that
// path and ~2 TB/s of HBM.  Here the roles are split the way the shapes suggest (tiny weights, long token stream):
//   * each workgroup (8 waves, one per CU) keeps a 128 KiB column slice of W in LDS for its whole life:
//     image [BN][K], BN = 65536 / K columns, 16-byte chunk c of row n stored at c ^ (n & 15): the ds_read_b128 fragment
//     reads of the 16-lane groups (MI355X_MICROARCH §LDS) touch 16 distinct bank quads;
//   * X never touches LDS: a wave owns 32-row strips and loads its MFMA fragments straight from global memory
//     (lane (r, h) reads row r, bytes 32 j + 16 h of every 512-byte row chunk: whole 128-byte lines per row over four
//     loads), one strip (or K chunk) ahead of the MFMAs that consume it - no barriers after the fill;
//   * D = W_frag x X_frag (transposed tiles): a lane owns one token row; two v_permlane32_swap rounds give 16
//     consecutive columns per lane and the register epilogue of gemm_bf16_glds.hip (bias / ReLU / gate / dropout /
//     residual, 128-byte lines stored back to back) is reused unchanged.
// X is read ceil(N / BN) times (from L2 when the slices' workgroups share an XCD: same strips, same XCD by
// construction), W once per workgroup (32 MB per launch in total).
// Eligibility is decided on the host (dsvg_gemm_bf16_ws_try); everything else runs on the tiled kernels.
#include "gemm_bf16.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int WS_WAVES = 8;
constexpr int WS_THREADS = 64 * WS_WAVES;
constexpr int WS_IMG_BYTES = 128 * 1024;
constexpr int WS_LDS_BYTES = WS_IMG_BYTES + 1024;      // + the bias slice (<= 256 floats)

union Frag8 {
    bf16x8 v;
    uint4 u;
};

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

// RAG: N is not a multiple of 8 (argument head): the last 8-column chunk of a row is finished element-wise
template <int K, bool BKC, int EPI, bool RAG>
__global__ __launch_bounds__(WS_THREADS) void gemm_bf16_ws_kernel(dsvg_gemm_desc p, int n_slices, int stride_waves) {
    constexpr int BN = 65536 / K;           // columns of the resident weight slice
    constexpr int KCH = K / 256;            // 256-wide K chunks (64 fragment VGPRs each)
    constexpr int NG = BN / 128;            // column groups of 4 MFMA tiles
    constexpr int ROWB = K * 2;             // bytes per image row
    static_assert((KCH == 1 && NG == 2) || (KCH == 2 && NG == 1), "K = 256 or 512");
    extern __shared__ __attribute__((aligned(1024))) char wimg[];

    // workgroup b runs on XCD b % 8; the slices of one strip set sit on the same XCD (X re-reads hit that L2)
    const int bid = blockIdx.x;
    const int xcd = bid % 8, local = bid / 8;
    const int slice = local % n_slices;
    const int wg = (local / n_slices) * 8 + xcd;
    const int n0 = slice * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- fill the weight image (once) ----------------------------------------------------------------------------
    // every load of a batch is issued before the first LDS write (a load -> write loop pays one L2 round trip per
    // iteration: 16 of them, all 256 workgroups at once, with nothing else to hide them)
    const bf16_t* Bw = (const bf16_t*)p.B;
    if (BKC) {
        constexpr int PER = BN * (K / 8) / WS_THREADS;          // 16 chunks of 16 bytes per thread
        static_assert(PER == 16, "fill geometry");
        uint4 t[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = u * WS_THREADS + tid;
            const int n = i / (K / 8), c = i % (K / 8);
            t[u] = *reinterpret_cast<const uint4*>(Bw + (size_t)min(n0 + n, p.N - 1) * p.ldb + 8 * c);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = u * WS_THREADS + tid;
            const int n = i / (K / 8), c = i % (K / 8);
            *reinterpret_cast<uint4*>(wimg + n * ROWB + ((c ^ (n & 15)) * 16)) = t[u];
        }
    } else {
        // W is stored [K][N] (input-gradient GEMMs): transpose while filling.  One wave task = 8 columns x 128 k:
        // lane l takes the k pair (2 l, 2 l + 1) of the task's k block, packs the two rows into dwords and writes
        // (column n, k pair) with ds_write_b32 - the 64 lanes of one write hit 64 distinct banks
        // (bank = 4 * (((k >> 3) ^ (n & 15)) & 15) + ((k & 7) >> 1), k = 2 l covers every value once)
        constexpr int TASKS = (K / 128) * (BN / 8);             // 64
        constexpr int PERW = TASKS / WS_WAVES;                  // 8 tasks per wave
        uint4 lo[PERW], hi[PERW];
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int task = wave * PERW + u;
            const int kb = task % (K / 128), n8 = task / (K / 128);
            const int k = kb * 128 + 2 * lane;
            const bf16_t* src = Bw + (size_t)k * p.ldb + min(n0 + 8 * n8, p.N - 8);
            lo[u] = *reinterpret_cast<const uint4*>(src);
            hi[u] = *reinterpret_cast<const uint4*>(src + p.ldb);
        }
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int task = wave * PERW + u;
            const int kb = task % (K / 128), n8 = task / (K / 128);
            const int k = kb * 128 + 2 * lane;
            const uint32_t a[4] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w};
            const uint32_t b[4] = {hi[u].x, hi[u].y, hi[u].z, hi[u].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int n = 8 * n8 + e;
                const uint32_t wa = (e & 1) ? (a[e >> 1] >> 16) : (a[e >> 1] & 0xffffu);
                const uint32_t wb = (e & 1) ? (b[e >> 1] & 0xffff0000u) : (b[e >> 1] << 16);
                *reinterpret_cast<uint32_t*>(wimg + n * ROWB + (((k >> 3) ^ (n & 15)) * 16) + (k & 7) * 2) = wa | wb;
            }
        }
    }
    // the slice's bias rides in LDS too: a global bias load at the head of every epilogue would wait (in-order vmcnt)
    // for the token prefetch issued just before it
    float* lbias = reinterpret_cast<float*>(wimg + WS_IMG_BYTES);
    if (tid < BN) lbias[tid] = (p.bias != nullptr && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
    __syncthreads();

    // ---- fragment addressing ---------------------------------------------------------------------------------------
    // W fragment of column tile t, K step J: row n = 32 t + r, chunk 2 J + h  ->  physical chunk (2 J + h) ^ (r & 15)
    uint32_t fk[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) fk[jj] = (uint32_t)(r * ROWB + ((((2 * jj) | h) ^ (r & 15)) * 16));
    // `wofs` is an opaque zero refreshed once per strip: the fragments of the resident image are loop-invariant, and
    // hoisting 128 of them out of the strip loop (the compiler does) spills every one of them to scratch
    int wofs = 0;
    auto wfrag = [&](int t, int J) __attribute__((always_inline)) -> bf16x8 {
        Frag8 f;
        f.u = *reinterpret_cast<const uint4*>(wimg + wofs + t * 32 * ROWB + (J >> 3) * 256 + fk[J & 7]);
        return f.v;
    };
    // X fragments of one 32-row strip, K chunk kc: lane (r, h) holds row r, k = 256 kc + 16 j + 8 h .. + 7
    const bf16_t* Ax = (const bf16_t*)p.A;
    auto load_x = [&](uint4 (&a)[16], int strip, int kc) __attribute__((always_inline)) {
        const int row = min(strip * 32 + r, p.M - 1);
        const uint4* src = reinterpret_cast<const uint4*>(Ax + (size_t)row * p.lda + kc * 256) + h;
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = src[2 * j];
    };
    // 16 K steps x 4 column tiles; the W fragments of step j + 1 are fetched from LDS while the MFMAs of step j run
    // (left to itself the compiler re-uses one fragment register set and waits lgkmcnt(0) before every MFMA)
    auto mfma_chunk = [&](const uint4 (&a)[16], int g, int kc, floatx16& c0, floatx16& c1, floatx16& c2, floatx16& c3) __attribute__((always_inline)) {
        bf16x8 e0, e1, e2, e3, o0, o1, o2, o3;
        const int t0 = 4 * g, J0 = kc * 16;
        e0 = wfrag(t0 + 0, J0); e1 = wfrag(t0 + 1, J0); e2 = wfrag(t0 + 2, J0); e3 = wfrag(t0 + 3, J0);
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            Frag8 x;
            o0 = wfrag(t0 + 0, J0 + j + 1); o1 = wfrag(t0 + 1, J0 + j + 1);
            o2 = wfrag(t0 + 2, J0 + j + 1); o3 = wfrag(t0 + 3, J0 + j + 1);
            __builtin_amdgcn_sched_barrier(0);      // keep the fetch of step j + 1 ahead of the MFMAs of step j
            x.u = a[j];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e0, x.v, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e1, x.v, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e2, x.v, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e3, x.v, c3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 < 16) {
                e0 = wfrag(t0 + 0, J0 + j + 2); e1 = wfrag(t0 + 1, J0 + j + 2);
                e2 = wfrag(t0 + 2, J0 + j + 2); e3 = wfrag(t0 + 3, J0 + j + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            x.u = a[j + 1];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o0, x.v, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o1, x.v, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o2, x.v, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o3, x.v, c3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- register epilogue (same math and store pattern as gemm_bf16_glds.hip) ---------------------------------------
    constexpr bool has_res = EPI == EPI_BIAS_RES_DROP;
    constexpr bool has_gate = EPI == EPI_GATE;
    constexpr bool relu = EPI == EPI_BIAS_RELU_DROP;
    constexpr bool may_drop = EPI == EPI_BIAS_RES_DROP || EPI == EPI_BIAS_RELU_DROP;
    const DropCtx dc = drop_make(may_drop ? p.drop_p : 0.f, p.seed, p.drop_site);
    const bool has_bias = !has_gate && p.bias != nullptr;
    const bool n_aligned = !(p.N & 7);

    // one transposed 32x32 accumulator tile of token row m: ONE v_permlane32_swap round turns the lane's four column
    // quads (8 g + 4 h + e) into two octets, columns n32 + 8 h + (0..7) and n32 + 16 + 8 h + (0..7): the two lanes of a row
    // then write 32 contiguous bytes per store instruction (16 consecutive columns per lane would put their pieces
    // 32 bytes apart).  Epilogue math on the two aligned 8-column chunks; results left packed in `pk`.
    auto tile16 = [&](const floatx16& c, int m, int n32, uint4 (&pk)[2], int (&nvalid)[2]) __attribute__((always_inline)) {
        uint32_t x[4][4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[gq][e] = __float_as_uint(c[4 * gq + e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            auto s01 = __builtin_amdgcn_permlane32_swap(x[0][e], x[1][e], false, false);
            auto s23 = __builtin_amdgcn_permlane32_swap(x[2][e], x[3][e], false, false);
            x[0][e] = s01[0]; x[1][e] = s01[1]; x[2][e] = s23[0]; x[3][e] = s23[1];
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(x[2 * cb][e]); v[4 + e] = __uint_as_float(x[2 * cb + 1][e]); }
            const int nb = n32 + 16 * cb + 8 * h;
            const int nv = (m < p.M && nb < p.N) ? min(8, p.N - nb) : 0;
            nvalid[cb] = nv;
            if (nv == 0) { pk[cb] = make_uint4(0u, 0u, 0u, 0u); continue; }
            if (has_bias) {         // (columns past N hold zeros in the LDS copy)
                const float4 b0 = *reinterpret_cast<const float4*>(lbias + (nb - n0));
                const float4 b1 = *reinterpret_cast<const float4*>(lbias + (nb - n0) + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (has_gate) {
                float gv[8];
                const bf16_t* gp = (const bf16_t*)p.gate + (size_t)m * p.ldgate + nb;
                if (!RAG || nv == 8) unpack8(*reinterpret_cast<const uint4*>(gp), gv);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[e] = e < nv ? bf2f(gp[e]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gv[e] > 0.f ? v[e] * p.gate_scale : 0.f;
            }
            if (may_drop && dc.on) {
                float dm[8];
                if (!RAG || n_aligned) {
                    drop_mult8(dc, (uint64_t)m * p.N + nb, dm);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dm[e] = drop_mult(dc, (uint64_t)m * p.N + nb + e);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dm[e];
            }
            if (has_res) {
                float rv[8];
                const bf16_t* rp = (const bf16_t*)p.res + (size_t)m * p.ldres + nb;
                if (!RAG || nv == 8) unpack8(*reinterpret_cast<const uint4*>(rp), rv);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] = e < nv ? bf2f(rp[e]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            pk[cb] = pack8(v);
        }
    };
    auto put8 = [&](bf16_t* cp, const uint4& v, int nv) __attribute__((always_inline)) {
        if (!RAG || nv == 8) {
            if (nv) *reinterpret_cast<uint4*>(cp) = v;
            return;
        }
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < nv) cp[e] = (bf16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
    };
    // two column tiles = one 128-byte line per token row: four 32-byte pieces (the row's lane pair) back to back
    auto row_block = [&](const floatx16& c0, const floatx16& c1, int m, int ncol) __attribute__((always_inline)) {
        uint4 pk0[2], pk1[2];
        int nv0[2], nv1[2];
        tile16(c0, m, ncol, pk0, nv0);
        tile16(c1, m, ncol + 32, pk1, nv1);
        bf16_t* cp = (bf16_t*)p.C + (size_t)m * p.ldc + ncol + 8 * h;
        put8(cp, pk0[0], nv0[0]);
        put8(cp + 16, pk0[1], nv0[1]);
        put8(cp + 32, pk1[0], nv1[0]);
        put8(cp + 48, pk1[1], nv1[1]);
    };
    auto zero4 = [](floatx16& c0, floatx16& c1, floatx16& c2, floatx16& c3) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
    };
    auto finish = [&](const floatx16& c0, const floatx16& c1, const floatx16& c2, const floatx16& c3, int strip, int g) __attribute__((always_inline)) {
        const int m = strip * 32 + r;
        const int ncol = n0 + g * 128;
        row_block(c0, c1, m, ncol);
        row_block(c2, c3, m, ncol + 64);
    };

    // ---- token strips: wave w of workgroup wg takes strips gw, gw + stride_waves, ... ---------------------------------
    const int n_strips = (p.M + 31) / 32;
    int s = __builtin_amdgcn_readfirstlane(wg * WS_WAVES + wave);
    uint4 xa[16], xb[16];
    floatx16 c0, c1, c2, c3;
    if (KCH == 1) {
        // whole strip in 64 VGPRs, next strip prefetched while this one runs through both column groups
        auto strip_all = [&](const uint4 (&a)[16], int strip) __attribute__((always_inline)) {
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {          // (not unrolled: the epilogue is long, keep the loop in the I-cache)
                if (n0 + g * 128 >= p.N) break;
                zero4(c0, c1, c2, c3);
                mfma_chunk(a, g, 0, c0, c1, c2, c3);
                finish(c0, c1, c2, c3, strip, g);
            }
        };
        if (s < n_strips) load_x(xa, s, 0);
        while (s < n_strips) {
            asm volatile("" : "+s"(wofs));
            const int s1 = s + stride_waves;
            if (s1 < n_strips) load_x(xb, s1, 0);
            strip_all(xa, s);
            const int s2 = s1 + stride_waves;
            if (s2 < n_strips) load_x(xa, s2, 0);
            if (s1 < n_strips) strip_all(xb, s1);
            s = s2;
        }
    } else {
        // two K chunks per strip: chunk 1 of this strip and chunk 0 of the next one are in flight behind the MFMAs
        if (s < n_strips) load_x(xa, s, 0);
        while (s < n_strips) {
            asm volatile("" : "+s"(wofs));
            load_x(xb, s, 1);
            zero4(c0, c1, c2, c3);
            mfma_chunk(xa, 0, 0, c0, c1, c2, c3);
            const int sn = s + stride_waves;
            if (sn < n_strips) load_x(xa, sn, 0);
            mfma_chunk(xb, 0, 1, c0, c1, c2, c3);
            finish(c0, c1, c2, c3, s, 0);
            s = sn;
        }
    }
}

template <int K, bool BKC, int EPI, bool RAG>
void launch(const dsvg_gemm_desc& d, hipStream_t st) {
    constexpr int BN = 65536 / K;
    const int n_slices = (d.N + BN - 1) / BN;
    static const int n_cu = [] {
        int dev = 0, cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        return cu > 0 ? cu : 256;
    }();
    // one resident workgroup per CU; the workgroups of one slice come in multiples of 8 (one per XCD)
    int q = n_cu / (8 * n_slices);
    if (q < 1) q = 1;
    const int n_strips = (d.M + 31) / 32;
    while (q > 1 && (q - 1) * 8 * WS_WAVES >= n_strips) --q;      // never more waves than strips
    const int wg_per_slice = 8 * q;
    auto kern = gemm_bf16_ws_kernel<K, BKC, EPI, RAG>;
    DSVG_ENSURE_LDS(kern, WS_LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(wg_per_slice * n_slices), dim3(WS_THREADS), WS_LDS_BYTES, st, d, n_slices,
                       wg_per_slice * WS_WAVES);
}

template <int K>
bool dispatch(const dsvg_gemm_desc& d, int epi, hipStream_t st) {
    const bool rag = (d.N & 7) != 0;
    if (rag) {      // only the plain-bias forward linear comes with a ragged N (the 2827-wide argument head)
        if (K == 256 && d.b_kc && epi == EPI_BIAS) launch<256, true, EPI_BIAS, true>(d, st);
        else return false;
    } else if (d.b_kc) {
        if (epi == EPI_BIAS) launch<K, true, EPI_BIAS, false>(d, st);
        else if (epi == EPI_BIAS_RES_DROP) launch<K, true, EPI_BIAS_RES_DROP, false>(d, st);
        else if (epi == EPI_BIAS_RELU_DROP) launch<K, true, EPI_BIAS_RELU_DROP, false>(d, st);
        else return false;
    } else {
        if (epi == EPI_BIAS) launch<K, false, EPI_BIAS, false>(d, st);
        else if (epi == EPI_GATE) launch<K, false, EPI_GATE, false>(d, st);
        else return false;
    }
    return true;
}

}  // namespace

// `epi`: the compile-time epilogue class the caller derived (16-byte aligned C / res / gate, bf16 output, no accumulate)
bool dsvg_gemm_bf16_ws_try(const dsvg_gemm_desc& d, int epi, hipStream_t st) {
    // DSVG_GEMM_WS: 0 = off (default), 1 = the K = 256 shapes, 2 = K = 512 too.  Measured against the tiled LDS-DMA kernel
    // (profiles/r01_gemm_ws_microbench.log): 1-9 % faster on the K = 256 shapes at 131072 tokens, 20-35 % slower at K = 512,
    // and no difference on the whole train step (10.58 vs 10.54 ms) - so it stays opt-in; see DESIGN.md "What bounds the
    // token-side GEMMs" for what the probes say about both kernels.
    static const int knob = getenv("DSVG_GEMM_WS") ? atoi(getenv("DSVG_GEMM_WS")) : 0;
    static const int min_m = getenv("DSVG_GEMM_WS_MIN_M") ? atoi(getenv("DSVG_GEMM_WS_MIN_M")) : 16384;
    if (d.impl != 0 && d.impl != 5) return false;
    if (d.impl == 0 && (!knob || d.M < min_m || (d.K == 512 && knob < 2))) return false;
    if (!d.a_kc || d.a_drop_p > 0.f || (d.K != 256 && d.K != 512)) return false;
    if ((d.lda & 7) || (d.ldb & 7) || ((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) return false;
    if (d.M < 1 || d.N < 8 || (!d.b_kc && (d.N & 7))) return false;
    if (d.bias && ((uintptr_t)d.bias & 15)) return false;
    return d.K == 256 ? dispatch<256>(d, epi, st) : dispatch<512>(d, epi, st);
}
