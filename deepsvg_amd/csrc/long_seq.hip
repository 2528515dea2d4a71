// Sequences of more than 64 tokens (the one-stage / autoregressive configs at the reference's default
// max_total_len = 240: 241- and 242-token sequences, deepsvg/model/config.py:43,74-89).  The short-sequence kernels keep a
// sequence's key-padding mask in one 64-bit word and its scores in registers; here the command masks - always "before the
// first EOS" (deepsvg/model/utils.py:7-32) - are a LENGTH per sequence, and attention streams over the keys:
//   dsvg_seq_lens            first-EOS index of every command sequence
//   dsvg_attention_long_*    lane per query row, K / V (backward: Q, K, V, dO) of one (sequence, head) in LDS as fp32,
//                            online softmax in the forward pass, two sweeps + a per-key pass in the backward pass
//                            (probabilities recomputed, no S x S tensor), optional causal mask, the same dropout
//                            element ids as attention.hip (so S <= 64 cases agree with the short kernels)
//   dsvg_prefix_mean_*       masked mean-pool over the valid prefix (deepsvg/model/model.py:137)
// These are correctness-first VALU kernels: at 242 tokens the score work is 8x that of the 31-token stages but still
// < 10 % of the layer's GEMM FLOPs.
#include "dsvg_common.h"
#include "../../include/dsvg.h"

namespace {
constexpr int AL_THREADS = 256;
constexpr int AL_LD = 33;           // fp32 row stride in LDS: a column walk over rows hits distinct banks
constexpr int AL_MAX_S = 256;

template <typename T>
__device__ __forceinline__ void stage_slab(float* dst, const T* __restrict__ src, long long ld, int S) {
    for (int idx = threadIdx.x; idx < S * 32; idx += AL_THREADS) {
        const int j = idx >> 5, c = idx & 31;
        dst[j * AL_LD + c] = Elem<T>::ld(src + (long long)j * ld + c);
    }
}

template <typename T>
__global__ __launch_bounds__(AL_THREADS) void attn_long_fwd_kernel(const T* __restrict__ qkv,
                                                                   const int32_t* __restrict__ seq_len,
                                                                   T* __restrict__ out, int S, int H, float scale,
                                                                   int causal, float drop_p, uint32_t site,
                                                                   const uint64_t* seed) {
    extern __shared__ float sm[];
    float* Ks = sm;
    float* Vs = sm + S * AL_LD;
    const int b = blockIdx.x, h = blockIdx.y, d = H * 32;
    const T* base = qkv + (size_t)b * S * 3 * d + h * 32;
    stage_slab<T>(Ks, base + d, 3LL * d, S);
    stage_slab<T>(Vs, base + 2 * d, 3LL * d, S);
    __syncthreads();
    const int len = seq_len ? min(max(seq_len[b], 0), S) : S;
    const DropCtx dc = drop_make(drop_p, seed, site);
    for (int i = threadIdx.x; i < S; i += AL_THREADS) {
        float q[32], o[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) { q[c] = Elem<T>::ld(base + (size_t)i * 3 * d + c) * scale; o[c] = 0.f; }
        const int jend = causal ? min(len, i + 1) : len;
        float m = -INFINITY, l = 0.f;
        const uint64_t drow = ((uint64_t)b * H + h) * S + i;         // dropout row (dsvg_common.h: attn_drop_*)
        uint32_t hrow = 0;
        for (int j = 0; j < jend; ++j) {
            if ((j & 31) == 0) hrow = attn_drop_row(dc, drow, j >> 5);
            const float* kr = Ks + j * AL_LD;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) s = fmaf(q[c], kr[c], s);
            const float mn = fmaxf(m, s);
            const float corr = __expf(m - mn), e = __expf(s - mn);
            l = l * corr + e;
            const float pe = e * attn_drop_key(dc, hrow, j);
            const float* vr = Vs + j * AL_LD;
#pragma unroll
            for (int c = 0; c < 32; ++c) o[c] = fmaf(pe, vr[c], o[c] * corr);
            m = mn;
        }
        const float inv = jend > 0 ? 1.f / l : 0.f;
        T* dst = out + ((size_t)b * S + i) * d + h * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) Elem<T>::st(dst + c, o[c] * inv);
    }
}

// One query row per (sequence, head): the incremental decoding step (query = the newest token `row`, keys = the cached
// rows j <= row that the length mask allows).  Lane j scores key j; block-wide max / sum; 32 lanes then accumulate the
// output columns.  O(S) per step instead of the O(S^2) of re-running every query row.
template <typename T>
__global__ __launch_bounds__(AL_THREADS) void attn_long_row_kernel(const T* __restrict__ qkv,
                                                                   const int32_t* __restrict__ seq_len,
                                                                   T* __restrict__ out, int S, int H, float scale,
                                                                   int row, float drop_p, uint32_t site,
                                                                   const uint64_t* seed) {
    extern __shared__ float sm[];
    float* Vs = sm;                     // [S][33]
    float* pr = Vs + S * AL_LD;         // [S] probabilities (already multiplied by the dropout mask)
    __shared__ float red[AL_THREADS / 64];
    __shared__ float bc[2];
    const int b = blockIdx.x, h = blockIdx.y, d = H * 32;
    const T* base = qkv + (size_t)b * S * 3 * d + h * 32;
    const int len = seq_len ? min(max(seq_len[b], 0), S) : S;
    const int jend = min(len, row + 1);
    stage_slab<T>(Vs, base + 2 * d, 3LL * d, jend);
    const DropCtx dc = drop_make(drop_p, seed, site);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s_own[(AL_MAX_S + AL_THREADS - 1) / AL_THREADS];
    float mx = -INFINITY;
    for (int j = threadIdx.x, k = 0; j < S; j += AL_THREADS, ++k) {
        float s = -INFINITY;
        if (j < jend) {
            s = 0.f;
            const T* qr = base + (size_t)row * 3 * d;
            const T* kr = base + (size_t)j * 3 * d + d;
#pragma unroll
            for (int c = 0; c < 32; ++c) s = fmaf(Elem<T>::ld(qr + c) * scale, Elem<T>::ld(kr + c), s);
        }
        s_own[k] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int w = 1; w < AL_THREADS / 64; ++w) m = fmaxf(m, red[w]);
        bc[0] = m;
    }
    __syncthreads();
    const float m = bc[0];
    float sum = 0.f;
    const uint64_t drow = ((uint64_t)b * H + h) * S + row;
    for (int j = threadIdx.x, k = 0; j < S; j += AL_THREADS, ++k) {
        const float e = j < jend ? __expf(s_own[k] - m) : 0.f;
        sum += e;
        if (j < S) pr[j] = e * attn_drop_mult(dc, drow, j);
    }
    sum = wave_sum(sum);
    __syncthreads();                    // red[] was read by thread 0 above
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = 0.f;
        for (int w = 0; w < AL_THREADS / 64; ++w) l += red[w];
        bc[1] = jend > 0 ? 1.f / l : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c = threadIdx.x;
        float o = 0.f;
        for (int j = 0; j < jend; ++j) o = fmaf(pr[j], Vs[j * AL_LD + c], o);
        Elem<T>::st(out + ((size_t)b * S + row) * d + h * 32 + c, o * bc[1]);
    }
}

template <typename T>
__global__ __launch_bounds__(AL_THREADS) void attn_long_bwd_kernel(const T* __restrict__ qkv,
                                                                   const int32_t* __restrict__ seq_len,
                                                                   const T* __restrict__ dout, T* __restrict__ dqkv,
                                                                   int S, int H, float scale, int causal, float drop_p,
                                                                   uint32_t site, const uint64_t* seed) {
    extern __shared__ float sm[];
    float* Qs = sm;
    float* Ks = Qs + S * AL_LD;
    float* Vs = Ks + S * AL_LD;
    float* Gs = Vs + S * AL_LD;
    float* lse_s = Gs + S * AL_LD;
    float* D_s = lse_s + S;
    const int b = blockIdx.x, h = blockIdx.y, d = H * 32;
    const T* base = qkv + (size_t)b * S * 3 * d + h * 32;
    stage_slab<T>(Qs, base, 3LL * d, S);
    stage_slab<T>(Ks, base + d, 3LL * d, S);
    stage_slab<T>(Vs, base + 2 * d, 3LL * d, S);
    stage_slab<T>(Gs, dout + (size_t)b * S * d + h * 32, (long long)d, S);
    __syncthreads();
    const int len = seq_len ? min(max(seq_len[b], 0), S) : S;
    const DropCtx dc = drop_make(drop_p, seed, site);
    const uint64_t hbase = ((uint64_t)b * H + h) * S;       // dropout row of query i = hbase + i
    T* dbase = dqkv + (size_t)b * S * 3 * d + h * 32;

    // ---- pass 1: lane = query row i: lse_i, D_i = sum_j P_ij dP_ij, dq_i ---------------------------------------
    for (int i = threadIdx.x; i < S; i += AL_THREADS) {
        float q[32], go[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) { q[c] = Qs[i * AL_LD + c] * scale; go[c] = Gs[i * AL_LD + c]; }
        const int jend = causal ? min(len, i + 1) : len;
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < jend; ++j) {
            const float* kr = Ks + j * AL_LD;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) s = fmaf(q[c], kr[c], s);
            const float mn = fmaxf(m, s);
            l = l * __expf(m - mn) + __expf(s - mn);
            m = mn;
        }
        const float lse = jend > 0 ? m + __logf(l) : INFINITY;
        float D = 0.f, A[32], Bv[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) { A[c] = 0.f; Bv[c] = 0.f; }
        uint32_t hrow = 0;
        for (int j = 0; j < jend; ++j) {
            if ((j & 31) == 0) hrow = attn_drop_row(dc, hbase + i, j >> 5);
            const float* kr = Ks + j * AL_LD;
            const float* vr = Vs + j * AL_LD;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) { s = fmaf(q[c], kr[c], s); dp = fmaf(go[c], vr[c], dp); }
            const float p = __expf(s - lse);
            dp *= attn_drop_key(dc, hrow, j);
            const float pd = p * dp;
            D += pd;
#pragma unroll
            for (int c = 0; c < 32; ++c) { A[c] = fmaf(pd, kr[c], A[c]); Bv[c] = fmaf(p, kr[c], Bv[c]); }
        }
        lse_s[i] = lse;
        D_s[i] = D;
        T* dq = dbase + (size_t)i * 3 * d;
#pragma unroll
        for (int c = 0; c < 32; ++c) Elem<T>::st(dq + c, (A[c] - D * Bv[c]) * scale);     // sum_j P (dP - D) k_j * scale
    }
    __syncthreads();

    // ---- pass 2: lane = key row j: dk_j, dv_j over the query rows that see it ---------------------------------------
    for (int j = threadIdx.x; j < S; j += AL_THREADS) {
        float dk[32], dv[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        if (j < len) {
            float k[32], v[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) { k[c] = Ks[j * AL_LD + c]; v[c] = Vs[j * AL_LD + c]; }
            for (int r = causal ? j : 0; r < S; ++r) {
                const float* qr = Qs + r * AL_LD;
                const float* gr = Gs + r * AL_LD;
                float s = 0.f, dpv = 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) { s = fmaf(qr[c], k[c], s); dpv = fmaf(gr[c], v[c], dpv); }
                const float p = __expf(s * scale - lse_s[r]);
                const float mult = attn_drop_mult(dc, hbase + r, j);
                const float pd = p * mult;
                const float ds = p * (dpv * mult - D_s[r]) * scale;
#pragma unroll
                for (int c = 0; c < 32; ++c) { dv[c] = fmaf(pd, gr[c], dv[c]); dk[c] = fmaf(ds, qr[c], dk[c]); }
            }
        }
        T* drow = dbase + (size_t)j * 3 * d;
#pragma unroll
        for (int c = 0; c < 32; ++c) { Elem<T>::st(drow + d + c, dk[c]); Elem<T>::st(drow + 2 * d + c, dv[c]); }
    }
}

__global__ void seq_lens_kernel(const float* __restrict__ commands, long long n_seq, int S, int eos,
                                int32_t* __restrict__ lens) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_seq) return;
    const float* row = commands + b * S;
    int fe = S;
    for (int s = S - 1; s >= 0; --s)
        if ((int)row[s] == eos) fe = s;
    lens[b] = fe;
}

template <typename T>
__global__ void prefix_mean_fwd_kernel(const T* __restrict__ x, const int32_t* __restrict__ lens, T* __restrict__ out,
                                       int S, int d) {
    const long long b = blockIdx.x;
    const int len = min(max(lens[b], 0), S);
    const float inv = 1.f / (float)len;                    // 0 valid tokens: 0/0 like the reference (model.py:137)
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float s = 0.f;
        const T* px = x + b * S * d + c;
        for (int i = 0; i < len; ++i, px += d) s += Elem<T>::ld(px);
        Elem<T>::st(out + b * d + c, s * inv);
    }
}
template <typename T>
__global__ void prefix_mean_bwd_kernel(const T* __restrict__ dout, const int32_t* __restrict__ lens, T* __restrict__ dx,
                                       int S, int d) {
    const long long b = blockIdx.x;
    const int len = min(max(lens[b], 0), S);
    const float inv = 1.f / (float)len;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const float g = Elem<T>::ld(dout + b * d + c) * inv;
        T* px = dx + b * S * d + c;
        for (int i = 0; i < S; ++i, px += d) Elem<T>::st(px, i < len ? g : 0.f);
    }
}
}  // namespace

extern "C" int dsvg_seq_lens(const float* commands, int64_t n_seq, int32_t S, int32_t eos_id, int32_t* lens,
                             void* stream) {
    DSVG_CHECK_ARG(commands && lens && n_seq > 0 && S > 0, "seq_lens: bad args");
    hipLaunchKernelGGL(seq_lens_kernel, dim3((unsigned)dsvg_cdiv(n_seq, 64)), dim3(64), 0, (hipStream_t)stream, commands,
                       (long long)n_seq, S, eos_id, lens);
    DSVG_LAUNCH_CHECK("seq_lens");
    return 0;
}

extern "C" int dsvg_attention_long_fwd(int32_t dtype, const void* qkv, const int32_t* seq_len, void* out, int64_t n_seq,
                                       int32_t S, int32_t n_heads, float scale, int32_t causal, int32_t only_row,
                                       float drop_p, uint32_t drop_site, const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(qkv && out && n_seq > 0 && S > 0 && S <= AL_MAX_S && n_heads > 0 && n_seq < (1ll << 31),
                   "attention_long_fwd: bad args (S=%d, at most %d)", S, AL_MAX_S);
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "attention_long_fwd: dropout needs a seed pointer");
    DSVG_CHECK_ARG(only_row < S && (only_row < 0 || causal), "attention_long_fwd: only_row is the causal decoding step");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n_seq, (unsigned)n_heads);
    if (only_row >= 0) {            // incremental decoding: one query row, everything else of `out` untouched
        const size_t lds1 = ((size_t)S * AL_LD + S) * sizeof(float);
        if (dtype == DSVG_F32) {
            auto kern = attn_long_row_kernel<float>;
            DSVG_ENSURE_LDS(kern, lds1);
            hipLaunchKernelGGL(kern, grid, dim3(AL_THREADS), lds1, st, (const float*)qkv, seq_len, (float*)out, S, n_heads,
                               scale, only_row, drop_p, drop_site, seed);
        } else if (dtype == DSVG_BF16) {
            auto kern = attn_long_row_kernel<bf16_t>;
            DSVG_ENSURE_LDS(kern, lds1);
            hipLaunchKernelGGL(kern, grid, dim3(AL_THREADS), lds1, st, (const bf16_t*)qkv, seq_len, (bf16_t*)out, S,
                               n_heads, scale, only_row, drop_p, drop_site, seed);
        } else { dsvg_set_error("attention_long_fwd: bad dtype"); return -1; }
        DSVG_LAUNCH_CHECK("attention_long_row");
        return 0;
    }
    const size_t lds = (size_t)2 * S * AL_LD * sizeof(float);
    if (dtype == DSVG_F32) {
        auto kern = attn_long_fwd_kernel<float>;
        DSVG_ENSURE_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(AL_THREADS), lds, st, (const float*)qkv, seq_len, (float*)out, S, n_heads, scale,
                           causal, drop_p, drop_site, seed);
    } else if (dtype == DSVG_BF16) {
        auto kern = attn_long_fwd_kernel<bf16_t>;
        DSVG_ENSURE_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(AL_THREADS), lds, st, (const bf16_t*)qkv, seq_len, (bf16_t*)out, S, n_heads,
                           scale, causal, drop_p, drop_site, seed);
    } else { dsvg_set_error("attention_long_fwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("attention_long_fwd");
    return 0;
}

extern "C" int dsvg_attention_long_bwd(int32_t dtype, const void* qkv, const int32_t* seq_len, const void* dout, void* dqkv,
                                       int64_t n_seq, int32_t S, int32_t n_heads, float scale, int32_t causal,
                                       float drop_p, uint32_t drop_site, const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(qkv && dout && dqkv && n_seq > 0 && S > 0 && S <= AL_MAX_S && n_heads > 0 && n_seq < (1ll << 31),
                   "attention_long_bwd: bad args (S=%d, at most %d)", S, AL_MAX_S);
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "attention_long_bwd: dropout needs a seed pointer");
    const size_t lds = ((size_t)4 * S * AL_LD + 2 * S) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n_seq, (unsigned)n_heads);
    if (dtype == DSVG_F32) {
        auto kern = attn_long_bwd_kernel<float>;
        DSVG_ENSURE_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(AL_THREADS), lds, st, (const float*)qkv, seq_len, (const float*)dout,
                           (float*)dqkv, S, n_heads, scale, causal, drop_p, drop_site, seed);
    } else if (dtype == DSVG_BF16) {
        auto kern = attn_long_bwd_kernel<bf16_t>;
        DSVG_ENSURE_LDS(kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(AL_THREADS), lds, st, (const bf16_t*)qkv, seq_len, (const bf16_t*)dout,
                           (bf16_t*)dqkv, S, n_heads, scale, causal, drop_p, drop_site, seed);
    } else { dsvg_set_error("attention_long_bwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("attention_long_bwd");
    return 0;
}

extern "C" int dsvg_prefix_mean_fwd(int32_t dtype, const void* x, const int32_t* lens, void* out, int64_t n_seq, int32_t S,
                                    int32_t d, void* stream) {
    DSVG_CHECK_ARG(x && lens && out && n_seq > 0 && S > 0 && d > 0, "prefix_mean_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(prefix_mean_fwd_kernel<float>, dim3((unsigned)n_seq), dim3(256), 0, st, (const float*)x, lens,
                           (float*)out, S, d);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(prefix_mean_fwd_kernel<bf16_t>, dim3((unsigned)n_seq), dim3(256), 0, st, (const bf16_t*)x, lens,
                           (bf16_t*)out, S, d);
    else { dsvg_set_error("prefix_mean_fwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("prefix_mean_fwd");
    return 0;
}
extern "C" int dsvg_prefix_mean_bwd(int32_t dtype, const void* dout, const int32_t* lens, void* dx, int64_t n_seq, int32_t S,
                                    int32_t d, void* stream) {
    DSVG_CHECK_ARG(dout && lens && dx && n_seq > 0 && S > 0 && d > 0, "prefix_mean_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(prefix_mean_bwd_kernel<float>, dim3((unsigned)n_seq), dim3(256), 0, st, (const float*)dout, lens,
                           (float*)dx, S, d);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(prefix_mean_bwd_kernel<bf16_t>, dim3((unsigned)n_seq), dim3(256), 0, st, (const bf16_t*)dout, lens,
                           (bf16_t*)dx, S, d);
    else { dsvg_set_error("prefix_mean_bwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("prefix_mean_bwd");
    return 0;
}
