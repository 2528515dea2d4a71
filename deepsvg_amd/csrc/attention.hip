// Fused multi-head self-attention core for the short DeepSVG sequences (S <= 64, head_dim = 32).
// One workgroup owns one (sequence, head-group): the q|k|v slabs of the group are staged once in LDS
// with coalesced 16-byte row loads, every lane owns one (head, query row) pair, K/V rows are LDS
// broadcasts, the softmax row lives in registers, and no S x S tensor ever reaches HBM.  The backward
// kernel recomputes the probabilities (flash-style two passes: query-major for dQ, key-major for dK/dV).
// Replaces deepsvg/model/layers/functional.py:168,197-248 and its autograd backward.
//
// Row stride of the LDS image = 3*W elements + 16 bytes, i.e. == 4 dwords (mod 64): the 16 lanes of a
// ds_read_b128 group that read 16 different rows hit 16 disjoint 4-bank slots (conflict-free); lanes
// reading the same K/V row broadcast.
#include "dsvg_common.h"
#include "../../include/dsvg.h"

// bf16 MFMA variant for 17..32-token sequences (attention_mfma.hip)
bool dsvg_attention_mfma_ok(int32_t dtype, int32_t S, int32_t n_heads);
int dsvg_attention_fwd_mfma(const void* qkv, const uint64_t* key_mask, const int32_t* seq_off, int64_t total_rows,
                            const int32_t* tile_first, void* out, int64_t n_seq, int32_t S, int32_t n_heads,
                            float scale, float drop_p, uint32_t drop_site, const uint64_t* seed, hipStream_t st);
int dsvg_attention_bwd_mfma(const void* qkv, const uint64_t* key_mask, const int32_t* seq_off, int64_t total_rows,
                            const int32_t* tile_first, const void* dout, void* dqkv, int64_t n_seq, int32_t S,
                            int32_t n_heads, float scale, float drop_p, uint32_t drop_site, const uint64_t* seed,
                            hipStream_t st, const void* wo_packed_bwd = nullptr);
int dsvg_attn_pack_bwd_launch(const float* flat, const int64_t* offs, int n_layers, void* img, hipStream_t st);

template <typename T, int SP, int HG>
struct AttnCfg {
    static constexpr int HPW = 64 / SP;          // heads per wave
    static constexpr int NW = HG / HPW;          // waves per workgroup
    static constexpr int NT = NW * 64;
    static constexpr int W = HG * 32;            // columns of one q/k/v slab of the head group
    static constexpr int PAD = 16 / (int)sizeof(T);
    static constexpr int LD = 3 * W + PAD;       // row stride of the qkv image
    static constexpr int LDO = W + PAD;          // row stride of the dO image (backward)
};

template <typename T, int NT>
__device__ __forceinline__ void tile_copy_in(T* dst, int ld_dst, const T* src, long long ld_src, int rows, int cols) {
    typedef typename Elem<T>::raw4 raw4;
    const int cpr = cols / 4;
    for (int idx = threadIdx.x; idx < rows * cpr; idx += NT) {
        const int r = idx / cpr, c = idx % cpr;
        *reinterpret_cast<raw4*>(dst + r * ld_dst + 4 * c) = *reinterpret_cast<const raw4*>(src + r * ld_src + 4 * c);
    }
}
template <typename T, int NT>
__device__ __forceinline__ void tile_copy_out(T* dst, long long ld_dst, const T* src, int ld_src, int rows, int cols) {
    typedef typename Elem<T>::raw4 raw4;
    const int cpr = cols / 4;
    for (int idx = threadIdx.x; idx < rows * cpr; idx += NT) {
        const int r = idx / cpr, c = idx % cpr;
        *reinterpret_cast<raw4*>(dst + r * ld_dst + 4 * c) = *reinterpret_cast<const raw4*>(src + r * ld_src + 4 * c);
    }
}

// packed (variable-length) layout: sequence b owns rows [seq_off[b], seq_off[b+1]) of the token-major buffers and all
// of its keys are valid; the workgroup with blockIdx.x == n_seq zero-fills the pad rows [seq_off[n_seq], total_rows)
// of its column slab so that later reductions over rows (weight gradients) see finite values / exact zeros.
template <typename T, int NT>
__device__ __forceinline__ void zero_rows(T* dst, long long ld, long long row_begin, long long row_end, int cols) {
    typedef typename Elem<T>::raw4 raw4;
    const int cpr = cols / 4;
    raw4 z;
    memset(&z, 0, sizeof(z));
    for (long long idx = threadIdx.x; idx < (row_end - row_begin) * cpr; idx += NT) {
        const long long r = row_begin + idx / cpr;
        const int c = (int)(idx % cpr);
        *reinterpret_cast<raw4*>(dst + r * ld + 4 * c) = z;
    }
}

template <typename T, int SP, int HG>
__global__ __launch_bounds__(HG * SP) void attn_fwd_kernel(
    const T* __restrict__ qkv, const uint64_t* __restrict__ key_mask, const int32_t* __restrict__ seq_off,
    long long total_rows, T* __restrict__ out, int Smax, int H, float scale, float drop_p, uint32_t drop_site,
    const uint64_t* seed, int causal) {
    typedef AttnCfg<T, SP, HG> C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);
    const int b = blockIdx.x, hg = blockIdx.y;
    const int d = H * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = lane / SP, i = lane % SP;
    const int hh = wave * C::HPW + hl;      // head inside the group
    const int h = hg * HG + hh;             // global head
    long long row0 = (long long)b * Smax;
    int S = Smax;                           // this sequence's length (dropout ids keep the Smax-based numbering)
    if (total_rows > 0 && b == (int)gridDim.x - 1) {     // tail workgroup: rows past the last sequence <- 0
        zero_rows<T, C::NT>(out + (size_t)hg * C::W, (long long)d, seq_off ? (long long)seq_off[b] : row0, total_rows, C::W);
        return;
    }
    if (seq_off) {
        row0 = seq_off[b];
        S = seq_off[b + 1] - seq_off[b];
    }
    const T* src = qkv + (size_t)row0 * 3 * d + (size_t)hg * C::W;

    tile_copy_in<T, C::NT>(tile, C::LD, src, 3LL * d, S, C::W);                          // q slab
    tile_copy_in<T, C::NT>(tile + C::W, C::LD, src + d, 3LL * d, S, C::W);               // k slab
    tile_copy_in<T, C::NT>(tile + 2 * C::W, C::LD, src + 2 * d, 3LL * d, S, C::W);       // v slab
    __syncthreads();

    // causal (autoregressive decoder, square_subsequent_mask of deepsvg/model/model.py:219-222,270): query row i sees
    // the keys j <= i - folded into the lane's private key mask
    const uint64_t km_all = key_mask ? key_mask[b] : ~0ull;
    const uint64_t km = causal ? (km_all & (i >= 63 ? ~0ull : ((2ull << i) - 1ull))) : km_all;
    const DropCtx dc = drop_make(drop_p, seed, drop_site);
    const bool active = i < S;
    float o[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) o[c] = 0.f;

    if (active) {
        float q[32];
        row32_load(tile + i * C::LD + hh * 32, q);
#pragma unroll
        for (int c = 0; c < 32; ++c) q[c] *= scale;
        float s[SP];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            s[j] = -INFINITY;
            if (j < S) {
                float kr[32];
                row32_load(tile + j * C::LD + C::W + hh * 32, kr);
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) acc = fmaf(q[c], kr[c], acc);
                if ((km >> j) & 1ull) s[j] = acc;
                m = fmaxf(m, s[j]);
            }
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            s[j] = (j < S) ? __expf(s[j] - m) : 0.f;
            l += s[j];
        }
        const float inv = 1.f / l;
        const uint64_t drow = ((uint64_t)b * H + h) * Smax + i;       // dropout row of this (sequence, head, query)
        const uint32_t hr0 = attn_drop_row(dc, drow, 0), hr1 = SP > 32 ? attn_drop_row(dc, drow, 1) : 0u;
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            if (j < S) {
                const float pj = s[j] * inv * attn_drop_key(dc, j < 32 ? hr0 : hr1, j);
                float vr[32];
                row32_load(tile + j * C::LD + 2 * C::W + hh * 32, vr);
#pragma unroll
                for (int c = 0; c < 32; ++c) o[c] = fmaf(pj, vr[c], o[c]);
            }
        }
        // each (row, head) q slot is private to its lane: reuse it as the output staging slot
        row32_store(tile + i * C::LD + hh * 32, o);
    }
    __syncthreads();
    tile_copy_out<T, C::NT>(out + (size_t)row0 * d + (size_t)hg * C::W, (long long)d, tile, C::LD, S, C::W);
}

template <typename T, int SP, int HG>
__global__ __launch_bounds__(HG * SP) void attn_bwd_kernel(
    const T* __restrict__ qkv, const uint64_t* __restrict__ key_mask, const int32_t* __restrict__ seq_off,
    long long total_rows, const T* __restrict__ dout, T* __restrict__ dqkv, int Smax, int H, float scale, float drop_p,
    uint32_t drop_site, const uint64_t* seed, int causal) {
    typedef AttnCfg<T, SP, HG> C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);                         // [Smax][LD]   q|k|v
    T* dtile = tile + Smax * C::LD;                                   // [Smax][LDO]  dO
    float* stat = reinterpret_cast<float*>(dtile + Smax * C::LDO);    // [HG][SP][2]  lse, D
    const int b = blockIdx.x, hg = blockIdx.y;
    const int d = H * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = lane / SP, i = lane % SP;
    const int hh = wave * C::HPW + hl;
    const int h = hg * HG + hh;
    long long row0 = (long long)b * Smax;
    int S = Smax;
    if (total_rows > 0 && b == (int)gridDim.x - 1) {     // tail workgroup: rows past the last sequence <- 0
        const long long first = seq_off ? (long long)seq_off[b] : row0;
        T* z = dqkv + (size_t)hg * C::W;
        zero_rows<T, C::NT>(z, 3LL * d, first, total_rows, C::W);
        zero_rows<T, C::NT>(z + d, 3LL * d, first, total_rows, C::W);
        zero_rows<T, C::NT>(z + 2 * d, 3LL * d, first, total_rows, C::W);
        return;
    }
    if (seq_off) {
        row0 = seq_off[b];
        S = seq_off[b + 1] - seq_off[b];
    }
    const T* src = qkv + (size_t)row0 * 3 * d + (size_t)hg * C::W;

    tile_copy_in<T, C::NT>(tile, C::LD, src, 3LL * d, S, C::W);
    tile_copy_in<T, C::NT>(tile + C::W, C::LD, src + d, 3LL * d, S, C::W);
    tile_copy_in<T, C::NT>(tile + 2 * C::W, C::LD, src + 2 * d, 3LL * d, S, C::W);
    tile_copy_in<T, C::NT>(dtile, C::LDO, dout + (size_t)row0 * d + (size_t)hg * C::W, (long long)d, S, C::W);
    __syncthreads();

    const uint64_t km_all = key_mask ? key_mask[b] : ~0ull;
    // pass 1 (lane = query row i): causal folds "keys j <= i" into the lane's key mask; pass 2 (lane = key row j = i)
    // then only visits the query rows r >= j
    const uint64_t km = causal ? (km_all & (i >= 63 ? ~0ull : ((2ull << i) - 1ull))) : km_all;
    const DropCtx dc = drop_make(drop_p, seed, drop_site);
    const bool active = i < S;
    const uint64_t hbase = ((uint64_t)b * H + h) * Smax;   // dropout row of query i = hbase + i, key j counted in the sequence

    float dq[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) dq[c] = 0.f;

    // ---- pass 1: this lane is query row i --------------------------------------------------------------
    if (active) {
        float q[32], go[32];
        row32_load(tile + i * C::LD + hh * 32, q);
        row32_load(dtile + i * C::LDO + hh * 32, go);
#pragma unroll
        for (int c = 0; c < 32; ++c) q[c] *= scale;
        float s[SP];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            s[j] = -INFINITY;
            if (j < S) {
                float kr[32];
                row32_load(tile + j * C::LD + C::W + hh * 32, kr);
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) acc = fmaf(q[c], kr[c], acc);
                if ((km >> j) & 1ull) s[j] = acc;
                m = fmaxf(m, s[j]);
            }
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            s[j] = (j < S) ? __expf(s[j] - m) : 0.f;
            l += s[j];
        }
        const float inv = 1.f / l;
        const float lse = m + __logf(l);
        float dp[SP];
        float D = 0.f;
        const uint32_t hr0 = attn_drop_row(dc, hbase + i, 0), hr1 = SP > 32 ? attn_drop_row(dc, hbase + i, 1) : 0u;
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            dp[j] = 0.f;
            if (j < S) {
                float vr[32];
                row32_load(tile + j * C::LD + 2 * C::W + hh * 32, vr);
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) acc = fmaf(go[c], vr[c], acc);
                s[j] *= inv;                                                  // P_ij
                dp[j] = acc * attn_drop_key(dc, j < 32 ? hr0 : hr1, j);        // dP_ij
                D = fmaf(s[j], dp[j], D);
            }
        }
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            if (j < S) {
                const float ds = s[j] * (dp[j] - D) * scale;
                float kr[32];
                row32_load(tile + j * C::LD + C::W + hh * 32, kr);
#pragma unroll
                for (int c = 0; c < 32; ++c) dq[c] = fmaf(ds, kr[c], dq[c]);
            }
        }
        stat[(hh * SP + i) * 2 + 0] = lse;
        stat[(hh * SP + i) * 2 + 1] = D;
    }
    __syncthreads();

    // ---- pass 2: this lane is key row j = i --------------------------------------------------------------
    float dk[32], dv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
    if (active) {
        const int j = i;
        const bool kvalid = (km_all >> j) & 1ull;
        float kr[32], vr[32];
        row32_load(tile + j * C::LD + C::W + hh * 32, kr);
        row32_load(tile + j * C::LD + 2 * C::W + hh * 32, vr);
        if (kvalid) {
            for (int r = causal ? j : 0; r < S; ++r) {
                float q[32], go[32];
                row32_load(tile + r * C::LD + hh * 32, q);
                row32_load(dtile + r * C::LDO + hh * 32, go);
                float sacc = 0.f, dacc = 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) { sacc = fmaf(q[c], kr[c], sacc); dacc = fmaf(go[c], vr[c], dacc); }
                const float lse = stat[(hh * SP + r) * 2 + 0];
                const float D = stat[(hh * SP + r) * 2 + 1];
                const float mult = attn_drop_mult(dc, hbase + r, j);
                const float p = __expf(sacc * scale - lse);
                const float pd = p * mult;                                   // dropped probability used in O = P~ V
                const float ds = p * (dacc * mult - D) * scale;             // dS_rj * scale (q unscaled below)
#pragma unroll
                for (int c = 0; c < 32; ++c) { dv[c] = fmaf(pd, go[c], dv[c]); dk[c] = fmaf(ds, q[c], dk[c]); }
            }
        }
    }
    __syncthreads();   // every read of the q/k/v image is done: reuse it as the dq|dk|dv staging image
    if (active) {
        row32_store(tile + i * C::LD + hh * 32, dq);
        row32_store(tile + i * C::LD + C::W + hh * 32, dk);
        row32_store(tile + i * C::LD + 2 * C::W + hh * 32, dv);
    }
    __syncthreads();
    T* dst = dqkv + (size_t)row0 * 3 * d + (size_t)hg * C::W;
    tile_copy_out<T, C::NT>(dst, 3LL * d, tile, C::LD, S, C::W);
    tile_copy_out<T, C::NT>(dst + d, 3LL * d, tile + C::W, C::LD, S, C::W);
    tile_copy_out<T, C::NT>(dst + 2 * d, 3LL * d, tile + 2 * C::W, C::LD, S, C::W);
}

template <typename T, int SP, int HG>
static int launch_fwd(const void* qkv, const uint64_t* km, const int32_t* seq_off, int64_t total_rows, void* out,
                      int64_t n_seq, int S, int H, float scale, float drop_p, uint32_t site, const uint64_t* seed,
                      int causal, hipStream_t st) {
    typedef AttnCfg<T, SP, HG> C;
    const size_t lds = (size_t)S * C::LD * sizeof(T);
    if (lds > 160 * 1024) { dsvg_set_error("attention_fwd: LDS image too large (%zu B)", lds); return -1; }
    auto kern = attn_fwd_kernel<T, SP, HG>;
    DSVG_ENSURE_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_seq + (total_rows > 0 ? 1u : 0u), H / HG), dim3(C::NT), lds, st, (const T*)qkv,
                       km, seq_off, (long long)total_rows, (T*)out, S, H, scale, drop_p, site, seed, causal);
    DSVG_LAUNCH_CHECK("attention_fwd");
    return 0;
}
template <typename T, int SP, int HG>
static int launch_bwd(const void* qkv, const uint64_t* km, const int32_t* seq_off, int64_t total_rows, const void* dout,
                      void* dqkv, int64_t n_seq, int S, int H, float scale, float drop_p, uint32_t site,
                      const uint64_t* seed, int causal, hipStream_t st) {
    typedef AttnCfg<T, SP, HG> C;
    const size_t lds = (size_t)S * (C::LD + C::LDO) * sizeof(T) + (size_t)HG * SP * 2 * sizeof(float);
    if (lds > 160 * 1024) { dsvg_set_error("attention_bwd: LDS image too large (%zu B)", lds); return -1; }
    auto kern = attn_bwd_kernel<T, SP, HG>;
    DSVG_ENSURE_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_seq + (total_rows > 0 ? 1u : 0u), H / HG), dim3(C::NT), lds, st, (const T*)qkv,
                       km, seq_off, (long long)total_rows, (const T*)dout, (T*)dqkv, S, H, scale, drop_p, site, seed,
                       causal);
    DSVG_LAUNCH_CHECK("attention_bwd");
    return 0;
}

// head-group choice: all 8 heads per workgroup while the LDS image fits, 4 heads for the 64-row variant
#define DSVG_ATTN_DISPATCH(FN, T, ...)                                              \
    do {                                                                            \
        if (S <= 8 && n_heads % 8 == 0) return FN<T, 8, 8>(__VA_ARGS__);            \
        if (S <= 16 && n_heads % 8 == 0) return FN<T, 16, 8>(__VA_ARGS__);          \
        if (S <= 32 && n_heads % 8 == 0) return FN<T, 32, 8>(__VA_ARGS__);          \
        if (S <= 32 && n_heads % 2 == 0) return FN<T, 32, 2>(__VA_ARGS__);          \
        if (S <= 64 && n_heads % 4 == 0) return FN<T, 64, 4>(__VA_ARGS__);          \
        if (S <= 64) return FN<T, 64, 1>(__VA_ARGS__);                              \
    } while (0)

// Grouping of the packed sequences into attention tiles of at most 32 rows (the MFMA kernel's score tile):
// tile j = sequences tile_first[j] .. tile_first[j+1]-1; tile_first[n_tiles] = n_seq; tile_first[n_seq + 1] = n_tiles.
// The average packed encoder sequence has ~10 valid tokens, so a tile carries ~3 of them.  Greedy inside segments of 64
// consecutive sequences (a tile never crosses a segment boundary: ~5 % more tiles than the global greedy walk, which is
// inherently sequential): one wave per segment, the 64 offsets in the 64 lanes, a ballot finds the first sequence that
// no longer fits the open tile and a scalar lane read fetches the row it starts at; the per-segment lists are then
// compacted with a workgroup scan.
__global__ __launch_bounds__(1024) void attention_tiles_kernel(const int32_t* __restrict__ seq_off, int n_seq,
                                                               int max_rows, int32_t* __restrict__ tile_first,
                                                               int32_t* __restrict__ scratch) {
    __shared__ int cnt[1024];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_segs = (n_seq + 63) / 64;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int seg0 = 0; seg0 < n_segs; seg0 += 1024) {
        const int seg1 = min(n_segs, seg0 + 1024);
        cnt[threadIdx.x] = 0;
        __syncthreads();
        for (int sg = seg0 + wave; sg < seg1; sg += 16) {
            const int i = sg * 64 + lane;
            const bool valid = i < n_seq;
            const int o0 = valid ? seq_off[i] : 0;
            const int o1 = valid ? seq_off[i + 1] : 0;
            int start = -(1 << 30), pos = 0, c = 0;
            while (true) {
                const unsigned long long cand = __ballot(valid && lane >= pos && (o1 - start > max_rows));
                if (!cand) break;
                const int j = __builtin_ctzll(cand);            // first sequence that does not fit: it opens a tile
                if (lane == j) scratch[sg * 64 + c] = i;
                ++c;
                start = __builtin_amdgcn_readlane(o0, j);       // j is wave-uniform: a scalar read, not a permute
                pos = j + 1;
            }
            if (lane == 0) cnt[sg - seg0] = c;
        }
        __syncthreads();
        const int mine = cnt[threadIdx.x];
        for (int o = 1; o < 1024; o <<= 1) {                    // inclusive scan of the segment counts
            const int v = threadIdx.x >= o ? cnt[threadIdx.x - o] : 0;
            __syncthreads();
            cnt[threadIdx.x] += v;
            __syncthreads();
        }
        const int total = cnt[1023];
        const int excl = cnt[threadIdx.x] - mine;
        __syncthreads();
        cnt[threadIdx.x] = excl;
        __syncthreads();
        __threadfence_block();
        for (int sg = seg0 + wave; sg < seg1; sg += 16) {
            const int first = carry + cnt[sg - seg0];
            const int c = (sg - seg0 + 1 < 1024 ? cnt[sg - seg0 + 1] : total) - cnt[sg - seg0];
            for (int q = lane; q < c; q += 64) tile_first[first + q] = scratch[sg * 64 + q];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tile_first[carry] = n_seq;
        tile_first[n_seq + 1] = carry;
    }
}
extern "C" int dsvg_attention_tiles(const int32_t* seq_off, int64_t n_seq, int32_t max_rows, int32_t* tile_first,
                                    int32_t* scratch, void* stream) {
    DSVG_CHECK_ARG(seq_off && tile_first && scratch && n_seq > 0 && n_seq < (1 << 30) && max_rows > 0,
                   "attention_tiles: bad args");
    hipLaunchKernelGGL(attention_tiles_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, seq_off, (int)n_seq, max_rows,
                       tile_first, scratch);
    DSVG_LAUNCH_CHECK("attention_tiles");
    return 0;
}

extern "C" int dsvg_attention_fwd(int32_t dtype, const void* qkv, const uint64_t* key_mask, const int32_t* seq_off,
                                  int64_t total_rows, const int32_t* tile_first, void* out, int64_t n_seq, int32_t S,
                                  int32_t n_heads, float scale, float drop_p, uint32_t drop_site, const uint64_t* seed,
                                  void* stream) {
    DSVG_CHECK_ARG(qkv && out && n_seq > 0 && S > 0 && S <= 64 && n_heads > 0, "attention_fwd: bad args (S=%d)", S);
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "attention_fwd: dropout needs a seed pointer");
    DSVG_CHECK_ARG(!seq_off || (!key_mask && total_rows > 0), "attention_fwd: packed layout takes no key mask");
    DSVG_CHECK_ARG(seq_off || total_rows == 0 || total_rows >= n_seq * S, "attention_fwd: total_rows < n_seq * S");
    if (!seq_off && total_rows <= n_seq * S) total_rows = 0;       // dense layout without a tail
    hipStream_t st = (hipStream_t)stream;
    if (dsvg_attention_mfma_ok(dtype, S, n_heads))
        return dsvg_attention_fwd_mfma(qkv, key_mask, seq_off, total_rows, seq_off ? tile_first : nullptr, out, n_seq, S,
                                       n_heads, scale, drop_p, drop_site, seed, st);
    if (dtype == DSVG_F32) {
        DSVG_ATTN_DISPATCH(launch_fwd, float, qkv, key_mask, seq_off, total_rows, out, n_seq, S, n_heads, scale, drop_p,
                           drop_site, seed, 0, st);
    } else if (dtype == DSVG_BF16) {
        DSVG_ATTN_DISPATCH(launch_fwd, bf16_t, qkv, key_mask, seq_off, total_rows, out, n_seq, S, n_heads, scale, drop_p,
                           drop_site, seed, 0, st);
    }
    dsvg_set_error("attention_fwd: unsupported dtype/shape (dtype=%d S=%d H=%d)", dtype, S, n_heads);
    return -1;
}

// causal self-attention of the autoregressive decoder (dense layout; lane-per-query kernels for every dtype / length)
extern "C" int dsvg_attention_causal_fwd(int32_t dtype, const void* qkv, const uint64_t* key_mask, void* out,
                                         int64_t n_seq, int32_t S, int32_t n_heads, float scale, float drop_p,
                                         uint32_t drop_site, const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(qkv && out && n_seq > 0 && S > 0 && S <= 64 && n_heads > 0, "attention_causal_fwd: bad args (S=%d)", S);
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "attention_causal_fwd: dropout needs a seed pointer");
    hipStream_t st = (hipStream_t)stream;
    const int32_t* seq_off = nullptr;
    const int64_t total_rows = 0;
    if (dtype == DSVG_F32) {
        DSVG_ATTN_DISPATCH(launch_fwd, float, qkv, key_mask, seq_off, total_rows, out, n_seq, S, n_heads, scale, drop_p,
                           drop_site, seed, 1, st);
    } else if (dtype == DSVG_BF16) {
        DSVG_ATTN_DISPATCH(launch_fwd, bf16_t, qkv, key_mask, seq_off, total_rows, out, n_seq, S, n_heads, scale, drop_p,
                           drop_site, seed, 1, st);
    }
    dsvg_set_error("attention_causal_fwd: unsupported dtype/shape (dtype=%d S=%d H=%d)", dtype, S, n_heads);
    return -1;
}

extern "C" int dsvg_attention_bwd(int32_t dtype, const void* qkv, const uint64_t* key_mask, const int32_t* seq_off,
                                  int64_t total_rows, const int32_t* tile_first, const void* dout, void* dqkv,
                                  int64_t n_seq, int32_t S, int32_t n_heads, float scale, float drop_p,
                                  uint32_t drop_site, const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(qkv && dout && dqkv && n_seq > 0 && S > 0 && S <= 64 && n_heads > 0, "attention_bwd: bad args");
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "attention_bwd: dropout needs a seed pointer");
    DSVG_CHECK_ARG(!seq_off || (!key_mask && total_rows > 0), "attention_bwd: packed layout takes no key mask");
    DSVG_CHECK_ARG(seq_off || total_rows == 0 || total_rows >= n_seq * S, "attention_bwd: total_rows < n_seq * S");
    if (!seq_off && total_rows <= n_seq * S) total_rows = 0;       // dense layout without a tail
    hipStream_t st = (hipStream_t)stream;
    if (dsvg_attention_mfma_ok(dtype, S, n_heads))
        return dsvg_attention_bwd_mfma(qkv, key_mask, seq_off, total_rows, seq_off ? tile_first : nullptr, dout, dqkv,
                                       n_seq, S, n_heads, scale, drop_p, drop_site, seed, st);
    if (dtype == DSVG_F32) {
        DSVG_ATTN_DISPATCH(launch_bwd, float, qkv, key_mask, seq_off, total_rows, dout, dqkv, n_seq, S, n_heads, scale,
                           drop_p, drop_site, seed, 0, st);
    } else if (dtype == DSVG_BF16) {
        DSVG_ATTN_DISPATCH(launch_bwd, bf16_t, qkv, key_mask, seq_off, total_rows, dout, dqkv, n_seq, S, n_heads, scale,
                           drop_p, drop_site, seed, 0, st);
    }
    dsvg_set_error("attention_bwd: unsupported dtype/shape (dtype=%d S=%d H=%d)", dtype, S, n_heads);
    return -1;
}

// The same with the out_proj backward inside (8 heads of 32, sequences of 17 .. 32 rows or the packed tile layout): dx1m is the
// gradient of the block's projected output with the residual dropout mask on it, wo_packed_bwd one layer of dsvg_attn_pack_bwd
extern "C" int dsvg_attention_bwd_outproj(const void* qkv, const uint64_t* key_mask, const int32_t* seq_off,
                                          int64_t total_rows, const int32_t* tile_first, const void* dx1m,
                                          const void* wo_packed_bwd, void* dqkv, int64_t n_seq, int32_t S, float scale,
                                          float drop_p, uint32_t drop_site, const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(qkv && dx1m && wo_packed_bwd && dqkv && n_seq > 0, "attention_bwd_outproj: bad args");
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "attention_bwd_outproj: dropout needs a seed pointer");
    DSVG_CHECK_ARG(!seq_off || (!key_mask && total_rows > 0 && tile_first), "attention_bwd_outproj: the packed layout takes its tile list, no key mask");
    DSVG_CHECK_ARG(seq_off || (S > 16 && S <= 32), "attention_bwd_outproj: dense sequences of 17 .. 32 rows");
    DSVG_CHECK_ARG(dsvg_attention_mfma_ok(DSVG_BF16, S, 8), "attention_bwd_outproj: the MFMA attention kernels are switched off");
    DSVG_CHECK_ARG(seq_off || total_rows == 0 || total_rows >= n_seq * S, "attention_bwd_outproj: total_rows < n_seq * S");
    DSVG_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)dx1m | (uintptr_t)wo_packed_bwd | (uintptr_t)dqkv) & 15) == 0,
                   "attention_bwd_outproj: operands must be 16-byte aligned");
    if (!seq_off && total_rows <= n_seq * S) total_rows = 0;
    return dsvg_attention_bwd_mfma(qkv, key_mask, seq_off, total_rows, seq_off ? tile_first : nullptr, dx1m, dqkv, n_seq, S,
                                   8, scale, drop_p, drop_site, seed, (hipStream_t)stream, wo_packed_bwd);
}

extern "C" int64_t dsvg_attn_pack_bwd_elems(int32_t n_layers) { return (int64_t)n_layers * 512 * 512; }
extern "C" int dsvg_attn_pack_bwd(const float* flat_f32, const int64_t* offs, int32_t n_layers, void* packed_bwd, void* stream) {
    DSVG_CHECK_ARG(flat_f32 && offs && packed_bwd && n_layers > 0 && ((uintptr_t)packed_bwd & 15) == 0, "attn_pack_bwd: bad args");
    return dsvg_attn_pack_bwd_launch(flat_f32, offs, n_layers, packed_bwd, (hipStream_t)stream);
}

extern "C" int dsvg_attention_causal_bwd(int32_t dtype, const void* qkv, const uint64_t* key_mask, const void* dout,
                                         void* dqkv, int64_t n_seq, int32_t S, int32_t n_heads, float scale,
                                         float drop_p, uint32_t drop_site, const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(qkv && dout && dqkv && n_seq > 0 && S > 0 && S <= 64 && n_heads > 0, "attention_causal_bwd: bad args");
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "attention_causal_bwd: dropout needs a seed pointer");
    hipStream_t st = (hipStream_t)stream;
    const int32_t* seq_off = nullptr;
    const int64_t total_rows = 0;
    if (dtype == DSVG_F32) {
        DSVG_ATTN_DISPATCH(launch_bwd, float, qkv, key_mask, seq_off, total_rows, dout, dqkv, n_seq, S, n_heads, scale,
                           drop_p, drop_site, seed, 1, st);
    } else if (dtype == DSVG_BF16) {
        DSVG_ATTN_DISPATCH(launch_bwd, bf16_t, qkv, key_mask, seq_off, total_rows, dout, dqkv, n_seq, S, n_heads, scale,
                           drop_p, drop_site, seed, 1, st);
    }
    dsvg_set_error("attention_causal_bwd: unsupported dtype/shape (dtype=%d S=%d H=%d)", dtype, S, n_heads);
    return -1;
}
