// Per-step weight images of a bf16 model: fp32 master parameters -> the bf16 flat copy and the MFMA fragment images of the
// fused kernels.  Every image is a pure function of the flat parameter buffer, one thread per 16-byte lane slot; the
// bodies live here so that the stand-alone launches (dsvg_ffn_pack, dsvg_attn_pack, dsvg_attn_pack_bwd, dsvg_gs_pack,
// dsvg_cast_weights) and the one-launch refresh of a training step (dsvg_pack_images, pack_images.hip) run the same code.
// Layouts are documented where the consumers live (ffn_fused.hip, attn_fused.hip, attention_mfma.hip, group_stage.hip).
#pragma once
#include "dsvg_common.h"

namespace dsvg_pack {

constexpr int D = 256;                  // d_model
constexpr int F = 512;                  // dim_feedforward
constexpr int H = 8;                    // heads
constexpr int FRAG_BYTES = 1024;        // bytes per packed MFMA fragment (64 lanes x 16 B)

__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]), f2bf_pk(v[4], v[5]), f2bf_pk(v[6], v[7]));
}

// the register order of a transposed 32 x 32 MFMA tile: value r of lane half h2 belongs to row / column rowmap(r, h2)
__host__ __device__ inline int rowmap(int r, int h2) { return (r & 3) + 8 * (r >> 2) + 4 * h2; }

// ---- fused FFN (ffn_fused.hip) ---------------------------------------------------------------------------------------
constexpr int FFN_CH = 32;              // hidden units per chunk
constexpr int FFN_NCH = F / FFN_CH;     // 16
constexpr int FFN_FWD_CHUNK = 32 * FRAG_BYTES;    // [W1 chunk: 16 fragments | W2 chunk: 16 fragments]
constexpr int FFN_BWD_CHUNK = 48 * FRAG_BYTES;    // [W1 chunk | W2^T chunk | W1^T chunk]
constexpr int FFN_SLOTS = FFN_NCH * 80 * 64;        // lane slots per layer: 32 forward + 48 backward fragments per chunk

// hidden unit (inside its chunk) that K slot (ks2, half, e) of GEMM 2 carries = the unit accumulator register
// r = 8 ks2 + e of GEMM 1's transposed tile holds in lane half `half`
__host__ __device__ inline int hidden_of(int ks2, int half, int e) { return (e & 3) + 8 * (2 * ks2 + (e >> 2)) + 4 * half; }
// position of hidden unit j in the fragment-ordered h / dpre matrices (and back: an involution): bits 2 and 3 swapped
__host__ __device__ inline int frag_pos(int j) { return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1); }

// The affine part of the LayerNorm in front of linear1 is folded into it:
//     linear1(gamma * xh + beta) = (W1 diag(gamma)) xh + (b1 + W1 beta) = W1' xh + b1'
// so the kernels only normalise (xh = (x - mean) * rstd) and never touch gamma / beta; the backward kernel gets the
// gradient with respect to xh straight from W1'^T, and dW1 / dgamma / dbeta are finished from G = dpre^T xh and
// db1 = sum_t dpre by dsvg_ffn_wgrad_finish (dW1 = G diag(gamma) + db1 beta^T, dgamma = colsum(W1 * G), dbeta = W1^T db1).
// offs[layer][0..4] = element offsets of linear1.weight, linear1.bias, linear2.weight, norm.weight, norm.bias in `flat`.
__device__ __forceinline__ void ffn_slot(long long gid, const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                         int n_layers, bf16_t* __restrict__ fwd, bf16_t* __restrict__ bwd) {
    if (gid >= (long long)n_layers * FFN_SLOTS) return;
    const int layer = (int)(gid / FFN_SLOTS);
    int s = (int)(gid % FFN_SLOTS);
    const int l = s & 63; s >>= 6;
    const int f = s % 80, c = s / 80;
    const int i = l & 31, half = l >> 5;
    const float* W1 = flat + offs[layer * 5 + 0];     // [512, 256] row-major
    const float* W2 = flat + offs[layer * 5 + 2];     // [256, 512] row-major
    const float* ga = flat + offs[layer * 5 + 3];
    float v[8];
    bf16_t* dst;
    if (f < 32) {
        dst = fwd + ((size_t)layer * FFN_NCH + c) * (FFN_FWD_CHUNK / 2) + (size_t)f * 512 + l * 8;
        if (f < 16) {               // W1' chunk, K step f: A[i = hidden 32 c + i][k = 16 f + 8 half + e]
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int k = 16 * f + 8 * half + e; v[e] = W1[(size_t)(FFN_CH * c + i) * D + k] * ga[k]; }
        } else {                    // W2 chunk, output tile t, K step ks2: A[i = out 32 t + i][k -> hidden_of(ks2, half, e)]
            const int t = (f - 16) >> 1, ks2 = (f - 16) & 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = W2[(size_t)(32 * t + i) * F + FFN_CH * c + hidden_of(ks2, half, e)];
        }
    } else {
        const int g = f - 32;
        dst = bwd + ((size_t)layer * FFN_NCH + c) * (FFN_BWD_CHUNK / 2) + (size_t)g * 512 + l * 8;
        if (g < 16) {               // W1' chunk again (recomputation of the hidden tile)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int k = 16 * g + 8 * half + e; v[e] = W1[(size_t)(FFN_CH * c + i) * D + k] * ga[k]; }
        } else if (g < 32) {        // W2^T chunk, K step ks over the 256 outputs: A[i = hidden 32 c + i][k = out 16 ks + 8 half + e]
            const int ks = g - 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = W2[(size_t)(16 * ks + 8 * half + e) * F + FFN_CH * c + i];
        } else {                    // W1'^T chunk, d tile t, K step ks2 over the chunk's hidden units (same K order as W2's)
            const int t = (g - 32) >> 1, ks2 = (g - 32) & 1;
            const float gd = ga[32 * t + i];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = W1[(size_t)(FFN_CH * c + hidden_of(ks2, half, e)) * D + 32 * t + i] * gd;
        }
    }
    *reinterpret_cast<uint4*>(dst) = pack8(v);
}

// w2p[layer][o][p] = bf16(W2[o][frag_pos(p)]): linear2.weight with fragment-ordered columns, a plain row-major matrix for
// the unfused input-gradient GEMM (dpre = dym . W2p, gated by the fragment-ordered h); one thread per 8 output elements
__device__ __forceinline__ void ffn_w2p_slot(long long gid, const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                             int n_layers, bf16_t* __restrict__ w2p) {
    if (gid >= (long long)n_layers * (D * F / 8)) return;
    const int layer = (int)(gid / (D * F / 8));
    const int r = (int)(gid % (D * F / 8));
    const int o = r / (F / 8), p0 = (r % (F / 8)) * 8;
    const float* W2 = flat + offs[layer * 5 + 2] + (size_t)o * F;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = W2[frag_pos(p0 + e)];
    *reinterpret_cast<uint4*>(w2p + (size_t)layer * D * F + (size_t)o * F + p0) = pack8(v);
}

// b1'[layer][j] = b1[j] + sum_k W1[j][k] beta[k]: one wave per hidden unit (row = layer * 512 + j)
__device__ __forceinline__ void ffn_fold_bias_row(int row, int lane, const float* __restrict__ flat,
                                                  const int64_t* __restrict__ offs, int n_layers, float* __restrict__ b1f) {
    if (row >= n_layers * F) return;
    const int layer = row / F, j = row % F;
    const float* W1 = flat + offs[layer * 5 + 0] + (size_t)j * D;
    const float* be = flat + offs[layer * 5 + 4];
    float s = 0.f;
#pragma unroll
    for (int k = lane; k < D; k += 64) s += W1[k] * be[k];
    s = wave_sum(s);
    if (lane == 0) b1f[row] = flat[offs[layer * 5 + 1] + j] + s;
}

// ---- fused attention forward (attn_fused.hip) ------------------------------------------------------------------------
constexpr int ATTN_IMG_FRAGS = 512;     // 384 in_proj + 128 out_proj fragments per layer

// offs[layer][0..1] = element offsets of in_proj_weight [768, 256] and out_proj.weight [256, 256] in `flat`.
// Fragment f of a layer, lane l = (i = l & 31, half = l >> 5), slot e:
//   f = 48 h + 16 sel + ks  (sel = 0 q, 1 k, 2 v):  Win[256 sel + 32 h + i][16 ks + 8 half + e]
//   f = 384 + 16 t + 2 h + ks2:                      Wo[32 t + i][32 h + rowmap(8 ks2 + e, half)]
__device__ __forceinline__ void attn_slot(long long gid, const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                          int n_layers, bf16_t* __restrict__ img) {
    if (gid >= (long long)n_layers * ATTN_IMG_FRAGS * 64) return;
    const int layer = (int)(gid / (ATTN_IMG_FRAGS * 64));
    const int s = (int)(gid % (ATTN_IMG_FRAGS * 64));
    const int l = s & 63, f = s >> 6;
    const int i = l & 31, half = l >> 5;
    float v[8];
    if (f < 384) {
        const float* Win = flat + offs[layer * 2 + 0];
        const int h = f / 48, g = f % 48, sel = g >> 4, ks = g & 15;
        const float* row = Win + (size_t)(256 * sel + 32 * h + i) * D + 16 * ks + 8 * half;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = row[e];
    } else {
        const float* Wo = flat + offs[layer * 2 + 1];
        const int g = f - 384, t = g >> 4, h = (g & 15) >> 1, ks2 = g & 1;
        const float* row = Wo + (size_t)(32 * t + i) * D + 32 * h;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = row[rowmap(8 * ks2 + e, half)];
    }
    *reinterpret_cast<uint4*>(img + ((size_t)layer * ATTN_IMG_FRAGS + f) * 512 + l * 8) = pack8(v);
}

// ---- attention backward (attention_mfma.hip, attn_bwd_dx.hip) ---------------------------------------------------------
// fragments 0 .. 127: packed_bwd[layer][head 8][K step 16][lane l][e] = Wo[16 ks + 8 (l >> 5) + e][32 head + (l & 31)]: the A
// fragments of dO^T = Wo^T-columns x dx1m^T per head (offs[layer][1] = element offset of out_proj.weight [256, 256] in `flat`)
// fragments 128 .. 511 (round 6): in_proj_weight^T as the K-step-major A fragments of dxn1^T = Win^T x dqkv^T
// (attn_bwd_dx_kernel): fragment 128 + 16 c + 2 t + ks2 (K step c of 32 = 24 steps, output tile t of 32 columns),
// lane l = (i = l & 31, half = l >> 5), slot e:  Win[32 c + 16 ks2 + 8 half + e][32 t + i]   (offs[layer][0] = in_proj_weight)
constexpr int ATTN_BWD_FRAGS = 512;
constexpr int ATTN_BWD_WO_FRAGS = 128;
constexpr int ATTN_BWD_SLOTS = ATTN_BWD_FRAGS * 64;
__device__ __forceinline__ void attn_bwd_slot(long long gid, const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                              int n_layers, bf16_t* __restrict__ img) {
    if (gid >= (long long)n_layers * ATTN_BWD_SLOTS) return;
    const int layer = (int)(gid / ATTN_BWD_SLOTS);
    const int s = (int)(gid % ATTN_BWD_SLOTS);
    const int l = s & 63, f = s >> 6;
    uint32_t w[4];
    if (f < ATTN_BWD_WO_FRAGS) {
        const int hh = f >> 4, ks = f & 15;
        const float* Wo = flat + offs[layer * 2 + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 16 * ks + 8 * (l >> 5) + 2 * e;
            w[e] = f2bf_pk(Wo[(size_t)k * 256 + 32 * hh + (l & 31)], Wo[(size_t)(k + 1) * 256 + 32 * hh + (l & 31)]);
        }
    } else {
        const int g = f - ATTN_BWD_WO_FRAGS, c = g >> 4, t = (g & 15) >> 1, ks2 = g & 1;
        const float* Win = flat + offs[layer * 2 + 0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 32 * c + 16 * ks2 + 8 * (l >> 5) + 2 * e;
            w[e] = f2bf_pk(Win[(size_t)k * 256 + 32 * t + (l & 31)], Win[(size_t)(k + 1) * 256 + 32 * t + (l & 31)]);
        }
    }
    *reinterpret_cast<uint4*>(img + gid * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- group-stage layer kernels (group_stage.hip) ----------------------------------------------------------------------
constexpr int GS_FRAGS = 128;           // weight fragments per wave and layer (both directions)
constexpr long long GS_SLOTS = 2ll * H * GS_FRAGS * 64;     // lane slots per layer (forward + backward image)

// bf16 MFMA A fragments, wave-major, in consumption order.
// offs[layer][0..3] = element offsets of in_proj_weight [768,256], out_proj.weight [256,256], linear1.weight [512,256],
// linear2.weight [256,512] in `flat`.  Fragment i of wave w, lane l = (row = l & 31, half = l >> 5), slot e; k = 16 ks + 8
// half + e is always the NATURAL index of the reduced dimension (the B operands come from row-major images):
//   forward image                                      backward image (the transposed products)
//   i <  48: Win[256 (i%3) + 32 w + row][k], ks = i/3   i <  32: W2[k][64 w + 32 (i&1) + row],  ks = i>>1   (dh  = dym . W2)
//   i <  64: Wo [32 w + row][k],            ks = i-48   i <  64: W1[k][32 w + row],             ks = i-32   (dxn2 = dpre . W1)
//   i <  96: W1 [64 w + 32 (i&1) + row][k], ks = (i-64)>>1   i <  80: Wo[k][32 w + row],         ks = i-64   (dao = dx1m . Wo)
//   i < 128: W2 [32 w + row][k],            ks = i-96   i < 128: Win[k][32 w + row],            ks = i-80   (dxn1 = dqkv . Win)
__device__ __forceinline__ void gs_slot(long long gid, const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                        int n_layers, bf16_t* __restrict__ fwd, bf16_t* __restrict__ bwd) {
    if (gid >= (long long)n_layers * GS_SLOTS) return;
    const int layer = (int)(gid / GS_SLOTS);
    int s = (int)(gid % GS_SLOTS);
    const int dir = s / (H * GS_FRAGS * 64);
    s %= H * GS_FRAGS * 64;
    const int l = s & 63, i = (s >> 6) % GS_FRAGS, w = s / (64 * GS_FRAGS);
    const int row = l & 31, half = l >> 5;
    const float* Win = flat + offs[layer * 4 + 0];
    const float* Wo = flat + offs[layer * 4 + 1];
    const float* W1 = flat + offs[layer * 4 + 2];
    const float* W2 = flat + offs[layer * 4 + 3];
    float v[8];
    if (dir == 0) {
        const float* src;
        if (i < 48) src = Win + (size_t)(256 * (i % 3) + 32 * w + row) * D + 16 * (i / 3) + 8 * half;
        else if (i < 64) src = Wo + (size_t)(32 * w + row) * D + 16 * (i - 48) + 8 * half;
        else if (i < 96) src = W1 + (size_t)(64 * w + 32 * (i & 1) + row) * D + 16 * ((i - 64) >> 1) + 8 * half;
        else src = W2 + (size_t)(32 * w + row) * F + 16 * (i - 96) + 8 * half;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e];
    } else {
        const float* src;
        size_t ld;
        if (i < 32) { src = W2 + (size_t)(16 * (i >> 1) + 8 * half) * F + 64 * w + 32 * (i & 1) + row; ld = F; }
        else if (i < 64) { src = W1 + (size_t)(16 * (i - 32) + 8 * half) * D + 32 * w + row; ld = D; }
        else if (i < 80) { src = Wo + (size_t)(16 * (i - 64) + 8 * half) * D + 32 * w + row; ld = D; }
        else { src = Win + (size_t)(16 * (i - 80) + 8 * half) * D + 32 * w + row; ld = D; }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(size_t)e * ld];
    }
    bf16_t* dst = (dir == 0 ? fwd : bwd) + (((size_t)layer * H + w) * GS_FRAGS + i) * 512 + l * 8;
    *reinterpret_cast<uint4*>(dst) = pack8(v);
}

// ---- the bf16 copy of the whole flat parameter buffer: group i = 8 consecutive elements, one 16-byte store -----------
__device__ __forceinline__ void cast8(long long i, const float* __restrict__ src, bf16_t* __restrict__ dst) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i];
    const float4 b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    reinterpret_cast<uint4*>(dst)[i] = make_uint4(f2bf_pk(a.x, a.y), f2bf_pk(a.z, a.w), f2bf_pk(b.x, b.y), f2bf_pk(b.z, b.w));
}

// ---- the step counter and the dropout seed of a training step (splitmix64) --------------------------------------------
__device__ __forceinline__ void advance(long long* counter, uint64_t* seed) {
    if (counter) *counter += 1;
    if (seed) {
        uint64_t s = *seed;
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z = z ^ (z >> 31);
        *seed = z;
    }
}

}  // namespace dsvg_pack
