// MFMA attention core for the bf16 path, sequences of 17..32 tokens, head_dim 32 (the two big stacks of the
// hierarchical model: S = 32 encoder, S = 31 decoder).  One 512-thread workgroup per sequence, one wave per head;
// the q|k|v slabs of the 8 heads are staged once in LDS (rows >= S zero-filled: 0 * garbage would be NaN in an
// MFMA), every matrix product of the head is 2 x v_mfma_f32_32x32x16_bf16:
//
//   forward   St = K Q^T   (A = K rows, B = Q rows)  -> lane (q = l&31, h2 = l>>5) holds S[q][key(r,h2)], r<16
//             softmax over keys = 16 local registers + one exchange with lane^32
//             O^T = V^T P^T (A = V^T by ds_read_b64_tr_b16, B = the lane's own P registers, key order permuted
//                            identically on both operands)  -> lane holds O[q][d(r,h2)]
//   backward  pass A (same orientation):  P, dP = V dO^T, D_q = sum_k P dP, dS;  dQ^T = K^T dS^T
//             pass B (operands swapped):  S2 = Q K^T, dP2 = dO V^T -> lane (key, h2) holds [q(r,h2)][key];
//                                          dK^T = Q^T dS2, dV^T = dO^T P~2      (lse_q, D_q through LDS)
//   with key(r,h2) = q(r,h2) = d(r,h2) = (r&3) + 8*(r>>2) + 4*h2  (the 32x32 MFMA C layout).
//
// ~20 MFMAs and a few hundred VALU ops per head instead of ~7000 FMAs per lane in the VALU kernel
// (attention.hip, kept for fp32 and for the 8-token group stages); the kernel is HBM-bound.
// Replaces deepsvg/model/layers/functional.py:168,197-248 and its autograd backward.
#include "dsvg_common.h"
#include "pack_images.h"
#include "../../include/dsvg.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int HG = 8;            // heads (= waves) per workgroup
constexpr int W = HG * 32;       // columns of one q/k/v slab
constexpr int LD = 3 * W + 8;    // row stride of the qkv image (elements); 388 dwords == 4 (mod 64)
constexpr int LDO = W + 8;       // row stride of the dO image

union F8 {
    bf16x8 v;
    shortx4 h[2];
    uint4 u;
};

__device__ __forceinline__ int rowmap(int r, int h2) { return (r & 3) + 8 * (r >> 2) + 4 * h2; }

__device__ __forceinline__ bf16x8 row_frag(const bf16_t* img, int ld, int row, int col0, int step, int h2) {
    F8 f;
    f.u = *reinterpret_cast<const uint4*>(&img[row * ld + col0 + 16 * step + 8 * h2]);
    return f.v;
}
// A[i = column c (lane&31)][k-slot e] = img[row(8*ks + e, h2)][col0 + c]: two hardware-transposed 4x16 reads
__device__ __forceinline__ bf16x8 col_frag(const bf16_t* img, int ld, int col0, int ks, int lane) {
    const int g = lane >> 4, q16 = lane & 15;
    const int row = 16 * ks + 4 * (g >> 1) + (q16 >> 2);
    const int col = col0 + 16 * (g & 1) + 4 * (q16 & 3);
    F8 f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(&img[row * ld + col]));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(&img[(row + 8) * ld + col]));
    return f.v;
}
__device__ __forceinline__ bf16x8 pack_regs(const float (&p)[16], int ks) {
    F8 f;
    f.u = make_uint4(f2bf_pk(p[8 * ks + 0], p[8 * ks + 1]), f2bf_pk(p[8 * ks + 2], p[8 * ks + 3]),
                     f2bf_pk(p[8 * ks + 4], p[8 * ks + 5]), f2bf_pk(p[8 * ks + 6], p[8 * ks + 7]));
    return f.v;
}
// lane holds out[row = lane&31][d = rowmap(r, h2)]: four 8-byte pieces per lane into the staging image
__device__ __forceinline__ void stage_rows(bf16_t* img, int ld, int row, int col0, int h2, const floatx16& v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint2 t;
        t.x = f2bf_pk(v[4 * c + 0], v[4 * c + 1]);
        t.y = f2bf_pk(v[4 * c + 2], v[4 * c + 3]);
        *reinterpret_cast<uint2*>(&img[row * ld + col0 + 8 * c + 4 * h2]) = t;
    }
}

// 32 rows x W columns of up to four slabs -> LDS, 16-byte pieces, rows [S,32) zero-filled.  W = 256: 1024 pieces per
// slab = 2 per thread.  All loads are issued before the first use, from row-clamped addresses with the zero selected
// afterwards: a load under `if (r < S)` is compiled as a branch + s_waitcnt vmcnt(0) per piece, i.e. 6-8 dependent
// HBM round trips at the start of every workgroup.
struct SlabSrc {
    const bf16_t* src;
    long long ld;
    bf16_t* dst;
    int ld_dst;
};
template <int NS>
__device__ __forceinline__ void load_slabs(const SlabSrc (&sl)[NS], int S) {
    constexpr int CPR = W / 8;                 // 16-byte pieces per row
    constexpr int PER = 32 * CPR / 512;        // pieces per thread and slab
    uint4 v[NS][PER];
    const int rmax = S > 0 ? S - 1 : 0;
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = threadIdx.x + 512 * k;
            const int r = idx / CPR, c = idx % CPR;
            v[n][k] = *reinterpret_cast<const uint4*>(sl[n].src + (long long)min(r, rmax) * sl[n].ld + 8 * c);
        }
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = threadIdx.x + 512 * k;
            const int r = idx / CPR, c = idx % CPR;
            *reinterpret_cast<uint4*>(sl[n].dst + r * sl[n].ld_dst + 8 * c) = r < S ? v[n][k] : make_uint4(0u, 0u, 0u, 0u);
        }
}
__device__ __forceinline__ void store_slab(bf16_t* dst, long long ld_dst, const bf16_t* src, int ld_src, int S, int cols) {
    const int cpr = cols / 8;
    for (int idx = threadIdx.x; idx < S * cpr; idx += 512) {
        const int r = idx / cpr, c = idx % cpr;
        *reinterpret_cast<uint4*>(dst + r * ld_dst + 8 * c) = *reinterpret_cast<const uint4*>(src + r * ld_src + 8 * c);
    }
}

// pad rows [row_begin, row_end) of one column slab <- 0 (packed layout, see attention.hip)
__device__ __forceinline__ void zero_slab_rows(bf16_t* dst, long long ld, long long row_begin, long long row_end, int cols) {
    const int cpr = cols / 8;
    for (long long idx = threadIdx.x; idx < (row_end - row_begin) * cpr; idx += 512) {
        const long long r = row_begin + idx / cpr;
        *reinterpret_cast<uint4*>(dst + r * ld + 8 * (int)(idx % cpr)) = make_uint4(0u, 0u, 0u, 0u);
    }
}

// MODE 0: dense, one sequence of 17..32 rows per workgroup.  MODE 1: packed layout, tiles of whole sequences from
// dsvg_attention_tiles.  MODE 2: dense layout with short sequences (<= 16 rows): 32 / Smax whole sequences per tile, an
// optional key mask per sequence (the 8-token group stages: 4 icons per tile instead of the lane-per-query VALU kernel).
template <int MODE>
__global__ __launch_bounds__(512) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv,
                                                            const uint64_t* __restrict__ key_mask,
                                                            const int32_t* __restrict__ seq_off, long long total_rows,
                                                            const int32_t* __restrict__ tile_first, int n_seq,
                                                            bf16_t* __restrict__ out, int Smax, int H, float scale,
                                                            float drop_p, uint32_t drop_site, const uint64_t* seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);   // [32][LD]
    int* soff = reinterpret_cast<int*>(tile + 32 * LD);   // [34] sequence offsets of the tile (TILED)
    const int b = blockIdx.x, hg = blockIdx.y, d = H * 32;
    const int lane = threadIdx.x & 63, hh = threadIdx.x >> 6;
    const int li = lane & 31, h2 = lane >> 5;
    const int h = hg * HG + hh;
    constexpr bool TILED = MODE != 0;
    long long row0 = (long long)b * Smax;
    int S = Smax;                   // rows of this workgroup's tile; dropout ids keep the Smax-based numbering
    if (total_rows > 0 && b == (int)gridDim.x - 1) {     // tail workgroup: rows past the last sequence <- 0
        zero_slab_rows(out + (size_t)hg * W, (long long)d,
                       seq_off ? (long long)seq_off[n_seq] : (MODE == 2 ? (long long)n_seq * Smax : row0), total_rows, W);
        return;
    }
    // tiles (MODE 1: dsvg_attention_tiles; MODE 2: 32 / Smax consecutive dense sequences): the workgroup owns the sequences
    // s_first .. s_last - 1, at most 32 rows in total, block-diagonal attention inside the 32x32 score tile
    int s_first = b, s_last = b + 1;
    if (MODE == 1) {
        if (b >= tile_first[n_seq + 1]) return;
        s_first = tile_first[b];
        s_last = tile_first[b + 1];
        if ((int)threadIdx.x <= s_last - s_first) soff[threadIdx.x] = seq_off[s_first + threadIdx.x];   // <= 33 entries
    } else if (MODE == 2) {
        const int per = 32 / Smax;
        s_first = b * per;
        s_last = min(n_seq, s_first + per);
        if ((int)threadIdx.x <= s_last - s_first) soff[threadIdx.x] = (s_first + (int)threadIdx.x) * Smax;
        row0 = (long long)s_first * Smax;
        S = (s_last - s_first) * Smax;
    }
    if (seq_off) {
        row0 = seq_off[s_first];
        S = seq_off[s_last] - seq_off[s_first];
    }
    const bf16_t* src = qkv + (size_t)row0 * 3 * d + (size_t)hg * W;
    {
        const SlabSrc sl[3] = {{src, 3LL * d, tile, LD}, {src + d, 3LL * d, tile + W, LD}, {src + 2 * d, 3LL * d, tile + 2 * W, LD}};
        load_slabs<3>(sl, S);
    }
    __syncthreads();
    // this lane's row (query in pass A, key in pass B): its sequence, first row of that sequence in the tile, length
    int my_seq = b, my_start = 0, my_len = S;
    if (TILED) {
        int qi = 0;
        for (int q = 1; q < s_last - s_first; ++q)
            if (row0 + li >= soff[q]) qi = q;
        my_seq = s_first + qi;
        my_start = soff[qi] - (int)row0;
        my_len = soff[qi + 1] - soff[qi];
    }

    // keys visible to this lane's query: the rows of its own sequence (MODE 2: those its key mask lets through)
    const uint32_t km = MODE == 2 ? (((key_mask ? (uint32_t)key_mask[my_seq] : ~0u) & (uint32_t)((1ull << my_len) - 1ull)) << my_start)
                        : TILED ? (uint32_t)(((1ull << my_len) - 1ull) << my_start)
                                : ((key_mask ? (uint32_t)key_mask[b] : ~0u) & (uint32_t)((1ull << S) - 1ull));
    const DropCtx dc = drop_make(drop_p, seed, drop_site);
    const int qc = hh * 32, kc = W + hh * 32, vc = 2 * W + hh * 32;

    floatx16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int step = 0; step < 2; ++step)
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tile, LD, li, kc, step, h2),
                                                     row_frag(tile, LD, li, qc, step, h2), st, 0, 0, 0);
    // st[r] = q_li . k_key(r,h2)
    float p[16];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = rowmap(r, h2);
        p[r] = ((km >> key) & 1u) ? st[r] * scale : -INFINITY;
        m = fmaxf(m, p[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        p[r] = (p[r] == -INFINITY) ? 0.f : __expf(p[r] - m);
        l += p[r];
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    // dropout row of (sequence s, head h, query i) = (s H + h) Smax + i; query and key counted inside the sequence
    const uint32_t hrow = attn_drop_row(dc, ((uint64_t)my_seq * H + h) * Smax + (li - my_start), 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = p[r] * inv * attn_drop_key(dc, hrow, (uint32_t)(rowmap(r, h2) - my_start));
    if (li >= S) {
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = 0.f;     // padded query row: keep NaNs of an all-masked row out of the MFMA
    }
    floatx16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(tile, LD, vc, ks, lane), pack_regs(p, ks), ot, 0, 0, 0);
    // ot[r] = O[q = li][d = rowmap(r,h2)]; the head's q slab is dead: reuse it as the output staging slab
    stage_rows(tile, LD, li, qc, h2, ot);
    __syncthreads();
    store_slab(out + (size_t)row0 * d + (size_t)hg * W, (long long)d, tile, LD, S, W);
}

// FO (8 heads, d_model 256): `dout` is the gradient of the attention block's OUTPUT PROJECTION result with the residual
// dropout mask already on it (dx1m [rows, 256]) and wo_img the packed out_proj weight (dsvg_attn_pack_bwd: A fragments of
// Wo^T per head): the head-output gradient dO = dx1m . Wo is formed per tile on chip - each wave 16 MFMAs for its head's 32
// columns, B operand = the staged dx1m rows - instead of by a separate GEMM launch that writes and re-reads it.
template <int MODE, bool FO>
__global__ __launch_bounds__(512, 4) void attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv,
                                                            const uint64_t* __restrict__ key_mask,
                                                            const int32_t* __restrict__ seq_off, long long total_rows,
                                                            const int32_t* __restrict__ tile_first, int n_seq,
                                                            const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                            int Smax, int H, float scale, float drop_p,
                                                            uint32_t drop_site, const uint64_t* seed,
                                                            const bf16_t* __restrict__ wo_img) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);          // [32][LD]   q|k|v
    bf16_t* dtile = tile + 32 * LD;                              // [32][LDO]  dO
    float* stat = reinterpret_cast<float*>(dtile + 32 * LDO);    // [HG][32][2] lse, D
    int* soff = reinterpret_cast<int*>(stat + HG * 96);          // [34] sequence offsets of the tile (TILED)
    const int b = blockIdx.x, hg = blockIdx.y, d = H * 32;
    const int lane = threadIdx.x & 63, hh = threadIdx.x >> 6;
    const int li = lane & 31, h2 = lane >> 5;
    const int h = hg * HG + hh;
    constexpr bool TILED = MODE != 0;
    long long row0 = (long long)b * Smax;
    int S = Smax;
    if (total_rows > 0 && b == (int)gridDim.x - 1) {     // tail workgroup: rows past the last sequence <- 0
        const long long first = seq_off ? (long long)seq_off[n_seq] : (MODE == 2 ? (long long)n_seq * Smax : row0);
        bf16_t* z = dqkv + (size_t)hg * W;
        zero_slab_rows(z, 3LL * d, first, total_rows, W);
        zero_slab_rows(z + d, 3LL * d, first, total_rows, W);
        zero_slab_rows(z + 2 * d, 3LL * d, first, total_rows, W);
        return;
    }
    int s_first = b, s_last = b + 1;        // tile modes: see the forward kernel
    if (MODE == 1) {
        if (b >= tile_first[n_seq + 1]) return;
        s_first = tile_first[b];
        s_last = tile_first[b + 1];
        if ((int)threadIdx.x <= s_last - s_first) soff[threadIdx.x] = seq_off[s_first + threadIdx.x];   // <= 33 entries
    } else if (MODE == 2) {
        const int per = 32 / Smax;
        s_first = b * per;
        s_last = min(n_seq, s_first + per);
        if ((int)threadIdx.x <= s_last - s_first) soff[threadIdx.x] = (s_first + (int)threadIdx.x) * Smax;
        row0 = (long long)s_first * Smax;
        S = (s_last - s_first) * Smax;
    }
    if (seq_off) {
        row0 = seq_off[s_first];
        S = seq_off[s_last] - seq_off[s_first];
    }

    const bf16_t* src = qkv + (size_t)row0 * 3 * d + (size_t)hg * W;
    // FO: the first half of this head's Wo^T fragments, issued ahead of the tile loads (L2-resident: 8 KiB per head)
    F8 wa[8];
    const uint4* wp = FO ? reinterpret_cast<const uint4*>(wo_img) + (size_t)(hh * 16) * 64 + lane : nullptr;
    if (FO) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wa[ks].u = wp[ks * 64];
    }
    if (!FO) {
        const SlabSrc sl[4] = {{src, 3LL * d, tile, LD}, {src + d, 3LL * d, tile + W, LD}, {src + 2 * d, 3LL * d, tile + 2 * W, LD},
                               {dout + (size_t)row0 * d + (size_t)hg * W, (long long)d, dtile, LDO}};
        load_slabs<4>(sl, S);
        __syncthreads();
    } else {
        // the dx1m tile first, then the q | k | v slabs: their loads are in flight (registers) while the tile's dO^T = Wo^T-
        // columns x dx1m^T runs - loads return in order, so the wait for the dx1m pieces leaves the six later ones pending
        constexpr int CPR = W / 8;
        const int rmax = S > 0 ? S - 1 : 0;
        uint4 vd[2], vq[3][2];
        const bf16_t* dsrc = dout + (size_t)row0 * d + (size_t)hg * W;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = threadIdx.x + 512 * k, r = idx / CPR, c = idx % CPR;
            vd[k] = *reinterpret_cast<const uint4*>(dsrc + (long long)min(r, rmax) * d + 8 * c);
        }
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = threadIdx.x + 512 * k, r = idx / CPR, c = idx % CPR;
                vq[n][k] = *reinterpret_cast<const uint4*>(src + (size_t)n * d + (long long)min(r, rmax) * 3 * d + 8 * c);
            }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = threadIdx.x + 512 * k, r = idx / CPR, c = idx % CPR;
            *reinterpret_cast<uint4*>(dtile + r * LDO + 8 * c) = r < S ? vd[k] : make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        // dO^T of this head: lane (token li, half h2) ends up with dO[li][32 hh + rowmap(r, h2)], the layout stage_rows takes
        floatx16 da;
#pragma unroll
        for (int r = 0; r < 16; ++r) da[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks].v, row_frag(dtile, LDO, li, 0, ks, h2), da, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wa[ks].u = wp[(8 + ks) * 64];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks].v, row_frag(dtile, LDO, li, 0, 8 + ks, h2), da, 0, 0, 0);
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = threadIdx.x + 512 * k, r = idx / CPR, c = idx % CPR;
                *reinterpret_cast<uint4*>(tile + n * W + r * LD + 8 * c) = r < S ? vq[n][k] : make_uint4(0u, 0u, 0u, 0u);
            }
        __syncthreads();                        // every wave has read the whole dx1m tile; q | k | v are staged
        stage_rows(dtile, LDO, li, hh * 32, h2, da);
        __builtin_amdgcn_wave_barrier();        // (the head's dO slab is written and read by this wave only: in-order LDS)
    }
    // this lane's row (query in pass A, key in pass B): its sequence, first row of that sequence in the tile, length
    int my_seq = b, my_start = 0, my_len = S;
    if (TILED) {
        int qi = 0;
        for (int q = 1; q < s_last - s_first; ++q)
            if (row0 + li >= soff[q]) qi = q;
        my_seq = s_first + qi;
        my_start = soff[qi] - (int)row0;
        my_len = soff[qi + 1] - soff[qi];
    }

    // km: the keys this lane's query sees (pass A); qm: the queries that see this lane's key (pass B) = the rows of the
    // lane's own sequence (queries are never masked)
    const uint32_t qm = TILED ? (uint32_t)(((1ull << my_len) - 1ull) << my_start) : 0u;
    const uint32_t km = MODE == 2 ? (((key_mask ? (uint32_t)key_mask[my_seq] : ~0u) << my_start) & qm)
                        : TILED ? qm : ((key_mask ? (uint32_t)key_mask[b] : ~0u) & (uint32_t)((1ull << S) - 1ull));
    const DropCtx dc = drop_make(drop_p, seed, drop_site);
    const int qc = hh * 32, kc = W + hh * 32, vc = 2 * W + hh * 32, oc = hh * 32;
    // dropout row of query tile-row q = hbase + q (= (seq H + h) Smax + q_local); keys counted inside the sequence
    const uint64_t hbase = ((uint64_t)my_seq * H + h) * Smax - my_start;
    float* my_stat = stat + hh * 96;        // [32 queries][lse, D, dropout row hash]

    floatx16 acc, acc2;
    float p[16], g[16];
    // ---------------- pass A: lane = (query li, half h2), registers over keys -----------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tile, LD, li, kc, step, h2),
                                                      row_frag(tile, LD, li, qc, step, h2), acc, 0, 0, 0);      // K Q^T
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tile, LD, li, vc, step, h2),
                                                       row_frag(dtile, LDO, li, oc, step, h2), acc2, 0, 0, 0);  // V dO^T
    }
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        p[r] = ((km >> rowmap(r, h2)) & 1u) ? acc[r] * scale : -INFINITY;
        m = fmaxf(m, p[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        p[r] = (p[r] == -INFINITY) ? 0.f : __expf(p[r] - m);
        l += p[r];
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    const float lse = m + __logf(l);
    float D = 0.f;
    const uint32_t hrow = attn_drop_row(dc, hbase + li, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        p[r] *= inv;                                                               // P[q][key]
        g[r] = acc2[r] * attn_drop_key(dc, hrow, (uint32_t)(rowmap(r, h2) - my_start));   // dP[q][key]
        D = fmaf(p[r], g[r], D);
    }
    D += __shfl_xor(D, 32, 64);
    const bool qvalid = li < S;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = qvalid ? p[r] * (g[r] - D) * scale : 0.f;  // scale * dS[q][key]
    if (h2 == 0) { my_stat[li * 3 + 0] = lse; my_stat[li * 3 + 1] = D; my_stat[li * 3 + 2] = __uint_as_float(hrow); }
    floatx16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(tile, LD, kc, ks, lane), pack_regs(g, ks), dq, 0, 0, 0);
    // dq[r] = dQ[q = li][d = rowmap(r,h2)]: packed to bf16 right away (8 instead of 16 live registers through pass B)
    uint32_t dq_pk[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dq_pk[c] = f2bf_pk(dq[2 * c], dq[2 * c + 1]);
    __builtin_amdgcn_wave_barrier();
    // pass B reads the same q / k / v / dO fragments as pass A in swapped roles: make the compiler re-read them from LDS
    // instead of keeping 32 fragment registers alive across the softmax code (that was 24 spilled dwords per lane)
    asm volatile("" ::: "memory");

    // ---------------- pass B: lane = (key li, half h2), registers over queries -----------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tile, LD, li, qc, step, h2),
                                                      row_frag(tile, LD, li, kc, step, h2), acc, 0, 0, 0);      // Q K^T
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dtile, LDO, li, oc, step, h2),
                                                       row_frag(tile, LD, li, vc, step, h2), acc2, 0, 0, 0);    // dO V^T
    }
    const bool kvalid = MODE == 1 ? (li < S) : (li < S && (bool)((km >> li) & 1u));
    // P~ and dS go straight to packed bf16 pairs (they are only MFMA operands from here on): 16 registers instead of 32
    uint32_t pp[8], gp[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float pv[2], gv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * c + e;
            const int q = rowmap(r, h2);
            const float lse_q = my_stat[q * 3 + 0], D_q = my_stat[q * 3 + 1];
            const bool ok = kvalid && q < S && (!TILED || ((qm >> q) & 1u));
            const float pr = ok ? __expf(acc[r] * scale - lse_q) : 0.f;               // P[q][key = li]
            const float mult = attn_drop_key(dc, __float_as_uint(my_stat[q * 3 + 2]), (uint32_t)(li - my_start));
            pv[e] = pr * mult;                                                        // P~ (as used by O = P~ V)
            gv[e] = ok ? pr * (acc2[r] * mult - D_q) * scale : 0.f;                   // scale * dS[q][key]
        }
        pp[c] = f2bf_pk(pv[0], pv[1]);
        gp[c] = f2bf_pk(gv[0], gv[1]);
    }
    floatx16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        F8 fg, fp;
        fg.u = make_uint4(gp[4 * ks], gp[4 * ks + 1], gp[4 * ks + 2], gp[4 * ks + 3]);
        fp.u = make_uint4(pp[4 * ks], pp[4 * ks + 1], pp[4 * ks + 2], pp[4 * ks + 3]);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(tile, LD, qc, ks, lane), fg.v, dk, 0, 0, 0);
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(dtile, LDO, oc, ks, lane), fp.v, dv, 0, 0, 0);
    }
    // every operand read of this head's slabs is done (same wave, in-order LDS): stage dq|dk|dv over q|k|v
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 4; ++c)
        *reinterpret_cast<uint2*>(&tile[li * LD + qc + 8 * c + 4 * h2]) = make_uint2(dq_pk[2 * c], dq_pk[2 * c + 1]);
    stage_rows(tile, LD, li, kc, h2, dk);
    stage_rows(tile, LD, li, vc, h2, dv);
    __syncthreads();
    bf16_t* dst = dqkv + (size_t)row0 * 3 * d + (size_t)hg * W;
    store_slab(dst, 3LL * d, tile, LD, S, W);
    store_slab(dst + d, 3LL * d, tile + W, LD, S, W);
    store_slab(dst + 2 * d, 3LL * d, tile + 2 * W, LD, S, W);
}

}  // namespace

bool dsvg_attention_mfma_ok(int32_t dtype, int32_t S, int32_t n_heads) {
    static const bool off = getenv("DSVG_ATTN_VALU") != nullptr;
    // S <= 16 (dense): several sequences per 32-row tile (MODE 2); DSVG_ATTN_MFMA_MIN_S raises the lower limit again
    static const int min_s = getenv("DSVG_ATTN_MFMA_MIN_S") ? atoi(getenv("DSVG_ATTN_MFMA_MIN_S")) : 2;
    return !off && dtype == DSVG_BF16 && S >= min_s && S <= 32 && (n_heads % HG) == 0;
}

int dsvg_attention_fwd_mfma(const void* qkv, const uint64_t* key_mask, const int32_t* seq_off, int64_t total_rows,
                            const int32_t* tile_first, void* out, int64_t n_seq, int32_t S, int32_t n_heads,
                            float scale, float drop_p, uint32_t drop_site, const uint64_t* seed, hipStream_t st) {
    const size_t lds = (size_t)32 * LD * sizeof(bf16_t) + 34 * sizeof(int);
    const bool multi = !tile_first && !seq_off && S <= 16;      // dense short sequences: 32 / S of them per tile
    auto kern = tile_first ? attn_fwd_mfma_kernel<1> : (multi ? attn_fwd_mfma_kernel<2> : attn_fwd_mfma_kernel<0>);
    const int64_t n_wg = multi ? (n_seq + 32 / S - 1) / (32 / S) : n_seq;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg + (total_rows > 0 ? 1u : 0u), n_heads / HG), dim3(512), lds, st,
                       (const bf16_t*)qkv, key_mask, seq_off, (long long)total_rows, tile_first, (int)n_seq, (bf16_t*)out,
                       S, n_heads, scale, drop_p, drop_site, seed);
    DSVG_LAUNCH_CHECK("attention_fwd_mfma");
    return 0;
}

int dsvg_attention_bwd_mfma(const void* qkv, const uint64_t* key_mask, const int32_t* seq_off, int64_t total_rows,
                            const int32_t* tile_first, const void* dout, void* dqkv, int64_t n_seq, int32_t S,
                            int32_t n_heads, float scale, float drop_p, uint32_t drop_site, const uint64_t* seed,
                            hipStream_t st, const void* wo_packed_bwd) {
    const size_t lds = (size_t)32 * (LD + LDO) * sizeof(bf16_t) + (size_t)HG * 96 * sizeof(float) + 34 * sizeof(int);
    const bool multi = !tile_first && !seq_off && S <= 16;
    const bool fo = wo_packed_bwd != nullptr;       // (the caller checked: 8 heads, not the multi-sequence mode)
    auto kern = fo ? (tile_first ? attn_bwd_mfma_kernel<1, true> : attn_bwd_mfma_kernel<0, true>)
                   : (tile_first ? attn_bwd_mfma_kernel<1, false>
                                 : (multi ? attn_bwd_mfma_kernel<2, false> : attn_bwd_mfma_kernel<0, false>));
    DSVG_ENSURE_LDS((attn_bwd_mfma_kernel<0, false>), lds);
    DSVG_ENSURE_LDS((attn_bwd_mfma_kernel<1, false>), lds);
    DSVG_ENSURE_LDS((attn_bwd_mfma_kernel<2, false>), lds);
    DSVG_ENSURE_LDS((attn_bwd_mfma_kernel<0, true>), lds);
    DSVG_ENSURE_LDS((attn_bwd_mfma_kernel<1, true>), lds);
    const int64_t n_wg = multi ? (n_seq + 32 / S - 1) / (32 / S) : n_seq;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg + (total_rows > 0 ? 1u : 0u), n_heads / HG), dim3(512), lds, st,
                       (const bf16_t*)qkv, key_mask, seq_off, (long long)total_rows, tile_first, (int)n_seq,
                       (const bf16_t*)dout, (bf16_t*)dqkv, S, n_heads, scale, drop_p, drop_site, seed,
                       (const bf16_t*)wo_packed_bwd);
    DSVG_LAUNCH_CHECK("attention_bwd_mfma");
    return 0;
}

// packed_bwd[layer]: fragments 0 .. 127 = [head 8][K step 16][lane l][e] = Wo[16 ks + 8 (l >> 5) + e][32 head + (l & 31)], the A
// fragments of dO^T = Wo^T-columns x dx1m^T per head (offs[layer][1] = element offset of out_proj.weight [256, 256] in `flat`);
// fragments 128 .. 511 = in_proj_weight^T for attn_bwd_dx.hip (layouts: pack_images.h)
__global__ __launch_bounds__(256) void attn_pack_bwd_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                                            int n_layers, bf16_t* __restrict__ img) {
    dsvg_pack::attn_bwd_slot((long long)blockIdx.x * 256 + threadIdx.x, flat, offs, n_layers, img);      // (pack_images.h)
}
int dsvg_attn_pack_bwd_launch(const float* flat, const int64_t* offs, int n_layers, void* img, hipStream_t st) {
    const long long n = (long long)n_layers * dsvg_pack::ATTN_BWD_SLOTS;
    hipLaunchKernelGGL(attn_pack_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, flat, offs, n_layers,
                       (bf16_t*)img);
    DSVG_LAUNCH_CHECK("attn_pack_bwd");
    return 0;
}
