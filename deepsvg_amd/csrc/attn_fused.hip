// Fused attention sub-block of the pre-LN transformer layers, d_model = 256, 8 heads of 32, sequences of at most 32
// tokens, bf16 storage, fp32 accumulation and statistics:
//     x1 = x + drop_r( Wo . MHA( LN(x) ) + bo ),   MHA = per-head softmax(q k^T * scale) with dropout on the probabilities
// (deepsvg/model/layers/improved_transformer.py:43-46,127-131 with layers/attention.py / layers/functional.py:168,197-248).
// One launch replaces LayerNorm + in_proj GEMM + attention + out_proj GEMM: q|k|v, the probabilities and the head outputs
// stay on the chip; the inference call reads 512 B and writes 512 B per token.  The training call also stores what the
// (unfused) backward pass reads: LN(x), q|k|v, the head outputs and the row statistics.
//
// Structure (same skeleton as ffn_fused.hip):
//   * token-stationary waves: a wave owns ONE attention tile - up to 32 consecutive rows that hold whole sequences (the
//     packed encoder layout of dsvg_attention_tiles: block-diagonal attention inside the tile; or one padded sequence of
//     the dense layouts) - and keeps its LayerNorm-ed rows as 16 MFMA operand fragments in registers.  A 512-thread
//     workgroup = 8 consecutive tiles.
//   * the weights stream through an LDS ring of 32 KiB slots as ready-made MFMA A fragments (dsvg_attn_pack), 20 chunks per
//     layer: per head h [Wq_h | Wk_h] (32 fragments) and [Wv_h] (16), then out_proj in 4 chunks of two 32-row blocks.
//   * per head: q^T, k^T, v^T = W_frag x X_frag (transposed accumulators: lane = token, registers = head dims), + bias,
//     -> bf16.  S^T = K Q^T takes the packed q / k registers directly as MFMA operands (K order = register order on both
//     sides); softmax over the keys = 16 registers + one exchange with lane ^ 32; v^T goes through a 2.5 KiB per-wave
//     staging tile and comes back as the A operand of O^T = V^T P^T by hardware-transposed LDS reads.  The packed head
//     output is the B operand of out_proj, whose K index runs in that register order (dsvg_attn_pack lays Wo out so).
//   * out_proj: per 32-row block of outputs 16 MFMAs over the 8 heads x 2 K steps, then bias, dropout (the library's
//     standard draws: dsvg_drop_apply replays the mask for the backward pass), residual, bf16 stores (32 B per lane).
//   * TRAIN: q, k, v, head output go through the staging tile so that every global store is a full 64-byte row segment;
//     the stores of a stage are held in registers and issued right behind the next ring synchronisation (vmcnt counts
//     stores, out of order with respect to loads: the only safe DMA wait is vmcnt(0), so stores need a stage to drain).
#include "fused_common.h"
#include "pack_images.h"
#include "../../include/dsvg.h"

typedef short shortx4 __attribute__((ext_vector_type(4)));
typedef float af2 __attribute__((ext_vector_type(2)));
typedef unsigned short au16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int AD = 256;                 // d_model
constexpr int AH = 8;                   // heads
constexpr int SLOT = 32 * FRAG;         // ring slot = the largest chunk
constexpr int N_CHUNK = 20;             // 8 x (qk, v) + 4 x out_proj
constexpr int IMG_FRAGS = 512;          // 384 in_proj + 128 out_proj fragments per layer
constexpr int TILES_PER_WG = 8;
constexpr int VLD = 32;                 // row stride (elements) of the per-wave staging tile [32 tokens][32 dims]: 64 B,
                                        // swizzled (stg_swz)
constexpr int SMALL_LDS = (768 + 256 + 256 + 256) * 4;      // in_proj bias | out_proj bias | gamma | beta
constexpr int STAGE_LDS = TILES_PER_WG * 32 * VLD * 2;

using dsvg_pack::rowmap;
static_assert(dsvg_pack::ATTN_IMG_FRAGS == IMG_FRAGS && dsvg_pack::D == AD && dsvg_pack::H == AH, "pack_images.h restates these");
// byte offset of chunk c in the layer image; chunks 2 h + 1 (v of head h) have 16 fragments, all others 32
__device__ __forceinline__ int chunk_off(int c) {
    return c < 16 ? ((c >> 1) * 48 + (c & 1) * 32) * FRAG : (384 + (c - 16) * 32) * FRAG;
}

// weight packing (dsvg_attn_pack): fp32 master parameters -> bf16 fragment images, body and layout in pack_images.h
__global__ __launch_bounds__(256) void attn_pack_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                                        int n_layers, bf16_t* __restrict__ img) {
    dsvg_pack::attn_slot((long long)blockIdx.x * 256 + threadIdx.x, flat, offs, n_layers, img);
}

// A[i = column c (lane & 31)][K slot e] = img[row rowmap(8 ks + e, lane >> 5)][col0 + c]: two hardware-transposed 4 x 16
// reads (the same access attention_mfma.hip uses for V^T)
__device__ __forceinline__ bf16x8 col_frag(const bf16_t* img, int ld, int col0, int ks, int lane) {
    const int g = lane >> 4, q16 = lane & 15;
    const int row = 16 * ks + 4 * (g >> 1) + (q16 >> 2);
    const int col = col0 + 16 * (g & 1) + 4 * (q16 & 3);
    union { bf16x8 v; shortx4 h[2]; } f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(&img[row * ld + col]));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(&img[(row + 8) * ld + col]));
    return f.v;
}

// The staging tile's swizzle: 64-byte rows, the eight 8-byte granules of row r stored at granule ^ stg_swz(r).
//   * writes (ds_write_b64, 16-lane groups = 16 consecutive rows, 32 banks): rows of equal parity share a 16-bank half and
//     get 8 different granule positions (stg_swz is a bijection of (r >> 1) & 7);
//   * row reads (ds_read_b64 pairs, 32-lane groups = 8 rows x 4 granules of one parity, 64 banks): rows r and r + 4 share a
//     quarter of the banks and read granules of opposite parity (bit 0 of stg_swz = bit 2 of r);
//   * transposed reads (32-lane groups = 4 consecutive rows x 8 granules): all 64 banks whatever the order inside a row.
// (Round 2's 80-byte rows put rows r, r + 16 - and r, r + 4 of a read group - on the same banks: 23 % of the LDS cycles of
// the training variant were bank conflicts, profiles/r02_attn_pmc_summary.txt.)
__device__ __forceinline__ int stg_swz(int row) {
    const int m = row >> 1;
    return ((m >> 1) & 1) | ((m & 1) << 1) | (m & 4);
}
__device__ __forceinline__ bf16x8 col_frag_swz(const bf16_t* img, int ks, int lane) {
    const int g = lane >> 4, q16 = lane & 15;
    const int row = 16 * ks + 4 * (g >> 1) + (q16 >> 2);
    const int gran = 4 * (g & 1) + (q16 & 3);                      // 8-byte granule of the row
    const int off = row * VLD + 4 * (gran ^ stg_swz(row));
    const int off8 = (row + 8) * VLD + 4 * (gran ^ stg_swz(row + 8));
    union { bf16x8 v; shortx4 h[2]; } f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(&img[off]));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(&img[off8]));
    return f.v;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
constexpr int NBUF = 3;         // ring slots of the weight stream
constexpr int DIST = NBUF - 1;  // chunks the DMA runs ahead

template <bool TRAIN, bool TILED>
__global__ __launch_bounds__(512, 2) void attn_block_fwd_kernel(
        const bf16_t* __restrict__ x, const bf16_t* __restrict__ img, const float* __restrict__ in_bias,
        const float* __restrict__ out_bias, const float* __restrict__ gamma, const float* __restrict__ beta,
        const uint64_t* __restrict__ key_mask, const int32_t* __restrict__ seq_off, const int32_t* __restrict__ tile_first,
        int n_seq, int Smax, long long total_rows, bf16_t* __restrict__ x1, bf16_t* __restrict__ xn_out,
        bf16_t* __restrict__ qkv_out, bf16_t* __restrict__ ao_out, float* __restrict__ mean_out,
        float* __restrict__ rstd_out, float eps, float scale, float drop_p, const uint64_t* __restrict__ seed,
        uint32_t site_p, uint32_t site_r, const bf16_t* __restrict__ gadd, long long gadd_ld, uint32_t site_g, int warm) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // [NBUF slots | biases, gamma, beta | 8 staging tiles]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h2 = lane >> 5;
    const bool late = wave >= 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    float* sbin = reinterpret_cast<float*>(smem + NBUF * SLOT);
    float* sbo = sbin + 768;
    float* sga = sbo + 256;
    float* sbe = sga + 256;
    bf16_t* stg = reinterpret_cast<bf16_t*>(smem + NBUF * SLOT + SMALL_LDS) + wave * (32 * VLD);

    // ---- weight stream ---------------------------------------------------------------------------------------------------
    const char* img_b = reinterpret_cast<const char*>(img) + lane * 16;
    auto issue = [&](int c) {
        const uint32_t slot = lds0 + (uint32_t)(c % NBUF) * SLOT;
        const char* src = img_b + chunk_off(c);
        if (c < 16 && (c & 1)) dma2(src + wave * 2048, __builtin_amdgcn_readfirstlane(slot + wave * 2048));
        else dma4(src + wave * 4096, __builtin_amdgcn_readfirstlane(slot + wave * 4096));
    };
    // 3 ring slots, the DMA two chunks ahead; every sync drains the wave's memory operations (vmcnt(0)).  (Round 4 measured a
    // 4-slot ring with counted waits - the training stores no longer drained at the 20 syncs - on MI355X: bit-identical, but 2 %
    // SLOWER per launch (176 vs 173 us at 127 k rows) and no change of the step; removed.  profiles/r04_experimental_attn_bench.log)
    // Round 6: the layer's image (512 KiB) is cold in this XCD's L2 when the launch starts and the workgroups walk it in
    // lockstep - every chunk would begin with an HBM miss all of them wait for.  The workgroups of an XCD (block b runs on XCD
    // b % 8) request it up front, 32 KiB each, into this wave's own 4 KiB of the ring slot its DMA of chunk 2 overwrites later
    // (same wave: in order); nobody reads that slot before chunk 2 has landed.
    if (warm)
        dma4(img_b + (size_t)((blockIdx.x >> 3) & 15u) * (32 * 1024) + wave * 4096,
             __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)DIST * SLOT + wave * 4096));
    issue(0);
    issue(1);

    for (int i = tid; i < 768; i += 512) sbin[i] = in_bias[i];
    if (tid < 256) { sbo[tid] = out_bias[tid]; sga[tid] = gamma[tid]; sbe[tid] = beta[tid]; }

    // ---- this wave's tile: rows [row0, row0 + S) ----------------------------------------------------------------------------
    const int b = blockIdx.x * TILES_PER_WG + wave;
    // dense layouts: a tile holds per = 32 / Smax whole sequences (one for the 17..32-token stages, four for 8-token groups)
    const int per = TILED ? 1 : 32 / Smax;
    const int n_tiles = TILED ? tile_first[n_seq + 1] : (n_seq + per - 1) / per;
    const long long real_rows = TILED ? (long long)seq_off[n_seq] : (long long)n_seq * Smax;
    long long row0 = 0;
    int S = 0, s_first = 0, n_in = 1;
    bool pad_tile = false;
    if (b < n_tiles) {
        if (TILED) {
            s_first = tile_first[b];
            n_in = tile_first[b + 1] - s_first;
            row0 = seq_off[s_first];
            S = seq_off[s_first + n_in] - (int)row0;
        } else {
            s_first = b * per;
            n_in = min(per, n_seq - s_first);
            row0 = (long long)s_first * Smax;
            S = n_in * Smax;
        }
    } else {        // rows past the last sequence (bucket padding): tiles of 32 rows that attend to themselves only
        pad_tile = true;
        row0 = real_rows + 32ll * (b - n_tiles);
        const long long left = total_rows - row0;
        S = left > 32 ? 32 : (left > 0 ? (int)left : 0);
    }
    S = __builtin_amdgcn_readfirstlane(S);
    if (__syncthreads_count(S > 0) == 0) {          // (also publishes the staged biases)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // the lane's query row: its sequence, the first row of that sequence inside the tile and its length
    int my_seq = s_first, my_start = 0, my_len = S;
    if (pad_tile) {
        my_start = li;
        my_len = 1;
    } else if (TILED) {
        const int so = seq_off[s_first + min(lane, n_in)];       // lanes 0 .. n_in (<= 32) hold the tile's offsets
        int qi = 0;
        for (int q = 1; q < n_in; ++q)
            if (row0 + li >= __shfl(so, q, 64)) qi = q;
        const int a0 = __shfl(so, qi, 64), a1 = __shfl(so, qi + 1, 64);
        my_seq = s_first + qi;
        my_start = a0 - (int)row0;
        my_len = a1 - a0;
    } else {
        const int qi = min(li / Smax, n_in - 1);
        my_seq = s_first + qi;
        my_start = qi * Smax;
        my_len = Smax;
    }
    uint32_t km = (uint32_t)((1ull << my_len) - 1ull);
    if (!TILED && !pad_tile && key_mask) km &= (uint32_t)key_mask[my_seq];
    km <<= my_start;
    const bool row_live = li < S;
    long long my_row = row0 + (S > 0 ? min(li, S - 1) : 0);     // rows past S: a clamped copy, computed but never stored
    if (S <= 0 || my_row >= total_rows) my_row = 0;

    // ---- LayerNorm in registers -> 16 operand fragments ----------------------------------------------------------------
    bf16x8 xf[16];
    {
        const char* xr = reinterpret_cast<const char*>(x) + (size_t)my_row * (AD * 2) + h2 * 16;
        uint4 raw[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) raw[ks] = *reinterpret_cast<const uint4*>(xr + 32 * ks);
        // (statistics from the packed words: ln_stats_packed, fused_common.h)
        float s, q, mean, rstd;
        ln_stats_packed(raw, s, q);
        ln_mean_rstd256(s, q, eps, mean, rstd);
        const float shift = -mean * rstd;
        char* xo = TRAIN ? reinterpret_cast<char*>(xn_out) + (size_t)my_row * (AD * 2) + h2 * 16 : nullptr;
        const bool st = TRAIN && row_live;
        if (st && h2 == 0) { mean_out[my_row] = mean; rstd_out[my_row] = rstd; }
        // gamma / beta come from LDS one K step at a time: `zoff` (always 0) is made to depend on the previous step's result,
        // otherwise hipcc hoists all 64 reads (256 registers) above the loop and spills the prologue
        int zoff = 8 * h2;
        asm volatile("" : "+v"(zoff));
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float v[8];
            unpack8(raw[ks], v);
            const float4 g0 = *reinterpret_cast<const float4*>(sga + 16 * ks + zoff);
            const float4 g1 = *reinterpret_cast<const float4*>(sga + 16 * ks + zoff + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(sbe + 16 * ks + zoff);
            const float4 b1 = *reinterpret_cast<const float4*>(sbe + 16 * ks + zoff + 4);
            v[0] = fmaf(fmaf(v[0], rstd, shift), g0.x, b0.x); v[1] = fmaf(fmaf(v[1], rstd, shift), g0.y, b0.y);
            v[2] = fmaf(fmaf(v[2], rstd, shift), g0.z, b0.z); v[3] = fmaf(fmaf(v[3], rstd, shift), g0.w, b0.w);
            v[4] = fmaf(fmaf(v[4], rstd, shift), g1.x, b1.x); v[5] = fmaf(fmaf(v[5], rstd, shift), g1.y, b1.y);
            v[6] = fmaf(fmaf(v[6], rstd, shift), g1.z, b1.z); v[7] = fmaf(fmaf(v[7], rstd, shift), g1.w, b1.w);
            Frag8 f;
            f.u = pack8(v);
            xf[ks] = f.v;
            if (st) *reinterpret_cast<uint4*>(xo + 32 * ks) = f.u;
            asm volatile("" : "+v"(zoff) : "v"(f.u.x), "v"(f.u.y), "v"(f.u.z), "v"(f.u.w) : "memory");
        }
    }

    __builtin_amdgcn_sched_barrier(0);      // (the zeroed operand queue below must not be hoisted above the LayerNorm)
    const DropCtx dp = drop_make(drop_p, seed, site_p);
    const DropCtx dr = drop_make(drop_p, seed, site_r);
    const DropCtx dg = drop_make(drop_p, seed, site_g);
    const char* lbase = smem + lane * 16;
    auto ld = [&](const char* p) -> uint4 { return *reinterpret_cast<const uint4*>(p); };

    // TRAIN: row segments waiting for their store slot (issued right behind the next sync point)
    // (a lane owns the 16-byte pieces `lane` and `lane + 64` of a staged 32 x 32 tile: token piece >> 2, dims 8 (piece & 3) ..)
    uint4 pa0, pa1, pb0, pb1;       // (named registers: as an array they end up in scratch memory)
    pa0 = pa1 = pb0 = pb1 = make_uint4(0u, 0u, 0u, 0u);
    const bool ok0 = TRAIN && (lane >> 2) < S, ok1 = TRAIN && (lane >> 2) + 16 < S;
    char* qb = TRAIN ? reinterpret_cast<char*>(qkv_out) + ((size_t)(row0 + (lane >> 2)) * (3 * AD) + 8 * (lane & 3)) * 2 : nullptr;
    char* ab = TRAIN ? reinterpret_cast<char*>(ao_out) + ((size_t)(row0 + (lane >> 2)) * AD + 8 * (lane & 3)) * 2 : nullptr;
    // sync(k): afterwards chunks <= k + 1 are readable and chunk k + 2 is on its way into the slot chunk k - 1 has left.
    // The two waves of a SIMD (w, w + 4) run one stage apart: waves 0-3 pass sync(k) BEFORE the stage of chunk k, waves 4-7
    // AFTER it (same straight-line code, only the barrier position differs), so the MFMA-bound projection stage of one
    // sits beside the VALU-bound softmax stage of the other.  Behind the barrier go the stores of the stage the wave has
    // just finished: the q | k tiles (chunk 2 h) or the v and head-output tiles (chunk 2 h + 1) of head h.
    auto sync = [&](int k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (k + DIST < N_CHUNK) issue(k + DIST);
        const int done = late ? k : k - 1;         // chunk whose stage this wave finished last
        if (TRAIN && done >= 0 && done < 16) {
            const int hp = done >> 1;
            const bool qk = !(done & 1);
            char* d0 = qk ? qb + 64 * hp : qb + 4 * AD + 64 * hp;
            char* d1 = qk ? qb + 2 * AD + 64 * hp : ab + 64 * hp;
            const int far0 = 16 * 3 * AD * 2, far1 = qk ? 16 * 3 * AD * 2 : 16 * AD * 2;
            if (ok0) { *reinterpret_cast<uint4*>(d0) = pa0; *reinterpret_cast<uint4*>(d1) = pb0; }
            if (ok1) { *reinterpret_cast<uint4*>(d0 + far0) = pa1; *reinterpret_cast<uint4*>(d1 + far1) = pb1; }
        }
    };
    // the packed 32 x 32 tile (lane: token li, dims 8 g + 4 h2 + 0..3 as piece g) -> staging tile [token][dim]
    auto stage_put = [&](const Frag8 (&f)[2]) {
        asm volatile("" ::: "memory");      // (one wave, in-order LDS: only the compiler must keep put / take / col_frag in order)
        const int sw = stg_swz(li);
        *reinterpret_cast<uint2*>(&stg[li * VLD + 4 * ((0 + h2) ^ sw)]) = make_uint2(f[0].u.x, f[0].u.y);
        *reinterpret_cast<uint2*>(&stg[li * VLD + 4 * ((2 + h2) ^ sw)]) = make_uint2(f[0].u.z, f[0].u.w);
        *reinterpret_cast<uint2*>(&stg[li * VLD + 4 * ((4 + h2) ^ sw)]) = make_uint2(f[1].u.x, f[1].u.y);
        *reinterpret_cast<uint2*>(&stg[li * VLD + 4 * ((6 + h2) ^ sw)]) = make_uint2(f[1].u.z, f[1].u.w);
        asm volatile("" ::: "memory");
    };
    // staging tile -> two pending 16-byte row pieces per lane: token (lane + 64 j) >> 2, dims 8 ((lane + 64 j) & 3) .. (rows 16
    // apart share the swizzle); each piece as its two granules
    const int tk_lo = (lane >> 2) * VLD + 4 * ((2 * (lane & 3)) ^ stg_swz(lane >> 2));
    const int tk_hi = (lane >> 2) * VLD + 4 * ((2 * (lane & 3) + 1) ^ stg_swz(lane >> 2));
    // (the far rows' offsets are made opaque: near + 1 KiB would be merged into ds_read2st64_b64, which is banked in 16-lane
    // groups over 32 banks and runs at half the rate of two ds_read_b64)
    int tk_lo_far = tk_lo + 16 * VLD, tk_hi_far = tk_hi + 16 * VLD;
    asm volatile("" : "+v"(tk_lo_far), "+v"(tk_hi_far));
    auto stage_take = [&](uint4& near, uint4& far) {
        const uint2 n0 = *reinterpret_cast<const uint2*>(&stg[tk_lo]), n1 = *reinterpret_cast<const uint2*>(&stg[tk_hi]);
        const uint2 f0 = *reinterpret_cast<const uint2*>(&stg[tk_lo_far]), f1 = *reinterpret_cast<const uint2*>(&stg[tk_hi_far]);
        near = make_uint4(n0.x, n0.y, n1.x, n1.y);
        far = make_uint4(f0.x, f0.y, f1.x, f1.y);
    };

    bf16x8 aof[16];         // the out_proj operand queue: every head shifts it by two fragments and appends its own

    // ---- per-lane constants of the softmax + dropout code (round 5: the head loop is bound by VALU issue, 21-27 instructions
    // per MFMA) ----------------------------------------------------------------------------------------------------------------
    //   * the key mask as an additive bias (0 / -inf), the score scale with log2(e) folded in: p = exp2(fma(st, sc2, mb) - m),
    //     one v_pk_fma + v_exp per element instead of bit test + multiply + select + compare + select + multiply + exp;
    //   * the dropout words of the probabilities: the lane's 16 keys are 4 runs of 4 consecutive ones (8 q + 4 h2 .. + 3); counted
    //     from the start of the lane's sequence, run q begins at ko = (8 q + 4 h2 - my_start) & 31 and takes the three words
    //     ko >> 1 .. + 2 of the row hash.  Their multipliers do not depend on the head: 12 registers, set once.  A pair of
    //     adjacent keys takes its two 16-bit draws from ONE word when ko is even, from the high half of one word and the low half
    //     of the next when it is odd: v_alignbit_b32 by 0 / 16 bits; the mask is then applied to the PACKED bf16 pair as in
    //     ffn_act_packed (two saturating 16-bit subtractions) - the same draws attn_drop_key hands the backward pass.
    float mb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) mb[r] = ((km >> rowmap(r, h2)) & 1u) ? 0.f : -INFINITY;
    const float sc2 = scale * 1.4426950408889634f;
    uint32_t cw[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t ko = (uint32_t)(8 * q + 4 * h2 - my_start) & 31u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t j = ((ko >> 1) + (uint32_t)i) & 15u;
            cw[q][i] = ((0x7feb352du * (j + 1u)) ^ (0x846ca68bu >> j)) | 1u;
        }
    }
    const uint32_t odd_sh = (my_start & 1) ? 16u : 0u;

    // ---- heads -----------------------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // chunks 0 and 1 are in LDS
#pragma unroll 1
    for (int h = 0; h < AH; ++h) {
        if (!late) sync(2 * h);
        Frag8 qf[2], kf[2];
        {
            const char* sl = lbase + ((2 * h) % NBUF) * SLOT;
            floatx16 qa, ka;
#pragma unroll
            for (int r = 0; r < 16; ++r) { qa[r] = 0.f; ka[r] = 0.f; }
            uint4 ring[4];
            // consumption order q0 k0 q1 k1 ...: fragment of step n = 16 (n & 1) + (n >> 1)
#pragma unroll
            for (int n = 0; n < 4; ++n) ring[n] = ld(sl + (16 * (n & 1) + (n >> 1)) * FRAG);
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                Frag8 a;
                a.u = ring[n & 3];
                if (n & 1) ka = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, xf[n >> 1], ka, 0, 0, 0);
                else qa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, xf[n >> 1], qa, 0, 0, 0);
                if (n + 4 < 32) ring[n & 3] = ld(sl + (16 * ((n + 4) & 1) + ((n + 4) >> 1)) * FRAG);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                float vq[8], vk[8];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q4 = 2 * ks2 + qq;
                    const float4 bq = *reinterpret_cast<const float4*>(sbin + 32 * h + 8 * q4 + 4 * h2);
                    const float4 bk = *reinterpret_cast<const float4*>(sbin + 256 + 32 * h + 8 * q4 + 4 * h2);
                    vq[4 * qq + 0] = qa[4 * q4 + 0] + bq.x; vq[4 * qq + 1] = qa[4 * q4 + 1] + bq.y;
                    vq[4 * qq + 2] = qa[4 * q4 + 2] + bq.z; vq[4 * qq + 3] = qa[4 * q4 + 3] + bq.w;
                    vk[4 * qq + 0] = ka[4 * q4 + 0] + bk.x; vk[4 * qq + 1] = ka[4 * q4 + 1] + bk.y;
                    vk[4 * qq + 2] = ka[4 * q4 + 2] + bk.z; vk[4 * qq + 3] = ka[4 * q4 + 3] + bk.w;
                }
                qf[ks2].u = pack8(vq);
                kf[ks2].u = pack8(vk);
            }
            if (TRAIN) {
                stage_put(qf);
                stage_take(pa0, pa1);
                stage_put(kf);
                stage_take(pb0, pb1);
            }
        }
        if (late) sync(2 * h);
        else sync(2 * h + 1);
        {
            const char* sl = lbase + ((2 * h + 1) % NBUF) * SLOT;
            // scores first (their operands are in registers): st[r] = q_li . k_key(r, h2)
            floatx16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0].v, qf[0].v, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1].v, qf[1].v, st, 0, 0, 0);
            floatx16 va;
#pragma unroll
            for (int r = 0; r < 16; ++r) va[r] = 0.f;
            uint4 ring[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) ring[n] = ld(sl + n * FRAG);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                Frag8 a;
                a.u = ring[n & 3];
                va = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, xf[n], va, 0, 0, 0);
                if (n + 4 < 16) ring[n & 3] = ld(sl + (n + 4) * FRAG);
                __builtin_amdgcn_sched_barrier(0);
            }
            // softmax over the keys of the lane's query row (constants: see above the head loop)
            float p[16];
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const af2 t = __builtin_elementwise_fma(af2{st[r], st[r + 1]}, af2{sc2, sc2}, af2{mb[r], mb[r + 1]});
                p[r] = t[0]; p[r + 1] = t[1];
                m = fmaxf(m, fmaxf(t[0], t[1]));
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            m = fmaxf(m, -1e30f);                   // (an all-masked row: exp2(-inf + 1e30) = 0, never -inf - -inf)
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(p[r] - m);
                l += p[r];
            }
            l += __shfl_xor(l, 32, 64);
            // normalisation and the dropout scale in one factor; padded query rows -> 0 (keeps the NaN of an all-masked row out
            // of the MFMA)
            const float cnorm = row_live ? dp.scale / l : 0.f;
            uint32_t ppk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const af2 t = af2{p[2 * k], p[2 * k + 1]} * af2{cnorm, cnorm};
                ppk[k] = f2bf_pk(t[0], t[1]);
            }
            if (!row_live) {
#pragma unroll
                for (int k = 0; k < 8; ++k) ppk[k] = 0u;
            }
            if (dp.on) {
                // dropout row of (sequence s, head h, query i) = (s H + h) Smax + i; i and the keys counted inside the sequence
                const uint32_t hrow = attn_drop_row(dp, ((uint64_t)my_seq * AH + h) * Smax + (li - my_start), 0);
                const au16x2 tp = {(unsigned short)dp.thresh, (unsigned short)dp.thresh}, z2 = {0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t w[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) w[i] = (hrow * cw[q][i]) ^ __umulhi(hrow, cw[q][i]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t d = __builtin_amdgcn_alignbit(w[j + 1], w[j], odd_sh);
                        const au16x2 rr = z2 - __builtin_elementwise_sub_sat(tp, __builtin_bit_cast(au16x2, d));
                        ppk[2 * q + j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(au16x2, ppk[2 * q + j]), rr));
                    }
                }
            }
            // v^T + bias -> staging tile (token-major), read back transposed as the A operand of O^T = V^T P^T
            Frag8 vf[2];
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                float vv[8];
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q4 = 2 * ks2 + qq;
                    const float4 bv = *reinterpret_cast<const float4*>(sbin + 512 + 32 * h + 8 * q4 + 4 * h2);
                    vv[4 * qq + 0] = va[4 * q4 + 0] + bv.x; vv[4 * qq + 1] = va[4 * q4 + 1] + bv.y;
                    vv[4 * qq + 2] = va[4 * q4 + 2] + bv.z; vv[4 * qq + 3] = va[4 * q4 + 3] + bv.w;
                }
                vf[ks2].u = pack8(vv);
            }
            stage_put(vf);
            if (TRAIN) stage_take(pa0, pa1);
            floatx16 ot;
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag8 pf;
                pf.u = make_uint4(ppk[4 * ks + 0], ppk[4 * ks + 1], ppk[4 * ks + 2], ppk[4 * ks + 3]);
                ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag_swz(stg, ks, lane), pf.v, ot, 0, 0, 0);
            }
            // ot[r] = O[token li][dim rowmap(r, h2)] -> packed, appended to the out_proj operand queue (oldest head first)
            Frag8 of[2];
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                float vo[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) vo[e] = ot[8 * ks2 + e];
                of[ks2].u = pack8(vo);
            }
#pragma unroll
            for (int i = 0; i < 14; ++i) aof[i] = aof[i + 2];
            aof[14] = of[0].v;
            aof[15] = of[1].v;
            if (TRAIN) {
                stage_put(of);
                stage_take(pb0, pb1);
            }
        }
        if (late) sync(2 * h + 1);
    }

    // ---- out_proj: two 32-row blocks of outputs per chunk, + bias, dropout, residual ---------------------------------------------
    const long long m = row0 + li;
    const char* xres = reinterpret_cast<const char*>(x) + (size_t)my_row * (AD * 2);
    // optional per-sequence conditioning row (the decoder's linear_global(z), improved_transformer.py:131-136): x1 += drop(g)
    // with one mask element per (sequence, channel) - ids seq * 256 + column - as dsvg_bcast_add_fwd draws them
    const char* grow = gadd ? reinterpret_cast<const char*>(gadd) + (size_t)my_seq * (size_t)gadd_ld * 2 : nullptr;
    char* yrow = reinterpret_cast<char*>(x1) + (size_t)my_row * (AD * 2);
#pragma unroll 1
    for (int u = 0; u < 4; ++u) {
        if (!late) sync(16 + u);
        const char* sl = lbase + ((16 + u) % NBUF) * SLOT;
        uint4 res[4], gr[4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            res[2 * tt] = *reinterpret_cast<const uint4*>(xres + (32 * (2 * u + tt) + 16 * h2) * 2);
            res[2 * tt + 1] = *reinterpret_cast<const uint4*>(xres + (32 * (2 * u + tt) + 16 * h2 + 8) * 2);
            if (grow) {
                gr[2 * tt] = *reinterpret_cast<const uint4*>(grow + (32 * (2 * u + tt) + 16 * h2) * 2);
                gr[2 * tt + 1] = *reinterpret_cast<const uint4*>(grow + (32 * (2 * u + tt) + 16 * h2 + 8) * 2);
            }
        }
        floatx16 ya[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { ya[0][r] = 0.f; ya[1][r] = 0.f; }
        uint4 ring[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) ring[n] = ld(sl + (16 * (n & 1) + (n >> 1)) * FRAG);
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            Frag8 a;
            a.u = ring[n & 3];
            ya[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, aof[n >> 1], ya[n & 1], 0, 0, 0);
            if (n + 4 < 32) ring[n & 3] = ld(sl + (16 * ((n + 4) & 1) + ((n + 4) >> 1)) * FRAG);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            uint32_t xc[4][4];
            tile_to_cols16(ya[tt], xc);
            const int n16 = 32 * (2 * u + tt) + 16 * h2;
            uint4 pk[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                float v[8], rv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(xc[2 * cb][e]); v[4 + e] = __uint_as_float(xc[2 * cb + 1][e]); }
                const float4 b0 = *reinterpret_cast<const float4*>(sbo + n16 + 8 * cb);
                const float4 b1 = *reinterpret_cast<const float4*>(sbo + n16 + 8 * cb + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                if (dr.on) {
                    float dm[8];
                    drop_mult8(dr, (uint64_t)m * AD + n16 + 8 * cb, dm);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= dm[e];
                }
                unpack8(res[2 * tt + cb], rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
                if (grow) {
                    float gv[8];
                    unpack8(gr[2 * tt + cb], gv);
                    if (dg.on) {
                        float gm[8];
                        drop_mult8(dg, (uint64_t)my_seq * AD + n16 + 8 * cb, gm);
#pragma unroll
                        for (int e = 0; e < 8; ++e) gv[e] *= gm[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += gv[e];
                }
                pk[cb] = pack8(v);
            }
            if (row_live) {
                *reinterpret_cast<uint4*>(yrow + n16 * 2) = pk[0];
                *reinterpret_cast<uint4*>(yrow + n16 * 2 + 16) = pk[1];
            }
        }
        if (late) sync(16 + u);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int64_t dsvg_attn_pack_bytes(int32_t n_layers) { return (int64_t)n_layers * IMG_FRAGS * FRAG; }

extern "C" int dsvg_attn_pack(const float* flat_f32, const int64_t* offs, int32_t n_layers, int32_t d_model, int32_t n_heads,
                              void* packed, void* stream) {
    DSVG_CHECK_ARG(flat_f32 && offs && packed, "attn_pack: null pointer");
    DSVG_CHECK_ARG(d_model == AD && n_heads == AH, "attn_pack: the fused attention block is built for d_model 256 / 8 heads");
    DSVG_CHECK_ARG(n_layers > 0, "attn_pack: bad layer count");
    const long long n = (long long)n_layers * IMG_FRAGS * 64;
    hipLaunchKernelGGL(attn_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat_f32, offs,
                       n_layers, (bf16_t*)packed);
    DSVG_LAUNCH_CHECK("attn_pack");
    return 0;
}

extern "C" int dsvg_attn_block_fwd(const void* x, const void* packed_layer, const float* in_bias, const float* out_bias,
                                   const float* gamma, const float* beta, const uint64_t* key_mask, const int32_t* seq_off,
                                   const int32_t* tile_first, int64_t n_seq, int32_t S, int64_t rows, void* x1, void* xn_out,
                                   void* qkv_out, void* ao_out, float* mean_out, float* rstd_out, float eps, float scale,
                                   float drop_p, uint32_t site_probs, uint32_t site_res, const void* seed,
                                   const void* seq_add, int64_t seq_add_ld, uint32_t site_seq_add, void* stream) {
    DSVG_CHECK_ARG(x && packed_layer && in_bias && out_bias && gamma && beta && x1, "attn_block_fwd: null pointer");
    DSVG_CHECK_ARG(S >= 1 && S <= 32, "attn_block_fwd: sequences of at most 32 tokens (got %d)", S);
    DSVG_CHECK_ARG(n_seq > 0 && rows > 0 && rows < (1ll << 31), "attn_block_fwd: bad sizes");
    DSVG_CHECK_ARG(!(drop_p > 0.f) || seed, "attn_block_fwd: dropout needs a seed");
    DSVG_CHECK_ARG(!(drop_p > 0.5f), "attn_block_fwd: dropout rates up to 0.5 (the packed mask code)");
    const bool tiled = seq_off != nullptr;
    DSVG_CHECK_ARG(!tiled || (tile_first && !key_mask), "attn_block_fwd: the packed layout needs its tile list and no key mask");
    DSVG_CHECK_ARG(tiled || rows >= n_seq * S, "attn_block_fwd: %lld rows for %lld sequences of %d", (long long)rows,
                   (long long)n_seq, S);
    const bool train = xn_out != nullptr;
    DSVG_CHECK_ARG(!train || (qkv_out && ao_out && mean_out && rstd_out), "attn_block_fwd: the training outputs come together");
    DSVG_CHECK_ARG((((uintptr_t)x | (uintptr_t)x1 | (uintptr_t)packed_layer | (uintptr_t)xn_out | (uintptr_t)qkv_out |
                     (uintptr_t)ao_out | (uintptr_t)seq_add) & 15) == 0, "attn_block_fwd: operands must be 16-byte aligned");
    DSVG_CHECK_ARG(!seq_add || !tiled, "attn_block_fwd: the per-sequence add is wired for the dense layouts");
    DSVG_CHECK_ARG(!seq_add || (seq_add_ld >= AD && seq_add_ld % 8 == 0), "attn_block_fwd: bad seq_add row stride %lld", (long long)seq_add_ld);
    // waves: one per attention tile + one per 32 rows of bucket padding behind the last sequence.  Adjacent tiles of the
    // packed layout hold more than 32 rows together, so there are at most rows / 16 + 1 of them (and at most n_seq)
    long long waves;
    if (tiled) {
        const long long a = n_seq + (rows + 31) / 32, b = rows / 16 + 2;
        waves = a < b ? a : b;
    } else {
        const int per = 32 / S;                     // whole sequences per tile
        waves = (n_seq + per - 1) / per + (rows - n_seq * S + 31) / 32;
    }
    const int nb = (int)((waves + TILES_PER_WG - 1) / TILES_PER_WG);
    hipStream_t st = (hipStream_t)stream;
    static const int w_warm = getenv("DSVG_W_WARM") ? atoi(getenv("DSVG_W_WARM")) : 1;      // A/B knob: weight image into L2 up front
#define DSVG_ATTN_FWD(TR, TI)                                                                                          \
    do {                                                                                                               \
        const size_t lds = (size_t)NBUF * SLOT + SMALL_LDS + STAGE_LDS;                                                \
        DSVG_ENSURE_LDS((attn_block_fwd_kernel<TR, TI>), lds);                                                     \
        hipLaunchKernelGGL((attn_block_fwd_kernel<TR, TI>), dim3(nb), dim3(512), lds, st, (const bf16_t*)x,        \
                           (const bf16_t*)packed_layer, in_bias, out_bias, gamma, beta, key_mask, seq_off, tile_first, \
                           (int)n_seq, (int)S, (long long)rows, (bf16_t*)x1, (bf16_t*)xn_out, (bf16_t*)qkv_out,        \
                           (bf16_t*)ao_out, mean_out, rstd_out, eps, scale, drop_p, (const uint64_t*)seed, site_probs, \
                           site_res, (const bf16_t*)seq_add, (long long)seq_add_ld, site_seq_add, w_warm);                                   \
    } while (0)
    if (train) { if (tiled) DSVG_ATTN_FWD(true, true); else DSVG_ATTN_FWD(true, false); }
    else { if (tiled) DSVG_ATTN_FWD(false, true); else DSVG_ATTN_FWD(false, false); }
#undef DSVG_ATTN_FWD
    DSVG_LAUNCH_CHECK("attn_block_fwd");
    return 0;
}
