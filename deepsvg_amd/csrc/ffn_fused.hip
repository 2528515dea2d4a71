// Fused FFN sub-block of the pre-LN transformer layers, d_model = 256, dim_feedforward = 512, bf16 storage, fp32
// accumulation and statistics:
//     y = x + drop_r( W2 . drop_h( relu( W1 . LN(x) + b1 ) ) + b2 )
// (deepsvg/model/layers/improved_transformer.py:51-53,138-140; the LayerNorm is norm2 of the same layer, :51,:138).
// One launch replaces LayerNorm + linear1 GEMM + linear2 GEMM: the normalised rows and the 512-wide hidden activations
// never leave the chip, HBM sees 512 B in + 512 B out per token (AI 512 FLOP/B, MFMA-bound).
//
// Structure (gfx950, 64-lane waves, v_mfma_f32_32x32x16_bf16):
//   * token-stationary waves: a 512-thread workgroup owns 256 token rows, each of its 8 waves 32 of them.  A wave keeps
//     its LayerNorm-ed rows as the B operands of GEMM 1 in registers for the whole kernel (16 fragments = 64 VGPRs) and the
//     transposed output tile y^T[256 x 32 tokens] in 8 accumulators (128 registers).
//   * the weights stream: the hidden dimension is cut into 16 chunks of 32 units; per chunk the workgroup needs
//     W1[32 c .. 32 c + 31, :] (16 KiB) and W2[:, 32 c .. 32 c + 31] (16 KiB).  dsvg_ffn_pack stores them once per
//     optimiser step as ready-made MFMA A fragments (1 KiB each: lane l's 16 bytes at l * 16), so a chunk is ONE
//     contiguous 32 KiB block that goes L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 4 pieces per wave) into a ring of
//     NBUF chunk slots, issued NBUF - 1 chunks ahead behind counted s_waitcnt vmcnt + one s_barrier per chunk, and every
//     fragment read is a conflict-free ds_read_b128 at `slot + fragment * 1024 + lane * 16` (one address register).
//     The DMA is issued from inline asm: hipcc drains vmcnt(0) before the next ds_read after a *builtin* LDS-DMA
//     (it cannot tell the ring slots apart), which would serialise the prefetch.
//   * GEMM 1 is issued transposed (D = W1_frag x X_frag): a lane then holds, for its token row (lane & 31), the hidden
//     units 8 q + 4 (lane >> 5) + e of the chunk - exactly a B-operand register image for GEMM 2 once packed to bf16, if
//     the K index of GEMM 2 is taken in that order.  dsvg_ffn_pack lays the W2 fragments out in the same K order, so
//     the hidden tile goes accumulator -> bias / ReLU / dropout -> bf16 -> MFMA operand without touching LDS.
//   * epilogue: two v_permlane32_swap rounds turn each 32 x 32 accumulator tile into 16 consecutive output columns per
//     lane; bias, dropout, residual and the bf16 stores (32 contiguous bytes per lane and tile) run from registers.
//   * dropout draws of the hidden site ("v2", private to the fused kernels): ONE counter hash per 16 consecutive elements
//     and a one-multiply finaliser per pair of 16-bit draws (0.75 v_mul_lo_u32 per element instead of 1.5), still 16-bit
//     thresholds (p = 0.1 -> 6554 / 65536).  The ids are taken in the order the lane owns them (tok * 512 + 32 c + 16
//     (lane >> 5) + r); the backward pass never re-draws them, it reads the gate off the stored h (h > 0 <=> the unit
//     passed the ReLU and was kept).  The residual site uses the library's standard draws (dsvg_drop_apply replays it).
#include "fused_common.h"
#include "pack_images.h"
#include "../../include/dsvg.h"

namespace {

constexpr int FD = 256;                 // d_model
constexpr int FF = 512;                 // dim_feedforward
constexpr int CH = 32;                  // hidden units per chunk
constexpr int NCH = FF / CH;            // 16
constexpr int FWD_CHUNK = 32 * FRAG;    // [W1 chunk: 16 fragments | W2 chunk: 16 fragments]
constexpr int BWD_CHUNK = 48 * FRAG;    // [W1 chunk | W2^T chunk | W1^T chunk]
constexpr int TOK_PER_WG = 256;

using dsvg_pack::hidden_of;         // hidden unit (inside its chunk) that K slot (ks2, half, e) of GEMM 2 carries
using dsvg_pack::frag_pos;          // position of hidden unit j in the fragment-ordered h / dpre matrices (an involution)
static_assert(dsvg_pack::D == FD && dsvg_pack::F == FF && dsvg_pack::FFN_CH == CH && dsvg_pack::FFN_FWD_CHUNK == FWD_CHUNK &&
              dsvg_pack::FFN_BWD_CHUNK == BWD_CHUNK && dsvg_pack::FRAG_BYTES == FRAG, "pack_images.h restates these");

// ---------------------------------------------------------------------------------------------------------------------
// weight packing: fp32 master parameters -> bf16 fragment-major chunk images (one thread per lane slot of 8 elements),
// the LayerNorm affine folded into linear1 (W1' = W1 diag(gamma), b1' = b1 + W1 beta): bodies in pack_images.h, which the
// one-launch refresh of a training step (dsvg_pack_images) shares
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                                       int n_layers, bf16_t* __restrict__ fwd, bf16_t* __restrict__ bwd) {
    dsvg_pack::ffn_slot((long long)blockIdx.x * 256 + threadIdx.x, flat, offs, n_layers, fwd, bwd);
}
__global__ __launch_bounds__(256) void ffn_w2p_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                                      int n_layers, bf16_t* __restrict__ w2p) {
    dsvg_pack::ffn_w2p_slot((long long)blockIdx.x * 256 + threadIdx.x, flat, offs, n_layers, w2p);
}
__global__ __launch_bounds__(256) void ffn_fold_bias_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                                            int n_layers, float* __restrict__ b1f) {
    dsvg_pack::ffn_fold_bias_row(blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, flat, offs, n_layers, b1f);
}

// ---------------------------------------------------------------------------------------------------------------------
// dropout draws, scheme v2 (see the header): group of 16 consecutive ids, word i = two 16-bit draws (slots 2 i, 2 i + 1)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t drop2_group(const DropCtx& c, uint64_t g16) {
    const uint32_t h = dsvg_hash32((uint32_t)g16 ^ c.s0);
    return (h ^ c.s1) + (uint32_t)(g16 >> 32) * 0x9e3779b1u;
}
__device__ __forceinline__ uint32_t drop2_word(uint32_t h, uint32_t i) { return drop_word(h, i); }   // (dsvg_common.h)
// multipliers of the 8 ids 16 g16 + 8 hi .. + 7 (hi = 0, 1) given the group hash
__device__ __forceinline__ void drop2_mult8(const DropCtx& c, uint32_t h, int hi, float (&m)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = drop2_word(h, 4 * hi + i);
        m[2 * i] = (w & 0xffffu) < c.thresh ? 0.f : c.scale;
        m[2 * i + 1] = (w >> 16) < c.thresh ? 0.f : c.scale;
    }
}

typedef unsigned short rs_u16x2 __attribute__((ext_vector_type(2)));
typedef short rs_s16x2 __attribute__((ext_vector_type(2)));
typedef float rs_f2 __attribute__((ext_vector_type(2)));

// Activation of a hidden tile on PACKED pairs (round 5; 5.5 instructions per pair instead of 19): the 16 accumulator values
// of a lane -> the two B-operand fragments of G2.
//   scale (v_pk_mul_f32) -> bf16 pair (v_cvt_pk_bf16_f32) -> ReLU as a signed 16-bit maximum with 0 (v_pk_max_i16: a negative
//   bf16 is a negative int16) -> dropout: word k of the lane's group hash holds the draws of elements 2 k (low half) and
//   2 k + 1, drop when draw < thresh: r = sat(thresh - draw) is 0 for a kept element, else 0 - r >= 65536 - thresh >= 0x8000
//   exceeds every non-negative bf16 pattern (thresh <= 32768: dropout rates up to 0.5, checked by the host), so one more
//   saturating subtraction clears exactly the dropped halves (2 x v_pk_sub_u16 clamp + v_pk_sub_u16).
// The same draws as drop2_mult8 (ids in lane order); the scale is applied before instead of after the ReLU: results differ
// from the scalar form by fp32 rounding order only.
// bs (BIAS): the lane's 16 bias values ALREADY MULTIPLIED BY THE SCALE, element 4 q + e at bs[8 q + e] (LDS): the
// token-stationary kernels start their hidden accumulator from zero (an initial value would be 16 more live registers under
// G2) and add the bias here, fused with the scale: v_pk_fma_f32.
template <bool BIAS>
__device__ __forceinline__ void ffn_act_packed(const float (&v)[16], const DropCtx& dh, uint32_t hh, uint32_t (&pk)[8],
                                               const float* bs = nullptr) {
    const rs_f2 sc2 = {dh.scale, dh.scale};
    const rs_s16x2 zero = {0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BIAS) bb = *reinterpret_cast<const float4*>(bs + 8 * q);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = 2 * q + j;
            rs_f2 a = rs_f2{v[2 * k], v[2 * k + 1]};
            if (BIAS) a = __builtin_elementwise_fma(a, sc2, j ? rs_f2{bb.z, bb.w} : rs_f2{bb.x, bb.y});
            else a = a * sc2;
            pk[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(rs_s16x2, f2bf_pk(a[0], a[1])), zero));
        }
    }
    if (dh.on) {
        const rs_u16x2 tpair = {(unsigned short)dh.thresh, (unsigned short)dh.thresh};
        const rs_u16x2 z = {0, 0};
        // (stage by stage over the 8 words, not word by word: back-to-back dependent packed operations cost a wait state each)
        rs_u16x2 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __builtin_bit_cast(rs_u16x2, drop2_word(hh, k));
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __builtin_elementwise_sub_sat(tpair, r[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = z - r[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(rs_u16x2, pk[k]), r[k]));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
// Schedule of the chunk loop (per workgroup; `sync(k)` = s_waitcnt vmcnt(0) + s_barrier + DMA issue of chunk k + 2):
//   * DMA runs two chunks ahead: after sync(k) chunks <= k + 1 are in LDS, chunk k + 2 is in flight into the slot chunk
//     k - 1 has left (NBUF = 3), so fragment reads may run ahead into the next chunk without waiting for anything.
//   * the two waves of a SIMD (w, w + 4) run the same three stages G1 (16 dependent MFMAs: hidden tile), E1 (VALU:
//     bias / ReLU / dropout / bf16) and G2 (16 MFMAs into the 8 output accumulators) one stage apart:
//        waves 0-3:  sync(c) | G1(c)   E1(c)     G2(c)
//        waves 4-7:  sync(c) | G2(c)   G1(c + 1) E1(c + 1)         (same code, their barrier sits before G2)
//     so every VALU stage of one wave sits beside an MFMA stage of its partner instead of beside the partner's VALU
//     stage (waves that leave a barrier together otherwise stay in lockstep and the matrix pipe idles during E1).
//     (tried in round 3, both without effect on the 52 us of a 65,536-row inference launch: the barrier of waves 4-7 between
//     the two halves of E1 - half a period apart instead of a stage - and two accumulator chains in G1, which cost 16
//     registers and with them in-loop spills: 60 us)
//   * A fragments go through a 4-deep register ring that is refilled right behind each MFMA (prefetch distance 4 MFMAs,
//     continuous across stages and chunks): ds_read latency is covered by the wave's own MFMAs, not only by its partner.
//   * TRAIN: the kernel also hands the unfused backward what it needs - h (bf16 [T, 512], hidden columns in FRAGMENT
//     ORDER: position 32 c + 16 s + 8 b + 4 a + e holds unit 32 c + 16 s + 8 a + 4 b + e, i.e. exactly the 8 values a
//     lane owns per K step: one aligned 16-byte store each), xh = (x - mean) * rstd (bf16) and rstd (fp32).  vmcnt
//     counts stores too, out of order with respect to loads, so the only safe DMA wait is vmcnt(0); the h stores of a
//     chunk are therefore held back in 8 registers and issued right BEHIND the next sync point, which gives them (and
//     the DMA) a whole iteration to drain before the next wait.
// development probe (dsvg_ffn_debug_clock): when set, wave `w` of every workgroup stores s_memtime at four points - kernel
// start, LayerNorm done (first chunk sync ahead), chunk loop done, last store issued - into dbg[(block * 8 + w) * 4 ..]
// (bit 0 of the buffer address set: s_memrealtime - the constant 100 MHz reference counter, ONE time base for the whole chip, so
// that start / end stamps of different workgroups can be compared: dispatch ramp and tail of a launch - instead of s_memtime,
// the shader clock (one counter per XCD: only differences inside a wave mean anything))
__device__ __forceinline__ void ffn_stamp(unsigned long long* d, int slot) {
    if (d && (threadIdx.x & 63) == 0) {
        const bool real = (reinterpret_cast<uintptr_t>(d) & 1) != 0;
        // (an explicit GLOBAL pointer: the integer round trip would make this a flat_store, and a flat operation pending in
        // vmcnt - these waves never wait on it - turns every LDS wait of the kernel into lgkmcnt(0), LLVM SIInsertWaitcnts)
        typedef unsigned long long __attribute__((address_space(1))) * gptr_t;
        gptr_t b = (gptr_t)(reinterpret_cast<uintptr_t>(d) & ~(uintptr_t)1);
        b[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + slot] = real ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
    }
}

// development probe of ffn_bwd_dx_kernel (dsvg_ffn_debug_clock with bit 1 of the buffer address set; scripts/ffn_bwd_dx_probe.py): 8
// stamps of the chip-wide 100 MHz counter per wave - start, first chunk ready, K loop done, x rows landed + statistics, LayerNorm
// math done, residual rows landed, stores issued, masked pass done
__device__ __forceinline__ void bwd_stamp(unsigned long long* d, int slot) {
    if (d && (threadIdx.x & 63) == 0) {
        typedef unsigned long long __attribute__((address_space(1))) * gptr_t;
        gptr_t b = (gptr_t)(reinterpret_cast<uintptr_t>(d) & ~(uintptr_t)3);
        b[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    }
}

// The wave's 32 rows, one row per lane pair (lane half h owns the columns 16 ks + 8 h .. + 7 of every K step): LayerNorm
// without the affine part (folded into W1' / b1') -> the 16 B-operand fragments xf; TRAIN also stores xh and rstd.
// The statistics come from the packed words (ln_stats_packed, fused_common.h: the prologue is instruction-issue-bound).
// Called between the row loads and the first use of b1 / b2 (__syncthreads).
template <bool TRAIN, bool SYNC = true>
__device__ __forceinline__ void ffn_ln_rows(const bf16_t* __restrict__ x, int my_row, int half, bool st,
                                            bf16_t* __restrict__ xh_out, float* __restrict__ rstd_out, float eps,
                                            bf16x8 (&xf)[16]) {
    const char* xr = reinterpret_cast<const char*>(x) + (size_t)my_row * (FD * 2) + half * 16;
    uint4 raw[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) raw[ks] = *reinterpret_cast<const uint4*>(xr + 32 * ks);
    if (SYNC) __syncthreads();  // b1 / b2 staged (the row loads above are in flight meanwhile)
    float s, q, mean, rstd;
    ln_stats_packed(raw, s, q);
    ln_mean_rstd256(s, q, eps, mean, rstd);
    const float shift = -mean * rstd;
    char* xo = TRAIN ? reinterpret_cast<char*>(xh_out) + (size_t)my_row * (FD * 2) + half * 16 : nullptr;
    if (TRAIN && st && half == 0 && rstd_out) rstd_out[my_row] = rstd;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {       // (gamma / beta live in the packed W1' / b1', see ffn_pack_kernel)
        float v[8];
        unpack8(raw[ks], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], rstd, shift);
        Frag8 f;
        f.u = pack8(v);
        xf[ks] = f.v;
        if (TRAIN && st) *reinterpret_cast<uint4*>(xo + 32 * ks) = f.u;
    }
}

template <int NBUF, bool TRAIN, bool PK>
__global__ __launch_bounds__(512, 2) void ffn_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ img,
                                                         const float* __restrict__ b1, const float* __restrict__ b2,
                                                         bf16_t* __restrict__ y, bf16_t* __restrict__ h_out,
                                                         bf16_t* __restrict__ xh_out, float* __restrict__ rstd_out,
                                                         int M, float eps, float drop_p,
                                                         const uint64_t* __restrict__ seed, uint32_t site_h,
                                                         uint32_t site_r, int n_chunks, unsigned long long* dbg, int warm) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [NBUF chunk slots | b1 (2 KiB) | b2 (1 KiB)]
    ffn_stamp(dbg, 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    const int tok = lane & 31, half = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    float* sb1 = reinterpret_cast<float*>(smem + NBUF * FWD_CHUNK);
    float* sb2 = sb1 + FF;

    // ---- weight stream: this wave moves pieces 4 wave .. 4 wave + 3 of every chunk -----------------------------------
    const char* my_src = reinterpret_cast<const char*>(img) + wave * 4096;         // (wave-uniform; + lane * 16 per lane)
    const uint32_t my_dst = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
    auto issue = [&](int c) { dma4s(my_src + (size_t)c * FWD_CHUNK, (uint32_t)lane * 16u, my_dst + (uint32_t)(c % NBUF) * FWD_CHUNK); };
    // DMA distance: DIST = NBUF - 1 chunks ahead of the compute (NBUF = 3: 2, NBUF = 4: 3)
    constexpr int DIST = NBUF - 1;
    // Round 6: the layer's chunk image (512 KiB) is cold in this XCD's L2 when the launch starts (inside a training step every
    // layer's image is touched once per step) and the workgroups walk it in lockstep: every chunk would begin with one HBM miss
    // that all of them wait for.  So the workgroups of an XCD (block b runs on XCD b % 8) request the chunks behind the first
    // DIST up front, one each: an ordinary chunk DMA into the ring's last slot - this wave's own pieces of it, which its
    // own, later DMA of chunk DIST overwrites in order; nobody reads the slot before that one has landed.
    if (warm && NBUF == 4 && n_chunks > DIST) {
        const int wc = DIST + (int)((blockIdx.x >> 3) % (unsigned)(n_chunks - DIST));
        dma4s(my_src + (size_t)wc * FWD_CHUNK, (uint32_t)lane * 16u, my_dst + (uint32_t)DIST * FWD_CHUNK);
    }
#pragma unroll
    for (int c = 0; c < DIST; ++c)
        if (c < n_chunks) issue(c);

    // (PK: b1' is staged multiplied by the dropout scale, see ffn_act_packed)
    sb1[tid] = PK ? b1[tid] * drop_make(drop_p, seed, site_h).scale : b1[tid];
    if (tid < FD) sb2[tid] = b2[tid];

    // ---- the wave's 32 rows: LayerNorm in registers -> 16 B-operand fragments ------------------------------------------
    const int row0 = blockIdx.x * TOK_PER_WG + wave * 32;
    const int my_row = min(row0 + tok, M - 1);          // rows past M are computed on a clamped copy, never stored
    bf16x8 xf[16];
    ffn_ln_rows<TRAIN>(x, my_row, half, TRAIN && row0 + tok < M, xh_out, rstd_out, eps, xf);
    __builtin_amdgcn_sched_barrier(0);      // (the 128 accumulator zeroes below must not be hoisted above the LayerNorm)

    const DropCtx dh = drop_make(drop_p, seed, site_h);
    const DropCtx dr = drop_make(drop_p, seed, site_r);

    floatx16 yacc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[t][r] = 0.f;
    floatx16 hid;
    bf16x8 hf[2];
    uint4 ring[4];

    const char* lbase = smem + lane * 16;
    auto slot_of = [&](int c) -> const char* { return lbase + (c % NBUF) * FWD_CHUNK; };
    auto ld = [&](const char* p) -> uint4 { return *reinterpret_cast<const uint4*>(p); };

    // sync(k): afterwards chunks <= k + 1 are readable (this wave's pieces: counted vmcnt, the others': the barrier) and
    // chunk k + DIST is on its way into the slot chunk k - 1 has left; with NBUF = 4 chunk k + 2 stays in flight across
    // the barrier (4 pieces per wave and chunk)
    uint4 stash[2];                 // TRAIN: the packed h chunk waiting for its store slot
    int stash_c = -1;
    char* hrow = TRAIN ? reinterpret_cast<char*>(h_out) + (size_t)my_row * (FF * 2) + half * 16 : nullptr;
    const bool hst = TRAIN && row0 + tok < M;
    auto flush = [&]() {
        if (TRAIN && stash_c >= 0) {
            if (hst) {
                *reinterpret_cast<uint4*>(hrow + (CH * stash_c) * 2) = stash[0];
                *reinterpret_cast<uint4*>(hrow + (CH * stash_c + 16) * 2) = stash[1];
            }
            stash_c = -1;
        }
    };
    auto sync = [&](int k) {
        // (TRAIN: the h stores behind each sync are in flight too.  Loads return in order, so "at most 4 outstanding" still
        // means chunk k + 1 - older than the 4 DMA loads of chunk k + 2 - has landed; with the 3-slot ring there is no
        // younger load to count on and every sync has to drain the stores as well)
        if (DIST == 3 && k + 2 < n_chunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (k + DIST < n_chunks) issue(k + DIST);
        flush();
    };
    // G1: hid[unit][token] = sum_k W1[32 c + unit][k] xn[token][k]; the ring runs on into `cont` (4 more fragments)
    auto g2frag = [](int p) -> int { return 2 * (p & 7) + (p >> 3); };      // G2 position -> fragment of the W2 chunk
    auto G1 = [&](const char* w1, const char* cont) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Frag8 a;
            a.u = ring[ks & 3];
            hid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, xf[ks], hid, 0, 0, 0);
            if (ks + 4 < 16) ring[ks & 3] = ld(w1 + (ks + 4) * FRAG);
            else if (cont) ring[ks & 3] = ld(cont + g2frag(ks + 4 - 16) * FRAG);
            __builtin_amdgcn_sched_barrier(0);      // keep "MFMA n, refill slot n" order: hipcc otherwise sinks every read
        }                                           // to just before its MFMA (prefetch distance 0)
    };
    // E1: bias, ReLU, dropout (ids tok * 512 + 32 c + 16 half + r), bf16 -> the two K steps of G2's B operand
    auto E1 = [&](int c) {
        const uint64_t id0 = (uint64_t)(row0 + tok) * FF + (uint32_t)(CH * c + 16 * half);
        const uint32_t hh = dh.on ? drop2_group(dh, id0 >> 4) : 0u;
        if (PK) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = hid[r];
            uint32_t pk[8];
            ffn_act_packed<true>(v, dh, hh, pk, sb1 + CH * c + 4 * half);
            Frag8 f0, f1;
            f0.u = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            f1.u = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            hf[0] = f0.v; hf[1] = f1.v;
            if (TRAIN) { stash[0] = f0.u; stash[1] = f1.u; stash_c = c; }
            return;
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            float v[8];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * ks2 + qq;
                const float4 bb = *reinterpret_cast<const float4*>(sb1 + CH * c + 8 * q + 4 * half);
                v[4 * qq + 0] = fmaxf(hid[4 * q + 0] + bb.x, 0.f);
                v[4 * qq + 1] = fmaxf(hid[4 * q + 1] + bb.y, 0.f);
                v[4 * qq + 2] = fmaxf(hid[4 * q + 2] + bb.z, 0.f);
                v[4 * qq + 3] = fmaxf(hid[4 * q + 3] + bb.w, 0.f);
            }
            if (dh.on) {
                float m[8];
                drop2_mult8(dh, hh, ks2, m);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= m[e];
            }
            Frag8 f;
            f.u = pack8(v);
            hf[ks2] = f.v;
            if (TRAIN) stash[ks2] = f.u;
        }
        if (TRAIN) stash_c = c;
    };
    // G2: y[out][token] += sum_unit W2[out][32 c + unit] hid[unit][token]; the ring runs on into `cont`
    auto G2 = [&](const char* w2, const char* cont) {
        // position n: output tile n & 7, K step n >> 3 (fragment 2 (n & 7) + (n >> 3) of the packed chunk): the two MFMAs of
        // an accumulator are 8 apart instead of adjacent with a ring refill between them
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            Frag8 a;
            a.u = ring[n & 3];
            yacc[n & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, hf[n >> 3], yacc[n & 7], 0, 0, 0);
            if (n + 4 < 16) ring[n & 3] = ld(w2 + g2frag(n + 4) * FRAG);
            else if (cont) ring[n & 3] = ld(cont + (n + 4 - 16) * FRAG);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- the chunk loop ---------------------------------------------------------------------------------------------------
    // Both wave groups run the SAME straight-line code G1 E1 G2 per chunk; only the position of their one barrier per
    // chunk differs (before G1 for waves 0-3, before G2 for waves 4-7), which holds waves 4-7 one stage (G1 + E1) ahead.
    ffn_stamp(dbg, 1);
    if (n_chunks > 0) {
        if (DIST == 3 && n_chunks > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // chunks 0 and 1 are in LDS
        {
            const char* s0 = slot_of(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[i] = ld(s0 + i * FRAG);
        }
        for (int c = 0; c < n_chunks; ++c) {
            const char* sc = slot_of(c);
            const char* sn = (c + 1 < n_chunks) ? slot_of(c + 1) : nullptr;
            if (!late) sync(c);
            G1(sc, sc + 16 * FRAG);
            E1(c);
            if (late) sync(c);
            G2(sc + 16 * FRAG, sn);
        }
        flush();
    }
    ffn_stamp(dbg, 2);

    // ---- epilogue: + b2, dropout, + residual, bf16 rows ---------------------------------------------------------------
    const int m = row0 + tok;
    const bool live = m < M;
    const char* xres = reinterpret_cast<const char*>(x) + (size_t)my_row * (FD * 2);
    char* yrow = reinterpret_cast<char*>(y) + (size_t)my_row * (FD * 2);
    // the residual row (L2-hot: this workgroup read it in the prologue) in ONE batch of loads - the B-operand registers
    // are free by now - instead of one dependent round trip per tile
    uint4 res[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        res[2 * t] = *reinterpret_cast<const uint4*>(xres + (32 * t + 16 * half) * 2);
        res[2 * t + 1] = *reinterpret_cast<const uint4*>(xres + (32 * t + 16 * half + 8) * 2);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        uint32_t xc[4][4];
        tile_to_cols16(yacc[t], xc);
        const int n16 = 32 * t + 16 * half;
        uint4 pk[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float v[8], rv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(xc[2 * cb][e]); v[4 + e] = __uint_as_float(xc[2 * cb + 1][e]); }
            const float4 b0 = *reinterpret_cast<const float4*>(sb2 + n16 + 8 * cb);
            const float4 b1v = *reinterpret_cast<const float4*>(sb2 + n16 + 8 * cb + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1v.x; v[5] += b1v.y; v[6] += b1v.z; v[7] += b1v.w;
            if (dr.on) {        // residual site: the library's standard draws (ids m * 256 + column, groups of 8), so that
                float dm[8];    // dsvg_drop_apply replays the mask for the unfused backward GEMMs
                drop_mult8(dr, (uint64_t)m * FD + n16 + 8 * cb, dm);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dm[e];
            }
            unpack8(res[2 * t + cb], rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
            pk[cb] = pack8(v);
        }
        if (live) {
            *reinterpret_cast<uint4*>(yrow + n16 * 2) = pk[0];
            *reinterpret_cast<uint4*>(yrow + n16 * 2 + 16) = pk[1];
        }
    }
    ffn_stamp(dbg, 3);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward, half-size workgroups: the same per-wave program (32 rows: LayerNorm -> G1 E1 G2 per chunk -> epilogue, results
// bit-identical to ffn_fwd_kernel) in workgroups of 4 waves = 128 rows with a ring of four 16 KiB HALF chunks (the W1 part /
// the W2 part of a chunk; 67 KiB of LDS), so that two workgroups fit on a CU.  Used for launches of at most 32,768 rows:
// 256 of these workgroups put ONE wave on every SIMD of the chip where 128 of the 256-row workgroups put two waves on
// every SIMD of half the CUs (1,000 rows: 28 us against 39; at full size the two kernels are equal, and delaying the second
// workgroup of a CU so that its prologue runs beside its partner's chunk loop - tried with a per-CU turn counter keyed by
// HW_ID / XCC_ID, removed again - gained nothing: two 128-row workgroups stream the weights twice, which saturates the
// L2 -> LDS path, see HISTORY.md "Round 3").
//   sync(hc) in front of every G: half chunk hc + 1 has landed (the A ring runs on into it), half chunk hc + 3 is issued
//   into the slot half chunk hc - 1 has left; hc + 2 stays in flight across the barrier (counted wait: 4 pieces per wave).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int HSLOT = 16 * FRAG;        // half chunk: 16 fragments = 16 KiB
constexpr int HNBUF = 4;

template <bool TRAIN, bool PK>
__global__ __launch_bounds__(256, 2) void ffn_fwd_half_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ img,
                                                              const float* __restrict__ b1, const float* __restrict__ b2,
                                                              bf16_t* __restrict__ y, bf16_t* __restrict__ h_out,
                                                              bf16_t* __restrict__ xh_out, float* __restrict__ rstd_out,
                                                              int M, float eps, float drop_p,
                                                              const uint64_t* __restrict__ seed, uint32_t site_h,
                                                              uint32_t site_r, int n_chunks, unsigned long long* dbg, int warm) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [4 half-chunk slots | b1 (2 KiB) | b2 (1 KiB)]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (dbg && lane == 0) dbg[((size_t)blockIdx.x * 4 + wave) * 4 + 0] = __builtin_amdgcn_s_memtime();
    const int tok = lane & 31, half = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    float* sb1 = reinterpret_cast<float*>(smem + HNBUF * HSLOT);
    float* sb2 = sb1 + FF;
    const int n_half = 2 * n_chunks;

    // ---- weight stream: this wave moves pieces 4 wave .. 4 wave + 3 of every half chunk ------------------------------
    const char* my_src = reinterpret_cast<const char*>(img) + wave * 4096;
    const uint32_t my_dst = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
    auto issue = [&](int hc) { dma4s(my_src + (size_t)hc * HSLOT, (uint32_t)lane * 16u, my_dst + (uint32_t)(hc % HNBUF) * HSLOT); };
    // (round 6, as in ffn_fwd_kernel: the half chunks behind the first three requested into this XCD's L2 up front, one per
    // workgroup, into this wave's own pieces of the free ring slot)
    if (warm && n_half > 3) dma4s(my_src + (size_t)(3 + (int)((blockIdx.x >> 3) % (unsigned)(n_half - 3))) * HSLOT, (uint32_t)lane * 16u, my_dst + 3u * HSLOT);
#pragma unroll
    for (int hc = 0; hc < 3; ++hc)
        if (hc < n_half) issue(hc);

    {
        const float bsc = PK ? drop_make(drop_p, seed, site_h).scale : 1.f;
        sb1[tid] = b1[tid] * bsc;
        sb1[tid + 256] = b1[tid + 256] * bsc;
    }
    sb2[tid] = b2[tid];

    // ---- the wave's 32 rows: LayerNorm in registers -> 16 B-operand fragments (as in ffn_fwd_kernel) -----------------
    const int row0 = blockIdx.x * 128 + wave * 32;
    const int my_row = min(row0 + tok, M - 1);
    bf16x8 xf[16];
    ffn_ln_rows<TRAIN>(x, my_row, half, TRAIN && row0 + tok < M, xh_out, rstd_out, eps, xf);
    __builtin_amdgcn_sched_barrier(0);

    const DropCtx dh = drop_make(drop_p, seed, site_h);
    const DropCtx dr = drop_make(drop_p, seed, site_r);

    floatx16 yacc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[t][r] = 0.f;
    floatx16 hid;
    bf16x8 hf[2];
    uint4 ring[4];

    const char* lbase = smem + lane * 16;
    auto slot_of = [&](int hc) -> const char* { return lbase + (hc % HNBUF) * HSLOT; };
    auto ld = [&](const char* p) -> uint4 { return *reinterpret_cast<const uint4*>(p); };

    uint4 stash[2];
    int stash_c = -1;
    char* hrow = TRAIN ? reinterpret_cast<char*>(h_out) + (size_t)my_row * (FF * 2) + half * 16 : nullptr;
    const bool hst = TRAIN && row0 + tok < M;
    auto flush = [&]() {
        if (TRAIN && stash_c >= 0) {
            if (hst) {
                *reinterpret_cast<uint4*>(hrow + (CH * stash_c) * 2) = stash[0];
                *reinterpret_cast<uint4*>(hrow + (CH * stash_c + 16) * 2) = stash[1];
            }
            stash_c = -1;
        }
    };
    auto sync = [&](int hc) {
        // loads return in order: "at most 4 outstanding" means half chunk hc + 1 - older than the 4 pieces of hc + 2 - has
        // landed; the h stores behind a sync can only make the wait stricter
        if (hc + 2 < n_half) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (hc + 3 < n_half) issue(hc + 3);
        flush();
    };
    auto g2frag = [](int p) -> int { return 2 * (p & 7) + (p >> 3); };      // G2 position -> fragment of the W2 chunk
    auto G1 = [&](const char* w1, const char* cont) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Frag8 a;
            a.u = ring[ks & 3];
            hid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, xf[ks], hid, 0, 0, 0);
            if (ks + 4 < 16) ring[ks & 3] = ld(w1 + (ks + 4) * FRAG);
            else if (cont) ring[ks & 3] = ld(cont + g2frag(ks + 4 - 16) * FRAG);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto E1 = [&](int c) {
        const uint64_t id0 = (uint64_t)(row0 + tok) * FF + (uint32_t)(CH * c + 16 * half);
        const uint32_t hh = dh.on ? drop2_group(dh, id0 >> 4) : 0u;
        if (PK) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = hid[r];
            uint32_t pk[8];
            ffn_act_packed<true>(v, dh, hh, pk, sb1 + CH * c + 4 * half);
            Frag8 f0, f1;
            f0.u = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            f1.u = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            hf[0] = f0.v; hf[1] = f1.v;
            if (TRAIN) { stash[0] = f0.u; stash[1] = f1.u; stash_c = c; }
            return;
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            float v[8];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * ks2 + qq;
                const float4 bb = *reinterpret_cast<const float4*>(sb1 + CH * c + 8 * q + 4 * half);
                v[4 * qq + 0] = fmaxf(hid[4 * q + 0] + bb.x, 0.f);
                v[4 * qq + 1] = fmaxf(hid[4 * q + 1] + bb.y, 0.f);
                v[4 * qq + 2] = fmaxf(hid[4 * q + 2] + bb.z, 0.f);
                v[4 * qq + 3] = fmaxf(hid[4 * q + 3] + bb.w, 0.f);
            }
            if (dh.on) {
                float m[8];
                drop2_mult8(dh, hh, ks2, m);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= m[e];
            }
            Frag8 f;
            f.u = pack8(v);
            hf[ks2] = f.v;
            if (TRAIN) stash[ks2] = f.u;
        }
        if (TRAIN) stash_c = c;
    };
    auto G2 = [&](const char* w2, const char* cont) {
        // position n: output tile n & 7, K step n >> 3 (fragment 2 (n & 7) + (n >> 3) of the packed chunk): the two MFMAs of
        // an accumulator are 8 apart instead of adjacent with a ring refill between them
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            Frag8 a;
            a.u = ring[n & 3];
            yacc[n & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, hf[n >> 3], yacc[n & 7], 0, 0, 0);
            if (n + 4 < 16) ring[n & 3] = ld(w2 + g2frag(n + 4) * FRAG);
            else if (cont) ring[n & 3] = ld(cont + (n + 4 - 16) * FRAG);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (dbg && lane == 0) dbg[((size_t)blockIdx.x * 4 + wave) * 4 + 1] = __builtin_amdgcn_s_memtime();
    if (n_chunks > 0) {
        // half chunk 0 has to be readable before the ring is primed: everything but the newest 4 operations (the last
        // xh stores when training, else the pieces of half chunk 2) done, and the other waves' pieces behind the barrier
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const char* s0 = slot_of(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[i] = ld(s0 + i * FRAG);
        }
        for (int c = 0; c < n_chunks; ++c) {
            sync(2 * c);
            G1(slot_of(2 * c), slot_of(2 * c + 1));
            E1(c);
            sync(2 * c + 1);
            G2(slot_of(2 * c + 1), (c + 1 < n_chunks) ? slot_of(2 * c + 2) : nullptr);
        }
        flush();
    }
    if (dbg && lane == 0) dbg[((size_t)blockIdx.x * 4 + wave) * 4 + 2] = __builtin_amdgcn_s_memtime();

    // ---- epilogue: + b2, dropout, + residual, bf16 rows (as in ffn_fwd_kernel) ---------------------------------------
    const int m = row0 + tok;
    const bool live = m < M;
    const char* xres = reinterpret_cast<const char*>(x) + (size_t)my_row * (FD * 2);
    char* yrow = reinterpret_cast<char*>(y) + (size_t)my_row * (FD * 2);
    uint4 res[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        res[2 * t] = *reinterpret_cast<const uint4*>(xres + (32 * t + 16 * half) * 2);
        res[2 * t + 1] = *reinterpret_cast<const uint4*>(xres + (32 * t + 16 * half + 8) * 2);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        uint32_t xc[4][4];
        tile_to_cols16(yacc[t], xc);
        const int n16 = 32 * t + 16 * half;
        uint4 pk[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float v[8], rv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(xc[2 * cb][e]); v[4 + e] = __uint_as_float(xc[2 * cb + 1][e]); }
            const float4 b0 = *reinterpret_cast<const float4*>(sb2 + n16 + 8 * cb);
            const float4 b1v = *reinterpret_cast<const float4*>(sb2 + n16 + 8 * cb + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1v.x; v[5] += b1v.y; v[6] += b1v.z; v[7] += b1v.w;
            if (dr.on) {
                float dm[8];
                drop_mult8(dr, (uint64_t)m * FD + n16 + 8 * cb, dm);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dm[e];
            }
            unpack8(res[2 * t + cb], rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
            pk[cb] = pack8(v);
        }
        if (live) {
            *reinterpret_cast<uint4*>(yrow + n16 * 2) = pk[0];
            *reinterpret_cast<uint4*>(yrow + n16 * 2 + 16) = pk[1];
        }
    }
    if (dbg && lane == 0) dbg[((size_t)blockIdx.x * 4 + wave) * 4 + 3] = __builtin_amdgcn_s_memtime();
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, kernel 1 of 2: the hidden tile.  Per 32-token wave and hidden chunk c:
//     pre  = W1'[chunk] . xh            (recomputed: the forward pass stores nothing but x)
//     dh   = W2[:, chunk]^T . dym       (dym = dy * residual-dropout mask, replayed)
//     h    = drop_h(relu(pre + b1')),  dpre = dh * [pre + b1' > 0] * drop_h mask
// and writes, for the weight-gradient GEMMs and kernel 2: h and dpre (bf16 [T, 512], hidden columns in FRAGMENT ORDER:
// position 32 c + 16 s + 8 b + 4 a + e holds unit 32 c + 16 s + 8 a + 4 b + e - exactly the 8 values a lane owns per K
// step, so every store is one aligned 16-byte piece and kernel 2 reads its B operands back with one 16-byte load),
// xh = (x - mean) * rstd and dym (bf16 [T, 256]).  The two 16-deep MFMA chains (pre, dh) are independent and issued
// alternately, so no dependent MFMA sits right behind its predecessor's LDS read.
// LDS ring: [W1' chunk | W2^T chunk] = the first 32 KiB of every 48 KiB backward chunk.
// (stores are in flight inside the loop and vmcnt also counts them, out of order with respect to loads: every wait is 0)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void ffn_bwd_hidden_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                const bf16_t* __restrict__ img, const float* __restrict__ b1,
                                                                bf16_t* __restrict__ h_out, bf16_t* __restrict__ dpre_out,
                                                                bf16_t* __restrict__ xh_out, bf16_t* __restrict__ dym_out,
                                                                int M, float eps, float drop_p,
                                                                const uint64_t* __restrict__ seed, uint32_t site_h,
                                                                uint32_t site_r) {
    constexpr int NBUF = 3;
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [3 x 32 KiB | b1' (2 KiB)]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    const int tok = lane & 31, half = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    float* sb1 = reinterpret_cast<float*>(smem + NBUF * FWD_CHUNK);

    const char* my_src = reinterpret_cast<const char*>(img) + wave * 4096 + lane * 16;
    const uint32_t my_dst = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
    auto issue = [&](int c) { dma4(my_src + (size_t)c * BWD_CHUNK, my_dst + (uint32_t)(c % NBUF) * FWD_CHUNK); };
    issue(0);
    issue(1);
    sb1[tid] = b1[tid];

    const int row0 = blockIdx.x * TOK_PER_WG + wave * 32;
    const int m = row0 + tok;
    const bool live = m < M;
    const int my_row = min(m, M - 1);
    const DropCtx dh_ctx = drop_make(drop_p, seed, site_h);
    const DropCtx dr_ctx = drop_make(drop_p, seed, site_r);

    // ---- xh fragments (LayerNorm without the affine part, bit-identical to the forward kernel's) and their copy for the
    // weight-gradient GEMM ----
    bf16x8 xf[16];
    ffn_ln_rows<true, false>(x, my_row, half, live, xh_out, nullptr, eps, xf);
    // ---- dym fragments: dy with the residual dropout replayed (ids m * 256 + column, standard draws) ----------------------
    bf16x8 df[16];
    {
        const char* dr = reinterpret_cast<const char*>(dy) + (size_t)my_row * (FD * 2) + half * 16;
        char* dm_o = reinterpret_cast<char*>(dym_out) + (size_t)my_row * (FD * 2) + half * 16;
        uint4 raw[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) raw[ks] = *reinterpret_cast<const uint4*>(dr + 32 * ks);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Frag8 f;
            if (dr_ctx.on) {
                float v[8], mm[8];
                unpack8(raw[ks], v);
                drop_mult8(dr_ctx, (uint64_t)m * FD + 16 * ks + 8 * half, mm);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= mm[e];
                f.u = pack8(v);
                if (live) *reinterpret_cast<uint4*>(dm_o + 32 * ks) = f.u;
            } else {
                f.u = raw[ks];      // no dropout: dym == dy, the caller passes dy itself to the weight-gradient GEMM
            }
            df[ks] = f.v;
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    floatx16 pre, dh;
    uint4 ring[4];
    const char* lbase = smem + lane * 16;
    auto slot_of = [&](int c) -> const char* { return lbase + (c % NBUF) * FWD_CHUNK; };
    auto ld = [&](const char* p) -> uint4 { return *reinterpret_cast<const uint4*>(p); };
    // vmcnt counts the h / dpre stores too, and stores retire out of order with respect to loads: the only safe wait is
    // vmcnt(0).  It sits right behind the GEMMs of a chunk - BEFORE that chunk's stores are issued - so that it covers
    // the DMA issued one GEMM phase earlier and the stores of the PREVIOUS chunk, which have had a whole GEMM phase to
    // drain; the barrier itself then needs no wait (every wave confirmed its pieces of chunk k + 1 before reaching it).
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto sync = [&](int k) {
        __builtin_amdgcn_s_barrier();
        if (k + 2 < NCH) issue(k + 2);
    };
    // fragment stream of a chunk in consumption order: W1'[0], W2T[0], W1'[1], W2T[1], ... (fragment n -> n / 2 + 16 (n & 1))
    auto frag_at = [&](const char* sc, int n) -> const char* { return sc + ((n >> 1) + 16 * (n & 1)) * FRAG; };
    auto GEMMS = [&](const char* sc, const char* sn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { pre[r] = 0.f; dh[r] = 0.f; }
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            Frag8 a;
            a.u = ring[n & 3];
            if (n & 1) dh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, df[n >> 1], dh, 0, 0, 0);
            else pre = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, xf[n >> 1], pre, 0, 0, 0);
            if (n + 4 < 32) ring[n & 3] = ld(frag_at(sc, n + 4));
            else if (sn) ring[n & 3] = ld(frag_at(sn, n + 4 - 32));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto EPI = [&](int c) {
        const uint64_t id0 = (uint64_t)m * FF + (uint32_t)(CH * c + 16 * half);
        const uint32_t hh = dh_ctx.on ? drop2_group(dh_ctx, id0 >> 4) : 0u;
        char* ho = reinterpret_cast<char*>(h_out) + (size_t)my_row * (FF * 2) + (CH * c + 8 * half) * 2;
        char* po = reinterpret_cast<char*>(dpre_out) + (size_t)my_row * (FF * 2) + (CH * c + 8 * half) * 2;
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            float hv[8], pv[8], mm[8];
            if (dh_ctx.on) drop2_mult8(dh_ctx, hh, ks2, mm);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * ks2 + qq;
                const float4 bb = *reinterpret_cast<const float4*>(sb1 + CH * c + 8 * q + 4 * half);
                const float bq[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = pre[4 * q + e] + bq[e];
                    const float keep = dh_ctx.on ? mm[4 * qq + e] : 1.f;
                    const bool act = p > 0.f;
                    hv[4 * qq + e] = act ? p * keep : 0.f;
                    pv[4 * qq + e] = act ? dh[4 * q + e] * keep : 0.f;
                }
            }
            if (live) {
                *reinterpret_cast<uint4*>(ho + 32 * ks2) = pack8(hv);
                *reinterpret_cast<uint4*>(po + 32 * ks2) = pack8(pv);
            }
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // chunks 0 and 1 (and b1') are in LDS
    {
        const char* s0 = slot_of(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[i] = ld(frag_at(s0, i));
    }
    for (int c = 0; c < NCH; ++c) {
        const char* sc = slot_of(c);
        const char* sn = (c + 1 < NCH) ? slot_of(c + 1) : nullptr;
        if (!late) sync(c);
        GEMMS(sc, sn);
        landed();
        if (late) sync(c);
        EPI(c);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, kernel 2 of 2: dx = dy + LayerNorm'( dpre . W1' ).  Token-stationary like the forward kernel; the B operands
// (dpre in fragment order, see kernel 1) come straight from global memory, one chunk ahead; W1'^T chunks (the last 16
// KiB of every 48 KiB backward chunk) stream through a 4-slot LDS ring.  The LayerNorm backward needs no gamma (it is
// inside W1'): dx = dy + rstd * (g - mean(g) - xh * mean(g * xh)), g = dxh, statistics over the row in registers.
// ---------------------------------------------------------------------------------------------------------------------
template <bool PROBE>       // (PROBE: the stamps below; the production instantiation carries none of their code - they cost it 33 spilled registers)
__global__ __launch_bounds__(512, 2) void ffn_bwd_dx_kernel(const bf16_t* __restrict__ dpre, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ dy, const bf16_t* __restrict__ img,
                                                            bf16_t* __restrict__ dx, int M, float eps,
                                                            bf16_t* __restrict__ dxm, float drop_p,
                                                            const uint64_t* __restrict__ seed, uint32_t site_m, int warm,
                                                            unsigned long long* dbg) {
    constexpr int NBUF = 4, SLOT = 16 * FRAG;
    if (PROBE) bwd_stamp(dbg, 0);
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [4 x 16 KiB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    // this wave moves pieces 2 wave, 2 wave + 1 of every 16-piece chunk
    const char* my_src = reinterpret_cast<const char*>(img) + 32 * FRAG + wave * 2048 + lane * 16;
    const uint32_t my_dst = __builtin_amdgcn_readfirstlane(lds0 + wave * 2048);
    auto issue = [&](int c) { dma2(my_src + (size_t)c * BWD_CHUNK, my_dst + (uint32_t)(c % NBUF) * SLOT); };
    // (round 6, as in ffn_fwd_kernel: the W1'^T chunks are cold in this XCD's L2 and walked in lockstep - the workgroups of an
    // XCD request the chunks behind the first two up front, one each, into this wave's own pieces of ring slot 3)
    if (warm) dma2(my_src + (size_t)(2 + (blockIdx.x >> 3) % (unsigned)(NCH - 2)) * BWD_CHUNK, my_dst + 3u * SLOT);
    issue(0);
    issue(1);

    const int row0 = blockIdx.x * TOK_PER_WG + wave * 32;
    const int m = row0 + tok;
    const int my_row = min(m, M - 1);
    const char* pr = reinterpret_cast<const char*>(dpre) + (size_t)my_row * (FF * 2) + half * 16;
    auto bfrag = [&](int c, int ks2) -> uint4 { return *reinterpret_cast<const uint4*>(pr + (CH * c + 16 * ks2) * 2); };

    floatx16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // B operands: a register ring 4 chunks deep (a chunk is 0.25 us of MFMA work per wave, a global load 1 - 2 us under
    // load).  Per iteration this wave issues 2 DMA pieces (chunk c + 2) and 2 B loads (chunk c + 4), all loads, returned
    // in order: "everything but the 4 newest has landed" = chunk c's DMA (issued 2 iterations ago) and its B operands.
    uint4 bq[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { bq[i][0] = bfrag(i, 0); bq[i][1] = bfrag(i, 1); }
    const char* lbase = smem + lane * 16;
    auto chunk = [&](int c, uint4 (&b)[2], bool more, int pend) {
        // pend = loads this wave may leave in flight: the previous iteration's (4 in the steady state, 2 / 0 in the tail)
        if (pend == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // chunk c landed for everybody; the slot of chunk c - 2 is free again
        if (PROBE && c == 0) bwd_stamp(dbg, 1);
        if (c + 2 < NCH) issue(c + 2);
        const char* sc = lbase + (c % NBUF) * SLOT;
        Frag8 b0, b1;
        b0.u = b[0]; b1.u = b[1];
        if (more) { b[0] = bfrag(c + 4, 0); b[1] = bfrag(c + 4, 1); }
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            Frag8 a;
            a.u = *reinterpret_cast<const uint4*>(sc + n * FRAG);
            acc[n >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, (n & 1) ? b1.v : b0.v, acc[n >> 1], 0, 0, 0);
        }
    };
    for (int c = 0; c < NCH - 4; c += 4) {
        chunk(c, bq[0], true, 4); chunk(c + 1, bq[1], true, 4); chunk(c + 2, bq[2], true, 4); chunk(c + 3, bq[3], true, 4);
    }
    // tail: no more B loads, only the last two DMA issues (2 pieces each) are left to overlap with
    chunk(NCH - 4, bq[0], false, 0); chunk(NCH - 3, bq[1], false, 2); chunk(NCH - 2, bq[2], false, 2);
    chunk(NCH - 1, bq[3], false, 0);

    if (PROBE) bwd_stamp(dbg, 2);
    // ---- epilogue: LayerNorm backward on the rows in registers ------------------------------------------------------------
    // (scripts/ffn_bwd_dx_probe.py, 63,488 rows: first chunk ready after 6 us, K loop 20 us, then three serialised bursts of the
    // whole launch - x rows 9.5 us, residual rows 10.6 us, stores 10 us.  Round 6 tried to request the x rows before the loop:
    // a wave's loads return IN ORDER, so as its oldest loads they delayed the first chunk by what the epilogue saved (first chunk
    // ready after 16 us, x phase 2 us, step +1.5 %); behind the B operands of chunk c + 2 they can be at most 3 chunks early.)
    // tiles -> the lane's 16 consecutive columns per tile (in place), x row in the same layout
    const char* xrow = reinterpret_cast<const char*>(x) + (size_t)my_row * (FD * 2);
    uint4 xr[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        xr[2 * t] = *reinterpret_cast<const uint4*>(xrow + (32 * t + 16 * half) * 2);
        xr[2 * t + 1] = *reinterpret_cast<const uint4*>(xrow + (32 * t + 16 * half + 8) * 2);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        uint32_t xc[4][4];
        tile_to_cols16(acc[t], xc);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = __uint_as_float(xc[q][e]);      // column 4 q + e of the 16
    }
    float s, q, mean, rstd;
    ln_stats_packed(xr, s, q);
    ln_mean_rstd256(s, q, eps, mean, rstd);
    if (PROBE) bwd_stamp(dbg, 3);
    const float shift = -mean * rstd;       // xh = x * rstd + shift
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float v[8];
            unpack8(xr[2 * t + cb], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = acc[t][8 * cb + e];
                c1 += g;
                c2 += g * fmaf(v[e], rstd, shift);
            }
        }
    c1 += __shfl_xor(c1, 32, 64);
    c2 += __shfl_xor(c2, 32, 64);
    c1 *= (1.f / FD);
    c2 *= (1.f / FD);
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(xr[i].x), "+v"(xr[i].y), "+v"(xr[i].z), "+v"(xr[i].w));
    // w = rstd * (g - c1 - xh * c2) in place; then the x registers are dead and make room for dy
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float v[8];
            unpack8(xr[2 * t + cb], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[t][8 * cb + e] = rstd * (acc[t][8 * cb + e] - c1 - fmaf(v[e], rstd, shift) * c2);
        }
    // (the dy loads below must not be hoisted above the pass that frees the x registers: an address offset that is
    // opaque to the compiler - always 0 - and ordered behind the last value of that pass pins them here)
    if (PROBE) bwd_stamp(dbg, 4);
    uint32_t zoff = 0;
    asm volatile("" : "+v"(zoff) : "v"(acc[7][15]), "v"(acc[0][0]));
    const char* dyrow = reinterpret_cast<const char*>(dy) + (size_t)my_row * (FD * 2) + zoff;
    char* orow = reinterpret_cast<char*>(dx) + (size_t)my_row * (FD * 2);
    // optional second output: dx with the dropout mask of the attention sub-block's residual site replayed on it (what
    // dsvg_drop_apply would make of the rounded dx: the next two GEMMs of the backward pass read it)
    char* mrow = dxm ? reinterpret_cast<char*>(dxm) + (size_t)my_row * (FD * 2) : nullptr;
    uint4 dr[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        dr[2 * t] = *reinterpret_cast<const uint4*>(dyrow + (32 * t + 16 * half) * 2);
        dr[2 * t + 1] = *reinterpret_cast<const uint4*>(dyrow + (32 * t + 16 * half + 8) * 2);
    }
    if (PROBE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); bwd_stamp(dbg, 5); }
    if (m < M) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                float v[8];
                unpack8(dr[2 * t + cb], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += acc[t][8 * cb + e];
                *reinterpret_cast<uint4*>(orow + (32 * t + 16 * half + 8 * cb) * 2) = pack8(v);
            }
    }
    if (PROBE) bwd_stamp(dbg, 6);
    if (mrow && m < M) {
        // second pass, from the rows this lane has just stored (L2-hot; the registers above are all in use in the first
        // pass): wait for the stores, then read back through an address the compiler cannot match with them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t zo2 = 0;
        asm volatile("" : "+v"(zo2));
        const char* back = orow + zo2;
        const DropCtx dcm = drop_make(drop_p, seed, site_m);
#pragma unroll 2
        for (int i = 0; i < 16; ++i) {
            const int col = 32 * (i >> 1) + 16 * half + 8 * (i & 1);
            const uint4 pk = *reinterpret_cast<const uint4*>(back + col * 2);
            float w[8], mm[8];
            unpack8(pk, w);
            drop_mult8(dcm, (uint64_t)m * FD + col, mm);
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] *= mm[e];
            *reinterpret_cast<uint4*>(mrow + col * 2) = pack8(w);
        }
    }
    if (PROBE) bwd_stamp(dbg, 7);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradients, last step.  The two split-K GEMMs deliver G1p = dpre^T xh [512 (fragment order), 256], its row sums
// db1p [512 (fragment order)] and G2p = dym^T h [256, 512 (fragment order)]; this kernel undoes the fragment order
// (position p(j): bits 2 and 3 of j swapped) and the LayerNorm fold (see ffn_pack_kernel):
//     dW1[j][k] = gamma[k] G1p[p(j)][k] + beta[k] db1p[p(j)],   db1[j] = db1p[p(j)],   dW2[o][j] = G2p[o][p(j)],
//     dgamma[k] = sum_j W1[j][k] G1p[p(j)][k],                  dbeta[k] = sum_j W1[j][k] db1p[p(j)]
// One workgroup per 4 columns k (and 4 rows o of dW2); fixed summation order (deterministic).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ffn_wgrad_finish_body(const float* __restrict__ G1p, const float* __restrict__ db1p,
                                                      const float* __restrict__ G2p, const float* __restrict__ W1,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ dW1, float* __restrict__ db1,
                                                      float* __restrict__ dW2, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, int block) {
    __shared__ float red[2][4][256];
    const int t = threadIdx.x;
    const int k0 = block * 4;
    const float4 ga = *reinterpret_cast<const float4*>(gamma + k0);
    const float4 be = *reinterpret_cast<const float4*>(beta + k0);
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = t + 256 * jj, pj = frag_pos(j);
        const float4 g = *reinterpret_cast<const float4*>(G1p + (size_t)pj * FD + k0);
        const float4 w = *reinterpret_cast<const float4*>(W1 + (size_t)j * FD + k0);
        const float b = db1p[pj];
        *reinterpret_cast<float4*>(dW1 + (size_t)j * FD + k0) =
            make_float4(ga.x * g.x + be.x * b, ga.y * g.y + be.y * b, ga.z * g.z + be.z * b, ga.w * g.w + be.w * b);
        dg[0] += w.x * g.x; dg[1] += w.y * g.y; dg[2] += w.z * g.z; dg[3] += w.w * g.w;
        db[0] += w.x * b; db[1] += w.y * b; db[2] += w.z * b; db[3] += w.w * b;
        if (block == 0) db1[j] = b;
        // rows o = k0 .. k0 + 3 of dW2 (block < 64 covers all 256 rows)
#pragma unroll
        for (int oo = 0; oo < 4; ++oo) dW2[(size_t)(k0 + oo) * FF + j] = G2p[(size_t)(k0 + oo) * FF + pj];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[0][i][t] = dg[i]; red[1][i][t] = db[i]; }
    __syncthreads();
    if (t < 8) {        // fixed-order sums: thread t < 4 -> dgamma[k0 + t], t >= 4 -> dbeta[k0 + t - 4]
        const float* r = red[t >> 2][t & 3];
        float a = 0.f;
        for (int i = 0; i < 256; ++i) a += r[i];
        if (t < 4) dgamma[k0 + t] = a; else dbeta[k0 + t - 4] = a;
    }
}
__global__ __launch_bounds__(256) void ffn_wgrad_finish_kernel(const float* __restrict__ G1p, const float* __restrict__ db1p,
                                                               const float* __restrict__ G2p, const float* __restrict__ W1,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ dW1, float* __restrict__ db1,
                                                               float* __restrict__ dW2, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    ffn_wgrad_finish_body(G1p, db1p, G2p, W1, gamma, beta, dW1, db1, dW2, dgamma, dbeta, (int)blockIdx.x);
}
// the same for up to 16 layers in one launch (the table travels in the kernel arguments): a backward pass finishes all of
// its fused-FFN layers behind the one batched reduction launch
constexpr int FINISH_MAX = 16;
struct FinishTable { const float* in[FINISH_MAX][6]; float* out[FINISH_MAX][5]; };
__global__ __launch_bounds__(256) void ffn_wgrad_finish_many_kernel(const FinishTable t) {
    const int layer = blockIdx.x / (FD / 4), block = blockIdx.x % (FD / 4);
    ffn_wgrad_finish_body(t.in[layer][0], t.in[layer][1], t.in[layer][2], t.in[layer][3], t.in[layer][4], t.in[layer][5],
                          t.out[layer][0], t.out[layer][1], t.out[layer][2], t.out[layer][3], t.out[layer][4], block);
}


// ---------------------------------------------------------------------------------------------------------------------
// forward, role-specialised (round 5).  profiles/r04_coissue_probe.log: a wave that issues only MFMAs and a wave that issues
// only VALU instructions share a SIMD without slowing each other (32.2 cycles per MFMA beside ~6 cycles per VALU
// instruction), while two waves that MIX the two kinds serialise (ffn_fwd_kernel: 3,600-3,800 cycles per chunk for 2,048
// cycles of matrix work).  So the roles are split.  A 512-thread workgroup owns 128 rows:
//   * waves 0-3 ("matrix waves", one per SIMD) own 32 rows each and issue nothing but MFMAs, the ds_read_b128 of their A
//     fragments and the two small hand-offs below.  The wave computes the hidden tile of chunk i (G1(i), 16 MFMAs) INTERLEAVED
//     with the second product of chunk i - 2 (G2(i - 2), 16 MFMAs into the 8 output accumulators), so two MFMAs on the same
//     accumulator are never adjacent and the activation of chunk i - 1 is computed elsewhere at the same time;
//   * waves 4-7 ("vector waves", wave w + 4 sits on the SIMD of wave w) do everything else: the weight stream (LDS-DMA of
//     iteration block i + 3 = [W1' chunk i + 3 | W2 chunk i + 1], 8 pieces per wave), the activation E1(i - 1) of their
//     partner's hidden tile (scale, bf16, ReLU, dropout - all on PACKED pairs: v_pk_mul_f32, v_cvt_pk_bf16_f32,
//     v_pk_max_i16, and the dropout as two saturating 16-bit subtractions), the h stores of the training variant, and half
//     of the epilogue.
//   * hand-offs through LDS, lane to lane (the hidden tile's accumulator layout IS the B-operand layout of G2, see the file
//     header): hid (fp32, 4 KiB per pair) matrix -> vector at the end of an iteration, hf (bf16, 2 KiB) vector -> matrix.
//     Two s_barrier per iteration keep both single-buffered:   [A] both sides READ their hand-off  [B] both sides WRITE.
//     b1' is the initial value of the hidden accumulator and b2 of the output accumulators (no bias adds anywhere).
//   * weight ring: 4 slots of one 32 KiB iteration block; block i + 3 is issued behind A(i) into the slot block i - 1 has
//     left; at the end of iteration i a vector wave waits with vmcnt(8) - its 8 pieces of block i + 3 may stay in flight, so
//     blocks <= i + 2 have landed (loads return in order; the h stores in flight can only make the wait stricter) - so that
//     behind A(i + 1) the matrix waves may read blocks i + 1 AND i + 2: their A-fragment ring runs across the barrier.
//     The stream is cyclic (chunk & 15): the blocks past the last one are harmless re-reads that keep the counts uniform.
//   * 18 iterations per 128-row tile (2 to fill, 2 to drain the G1 -> E1 -> G2 pipeline); epilogue: the matrix wave hands
//     output tiles 4-7 to its partner through LDS (the ring is free by then) and finishes tiles 0-3 itself.
// Results: h / y differ from ffn_fwd_kernel's by fp32 summation order only (the bias is the accumulator's initial value
// and the dropout scale is applied before instead of after the ReLU): rounding-level, checked in tests/test_kernels_gpu.py.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RS_ROWS = 128;
constexpr int RS_BLOCK = 32 * FRAG;                 // [W1' chunk i | W2 chunk i - 2]
constexpr int RS_NB = 4;
constexpr int RS_HID = RS_NB * RS_BLOCK;            // 4 x 4 KiB
constexpr int RS_HF = RS_HID + 4 * 4096;            // 4 x 2 KiB
constexpr int RS_B1 = RS_HF + 4 * 2048;
constexpr int RS_B2 = RS_B1 + FF * 4;
constexpr int RS_LDS = RS_B2 + FD * 4;              // 158,720 B
constexpr int RS_DUMP_ROW = 128 * 4 + 16;           // epilogue hand-off: 128 fp32 columns per row, padded (bank spread)
constexpr int RS_ITERS = NCH + 2;

// iteration kinds: G1 only (fill), G1 + G2, G2 only (drain)
constexpr int RS_T1 = 1, RS_T2 = 2, RS_T12 = 3, RS_END = 0;
__host__ __device__ constexpr int rs_len(int ty) { return ty == RS_T12 ? 32 : (ty == RS_END ? 0 : 16); }
// position p of an iteration: which product (0 = G1, 1 = G2) and which of its 16 MFMAs.  G1 + G2: four G1 first (G2 waits
// for its B operand), then alternating, four G2 last (they cover the latency of the last G1 and the hid hand-off)
__host__ __device__ constexpr int rs_kind(int ty, int p) {
    return ty == RS_T1 ? 0 : ty == RS_T2 ? 1 : (p < 4 ? 0 : p >= 28 ? 1 : ((p & 1) ? 0 : 1));
}
__host__ __device__ constexpr int rs_idx(int ty, int p) {
    return ty != RS_T12 ? p : (p < 4 ? p : p >= 28 ? p - 16 : ((p & 1) ? 4 + (p - 5) / 2 : (p - 4) / 2));
}
// fragment of the 32-fragment block that position p multiplies (G2 position n: output tile n & 7, K step n >> 3)
__host__ __device__ constexpr int rs_frag(int ty, int p) {
    return rs_kind(ty, p) == 0 ? rs_idx(ty, p) : 16 + 2 * (rs_idx(ty, p) & 7) + (rs_idx(ty, p) >> 3);
}
__host__ __device__ constexpr bool rs_has1(int ty) { return ty == RS_T1 || ty == RS_T12; }
__host__ __device__ constexpr bool rs_has2(int ty) { return ty == RS_T2 || ty == RS_T12; }

// development probe of the role-specialised kernel (dsvg_ffn_debug_clock): 16 stamps per wave at dbg[(block * 8 + wave) * 16 ..]:
// 0 start, 1 before A(0), 2 / 3 / 4 behind iterations 1 / 8 / 15, 5 behind E_A, 6 behind E_B, 7 end (bit 0 of the address:
// s_memrealtime, see ffn_stamp)
__device__ __forceinline__ void rs_stamp(unsigned long long* d, int slot) {
    if (d && (threadIdx.x & 63) == 0) {
        const bool real = (reinterpret_cast<uintptr_t>(d) & 1) != 0;
        typedef unsigned long long __attribute__((address_space(1))) * gptr_t;
        gptr_t b = (gptr_t)(reinterpret_cast<uintptr_t>(d) & ~(uintptr_t)1);
        b[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + slot] = real ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
    }
}

// one iteration of a matrix wave (behind barrier A(i); ends with barrier A(i + 1))
template <int TY, int NX>
__device__ __forceinline__ void rs_matrix_iter(int i, const char* lbase, const bf16x8 (&xf)[16], floatx16 (&yacc)[8],
                                               floatx16& hid, uint4 (&ring)[4], char* hid_buf, const char* hf_buf,
                                               const float* sb1h) {
    constexpr int L = rs_len(TY);
    constexpr int PB = 4;                                   // barrier B behind this position (the first G2 has its operand)
    constexpr int PW = TY == RS_T12 ? 29 : 15;              // hid hand-off behind this position (2 MFMAs behind the last G1)
    const char* cur = lbase + (i & (RS_NB - 1)) * RS_BLOCK;
    const char* nxt = lbase + ((i + 1) & (RS_NB - 1)) * RS_BLOCK;
    Frag8 hf0, hf1;
    if (rs_has2(TY)) {
        hf0.u = *reinterpret_cast<const uint4*>(hf_buf);
        hf1.u = *reinterpret_cast<const uint4*>(hf_buf + FRAG);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < L; ++p) {
        Frag8 a;
        a.u = ring[p & 3];
        const int n = rs_idx(TY, p);
        if (rs_kind(TY, p) == 0) hid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, xf[n], hid, 0, 0, 0);
        else yacc[n & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, (n >> 3) ? hf1.v : hf0.v, yacc[n & 7], 0, 0, 0);
        if (p + 4 < L) ring[p & 3] = *reinterpret_cast<const uint4*>(cur + rs_frag(TY, p + 4) * FRAG);
        else if (NX != RS_END) ring[p & 3] = *reinterpret_cast<const uint4*>(nxt + rs_frag(NX, p + 4 - L) * FRAG);
        __builtin_amdgcn_sched_barrier(0);
        if (p == PB) {
            // both hf reads have returned: LDS operations of a wave return in order and exactly PB + 1 fragment reads were
            // issued behind them
            if (rs_has2(TY)) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // B(i): every hand-off buffer has been read
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rs_has1(TY) && p == PW) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(hid_buf + q * FRAG) = make_float4(hid[4 * q], hid[4 * q + 1], hid[4 * q + 2], hid[4 * q + 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (rs_has1(NX)) {                              // the next hidden tile starts from b1'
                const float* b = sb1h + CH * (i + 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bb = *reinterpret_cast<const float4*>(b + 8 * q);
                    hid[4 * q] = bb.x; hid[4 * q + 1] = bb.y; hid[4 * q + 2] = bb.z; hid[4 * q + 3] = bb.w;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // A(i + 1)
    __builtin_amdgcn_sched_barrier(0);
}

template <bool TRAIN>
__global__ __launch_bounds__(512, 2) void ffn_fwd_rs_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ img,
                                                            const float* __restrict__ b1, const float* __restrict__ b2,
                                                            bf16_t* __restrict__ y, bf16_t* __restrict__ h_out,
                                                            bf16_t* __restrict__ xh_out, float* __restrict__ rstd_out,
                                                            int M, float eps, float drop_p,
                                                            const uint64_t* __restrict__ seed, uint32_t site_h,
                                                            uint32_t site_r, unsigned long long* dbg, int probe) {
    // probe (DSVG_FFN_RS_PROBE, timing experiments only - the results are wrong): 1 no activation arithmetic, 2 no weight
    // DMA inside the loop, 4 no h stores
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    rs_stamp(dbg, 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave & 3;                                // the pair's 32-row tile
    const int tok = lane & 31, half = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    float* sb1 = reinterpret_cast<float*>(smem + RS_B1);
    float* sb2 = reinterpret_cast<float*>(smem + RS_B2);
    sb1[tid] = b1[tid];
    if (tid < FD) sb2[tid] = b2[tid];
    const int row0 = blockIdx.x * RS_ROWS + pw * 32;
    const int m = row0 + tok;
    const bool live = m < M;
    const int my_row = min(m, M - 1);                       // rows past M are computed on a clamped copy, never stored
    const char* lbase = smem + lane * 16;
    char* hid_buf = smem + RS_HID + pw * 4096 + lane * 16;
    char* hf_buf = smem + RS_HF + pw * 2048 + lane * 16;
    char* dump = smem + pw * (32 * RS_DUMP_ROW);            // epilogue: the pair's output tiles 4-7, fp32 [32 rows][128]
    const DropCtx dr = drop_make(drop_p, seed, site_r);
    char* yrow = reinterpret_cast<char*>(y) + (size_t)my_row * (FD * 2);

    if (wave < 4) {
        // ================================================ matrix wave ================================================
        bf16x8 xf[16];
        ffn_ln_rows<TRAIN, false>(x, my_row, half, TRAIN && live, xh_out, rstd_out, eps, xf);
        // (the normalised fragments are first USED in the chunk loop: without an opaque use here the compiler sinks their
        // computation behind the accumulators' initial values, with 128 unpacked row values live across them - spills)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Frag8 f;
            f.v = xf[ks];
            asm volatile("" : "+v"(f.u.x), "+v"(f.u.y), "+v"(f.u.z), "+v"(f.u.w));
            xf[ks] = f.v;
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                    // P: b1' / b2 staged
        __builtin_amdgcn_sched_barrier(0);
        floatx16 yacc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4*>(sb2 + 32 * t + 8 * q + 4 * half);
                yacc[t][4 * q] = bb.x; yacc[t][4 * q + 1] = bb.y; yacc[t][4 * q + 2] = bb.z; yacc[t][4 * q + 3] = bb.w;
            }
        floatx16 hid;
        const float* sb1h = sb1 + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(sb1h + 8 * q);
            hid[4 * q] = bb.x; hid[4 * q + 1] = bb.y; hid[4 * q + 2] = bb.z; hid[4 * q + 3] = bb.w;
        }
        uint4 ring[4];
        __builtin_amdgcn_sched_barrier(0);
        rs_stamp(dbg, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // A(0): blocks 0 .. 2 have landed
#pragma unroll
        for (int p = 0; p < 4; ++p) ring[p] = *reinterpret_cast<const uint4*>(lbase + rs_frag(RS_T1, p) * FRAG);
        __builtin_amdgcn_sched_barrier(0);
        rs_matrix_iter<RS_T1, RS_T1>(0, lbase, xf, yacc, hid, ring, hid_buf, hf_buf, sb1h);
        rs_matrix_iter<RS_T1, RS_T12>(1, lbase, xf, yacc, hid, ring, hid_buf, hf_buf, sb1h);
        rs_stamp(dbg, 2);
        for (int i = 2; i < NCH - 1; ++i) {
            rs_matrix_iter<RS_T12, RS_T12>(i, lbase, xf, yacc, hid, ring, hid_buf, hf_buf, sb1h);
            if (i == 8) rs_stamp(dbg, 3);
        }
        rs_matrix_iter<RS_T12, RS_T2>(NCH - 1, lbase, xf, yacc, hid, ring, hid_buf, hf_buf, sb1h);
        rs_stamp(dbg, 4);
        // the residual rows of the tiles this wave finishes itself (the B-operand registers are free now)
        const char* xres = reinterpret_cast<const char*>(x) + (size_t)my_row * (FD * 2);
        uint4 res[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            res[2 * t] = *reinterpret_cast<const uint4*>(xres + (32 * t + 16 * half) * 2);
            res[2 * t + 1] = *reinterpret_cast<const uint4*>(xres + (32 * t + 16 * half + 8) * 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        rs_matrix_iter<RS_T2, RS_T2>(NCH, lbase, xf, yacc, hid, ring, hid_buf, hf_buf, sb1h);
        rs_matrix_iter<RS_T2, RS_END>(NCH + 1, lbase, xf, yacc, hid, ring, hid_buf, hf_buf, sb1h);     // ends with E_A
        rs_stamp(dbg, 5);
        // ---- epilogue: tiles 4-7 to the partner (fp32, row `tok`, column 32 (t - 4) + 8 q + 4 half + e) ----------------
#pragma unroll
        for (int t = 4; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(dump + tok * RS_DUMP_ROW + (32 * (t - 4) + 8 * q + 4 * half) * 4) =
                    make_float4(yacc[t][4 * q], yacc[t][4 * q + 1], yacc[t][4 * q + 2], yacc[t][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // E_B
        rs_stamp(dbg, 6);
        // tiles 0-3: dropout, + residual, bf16 rows (as ffn_fwd_kernel's epilogue, b2 is already inside)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t xc[4][4];
            tile_to_cols16(yacc[t], xc);
            const int n16 = 32 * t + 16 * half;
            uint4 pk[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                float v[8], rv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(xc[2 * cb][e]); v[4 + e] = __uint_as_float(xc[2 * cb + 1][e]); }
                unpack8(res[2 * t + cb], rv);
                if (dr.on) {
                    float dm[8];
                    drop_mult8(dr, (uint64_t)m * FD + n16 + 8 * cb, dm);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], dm[e], rv[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                pk[cb] = pack8(v);
            }
            if (live) {
                *reinterpret_cast<uint4*>(yrow + n16 * 2) = pk[0];
                *reinterpret_cast<uint4*>(yrow + n16 * 2 + 16) = pk[1];
            }
        }
        rs_stamp(dbg, 7);
    } else {
        // ================================================ vector wave ================================================
        // weight stream: this wave moves bytes [8 KiB v, 8 KiB (v + 1)) of every 32 KiB block: v = 0, 1 the W1' half of
        // chunk `block`, v = 2, 3 the W2 half of chunk `block - 2` (cyclic)
        const char* my_src = reinterpret_cast<const char*>(img) + pw * 8192;
        const uint32_t my_dst = __builtin_amdgcn_readfirstlane(lds0 + pw * 8192);
        const int lag = pw < 2 ? 0 : 2;
        auto issue = [&](int blk) {
            const char* s = my_src + (size_t)((blk - lag) & (NCH - 1)) * FWD_CHUNK;
            const uint32_t d = my_dst + (uint32_t)(blk & (RS_NB - 1)) * RS_BLOCK;
            dma4s(s, (uint32_t)lane * 16u, d);
            dma4s(s + 4096, (uint32_t)lane * 16u, d + 4096);
        };
        issue(0);
        issue(1);
        issue(2);
        // the residual rows of the output tiles this wave finishes (columns 16 ks + 8 half .. + 7, ks = 8 .. 15)
        const char* xr = reinterpret_cast<const char*>(x) + (size_t)my_row * (FD * 2) + half * 16;
        uint4 res[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) res[k] = *reinterpret_cast<const uint4*>(xr + 32 * (8 + k));
        const DropCtx dh = drop_make(drop_p, seed, site_h);
        // hidden-site draws: group of 16 ids = (row0 + tok) * 512 + 32 c + 16 half .. + 15 -> g = (row0 + tok) * 32 + 2 c + half
        const uint64_t g0 = (uint64_t)m * 32 + half;
        const uint32_t g0lo = (uint32_t)g0, ghi_term = (uint32_t)(g0 >> 32) * 0x9e3779b1u;
        char* hrow = TRAIN ? reinterpret_cast<char*>(h_out) + (size_t)my_row * (FF * 2) + half * 16 : nullptr;
        __syncthreads();                                    // P
        rs_stamp(dbg, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // A(0)
        for (int i = 0; i < RS_ITERS; ++i) {
            const bool act = i >= 1 && i <= NCH;            // E1(i - 1)
            uint4 hv[4];
            if (act) {
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const uint4*>(hid_buf + q * FRAG);
            }
            if (i < NCH && !(probe & 2)) issue(i + 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // B(i)
            if (act && !(probe & 1)) {
                const int c = i - 1;
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[4 * q] = __uint_as_float(hv[q].x); v[4 * q + 1] = __uint_as_float(hv[q].y);
                    v[4 * q + 2] = __uint_as_float(hv[q].z); v[4 * q + 3] = __uint_as_float(hv[q].w);
                }
                uint32_t pk[8];
                const uint32_t hh = dh.on ? (dsvg_hash32((g0lo + 2u * (uint32_t)c) ^ dh.s0) ^ dh.s1) + ghi_term : 0u;
                ffn_act_packed<false>(v, dh, hh, pk);
                const uint4 f0 = make_uint4(pk[0], pk[1], pk[2], pk[3]), f1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                *reinterpret_cast<uint4*>(hf_buf) = f0;
                *reinterpret_cast<uint4*>(hf_buf + FRAG) = f1;
                if (TRAIN && live && !(probe & 4)) {
                    *reinterpret_cast<uint4*>(hrow + (CH * c) * 2) = f0;
                    *reinterpret_cast<uint4*>(hrow + (CH * c + 16) * 2) = f1;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i < NCH) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (i == RS_ITERS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // A(i + 1); the last one is E_A
            if (i == 1) rs_stamp(dbg, 2);
            else if (i == 8) rs_stamp(dbg, 3);
            else if (i == NCH - 1) rs_stamp(dbg, 4);
        }
        rs_stamp(dbg, 5);
        __builtin_amdgcn_s_barrier();                       // E_B: the partner's tiles 4-7 are in LDS
        rs_stamp(dbg, 6);
        // output columns 32 t + 16 p + 8 half .. + 7 (t = 4 .. 7, p = 0, 1): the lane's LayerNorm-layout piece ks = 2 t + p
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = 16 * (8 + k) + 8 * half;
            const char* d = dump + tok * RS_DUMP_ROW + (16 * k + 8 * half) * 4;
            const float4 a0 = *reinterpret_cast<const float4*>(d), a1 = *reinterpret_cast<const float4*>(d + 16);
            float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, rv[8];
            unpack8(res[k], rv);
            if (dr.on) {
                float dm[8];
                drop_mult8(dr, (uint64_t)m * FD + col, dm);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], dm[e], rv[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            if (live) *reinterpret_cast<uint4*>(yrow + col * 2) = pack8(v);
        }
        rs_stamp(dbg, 7);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int64_t dsvg_ffn_pack_bytes(int32_t n_layers, int32_t which) {
    return (int64_t)n_layers * NCH * (which == 0 ? FWD_CHUNK : BWD_CHUNK);
}

extern "C" int dsvg_ffn_pack(const float* flat_f32, const int64_t* offs, int32_t n_layers, int32_t d_model, int32_t d_ff,
                             void* packed_fwd, void* packed_bwd, float* b1_folded, void* w2p, void* stream) {
    DSVG_CHECK_ARG(flat_f32 && offs && packed_fwd && packed_bwd && b1_folded, "ffn_pack: null pointer");
    DSVG_CHECK_ARG(d_model == FD && d_ff == FF, "ffn_pack: the fused FFN kernels are built for d_model 256 / dim_ff 512");
    DSVG_CHECK_ARG(n_layers > 0, "ffn_pack: bad layer count");
    const long long n = (long long)n_layers * NCH * 80 * 64;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat_f32, offs,
                       n_layers, (bf16_t*)packed_fwd, (bf16_t*)packed_bwd);
    hipLaunchKernelGGL(ffn_fold_bias_kernel, dim3((unsigned)((n_layers * FF + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       flat_f32, offs, n_layers, b1_folded);
    if (w2p) {
        const long long n8 = (long long)n_layers * (FD * FF / 8);
        hipLaunchKernelGGL(ffn_w2p_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat_f32, offs,
                           n_layers, (bf16_t*)w2p);
    }
    DSVG_LAUNCH_CHECK("ffn_pack");
    return 0;
}

static unsigned long long* g_ffn_dbg_host = nullptr;      // development probe, see dsvg_ffn_debug_clock

extern "C" int dsvg_ffn_fwd(const void* x, const void* packed_fwd_layer, const float* b1_folded, const float* b2, void* y,
                            void* h_out, void* xh_out, float* rstd_out, int64_t rows, float eps, float drop_p,
                            uint32_t site_hidden, uint32_t site_res, const void* seed, int32_t stages, void* stream) {
    DSVG_CHECK_ARG(x && packed_fwd_layer && b1_folded && b2 && y, "ffn_fwd: null pointer");
    DSVG_CHECK_ARG(rows > 0 && rows < (1ll << 31) - TOK_PER_WG, "ffn_fwd: bad row count");
    DSVG_CHECK_ARG(!(drop_p > 0.f) || seed, "ffn_fwd: dropout needs a seed");
    const bool train = h_out != nullptr;
    DSVG_CHECK_ARG(!train || (xh_out && rstd_out), "ffn_fwd: h_out, xh_out and rstd_out come together");
    DSVG_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)packed_fwd_layer | (uintptr_t)h_out | (uintptr_t)xh_out) & 15) == 0,
                   "ffn_fwd: operands must be 16-byte aligned");
    const int nb = (int)((rows + TOK_PER_WG - 1) / TOK_PER_WG);
    hipStream_t st = (hipStream_t)stream;
    const int stages_arg = stages;       // 0 default, 2 half-size workgroups, 3 / 4 ring slots of the 256-row kernel
    if (stages == 0 || stages == 2 || stages >= 5) stages = 4;
    // training variant: 4 slots (chunk k + 2's DMA in flight across the syncs, counted waits) or 3 (every sync drains the
    // wave's h stores too); DSVG_FFN_TRAIN_STAGES for the A/B
    static const int train_stages = getenv("DSVG_FFN_TRAIN_STAGES") ? atoi(getenv("DSVG_FFN_TRAIN_STAGES")) : 4;
    if (train) stages = train_stages == 3 ? 3 : 4;
    // timing probe only (results are wrong below 16): number of hidden chunks actually processed
    static const int dbg_chunks = getenv("DSVG_FFN_DBG_CHUNKS") ? atoi(getenv("DSVG_FFN_DBG_CHUNKS")) : NCH;
    static const int w_warm = getenv("DSVG_W_WARM") ? atoi(getenv("DSVG_W_WARM")) : 1;      // A/B knob: weight image into L2 up front
#define DSVG_FFN_FWD(NB, TR, PK)                                                                                      \
    do {                                                                                                              \
        const size_t lds = (size_t)NB * FWD_CHUNK + 3072;                                                             \
        DSVG_ENSURE_LDS((ffn_fwd_kernel<NB, TR, PK>), lds);                                                           \
        hipLaunchKernelGGL((ffn_fwd_kernel<NB, TR, PK>), dim3(nb), dim3(512), lds, st, (const bf16_t*)x,               \
                           (const bf16_t*)packed_fwd_layer, b1_folded, b2, (bf16_t*)y, (bf16_t*)h_out, (bf16_t*)xh_out,\
                           rstd_out, (int)rows, eps, drop_p, (const uint64_t*)seed, site_hidden, site_res, dbg_chunks,    \
                           g_ffn_dbg_host, w_warm);                                                                   \
    } while (0)
    // half-size workgroups (stages 2, or by default up to 32,768 rows; DSVG_FFN_HALF=0 / 1 forces the choice)
    static const int half_env = getenv("DSVG_FFN_HALF") ? atoi(getenv("DSVG_FFN_HALF")) : -1;
    // packed activation (ffn_act_packed; dropout rates up to 0.5): stages 6 / 7 = the 256-row / half-size workgroups with it,
    // 0 = the library's choice (DSVG_FFN_PACKED=0 / 1 forces it for the default)
    static const int pk_env = getenv("DSVG_FFN_PACKED") ? atoi(getenv("DSVG_FFN_PACKED")) : 1;
    const bool pk = !(drop_p > 0.5f) && (stages_arg == 6 || stages_arg == 7 || (stages_arg == 0 && pk_env != 0));
    const bool half_on = stages_arg == 2 || stages_arg == 7 || (stages_arg == 0 && (half_env >= 0 ? half_env != 0 : rows <= 32768));
    if (half_on) {
        const int nbh = (int)((rows + 127) / 128);
        const size_t lds = (size_t)HNBUF * HSLOT + 3072;
#define DSVG_FFN_FWD_HALF(TR, PK)                                                                                     \
    do {                                                                                                              \
        DSVG_ENSURE_LDS((ffn_fwd_half_kernel<TR, PK>), lds);                                                          \
        hipLaunchKernelGGL((ffn_fwd_half_kernel<TR, PK>), dim3(nbh), dim3(256), lds, st, (const bf16_t*)x,             \
                           (const bf16_t*)packed_fwd_layer, b1_folded, b2, (bf16_t*)y, (bf16_t*)h_out, (bf16_t*)xh_out,\
                           rstd_out, (int)rows, eps, drop_p, (const uint64_t*)seed, site_hidden, site_res, dbg_chunks,    \
                           g_ffn_dbg_host, w_warm);                                                                   \
    } while (0)
        if (train && pk) DSVG_FFN_FWD_HALF(true, true);
        else if (train) DSVG_FFN_FWD_HALF(true, false);
        else if (pk) DSVG_FFN_FWD_HALF(false, true);
        else DSVG_FFN_FWD_HALF(false, false);
#undef DSVG_FFN_FWD_HALF
        DSVG_LAUNCH_CHECK("ffn_fwd (half-size workgroups)");
        return 0;
    }
    if (stages_arg == 5) {      // role-specialised 128-row workgroups (matrix waves + vector waves)
        DSVG_CHECK_ARG(!(drop_p > 0.5f), "ffn_fwd: the role-specialised kernel takes dropout rates up to 0.5");
        const int nbr = (int)((rows + RS_ROWS - 1) / RS_ROWS);
        static const int rs_probe = getenv("DSVG_FFN_RS_PROBE") ? atoi(getenv("DSVG_FFN_RS_PROBE")) : 0;
#define DSVG_FFN_FWD_RS(TR)                                                                                           \
    do {                                                                                                              \
        DSVG_ENSURE_LDS((ffn_fwd_rs_kernel<TR>), RS_LDS);                                                             \
        hipLaunchKernelGGL((ffn_fwd_rs_kernel<TR>), dim3(nbr), dim3(512), RS_LDS, st, (const bf16_t*)x,                \
                           (const bf16_t*)packed_fwd_layer, b1_folded, b2, (bf16_t*)y, (bf16_t*)h_out, (bf16_t*)xh_out,\
                           rstd_out, (int)rows, eps, drop_p, (const uint64_t*)seed, site_hidden, site_res,              \
                           g_ffn_dbg_host, rs_probe);                                                                 \
    } while (0)
        if (train) DSVG_FFN_FWD_RS(true);
        else DSVG_FFN_FWD_RS(false);
#undef DSVG_FFN_FWD_RS
        DSVG_LAUNCH_CHECK("ffn_fwd (role-specialised)");
        return 0;
    }
    if (stages != 3 && stages != 4) { dsvg_set_error("ffn_fwd: stages must be 0 (default), 2 / 7 (half-size workgroups), 3, 4, 5 or 6"); return -1; }
    if (pk) {
        if (train) DSVG_FFN_FWD(4, true, true);
        else DSVG_FFN_FWD(4, false, true);
    } else if (train && stages == 3) DSVG_FFN_FWD(3, true, false);
    else if (train) DSVG_FFN_FWD(4, true, false);
    else if (stages == 3) DSVG_FFN_FWD(3, false, false);
    else DSVG_FFN_FWD(4, false, false);
#undef DSVG_FFN_FWD
    DSVG_LAUNCH_CHECK("ffn_fwd");
    return 0;
}

/* development probe: buf = device buffer of (workgroups * 8 * 4) uint64 or NULL (off); see ffn_stamp */
extern "C" int dsvg_ffn_debug_clock(void* buf) {
    g_ffn_dbg_host = (unsigned long long*)buf;
    return 0;
}

extern "C" int dsvg_ffn_bwd(const void* x, const void* dy, const void* packed_bwd_layer, const float* b1_folded, void* h,
                            void* dpre, void* xh, void* dym, void* dx, int64_t rows, float eps, float drop_p,
                            uint32_t site_hidden, uint32_t site_res, const void* seed, void* stream) {
    DSVG_CHECK_ARG(x && dy && packed_bwd_layer && b1_folded && h && dpre && xh && dx, "ffn_bwd: null pointer");
    DSVG_CHECK_ARG(rows > 0 && rows < (1ll << 31) - TOK_PER_WG, "ffn_bwd: bad row count");
    DSVG_CHECK_ARG(!(drop_p > 0.f) || (seed && dym), "ffn_bwd: dropout needs a seed and the dym buffer");
    DSVG_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)h | (uintptr_t)dpre | (uintptr_t)xh | (uintptr_t)dym |
                     (uintptr_t)dx | (uintptr_t)packed_bwd_layer) & 15) == 0, "ffn_bwd: operands must be 16-byte aligned");
    const int nb = (int)((rows + TOK_PER_WG - 1) / TOK_PER_WG);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds1 = 3 * FWD_CHUNK + 2048, lds2 = 4 * 16 * FRAG;
    DSVG_ENSURE_LDS(ffn_bwd_hidden_kernel, lds1);
    hipLaunchKernelGGL(ffn_bwd_hidden_kernel, dim3(nb), dim3(512), lds1, st, (const bf16_t*)x, (const bf16_t*)dy,
                       (const bf16_t*)packed_bwd_layer, b1_folded, (bf16_t*)h, (bf16_t*)dpre, (bf16_t*)xh, (bf16_t*)dym,
                       (int)rows, eps, drop_p, (const uint64_t*)seed, site_hidden, site_res);
    DSVG_LAUNCH_CHECK("ffn_bwd (hidden)");
    hipLaunchKernelGGL(ffn_bwd_dx_kernel<false>, dim3(nb), dim3(512), lds2, st, (const bf16_t*)dpre, (const bf16_t*)x,
                       (const bf16_t*)dy, (const bf16_t*)packed_bwd_layer, (bf16_t*)dx, (int)rows, eps, (bf16_t*)nullptr, 0.f,
                       (const uint64_t*)nullptr, 0u, 0, (unsigned long long*)nullptr);
    DSVG_LAUNCH_CHECK("ffn_bwd (dx)");
    return 0;
}

extern "C" int dsvg_ffn_bwd_dx(const void* dpre, const void* x, const void* dy, const void* packed_bwd_layer, void* dx,
                               int64_t rows, float eps, void* dx_masked, float drop_p, uint32_t drop_site, const void* seed,
                               void* stream) {
    DSVG_CHECK_ARG(dpre && x && dy && packed_bwd_layer && dx, "ffn_bwd_dx: null pointer");
    DSVG_CHECK_ARG(!dx_masked || !(drop_p > 0.f) || seed, "ffn_bwd_dx: the masked output needs a seed");
    DSVG_CHECK_ARG(((uintptr_t)dx_masked & 15) == 0, "ffn_bwd_dx: operands must be 16-byte aligned");
    DSVG_CHECK_ARG(rows > 0 && rows < (1ll << 31) - TOK_PER_WG, "ffn_bwd_dx: bad row count");
    DSVG_CHECK_ARG((((uintptr_t)dpre | (uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)packed_bwd_layer) & 15) == 0,
                   "ffn_bwd_dx: operands must be 16-byte aligned");
    const int nb = (int)((rows + TOK_PER_WG - 1) / TOK_PER_WG);
    const size_t lds2 = 4 * 16 * FRAG;
    static const int w_warm = getenv("DSVG_W_WARM") ? atoi(getenv("DSVG_W_WARM")) : 1;      // A/B knob
    unsigned long long* probe = ((uintptr_t)g_ffn_dbg_host & 2) ? g_ffn_dbg_host : nullptr;
    if (probe)
        hipLaunchKernelGGL(ffn_bwd_dx_kernel<true>, dim3(nb), dim3(512), lds2, (hipStream_t)stream, (const bf16_t*)dpre,
                       (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)packed_bwd_layer, (bf16_t*)dx, (int)rows, eps,
                       (bf16_t*)dx_masked, drop_p, (const uint64_t*)seed, drop_site, w_warm,
                       probe);
    else
        hipLaunchKernelGGL(ffn_bwd_dx_kernel<false>, dim3(nb), dim3(512), lds2, (hipStream_t)stream, (const bf16_t*)dpre,
                       (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)packed_bwd_layer, (bf16_t*)dx, (int)rows, eps,
                       (bf16_t*)dx_masked, drop_p, (const uint64_t*)seed, drop_site, w_warm,
                       (unsigned long long*)nullptr);
    DSVG_LAUNCH_CHECK("ffn_bwd_dx");
    return 0;
}

extern "C" int dsvg_ffn_wgrad_finish(const float* g1p, const float* db1p, const float* g2p, const float* w1,
                                     const float* gamma, const float* beta, float* dw1, float* db1, float* dw2,
                                     float* dgamma, float* dbeta, void* stream) {
    DSVG_CHECK_ARG(g1p && db1p && g2p && w1 && gamma && beta && dw1 && db1 && dw2 && dgamma && dbeta,
                   "ffn_wgrad_finish: null pointer");
    hipLaunchKernelGGL(ffn_wgrad_finish_kernel, dim3(FD / 4), dim3(256), 0, (hipStream_t)stream, g1p, db1p, g2p, w1, gamma,
                       beta, dw1, db1, dw2, dgamma, dbeta);
    DSVG_LAUNCH_CHECK("ffn_wgrad_finish");
    return 0;
}

extern "C" int dsvg_ffn_wgrad_finish_many(const void* const* ptrs, int32_t n_layers, void* stream) {
    DSVG_CHECK_ARG(ptrs && n_layers > 0, "ffn_wgrad_finish_many: bad arguments");
    for (int base = 0; base < n_layers; base += FINISH_MAX) {
        const int n = n_layers - base < FINISH_MAX ? n_layers - base : FINISH_MAX;
        FinishTable t{};
        for (int i = 0; i < n; ++i) {
            const void* const* p = ptrs + (size_t)(base + i) * 11;
            for (int k = 0; k < 11; ++k) DSVG_CHECK_ARG(p[k], "ffn_wgrad_finish_many: null pointer");
            for (int k = 0; k < 6; ++k) t.in[i][k] = (const float*)p[k];
            for (int k = 0; k < 5; ++k) t.out[i][k] = (float*)const_cast<void*>(p[6 + k]);
        }
        hipLaunchKernelGGL(ffn_wgrad_finish_many_kernel, dim3(n * (FD / 4)), dim3(256), 0, (hipStream_t)stream, t);
        DSVG_LAUNCH_CHECK("ffn_wgrad_finish_many");
    }
    return 0;
}
