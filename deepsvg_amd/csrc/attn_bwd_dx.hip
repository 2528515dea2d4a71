// Input gradient of the attention sub-block of the pre-LN transformer layers, d_model = 256, bf16 storage, fp32 accumulation:
//     dx = dx1 + LayerNorm'( dqkv . Win )            (+ the norm's dgamma / dbeta partial sums)
// i.e. the backward of `x1 = x + drop(out_proj(MHA(in_proj(norm1(x)))))` from the packed q|k|v gradient on
// (deepsvg/model/layers/improved_transformer.py:43-45,127-129 under autograd, deepsvg/train.py:98).  One launch replaces
// the input-gradient GEMM dxn1 = dqkv . Win (its [T, 256] result written and read once) and dsvg_layernorm_bwd behind it:
// the mirror image of ffn_bwd_dx_kernel for the other half of the layer.
//
// Structure (gfx950, 64-lane waves, v_mfma_f32_32x32x16_bf16):
//   * a 256-thread workgroup owns 128 token rows, each of its 4 waves 32 of them with ALL 256 output columns (8 transposed
//     accumulator tiles = 128 registers): the LayerNorm backward needs whole rows, so its statistics stay inside a lane
//     pair.  Two workgroups per CU (80 KiB of LDS each).
//   * K = 768 is walked in 24 steps of 32.  Both operands go HBM / L2 -> LDS by LDS-DMA, no staging registers:
//       - Win^T as ready-made A fragments (dsvg_attn_pack_bwd, fragments 128 .. 511 of a layer's image: 16 KiB per step,
//         contiguous), ring of 3 steps, issued 2 steps ahead;
//       - the dqkv rows as a [128 rows][32 k] image (64 B per row and step), ring of 4 steps, issued 3 steps ahead (these
//         bytes come from HBM, the weights from L2).  The DMA writes lane-linear, so the bank swizzle is applied to the
//         SOURCE address: 16-byte piece c of row r lives at slot c ^ ((r >> 2) & 3) - conflict-free ds_read_b128 B-operand
//         reads (lane = token row, 8 consecutive k).
//     One s_barrier and one counted s_waitcnt vmcnt per step (the DMA is issued from inline asm, see fused_common.h).
//   * the x rows of the epilogue are requested two steps before the loop ends; gamma is staged into a dead ring slot.
//   * epilogue, per lane one token row x 16 consecutive columns per tile: g = gamma * dxn1, the row means of g and g * xh
//     from registers + one lane-pair exchange, dx = res + rstd * (g - mean(g) - xh * mean(g * xh)) with the forward pass's
//     stored mean / rstd; optional second output dx with a dropout mask replayed on it (what dsvg_layernorm_bwd_masked
//     hands to the layer below).  dgamma = sum_t dxn1 * xh and dbeta = sum_t dxn1: per wave and tile a 32 x 32 fp32 block
//     goes through a swizzled LDS scratch area (the dead ring) and comes back as column sums, the four waves are added in
//     a fixed order -> one partial row [2][256] per workgroup, reduced by the library's deterministic partial reduction
//     (queued under dsvg_defer_scope like dsvg_layernorm_bwd's).
#include "fused_common.h"
#include "pack_images.h"
#include <utility>
#include "../../include/dsvg.h"

namespace {

constexpr int AD = 256;                 // d_model
constexpr int AK = 768;                 // q | k | v
constexpr int KSTEP = 32;
constexpr int NSTEP = AK / KSTEP;       // 24
constexpr int WSTAGE = 16 * FRAG;       // Win^T fragments of one K step
constexpr int WSLOTS = 4, WA = 2;       // ring slots / steps the weight DMA runs ahead (a slot is refilled two barriers after its last read)
constexpr int QA = 4;                   // steps the dqkv row loads run ahead (registers: 8 per step and lane)
constexpr int XWAVES = 4;
constexpr int ROWS_WG = 32 * XWAVES;    // 128
constexpr int QSTAGE = 32 * KSTEP * 2;  // one wave's dqkv image of one K step: 2 KiB, two private slots per wave
constexpr int LDS_BYTES = WSLOTS * WSTAGE + XWAVES * 2 * QSTAGE;       // 64 + 16 KiB
constexpr int AHEAD = 3, AHEAD2 = 2;    // row tiles requested ahead of their use in the epilogue's pass 1 (x) / pass 2 (x and res)
static_assert(LDS_BYTES == 81920, "two workgroups per CU");
static_assert(QA > WA + 1, "the dqkv rows of step s + 1 must be older than the weights of step s (one counted wait covers both)");
static_assert(NSTEP % 4 == 0 && WSLOTS == 4 && QA == 4, "the K loop is a real loop of period 4 (ring indices are compile-time)");
static_assert(dsvg_pack::ATTN_BWD_FRAGS - dsvg_pack::ATTN_BWD_WO_FRAGS == NSTEP * 16, "pack_images.h restates the image");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr bool W_PREFETCH = true;

// VMEM operations a wave issues per step behind its barrier: weights(s + WA) [4 DMA pieces], rows(s + QA) [2 loads] - ALWAYS,
// past the end of K the last step's addresses again (the loop body is one piece of code for every step: 44 KB of straight-line
// code ran 1.5 x slower inside the training step than in a loop of its own launches - cold instruction cache).  Loads return
// in order, so `s_waitcnt vmcnt(STEADY)` in front of barrier s = "the weights of step s (and everything older: the dqkv rows of
// step s + 1) have landed": behind weights(s) come rows(s - WA + QA) and the WA - 1 steps in between.  The prologue issues in
// the same pattern (virtual steps -WA .. -1), so the count holds from step 0 on.
constexpr int STEADY = 2 + 6 * (WA - 1);

// the wait in front of a step's barrier: this wave's DMA pieces of the step have landed (counted vmcnt) AND every LDS read it has
// issued has returned (lgkmcnt(0)).  The second half is not optional: hipcc schedules the last MFMAs of a step behind the next
// barrier, their fragment reads are ISSUED in front of it but may still sit in the LDS queue (two workgroups per CU: ~100
// reads ahead of them) when another wave, released by the barrier, refills that ring slot - measured on the first version of
// this kernel: a few 32-row waves per 63,488-row launch with the last two fragments of a step replaced by later weights.
template <int N>
__device__ __forceinline__ void wait_step() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
// a row load the compiler does not track (its own vmcnt model does not see the asm DMA between these loads and would drain the
// queue to the last two steps): the destination is valid only behind one of the counted waits above.
// address = wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset
__device__ __forceinline__ void ld_async(u32x4& d, const void* base_uniform, uint32_t lane_off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(lane_off), "s"(base_uniform) : "memory");
}

// development probe (dsvg_attn_bwd_dx_debug_clock): every wave stores the chip-wide 100 MHz counter at 6 points of its life
__device__ __forceinline__ void abd_stamp(unsigned long long* d, int slot) {
    if (d && (threadIdx.x & 63) == 0) {
        typedef unsigned long long __attribute__((address_space(1))) * gptr_t;     // (a global_store, not a flat one: fused_common)
        gptr_t b = (gptr_t)(reinterpret_cast<uintptr_t>(d));
        b[((size_t)blockIdx.x * XWAVES + (threadIdx.x >> 6)) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int... I, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dx_kernel(const bf16_t* __restrict__ dqkv, const bf16_t* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const bf16_t* __restrict__ res,
                                                             const bf16_t* __restrict__ wimg, bf16_t* __restrict__ dx,
                                                             float* __restrict__ part, int M, bf16_t* __restrict__ dxm,
                                                             float drop_p, const uint64_t* __restrict__ seed, uint32_t site_m,
                                                             unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    const int row0 = blockIdx.x * ROWS_WG;

    // ---- sources ---------------------------------------------------------------------------------------------------------
    // Win^T: this wave moves pieces 4 wave .. 4 wave + 3 of every 16-piece step
    const uint32_t wlane = (uint32_t)(wave * 4096 + lane * 16);
    const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
    auto issue_w = [&](int s, int slot) {       // s: wave-uniform, slot: compile-time after unrolling
        const char* base = reinterpret_cast<const char*>(wimg) + (size_t)min(s, NSTEP - 1) * WSTAGE;
        dma4s(base, wlane, wdst + (uint32_t)slot * WSTAGE);
    };
    // dqkv: the wave's own 32 rows, 64 bytes per row and step as two coalesced 1 KiB loads (lane l: row (l >> 2) [+ 16], 16-byte
    // piece (l & 3) ^ ((row >> 2) & 3) - the swizzle of the LDS image, applied on the source side so that the lane's store is
    // lane-linear), QA steps ahead in registers, then one step ahead into the wave's private two-slot LDS image
    const int qsw = (lane & 3) ^ ((lane >> 4) & 3);
    const int qra = min(row0 + 32 * wave + (lane >> 2), M - 1), qrb = min(row0 + 32 * wave + 16 + (lane >> 2), M - 1);
    const uint32_t qoa = (uint32_t)qra * (AK * 2) + qsw * 16, qob = (uint32_t)qrb * (AK * 2) + qsw * 16;
    u32x4 qr[QA][2];
    auto issue_q = [&](int s, int slot) {
        const char* base = reinterpret_cast<const char*>(dqkv) + min(s, NSTEP - 1) * (KSTEP * 2);
        ld_async(qr[slot][0], base, qoa);
        ld_async(qr[slot][1], base, qob);
    };
    char* qimg = smem + WSLOTS * WSTAGE + wave * (2 * QSTAGE);
    auto stage_q = [&](int slot) {      // registers of ring entry `slot` -> image slot `slot & 1` (rows 0 .. 15 | rows 16 .. 31), lane-linear
        char* d = qimg + (slot & 1) * QSTAGE + lane * 16;
        *reinterpret_cast<u32x4*>(d) = qr[slot][0];
        *reinterpret_cast<u32x4*>(d + 1024) = qr[slot][1];
    };

    abd_stamp(dbg, 0);
    const int m = row0 + 32 * wave + tok;
    const int my_row = min(m, M - 1);
    const float mu = mean[my_row], rs = rstd[my_row];

    u32x4 gq;           // gamma (wave 0 stages it into LDS behind the loop): the oldest load, covered by every wait
    ld_async(gq, gamma, (uint32_t)lane * 16);
    // The weight image (384 KiB) is cold in this XCD's L2 when the launch starts (inside a training step every layer's image is
    // touched once per step), and the workgroups walk it in lockstep: every K step would begin with one HBM miss that all of them
    // wait for.  So the waves of an XCD (workgroup b runs on XCD b % 8) request the whole image up front, 2 KiB each.
    u32x4 sink0, sink1;
    if (W_PREFETCH) {
        const uint32_t widx = (uint32_t)((blockIdx.x >> 3) * XWAVES + wave) * 2u;
        ld_async(sink0, wimg, ((widx % 384u) << 10) + (uint32_t)lane * 16);
        ld_async(sink1, wimg, (((widx + 1u) % 384u) << 10) + (uint32_t)lane * 16);
    }
    // prologue in the loop's own issue pattern: rows(0), rows(1) | weights(0), rows(2) | weights(1), rows(3)
    issue_q(0, 0); issue_q(1, 1);
    issue_w(0, 0); issue_q(2, 2);
    issue_w(1, 1); issue_q(3, 3);

    floatx16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // B operand: row tok of the wave's image, pieces 2 ks2 + half, swizzled by (row >> 2) & 3
    const int bsw = (tok >> 2) & 3;
    const char* qrow = qimg + tok * (KSTEP * 2);
    const int boff0 = ((0 + half) ^ bsw) * 16, boff1 = ((2 + half) ^ bsw) * 16;
    const char* wl = smem + lane * 16;
    const char* xrow = reinterpret_cast<const char*>(x) + (size_t)my_row * (AD * 2) + half * 32;
    const char* rrow = reinterpret_cast<const char*>(res) + (size_t)my_row * (AD * 2) + half * 32;
    // the lane's 16 columns of tile t of a row: two 16-byte pieces
    auto ld_tile = [&](const char* row, int t, uint4 (&d)[2]) {
        d[0] = *reinterpret_cast<const uint4*>(row + 64 * t);
        d[1] = *reinterpret_cast<const uint4*>(row + 64 * t + 16);
    };
    uint4 xa[8][2];

    // rows of step 0: the oldest row loads of the prologue (behind them: rows(1), weights(0), rows(2), weights(1), rows(3))
    wait_step<2 + 4 + 2 + 4 + 2>();
    // (the destination registers of an untracked load stay reserved until it has landed: a register hipcc considers dead would
    // be handed to another value and overwritten when the data arrives)
    if (W_PREFETCH) asm volatile("" ::"v"(sink0), "v"(sink1));
    stage_q(0);
    abd_stamp(dbg, 1);

#pragma unroll 1
    for (int it = 0; it < NSTEP / 4; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = 4 * it + u;
            wait_step<STEADY>();
            __builtin_amdgcn_s_barrier();       // weights of step s landed for everybody; the slot of step s - 2 is free again
            issue_w(s + WA, (u + WA) & 3);
            issue_q(s + QA, u);                 // (ring entry u held the rows of step s: staged one step ago)
            stage_q((u + 1) & 3);               // rows of step s + 1 (landed: older than the weights of step s)
            const char* wsl = wl + u * WSTAGE;
            const char* qsl = qrow + (u & 1) * QSTAGE;
            Frag8 b0, b1;
            b0.u = *reinterpret_cast<const uint4*>(qsl + boff0);
            b1.u = *reinterpret_cast<const uint4*>(qsl + boff1);
            // A fragments: a ring of 8, refilled right behind the MFMA that has consumed the entry
            Frag8 a[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) a[n].u = *reinterpret_cast<const uint4*>(wsl + n * FRAG);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                acc[n >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n & 7].v, (n & 1) ? b1.v : b0.v, acc[n >> 1], 0, 0, 0);
                if (n + 8 < 16) a[n & 7].u = *reinterpret_cast<const uint4*>(wsl + (n + 8) * FRAG);
            }
        }
    }
    // the epilogue's first x pieces; then the surplus loads / DMA of the last steps drain before the ring is reused
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) ld_tile(xrow, t, xa[t]);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * AHEAD) : "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == 0) *reinterpret_cast<u32x4*>(smem + 1 * WSTAGE + lane * 16) = gq;
    abd_stamp(dbg, 2);
    __syncthreads();        // (every wave is done with the ring: the epilogue reuses all of it)

    // ---- epilogue --------------------------------------------------------------------------------------------------------
    // LDS, all of it dead ring by now:
    //   [0, 16 KiB) weight slot 0: the waves' column sums [4][2][256] (8 KiB);  [16 KiB, 17 KiB): gamma (staged at step NSTEP - 2)
    //   [32 KiB, 64 KiB) weight slots 2, 3: per wave 8 KiB, one 32 x 32 fp32 block of dxn1 * xh and one of dxn1
    const float* gam = reinterpret_cast<const float*>(smem + 1 * WSTAGE);
    float* comb = reinterpret_cast<float*>(smem);
    char* blk = smem + 2 * WSTAGE + wave * 8192;
    const bool live = m < M;
    float c1 = 0.f, c2 = 0.f;
    const int rq = lane >> 5, rc = lane & 31;       // column-sum role: block rq (0: dxn1 * xh, 1: dxn1), column rc of the tile
    const char* csrc = blk + rq * 4096 + (rc & 3) * 4;
    // ---- pass A: the accumulator tiles into row order (in place), the row statistics c1 = mean(g), c2 = mean(g * xh) -----------
    // (tile by tile; the statistics are pinned per tile: hipcc otherwise postpones both sums to the end of the pass and parks
    // their terms in scratch memory)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        asm volatile("" ::: "memory");
        if (t + AHEAD < 8) ld_tile(xrow, t + AHEAD, xa[t + AHEAD]);
        uint32_t xc[4][4];
        tile_to_cols16(acc[t], xc);                 // xc[q][e] = column 4 q + e of the lane's 16
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[8];
            unpack8(xa[t][q >> 1], v);
            const float4 g4 = *reinterpret_cast<const float4*>(gam + 32 * t + 16 * half + 4 * q);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
            float g[4], gx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = live ? __uint_as_float(xc[q][e]) : 0.f;
                acc[t][4 * q + e] = u;
                g[e] = u * gg[e];
                gx[e] = g[e] * ((v[4 * (q & 1) + e] - mu) * rs);
            }
            s1 += (g[0] + g[1]) + (g[2] + g[3]);
            s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        }
        c1 += s1;
        c2 += s2;
        asm volatile("" : "+v"(c1), "+v"(c2));
    }
    abd_stamp(dbg, 3);
    asm volatile("" ::: "memory");
    // the rows again for pass B (x from L2 this time, through a pointer the compiler cannot relate to pass A's: it would otherwise
    // keep the normalised values of pass A alive - in scratch memory) and the residual rows
    // (an opaque ZERO OFFSET, not an opaque pointer: a laundered pointer is a generic one - flat_load, whose waits cannot be counted)
    unsigned zoff = 0;
    asm volatile("" : "+v"(zoff));
    const char* xrow2 = xrow + zoff;
    uint4 xb[8][2], rb[8][2];
#pragma unroll
    for (int t = 0; t < AHEAD2; ++t) { ld_tile(xrow2, t, xb[t]); ld_tile(rrow, t, rb[t]); }
    c1 += __shfl_xor(c1, 32, 64);
    c2 += __shfl_xor(c2, 32, 64);
    c1 *= (1.f / AD);
    c2 *= (1.f / AD);
    // ---- pass B, tile by tile: dx = res + rstd * (g - c1 - xh * c2) -> stores; dxn1 * xh and dxn1 -> column sums -------------------
    char* orow = reinterpret_cast<char*>(dx) + (size_t)my_row * (AD * 2) + half * 32;
    char* mrow = dxm ? reinterpret_cast<char*>(dxm) + (size_t)my_row * (AD * 2) + half * 32 : nullptr;
    const DropCtx dcm = drop_make(mrow ? drop_p : 0.f, seed, site_m);
    float cs[8];
    auto passB = [&](auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            asm volatile("" ::: "memory");
            if (t + AHEAD2 < 8) { ld_tile(xrow2, t + AHEAD2, xb[t + AHEAD2]); ld_tile(rrow, t + AHEAD2, rb[t + AHEAD2]); }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                float v[8], r8[8];
                unpack8(xb[t][cb], v);
                unpack8(rb[t][cb], r8);
                const float4 ga = *reinterpret_cast<const float4*>(gam + 32 * t + 16 * half + 8 * cb);
                const float4 gb = *reinterpret_cast<const float4*>(gam + 32 * t + 16 * half + 8 * cb + 4);
                const float gg[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                float p[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float u = acc[t][8 * cb + e];
                    const float xh = (v[e] - mu) * rs;
                    p[e] = u * xh;
                    r8[e] += rs * (u * gg[e] - c1 - xh * c2);
                }
                // the lane's columns of both blocks: 4-float groups 4 half + 2 cb (+ 1) of row tok, group slot swizzled by tok & 7
                // (conflict-free 16-byte stores and column reads)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    char* dst = blk + tok * 128 + (((4 * half + 2 * cb + q2) ^ (tok & 7)) * 16);
                    *reinterpret_cast<float4*>(dst) = make_float4(p[4 * q2], p[4 * q2 + 1], p[4 * q2 + 2], p[4 * q2 + 3]);
                    *reinterpret_cast<float4*>(dst + 4096) = make_float4(acc[t][8 * cb + 4 * q2], acc[t][8 * cb + 4 * q2 + 1],
                                                                          acc[t][8 * cb + 4 * q2 + 2], acc[t][8 * cb + 4 * q2 + 3]);
                }
                const uint4 pk = pack8(r8);
                if (live) *reinterpret_cast<uint4*>(orow + 64 * t + 16 * cb) = pk;
                if (MASKED) {   // dx as stored, with the mask of the consumer's dropout site on it (what dsvg_drop_apply would make)
                    float w[8], mm[8];
                    unpack8(pk, w);
                    drop_mult8(dcm, (uint64_t)m * AD + 32 * t + 16 * half + 8 * cb, mm);
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] *= mm[e];
                    if (live) *reinterpret_cast<uint4*>(mrow + 64 * t + 16 * cb) = pack8(w);
                }
            }
            __builtin_amdgcn_wave_barrier();
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) sum += *reinterpret_cast<const float*>(csrc + r * 128 + (((rc >> 2) ^ (r & 7)) * 16));
            asm volatile("" : "+v"(sum));       // (due here, not at the end of the pass: see c1 / c2)
            cs[t] = sum;
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (mrow) passB(std::true_type{});
    else passB(std::false_type{});
    abd_stamp(dbg, 4);
    // the waves' column sums, added in a fixed order
#pragma unroll
    for (int t = 0; t < 8; ++t) comb[(wave * 2 + rq) * AD + 32 * t + rc] = cs[t];
    __syncthreads();
    {
        float* pg = part + (size_t)blockIdx.x * (2 * AD);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < XWAVES; ++w) sum += comb[(w * 2 + k) * AD + tid];
            pg[k * AD + tid] = sum;
        }
    }
    abd_stamp(dbg, 5);
}

}  // namespace

static unsigned long long* g_abd_dbg = nullptr;
/* development probe: buf = device buffer of (workgroups * 4 waves * 8) uint64 or NULL (off); see abd_stamp */
extern "C" int dsvg_attn_bwd_dx_debug_clock(void* buf) {
    g_abd_dbg = (unsigned long long*)buf;
    return 0;
}

extern "C" int64_t dsvg_attn_bwd_dx_workspace_bytes(int64_t rows) {
    return ((rows + ROWS_WG - 1) / ROWS_WG) * 2 * AD * (int64_t)sizeof(float);
}

extern "C" int dsvg_attn_bwd_dx(const void* dqkv, const void* x, const float* mean, const float* rstd, const float* gamma,
                                const void* res, const void* packed_bwd_layer, void* dx, float* dgamma, float* dbeta,
                                int32_t accumulate, int64_t rows, float* workspace, int64_t workspace_bytes, void* dx_masked,
                                float drop_p, uint32_t drop_site, const void* seed, void* stream) {
    DSVG_CHECK_ARG(dqkv && x && mean && rstd && gamma && res && packed_bwd_layer && dx && dgamma && dbeta,
                   "attn_bwd_dx: null pointer");
    DSVG_CHECK_ARG(rows > 0 && rows < (1ll << 31) - ROWS_WG, "attn_bwd_dx: bad row count");
    DSVG_CHECK_ARG((((uintptr_t)dqkv | (uintptr_t)x | (uintptr_t)res | (uintptr_t)dx | (uintptr_t)packed_bwd_layer |
                     (uintptr_t)gamma | (uintptr_t)dx_masked) & 15) == 0, "attn_bwd_dx: operands must be 16-byte aligned");
    DSVG_CHECK_ARG(!dx_masked || dx_masked != dx, "attn_bwd_dx: the masked output needs its own buffer");
    DSVG_CHECK_ARG(!dx_masked || !(drop_p > 0.f) || seed, "attn_bwd_dx: the masked output needs a seed");
    DSVG_CHECK_ARG(workspace && workspace_bytes >= dsvg_attn_bwd_dx_workspace_bytes(rows), "attn_bwd_dx: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)((rows + ROWS_WG - 1) / ROWS_WG);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        once = true;
    }
    // fragments 128 .. 511 of the layer's backward image = in_proj_weight^T (pack_images.h)
    const bf16_t* wimg = (const bf16_t*)packed_bwd_layer + (size_t)dsvg_pack::ATTN_BWD_WO_FRAGS * 512;
    hipLaunchKernelGGL(attn_bwd_dx_kernel, dim3(nb), dim3(256), LDS_BYTES, st, (const bf16_t*)dqkv, (const bf16_t*)x, mean, rstd,
                       gamma, (const bf16_t*)res, wimg, (bf16_t*)dx, workspace, (int)rows, (bf16_t*)dx_masked, drop_p,
                       (const uint64_t*)seed, drop_site, g_abd_dbg);
    DSVG_LAUNCH_CHECK("attn_bwd_dx");
    // workspace rows are [dgamma(256) | dbeta(256)]; in the flat gradient buffer norm.bias follows norm.weight: the usual
    // case is ONE deterministic reduction of 512 columns (queued while a deferral scope is open), otherwise two strided ones
    if (dbeta == dgamma + AD) return dsvg_reduce_partials_strided(workspace, nb, 2 * AD, 2 * AD, dgamma, accumulate, st);
    int rc = dsvg_reduce_partials_strided(workspace, nb, 2 * AD, AD, dgamma, accumulate, st);
    if (rc) return rc;
    return dsvg_reduce_partials_strided(workspace + AD, nb, 2 * AD, AD, dbeta, accumulate, st);
}
