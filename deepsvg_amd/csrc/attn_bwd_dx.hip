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
#include "../../include/dsvg.h"

namespace {

constexpr int AD = 256;                 // d_model
constexpr int AK = 768;                 // q | k | v
constexpr int KSTEP = 32;
constexpr int NSTEP = AK / KSTEP;       // 24
constexpr int WSTAGE = 16 * FRAG;       // Win^T fragments of one K step
constexpr int WSLOTS = 3;
constexpr int XWAVES = 4;
constexpr int ROWS_WG = 32 * XWAVES;    // 128
constexpr int QSTAGE = ROWS_WG * KSTEP * 2;     // dqkv image of one K step: 8 KiB
constexpr int QSLOTS = 4;
constexpr int LDS_BYTES = WSLOTS * WSTAGE + QSLOTS * QSTAGE;       // 80 KiB
static_assert(LDS_BYTES == 81920, "two workgroups per CU");
static_assert(dsvg_pack::ATTN_BWD_FRAGS - dsvg_pack::ATTN_BWD_WO_FRAGS == NSTEP * 16, "pack_images.h restates the image");

// two 1 KiB LDS-DMA pieces with independent per-lane source addresses into consecutive KiB of LDS
__device__ __forceinline__ void dma1x2(const void* a, const void* b, uint32_t lds) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(a), "v"(b), "s"(lds) : "memory", "scc");
}

// the wait in front of a step's barrier: this wave's DMA pieces of the step have landed (counted vmcnt) AND every LDS read it has
// issued has returned (lgkmcnt(0)).  The second half is not optional: hipcc schedules the last MFMAs of a step behind the next
// barrier, their fragment reads are ISSUED in front of it but may still sit in the LDS queue (two workgroups per CU: ~100
// reads ahead of them) when another wave, released by the barrier, refills that ring slot - measured: a few 32-row waves per
// 63,488-row launch with the last two fragments of a step replaced by the weights of three steps later.
template <int N>
__device__ __forceinline__ void wait_step() {
    if (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else static_assert(N == 6 || N == 8, "add the count");
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dx_kernel(const bf16_t* __restrict__ dqkv, const bf16_t* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const bf16_t* __restrict__ res,
                                                             const bf16_t* __restrict__ wimg, bf16_t* __restrict__ dx,
                                                             float* __restrict__ part, int M, bf16_t* __restrict__ dxm,
                                                             float drop_p, const uint64_t* __restrict__ seed, uint32_t site_m) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    const int row0 = blockIdx.x * ROWS_WG;

    // ---- DMA sources ---------------------------------------------------------------------------------------------------
    // Win^T: this wave moves pieces 4 wave .. 4 wave + 3 of every 16-piece step
    const char* wsrc = reinterpret_cast<const char*>(wimg) + wave * 4096 + lane * 16;
    const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
    auto issue_w = [&](int s) { dma4(wsrc + (size_t)s * WSTAGE, wdst + (uint32_t)(s % WSLOTS) * WSTAGE); };
    // dqkv: pieces 2 wave, 2 wave + 1 of the 8-piece image = the wave's own 32 rows; lane l fills slot (l & 3) of row
    // (l >> 2) of its piece with source piece (l & 3) ^ ((row >> 2) & 3)
    const int qsw = (lane & 3) ^ ((lane >> 4) & 3);
    const int qra = min(row0 + 32 * wave + (lane >> 2), M - 1), qrb = min(row0 + 32 * wave + 16 + (lane >> 2), M - 1);
    const char* qa = reinterpret_cast<const char*>(dqkv) + (size_t)qra * (AK * 2) + qsw * 16;
    const char* qb = reinterpret_cast<const char*>(dqkv) + (size_t)qrb * (AK * 2) + qsw * 16;
    const uint32_t qdst = __builtin_amdgcn_readfirstlane(lds0 + WSLOTS * WSTAGE + wave * 2048);
    auto issue_q = [&](int s) { dma1x2(qa + s * (KSTEP * 2), qb + s * (KSTEP * 2), qdst + (uint32_t)(s % QSLOTS) * QSTAGE); };

    const int m = row0 + 32 * wave + tok;
    const int my_row = min(m, M - 1);
    const float mu = mean[my_row], rs = rstd[my_row];

    // issue order (per wave, loads return in order): q(0) | w(0) q(1) | w(1) q(2) | then per step s: w(s + 2) q(s + 3)
    issue_q(0);
    issue_w(0); issue_q(1);
    issue_w(1); issue_q(2);

    floatx16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // B operand: row (32 wave + tok) of the step's image, pieces 2 ks2 + half, swizzled by (row >> 2) & 3 = (tok >> 2) & 3
    const int bsw = (tok >> 2) & 3;
    const char* qrow = smem + WSLOTS * WSTAGE + (32 * wave + tok) * (KSTEP * 2);
    const int boff0 = ((0 + half) ^ bsw) * 16, boff1 = ((2 + half) ^ bsw) * 16;
    const char* wl = smem + lane * 16;
    const char* xrow = reinterpret_cast<const char*>(x) + (size_t)my_row * (AD * 2) + half * 32;
    const char* rrow = reinterpret_cast<const char*>(res) + (size_t)my_row * (AD * 2) + half * 32;
    // the lane's 16 columns of tile t of a row: two 16-byte pieces
    auto ld_tile = [&](const char* row, int t, uint4 (&d)[2]) {
        d[0] = *reinterpret_cast<const uint4*>(row + 64 * t);
        d[1] = *reinterpret_cast<const uint4*>(row + 64 * t + 16);
    };
    constexpr int AHEAD = 3;            // row tiles requested ahead of their use in the epilogue passes
    uint4 xa[8][2];

#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        // loads this wave may leave in flight: everything issued behind w(s) - q(s + 1), w(s + 1), q(s + 2) (2 + 4 + 2), in the
        // tail what is left of them, at the last step the first 6 row loads of the epilogue
        if (s <= NSTEP - 3) wait_step<8>();
        else wait_step<6>();
        __builtin_amdgcn_s_barrier();           // step s landed for everybody; the slots of step s - 1 are free again
        if (s + 2 < NSTEP) issue_w(s + 2);
        if (s + 3 < NSTEP) issue_q(s + 3);
        if (s == NSTEP - 2) {
            // the epilogue's first x pieces, two steps early; gamma into the dq slot of step 20 (dead: everybody is past step 21)
#pragma unroll
            for (int t = 0; t < AHEAD; ++t) ld_tile(xrow, t, xa[t]);
            if (wave == 0) {
                *reinterpret_cast<float4*>(smem + WSLOTS * WSTAGE + lane * 16) = *reinterpret_cast<const float4*>(gamma + lane * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        const char* wsl = wl + (s % WSLOTS) * WSTAGE;
        const char* qsl = qrow + (s % QSLOTS) * QSTAGE;
        Frag8 b0, b1;
        b0.u = *reinterpret_cast<const uint4*>(qsl + boff0);
        b1.u = *reinterpret_cast<const uint4*>(qsl + boff1);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            Frag8 a;
            a.u = *reinterpret_cast<const uint4*>(wsl + n * FRAG);
            acc[n >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, (n & 1) ? b1.v : b0.v, acc[n >> 1], 0, 0, 0);
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------------------------------
    // LDS, all of it dead ring by now (the slots of step 23 - W slot 2, dq slot 3 - are not touched):
    //   [0, 32 KiB)        per wave 8 KiB: one 32 x 32 fp32 block of dxn1 * xh and one of dxn1 (W slots 0, 1)
    //   [48 KiB, 49 KiB)   gamma;   [49 KiB, 57 KiB)  the waves' column sums [4][2][256] (dq slots 0, 1)
    const float* gam = reinterpret_cast<const float*>(smem + WSLOTS * WSTAGE);
    float* comb = reinterpret_cast<float*>(smem + WSLOTS * WSTAGE + 1024);
    char* blk = smem + wave * 8192;
    const bool live = m < M;
    float c1 = 0.f, c2 = 0.f;
    float cs[8];
    const int rq = lane >> 5, rc = lane & 31;       // column-sum role: block rq (0: dxn1 * xh, 1: dxn1), column rc of the tile
    const char* csrc = blk + rq * 4096 + (rc & 3) * 4;
    // ---- pass 1, tile by tile (one tile's temporaries and LDS traffic at a time: the accumulators fill half the registers) -----
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        asm volatile("" ::: "memory");
        if (t + AHEAD < 8) ld_tile(xrow, t + AHEAD, xa[t + AHEAD]);
        uint32_t xc[4][4];
        tile_to_cols16(acc[t], xc);                 // xc[q][e] = column 4 q + e of the lane's 16
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[8];
            unpack8(xa[t][q >> 1], v);
            const float4 g4 = *reinterpret_cast<const float4*>(gam + 32 * t + 16 * half + 4 * q);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
            float p[4], u4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = live ? __uint_as_float(xc[q][e]) : 0.f;
                const float xh = (v[4 * (q & 1) + e] - mu) * rs;
                const float g = u * gg[e];
                c1 += g;
                c2 += g * xh;
                p[e] = u * xh;
                u4[e] = u;
                acc[t][4 * q + e] = g;
            }
            // 4-float group 4 half + q of row tok, group slot swizzled by tok & 7 (conflict-free 16-byte stores and column reads)
            char* dst = blk + tok * 128 + (((4 * half + q) ^ (tok & 7)) * 16);
            *reinterpret_cast<float4*>(dst) = make_float4(p[0], p[1], p[2], p[3]);
            *reinterpret_cast<float4*>(dst + 4096) = make_float4(u4[0], u4[1], u4[2], u4[3]);
        }
        __builtin_amdgcn_wave_barrier();
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) sum += *reinterpret_cast<const float*>(csrc + r * 128 + (((rc >> 2) ^ (r & 7)) * 16));
        cs[t] = sum;
        __builtin_amdgcn_wave_barrier();
    }
    asm volatile("" ::: "memory");
    // the rows again for pass 2 (x from L2 this time) and the residual rows
    uint4 xb[8][2], rb[8][2];
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) { ld_tile(xrow, t, xb[t]); ld_tile(rrow, t, rb[t]); }
    c1 += __shfl_xor(c1, 32, 64);
    c2 += __shfl_xor(c2, 32, 64);
    c1 *= (1.f / AD);
    c2 *= (1.f / AD);
    // the waves' column sums, added in a fixed order while those rows are in flight
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
    // ---- pass 2: dx = res + rstd * (g - c1 - xh * c2), tile by tile ------------------------------------------------------------
    char* orow = reinterpret_cast<char*>(dx) + (size_t)my_row * (AD * 2) + half * 32;
    char* mrow = dxm ? reinterpret_cast<char*>(dxm) + (size_t)my_row * (AD * 2) + half * 32 : nullptr;
    const DropCtx dcm = drop_make(mrow ? drop_p : 0.f, seed, site_m);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        asm volatile("" ::: "memory");
        if (t + AHEAD < 8) { ld_tile(xrow, t + AHEAD, xb[t + AHEAD]); ld_tile(rrow, t + AHEAD, rb[t + AHEAD]); }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float v[8], r8[8];
            unpack8(xb[t][cb], v);
            unpack8(rb[t][cb], r8);
#pragma unroll
            for (int e = 0; e < 8; ++e) r8[e] += rs * (acc[t][8 * cb + e] - c1 - (v[e] - mu) * rs * c2);
            const uint4 pk = pack8(r8);
            if (live) *reinterpret_cast<uint4*>(orow + 64 * t + 16 * cb) = pk;
            if (mrow) {     // dx as stored, with the mask of the consumer's dropout site on it (what dsvg_drop_apply would make)
                float w[8], mm[8];
                unpack8(pk, w);
                drop_mult8(dcm, (uint64_t)m * AD + 32 * t + 16 * half + 8 * cb, mm);
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] *= mm[e];
                if (live) *reinterpret_cast<uint4*>(mrow + 64 * t + 16 * cb) = pack8(w);
            }
        }
    }
}

}  // namespace

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
                       (const uint64_t*)seed, drop_site);
    DSVG_LAUNCH_CHECK("attn_bwd_dx");
    // workspace rows are [dgamma(256) | dbeta(256)]; in the flat gradient buffer norm.bias follows norm.weight: the usual
    // case is ONE deterministic reduction of 512 columns (queued while a deferral scope is open), otherwise two strided ones
    if (dbeta == dgamma + AD) return dsvg_reduce_partials_strided(workspace, nb, 2 * AD, 2 * AD, dgamma, accumulate, st);
    int rc = dsvg_reduce_partials_strided(workspace, nb, 2 * AD, AD, dgamma, accumulate, st);
    if (rc) return rc;
    return dsvg_reduce_partials_strided(workspace + AD, nb, 2 * AD, AD, dbeta, accumulate, st);
}
