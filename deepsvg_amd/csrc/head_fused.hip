// The argument head (args_fcn: Linear(d_model 256 -> n_args * args_dim = 11 x 257 logits), deepsvg/model/model.py:228-246) fused
// with what consumes its logits, so that the [tokens, 2827] logit tensor - the largest stream of the model - never reaches
// HBM:
//   MODE_ARGMAX   decoding: arg-max per (token, slot) - the temperature-0 limit of deepsvg/model/utils.py:75-80 - or, with a
//                 temperature > 0, the reference's categorical draw itself as a Gumbel arg-max (logit + T g, dsvg_gumbel)
//   MODE_LSE      SVGLoss's cross-entropy forward (deepsvg/model/loss.py:51-57): log-sum-exp per (token, slot) and the
//                 weighted sum / count of (lse - logit[target])
//   MODE_DLOGITS  its backward: the logit tile is recomputed and leaves as dlogits = w g (softmax - onehot) in bf16, the
//                 operand of the head's two gradient GEMMs
// One skeleton (the GEMM-1 half of ffn_fused.hip): a 512-thread workgroup owns 256 token rows, a wave keeps its 32 rows as
// the 16 MFMA B fragments of the K = 256 reduction in registers for the whole kernel; the head's weight rows stream by in
// chunks of 64 output columns - dsvg_head_pack stores them as ready-made A fragments, 32 KiB per chunk - through a 3-slot
// LDS ring filled by LDS-DMA two chunks ahead (counted vmcnt: loads return in order, so "at most the newest issue is in
// flight" holds whatever stores are pending).  D = W_chunk x X^T: a lane holds, for its token (lane & 31), 16 of a tile's 32
// columns - rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) - so every per-(token, slot) statistic is a chain over the lane's own
// registers plus ONE exchange with lane ^ 32 when a slot ends.  Slots are runs of C consecutive columns; a chunk holds at
// most one slot edge per tile (C >= 64 is required), handled by masking.
#include "fused_common.h"
#include "../../include/dsvg.h"

namespace {

constexpr int HK = 256;                 // d_model = the reduction
constexpr int HCH = 64;                 // output columns per chunk (two 32 x 32 tiles)
constexpr int HTOK = 256;               // token rows per workgroup
constexpr int HSLOT = 32 * FRAG;        // bytes of a chunk image: [tile 2][K step 16] fragments
constexpr int HNBUF = 3;
constexpr int H_MAX_COLS = 3072;        // bias staging
enum { MODE_ARGMAX = 0, MODE_LSE = 1, MODE_DLOGITS = 2 };

__host__ __device__ inline int hrowmap(int r, int h2) { return (r & 3) + 8 * (r >> 2) + 4 * h2; }

// img[chunk][tile t][K step ks][lane l][e] = W[64 chunk + 32 t + (l & 31)][16 ks + 8 (l >> 5) + e], zero rows past n_out
__global__ __launch_bounds__(256) void head_pack_kernel(const bf16_t* __restrict__ w, int n_out, int n_chunks,
                                                        bf16_t* __restrict__ img) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;       // one thread per 16-byte lane slot
    if (gid >= (long long)n_chunks * 32 * 64) return;
    const int l = (int)(gid & 63), f = (int)((gid >> 6) & 31), c = (int)(gid >> 11);
    const int col = HCH * c + 32 * (f >> 4) + (l & 31);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (col < n_out) v = *reinterpret_cast<const uint4*>(w + (size_t)col * HK + 16 * (f & 15) + 8 * (l >> 5));
    *reinterpret_cast<uint4*>(img + gid * 8) = v;
}

struct HeadArgs {
    const bf16_t* x;            // [rows, 256]
    const bf16_t* img;          // dsvg_head_pack
    const float* bias;          // [n_out]
    long long rows;
    int n_out, C, group;        // n_out = group * C columns in use
    // per (token, slot) inputs, indexed tok * group + slot with tok = tok_idx ? tok_idx[row] : row (negative: weight 0)
    const int32_t* target;
    const float* w;
    const int32_t* tok_idx;
    // MODE_ARGMAX
    int32_t* out_idx;           // [rows * group]
    float temperature;          // > 0: sample from softmax(logits / temperature) (Gumbel arg-max), 0: plain arg-max
    const uint64_t* seed;
    uint32_t site;
    // MODE_LSE
    float* lse;                 // [rows * group] (out; in for MODE_DLOGITS)
    float* part;                // [gridDim.x * 2]: (sum of w (lse - logit[target]), sum of w) per workgroup
    // MODE_DLOGITS
    const float* sum_count;     // [2]: the reduced pair (count = [1])
    const float* gscale;        // optional upstream gradient scalar
    float coef;
    bf16_t* dl;                 // [rows, ld_d]
    long long ld_d;
};

template <int MODE>
__global__ __launch_bounds__(512, 1) void head_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [3 x 32 KiB ring | bias fp32 | 8 x 2 floats]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, h2 = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem);
    float* sbias = reinterpret_cast<float*>(smem + HNBUF * HSLOT);
    float* sred = sbias + H_MAX_COLS;
    const int n_chunks = (a.n_out + HCH - 1) / HCH;
    // this wave moves fragments 4 wave .. 4 wave + 3 of every chunk
    const char* my_src = reinterpret_cast<const char*>(a.img) + wave * 4096 + lane * 16;
    const uint32_t my_dst = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
    auto issue = [&](int c) { dma4(my_src + (size_t)c * HSLOT, my_dst + (uint32_t)(c % HNBUF) * HSLOT); };
    issue(0);
    if (n_chunks > 1) issue(1);
    for (int i = tid; i < n_chunks * HCH; i += 512) sbias[i] = i < a.n_out ? a.bias[i] : 0.f;
    __syncthreads();            // (the chunk loop's barriers are bare s_barrier: publish the staged bias here)

    const long long row = (long long)blockIdx.x * HTOK + wave * 32 + tok;
    const long long my_row = row < a.rows ? row : a.rows - 1;
    const bool row_ok = row < a.rows;
    bf16x8 xf[16];
    {
        const char* xr = reinterpret_cast<const char*>(a.x) + (size_t)my_row * (HK * 2) + h2 * 16;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Frag8 f;
            f.u = *reinterpret_cast<const uint4*>(xr + 32 * ks);
            xf[ks] = f.v;
        }
    }
    long long src_tok = my_row;
    if (MODE != MODE_ARGMAX && a.tok_idx) src_tok = a.tok_idx[my_row];
    const bool tok_ok = row_ok && src_tok >= 0;
    const long long r0 = (tok_ok ? src_tok : 0) * a.group;        // index of (token, slot 0) in target / w
    const long long o0 = my_row * a.group;                         // ... in lse / out_idx
    const int C = a.C;
    float g = 0.f;
    if (MODE == MODE_DLOGITS) g = a.coef * (a.gscale ? *a.gscale : 1.f) / a.sum_count[1];
    const bool noisy = MODE == MODE_ARGMAX && a.temperature > 0.f;
    const DropCtx gctx = drop_make(noisy ? 0.5f : 0.f, a.seed, a.site);      // (only its mixed seed words are used)
    const uint32_t k4 = (uint32_t)(a.n_out + 3) >> 2;

    // ---- running state of the slot this wave is in ----------------------------------------------------------------------------
    int slot = 0, slot_lo = 0, slot_hi = C;                         // columns [slot_lo, slot_hi)
    float st_m = -INFINITY, st_s = 0.f, st_t = 0.f;                 // LSE: running max, sum of exp, logit of the target
    int st_i = 0;                                                   // ARGMAX: best column (st_m = its value)
    float acc_l = 0.f, acc_w = 0.f;                                 // LSE: this lane's share of the loss sum / count
    // per-(token, slot) scalars of the current slot and, prefetched, of the next one
    int tcol = -1, tcol_n = -1;
    float wr = 0.f, wr_n = 0.f, ls = 0.f, ls_n = 0.f;
    auto fetch = [&](int s, int& tc, float& wv, float& lv) {
        if (MODE == MODE_ARGMAX) return;
        const int sc = min(s, a.group - 1);
        const int t = a.target[r0 + sc];
        tc = sc * C + min(max(t, 0), C - 1);
        wv = tok_ok ? (a.w ? a.w[r0 + sc] : 1.f) : 0.f;
        if (MODE == MODE_DLOGITS) lv = a.lse[o0 + sc];
    };
    fetch(0, tcol, wr, ls);
    fetch(1, tcol_n, wr_n, ls_n);

    auto flush = [&]() {        // the slot is complete: combine the two lane halves of a token, write its result
        if (MODE == MODE_ARGMAX) {
            const float om = __shfl_xor(st_m, 32, 64);
            const int oi = __shfl_xor(st_i, 32, 64);
            if (om > st_m || (om == st_m && oi < st_i)) { st_m = om; st_i = oi; }
            if (h2 == 0 && row_ok) a.out_idx[o0 + slot] = st_i - slot_lo;
        } else if (MODE == MODE_LSE) {
            const float om = __shfl_xor(st_m, 32, 64), os = __shfl_xor(st_s, 32, 64), ot = __shfl_xor(st_t, 32, 64);
            const float M = fmaxf(st_m, om);
            const float S = st_s * __expf(st_m - M) + os * __expf(om - M);
            const float l = M + __logf(S);
            if (h2 == 0 && row_ok) {
                a.lse[o0 + slot] = wr != 0.f ? l : 0.f;
                acc_l += wr != 0.f ? wr * (l - (st_t + ot)) : 0.f;
                acc_w += wr;
            }
        }
        ++slot;
        slot_lo = slot_hi;
        slot_hi += C;
        st_m = -INFINITY; st_s = 0.f; st_t = 0.f; st_i = 0;
        tcol = tcol_n; wr = wr_n; ls = ls_n;
        fetch(slot + 1, tcol_n, wr_n, ls_n);
    };

    const char* lbase = smem + lane * 16;
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c) {
        // chunk c has landed for everybody (this wave's pieces: all but the newest issue are done; the others': the barrier),
        // and the slot of chunk c - 1 is free for chunk c + 2
        if (c + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + 2 < n_chunks) issue(c + 2);
        const char* sc = lbase + (c % HNBUF) * HSLOT;
        floatx16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Frag8 a0, a1;
            a0.u = *reinterpret_cast<const uint4*>(sc + ks * FRAG);
            a1.u = *reinterpret_cast<const uint4*>(sc + (16 + ks) * FRAG);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, xf[ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, xf[ks], acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int cb = HCH * c + 32 * t;                        // first column of the tile
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *reinterpret_cast<const float4*>(sbias + cb + 8 * q + 4 * h2);
                v[4 * q + 0] = acc[t][4 * q + 0] + b.x; v[4 * q + 1] = acc[t][4 * q + 1] + b.y;
                v[4 * q + 2] = acc[t][4 * q + 2] + b.z; v[4 * q + 3] = acc[t][4 * q + 3] + b.w;
            }
            if (noisy) {
                // the lane's columns come in runs of 4 (cb + 8 q + 4 h2 + e): one key hash per run, one word per element
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t col0 = (uint32_t)(cb + 8 * q + 4 * h2);
                    const uint32_t hk = drop_group(gctx, (uint64_t)row * k4 + (col0 >> 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[4 * q + e] = fmaf(a.temperature, dsvg_gumbel_from_word(drop_word(hk, (uint32_t)e)), v[4 * q + e]);
                }
            }
            if (MODE == MODE_DLOGITS) {
                // every column of the tile belongs to `slot` or - at and past slot_hi - to the next one
                float d[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = cb + hrowmap(r, h2);
                    const bool nx = col >= slot_hi;
                    const float w_ = nx ? wr_n : wr;
                    const float sm = __expf(v[r] - (nx ? ls_n : ls)) - (col == (nx ? tcol_n : tcol) ? 1.f : 0.f);
                    d[r] = (col < a.n_out && w_ != 0.f) ? w_ * g * sm : 0.f;
                }
                floatx16 dv;
#pragma unroll
                for (int r = 0; r < 16; ++r) dv[r] = d[r];
                uint32_t xc[4][4];
                tile_to_cols16(dv, xc);                             // -> this lane's columns cb + 16 h2 .. + 15
                float o[16];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[4 * q + e] = __uint_as_float(xc[q][e]);
                bf16_t* dst = a.dl + my_row * a.ld_d + cb + 16 * h2;
                if (row_ok && cb + 16 * h2 < a.ld_d)
                    *reinterpret_cast<uint4*>(dst) = make_uint4(f2bf_pk(o[0], o[1]), f2bf_pk(o[2], o[3]), f2bf_pk(o[4], o[5]),
                                                                f2bf_pk(o[6], o[7]));
                if (row_ok && cb + 16 * h2 + 8 < a.ld_d)
                    *reinterpret_cast<uint4*>(dst + 8) = make_uint4(f2bf_pk(o[8], o[9]), f2bf_pk(o[10], o[11]),
                                                                    f2bf_pk(o[12], o[13]), f2bf_pk(o[14], o[15]));
                if (slot_hi <= cb + 32 && slot + 1 < a.group) flush();
            } else {
                // the part of the tile that belongs to the current slot, then - if the slot ends inside it - the rest
                int lo = cb;
                for (;;) {
                    const int hi = min(min(cb + 32, slot_hi), a.n_out);
                    if (MODE == MODE_ARGMAX) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int col = cb + hrowmap(r, h2);
                            const bool in = col >= lo && col < hi;
                            if (in && v[r] > st_m) { st_m = v[r]; st_i = col; }
                        }
                    } else {
                        float u[16], cm = -INFINITY;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int col = cb + hrowmap(r, h2);
                            u[r] = (col >= lo && col < hi) ? v[r] : -INFINITY;
                            cm = fmaxf(cm, u[r]);
                            st_t += (col == tcol && col >= lo && col < hi) ? v[r] : 0.f;
                        }
                        if (cm > -INFINITY) {       // (a lane may hold no column of the range: nothing to add)
                            const float mn = fmaxf(st_m, cm);
                            float add = 0.f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) add += __expf(u[r] - mn);
                            st_s = st_s * __expf(st_m - mn) + add;
                            st_m = mn;
                        }
                    }
                    if (slot_hi > cb + 32 || slot_hi > a.n_out) break;         // the slot goes on in the next tile
                    flush();
                    if (slot >= a.group) break;
                    lo = slot_lo;
                    if (lo >= cb + 32) break;
                }
            }
        }
    }
    if (MODE == MODE_LSE) {
        // fixed-order sums: lanes of a wave, then the 8 waves
        acc_l = wave_sum(acc_l);
        acc_w = wave_sum(acc_w);
        if (lane == 0) { sred[2 * wave] = acc_l; sred[2 * wave + 1] = acc_w; }
        __syncthreads();
        if (tid == 0) {
            float sl = 0.f, sw = 0.f;
            for (int i = 0; i < 8; ++i) { sl += sred[2 * i]; sw += sred[2 * i + 1]; }
            a.part[blockIdx.x * 2 + 0] = sl;
            a.part[blockIdx.x * 2 + 1] = sw;
        }
    }
}

constexpr size_t HEAD_LDS = (size_t)HNBUF * HSLOT + H_MAX_COLS * 4 + 64;

int head_check(const void* x, const void* img, const float* bias, int64_t rows, int32_t n_out, int32_t C, const char* who) {
    if (!x || !img || !bias || rows <= 0 || n_out <= 0 || C < 64 || n_out % C || n_out > H_MAX_COLS - HCH
        || rows >= (1ll << 31) - HTOK || (((uintptr_t)x | (uintptr_t)img) & 15)) {
        (void)who;
        dsvg_set_error("fused argument head: bad arguments (n_out = group * C <= 3008, C >= 64, 16-byte aligned x / image)");
        return -1;
    }
    return 0;
}

template <int MODE>
int head_launch(const HeadArgs& a, hipStream_t st, const char* who) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(head_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)HEAD_LDS) != hipSuccess) {
            dsvg_set_error("head kernel: cannot reserve its dynamic LDS");
            return -1;
        }
        attr = true;
    }
    const int nb = (int)((a.rows + HTOK - 1) / HTOK);
    hipLaunchKernelGGL(head_kernel<MODE>, dim3(nb), dim3(512), HEAD_LDS, st, a);
    DSVG_LAUNCH_CHECK(who);
    return 0;
}

}  // namespace

extern "C" int64_t dsvg_head_pack_elems(int32_t n_out) { return (int64_t)((n_out + HCH - 1) / HCH) * (HSLOT / 2); }

extern "C" int dsvg_head_pack(const void* weight_bf16, int32_t n_out, void* packed, void* stream) {
    DSVG_CHECK_ARG(weight_bf16 && packed && n_out > 0, "head_pack: bad arguments");
    DSVG_CHECK_ARG((((uintptr_t)weight_bf16 | (uintptr_t)packed) & 15) == 0, "head_pack: operands must be 16-byte aligned");
    const int n_chunks = (n_out + HCH - 1) / HCH;
    const long long n = (long long)n_chunks * 32 * 64;
    hipLaunchKernelGGL(head_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)weight_bf16, n_out, n_chunks, (bf16_t*)packed);
    DSVG_LAUNCH_CHECK("head_pack");
    return 0;
}

extern "C" int dsvg_head_argmax(const void* x, const void* packed, const float* bias, int64_t rows, int32_t n_out, int32_t C,
                                int32_t* out_idx, void* stream) {
    if (head_check(x, packed, bias, rows, n_out, C, "head_argmax")) return -1;
    DSVG_CHECK_ARG(out_idx, "head_argmax: null output");
    HeadArgs a{};
    a.x = (const bf16_t*)x; a.img = (const bf16_t*)packed; a.bias = bias; a.rows = rows; a.n_out = n_out; a.C = C;
    a.group = n_out / C; a.out_idx = out_idx;
    return head_launch<MODE_ARGMAX>(a, (hipStream_t)stream, "head_argmax");
}

extern "C" int dsvg_head_sample(const void* x, const void* packed, const float* bias, int64_t rows, int32_t n_out, int32_t C,
                                float temperature, const void* seed, uint32_t site, int32_t* out_idx, void* stream) {
    if (head_check(x, packed, bias, rows, n_out, C, "head_sample")) return -1;
    DSVG_CHECK_ARG(out_idx && seed && temperature > 0.f, "head_sample: needs an output, a seed and a temperature > 0");
    HeadArgs a{};
    a.x = (const bf16_t*)x; a.img = (const bf16_t*)packed; a.bias = bias; a.rows = rows; a.n_out = n_out; a.C = C;
    a.group = n_out / C; a.out_idx = out_idx; a.temperature = temperature; a.seed = (const uint64_t*)seed; a.site = site;
    return head_launch<MODE_ARGMAX>(a, (hipStream_t)stream, "head_sample");
}

extern "C" int64_t dsvg_head_lse_workspace_bytes(int64_t rows) { return ((rows + HTOK - 1) / HTOK) * 2 * (int64_t)sizeof(float); }

extern "C" int dsvg_head_lse(const void* x, const void* packed, const float* bias, int64_t rows, int32_t n_out, int32_t C,
                             const int32_t* target, const float* w, const int32_t* tok_idx, float* lse, float* sum_count,
                             float* workspace, int64_t workspace_bytes, void* stream) {
    if (head_check(x, packed, bias, rows, n_out, C, "head_lse")) return -1;
    DSVG_CHECK_ARG(target && lse && sum_count && workspace && workspace_bytes >= dsvg_head_lse_workspace_bytes(rows),
                   "head_lse: null pointer / workspace too small");
    HeadArgs a{};
    a.x = (const bf16_t*)x; a.img = (const bf16_t*)packed; a.bias = bias; a.rows = rows; a.n_out = n_out; a.C = C;
    a.group = n_out / C; a.target = target; a.w = w; a.tok_idx = tok_idx; a.lse = lse; a.part = workspace;
    if (head_launch<MODE_LSE>(a, (hipStream_t)stream, "head_lse")) return -1;
    return dsvg_reduce_partials_strided(workspace, (rows + HTOK - 1) / HTOK, 2, 2, sum_count, 0, (hipStream_t)stream);
}

extern "C" int dsvg_head_dlogits(const void* x, const void* packed, const float* bias, int64_t rows, int32_t n_out, int32_t C,
                                 const int32_t* target, const float* w, const int32_t* tok_idx, const float* lse,
                                 const float* sum_count, const float* gscale, float coef, void* dlogits, int64_t ld_d,
                                 void* stream) {
    if (head_check(x, packed, bias, rows, n_out, C, "head_dlogits")) return -1;
    DSVG_CHECK_ARG(target && lse && sum_count && dlogits, "head_dlogits: null pointer");
    DSVG_CHECK_ARG(ld_d >= n_out && (ld_d % 8) == 0 && ((uintptr_t)dlogits & 15) == 0 && ld_d <= ((n_out + HCH - 1) / HCH) * HCH,
                   "head_dlogits: ld_d must be n_out rounded up to a multiple of 8 (at most to the next multiple of 64)");
    HeadArgs a{};
    a.x = (const bf16_t*)x; a.img = (const bf16_t*)packed; a.bias = bias; a.rows = rows; a.n_out = n_out; a.C = C;
    a.group = n_out / C; a.target = target; a.w = w; a.tok_idx = tok_idx; a.lse = const_cast<float*>(lse);
    a.sum_count = sum_count; a.gscale = gscale; a.coef = coef; a.dl = (bf16_t*)dlogits; a.ld_d = ld_d;
    return head_launch<MODE_DLOGITS>(a, (hipStream_t)stream, "head_dlogits");
}
