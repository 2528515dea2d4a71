// Hungarian self-matching of predicted groups to target groups (HierarchicalSelfMatching,
// deepsvg/model/config.py:101-108): SVGTransformer.perfect_matching, deepsvg/model/model.py:311-350.
//
// The reference repeats every logit tensor G times, takes three cross-entropies of the repeated tensors
// (N*G*Gp*S*n_args*257 floats: 11.5 GB at 512 icons), copies the cost matrices to the host and calls scipy's
// linear_sum_assignment once per icon.  Here:
//   dsvg_match_costs   one workgroup per (icon, predicted group): log-sum-exp of that group's logits ONCE (the
//                      dense args_logits are read exactly once, HBM-bound), then the masked target log-likelihoods of
//                      all G target groups against it  ->  cost[n, g, p] = 2 * args + 1 * cmd + 1 * visibility
//   dsvg_match_assign  one workgroup per icon: exact minimum over all injective maps visible target -> prediction
//                      (at most 8! = 40320 for the 8 groups of the model) - no host round trip, no per-icon loop
#include "dsvg_common.h"
#include "../../include/dsvg.h"

namespace {
constexpr int MC_THREADS = 256;
constexpr int MC_MAX_ROWS = 64 * 11;      // S * n_args rows of one predicted group
constexpr int MC_MAX_G = 8;

template <typename T>
__global__ __launch_bounds__(MC_THREADS) void match_costs_kernel(
    const T* __restrict__ cmd_logits, long long ld_c, const T* __restrict__ args_logits, long long ld_a,
    const T* __restrict__ vis_logits, long long ld_v, const float* __restrict__ tgt_commands,
    const float* __restrict__ tgt_args, const float* __restrict__ cam, int G, int Gp, int S1, int A, int C, int n_cmd,
    int eos, float w_args, float w_cmd, float w_vis, float* __restrict__ cost, int32_t* __restrict__ vis_out) {
    __shared__ float lse_a[MC_MAX_ROWS];
    __shared__ float lse_c[64];
    __shared__ float lse_v;
    __shared__ int first_eos[MC_MAX_G], visible[MC_MAX_G];
    __shared__ float red[4][4];
    const int S = S1 - 1;
    const long long np = blockIdx.x;            // n * Gp + p
    const long long n = np / Gp;
    const int p = (int)(np % Gp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- log-sum-exp of every row of this predicted group -------------------------------------------------------
    const T* arow = args_logits + np * S * ld_a;
    for (int row = wave; row < S * A; row += MC_THREADS / 64) {
        const T* x = arow + (long long)(row / A) * ld_a + (row % A) * C;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, Elem<T>::ld(x + c));
        m = wave_max(m);
        float e = 0.f;
        for (int c = lane; c < C; c += 64) e += __expf(Elem<T>::ld(x + c) - m);
        e = wave_sum(e);
        if (lane == 0) lse_a[row] = m + __logf(e);
    }
    const T* crow = cmd_logits + np * S * ld_c;
    if (tid < S) {
        const T* x = crow + (long long)tid * ld_c;
        float m = -INFINITY;
        for (int c = 0; c < n_cmd; ++c) m = fmaxf(m, Elem<T>::ld(x + c));
        float e = 0.f;
        for (int c = 0; c < n_cmd; ++c) e += __expf(Elem<T>::ld(x + c) - m);
        lse_c[tid] = m + __logf(e);
    }
    const T* vrow = vis_logits + np * ld_v;
    if (tid == 0) {
        const float a = Elem<T>::ld(vrow), b = Elem<T>::ld(vrow + 1);
        const float m = fmaxf(a, b);
        lse_v = m + __logf(__expf(a - m) + __expf(b - m));
    }
    // ---- masks of the target groups, on the sequence WITHOUT its SOS column (model.py:314-315,388) ---------------
    if (tid < G) {
        const float* tc = tgt_commands + (n * G + tid) * S1 + 1;
        int fe = S, ne = 0;
        for (int s = 0; s < S; ++s) {
            const bool e = ((int)tc[s] == eos);
            ne += e;
            if (e && fe == S) fe = s;
        }
        first_eos[tid] = fe;
        visible[tid] = ne < S - 1 ? 1 : 0;       // _get_visibility_mask (utils.py:45-56)
        if (p == 0) vis_out[n * G + tid] = visible[tid];
    }
    __syncthreads();

    // ---- cost of pairing each target group g with this prediction -----------------------------------------------
    for (int g = 0; g < G; ++g) {
        const float* tc = tgt_commands + (n * G + g) * S1 + 1;
        const float* ta = tgt_args + ((n * G + g) * S1 + 1) * A;
        float sa = 0.f, na = 0.f, sc = 0.f, nc = 0.f;
        for (int i = tid; i < S * A; i += MC_THREADS) {
            const int s = i / A, a = i % A;
            const int c = min(max((int)tc[s], 0), n_cmd - 1);
            const float mk = cam[c * A + a];                              // CMD_ARGS_MASK[tgt_commands] (:327)
            if (mk != 0.f) {
                const int t = min(max((int)ta[s * A + a] + 1, 0), C - 1);      // shift due to -1 PAD_VAL (:329)
                sa += mk * (lse_a[i] - Elem<T>::ld(arow + (long long)s * ld_a + a * C + t));
                na += mk;
            }
        }
        const int fe = first_eos[g];
        if (tid < S) {
            // _get_padding_mask(extended=True) * visibility (:315): valid before the first EOS, extended by 3 positions
            const int s = tid;
            const bool on = visible[g] && (s < fe || (s >= 3 && s - 3 < fe));
            if (on) {
                const int c = min(max((int)tc[s], 0), n_cmd - 1);
                sc += lse_c[s] - Elem<T>::ld(crow + (long long)s * ld_c + c);
                nc += 1.f;
            }
        }
        sa = wave_sum(sa); na = wave_sum(na); sc = wave_sum(sc); nc = wave_sum(nc);
        if (lane == 0) { red[wave][0] = sa; red[wave][1] = na; red[wave][2] = sc; red[wave][3] = nc; }
        __syncthreads();
        if (tid == 0) {
            float r[4] = {0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < MC_THREADS / 64; ++w)
                for (int k = 0; k < 4; ++k) r[k] += red[w][k];
            const float l_args = r[0] / r[1];                              // 0/0 on groups without arguments (:336)
            const float l_cmd = r[2] / r[3];                               // (:337)
            const float l_vis = lse_v - Elem<T>::ld(vrow + visible[g]);   // (:334)
            cost[(n * G + g) * Gp + p] = w_args * l_args + w_cmd * l_cmd + w_vis * l_vis;      // (:339)
        }
        __syncthreads();
    }
}

// decode the id-th injective map rows 0..nv-1 -> distinct columns of 0..Gp-1 (mixed radix Gp, Gp-1, ...)
__device__ __forceinline__ void decode_map(int id, int nv, int Gp, int (&col)[MC_MAX_G]) {
    unsigned avail = (1u << Gp) - 1u;
    for (int i = 0; i < nv; ++i) {
        const int base = Gp - i;
        int d = id % base;
        id /= base;
        unsigned a = avail;
        while (d--) a &= a - 1;                 // drop the d lowest available columns
        const int c = __builtin_ctz(a);
        col[i] = c;
        avail &= ~(1u << c);
    }
}

__global__ __launch_bounds__(256) void match_assign_kernel(const float* __restrict__ cost, const int32_t* __restrict__ vis,
                                                           int G, int Gp, int32_t* __restrict__ assign,
                                                           int32_t* __restrict__ idx, int32_t* __restrict__ inv) {
    __shared__ float c[MC_MAX_G][MC_MAX_G];
    __shared__ int vrow[MC_MAX_G];
    __shared__ int nvs;
    __shared__ float bcost[256];
    __shared__ int bid[256];
    const long long n = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < G * Gp) c[tid / Gp][tid % Gp] = cost[n * G * Gp + tid];
    if (tid == 0) {
        int k = 0;
        for (int g = 0; g < G; ++g)
            if (vis[n * G + g]) vrow[k++] = g;   // rows of costs[mask] (model.py:344): the visible targets, in order
        nvs = k < Gp ? k : Gp;
    }
    __syncthreads();
    const int nv = nvs;
    int total = 1;
    for (int i = 0; i < nv; ++i) total *= Gp - i;
    float best = INFINITY;
    int best_id = 0x7fffffff;
    for (int id = tid; id < total; id += 256) {
        int col[MC_MAX_G];
        decode_map(id, nv, Gp, col);
        float s = 0.f;
        for (int i = 0; i < nv; ++i) s += c[vrow[i]][col[i]];
        if (s < best) { best = s; best_id = id; }
    }
    bcost[tid] = best;
    bid[tid] = best_id;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float oc = bcost[tid + o];
            const int oi = bid[tid + o];
            if (oc < bcost[tid] || (oc == bcost[tid] && oi < bid[tid])) { bcost[tid] = oc; bid[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        int col[MC_MAX_G];
        const int win = bid[0] == 0x7fffffff ? 0 : bid[0];
        decode_map(win, nv, Gp, col);
        unsigned used = 0;
        for (int i = 0; i < nv; ++i) used |= 1u << col[i];
        int k = nv;
        for (int cidx = 0; cidx < Gp; ++cidx)       // the unmatched predictions follow in ascending order (:347)
            if (!(used & (1u << cidx))) col[k++] = cidx;
        for (int j = 0; j < Gp; ++j) {
            assign[n * Gp + j] = col[j];
            idx[n * Gp + j] = (int32_t)(n * Gp + col[j]);
            inv[n * Gp + col[j]] = (int32_t)(n * Gp + j);
        }
    }
}
}  // namespace

extern "C" int dsvg_match_costs(int32_t dtype, const void* cmd_logits, int64_t ld_c, const void* args_logits,
                                int64_t ld_a, const void* vis_logits, int64_t ld_v, const float* tgt_commands,
                                const float* tgt_args, const float* cmd_args_mask, int64_t N, int32_t G, int32_t Gp,
                                int32_t S1, int32_t n_args, int32_t args_dim, int32_t n_cmd, int32_t eos_id,
                                float w_args, float w_cmd, float w_vis, float* cost, int32_t* visible, void* stream) {
    DSVG_CHECK_ARG(cmd_logits && args_logits && vis_logits && tgt_commands && tgt_args && cmd_args_mask && cost && visible,
                   "match_costs: null pointer");
    DSVG_CHECK_ARG(N > 0 && G > 0 && G <= MC_MAX_G && Gp > 0 && S1 > 1 && S1 <= 64 && (S1 - 1) * n_args <= MC_MAX_ROWS &&
                       n_args > 0 && args_dim > 1 && n_cmd > 0 && N * Gp < (1ll << 31),
                   "match_costs: bad shape (N=%lld G=%d Gp=%d S1=%d)", (long long)N, G, Gp, S1);
    DSVG_CHECK_ARG(ld_a >= (int64_t)n_args * args_dim && ld_c >= n_cmd && ld_v >= 2, "match_costs: bad row stride");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(match_costs_kernel<bf16_t>, dim3((unsigned)(N * Gp)), dim3(MC_THREADS), 0, st,
                           (const bf16_t*)cmd_logits, (long long)ld_c, (const bf16_t*)args_logits, (long long)ld_a,
                           (const bf16_t*)vis_logits, (long long)ld_v, tgt_commands, tgt_args, cmd_args_mask, G, Gp, S1,
                           n_args, args_dim, n_cmd, eos_id, w_args, w_cmd, w_vis, cost, visible);
    else
        hipLaunchKernelGGL(match_costs_kernel<float>, dim3((unsigned)(N * Gp)), dim3(MC_THREADS), 0, st,
                           (const float*)cmd_logits, (long long)ld_c, (const float*)args_logits, (long long)ld_a,
                           (const float*)vis_logits, (long long)ld_v, tgt_commands, tgt_args, cmd_args_mask, G, Gp, S1,
                           n_args, args_dim, n_cmd, eos_id, w_args, w_cmd, w_vis, cost, visible);
    DSVG_LAUNCH_CHECK("match_costs");
    return 0;
}

extern "C" int dsvg_match_assign(const float* cost, const int32_t* visible, int64_t N, int32_t G, int32_t Gp,
                                 int32_t* assign, int32_t* idx, int32_t* inv, void* stream) {
    DSVG_CHECK_ARG(cost && visible && assign && idx && inv, "match_assign: null pointer");
    DSVG_CHECK_ARG(N > 0 && G > 0 && G <= MC_MAX_G && Gp >= G && Gp <= MC_MAX_G && N * Gp < (1ll << 31),
                   "match_assign: exhaustive search covers up to %d groups (G=%d Gp=%d)", MC_MAX_G, G, Gp);
    hipLaunchKernelGGL(match_assign_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, cost, visible, G, Gp,
                       assign, idx, inv);
    DSVG_LAUNCH_CHECK("match_assign");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// arg-max over the class axis of a logits matrix: logical row r of [rows, C] lives at logits + (r / group) * ld +
// (r % group) * C (same addressing as the masked cross-entropy).  The temperature -> 0 limit of the reference's
// _sample_categorical (deepsvg/model/utils.py:75-80), used by greedy_sample(temperature=0): no fp32 copy of the
// logits, no softmax, no multinomial draw.  Ties go to the lowest class index.  One wave per row.
// ---------------------------------------------------------------------------------------------------------------
namespace {
// temperature > 0: a draw from softmax(logits / temperature) as a Gumbel arg-max (dsvg_gumbel, dsvg_common.h): the noise of
// logical row r = (token r / group, slot r % group), class c is that of element (token, slot * C + c) of the token's
// [group * C] logit row - the same element dsvg_head_sample perturbs, so both entry points draw the same sample
template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const T* __restrict__ logits, long long ld, int group,
                                                          long long rows, int C, int32_t* __restrict__ out,
                                                          float temperature, const uint64_t* __restrict__ seed, uint32_t site) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* x = logits + (r / group) * ld + (r % group) * (long long)C;
    const bool noisy = temperature > 0.f;
    const DropCtx gctx = drop_make(noisy ? 0.5f : 0.f, seed, site);
    const uint32_t k4 = (uint32_t)(group * C + 3) >> 2, col0 = (uint32_t)(r % group) * (uint32_t)C;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        float v = Elem<T>::ld(x + c);
        if (noisy) v = fmaf(temperature, dsvg_gumbel(gctx, (uint64_t)(r / group), k4, col0 + (uint32_t)c), v);
        if (v > best) { best = v; bi = c; }          // ascending c per lane: the first maximum wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) out[r] = bi == 0x7fffffff ? 0 : bi;
}
}  // namespace

static int argmax_rows_launch(int32_t dtype, const void* logits, int64_t ld, int32_t group, int64_t rows, int32_t C,
                              int32_t* out, float temperature, const void* seed, uint32_t site, void* stream, const char* who) {
    const unsigned grid = (unsigned)dsvg_cdiv(rows, 4);
    if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(argmax_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)logits, (long long)ld, group, (long long)rows, C, out, temperature,
                           (const uint64_t*)seed, site);
    else
        hipLaunchKernelGGL(argmax_rows_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)logits, (long long)ld, group, (long long)rows, C, out, temperature,
                           (const uint64_t*)seed, site);
    DSVG_LAUNCH_CHECK(who);
    return 0;
}

extern "C" int dsvg_argmax_rows(int32_t dtype, const void* logits, int64_t ld, int32_t group, int64_t rows, int32_t C,
                                int32_t* out, void* stream) {
    DSVG_CHECK_ARG(logits && out && rows > 0 && group > 0 && C > 0 && ld >= (int64_t)group * C, "argmax_rows: bad args");
    return argmax_rows_launch(dtype, logits, ld, group, rows, C, out, 0.f, nullptr, 0u, stream, "argmax_rows");
}

extern "C" int dsvg_sample_rows(int32_t dtype, const void* logits, int64_t ld, int32_t group, int64_t rows, int32_t C,
                                float temperature, const void* seed, uint32_t site, int32_t* out, void* stream) {
    DSVG_CHECK_ARG(logits && out && rows > 0 && group > 0 && C > 0 && ld >= (int64_t)group * C, "sample_rows: bad args");
    DSVG_CHECK_ARG(seed && temperature > 0.f, "sample_rows: needs a seed and a temperature > 0");
    return argmax_rows_launch(dtype, logits, ld, group, rows, C, out, temperature, seed, site, stream, "sample_rows");
}
