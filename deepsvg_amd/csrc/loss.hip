// SVGLoss pieces (deepsvg/model/loss.py:19-65): target/mask construction and a masked cross-entropy
// that never gathers rows (the reference uses boolean-mask indexing = dynamic shapes + host syncs).
// One 64-lane wave per logits row; rows with weight 0 are skipped entirely in the forward pass, so only
// the ~15-20 % of args_logits rows enabled by CMD_ARGS_MASK are ever read.
#include "dsvg_common.h"
#include "../../include/dsvg.h"

// ---------------------------------------------------------------------------------------------
// targets and weights.  tgt_commands [n_seq, S1], tgt_args [n_seq, S1, n_args], S = S1 - 1.
//   pm[s]  = no EOS at positions <= s                                   (model/utils.py:22-23)
//   ext[s] = min(1, pm[s] + pm[s-3])   (canonical, non-aliased reading of model/utils.py:25-28)
//   vis    = (#EOS < S1 - 1)                                            (model/utils.py:51)
//   cmd_w[s'] = ext[s'+1] * vis, cmd_tgt[s'] = cmd[s'+1]                (loss.py:35-36,49,53)
//   arg_w[s',a] = CMD_ARGS_MASK[cmd[s'+1], a], arg_tgt = arg[s'+1,a]+1  (loss.py:51,54)
// ---------------------------------------------------------------------------------------------
__global__ void loss_targets_kernel(const float* __restrict__ tc, const float* __restrict__ ta,
                                    const float* __restrict__ cam, long long n_seq, int S1, int n_args, int n_cmd,
                                    int eos, int* __restrict__ cmd_tgt, float* __restrict__ cmd_w,
                                    int* __restrict__ arg_tgt, float* __restrict__ arg_w, int* __restrict__ vis_tgt,
                                    const int32_t* __restrict__ seq_perm) {
    // seq_perm != null: output sequence b is source sequence seq_perm[b] (the visible-first order of the second decoder
    // stage: the heads then read their targets in the order the stage's rows are in); vis_tgt stays in source order
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_seq) return;
    const long long src = seq_perm ? seq_perm[b] : b;
    const int S = S1 - 1;
    const float* row = tc + src * S1;
    // padding mask = positions before the first EOS (utils.py:20-24), `extended` by the mask shifted 3 positions
    // (utils.py:25-30, canonical non-aliased reading): position t is on iff t < fe or (t >= 3 and t - 3 < fe)
    int n_eos = 0, fe = S1;
    for (int s = S1 - 1; s >= 0; --s) {
        const bool e = ((int)row[s] == eos);
        n_eos += e;
        if (e) fe = s;
    }
    const int vis = (n_eos < S1 - 1) ? 1 : 0;
    vis_tgt[src] = vis;
    for (int s = 0; s < S; ++s) {
        int c = (int)row[s + 1];
        c = min(max(c, 0), n_cmd - 1);
        cmd_tgt[b * S + s] = c;
        const int t = s + 1;
        const bool ext = t < fe || (t >= 3 && t - 3 < fe);
        cmd_w[b * S + s] = (ext && vis) ? 1.f : 0.f;
        for (int a = 0; a < n_args; ++a) {
            arg_tgt[(b * S + s) * n_args + a] = (int)ta[(src * S1 + s + 1) * n_args + a] + 1;
            arg_w[(b * S + s) * n_args + a] = cam[c * n_args + a];
        }
    }
}

extern "C" int dsvg_loss_targets(const float* tgt_commands, const float* tgt_args, const float* cmd_args_mask,
                                 int64_t n_seq, int32_t S1, int32_t n_args, int32_t n_cmd, int32_t eos_id,
                                 int32_t* cmd_tgt, float* cmd_w, int32_t* arg_tgt, float* arg_w, int32_t* vis_tgt,
                                 const int32_t* seq_perm, void* stream) {
    DSVG_CHECK_ARG(tgt_commands && tgt_args && cmd_args_mask && cmd_tgt && cmd_w && arg_tgt && arg_w && vis_tgt,
                   "loss_targets: null pointer");
    DSVG_CHECK_ARG(n_seq > 0 && S1 > 1 && S1 <= 4096, "loss_targets: bad shape (S1=%d)", S1);
    hipLaunchKernelGGL(loss_targets_kernel, dim3(dsvg_cdiv(n_seq, 64)), dim3(64), 0, (hipStream_t)stream, tgt_commands,
                       tgt_args, cmd_args_mask, (long long)n_seq, S1, n_args, n_cmd, eos_id, cmd_tgt, cmd_w, arg_tgt,
                       arg_w, vis_tgt, seq_perm);
    DSVG_LAUNCH_CHECK("loss_targets");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// masked CE.  Row r of the logical [rows, C] matrix lives at logits + (r / group) * ld + (r % group) * C
// (group = 11 arg slots per token for args_logits, 1 otherwise).
// ---------------------------------------------------------------------------------------------
constexpr int CE_MAX_BLOCKS = 2048;

// Forward.  A wave takes RC (16 or 64) consecutive rows at a time: one coalesced load of their weights and a ballot
// pick the rows that carry loss (88 % of the 1.4 M argument rows do not; per-row weight loads were a chain of
// dependent global loads); each quarter-wave (16 lanes) then handles one of those rows with the row's logits held in
// registers (one pass over memory, C <= 272), i.e. 4 rows in flight per wave.  The generic kernel below covers C > 272.
constexpr int CE_NV = 17;           // register-resident row: 16 lanes x 17 elements
template <typename T>
__global__ __launch_bounds__(256) void masked_ce_fwd16_kernel(const T* __restrict__ logits, long long ld, int group,
                                                              const int* __restrict__ target,
                                                              const float* __restrict__ w, long long rows, int C,
                                                              float* __restrict__ lse, float* __restrict__ part, int RC,
                                                              const int32_t* __restrict__ tok_idx) {
    // tok_idx != null: COMPACT logits - row ro of `logits` / `lse` belongs to token tok_idx[ro / group] (negative =
    // list padding, weight 0); targets and weights stay indexed by the source token
    __shared__ float red[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, sl = lane & 15;
    float acc_l = 0.f, acc_w = 0.f;
    for (long long r0 = ((long long)blockIdx.x * 4 + wave) * RC; r0 < rows; r0 += (long long)gridDim.x * 4 * RC) {
        const long long rl = r0 + lane;
        const bool mine = lane < RC && rl < rows;
        long long rsrc = rl;                                   // source row (targets, weights)
        if (mine && tok_idx) {
            const long long tk = tok_idx[rl / group];
            rsrc = tk < 0 ? -1 : tk * group + rl % group;
        }
        const float wl = (mine && rsrc >= 0) ? (w ? w[rsrc] : 1.f) : 0.f;
        if (mine && wl == 0.f) lse[rl] = 0.f;
        const unsigned long long live = __ballot(wl != 0.f);
        const int n_live = __popcll(live);
        const int rank = __popcll(live & ((1ull << lane) - 1ull));       // rank of this lane's row among the live ones
        for (int k = 0; k < n_live; k += 4) {
            // quarter-wave `sub` takes the live row of rank k + sub
            int j = -1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long hit = __ballot(wl != 0.f && rank == k + q);
                if (q == sub && hit) j = __builtin_ctzll(hit);
            }
            const float wr = __shfl(wl, j < 0 ? 0 : j, 64);
            const long long rs_j = __shfl(rsrc, j < 0 ? 0 : j, 64);
            if (j >= 0) {
                const long long r = r0 + j;
                const T* p = logits + (r / group) * ld + (r % group) * (long long)C;
                float v[CE_NV];
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < CE_NV; ++i) {
                    // unconditional load from a clamped column, select afterwards: a guarded load makes hipcc branch
                    // around every load and wait vmcnt(0) after each (17 dependent round trips per row)
                    const int c = sl + 16 * i;
                    const float x = Elem<T>::ld(p + min(c, C - 1));
                    v[i] = c < C ? x : -INFINITY;
                    m = fmaxf(m, v[i]);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
                float sacc = 0.f;
#pragma unroll
                for (int i = 0; i < CE_NV; ++i) sacc += (sl + 16 * i < C) ? __expf(v[i] - m) : 0.f;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 16);
                const float l = m + __logf(sacc);
                if (sl == 0) {
                    lse[r] = l;
                    int t = target[rs_j];
                    t = min(max(t, 0), C - 1);
                    acc_l += wr * (l - Elem<T>::ld(p + t));
                    acc_w += wr;
                }
            }
        }
    }
    // lanes 0, 16, 32, 48 hold the quarter-wave sums: fixed-order combine
    acc_l = acc_l + __shfl(acc_l, 16, 64) + (__shfl(acc_l, 32, 64) + __shfl(acc_l, 48, 64));
    acc_w = acc_w + __shfl(acc_w, 16, 64) + (__shfl(acc_w, 32, 64) + __shfl(acc_w, 48, 64));
    if (lane == 0) { red[wave][0] = acc_l; red[wave][1] = acc_w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        part[blockIdx.x * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void masked_ce_fwd_kernel(const T* __restrict__ logits, long long ld, int group,
                                                            const int* __restrict__ target, const float* __restrict__ w,
                                                            long long rows, int C, float* __restrict__ lse,
                                                            float* __restrict__ part,
                                                            const int32_t* __restrict__ tok_idx) {
    __shared__ float red[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc_l = 0.f, acc_w = 0.f;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        long long rsrc = r;
        if (tok_idx) {
            const long long tk = tok_idx[r / group];
            rsrc = tk < 0 ? -1 : tk * group + r % group;
        }
        const float wr = rsrc < 0 ? 0.f : (w ? w[rsrc] : 1.f);
        if (wr == 0.f) { if (lane == 0) lse[r] = 0.f; continue; }
        const T* p = logits + (r / group) * ld + (r % group) * (long long)C;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, Elem<T>::ld(p + c));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += __expf(Elem<T>::ld(p + c) - m);
        s = wave_sum(s);
        const float l = m + __logf(s);
        if (lane == 0) {
            lse[r] = l;
            int t = target[rsrc];
            t = min(max(t, 0), C - 1);
            acc_l += wr * (l - Elem<T>::ld(p + t));
            acc_w += wr;
        }
    }
    if (lane == 0) { red[wave][0] = acc_l; red[wave][1] = acc_w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        part[blockIdx.x * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void masked_ce_bwd_kernel(const T* __restrict__ logits, long long ld, int group,
                                                            const int* __restrict__ target, const float* __restrict__ w,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ sum_count,
                                                            const float* __restrict__ gscale, float coef,
                                                            T* __restrict__ dlogits, long long ld_d, long long rows,
                                                            int C, const int32_t* __restrict__ tok_idx,
                                                            int logits_compact) {
    // tok_idx != null: output token i is source token tok_idx[i] (a compact list of the tokens that carry loss;
    // negative = padding of the list -> a zero row); `rows` counts OUTPUT rows.  logits_compact: `logits` and `lse`
    // are indexed like the output (compact) instead of by source token
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = coef * (gscale ? *gscale : 1.f) / sum_count[1];
    for (long long ro = (long long)blockIdx.x * 4 + wave; ro < rows; ro += (long long)gridDim.x * 4) {
        const long long tok_o = ro / group;
        const int slot = (int)(ro % group);
        const long long tok = tok_idx ? (long long)tok_idx[tok_o] : tok_o;
        const long long r = tok * group + slot;                 // source row
        const float wr = tok < 0 ? 0.f : (w ? w[r] : 1.f);
        T* q = dlogits + tok_o * ld_d + slot * (long long)C;
        if (wr == 0.f) {
            for (int c = lane; c < C; c += 64) Elem<T>::st(q + c, 0.f);
        } else {
            const T* p = logits + (logits_compact ? tok_o : tok) * ld + slot * (long long)C;
            const float l = lse[logits_compact ? ro : r];
            int t = target[r];
            t = min(max(t, 0), C - 1);
            if (C <= 320) {     // all loads of the row in flight at once (clamped column, see masked_ce_fwd16_kernel)
                float xv[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) xv[i] = Elem<T>::ld(p + min(lane + 64 * i, C - 1));
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int c = lane + 64 * i;
                    if (c < C) Elem<T>::st(q + c, wr * g * (__expf(xv[i] - l) - (c == t ? 1.f : 0.f)));
                }
            } else {
                for (int c = lane; c < C; c += 64) {
                    const float sm = __expf(Elem<T>::ld(p + c) - l);
                    Elem<T>::st(q + c, wr * g * (sm - (c == t ? 1.f : 0.f)));
                }
            }
        }
        if (slot == group - 1) {   // zero the row padding so padded-K GEMMs read zeros
            for (long long c = (long long)group * C + lane; c < ld_d; c += 64) Elem<T>::st(dlogits + tok_o * ld_d + c, 0.f);
        }
    }
}

// The same for bf16 rows whose `group` slots lie back to back in 16-byte-aligned token rows (the argument head: 6 - 11 slots of
// 257 logits): a wave per TOKEN, a lane per 16-byte piece - the per-(token, slot) rows of the kernel above start at odd
// 2-byte offsets (514 B apart), so its accesses are 2 bytes per lane (1.4 TB/s measured on the compact argument logits).
// A piece of 8 columns touches at most two slots (C >= 8); the slots' lse / target / weight sit in lanes 0 .. group - 1.
__global__ __launch_bounds__(256) void masked_ce_bwd_tok_kernel(const bf16_t* __restrict__ logits, long long ld, int group,
                                                                const int* __restrict__ target, const float* __restrict__ w,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ sum_count,
                                                                const float* __restrict__ gscale, float coef,
                                                                bf16_t* __restrict__ dlogits, long long ld_d, long long n_tok,
                                                                int C, const int32_t* __restrict__ tok_idx,
                                                                int logits_compact) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = coef * (gscale ? *gscale : 1.f) / sum_count[1];
    const int width = group * C;
    for (long long tok_o = (long long)blockIdx.x * 4 + wave; tok_o < n_tok; tok_o += (long long)gridDim.x * 4) {
        const long long tok = tok_idx ? (long long)tok_idx[tok_o] : tok_o;
        float my_w = 0.f, my_l = 0.f;
        int my_t = 0;
        if (lane < group && tok >= 0) {
            const long long r = tok * group + lane;
            my_w = w ? w[r] : 1.f;
            my_l = lse[logits_compact ? tok_o * group + lane : r];
            my_t = min(max(target[r], 0), C - 1) + lane * C;        // column of the token row
        }
        const bf16_t* p = logits + (logits_compact ? tok_o : max(tok, 0LL)) * ld;
        bf16_t* q = dlogits + tok_o * ld_d;
        for (int base = 0; base < ld_d; base += 512) {      // (wave-uniform trip count: the shuffles read lanes < group)
            const int c0 = base + 8 * lane;
            const int s0 = min(c0 / C, group - 1), s1 = min(s0 + 1, group - 1);
            const int edge = (s0 + 1) * C;                          // first column of the next slot
            const float w0 = __shfl(my_w, s0, 64), w1 = __shfl(my_w, s1, 64);
            const float l0 = __shfl(my_l, s0, 64), l1 = __shfl(my_l, s1, 64);
            const int t0 = __shfl(my_t, s0, 64), t1 = __shfl(my_t, s1, 64);
            float v[8];
            if (c0 < width && (w0 != 0.f || w1 != 0.f)) {
                const uint4 raw = *reinterpret_cast<const uint4*>(p + min(c0, (int)ld - 8));
                const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = __uint_as_float(rw[e] << 16);
                    v[2 * e + 1] = __uint_as_float(rw[e] & 0xffff0000u);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c0 + e;
                    const bool hi = c >= edge;
                    const float wr = hi ? w1 : w0;
                    const float sm = __expf(v[e] - (hi ? l1 : l0)) - (c == (hi ? t1 : t0) ? 1.f : 0.f);
                    v[e] = (c < width && wr != 0.f) ? wr * g * sm : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            if (c0 < ld_d)
                *reinterpret_cast<uint4*>(q + c0) =
                    make_uint4(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]), f2bf_pk(v[4], v[5]), f2bf_pk(v[6], v[7]));
        }
    }
}

// ... and for the narrow heads (command head: 7 classes, visibility head: 2; one slot per token, bf16 rows padded to 8
// columns in dlogits): a THREAD per row - the wave-per-row kernel above spends a wave on 7 elements (34 us for 127 k rows)
__global__ __launch_bounds__(256) void masked_ce_bwd_narrow_kernel(const bf16_t* __restrict__ logits, long long ld,
                                                                   const int* __restrict__ target, const float* __restrict__ w,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ sum_count,
                                                                   const float* __restrict__ gscale, float coef,
                                                                   bf16_t* __restrict__ dlogits, long long rows, int C) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float g = coef * (gscale ? *gscale : 1.f) / sum_count[1];
    const float wr = w ? w[r] : 1.f;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.f;
    if (wr != 0.f) {
        const float l = lse[r];
        const int t = min(max(target[r], 0), C - 1);
        const bf16_t* p = logits + r * ld;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) v[c] = wr * g * (__expf(bf2f(p[c]) - l) - (c == t ? 1.f : 0.f));
    }
    *reinterpret_cast<uint4*>(dlogits + r * 8) =
        make_uint4(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]), f2bf_pk(v[4], v[5]), f2bf_pk(v[6], v[7]));
}

static int ce_grid(long long rows) {
    long long nb = (rows + 3) / 4;
    return (int)(nb < CE_MAX_BLOCKS ? nb : CE_MAX_BLOCKS);
}

extern "C" int64_t dsvg_masked_ce_workspace_bytes(int64_t rows) { return (int64_t)ce_grid(rows) * 2 * sizeof(float); }

extern "C" int dsvg_masked_ce_fwd(int32_t dtype, const void* logits, int64_t ld, int32_t group, const int32_t* target,
                                  const float* w, int64_t rows, int32_t C, float* lse, float* sum_count,
                                  float* workspace, int64_t workspace_bytes, const int32_t* tok_idx, void* stream) {
    DSVG_CHECK_ARG(logits && target && lse && sum_count && rows > 0 && C > 0 && group > 0, "masked_ce_fwd: bad args");
    DSVG_CHECK_ARG(workspace && workspace_bytes >= dsvg_masked_ce_workspace_bytes(rows), "masked_ce_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ce_grid(rows);       // the workspace holds at least this many partial rows
    int nparts = nb;
    if (C <= 16 * CE_NV) {
        const int RC = rows >= 64LL * 4 * CE_MAX_BLOCKS ? 64 : 16;    // rows per wave sweep
        const int nbq = (int)min((long long)nb, (rows + 4LL * RC - 1) / (4LL * RC));
        nparts = nbq;
        if (dtype == DSVG_F32)
            hipLaunchKernelGGL(masked_ce_fwd16_kernel<float>, dim3(nbq), dim3(256), 0, st, (const float*)logits,
                               (long long)ld, group, target, w, (long long)rows, C, lse, workspace, RC, tok_idx);
        else if (dtype == DSVG_BF16)
            hipLaunchKernelGGL(masked_ce_fwd16_kernel<bf16_t>, dim3(nbq), dim3(256), 0, st, (const bf16_t*)logits,
                               (long long)ld, group, target, w, (long long)rows, C, lse, workspace, RC, tok_idx);
        else { dsvg_set_error("masked_ce_fwd: bad dtype"); return -1; }
    } else if (dtype == DSVG_F32)
        hipLaunchKernelGGL(masked_ce_fwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)logits, (long long)ld,
                           group, target, w, (long long)rows, C, lse, workspace, tok_idx);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(masked_ce_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)logits,
                           (long long)ld, group, target, w, (long long)rows, C, lse, workspace, tok_idx);
    else { dsvg_set_error("masked_ce_fwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("masked_ce_fwd");
    return dsvg_reduce_partials_strided(workspace, nparts, 2, 2, sum_count, 0, st);
}

extern "C" int dsvg_masked_ce_bwd(int32_t dtype, const void* logits, int64_t ld, int32_t group, const int32_t* target,
                                  const float* w, const float* lse, const float* sum_count, const float* gscale,
                                  float coef, void* dlogits, int64_t ld_d, int64_t rows, int32_t C,
                                  const int32_t* tok_idx, int32_t logits_compact, void* stream) {
    DSVG_CHECK_ARG(logits && target && lse && sum_count && dlogits && rows > 0 && C > 0 && group > 0,
                   "masked_ce_bwd: bad args");
    DSVG_CHECK_ARG(ld_d >= (int64_t)group * C, "masked_ce_bwd: ld_d too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_BF16 && C >= 8 && group > 1 && group <= 64 && (ld % 8) == 0 && (ld_d % 8) == 0 && ld >= (int64_t)group * C
        && ld >= 8 && (rows % group) == 0 && (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0 && ld_d < (1 << 30)) {
        const long long n_tok = rows / group;
        hipLaunchKernelGGL(masked_ce_bwd_tok_kernel, dim3(ce_grid(n_tok)), dim3(256), 0, st, (const bf16_t*)logits,
                           (long long)ld, group, target, w, lse, sum_count, gscale, coef, (bf16_t*)dlogits, (long long)ld_d,
                           n_tok, C, tok_idx, logits_compact);
        DSVG_LAUNCH_CHECK("masked_ce_bwd (token rows)");
        return 0;
    }
    if (dtype == DSVG_BF16 && group == 1 && C <= 8 && ld_d == 8 && !tok_idx && ((uintptr_t)dlogits & 15) == 0) {
        hipLaunchKernelGGL(masked_ce_bwd_narrow_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st,
                           (const bf16_t*)logits, (long long)ld, target, w, lse, sum_count, gscale, coef, (bf16_t*)dlogits,
                           (long long)rows, C);
        DSVG_LAUNCH_CHECK("masked_ce_bwd (narrow rows)");
        return 0;
    }
    const int nb = ce_grid(rows);
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(masked_ce_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)logits, (long long)ld,
                           group, target, w, lse, sum_count, gscale, coef, (float*)dlogits, (long long)ld_d,
                           (long long)rows, C, tok_idx, logits_compact);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(masked_ce_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)logits,
                           (long long)ld, group, target, w, lse, sum_count, gscale, coef, (bf16_t*)dlogits,
                           (long long)ld_d, (long long)rows, C, tok_idx, logits_compact);
    else { dsvg_set_error("masked_ce_bwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("masked_ce_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Tokens that carry argument loss: live[i] = index of the i-th token t with any w[t*group + a] != 0 (ascending),
// entries past the count = -1 (capacity n_tok), *count = number of such tokens.  The backward pass of the
// 2827-wide argument head then runs on that compact list (about 30 % of the decoder tokens): rows of dlogits that
// the loss masks out are exact zeros (loss.py:51-54) and contribute neither to dX nor to dW.
// ---------------------------------------------------------------------------------------------
// three small launches: per-block (1024 tokens) flags + local ranks, scan of the block totals, scatter of the list
__global__ __launch_bounds__(1024) void live_rows_rank_kernel(const float* __restrict__ w, long long n_tok, int group,
                                                              int32_t* __restrict__ rank, int32_t* __restrict__ block_sum) {
    __shared__ int part[1024];
    const long long t = (long long)blockIdx.x * 1024 + threadIdx.x;
    int v = 0;
    if (t < n_tok)
        for (int a = 0; a < group; ++a) v |= (w[t * group + a] != 0.f) ? 1 : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int u = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += u;
        __syncthreads();
    }
    if (t < n_tok) rank[t] = v ? part[threadIdx.x] - 1 : -1;       // rank inside the block, -1 = not listed
    if (threadIdx.x == 1023) block_sum[blockIdx.x] = part[1023];
}
__global__ __launch_bounds__(1024) void live_rows_scan_kernel(int32_t* __restrict__ block_sum, int nb,
                                                              int32_t* __restrict__ count) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int b = base + threadIdx.x;
        const int v = b < nb ? block_sum[b] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int u = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += u;
            __syncthreads();
        }
        if (b < nb) block_sum[b] = carry + part[threadIdx.x] - v;      // exclusive offset of the block
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = carry;
}
__global__ __launch_bounds__(1024) void live_rows_fill_kernel(const int32_t* __restrict__ rank,
                                                              const int32_t* __restrict__ block_off,
                                                              const int32_t* __restrict__ count, long long n_tok,
                                                              int32_t* __restrict__ live) {
    const long long t = (long long)blockIdx.x * 1024 + threadIdx.x;
    if (t >= n_tok) return;
    if (t >= *count) live[t] = -1;                 // tail padding (positions < count are written by their owners)
    const int r = rank[t];
    if (r >= 0) live[block_off[blockIdx.x] + r] = (int32_t)t;
}
extern "C" int64_t dsvg_live_rows_workspace_bytes(int64_t n_tok) {
    return (n_tok + (n_tok + 1023) / 1024) * (int64_t)sizeof(int32_t);
}
extern "C" int dsvg_live_rows(const float* w, int64_t n_tok, int32_t group, int32_t* live, int32_t* count,
                              int32_t* workspace, int64_t workspace_bytes, void* stream) {
    DSVG_CHECK_ARG(w && live && count && n_tok > 0 && group > 0 && n_tok < (1ll << 31), "live_rows: bad args");
    DSVG_CHECK_ARG(workspace && workspace_bytes >= dsvg_live_rows_workspace_bytes(n_tok), "live_rows: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)((n_tok + 1023) / 1024);
    int32_t* rank = workspace;
    int32_t* block_sum = workspace + n_tok;
    hipLaunchKernelGGL(live_rows_rank_kernel, dim3(nb), dim3(1024), 0, st, w, (long long)n_tok, group, rank, block_sum);
    hipLaunchKernelGGL(live_rows_scan_kernel, dim3(1), dim3(1024), 0, st, block_sum, nb, count);
    hipLaunchKernelGGL(live_rows_fill_kernel, dim3(nb), dim3(1024), 0, st, rank, block_sum, count, (long long)n_tok, live);
    DSVG_LAUNCH_CHECK("live_rows");
    return 0;
}

// dst[idx[i], :] (+)= src[i, :] for idx[i] >= 0 (rows of dst not named by idx keep their content; idx has no duplicates)
template <typename T, bool ACC>
__global__ void scatter_rows_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, T* __restrict__ dst,
                                    long long n_rows, int width) {
    typedef typename Elem<T>::raw4 raw4;
    const int cpr = width / 4;
    const long long total = n_rows * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / cpr;
        const int c = (int)(i - row * cpr);
        const int t = idx[row];
        if (t < 0) continue;
        if (ACC) {
            float a[4], b[4];
            Elem<T>::ld4(src + row * width + 4 * c, a);
            Elem<T>::ld4(dst + (long long)t * width + 4 * c, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += b[e];
            Elem<T>::st4(dst + (long long)t * width + 4 * c, a);
        } else {
            reinterpret_cast<raw4*>(dst + (long long)t * width)[c] = reinterpret_cast<const raw4*>(src + row * width)[c];
        }
    }
}
extern "C" int dsvg_scatter_rows(int32_t dtype, const void* src, const int32_t* idx, void* dst, int64_t n_rows,
                                 int32_t width, int32_t accumulate, void* stream) {
    DSVG_CHECK_ARG(src && idx && dst && n_rows > 0 && width > 0 && (width % 4) == 0, "scatter_rows: bad args");
    const long long total = n_rows * (long long)(width / 4);
    const int nb = (int)min((long long)dsvg_cdiv(total, 256), 8192LL);
    hipStream_t st = (hipStream_t)stream;
#define DSVG_SR(T, A) hipLaunchKernelGGL((scatter_rows_kernel<T, A>), dim3(nb), dim3(256), 0, st, (const T*)src, idx, (T*)dst, \
                                         (long long)n_rows, width)
    if (dtype == DSVG_F32) { if (accumulate) DSVG_SR(float, true); else DSVG_SR(float, false); }
    else if (dtype == DSVG_BF16) { if (accumulate) DSVG_SR(bf16_t, true); else DSVG_SR(bf16_t, false); }
    else { dsvg_set_error("scatter_rows: bad dtype"); return -1; }
#undef DSVG_SR
    DSVG_LAUNCH_CHECK("scatter_rows");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// loss terms and their weighted total from the (sum, count) pairs of up to 4 cross-entropies, and the backward of that:
//     out[1 + i] = sum_i / count_i,   out[0] = sum_i w_i * out[1 + i]              (deepsvg/model/loss.py:43-57)
//     dsc[i] = { dout0 * w_i + dterm_i, 0 }   (the masked-CE backward divides by the count itself)
// One launch each instead of a dozen scalar elementwise launches of 5 us.
// ---------------------------------------------------------------------------------------------------------------------
struct LossCombineArgs {
    const float* sc[4];
    const float* dterm[4];
    float w[4];
    int n;
};
__global__ void loss_combine_fwd_kernel(const LossCombineArgs a, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    for (int i = 0; i < a.n; ++i) {
        const float l = a.sc[i][0] / a.sc[i][1];
        out[1 + i] = l;
        total += a.w[i] * l;
    }
    out[0] = total;
}
__global__ void loss_combine_bwd_kernel(const LossCombineArgs a, const float* __restrict__ dtotal, float* __restrict__ dsc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float g = dtotal ? *dtotal : 0.f;
    for (int i = 0; i < a.n; ++i) {
        dsc[2 * i] = g * a.w[i] + (a.dterm[i] ? *a.dterm[i] : 0.f);
        dsc[2 * i + 1] = 0.f;
    }
}

extern "C" int dsvg_loss_combine_fwd(const float* const* sum_count, const float* weights, int32_t n, float* out, void* stream) {
    DSVG_CHECK_ARG(sum_count && weights && out && n >= 1 && n <= 4, "loss_combine_fwd: bad args");
    LossCombineArgs a;
    a.n = n;
    for (int i = 0; i < 4; ++i) { a.sc[i] = i < n ? sum_count[i] : nullptr; a.dterm[i] = nullptr; a.w[i] = i < n ? weights[i] : 0.f; }
    for (int i = 0; i < n; ++i) DSVG_CHECK_ARG(a.sc[i], "loss_combine_fwd: null (sum, count) pair %d", i);
    hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, out);
    DSVG_LAUNCH_CHECK("loss_combine_fwd");
    return 0;
}

extern "C" int dsvg_loss_combine_bwd(const float* dtotal, const float* const* dterms, const float* weights, int32_t n,
                                     float* dsum_count, void* stream) {
    DSVG_CHECK_ARG(weights && dsum_count && n >= 1 && n <= 4, "loss_combine_bwd: bad args");
    LossCombineArgs a;
    a.n = n;
    for (int i = 0; i < 4; ++i) { a.sc[i] = nullptr; a.dterm[i] = (dterms && i < n) ? dterms[i] : nullptr; a.w[i] = i < n ? weights[i] : 0.f; }
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, dtotal, dsum_count);
    DSVG_LAUNCH_CHECK("loss_combine_bwd");
    return 0;
}
