// Mask construction, embedding gather/scatter, positional add, masked mean-pool and the broadcast add
// of linear_global(z): the HBM-bound glue of the DeepSVG encoder/decoder (reference call sites in
// include/dsvg.h).  All kernels are thread-per-feature-column with coalesced row accesses; parameter
// gradients are accumulated privately (registers / LDS columns owned by one thread) and reduced through
// per-workgroup partial rows -> no global atomics anywhere, bit-reproducible results.
#include "dsvg_common.h"
#include "../../include/dsvg.h"

// ---------------------------------------------------------------------------------------------
// masks (deepsvg/model/utils.py:7-66)
// ---------------------------------------------------------------------------------------------
__global__ void build_masks_kernel(const float* __restrict__ commands, long long n_seq, int S, int G, int eos,
                                   uint64_t* __restrict__ key_mask, int* __restrict__ seq_visible,
                                   uint64_t* __restrict__ group_mask) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_seq) return;
    const float* row = commands + b * S;
    uint64_t km = 0;
    int n_eos = 0;
    bool seen = false;
    for (int s = 0; s < S; ++s) {
        const bool is_eos = ((int)row[s] == eos);
        seen |= is_eos;
        n_eos += is_eos;
        if (!seen) km |= (1ull << s);
    }
    if (key_mask) key_mask[b] = km;
    if (seq_visible) seq_visible[b] = (n_eos < S - 1) ? 1 : 0;
}
__global__ void group_mask_kernel(const int* __restrict__ seq_visible, long long n_icons, int G,
                                  uint64_t* __restrict__ group_mask) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_icons) return;
    uint64_t gm = 0;
    for (int g = 0; g < G; ++g)
        if (seq_visible[n * G + g]) gm |= (1ull << g);
    group_mask[n] = gm;
}

extern "C" int dsvg_build_masks(const float* commands, int64_t n_seq, int32_t S, int32_t G, int32_t eos_id,
                                uint64_t* key_mask, int32_t* seq_visible, uint64_t* group_mask, void* stream) {
    DSVG_CHECK_ARG(commands && n_seq > 0 && S > 0 && S <= 64, "build_masks: bad args (S=%d)", S);
    DSVG_CHECK_ARG(!group_mask || (seq_visible && G > 0 && G <= 64 && n_seq % G == 0), "build_masks: bad group args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(build_masks_kernel, dim3(dsvg_cdiv(n_seq, 256)), dim3(256), 0, st, commands, (long long)n_seq, S,
                       G, eos_id, key_mask, seq_visible, group_mask);
    DSVG_LAUNCH_CHECK("build_masks");
    if (group_mask) {
        hipLaunchKernelGGL(group_mask_kernel, dim3(dsvg_cdiv(n_seq / G, 256)), dim3(256), 0, st, seq_visible,
                           (long long)(n_seq / G), G, group_mask);
        DSVG_LAUNCH_CHECK("group_mask");
    }
    return 0;
}

__global__ void group_index_kernel(const float* __restrict__ commands, long long n_seq, int S, int m_id,
                                   int* __restrict__ groups) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_seq) return;
    int cnt = 0;
    for (int s = 0; s < S; ++s) {
        cnt += ((int)commands[b * S + s] == m_id);
        groups[b * S + s] = cnt;
    }
}
extern "C" int dsvg_group_index(const float* commands, int64_t n_seq, int32_t S, int32_t m_id, int32_t* groups,
                                void* stream) {
    DSVG_CHECK_ARG(commands && groups && n_seq > 0 && S > 0, "group_index: bad args");
    hipLaunchKernelGGL(group_index_kernel, dim3(dsvg_cdiv(n_seq, 256)), dim3(256), 0, (hipStream_t)stream, commands,
                       (long long)n_seq, S, m_id, groups);
    DSVG_LAUNCH_CHECK("group_index");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Visible-first ordering of the second decoder stage.  The sequences of that stage are independent (one per group,
// model.py:250-262) and the loss excludes every position of an invisible target group (loss.py:36,51-54), so the
// backward pass of an invisible group's sequence is identically zero: with the visible sequences first, backward runs
// on a row prefix only.  new_of_old[b] = position of sequence b in the new order (stable partition), old_of_new = its
// inverse, *n_visible = number of visible sequences.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void visible_first_kernel(const int32_t* __restrict__ visible, long long n,
                                                             int32_t* __restrict__ new_of_old,
                                                             int32_t* __restrict__ old_of_new,
                                                             int32_t* __restrict__ n_visible) {
    __shared__ int part[1024];
    __shared__ int carry, total;
    // pass 0 counts the visible sequences, pass 1 assigns positions
    for (int pass = 0; pass < 2; ++pass) {
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (long long base = 0; base < n; base += 1024) {
            const long long b = base + threadIdx.x;
            const int v = (b < n && visible[b]) ? 1 : 0;
            part[threadIdx.x] = v;
            __syncthreads();
            for (int o = 1; o < 1024; o <<= 1) {
                const int t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
                __syncthreads();
                part[threadIdx.x] += t;
                __syncthreads();
            }
            if (pass == 1 && b < n) {
                const int vis_before = carry + part[threadIdx.x] - v;
                const int pos = v ? vis_before : total + (int)(b - vis_before);
                new_of_old[b] = pos;
                old_of_new[pos] = (int)b;
            }
            __syncthreads();
            if (threadIdx.x == 1023) carry += part[1023];
            __syncthreads();
        }
        if (pass == 0) {
            if (threadIdx.x == 0) { total = carry; *n_visible = carry; }
            __syncthreads();
        }
    }
}
extern "C" int dsvg_visible_first(const int32_t* visible, int64_t n, int32_t* new_of_old, int32_t* old_of_new,
                                  int32_t* n_visible, void* stream) {
    DSVG_CHECK_ARG(visible && new_of_old && old_of_new && n_visible && n > 0, "visible_first: bad args");
    hipLaunchKernelGGL(visible_first_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, visible, (long long)n,
                       new_of_old, old_of_new, n_visible);
    DSVG_LAUNCH_CHECK("visible_first");
    return 0;
}

// dst[g * S + s, :] = src[max(idx[g], 0) * S + s, :] for g < n_groups  (whole-sequence row gather; 16-byte pieces;
// a negative index marks list padding and reads group 0; an index >= n_src - a group the source does not hold - gives a
// zero row)
template <typename T>
__global__ void gather_groups_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, T* __restrict__ dst,
                                     long long n_groups, int S, int width, long long n_src) {
    typedef typename Elem<T>::raw4 raw4;
    const int cpr = width / 4;
    const long long total = n_groups * S * cpr;
    raw4 zero;
    __builtin_memset(&zero, 0, sizeof(zero));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / cpr;
        const int c = (int)(i - row * cpr);
        const long long g = row / S;
        const int sidx = (int)(row - g * S);
        const long long sg = max(idx[g], 0);
        reinterpret_cast<raw4*>(dst + row * width)[c] =
            sg < n_src ? reinterpret_cast<const raw4*>(src + (sg * S + sidx) * width)[c] : zero;
    }
}
extern "C" int dsvg_gather_groups(int32_t dtype, const void* src, const int32_t* idx, void* dst, int64_t n_groups,
                                  int32_t S, int32_t width, int64_t n_src, void* stream) {
    DSVG_CHECK_ARG(src && idx && dst && n_groups > 0 && S > 0 && width > 0 && (width % 4) == 0, "gather_groups: bad args");
    DSVG_CHECK_ARG(n_src > 0, "gather_groups: n_src = number of groups in src (> 0)");
    const long long total = n_groups * S * (long long)(width / 4);
    const int nb = (int)min((long long)dsvg_cdiv(total, 256), 8192LL);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(gather_groups_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)src, idx, (float*)dst,
                           (long long)n_groups, S, width, (long long)n_src);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(gather_groups_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)src, idx,
                           (bf16_t*)dst, (long long)n_groups, S, width, (long long)n_src);
    else { dsvg_set_error("gather_groups: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("gather_groups");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Packed (variable-length) token layout for the first encoder stage.  Only keys are masked there
// (layers/functional.py:234-239) and padded query rows are dropped by the masked mean-pool (model.py:137), so rows
// past a sequence's first EOS influence neither an output nor a gradient: the encoder runs on the valid tokens only.
//   seq_off[b]  = number of valid tokens in sequences < b (exclusive scan of popcount(key_mask)), seq_off[n_seq] = total
//   row t of the packed buffers = token (b, s) with t = seq_off[b] + s, s < len_b
//   rows >= total (up to the capacity n_seq * S) replicate token 0: finite inputs, their gradients are exact zeros
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void seq_offsets_kernel(const uint64_t* __restrict__ key_mask, long long n_seq,
                                                           int32_t* __restrict__ seq_off) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long base = 0; base < n_seq; base += 1024) {
        const long long b = base + threadIdx.x;
        const int len = b < n_seq ? __popcll(key_mask[b]) : 0;
        part[threadIdx.x] = len;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {            // Hillis-Steele inclusive scan
            const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        if (b < n_seq) seq_off[b] = carry + part[threadIdx.x] - len;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) seq_off[n_seq] = carry;
}
__global__ void pack_tokens_kernel(const float* __restrict__ commands, const float* __restrict__ args,
                                   const int32_t* __restrict__ seq_off, long long n_seq, int S, int n_args,
                                   float* __restrict__ pcmd, float* __restrict__ pargs, int32_t* __restrict__ ppos) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // dense token index
    if (t >= n_seq * S) return;
    const long long b = t / S;
    const int s = (int)(t % S);
    const int total = seq_off[n_seq];
    if (s < seq_off[b + 1] - seq_off[b]) {
        const long long r = seq_off[b] + s;
        pcmd[r] = commands[t];
        ppos[r] = s;
        for (int a = 0; a < n_args; ++a) pargs[r * n_args + a] = args[t * n_args + a];
    }
    if (t >= total) {                       // pad rows of the packed buffers
        pcmd[t] = commands[0];
        ppos[t] = 0;
        for (int a = 0; a < n_args; ++a) pargs[t * n_args + a] = args[a];
    }
}
extern "C" int dsvg_pack_tokens(const float* commands, const float* args, const uint64_t* key_mask, int64_t n_seq,
                                int32_t S, int32_t n_args, int32_t* seq_off, float* packed_commands, float* packed_args,
                                int32_t* packed_pos, void* stream) {
    DSVG_CHECK_ARG(commands && args && key_mask && seq_off && packed_commands && packed_args && packed_pos,
                   "pack_tokens: null pointer");
    DSVG_CHECK_ARG(n_seq > 0 && S > 0 && S <= 64 && n_args > 0 && n_seq * S < (1ll << 31), "pack_tokens: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(1024), 0, st, key_mask, (long long)n_seq, seq_off);
    DSVG_LAUNCH_CHECK("seq_offsets");
    hipLaunchKernelGGL(pack_tokens_kernel, dim3(dsvg_cdiv(n_seq * S, 256)), dim3(256), 0, st, commands, args, seq_off,
                       (long long)n_seq, S, n_args, packed_commands, packed_args, packed_pos);
    DSVG_LAUNCH_CHECK("pack_tokens");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// embedding gather (deepsvg/model/model.py:49-53)
//   A[t, a*E + e] = arg_embed[args[t,a] + 1, e];   R[t, c] = command_embed[cmd[t], c] (+ group_embed[grp[t], c])
// one thread per 4 output elements
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_gather_kernel(const float* __restrict__ commands, const float* __restrict__ args,
                                    const float* __restrict__ command_embed, const float* __restrict__ arg_embed,
                                    const float* __restrict__ group_embed, const int* __restrict__ groups,
                                    T* __restrict__ A, T* __restrict__ R, long long T_tok, int n_args, int E, int d,
                                    int n_cmd, int n_argvals) {
    const int wa = n_args * E / 4, wr = d / 4;
    const long long total = T_tok * (wa + wr);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long t = idx / (wa + wr);
        const int c = (int)(idx % (wa + wr));
        float v[4];
        if (c < wa) {
            const int a = (4 * c) / E, e = (4 * c) % E;
            int iv = (int)args[t * n_args + a] + 1;
            iv = min(max(iv, 0), n_argvals - 1);
            const float4 w = *reinterpret_cast<const float4*>(arg_embed + (size_t)iv * E + e);
            v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
            Elem<T>::st4(A + t * (long long)(n_args * E) + 4 * c, v);
        } else {
            const int cc = 4 * (c - wa);
            int ic = (int)commands[t];
            ic = min(max(ic, 0), n_cmd - 1);
            float4 w = *reinterpret_cast<const float4*>(command_embed + (size_t)ic * d + cc);
            if (group_embed) {
                const float4 g = *reinterpret_cast<const float4*>(group_embed + (size_t)groups[t] * d + cc);
                w.x += g.x; w.y += g.y; w.z += g.z; w.w += g.w;
            }
            v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
            Elem<T>::st4(R + t * (long long)d + cc, v);
        }
    }
}

extern "C" int dsvg_embed_gather(int32_t dtype, const float* commands, const float* args, const float* command_embed,
                                 const float* arg_embed, const float* group_embed, const int32_t* groups, void* A,
                                 void* R, int64_t T_tok, int32_t n_args, int32_t E, int32_t d, int32_t n_cmd,
                                 int32_t n_argvals, void* stream) {
    DSVG_CHECK_ARG(commands && args && command_embed && arg_embed && A && R, "embed_gather: null pointer");
    DSVG_CHECK_ARG(T_tok > 0 && (E % 4) == 0 && (d % 4) == 0, "embed_gather: bad shape");
    DSVG_CHECK_ARG(!group_embed || groups, "embed_gather: group_embed needs groups");
    const long long total = T_tok * (long long)(n_args * E / 4 + d / 4);
    const int nb = (int)min((long long)dsvg_cdiv(total, 256), 8192LL);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(embed_gather_kernel<float>, dim3(nb), dim3(256), 0, st, commands, args, command_embed,
                           arg_embed, group_embed, groups, (float*)A, (float*)R, (long long)T_tok, n_args, E, d, n_cmd,
                           n_argvals);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(embed_gather_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, commands, args, command_embed,
                           arg_embed, group_embed, groups, (bf16_t*)A, (bf16_t*)R, (long long)T_tok, n_args, E, d,
                           n_cmd, n_argvals);
    else { dsvg_set_error("embed_gather: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("embed_gather");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// embedding scatter-add (autograd of the nn.Embedding lookups, aten::embedding_dense_backward)
// Each workgroup owns a contiguous chunk of tokens and a private LDS copy of the gradient table
// (arg table 257x64 fp32 = 66 KB; command/group tables (7+G)x256).  Gradient rows are streamed with
// coalesced loads over the flattened (token, column) range and added with LDS atomics (ds_add_f32);
// the per-workgroup tables are written to the workspace and reduced in a fixed order, so there are no
// global atomics.  (LDS float atomics make the sum order inside one workgroup schedule-dependent, i.e.
// reproducible to ~1e-7 relative, not bitwise.)
// ---------------------------------------------------------------------------------------------
constexpr int ES_TOK_PER_BLOCK = 128;

// Each thread owns up to 4 columns (c = tid + 256 k) of the [n_args * E] gradient row, i.e. fixed (arg slot, e) pairs;
// tokens are processed 4 at a time with all of their loads issued before the first LDS atomic (the loop is otherwise a
// chain of dependent ~2 us loads), no integer divisions inside the loop.  64 tokens per workgroup so that even the
// packed encoder (~40k tokens at 512 icons) fills the chip.
template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_arg_kernel(const float* __restrict__ args, const T* __restrict__ dA,
                                                                float* __restrict__ part, long long T_tok, int n_args,
                                                                int E, int n_argvals) {
    extern __shared__ float acc[];   // [n_argvals * E]
    const int tab = n_argvals * E;
    for (int i = threadIdx.x; i < tab; i += 256) acc[i] = 0.f;
    __syncthreads();
    const long long t0 = (long long)blockIdx.x * ES_TOK_PER_BLOCK;
    const long long t1 = min(T_tok, t0 + ES_TOK_PER_BLOCK);
    const int width = n_args * E;
    constexpr int KC = 4, TU = 8;
    int a_of[KC], e_of[KC];
    float acc0[KC] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = threadIdx.x + 256 * k;
        a_of[k] = c < width ? c / E : -1;
        e_of[k] = c < width ? c % E : 0;
    }
    for (int c = threadIdx.x + 256 * KC; c < width; c += 256) {      // rows wider than 1024 columns: plain path
        for (long long t = t0; t < t1; ++t) {
            const float g = Elem<T>::ld(dA + t * width + c);
            int iv = (int)args[t * n_args + c / E] + 1;
            iv = min(max(iv, 0), n_argvals - 1);
            if (g != 0.f) atomicAdd(&acc[iv * E + c % E], g);
        }
    }
    for (long long tb = t0; tb < t1; tb += TU) {
        float g[TU][KC];
        int iv[TU][KC];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const long long t = tb + u;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                // unconditional loads from clamped addresses, select afterwards (a guarded load is compiled as a branch
                // plus s_waitcnt vmcnt(0) per element: the loads would not overlap)
                const bool ok = t < t1 && a_of[k] >= 0;
                const long long tt = min(t, t1 - 1);
                const float gv = Elem<T>::ld(dA + tt * width + (a_of[k] >= 0 ? threadIdx.x + 256 * k : 0));
                const int av = (int)args[tt * n_args + max(a_of[k], 0)] + 1;
                g[u][k] = ok ? gv : 0.f;
                iv[u][k] = ok ? av : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int row = min(max(iv[u][k], 0), n_argvals - 1);
                // table row 0 (argument value -1 = "unused slot") receives ~60 % of all contributions: those are summed
                // in a register of the thread that owns the column (LDS float atomics run at a fraction of a lane per
                // clock) and added once at the end
                if (row == 0) acc0[k] += g[u][k];
                else if (g[u][k] != 0.f) atomicAdd(&acc[row * E + e_of[k]], g[u][k]);
            }
    }
#pragma unroll
    for (int k = 0; k < KC; ++k)
        if (a_of[k] >= 0 && acc0[k] != 0.f) atomicAdd(&acc[e_of[k]], acc0[k]);
    __syncthreads();
    float* dst = part + (size_t)blockIdx.x * tab;
    for (int i = threadIdx.x; i < tab; i += 256) dst[i] = acc[i];
}

// The same table gradient as ONE-HOT MATRIX PRODUCTS on the matrix cores (bf16 gradients, E = 64): per argument slot a,
//     dTable^T[e][v] += sum_t dA[t][a][e] * [arg(t, a) + 1 == v]
// i.e. A = a hardware-transposed 32-token x 32-column block of the staged gradient rows (ds_read_b64_tr_b16), B = the
// one-hot of the same 32 tokens' indices against this wave's 32 table rows, built in registers from the staged indices.
// A workgroup of 8 waves owns ES_TOK_PER_BLOCK tokens; wave w accumulates table rows 32 w .. 32 w + 31 (wave 0 also the
// rows from 256 on), both 32-column halves: fp32 accumulation in a FIXED order - unlike the LDS float atomics of the
// kernel above, which also run at a fraction of a lane per clock (93 us for 41 k tokens; this one is bound by reading dA).
typedef __bf16 es_bf16x8 __attribute__((ext_vector_type(8)));
typedef short es_shortx4 __attribute__((ext_vector_type(4)));
typedef float es_floatx16 __attribute__((ext_vector_type(16)));
constexpr int ESM_LD = 11 * 64 + 8;         // row stride (elements) of the staged gradient rows: 1424 B = 89 x 16 B
__device__ __forceinline__ int es_rowmap(int r, int h2) { return (r & 3) + 8 * (r >> 2) + 4 * h2; }
__global__ __launch_bounds__(512) void embed_scatter_arg_mfma_kernel(const float* __restrict__ args,
                                                                     const bf16_t* __restrict__ dA, float* __restrict__ part,
                                                                     long long T_tok, int n_args, int n_argvals) {
    extern __shared__ __attribute__((aligned(16))) unsigned char es_smem[];
    bf16_t* img = reinterpret_cast<bf16_t*>(es_smem);                   // [32 tokens][ESM_LD]
    int* sidx = reinterpret_cast<int*>(es_smem + 32 * ESM_LD * 2);      // [32 tokens][16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h2 = lane >> 5;
    const int width = n_args * 64, cpr = width / 8;
    const long long t0 = (long long)blockIdx.x * ES_TOK_PER_BLOCK;
    const long long t1 = min(T_tok, t0 + ES_TOK_PER_BLOCK);
    const int n_vt = wave == 0 && n_argvals > 256 ? 2 : 1;             // value tiles of this wave: w (and 8 for wave 0)
    es_floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (long long tb = t0; tb < t1; tb += 32) {
        const int nt = (int)min(32LL, t1 - tb);
        // stage the chunk: gradient rows (rows past the end: zeros) and the +1-shifted, clamped indices
        for (int idx = threadIdx.x; idx < 32 * cpr; idx += 512) {
            const int r = idx / cpr, c = idx % cpr;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (r < nt) v = *reinterpret_cast<const uint4*>(dA + (tb + r) * width + 8 * c);
            *reinterpret_cast<uint4*>(img + r * ESM_LD + 8 * c) = v;
        }
        if (threadIdx.x < 32 * 16) {
            const int r = threadIdx.x >> 4, a = threadIdx.x & 15;
            int iv = -1;                                                // (no table row: padding tokens / slots)
            if (r < nt && a < n_args) iv = min(max((int)args[(tb + r) * n_args + a] + 1, 0), n_argvals - 1);
            sidx[r * 16 + a] = iv;
        }
        __syncthreads();
        for (int a = 0; a < n_args; ++a) {
            // the one-hot B fragments of this slot: lane (table row li of the tile, half h2), K slot e of step ks = token
            // row es_rowmap(8 ks + e, h2) - the K order of the transposed reads below
            es_bf16x8 oh[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                int tok_idx[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) tok_idx[e] = sidx[es_rowmap(8 * ks + e, h2) * 16 + a];
#pragma unroll
                for (int vt = 0; vt < 2; ++vt) {
                    if (vt >= n_vt) break;      // (wave-uniform)
                    const int v = (vt == 0 ? 32 * wave : 256) + li;
                    uint32_t w[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        w[e] = (tok_idx[2 * e] == v ? 0x3f80u : 0u) | (tok_idx[2 * e + 1] == v ? 0x3f800000u : 0u);
                    union { es_bf16x8 v8; uint4 u; } f;
                    f.u = make_uint4(w[0], w[1], w[2], w[3]);
                    oh[vt][ks] = f.v8;
                }
            }
#pragma unroll
            for (int ntile = 0; ntile < 2; ++ntile) {
                const int col0 = a * 64 + 32 * ntile;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    // A[i = column col0 + (lane & 31)][K slot e] = img[row es_rowmap(8 ks + e, lane >> 5)][col0 + i]
                    const int g = lane >> 4, q16 = lane & 15;
                    const int row = 16 * ks + 4 * (g >> 1) + (q16 >> 2);
                    const int col = col0 + 16 * (g & 1) + 4 * (q16 & 3);
                    union { es_bf16x8 v8; es_shortx4 h[2]; } fa;
                    fa.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (es_shortx4 __attribute__((address_space(3)))*)(&img[row * ESM_LD + col]));
                    fa.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (es_shortx4 __attribute__((address_space(3)))*)(&img[(row + 8) * ESM_LD + col]));
                    acc[0][ntile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v8, oh[0][ks], acc[0][ntile], 0, 0, 0);
                    if (n_vt == 2)
                        acc[1][ntile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v8, oh[1][ks], acc[1][ntile], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // acc[vt][ntile][r] = dTable[v = tile base + li][e = 32 ntile + es_rowmap(r, h2)]
    float* dst = part + (size_t)blockIdx.x * n_argvals * 64;
#pragma unroll
    for (int vt = 0; vt < 2; ++vt) {
        if (vt >= n_vt) break;
        const int v = (vt == 0 ? 32 * wave : 256) + li;
        if (v >= n_argvals) continue;
#pragma unroll
        for (int ntile = 0; ntile < 2; ++ntile)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)v * 64 + 32 * ntile + es_rowmap(r, h2)] = acc[vt][ntile][r];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_row_kernel(const float* __restrict__ commands,
                                                                const int* __restrict__ groups,
                                                                const T* __restrict__ dR, float* __restrict__ part_cmd,
                                                                float* __restrict__ part_grp, long long T_tok, int d,
                                                                int n_cmd, int n_groups, int g_lo, int g_win) {
    // one pass covers the group-table rows [g_lo, g_lo + g_win) (a long group table - the autoregressive decoder's has
    // max_total_len + 2 rows - does not fit LDS in one piece); the command table rides on the first pass
    extern __shared__ float acc[];   // [n_cmd * d] then [g_win * d]
    float* acc_c = acc;
    float* acc_g = acc + n_cmd * d;
    const bool do_cmd = g_lo == 0;
    for (int i = threadIdx.x; i < (n_cmd + g_win) * d; i += 256) acc[i] = 0.f;
    __syncthreads();
    const long long t0 = (long long)blockIdx.x * ES_TOK_PER_BLOCK;
    const long long t1 = min(T_tok, t0 + ES_TOK_PER_BLOCK);
    // tokens 8 at a time with all of their loads (command, group, gradient element) issued before the first table update:
    // one token per iteration was a chain of 128 dependent ~1 us global loads per workgroup
    constexpr int TU = 8;
    for (int c = threadIdx.x; c < d; c += 256) {
        for (long long tb = t0; tb < t1; tb += TU) {
            float g[TU];
            int ic[TU], ig[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const long long t = min(tb + u, t1 - 1);        // clamped address, selected below
                g[u] = Elem<T>::ld(dR + t * d + c);
                ic[u] = (int)commands[t];
                ig[u] = part_grp ? groups[t] - g_lo : -1;
            }
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                if (tb + u < t1 && g[u] != 0.f) {   // column c of every table row belongs to this thread alone: plain RMW
                    if (do_cmd) acc_c[min(max(ic[u], 0), n_cmd - 1) * d + c] += g[u];
                    if (ig[u] >= 0 && ig[u] < g_win) acc_g[ig[u] * d + c] += g[u];
                }
            }
        }
    }
    __syncthreads();
    if (do_cmd)
        for (int i = threadIdx.x; i < n_cmd * d; i += 256) part_cmd[(size_t)blockIdx.x * n_cmd * d + i] = acc_c[i];
    if (part_grp)
        for (int i = threadIdx.x; i < g_win * d; i += 256)
            part_grp[(size_t)blockIdx.x * n_groups * d + (size_t)g_lo * d + i] = acc_g[i];
}

extern "C" int64_t dsvg_embed_scatter_workspace_bytes(int64_t T_tok, int32_t n_args, int32_t E, int32_t d,
                                                      int32_t n_cmd, int32_t n_argvals, int32_t n_groups) {
    const int64_t nb = dsvg_cdiv(T_tok, ES_TOK_PER_BLOCK);
    return nb * ((int64_t)n_argvals * E + (int64_t)(n_cmd + n_groups) * d) * (int64_t)sizeof(float);
}

extern "C" int dsvg_embed_scatter(int32_t dtype, const float* commands, const float* args, const int32_t* groups,
                                  const void* dA, const void* dR, float* d_arg_embed, float* d_command_embed,
                                  float* d_group_embed, int64_t T_tok, int32_t n_args, int32_t E, int32_t d,
                                  int32_t n_cmd, int32_t n_argvals, int32_t n_groups, float* workspace,
                                  int64_t workspace_bytes, void* stream) {
    DSVG_CHECK_ARG(commands && args && dA && dR && d_arg_embed && d_command_embed, "embed_scatter: null pointer");
    DSVG_CHECK_ARG(!d_group_embed || (groups && n_groups > 0), "embed_scatter: group table needs groups");
    if (!d_group_embed) n_groups = 0;
    DSVG_CHECK_ARG(workspace && workspace_bytes >= dsvg_embed_scatter_workspace_bytes(T_tok, n_args, E, d, n_cmd,
                                                                                     n_argvals, n_groups),
                   "embed_scatter: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nb = dsvg_cdiv(T_tok, ES_TOK_PER_BLOCK);
    const int tab = n_argvals * E;
    float* part_arg = workspace;
    float* part_cmd = part_arg + (size_t)nb * tab;
    float* part_grp = n_groups ? part_cmd + (size_t)nb * n_cmd * d : nullptr;
    const size_t lds_a = (size_t)tab * sizeof(float);
    // group-table rows per pass: what fits 128 KiB next to the command table
    int g_cap = (int)((128 * 1024) / ((size_t)d * sizeof(float))) - n_cmd;
    DSVG_CHECK_ARG(lds_a <= 160 * 1024 && g_cap >= 1, "embed_scatter: tables too large for LDS");
    const int g_win0 = n_groups < g_cap ? n_groups : g_cap;
    const size_t lds_r = (size_t)(n_cmd + g_win0) * d * sizeof(float);
    if (dtype == DSVG_F32) {
        auto ka = embed_scatter_arg_kernel<float>;
        auto kr = embed_scatter_row_kernel<float>;
        DSVG_ENSURE_LDS(ka, lds_a);
        DSVG_ENSURE_LDS(kr, lds_r);
        hipLaunchKernelGGL(ka, dim3(nb), dim3(256), lds_a, st, args, (const float*)dA, part_arg, (long long)T_tok, n_args, E, n_argvals);
        for (int g_lo = 0; g_lo == 0 || g_lo < n_groups; g_lo += g_cap) {
            const int win = n_groups - g_lo < g_cap ? n_groups - g_lo : g_cap;
            hipLaunchKernelGGL(kr, dim3(nb), dim3(256), lds_r, st, commands, groups, (const float*)dR, part_cmd, part_grp,
                               (long long)T_tok, d, n_cmd, n_groups, g_lo, win > 0 ? win : 0);
        }
    } else if (dtype == DSVG_BF16) {
        auto ka = embed_scatter_arg_kernel<bf16_t>;
        auto kr = embed_scatter_row_kernel<bf16_t>;
        DSVG_ENSURE_LDS(ka, lds_a);
        DSVG_ENSURE_LDS(kr, lds_r);
        static const bool mfma_off = getenv("DSVG_EMBED_SCATTER_MFMA") && atoi(getenv("DSVG_EMBED_SCATTER_MFMA")) == 0;
        if (!mfma_off && E == 64 && n_args == 11 && n_argvals > 224 && n_argvals <= 288 && ((uintptr_t)dA & 15) == 0) {
            // (table rows 32 w .. of wave w; rows >= 256: wave 0's second tile - the tiles cover 288 rows)
            const size_t lds_m = (size_t)32 * ESM_LD * 2 + 32 * 16 * sizeof(int);
            DSVG_ENSURE_LDS(embed_scatter_arg_mfma_kernel, lds_m);
            hipLaunchKernelGGL(embed_scatter_arg_mfma_kernel, dim3(nb), dim3(512), lds_m, st, args, (const bf16_t*)dA, part_arg,
                               (long long)T_tok, n_args, n_argvals);
        } else
        hipLaunchKernelGGL(ka, dim3(nb), dim3(256), lds_a, st, args, (const bf16_t*)dA, part_arg, (long long)T_tok, n_args, E, n_argvals);
        for (int g_lo = 0; g_lo == 0 || g_lo < n_groups; g_lo += g_cap) {
            const int win = n_groups - g_lo < g_cap ? n_groups - g_lo : g_cap;
            hipLaunchKernelGGL(kr, dim3(nb), dim3(256), lds_r, st, commands, groups, (const bf16_t*)dR, part_cmd, part_grp,
                               (long long)T_tok, d, n_cmd, n_groups, g_lo, win > 0 ? win : 0);
        }
    } else { dsvg_set_error("embed_scatter: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("embed_scatter");
    int rc = dsvg_reduce_partials_strided(part_arg, nb, tab, tab, d_arg_embed, 0, st);
    if (rc) return rc;
    rc = dsvg_reduce_partials_strided(part_cmd, nb, (int64_t)n_cmd * d, (int64_t)n_cmd * d, d_command_embed, 0, st);
    if (rc) return rc;
    if (n_groups) rc = dsvg_reduce_partials_strided(part_grp, nb, (int64_t)n_groups * d, (int64_t)n_groups * d, d_group_embed, 0, st);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// y = drop(x + pos[t % S])      (positional_encoding.py:40-43, model.py:70-73)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void add_pos_fwd_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ y,
                                   long long n_tok, int S, int d, float drop_p, uint32_t site, const uint64_t* seed) {
    const DropCtx dc = drop_make(drop_p, seed, site);
    if ((d & 7) == 0) {         // 8 elements per thread: one group hash per 8 draws (drop_mult8) instead of one per element
        const int vpr8 = d / 8;
        const long long total8 = n_tok * vpr8;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total8;
             idx += (long long)gridDim.x * blockDim.x) {
            const long long t = idx / vpr8;
            const int c = 8 * (int)(idx % vpr8);
            const int s = (int)(t % S);
            float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, m[8];
            if (x) { Elem<T>::ld4(x + t * d + c, a); Elem<T>::ld4(x + t * d + c + 4, b); }
            const float4 p0 = *reinterpret_cast<const float4*>(pos + (size_t)s * d + c);
            const float4 p1 = *reinterpret_cast<const float4*>(pos + (size_t)s * d + c + 4);
            drop_mult8(dc, (uint64_t)t * d + c, m);
            a[0] = (a[0] + p0.x) * m[0]; a[1] = (a[1] + p0.y) * m[1]; a[2] = (a[2] + p0.z) * m[2]; a[3] = (a[3] + p0.w) * m[3];
            b[0] = (b[0] + p1.x) * m[4]; b[1] = (b[1] + p1.y) * m[5]; b[2] = (b[2] + p1.z) * m[6]; b[3] = (b[3] + p1.w) * m[7];
            Elem<T>::st4(y + t * d + c, a);
            Elem<T>::st4(y + t * d + c + 4, b);
        }
        return;
    }
    const int vpr = d / 4;
    const long long total = n_tok * vpr;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long t = idx / vpr;
        const int c = 4 * (int)(idx % vpr);
        const int s = (int)(t % S);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (x) Elem<T>::ld4(x + t * d + c, v);
        const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)s * d + c);
        const uint64_t e = (uint64_t)t * d + c;
        v[0] = (v[0] + pe.x) * drop_mult(dc, e);
        v[1] = (v[1] + pe.y) * drop_mult(dc, e + 1);
        v[2] = (v[2] + pe.z) * drop_mult(dc, e + 2);
        v[3] = (v[3] + pe.w) * drop_mult(dc, e + 3);
        Elem<T>::st4(y + t * d + c, v);
    }
}

// dx = dy * dropmask (elementwise, 4 elements per lane)
template <typename T>
__global__ void drop_copy_kernel(const T* __restrict__ dy, T* __restrict__ dx, long long n4, float drop_p,
                                 uint32_t site, const uint64_t* seed) {
    const DropCtx dc = drop_make(drop_p, seed, site);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float v[4];
        Elem<T>::ld4(dy + 4 * i, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= drop_mult(dc, (uint64_t)(4 * i + e));
        Elem<T>::st4(dx + 4 * i, v);
    }
}

// y = x * dropmask, 8 elements per lane (one hash group), element id = flat index.  Materialises the masked
// gradient once per dropout site so that the dX GEMM, the dW GEMM and the bias column-sum that all consume it
// read plain data instead of re-hashing it per output tile.
template <typename T>
__global__ void drop_apply8_kernel(const T* __restrict__ x, T* __restrict__ y, long long n8, float drop_p,
                                   uint32_t site, const uint64_t* seed) {
    const DropCtx dc = drop_make(drop_p, seed, site);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float a[4], b[4], m[8];
        Elem<T>::ld4(x + 8 * i, a);
        Elem<T>::ld4(x + 8 * i + 4, b);
        drop_mult8(dc, (uint64_t)(8 * i), m);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] *= m[e]; b[e] *= m[4 + e]; }
        Elem<T>::st4(y + 8 * i, a);
        Elem<T>::st4(y + 8 * i + 4, b);
    }
}
extern "C" int dsvg_drop_apply(int32_t dtype, const void* x, void* y, int64_t n, float drop_p, uint32_t drop_site,
                               const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(x && y && n > 0 && (n % 8) == 0, "drop_apply: n must be a positive multiple of 8");
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "drop_apply: dropout needs a seed pointer");
    DSVG_CHECK_ARG(!((uintptr_t)x & 15) && !((uintptr_t)y & 15), "drop_apply: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long long n8 = n / 8;
    const int nb = (int)min((long long)dsvg_cdiv(n8, 256), 8192LL);
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(drop_apply8_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, (float*)y, n8, drop_p,
                           drop_site, seed);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(drop_apply8_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n8,
                           drop_p, drop_site, seed);
    else { dsvg_set_error("drop_apply: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("drop_apply");
    return 0;
}

extern "C" int dsvg_add_pos_fwd(int32_t dtype, const void* x, const float* pos, void* y, int64_t n_seq, int32_t S,
                                int32_t d, float drop_p, uint32_t drop_site, const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(pos && y && n_seq > 0 && S > 0 && (d % 4) == 0, "add_pos_fwd: bad args");
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "add_pos_fwd: dropout needs a seed pointer");
    const long long total = n_seq * S * (long long)(d / 4);
    const int nb = (int)min((long long)dsvg_cdiv(total, 256), 8192LL);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(add_pos_fwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, pos, (float*)y,
                           (long long)(n_seq * S), S, d, drop_p, drop_site, seed);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(add_pos_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)x, pos, (bf16_t*)y,
                           (long long)(n_seq * S), S, d, drop_p, drop_site, seed);
    else { dsvg_set_error("add_pos_fwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("add_pos_fwd");
    return 0;
}

// d_pos[s, c] = sum_b (dy*mask)[b*S+s, c] is the column sum of dy viewed as [n_seq, S*d] (element ids coincide:
// b*(S*d) + s*d + c = t*d + c), so the generic column-sum kernel of gemm.hip does the reduction.
extern "C" int64_t dsvg_add_pos_bwd_workspace_bytes(int64_t n_seq, int32_t S, int32_t d) {
    return dsvg_colsum_workspace_bytes(n_seq, S * d);
}

extern "C" int dsvg_add_pos_bwd(int32_t dtype, const void* dy, void* dx, float* d_pos, int32_t accumulate,
                                int64_t n_seq, int32_t S, int32_t d, float drop_p, uint32_t drop_site,
                                const uint64_t* seed, float* workspace, int64_t workspace_bytes, void* stream) {
    DSVG_CHECK_ARG(dy && d_pos && n_seq > 0 && S > 0 && d > 0 && (d % 4) == 0, "add_pos_bwd: bad args");
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "add_pos_bwd: dropout needs a seed pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const long long n4 = n_seq * S * (long long)d / 4;
        const int nb = (int)min((long long)dsvg_cdiv(n4, 256), 8192LL);
        if (dtype == DSVG_F32)
            hipLaunchKernelGGL(drop_copy_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)dy, (float*)dx, n4,
                               drop_p, drop_site, seed);
        else if (dtype == DSVG_BF16)
            hipLaunchKernelGGL(drop_copy_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, n4,
                               drop_p, drop_site, seed);
        else { dsvg_set_error("add_pos_bwd: bad dtype"); return -1; }
        DSVG_LAUNCH_CHECK("add_pos_bwd(dx)");
    }
    return dsvg_colsum(dtype, dy, (int64_t)S * d, n_seq, S * d, d_pos, accumulate, drop_p, drop_site, seed, workspace,
                       workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// masked mean over the sequence axis (deepsvg/model/model.py:137,161); one block per sequence
// ---------------------------------------------------------------------------------------------
// packed layout (seq_off != null): sequence b owns rows [seq_off[b], seq_off[b+1]), all valid; the backward
// workgroup with blockIdx.x == n_seq zero-fills the pad rows [seq_off[n_seq], total_rows)
template <typename T>
__global__ void masked_mean_fwd_kernel(const T* __restrict__ x, const uint64_t* __restrict__ mask,
                                       const int32_t* __restrict__ seq_off, T* __restrict__ out, int S, int d) {
    const long long b = blockIdx.x;
    long long row0 = b * S;
    uint64_t m;
    if (seq_off) {
        row0 = seq_off[b];
        const int len = seq_off[b + 1] - seq_off[b];
        m = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
        S = len;
    } else {
        m = mask[b];
    }
    const float inv = 1.f / (float)__popcll(m);
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < S; ++i)
            if ((m >> i) & 1ull) s += Elem<T>::ld(x + (row0 + i) * d + c);
        Elem<T>::st(out + b * d + c, s * inv);
    }
}
template <typename T>
__global__ void masked_mean_bwd_kernel(const T* __restrict__ dout, const uint64_t* __restrict__ mask,
                                       const int32_t* __restrict__ seq_off, long long total_rows, T* __restrict__ dx,
                                       int S, int d) {
    const long long b = blockIdx.x;
    long long row0 = b * S;
    uint64_t m;
    if (seq_off) {
        if (b == (long long)gridDim.x - 1) {       // rows behind the last sequence (bucket padding) <- 0, 16 bytes per store
            const long long e0 = (long long)seq_off[b] * d, e1 = total_rows * d;
            constexpr int V = 16 / (int)sizeof(T);
            if (((uintptr_t)(dx + e0) & 15) == 0 && ((e1 - e0) % V) == 0) {
                uint4* z = reinterpret_cast<uint4*>(dx + e0);
                for (long long i = threadIdx.x; i < (e1 - e0) / V; i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                for (long long i = e0 + threadIdx.x; i < e1; i += blockDim.x) Elem<T>::st(dx + i, 0.f);
            }
            return;
        }
        row0 = seq_off[b];
        const int len = seq_off[b + 1] - seq_off[b];
        m = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
        S = len;
    } else {
        m = mask[b];
    }
    const float inv = 1.f / (float)__popcll(m);
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const float g = Elem<T>::ld(dout + b * d + c) * inv;
        for (int i = 0; i < S; ++i) Elem<T>::st(dx + (row0 + i) * d + c, ((m >> i) & 1ull) ? g : 0.f);
    }
}
// Round 6: the bf16 FORWARD kernel with a thread per (sequence, 8 columns): 16-byte loads instead of 2-byte ones (the one-column
// threads above ran the 41 k-row pooling at 1.2 TB/s: 18.2 -> 9.3 us).  Every column is still summed over its rows in increasing order:
// bit-identical.
__device__ __forceinline__ void mm_unpack8(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(w[e] << 16);
        v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ bool mm_seq(const uint64_t* mask, const int32_t* seq_off, long long b, int& S, long long& row0,
                                       uint64_t& m) {
    if (seq_off) {
        row0 = seq_off[b];
        const int len = seq_off[b + 1] - seq_off[b];
        m = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
        S = len;
    } else {
        row0 = b * S;
        m = mask[b];
    }
    return true;
}
__global__ __launch_bounds__(256) void masked_mean_fwd8_kernel(const bf16_t* __restrict__ x, const uint64_t* __restrict__ mask,
                                                               const int32_t* __restrict__ seq_off, bf16_t* __restrict__ out,
                                                               long long n_seq, int S, int d) {
    const int cpr = d >> 3;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = gid / cpr;
    if (b >= n_seq) return;
    const int c = (int)(gid % cpr) * 8;
    long long row0;
    uint64_t m;
    mm_seq(mask, seq_off, b, S, row0, m);
    const float inv = 1.f / (float)__popcll(m);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t* px = x + row0 * d + c;
    int i = 0;
    for (; i + 4 <= S; i += 4) {        // four rows in flight
        uint4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const uint4*>(px + (long long)(i + k) * d);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[8];
            mm_unpack8(t[k], v);
            if ((m >> (i + k)) & 1ull) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e];
            }
        }
    }
    for (; i < S; ++i) {
        if ((m >> i) & 1ull) {
            float v[8];
            mm_unpack8(*reinterpret_cast<const uint4*>(px + (long long)i * d), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
    }
    *reinterpret_cast<uint4*>(out + b * d + c) = make_uint4(f2bf_pk(s[0] * inv, s[1] * inv), f2bf_pk(s[2] * inv, s[3] * inv),
                                                            f2bf_pk(s[4] * inv, s[5] * inv), f2bf_pk(s[6] * inv, s[7] * inv));
}
static bool mm_vec_ok(int32_t dtype, const void* a, const void* b, int32_t d) {
    static const bool off = getenv("DSVG_MEAN_VEC") && atoi(getenv("DSVG_MEAN_VEC")) == 0;      // A/B knob
    return !off && dtype == DSVG_BF16 && (d % 8) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}

extern "C" int dsvg_masked_mean_fwd(int32_t dtype, const void* x, const uint64_t* mask, const int32_t* seq_off, void* out,
                                    int64_t n_seq, int32_t S, int32_t d, void* stream) {
    DSVG_CHECK_ARG(x && (mask || seq_off) && out && n_seq > 0 && S > 0 && S <= 64 && d > 0, "masked_mean_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (mm_vec_ok(dtype, x, out, d)) {
        const long long threads = (long long)n_seq * (d / 8);
        hipLaunchKernelGGL(masked_mean_fwd8_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const bf16_t*)x, mask,
                           seq_off, (bf16_t*)out, (long long)n_seq, S, d);
        DSVG_LAUNCH_CHECK("masked_mean_fwd (16-byte)");
        return 0;
    }
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(masked_mean_fwd_kernel<float>, dim3((unsigned)n_seq), dim3(256), 0, st, (const float*)x, mask,
                           seq_off, (float*)out, S, d);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(masked_mean_fwd_kernel<bf16_t>, dim3((unsigned)n_seq), dim3(256), 0, st, (const bf16_t*)x,
                           mask, seq_off, (bf16_t*)out, S, d);
    else { dsvg_set_error("masked_mean_fwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("masked_mean_fwd");
    return 0;
}
extern "C" int dsvg_masked_mean_bwd(int32_t dtype, const void* dout, const uint64_t* mask, const int32_t* seq_off,
                                    int64_t total_rows, void* dx, int64_t n_seq, int32_t S, int32_t d, void* stream) {
    DSVG_CHECK_ARG(dout && (mask || seq_off) && dx && n_seq > 0 && S > 0 && S <= 64 && d > 0, "masked_mean_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    // (a 16-byte variant of this kernel - a thread per sequence and 8 columns, or per row quarter of it - measured SLOWER than the
    // one-column threads below: 13.6 / 19.3 vs 9.8 us on the 41 k-row pooling; only the forward kernel was replaced)
    const unsigned nb = (unsigned)n_seq + (seq_off ? 1u : 0u);
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(masked_mean_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)dout, mask, seq_off,
                           (long long)total_rows, (float*)dx, S, d);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(masked_mean_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)dout, mask, seq_off,
                           (long long)total_rows, (bf16_t*)dx, S, d);
    else { dsvg_set_error("masked_mean_bwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("masked_mean_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// x[t] += drop(g)[t / S]   (improved_transformer.py:131-136), one block per sequence.  The reference applies the dropout
// to linear_global(memory) - one row per sequence - BEFORE the implicit broadcast, so a dropped (sequence, channel) is
// dropped at every position: the mask id is b * d + c, drawn once per thread, not per token.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void bcast_add_fwd_kernel(T* __restrict__ x, const T* __restrict__ g, int S, int d, float drop_p,
                                     uint32_t site, const uint64_t* seed) {
    const DropCtx dc = drop_make(drop_p, seed, site);
    const long long b = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const float gv = Elem<T>::ld(g + b * d + c) * drop_mult(dc, (uint64_t)b * d + c);
        T* px = x + b * S * d + c;
        for (int i = 0; i < S; ++i, px += d) Elem<T>::st(px, Elem<T>::ld(px) + gv);
    }
}
template <typename T>
__global__ void bcast_add_bwd_kernel(const T* __restrict__ dx, T* __restrict__ dg, long long n_seq, int S, int d,
                                     float drop_p, uint32_t site, const uint64_t* seed) {
    const long long b = blockIdx.x;
    if (b >= n_seq) {           // sequences past the live prefix: zero gradient
        for (int c = threadIdx.x; c < d; c += blockDim.x) Elem<T>::st(dg + b * d + c, 0.f);
        return;
    }
    const DropCtx dc = drop_make(drop_p, seed, site);
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float s = 0.f;
        const T* px = dx + b * S * d + c;
        for (int i = 0; i < S; ++i, px += d) s += Elem<T>::ld(px);
        Elem<T>::st(dg + b * d + c, s * drop_mult(dc, (uint64_t)b * d + c));
    }
}
// bf16 rows of d % 8 == 0 columns: 16-byte loads, 8 columns of one sequence per thread and BA_SPLIT threads per such piece
// (thread q takes the rows q, q + BA_SPLIT, ..: 2,304 sequences of 31 rows are 18 k pieces of 31 dependent-free loads - one
// thread per piece left the chip at one wave per SIMD and the launch at 9.4 us, bound by issue and latency); the partial
// sums meet in LDS in a fixed order.
// dxm (optional): the rows themselves with the mask of ANOTHER dropout site replayed on them (what dsvg_drop_apply(dx, site_m)
// writes) - the large decoder layers' backward needs both of dx1, and this kernel already reads every element of it once
constexpr int BA_SPLIT = 4;
__global__ __launch_bounds__(256) void bcast_add_bwd8_kernel(const bf16_t* __restrict__ dx, bf16_t* __restrict__ dg,
                                                             long long n_seq, long long n_seq_out, int S, int d,
                                                             float drop_p, uint32_t site, const uint64_t* seed,
                                                             bf16_t* __restrict__ dxm, uint32_t site_m, long long rows_m,
                                                             long long dg_ld) {
    // dg_ld: row stride (elements) of dg - d, or wider: dg is a column block of a longer row (ABI 9: the conditioning gradients of a
    // decoder stack's layers as the column blocks of ONE buffer, no concatenation launch)
    // rows_m (with dxm): dx / dxm have rows_m >= n_seq * S rows (a live row prefix rounded up): the rows past the summed
    // sequences are masked too, by the workgroups of the sequences they would belong to
    __shared__ float part[256][8];
    const int cpr = d / 8;                                  // pieces per row
    const int per = 256 / (cpr * BA_SPLIT);                 // sequences per workgroup (>= 1: d <= 512; see the host side)
    const int t = threadIdx.x;
    const int piece = t % cpr, q = (t / cpr) % BA_SPLIT, sl = t / (cpr * BA_SPLIT);
    const long long b = (long long)blockIdx.x * per + sl;
    const int c8 = piece * 8;
    const bool mine = sl < per && b < n_seq_out;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const bool summing = b < n_seq;
    if (mine && (summing || (dxm && b * S < rows_m))) {
        const DropCtx dm_ctx = drop_make(dxm ? drop_p : 0.f, seed, site_m);
        const bf16_t* px = dx + (b * S) * d + c8;
        // the masked copy of one row piece: ids (row * d + c8) .. + 7 are one aligned group of the standard draws
        auto put_masked = [&](long long row, const uint32_t (&w)[4]) {
            if (!dxm) return;
            float v[8], m[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
            drop_mult8(dm_ctx, (uint64_t)row * d + c8, m);
            *reinterpret_cast<uint4*>(dxm + row * d + c8) =
                make_uint4(f2bf_pk(v[0] * m[0], v[1] * m[1]), f2bf_pk(v[2] * m[2], v[3] * m[3]),
                           f2bf_pk(v[4] * m[4], v[5] * m[5]), f2bf_pk(v[6] * m[6], v[7] * m[7]));
        };
        // rows of this sequence that exist: all S of a summed sequence, those below rows_m of a masked-only one
        const int S_here = summing ? S : (int)min((long long)S, rows_m - b * S);
        for (int i = q; i < S_here; i += 4 * BA_SPLIT) {     // 4 rows of this thread in flight
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * BA_SPLIT < S_here) v[u] = *reinterpret_cast<const uint4*>(px + (long long)(i + u * BA_SPLIT) * d);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i + u * BA_SPLIT >= S_here) break;
                const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                if (summing) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[2 * e] += __uint_as_float(w[e] << 16);
                        s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                    }
                }
                put_masked(b * S + i + u * BA_SPLIT, w);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[t][e] = s[e];
    __syncthreads();
    if (!mine || q != 0) return;
#pragma unroll
    for (int k = 1; k < BA_SPLIT; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += part[t + k * cpr][e];
    if (b < n_seq) {
        const DropCtx dc = drop_make(drop_p, seed, site);
        if (dc.on) {
            float dm[8];
            drop_mult8(dc, (uint64_t)b * d + c8, dm);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] *= dm[e];
        }
    }
    *reinterpret_cast<uint4*>(dg + b * dg_ld + c8) =
        make_uint4(f2bf_pk(s[0], s[1]), f2bf_pk(s[2], s[3]), f2bf_pk(s[4], s[5]), f2bf_pk(s[6], s[7]));
}
extern "C" int dsvg_bcast_add_fwd(int32_t dtype, void* x, const void* g, int64_t n_seq, int32_t S, int32_t d,
                                  float drop_p, uint32_t drop_site, const uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(x && g && n_seq > 0 && S > 0 && d > 0, "bcast_add_fwd: bad args");
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "bcast_add_fwd: dropout needs a seed pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(bcast_add_fwd_kernel<float>, dim3((unsigned)n_seq), dim3(256), 0, st, (float*)x,
                           (const float*)g, S, d, drop_p, drop_site, seed);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(bcast_add_fwd_kernel<bf16_t>, dim3((unsigned)n_seq), dim3(256), 0, st, (bf16_t*)x,
                           (const bf16_t*)g, S, d, drop_p, drop_site, seed);
    else { dsvg_set_error("bcast_add_fwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("bcast_add_fwd");
    return 0;
}
extern "C" int dsvg_bcast_add_bwd(int32_t dtype, const void* dx, void* dg, int64_t n_seq, int64_t n_seq_out, int32_t S,
                                  int32_t d, float drop_p, uint32_t drop_site, const uint64_t* seed, int64_t dg_ld, void* stream) {
    DSVG_CHECK_ARG(dx && dg && n_seq > 0 && n_seq_out >= n_seq && S > 0 && d > 0, "bcast_add_bwd: bad args");
    DSVG_CHECK_ARG(dg_ld == d || (dtype == DSVG_BF16 && dg_ld > d && (dg_ld % 8) == 0 && (d % 8) == 0 && d <= 512 &&
                                  (((uintptr_t)dx | (uintptr_t)dg) & 15) == 0),
                   "bcast_add_bwd: a strided dg needs the bf16 kernel's conditions (d %% 8 == 0, d <= 512, 16-byte alignment)");
    DSVG_CHECK_ARG(drop_p <= 0.f || seed, "bcast_add_bwd: dropout needs a seed pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(bcast_add_bwd_kernel<float>, dim3((unsigned)n_seq_out), dim3(256), 0, st, (const float*)dx,
                           (float*)dg, (long long)n_seq, S, d, drop_p, drop_site, seed);
    else if (dtype == DSVG_BF16 && (d % 8) == 0 && d <= 512 && (((uintptr_t)dx | (uintptr_t)dg) & 15) == 0) {
        const int per = 256 / ((d / 8) * BA_SPLIT);
        hipLaunchKernelGGL(bcast_add_bwd8_kernel, dim3((unsigned)((n_seq_out + per - 1) / per)), dim3(256), 0, st,
                           (const bf16_t*)dx, (bf16_t*)dg, (long long)n_seq, (long long)n_seq_out, S, d, drop_p, drop_site,
                           seed, (bf16_t*)nullptr, 0u, 0ll, (long long)dg_ld);
    } else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(bcast_add_bwd_kernel<bf16_t>, dim3((unsigned)n_seq_out), dim3(256), 0, st, (const bf16_t*)dx,
                           (bf16_t*)dg, (long long)n_seq, S, d, drop_p, drop_site, seed);
    else { dsvg_set_error("bcast_add_bwd: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("bcast_add_bwd");
    return 0;
}

extern "C" int dsvg_bcast_add_bwd_masked(const void* dx, void* dg, void* dx_masked, int64_t n_seq, int64_t n_seq_out, int32_t S,
                                         int32_t d, int64_t rows, float drop_p, uint32_t drop_site, uint32_t mask_site,
                                         const uint64_t* seed, int64_t dg_ld, void* stream) {
    DSVG_CHECK_ARG(dg_ld >= d && (dg_ld % 8) == 0, "bcast_add_bwd_masked: bad dg row stride %lld", (long long)dg_ld);
    DSVG_CHECK_ARG(dx && dg && dx_masked && n_seq > 0 && n_seq_out >= n_seq && S > 0 && d > 0, "bcast_add_bwd_masked: bad args");
    DSVG_CHECK_ARG(rows >= n_seq * S && rows <= n_seq_out * S, "bcast_add_bwd_masked: rows must lie in [n_seq * S, n_seq_out * S]");
    DSVG_CHECK_ARG(drop_p > 0.f && seed, "bcast_add_bwd_masked: needs dropout and a seed pointer (use dsvg_bcast_add_bwd otherwise)");
    DSVG_CHECK_ARG((d % 8) == 0 && d <= 512 && (((uintptr_t)dx | (uintptr_t)dg | (uintptr_t)dx_masked) & 15) == 0,
                   "bcast_add_bwd_masked: bf16 rows of d % 8 == 0 <= 512 columns, 16-byte aligned");
    const int per = 256 / ((d / 8) * BA_SPLIT);
    hipLaunchKernelGGL(bcast_add_bwd8_kernel, dim3((unsigned)((n_seq_out + per - 1) / per)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dx, (bf16_t*)dg, (long long)n_seq, (long long)n_seq_out, S, d, drop_p, drop_site, seed,
                       (bf16_t*)dx_masked, mask_site, (long long)rows, (long long)dg_ld);
    DSVG_LAUNCH_CHECK("bcast_add_bwd_masked");
    return 0;
}
