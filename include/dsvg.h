/*
 * dsvg.h — C ABI of libdsvg_hip.so: the gfx950 (MI355X) kernels behind the DeepSVG
 * SVGTransformer forward/backward hot path.
 *
 * The reference (alexandre01/deepsvg) has no FFI for this path: every op is a stock aten op
 * reached through torch.nn / torch.nn.functional.  Each entry point below therefore cites the
 * reference call site (file:line under /root/reference) whose arithmetic it replaces.  The
 * Python host (deepsvg_amd/ops.py) binds these with ctypes; see INTEGRATION.md.
 *
 * Conventions
 *  - Every buffer is owned by the caller (PyTorch); the library allocates nothing persistent.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*); no implicit syncs.
 *  - Return 0 on success, <0 on error; dsvg_last_error() returns a thread-local message.
 *  - dtype: DSVG_F32 (parity path, exact-fp32 MFMA) or DSVG_BF16 (bf16 storage, fp32 accumulate).
 *  - "seed" is a device pointer to a uint64 so a captured hipGraph re-reads it on every replay.
 *  - Token-major activations: row index t = sequence * S + position; feature index contiguous.
 */
#ifndef DSVG_H
#define DSVG_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSVG_F32 0
#define DSVG_BF16 1

/* ABI version: bumped on EVERY change of an exported signature (round 4: 2 - dsvg_ffn_bwd_one and
 * dsvg_attn_block_fwd_stages removed, round 3's signature changes of dsvg_defer_scope / dsvg_gather_groups /
 * dsvg_bcast_add_bwd / dsvg_loss_targets / dsvg_scatter_rows counted).  dsvg_version() returns the value the library was
 * built with; a caller compiled against another header must refuse to run (deepsvg_amd/lib.py does).
 * 3: dsvg_latent_chain_fwd / dsvg_latent_chain_bwd added.  4: seq_add_ld argument of dsvg_attn_block_fwd / dsvg_gs_layer_fwd.
 * 5 (round 5): dsvg_sample_rows / dsvg_head_sample (categorical sampling on the device), dsvg_layernorm_bwd_masked added.
 * 6: dsvg_pack_images, dsvg_defer_zero added; dg argument of dsvg_gs_layer_bwd.
 * 7 (round 6): dsvg_attn_bwd_dx added; a layer of dsvg_attn_pack_bwd grew from 128 to 512 fragments (in_proj_weight^T behind
 *    out_proj.weight^T).
 * 8 (round 6): dsvg_gs_stack_fwd / dsvg_gs_stack_bwd added (one launch per STACK of group-stage layers).
 * 9 (round 6): dg_ld argument of dsvg_bcast_add_bwd / dsvg_bcast_add_bwd_masked (dg as a column block of a wider buffer). */
#define DSVG_ABI_VERSION 9

const char* dsvg_last_error(void);
int dsvg_version(void);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused prologue/epilogue:  C = epi( sum_k A(m,k) * B(n,k) )
 *   epi(v) = [C +] [res +] drop( gate( act( v + bias [+ res if res_pre] ) ) )
 * Replaces aten::addmm / aten::mm of every nn.Linear on the path:
 *   deepsvg/model/layers/functional.py:92,249 (QKV, out-proj),
 *   deepsvg/model/layers/improved_transformer.py:52,131,139 (FFN, linear_global),
 *   deepsvg/model/model.py:50,183,197 (embed_fcn, VAE, bottleneck),
 *   deepsvg/model/basic_blocks.py:18,20,36-37,60-63 (heads, ResNet),
 * and the two backward matmuls autograd derives for each of them (deepsvg/train.py:98).
 * ------------------------------------------------------------------------------------------ */
typedef struct dsvg_gemm_desc {
    int32_t dtype;            /* element type of A, B, res, gate and (unless c_f32) C             */
    int32_t M, N, K;
    const void* A; int64_t lda; int32_t a_kc;   /* a_kc=1: A(m,k)=A[m*lda+k]  0: A[k*lda+m]      */
    const void* B; int64_t ldb; int32_t b_kc;   /* b_kc=1: B(n,k)=B[n*ldb+k]  0: B[k*ldb+n]      */
    void* C; int64_t ldc; int32_t c_f32;        /* C[m*ldc+n]; c_f32=1 forces fp32 output        */
    const float* bias;                          /* [N] fp32 or NULL                              */
    const void* res; int64_t ldres; int32_t res_pre; /* residual (may alias C)                   */
    int32_t act;                                /* 0 none, 1 relu                                */
    const void* gate; int64_t ldgate; float gate_scale; /* v *= gate_scale*(gate[m,n]>0)         */
    float drop_p; uint32_t drop_site;           /* epilogue dropout, element id = m*N+n          */
    float a_drop_p; uint32_t a_drop_site; int64_t a_drop_ld; /* dropout replay on operand A,
                                                   element id = storage_row*a_drop_ld+storage_col */
    const uint64_t* seed;                       /* device pointer (may be NULL when p==0)        */
    int32_t accumulate;                         /* C += result                                   */
    int32_t split_k; float* workspace; int64_t workspace_bytes; /* split over K (weight grads);
                                                   needs c_f32 output and no epilogue            */
    float* rowsum;                              /* optional fp32 [M]: rowsum[m] = sum_k A(m,k), i.e.
                                                   the bias gradient for free inside the weight-
                                                   gradient GEMM (only with split_k > 1)         */
    int32_t impl;                               /* 0 = best MFMA kernel, 1 = one-thread-per-output,
                                                   2 = register-staged MFMA kernel only, 3 / 4 = LDS-DMA
                                                   kernel with 2 / 1 LDS stages when eligible, 5 = weight-
                                                   stationary kernel when eligible, 6 = LDS-DMA kernel with
                                                   4 stages (asm DMA three K steps ahead; the default for
                                                   launches of <= 256 workgroups) (test knobs)      */
} dsvg_gemm_desc;

int dsvg_gemm(const dsvg_gemm_desc* d, void* stream);
/* workspace bytes dsvg_gemm needs for a given split_k (0 when split_k<=1) */
int64_t dsvg_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k);

/* out[j] = (accumulate? out[j] : 0) + sum_{p<P} partial[p*n+j]   (fp32; deterministic order) */
int dsvg_reduce_partials(const float* partial, int64_t P, int64_t n, float* out, int32_t accumulate,
                         void* stream);

/* Deferred parameter-gradient reductions.  While a scope is open ON A STREAM (dsvg_defer_scope(1, s) ...
 * dsvg_defer_scope(0, s)) every partial-sum reduction this library is asked to launch on that stream (split-K slices of dsvg_gemm, the gamma/beta partials of
 * dsvg_layernorm_bwd, dsvg_colsum, dsvg_add_pos_bwd, dsvg_embed_scatter) whose data allows 16-byte accesses is queued
 * instead of launched: its destination is NOT valid and its workspace must stay untouched until dsvg_flush_deferred,
 * which performs all queued reductions in one launch per 64 of them (a backward pass of the benchmark model queues
 * ~130: the per-reduction launches were 0.85 ms of a 10.5 ms step).  A reduction whose destination overlaps a queued
 * one flushes the queue first, so write-after-write order is kept; reads of a queued destination are the caller's
 * responsibility (deepsvg_amd/trainer.py flushes right after loss.backward(), the reference's train.py:98).
 * The queue is keyed by the stream (hence by device): launches on other streams - another model, another trainer,
 * nn.DataParallel's per-device threads - are neither queued nor flushed by it, and the flush runs on the stream that
 * produced the queued partials, ordered behind them.  No state besides these per-stream queues is kept.
 * dsvg_defer_scope returns the number of reductions currently queued on that stream. */
int dsvg_defer_scope(int32_t on, void* stream);
/* out[0 .. n) = 0 (fp32): queued as a reduction over zero partial rows while a scope is open on the stream (the fill then costs
 * no launch of its own: the rows of a head's weight gradient that saw no loss term, functional.ArgsHeadLossFn), a plain
 * asynchronous memset otherwise.  Same ordering rules as the queued reductions. */
int dsvg_defer_zero(float* out, int64_t n, void* stream);
int dsvg_flush_deferred(void* stream);
/* Grouped weight-gradient launches.  While a group scope is open on a stream (dsvg_gemm_group_scope(1, s) ... (0, s)) every
 * split-K weight-gradient dsvg_gemm on that stream that would take the 4-stage LDS-DMA kernel (bf16, both operands
 * token-major, at most 256 workgroups) is queued, and closing the scope runs all of them as ONE launch (problem table in
 * the kernel arguments; each problem keeps its own grid, slices and workspace: bit-identical results).  The operands and
 * workspaces must stay alive and unchanged until the scope is closed; the split-K reductions of queued problems never run
 * before them (a flush of the deferred reductions, or an immediate reduction, launches the queued group first). */
int dsvg_gemm_group_scope(int32_t on, void* stream);

/* out[n] (+)= sum_m drop(A[m*lda+n])  — bias gradients (autograd of the `+ b` in every nn.Linear).
 * workspace: at least dsvg_colsum_workspace_bytes(M,N) bytes. */
int dsvg_colsum(int32_t dtype, const void* A, int64_t lda, int64_t M, int32_t N, float* out,
                int32_t accumulate, float drop_p, uint32_t drop_site, const uint64_t* seed,
                float* workspace, int64_t workspace_bytes, void* stream);
int64_t dsvg_colsum_workspace_bytes(int64_t M, int32_t N);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (d % 4 == 0, d <= 1024), eps as given.
 * Replaces torch.nn.LayerNorm at deepsvg/model/layers/improved_transformer.py:43,51,127,138 and
 * deepsvg/model/layers/transformer.py:185-186,239-240.
 * bwd: dx = [res +] LN'(dy);  dgamma/dbeta partials go to workspace then are reduced into
 * dgamma/dbeta (fp32, overwritten unless accumulate).
 * ------------------------------------------------------------------------------------------ */
int dsvg_layernorm_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int64_t rows, int32_t d, float eps, void* stream);
int dsvg_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* mean,
                       const float* rstd, const float* gamma, const void* res, void* dx,
                       float* dgamma, float* dbeta, int32_t accumulate, int64_t rows, int32_t d,
                       float* workspace, int64_t workspace_bytes, void* stream);
int64_t dsvg_layernorm_bwd_workspace_bytes(int64_t rows, int32_t d);
/* the same with a second output: dx_masked = dsvg_drop_apply(dx, drop_p, drop_site) - the stored (rounded) dx with that
 * site's dropout mask replayed on it - from the same pass over the row (round 5: the FFN half of the layer below reads its
 * incoming gradient only through the mask of its residual dropout, deepsvg/model/layers/improved_transformer.py:53,140) */
int dsvg_layernorm_bwd_masked(int32_t dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                              const float* gamma, const void* res, void* dx, float* dgamma, float* dbeta,
                              int32_t accumulate, int64_t rows, int32_t d, float* workspace, int64_t workspace_bytes,
                              void* dx_masked, float drop_p, uint32_t drop_site, const void* seed, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head self-attention core on packed QKV (after the in-projection):
 *   qkv[t, 0:d]=q, [d:2d]=k, [2d:3d]=v ; head h uses columns h*32..h*32+31 (head_dim must be 32);
 *   sequence b owns rows b*S..b*S+S-1 (S <= 64); key j of sequence b is visible iff bit j of
 *   key_mask[b] is set (NULL = all S keys).  q is scaled by `scale` after the bias add.
 *   out[t, h*32+c] = sum_j drop(softmax_j(scale*q_i.k_j))*v_j
 * Replaces deepsvg/model/layers/functional.py:168,197-248 (scale, reshape, bmm, masked_fill,
 * softmax, dropout, bmm, head merge).  bwd recomputes the probabilities (no S×S tensor in HBM).
 * Packed layout (seq_off != NULL, key_mask == NULL): sequence b owns rows seq_off[b]..seq_off[b+1]-1
 * (at most S of them, all visible); rows seq_off[n_seq]..total_rows-1 of out / dqkv are zero-filled.
 * Dense layout with total_rows > n_seq*S: rows n_seq*S..total_rows-1 of out / dqkv are zero-filled
 * (backward over a row prefix, see dsvg_visible_first); total_rows == 0: no tail.
 * tile_first (optional, packed layout only; from dsvg_attention_tiles): groups of consecutive sequences with
 * at most 32 rows in total are processed by one workgroup as a block-diagonal 32x32 score tile (bf16, S <= 32;
 * ignored by the other kernel variants).  Same results, ~3x fewer workgroups on the packed encoder.
 * ------------------------------------------------------------------------------------------ */
int dsvg_attention_fwd(int32_t dtype, const void* qkv, const uint64_t* key_mask, const int32_t* seq_off,
                       int64_t total_rows, const int32_t* tile_first, void* out, int64_t n_seq, int32_t S,
                       int32_t n_heads, float scale, float drop_p, uint32_t drop_site, const uint64_t* seed,
                       void* stream);
int dsvg_attention_bwd(int32_t dtype, const void* qkv, const uint64_t* key_mask, const int32_t* seq_off,
                       int64_t total_rows, const int32_t* tile_first, const void* dout, void* dqkv, int64_t n_seq,
                       int32_t S, int32_t n_heads, float scale, float drop_p, uint32_t drop_site,
                       const uint64_t* seed, void* stream);
/* tile_first: int32 [n_seq + 2]; tile j = sequences tile_first[j]..tile_first[j+1]-1 (sum of their lengths
 * <= max_rows; greedy inside segments of 64 consecutive sequences), tile_first[n_tiles] = n_seq,
 * tile_first[n_seq + 1] = n_tiles; scratch: int32 [ceil(n_seq / 64) * 64] */
int dsvg_attention_tiles(const int32_t* seq_off, int64_t n_seq, int32_t max_rows, int32_t* tile_first,
                         int32_t* scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Masks from the command tensor (deepsvg/model/utils.py:7-66).  commands: float32 [n_seq, S]
 * (batch-first rows, values are command indices; EOS = eos_id).
 *   key_mask[b]   bit j set  <=>  no EOS at positions <= j   (complement of _get_key_padding_mask)
 *   seq_visible[b] = (#EOS in row b) < S-1                    (_get_visibility_mask)
 * and per group-of-G rows: group_mask[n] bit g = seq_visible[n*G+g]  (_get_key_visibility_mask)
 * ------------------------------------------------------------------------------------------ */
int dsvg_build_masks(const float* commands, int64_t n_seq, int32_t S, int32_t G, int32_t eos_id,
                     uint64_t* key_mask, int32_t* seq_visible, uint64_t* group_mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * SVGEmbedding pieces (deepsvg/model/model.py:46-57):
 *  gather:  A[t, a*E:(a+1)*E] = arg_embed[args[t,a]+1]   (fp32 table -> dtype)  and
 *           R[t, :] = command_embed[commands[t]] (+ group_embed[groups[t]] when given)
 *  bwd:     d_arg_embed += scatter(dA), d_command_embed += scatter(dR), d_group_embed likewise.
 * ------------------------------------------------------------------------------------------ */
int dsvg_embed_gather(int32_t dtype, const float* commands, const float* args,
                      const float* command_embed, const float* arg_embed, const float* group_embed,
                      const int32_t* groups, void* A, void* R, int64_t T, int32_t n_args, int32_t E,
                      int32_t d, int32_t n_cmd, int32_t n_argvals, void* stream);
int dsvg_embed_scatter(int32_t dtype, const float* commands, const float* args,
                       const int32_t* groups, const void* dA, const void* dR, float* d_arg_embed,
                       float* d_command_embed, float* d_group_embed, int64_t T, int32_t n_args,
                       int32_t E, int32_t d, int32_t n_cmd, int32_t n_argvals, int32_t n_groups,
                       float* workspace, int64_t workspace_bytes, void* stream);
int64_t dsvg_embed_scatter_workspace_bytes(int64_t T, int32_t n_args, int32_t E, int32_t d,
                                           int32_t n_cmd, int32_t n_argvals, int32_t n_groups);
/* group index per token: groups[b*S+s] = #{ s' <= s : commands[b,s'] == m_id }  (utils.py:35-42) */
int dsvg_group_index(const float* commands, int64_t n_seq, int32_t S, int32_t m_id, int32_t* groups,
                     void* stream);
/* Visible-first order of the second decoder stage: its sequences are independent (one per group,
 * model/model.py:250-262) and SVGLoss excludes every position of an invisible target group (loss.py:36,51-54),
 * so those sequences have an identically zero backward pass; with the visible sequences first the backward
 * kernels run on a row prefix.  new_of_old / old_of_new: stable partition and its inverse. */
int dsvg_visible_first(const int32_t* visible, int64_t n, int32_t* new_of_old, int32_t* old_of_new,
                       int32_t* n_visible, void* stream);
/* dst[g*S + s, :] = src[idx[g]*S + s, :], g < n_groups: whole-sequence row gather (width % 4 == 0).  src holds n_src
 * groups: idx[g] >= n_src gives a zero sequence (the second decoder stage's forward runs on the visible sequences only;
 * the output rows of the others are zeros until somebody reads their logits), idx[g] < 0 (list padding) reads group 0 */
int dsvg_gather_groups(int32_t dtype, const void* src, const int32_t* idx, void* dst, int64_t n_groups,
                       int32_t S, int32_t width, int64_t n_src, void* stream);
/* Packed token layout of the first encoder stage: only keys are masked there (layers/functional.py:234-239)
 * and padded query rows are dropped by the mean-pool (model/model.py:137), so the encoder can run on the
 * valid tokens only, bit-for-bit safe.  seq_off[b] = exclusive scan of popcount(key_mask) (n_seq+1 entries,
 * seq_off[n_seq] = number of valid tokens); packed row seq_off[b]+s = token (b, s); rows past the total, up
 * to the capacity n_seq*S, replicate token 0.  packed_pos[row] = s (position index). */
int dsvg_pack_tokens(const float* commands, const float* args, const uint64_t* key_mask, int64_t n_seq,
                     int32_t S, int32_t n_args, int32_t* seq_off, float* packed_commands,
                     float* packed_args, int32_t* packed_pos, void* stream);

/* ------------------------------------------------------------------------------------------
 * y[t,:] = drop( (x ? x[t,:] : 0) + pos_embed[t % S, :] )
 * PositionalEncodingLUT.forward (deepsvg/model/layers/positional_encoding.py:40-43) and
 * ConstEmbedding.forward (deepsvg/model/model.py:70-73, x == NULL).
 * bwd: dx = dy*mask (if dx != NULL), d_pos[s,:] (+)= sum_b (dy*mask)[b*S+s,:]
 * ------------------------------------------------------------------------------------------ */
int dsvg_add_pos_fwd(int32_t dtype, const void* x, const float* pos, void* y, int64_t n_seq, int32_t S,
                     int32_t d, float drop_p, uint32_t drop_site, const uint64_t* seed, void* stream);
int dsvg_add_pos_bwd(int32_t dtype, const void* dy, void* dx, float* d_pos, int32_t accumulate,
                     int64_t n_seq, int32_t S, int32_t d, float drop_p, uint32_t drop_site,
                     const uint64_t* seed, float* workspace, int64_t workspace_bytes, void* stream);
int64_t dsvg_add_pos_bwd_workspace_bytes(int64_t n_seq, int32_t S, int32_t d);

/* ------------------------------------------------------------------------------------------
 * Masked mean over the sequence axis (deepsvg/model/model.py:137,161):
 *   out[b,:] = sum_{s in mask[b]} x[b*S+s,:] / popcount(mask[b])
 * Packed layout (seq_off != NULL): mean over rows seq_off[b]..seq_off[b+1]-1; bwd zero-fills the rows
 * seq_off[n_seq]..total_rows-1.
 * ------------------------------------------------------------------------------------------ */
int dsvg_masked_mean_fwd(int32_t dtype, const void* x, const uint64_t* mask, const int32_t* seq_off,
                         void* out, int64_t n_seq, int32_t S, int32_t d, void* stream);
int dsvg_masked_mean_bwd(int32_t dtype, const void* dout, const uint64_t* mask, const int32_t* seq_off,
                         int64_t total_rows, void* dx, int64_t n_seq, int32_t S, int32_t d, void* stream);

/* ------------------------------------------------------------------------------------------
 * x[t,:] += drop(g)[t / S, :]      "implicit broadcast" add of linear_global(z)
 * (deepsvg/model/layers/improved_transformer.py:131-136): the dropout acts on the per-sequence row BEFORE the
 * broadcast, as in the reference, so one mask element (id b*d + c) covers every position of sequence b.
 * bwd: dg[b,:] = mask[b,:] * sum_s dx[b*S+s,:] for b < n_seq; rows n_seq <= b < n_seq_out of dg are written as zeros (a
 *      backward pass restricted to a live prefix of the sequences: the others have zero gradient)
 * ------------------------------------------------------------------------------------------ */
int dsvg_bcast_add_fwd(int32_t dtype, void* x, const void* g, int64_t n_seq, int32_t S, int32_t d,
                       float drop_p, uint32_t drop_site, const uint64_t* seed, void* stream);
/* dg_ld (ABI 9): row stride (elements) of dg: d, or wider (bf16, d % 8 == 0, d <= 512, 16-byte aligned): dg is a column block of a
 * longer row - the layers of a decoder stack write their conditioning gradients side by side into ONE [n_seq_out, n_layers * d]
 * buffer, which is the concatenation GlobalCondFn's products read (no concatenation launch) */
int dsvg_bcast_add_bwd(int32_t dtype, const void* dx, void* dg, int64_t n_seq, int64_t n_seq_out, int32_t S, int32_t d,
                       float drop_p, uint32_t drop_site, const uint64_t* seed, int64_t dg_ld, void* stream);
/* bf16 only: dsvg_bcast_add_bwd AND dx_masked = dsvg_drop_apply(dx, drop_p, mask_site) over all `rows` rows of dx (n_seq * S <=
 * rows <= n_seq_out * S: a live row prefix may be rounded up past the summed sequences), from ONE read of dx (the large decoder
 * layers' backward needs both; d % 8 == 0, d <= 512, 16-byte aligned buffers, drop_p > 0) */
int dsvg_bcast_add_bwd_masked(const void* dx, void* dg, void* dx_masked, int64_t n_seq, int64_t n_seq_out, int32_t S, int32_t d,
                              int64_t rows, float drop_p, uint32_t drop_site, uint32_t mask_site, const uint64_t* seed,
                              int64_t dg_ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * SVGLoss (deepsvg/model/loss.py:19-65).
 *  targets: from tgt_commands [n_seq,S1] / tgt_args [n_seq,S1,n_args] (float32, S1=S+1 incl. SOS)
 *    cmd_tgt[n_seq,S], cmd_w[n_seq,S] (extended padding mask x visibility, loss.py:35-36,49),
 *    arg_tgt[n_seq,S,n_args] (=arg+1), arg_w = CMD_ARGS_MASK[cmd] (loss.py:51), vis_tgt[n_seq];
 *    seq_perm (optional, int32 [n_seq]): the token-level outputs of sequence b are those of source sequence seq_perm[b]
 *    (the second decoder stage's visible-first order), vis_tgt stays in source order
 *  masked CE: logical row r of the [rows, C] matrix lives at logits + (r/group)*ld + (r%group)*C
 *    (group = n_args for args_logits, whose 11x257 slots are contiguous per token; 1 otherwise);
 *    lse per row, sum_count = {sum_r w_r*(lse_r - logit[r,target_r]), sum_r w_r}; rows with w==0
 *    are never read.  loss = sum/count is formed by the caller on device.
 *  ce_bwd: dlogits = w * (softmax - onehot) * coef * (*gscale) / count  [zeros where w == 0, and
 *    zeros in the [group*C, ld_d) row padding]
 * ------------------------------------------------------------------------------------------ */
int dsvg_loss_targets(const float* tgt_commands, const float* tgt_args, const float* cmd_args_mask,
                      int64_t n_seq, int32_t S1, int32_t n_args, int32_t n_cmd, int32_t eos_id,
                      int32_t* cmd_tgt, float* cmd_w, int32_t* arg_tgt, float* arg_w, int32_t* vis_tgt,
                      const int32_t* seq_perm, void* stream);
int dsvg_masked_ce_fwd(int32_t dtype, const void* logits, int64_t ld, int32_t group, const int32_t* target,
                       const float* w, int64_t rows, int32_t C, float* lse, float* sum_count,
                       float* workspace, int64_t workspace_bytes, const int32_t* tok_idx, void* stream);
int64_t dsvg_masked_ce_workspace_bytes(int64_t rows);
int dsvg_masked_ce_bwd(int32_t dtype, const void* logits, int64_t ld, int32_t group, const int32_t* target,
                       const float* w, const float* lse, const float* sum_count, const float* gscale,
                       float coef, void* dlogits, int64_t ld_d, int64_t rows, int32_t C,
                       const int32_t* tok_idx, int32_t logits_compact, void* stream);
/* The loss terms and their weighted total from the (sum, count) pairs of n <= 4 cross-entropies in one launch (host arrays
 * of device pointers / host weights): out[1 + i] = sum_i / count_i, out[0] = sum_i weights[i] * out[1 + i]
 * (deepsvg/model/loss.py:43-57), and its backward: dsum_count[2 i] = *dtotal * weights[i] + *dterms[i] (NULL pointers
 * count as zero), dsum_count[2 i + 1] = 0 - the value dsvg_masked_ce_bwd takes as `gscale`. */
int dsvg_loss_combine_fwd(const float* const* sum_count, const float* weights, int32_t n, float* out, void* stream);
int dsvg_loss_combine_bwd(const float* dtotal, const float* const* dterms, const float* weights, int32_t n,
                          float* dsum_count, void* stream);
/* tok_idx (optional): compact token list, output token i = source token tok_idx[i] (negative -> zero row / zero
 * weight), `rows` = listed tokens * group; targets and weights are always indexed by the source token.
 * fwd with tok_idx: `logits` and `lse` are compact (row i = listed token i): the loss of the argument head is
 * computed from logits of the loss-carrying tokens only.  bwd: logits_compact selects the same for its inputs.  dsvg_live_rows builds that list: the tokens with any non-zero weight among their
 * `group` rows, ascending, padded with -1 up to n_tok entries; *count = their number.  Rows the loss masks out
 * have exactly zero dlogits (deepsvg/model/loss.py:51-54), so the argument head's dX / dW need only those tokens. */
int dsvg_live_rows(const float* w, int64_t n_tok, int32_t group, int32_t* live, int32_t* count,
                   int32_t* workspace, int64_t workspace_bytes, void* stream);
int64_t dsvg_live_rows_workspace_bytes(int64_t n_tok);
/* dst[idx[i], :] = src[i, :] (accumulate: +=) for idx[i] >= 0 (width % 4 == 0, idx without duplicates); other rows of dst are
 * left untouched */
int dsvg_scatter_rows(int32_t dtype, const void* src, const int32_t* idx, void* dst, int64_t n_rows,
                      int32_t width, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flat-buffer optimizer step: clip_grad_norm_ (deepsvg/train.py:100) + AdamW
 * (deepsvg/config.py:64-65, torch.optim.AdamW defaults) on one contiguous fp32 parameter buffer.
 * ------------------------------------------------------------------------------------------ */
int dsvg_sumsq(const float* x, int64_t n, float* out /*[1]*/, float* workspace, int64_t workspace_bytes,
               void* stream);
int64_t dsvg_sumsq_workspace_bytes(int64_t n);
int dsvg_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr,
                    float beta1, float beta2, float eps, float weight_decay, const int64_t* step,
                    const float* gnorm_sq, float max_norm, float grad_scale, void* stream);
/* dst (dtype) = src (fp32), optional transposed copy: dst_t[c*rows+r] = src[r*cols+c] */
int dsvg_cast_weights(int32_t dtype, const float* src, void* dst, void* dst_t, int64_t rows,
                      int64_t cols, void* stream);
/* *counter += 1 ; *seed = hash(*seed)  — per-step dropout seed advance, graph-capturable */
int dsvg_advance_step(int64_t* counter, uint64_t* seed, void* stream);
/* Every per-step weight image of a bf16 model in ONE launch (csrc/pack_images.hip): what dsvg_cast_weights(DSVG_BF16, flat)
 * + dsvg_ffn_pack + dsvg_attn_pack + dsvg_attn_pack_bwd + dsvg_gs_pack + dsvg_advance_step write, bit for bit (the same
 * device code, csrc/pack_images.h), i.e. the `.to(bf16)` copies of the parameters an autocast forward of
 * deepsvg/model/model.py makes, laid out for the fused kernels.  n = elements of the flat buffers (a multiple of 8, both
 * 16-byte aligned).  A family with 0 layers is skipped; ffn_w2p, attn_bwd (only a backward pass reads it), counter and seed
 * may each be NULL.  offs tables as in the stand-alone calls. */
int dsvg_pack_images(const float* flat_f32, void* flat_bf16, int64_t n,
                     const int64_t* ffn_offs, int32_t ffn_layers, void* ffn_fwd, void* ffn_bwd, float* ffn_b1f, void* ffn_w2p,
                     const int64_t* attn_offs, int32_t attn_layers, void* attn_img, void* attn_bwd,
                     const int64_t* gs_offs, int32_t gs_layers, void* gs_fwd, void* gs_bwd,
                     int64_t* counter, uint64_t* seed, void* stream);
/* n device-to-device copies dst[i][0 .. bytes[i]) = src[i][...] in one launch per 32 entries (the table travels in the kernel
 * arguments: capturable, no staging).  deepsvg_amd/trainer.py refreshes the static inputs and the layout plan of a hipGraph
 * step with it between two replays (they were ~20 separate copy launches). */
int dsvg_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream);
/* out[i] = y[i] > 0 ? dy[i]*scale : 0 — backward of ReLU (basic_blocks.py:47-57) from the saved output */
int dsvg_gate_mul(int32_t dtype, const void* dy, const void* y, void* out, int64_t n, float scale, void* stream);
/* y[i] = x[i] * dropmask(seed, site, i): replay of an nn.Dropout mask on a gradient (n % 8 == 0) */
int dsvg_drop_apply(int32_t dtype, const void* x, void* y, int64_t n, float drop_p, uint32_t drop_site,
                    const uint64_t* seed, void* stream);
/* out[i] = a[i] + b[i] — the residual add of the latent ResNet (basic_blocks.py:59-65) */
int dsvg_add(int32_t dtype, const void* a, const void* b, void* out, int64_t n, void* stream);
/* Causal variant for the autoregressive decoder (pred_mode = "autoregressive": tgt_mask =
 * square_subsequent_mask, deepsvg/model/model.py:219-222,270; layers/functional.py:228-233): query row i attends the
 * keys j <= i that key_mask allows.  Dense layout only. */
int dsvg_attention_causal_fwd(int32_t dtype, const void* qkv, const uint64_t* key_mask, void* out, int64_t n_seq,
                              int32_t S, int32_t n_heads, float scale, float drop_p, uint32_t drop_site,
                              const uint64_t* seed, void* stream);
/* dsvg_attention_bwd with the out_proj backward inside (bf16, 8 heads of 32; dense sequences of 17 .. 32 rows, or the packed
 * tile layout): dx1m = gradient of the block's projected output with the residual dropout mask on it [rows, 256] (what
 * dsvg_drop_apply / dsvg_ffn_bwd_dx hand over), wo_packed_bwd = one layer of dsvg_attn_pack_bwd.  The head-output gradient
 * dO = dx1m . Wo is formed per tile on chip: replaces the `dao = dx1m @ out_proj.weight` GEMM launch (its 512 B / token
 * written and re-read) in front of dsvg_attention_bwd (autograd of layers/functional.py:248-249 + :197-247).
 * dsvg_attn_pack_bwd: offs[layer][1] = element offset of out_proj.weight in flat_f32 (the offs table of dsvg_attn_pack);
 * packed_bwd: dsvg_attn_pack_bwd_elems(n_layers) bf16 elements (512 x 512 per layer: 128 fragments of out_proj.weight^T, then
 * 384 fragments of in_proj_weight^T for dsvg_attn_bwd_dx; offs[layer][0] = element offset of in_proj_weight). */
int dsvg_attention_bwd_outproj(const void* qkv, const uint64_t* key_mask, const int32_t* seq_off, int64_t total_rows,
                               const int32_t* tile_first, const void* dx1m, const void* wo_packed_bwd, void* dqkv,
                               int64_t n_seq, int32_t S, float scale, float drop_p, uint32_t drop_site,
                               const uint64_t* seed, void* stream);
/* Input gradient of the attention sub-block in ONE launch (bf16, d_model 256; csrc/attn_bwd_dx.hip):
 *     dx = res + LayerNorm'( dqkv . in_proj_weight ),   dgamma = sum_t dxn1 * xh,   dbeta = sum_t dxn1
 * with dxn1 = dqkv . in_proj_weight [rows, 256] never written: replaces the input-gradient dsvg_gemm behind dsvg_attention_bwd
 * / dsvg_attention_bwd_outproj and the dsvg_layernorm_bwd (_masked) of norm1 behind it - autograd of
 * deepsvg/model/layers/improved_transformer.py:43-45 / :127-129 (norm1 + in_proj of layers/functional.py:92).
 * dqkv [rows, 768], x / res / dx [rows, 256] bf16 row-major; mean / rstd: the statistics dsvg_attn_block_fwd /
 * dsvg_layernorm_fwd stored; gamma fp32 [256]; packed_bwd_layer: one layer of dsvg_attn_pack_bwd (fragments 128 .. 511 =
 * in_proj_weight^T); dgamma / dbeta fp32 [256], written (accumulate = 0) or added to, through the deterministic partial
 * reduction (queued inside a dsvg_defer_scope); workspace: dsvg_attn_bwd_dx_workspace_bytes(rows).
 * dx_masked (optional): second output = dsvg_drop_apply(dx, drop_p, drop_site) as in dsvg_layernorm_bwd_masked. */
int64_t dsvg_attn_bwd_dx_workspace_bytes(int64_t rows);
/* development probe of dsvg_attn_bwd_dx: buf = device buffer of (workgroups x 4 waves x 8) uint64 stamps of the 100 MHz
 * counter (wave start, first rows staged, K loop done, pass 1 done, sums published, stores issued) or NULL (off);
 * scripts/attn_bwd_dx_probe.py */
int dsvg_attn_bwd_dx_debug_clock(void* buf);
int dsvg_attn_bwd_dx(const void* dqkv, const void* x, const float* mean, const float* rstd, const float* gamma,
                     const void* res, const void* packed_bwd_layer, void* dx, float* dgamma, float* dbeta,
                     int32_t accumulate, int64_t rows, float* workspace, int64_t workspace_bytes, void* dx_masked,
                     float drop_p, uint32_t drop_site, const void* seed, void* stream);
int64_t dsvg_attn_pack_bwd_elems(int32_t n_layers);
int dsvg_attn_pack_bwd(const float* flat_f32, const int64_t* offs, int32_t n_layers, void* packed_bwd, void* stream);
int dsvg_attention_causal_bwd(int32_t dtype, const void* qkv, const uint64_t* key_mask, const void* dout, void* dqkv,
                              int64_t n_seq, int32_t S, int32_t n_heads, float scale, float drop_p,
                              uint32_t drop_site, const uint64_t* seed, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sequences of more than 64 tokens (one-stage / autoregressive configs at the reference's default
 * max_total_len = 240, deepsvg/model/config.py:43,74-89).  The command masks are "before the first EOS"
 * (deepsvg/model/utils.py:7-32), i.e. a valid-prefix LENGTH per sequence instead of a 64-bit word:
 *   seq_lens:       lens[b] = index of the first EOS of commands[b, :] (S if there is none)
 *   attention_long: same contract as dsvg_attention_fwd/bwd on the dense layout (keys j < seq_len[b], NULL = all;
 *                   causal != 0: additionally j <= i), S <= 256, same dropout element ids; only_row >= 0 (forward,
 *                   causal): compute that query row alone (incremental decoding step), other rows of out untouched
 *   prefix_mean:    out[b,:] = mean_{s < lens[b]} x[b*S + s, :]  (deepsvg/model/model.py:137) and its backward
 * ------------------------------------------------------------------------------------------ */
int dsvg_seq_lens(const float* commands, int64_t n_seq, int32_t S, int32_t eos_id, int32_t* lens, void* stream);
int dsvg_attention_long_fwd(int32_t dtype, const void* qkv, const int32_t* seq_len, void* out, int64_t n_seq, int32_t S,
                            int32_t n_heads, float scale, int32_t causal, int32_t only_row, float drop_p,
                            uint32_t drop_site, const uint64_t* seed, void* stream);
int dsvg_attention_long_bwd(int32_t dtype, const void* qkv, const int32_t* seq_len, const void* dout, void* dqkv,
                            int64_t n_seq, int32_t S, int32_t n_heads, float scale, int32_t causal, float drop_p,
                            uint32_t drop_site, const uint64_t* seed, void* stream);
int dsvg_prefix_mean_fwd(int32_t dtype, const void* x, const int32_t* lens, void* out, int64_t n_seq, int32_t S, int32_t d,
                         void* stream);
int dsvg_prefix_mean_bwd(int32_t dtype, const void* dout, const int32_t* lens, void* dx, int64_t n_seq, int32_t S,
                         int32_t d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Hungarian self-matching (HierarchicalSelfMatching, deepsvg/model/config.py:101-108):
 * SVGTransformer.perfect_matching, deepsvg/model/model.py:311-350.
 *  match_costs: cost[n, g, p] = w_args * mean_{masked (s,a)} CE(args_logits[n,p,s,a,:], tgt_args[n,g,s+1,a] + 1)
 *                             + w_cmd  * mean_{extended valid s} CE(command_logits[n,p,s,:], tgt_commands[n,g,s+1])
 *                             + w_vis  * CE(visibility_logits[n,p,:], visible[n,g])            (:329-339; 2, 1, 1)
 *    logits: rows (n*Gp + p)*S + s (S = S1 - 1) with row strides ld_* (elements); targets are the float32
 *    tgt_commands [N,G,S1] / tgt_args [N,G,S1,n_args] including their SOS column; the masks are taken on the
 *    sequence without it, as the reference does (:388, :314-315); visible[n,g] is returned for match_assign.
 *    Groups without argument slots get 0/0 like the reference; they are invisible and never assigned.
 *  match_assign: scipy.optimize.linear_sum_assignment(cost[n][visible rows]) (:344) as an exhaustive search
 *    (G, Gp <= 8): assign[n, j] = prediction matched to the j-th visible target, then the unmatched predictions in
 *    ascending order (:345-347); idx[n*Gp + j] = n*Gp + assign[n, j] and its inverse `inv` are the whole-sequence
 *    gather lists for dsvg_gather_groups (:390-393). */
int dsvg_match_costs(int32_t dtype, const void* cmd_logits, int64_t ld_c, const void* args_logits, int64_t ld_a,
                     const void* vis_logits, int64_t ld_v, const float* tgt_commands, const float* tgt_args,
                     const float* cmd_args_mask, int64_t N, int32_t G, int32_t Gp, int32_t S1, int32_t n_args,
                     int32_t args_dim, int32_t n_cmd, int32_t eos_id, float w_args, float w_cmd, float w_vis,
                     float* cost, int32_t* visible, void* stream);
int dsvg_match_assign(const float* cost, const int32_t* visible, int64_t N, int32_t G, int32_t Gp,
                      int32_t* assign, int32_t* idx, int32_t* inv, void* stream);

/* out[r] = argmax_c logits(r, c) for the logical rows r of a [rows, C] matrix stored like the masked-CE operand
 * (row r at logits + (r / group) * ld + (r % group) * C); ties -> lowest class.  The temperature -> 0 limit of
 * _sample_categorical (deepsvg/model/utils.py:75-80), used by greedy_sample(temperature=0). */
int dsvg_argmax_rows(int32_t dtype, const void* logits, int64_t ld, int32_t group, int64_t rows, int32_t C,
                     int32_t* out, void* stream);
/* out[r] = a draw from softmax(logits(r, :) / temperature), temperature > 0: _sample_categorical itself
 * (deepsvg/model/utils.py:75-79, torch.distributions.Categorical(logits = logits / T).sample()) as a Gumbel arg-max
 * arg-max_c (logit_c + T g_c) with the noise of the counter hash (seed: 8 bytes of device memory, site: stream id) - no
 * softmax, no cumulative sum, no fp32 copy of the logits.  Same addressing as dsvg_argmax_rows; the noise of logical row r,
 * class c is that of element (r / group, (r % group) * C + c), shared with dsvg_head_sample. */
int dsvg_sample_rows(int32_t dtype, const void* logits, int64_t ld, int32_t group, int64_t rows, int32_t C,
                     float temperature, const void* seed, uint32_t site, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * The argument head fused with its consumers (csrc/head_fused.hip): args_fcn = Linear(256 -> n_args *
 * args_dim) of deepsvg/model/model.py:228-246 evaluated tile by tile on chip - the [tokens, group * C] logits never reach
 * memory.  bf16 x [rows, 256]; `packed` = dsvg_head_pack of the head's weight rows in use ([n_out, 256] bf16, row-major,
 * n_out = group * C with slots of C >= 64 consecutive outputs, n_out <= 3008); bias fp32 [n_out].
 *   dsvg_head_argmax   out_idx[row * group + slot] = argmax_c logits(row, slot, c), ties -> lowest class: the temperature
 *                      -> 0 limit of _sample_categorical (deepsvg/model/utils.py:75-80) without the logits (replaces
 *                      the head GEMM + dsvg_argmax_rows in greedy_sample(temperature=0)).
 *   dsvg_head_sample   the same at a temperature > 0: out_idx = a draw from softmax(logits(row, slot, :) / temperature), the
 *                      reference's default decoding (greedy_sample(temperature = 1e-4), deepsvg/model/model.py:414-418) as a
 *                      Gumbel arg-max on the on-chip logit tile; draws identical to dsvg_sample_rows on the dense logits.
 *   dsvg_head_lse      the masked cross-entropy forward of SVGLoss (deepsvg/model/loss.py:51-57) on the compact token list:
 *                      lse[row * group + slot] (0 where the weight is 0) and sum_count = (sum of w (lse - logit[target]),
 *                      sum of w).  target / w are indexed tok * group + slot with tok = tok_idx ? tok_idx[row] : row
 *                      (tok < 0: list padding, weight 0) exactly as dsvg_masked_ce_fwd takes them (replaces head GEMM +
 *                      dsvg_masked_ce_fwd).  workspace: dsvg_head_lse_workspace_bytes(rows).
 *   dsvg_head_dlogits  its backward: dlogits[row, c] = w g (softmax - onehot), g = coef * (gscale ? *gscale : 1) /
 *                      sum_count[1], bf16 [rows, ld_d], ld_d = n_out rounded up to a multiple of 8, padding columns zero
 *                      (replaces a second head GEMM + dsvg_masked_ce_bwd; the logits are recomputed on chip). */
int64_t dsvg_head_pack_elems(int32_t n_out);
int dsvg_head_pack(const void* weight_bf16, int32_t n_out, void* packed, void* stream);
int dsvg_head_argmax(const void* x, const void* packed, const float* bias, int64_t rows, int32_t n_out, int32_t C,
                     int32_t* out_idx, void* stream);
int dsvg_head_sample(const void* x, const void* packed, const float* bias, int64_t rows, int32_t n_out, int32_t C,
                     float temperature, const void* seed, uint32_t site, int32_t* out_idx, void* stream);
int64_t dsvg_head_lse_workspace_bytes(int64_t rows);
int dsvg_head_lse(const void* x, const void* packed, const float* bias, int64_t rows, int32_t n_out, int32_t C,
                  const int32_t* target, const float* w, const int32_t* tok_idx, float* lse, float* sum_count,
                  float* workspace, int64_t workspace_bytes, void* stream);
int dsvg_head_dlogits(const void* x, const void* packed, const float* bias, int64_t rows, int32_t n_out, int32_t C,
                      const int32_t* target, const float* w, const int32_t* tok_idx, const float* lse,
                      const float* sum_count, const float* gscale, float coef, void* dlogits, int64_t ld_d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Device-side batch assembly.  Replaces, for a whole batch, the per-item chain of
 * SVGTensorDataset.get_data (deepsvg/svgtensor_dataset.py:164-205): SVGTensor.from_data(...).add_eos().add_sos()
 * .pad(seq_len) (deepsvg/difflib/tensor.py:85-88,108-116,125-143), .cmds()/.args()/.get_relative_args()
 * (tensor.py:155-162,172-189) and the DataLoader's default collate (deepsvg/train.py:27-28).
 *   rows     int16 [n_rows, 12]: (command, 11 arguments in SVGTensor.arg_keys order) per stored drawing command
 *   slot_off int32 [n_slots+1]:  row range of slot variant*G + group; a variant is one stored (icon, augmentation)
 *                                pair (the list entries of the .pkl "tensors", svgtensor_dataset.py:106-109,155-156)
 *   variant  int32 [n_items]:    which stored variant each batch item takes
 * grouped == 0: sequences (item, group), commands [n_items, G, L], args / args_rel [n_items, G, L, 11], L = S+2;
 * grouped != 0: one sequence per item over all its groups, commands [n_items, 1, L], L = max_total_len+2.
 * Each sequence is SOS, the stored rows, EOS, then EOS padding; argument slots of SOS/EOS/pad rows are pad_val.
 * args and args_rel are optional (NULL = not wanted); args_rel follows get_relative_args with ARGS_DIM=args_dim.
 * Sequences longer than L-2 are cut (the host refuses to build such a store). */
int dsvg_assemble_batch(const int16_t* rows, int64_t n_rows, const int32_t* slot_off, int64_t n_slots,
                        const int32_t* variant, int64_t n_items, int32_t G, int32_t grouped, int32_t L,
                        float pad_val, int32_t args_dim, float* commands, float* args, float* args_rel,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused FFN sub-block, bf16, d_model = 256 / dim_feedforward = 512 (csrc/ffn_fused.hip):
 *     y = x + drop_r( linear2( drop_h( relu( linear1( LayerNorm(x) ) ) ) ) )
 * replaces norm2 + linear1 + activation + dropout + linear2 + dropout2 + the residual add of
 * deepsvg/model/layers/improved_transformer.py:51-53 (encoder layer) and :138-140 (decoder layer) in ONE launch;
 * the normalised rows and the hidden activations never reach HBM.
 *   dsvg_ffn_pack   once per optimiser step: re-lays linear1.weight [512,256] / linear2.weight [256,512] of n_layers
 *                   layers out as bf16 MFMA-fragment-major chunk images, straight from the fp32 master parameters
 *                   `flat_f32`; offs = int64 device array [n_layers][5] of element offsets of (linear1.weight,
 *                   linear1.bias, linear2.weight, norm.weight, norm.bias).  The LayerNorm's affine part is folded into
 *                   linear1: W1' = W1 diag(gamma), b1' = b1 + W1 beta (b1_folded, fp32 [n_layers][512]).
 *                   packed_fwd [n_layers][16][32 KiB], packed_bwd [n_layers][16][48 KiB] (dsvg_ffn_pack_bytes(n, 0 | 1)).
 *                   w2p (optional, bf16 [n_layers][256][512]): linear2.weight with its columns in fragment order, the B
 *                   operand of the unfused input-gradient GEMM dpre = dym . W2p.
 *   dsvg_ffn_fwd    x, y bf16 [rows, 256] (row stride 256); packed_fwd_layer / b1_folded = that layer's slices; b2 fp32;
 *                   dropout sites / seed as everywhere else (draw scheme "v2", private to the fused kernels);
 *                   stages: 0 = default (workgroups of 128 rows up to 32,768 rows, of 256 rows above), 2 = 128-row
 *                   workgroups, 3 / 4 = 256-row workgroups with that many weight-ring slots; all bit-identical.
 *                   Training calls pass h_out (bf16 [rows, 512], hidden columns in
 *                   FRAGMENT ORDER: position p(j) = j with bits 2 and 3 swapped), xh_out = (x - mean) * rstd (bf16
 *                   [rows, 256]) and rstd_out (fp32 [rows]) for the backward pass; inference passes NULL for all three.
 * Buffers are caller-owned; all work is enqueued on `stream`. */
int64_t dsvg_ffn_pack_bytes(int32_t n_layers, int32_t which);
int dsvg_ffn_pack(const float* flat_f32, const int64_t* offs, int32_t n_layers, int32_t d_model, int32_t d_ff,
                  void* packed_fwd, void* packed_bwd, float* b1_folded, void* w2p, void* stream);
int dsvg_ffn_fwd(const void* x, const void* packed_fwd_layer, const float* b1_folded, const float* b2, void* y,
                 void* h_out, void* xh_out, float* rstd_out, int64_t rows, float eps, float drop_p,
                 uint32_t site_hidden, uint32_t site_res, const void* seed, int32_t stages, void* stream);
/* Backward of the fused FFN sub-block (two launches; csrc/ffn_fused.hip):
 *   kernel 1 recomputes the hidden tile from x, replays both dropout masks and writes what the weight-gradient GEMMs
 *            need: h, dpre bf16 [rows, 512] with the hidden columns in FRAGMENT ORDER (position p(j) = j with bits 2 and 3
 *            swapped), xh = (x - mean) * rstd and dym = dy * residual-dropout mask, bf16 [rows, 256] (dym may be NULL
 *            when drop_p == 0: then dym == dy);
 *   kernel 2 dx = dy + LayerNorm'(dpre . W1').
 * The caller then runs G2p = dym^T h [256, 512], G1p = dpre^T xh [512, 256] (+ row sums db1p, and db2 = colsum(dym))
 * with dsvg_gemm, and dsvg_ffn_wgrad_finish turns (G1p, db1p, G2p) into the gradients of linear1.weight / bias,
 * linear2.weight and of the LayerNorm's gamma / beta (w1 = fp32 master linear1.weight [512, 256]).
 * Replaces the autograd backward of deepsvg/model/layers/improved_transformer.py:51-53 / :138-140. */
int dsvg_ffn_bwd(const void* x, const void* dy, const void* packed_bwd_layer, const float* b1_folded, void* h, void* dpre,
                 void* xh, void* dym, void* dx, int64_t rows, float eps, float drop_p, uint32_t site_hidden,
                 uint32_t site_res, const void* seed, void* stream);
/* kernel 2 alone: dx = dy + LayerNorm'(dpre . W1') from a dpre (bf16 [rows, 512], fragment order) the caller produced
 * (training default: dpre = (dym . W2p) gated by the h the forward kernel stored, one dsvg_gemm).
 * dx_masked (optional): a second output, dx with the dropout mask (drop_p, drop_site, ids row * 256 + column) replayed on
 * it - exactly dsvg_drop_apply(dx) - for the attention sub-block's backward, which starts from it */
int dsvg_ffn_bwd_dx(const void* dpre, const void* x, const void* dy, const void* packed_bwd_layer, void* dx,
                    int64_t rows, float eps, void* dx_masked, float drop_p, uint32_t drop_site, const void* seed,
                    void* stream);
int dsvg_ffn_wgrad_finish(const float* g1p, const float* db1p, const float* g2p, const float* w1, const float* gamma,
                          const float* beta, float* dw1, float* db1, float* dw2, float* dgamma, float* dbeta,
                          void* stream);
/* the same for n_layers layers in one launch per 16: ptrs = n_layers x 11 pointers in the argument order of
 * dsvg_ffn_wgrad_finish (g1p, db1p, g2p, w1, gamma, beta, dw1, db1, dw2, dgamma, dbeta), a host array */
int dsvg_ffn_wgrad_finish_many(const void* const* ptrs, int32_t n_layers, void* stream);
/* development probe of dsvg_ffn_fwd: buf = device buffer of (workgroups x 8 waves x 4) uint64 time stamps (kernel start,
 * LayerNorm done, chunk loop done, stores issued) or NULL to switch it off (scripts/ffn_phase_probe.py) */
int dsvg_ffn_debug_clock(void* buf);
/* the same for dsvg_gs_layer_fwd: (workgroups * 8 * 8) uint64 - kernel start, LayerNorm 1, in_proj + attention, out_proj +
 * LayerNorm 2, linear1, linear2, stores issued (scripts/gs_phase_probe.py) */
int dsvg_gs_debug_clock(void* buf);

/* ------------------------------------------------------------------------------------------
 * Fused attention sub-block (d_model 256, 8 heads of 32, sequences of at most 32 tokens, bf16):
 *     x1 = x + drop_r( out_proj( MHA( LayerNorm(x) ) ) )
 * one launch instead of LayerNorm + in_proj GEMM + attention + out_proj GEMM
 * (deepsvg/model/layers/improved_transformer.py:43-46 and :127-131, layers/attention.py, layers/functional.py:168,197-248).
 * dsvg_attn_pack: bf16 MFMA-fragment images (dsvg_attn_pack_bytes(n_layers) bytes, 512 KiB per layer) of in_proj_weight
 *   and out_proj.weight straight from the fp32 master buffer; offs = int64 [n_layers][2] element offsets of
 *   (in_proj_weight, out_proj.weight).  Re-pack after every optimiser step.
 * dsvg_attn_block_fwd: x bf16 [rows, 256].  Layouts as dsvg_attention_fwd: dense (seq_off NULL: sequence b = rows
 *   b*S .. b*S+S-1, optional key_mask bit j = key j visible) or packed (seq_off + tile_first from dsvg_attention_tiles,
 *   block-diagonal attention inside each tile of <= 32 rows).  Rows behind the last sequence (bucket padding up to
 *   `rows`) get finite values.  Dropout: site_probs on the probabilities (same element ids as dsvg_attention_fwd),
 *   site_res on the residual branch (ids row*256 + col, replayable by dsvg_drop_apply).
 *   seq_add (optional, dense layouts): bf16 [n_seq, 256] with row stride seq_add_ld elements (256 = contiguous; a column block
 *   of a wider matrix otherwise: the conditioning rows of all the layers of a stack come from ONE GEMM), x1 += drop(seq_add[sequence]) with one mask element per
 *   (sequence, channel), site_seq_add - the decoder's linear_global(z) term (improved_transformer.py:131-136), same draws
 *   as dsvg_bcast_add_fwd, whose backward (dsvg_bcast_add_bwd) applies unchanged.
 *   Training outputs (all NULL for inference, all set otherwise) are what the unfused backward reads:
 *   xn_out = LayerNorm(x) bf16 [rows,256], qkv_out bf16 [rows,768], ao_out = head outputs bf16 [rows,256],
 *   mean_out / rstd_out fp32 [rows].
 * ------------------------------------------------------------------------------------------ */
int64_t dsvg_attn_pack_bytes(int32_t n_layers);
int dsvg_attn_pack(const float* flat_f32, const int64_t* offs, int32_t n_layers, int32_t d_model, int32_t n_heads,
                   void* packed, void* stream);
int dsvg_attn_block_fwd(const void* x, const void* packed_layer, const float* in_bias, const float* out_bias,
                        const float* gamma, const float* beta, const uint64_t* key_mask, const int32_t* seq_off,
                        const int32_t* tile_first, int64_t n_seq, int32_t S, int64_t rows, void* x1, void* xn_out,
                        void* qkv_out, void* ao_out, float* mean_out, float* rstd_out, float eps, float scale,
                        float drop_p, uint32_t site_probs, uint32_t site_res, const void* seed,
                        const void* seq_add, int64_t seq_add_ld, uint32_t site_seq_add, void* stream);
/* ------------------------------------------------------------------------------------------
 * One launch per layer and direction for the short-sequence ("group") stages: the whole pre-LN block
 *     x1 = x  + drop( out_proj( MHA( LayerNorm1(x) ) ) ) [+ drop( seq_add[sequence] )]
 *     x2 = x1 + drop( linear2( drop( relu( linear1( LayerNorm2(x1) ) ) ) ) )
 * of deepsvg/model/layers/improved_transformer.py:42-54 and :126-141 as used by hierarchical_encoder (model/model.py:153-161)
 * and hierarchical_decoder (:246-254): d_model 256, dim_ff 512, 8 heads, bf16, dense sequences of S tokens with 32 % S == 0,
 * rows = n_seq * S.  They replace ~15 (forward) / ~25 (backward) launches of 5-12 us each; what they read and write is
 * exactly what those launches read and write, with the same dropout draws (sites site0 + 0 probabilities, + 1 attention
 * residual, + 2 per-sequence term, + 3 hidden, + 4 FFN residual), so fused and unfused launches can be mixed freely.
 * dsvg_gs_pack: bf16 MFMA-fragment images (dsvg_gs_pack_bytes(n_layers) bytes = 1 MiB per layer, each) of the layer's four
 *   weight matrices for the forward and for the backward kernel, straight from the fp32 master buffer; offs = int64
 *   [n_layers][4] element offsets of (in_proj_weight, out_proj.weight, linear1.weight, linear2.weight).  Re-pack after
 *   every optimiser step.
 * dsvg_gs_layer_fwd: x bf16 [rows,256] -> x2.  key_mask (optional): bit j of key_mask[b] = key j of sequence b visible.
 *   seq_add (optional): bf16 [n_seq,256] (the decoder's linear_global(z) term, drawn like dsvg_bcast_add_fwd).
 *   Training outputs (all NULL for inference, all set otherwise) = what the backward pass reads, in the layouts of the
 *   unfused launches: mean1 / rstd1 / mean2 / rstd2 fp32 [rows], xn1 = LayerNorm1(x), ao (head outputs), x1,
 *   xn2 = LayerNorm2(x1) bf16 [rows,256], qkv bf16 [rows,768], h bf16 [rows,512] (after ReLU and dropout).
 *   The forward call also takes sequences of any length 1 .. 32 (32 / S whole sequences per tile) and a sequence offset:
 *   seq_base > 0 = the launch covers sequences seq_base .. seq_base + n_seq - 1 of a longer buffer (all pointers at their
 *   first row; dropout draws indexed from the buffer's first row), ffn_format != 0 = xn2 and h come out as the fused FFN
 *   kernels' backward reads them (xn2 without gamma / beta, h with fragment-ordered columns, dsvg_ffn_fwd).  Together: the
 *   training forward of a large dense stage runs the sequences that fill whole rounds of the chip on dsvg_attn_block_fwd +
 *   dsvg_ffn_fwd and the remainder on this kernel, into the same saved tensors.
 * dsvg_gs_layer_bwd: dx2 = dL/dx2 bf16 [rows,256] and the saved tensors -> dx = dL/dx, plus the token-major operands of
 *   the four weight-gradient GEMMs the caller runs afterwards - dym = dx2 * mask(site0+4) (with h: linear2), dpre (with
 *   xn2: linear1), dx1m = dx1 * mask(site0+1) (with ao: out_proj), dqkv (with xn1: in_proj); their column sums are the bias
 *   gradients - dx1 (optional, for dsvg_bcast_add_bwd of the per-sequence term) and the four LayerNorm parameter
 *   gradients (fp32 [256] each; reduced from per-tile partials in `workspace`, dsvg_gs_bwd_workspace_bytes(n_seq, S)
 *   bytes, through the deferred-reduction queue when a scope is open on the stream).
 *   dg (optional, bf16 [n_seq, 256]; ABI 6): the gradient of the per-sequence term, = dsvg_bcast_add_bwd(dx1, site0 + 2) bit
 *   for bit, formed from the tile while dx1 is on chip - that launch and the dx1 store (pass dx1 = NULL) are then not needed.
 * ------------------------------------------------------------------------------------------ */
int64_t dsvg_gs_pack_bytes(int32_t n_layers);
int dsvg_gs_pack(const float* flat_f32, const int64_t* offs, int32_t n_layers, int32_t d_model, int32_t d_ff,
                 int32_t n_heads, void* packed_fwd, void* packed_bwd, void* stream);
int dsvg_gs_layer_fwd(const void* x, const void* packed_fwd_layer, const float* in_bias, const float* out_bias,
                      const float* b1, const float* b2, const float* gamma1, const float* beta1, const float* gamma2,
                      const float* beta2, const uint64_t* key_mask, const void* seq_add, int64_t seq_add_ld, int64_t n_seq, int32_t S,
                      void* x2, float* mean1, float* rstd1, void* xn1, void* qkv, void* ao, void* x1, float* mean2,
                      float* rstd2, void* xn2, void* h, float eps, float scale, float drop_p, uint32_t site0,
                      const void* seed, int64_t seq_base, int32_t ffn_format, void* stream);
int64_t dsvg_gs_bwd_workspace_bytes(int64_t n_seq, int32_t S);
int dsvg_gs_layer_bwd(const void* dx2, const void* packed_bwd_layer, const void* x, const float* mean1, const float* rstd1,
                      const void* qkv, const void* x1, const float* mean2, const float* rstd2, const void* h,
                      const float* gamma1, const float* gamma2, const uint64_t* key_mask, int64_t n_seq, int32_t S,
                      void* dx, void* dx1, void* dym, void* dpre, void* dx1m, void* dqkv, float* dgamma2, float* dbeta2,
                      float* dgamma1, float* dbeta1, float scale, float drop_p, uint32_t site0, const void* seed,
                      void* workspace, int64_t workspace_bytes, void* dg, void* stream);
/* One launch per STACK and direction (ABI 8): the tiles (32 rows = whole sequences) are independent across the layers, so a
 * workgroup carries its rows through up to 4 layers - reference: the layer loops of deepsvg/model/layers/transformer.py:168-188
 * (TransformerEncoder.forward) and :214-242 (TransformerDecoder.forward) over the 4 layers of hierarchical_encoder /
 * hierarchical_decoder.  Same arithmetic, same stores and same dropout draws as n_layers calls of dsvg_gs_layer_fwd /
 * dsvg_gs_layer_bwd in a row (bit-identical; tests/test_kernels_gpu.py::test_gs_stack_*), minus the launch ramps, the
 * reloads of the rows between the layers, and the gradient stores between the layers of the backward pass.
 *   dsvg_gs_stack_fwd: x -> layers[n_layers - 1].x2.  Per layer: the arguments of dsvg_gs_layer_fwd; training outputs for every
 *     layer or for none; x2 of an inner layer may be NULL for inference.  seq_add of every layer has the row stride seq_add_ld
 *     (column blocks of one [n_seq, n_layers * 256] product).  seq_base = 0, ffn_format = 0.
 *   dsvg_gs_stack_bwd: dx2 = dL/d(layers[n_layers - 1].x2) -> layers[0].dx; `layers` in FORWARD order, walked from the last
 *     one.  Per layer: the arguments of dsvg_gs_layer_bwd; dx of layers 1 .. may be NULL (not stored); every layer's workspace
 *     has workspace_bytes >= dsvg_gs_bwd_workspace_bytes(n_seq, S); dg (optional) has the row stride dg_ld (column blocks of one
 *     [n_seq, n_layers * 256] buffer: the concatenated gradient of the conditioning product, no concatenation launch). */
typedef struct dsvg_gs_fwd_layer {
    const void* packed_fwd_layer;
    const float *in_bias, *out_bias, *b1, *b2, *gamma1, *beta1, *gamma2, *beta2;
    const void* seq_add;            /* or NULL */
    void* x2;
    float *mean1, *rstd1;
    void *xn1, *qkv, *ao, *x1;
    float *mean2, *rstd2;
    void *xn2, *h;
    uint32_t site0;
    uint32_t reserved_;
} dsvg_gs_fwd_layer;
typedef struct dsvg_gs_bwd_layer {
    const void* packed_bwd_layer;
    const void* x;
    const float *mean1, *rstd1;
    const void *qkv, *x1;
    const float *mean2, *rstd2;
    const void* h;
    const float *gamma1, *gamma2;
    void *dx, *dx1, *dym, *dpre, *dx1m, *dqkv, *dg;
    float *dgamma2, *dbeta2, *dgamma1, *dbeta1;
    void* workspace;
    uint32_t site0;
    uint32_t reserved_;
} dsvg_gs_bwd_layer;
int dsvg_gs_stack_fwd(const void* x, const dsvg_gs_fwd_layer* layers, int32_t n_layers, const uint64_t* key_mask,
                      int64_t seq_add_ld, int64_t n_seq, int32_t S, float eps, float scale, float drop_p, const void* seed,
                      void* stream);
int dsvg_gs_stack_bwd(const void* dx2, const dsvg_gs_bwd_layer* layers, int32_t n_layers, const uint64_t* key_mask,
                      int64_t n_seq, int32_t S, float scale, float drop_p, const void* seed, int64_t workspace_bytes,
                      int64_t dg_ld, void* stream);
/* test hook: raw ds_read_b64_tr_b16 on a 4 KiB LDS image img[i]=i, lane l reads at byte offset off[l] */
int dsvg_probe_trread(const int* off, short* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * The latent chain between the encoder and the decoder as ONE launch per direction (csrc/group_stage.hip):
 *     z_i = z_{i-1} + relu(W_i z_{i-1} + b_i), i = 1 .. n_res   (ResNet, deepsvg/model/basic_blocks.py:59-65)
 *     out = W_b z_{n_res} + b_b                                  (Bottleneck, deepsvg/model/model.py:193-198)
 * bf16, d_model = dim_z = 256, one row per icon.  weights / biases: host arrays of n_res + 1 device pointers (row-major bf16
 * [256, 256] / fp32 [256]; the final linear last); z_out / r_out: host arrays of n_res device pointers for the training
 * outputs z_1 .. z_n and the ReLU outputs r_1 .. r_n (bf16 [rows, 256]; pass NULL arrays for inference).
 * Backward: dout = dL/dout; dpre_out[i] = dz_{i+1} where r_{i+1} > 0 (the token-major operand of dW_{i+1} = dpre^T z_i; the
 * final linear's dW takes dout and z_n), dz0 = dL/dz0.  Replaces 2 launches per block forward and 3 backward (the autograd
 * backward of the same lines). */
int dsvg_latent_chain_fwd(const void* z0, const void* const* weights, const float* const* biases, int32_t n_res,
                          void* const* z_out, void* const* r_out, void* out, int64_t rows, void* stream);
int dsvg_latent_chain_bwd(const void* dout, const void* const* weights, const void* const* r, int32_t n_res,
                          void* const* dpre_out, void* dz0, int64_t rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSVG_H */
