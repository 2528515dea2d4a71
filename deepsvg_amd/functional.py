"""autograd.Function wrappers that stitch the HIP ops (deepsvg_amd/ops.py) into the blocks of the reference
model.  One Function per reference block, so a training step is ~40 autograd nodes instead of ~700:

  EmbedFn      SVGEmbedding.forward                         deepsvg/model/model.py:46-57
  LayerFn      TransformerEncoderLayerImproved.forward /    deepsvg/model/layers/improved_transformer.py:42-54
               TransformerDecoderLayerGlobalImproved.forward                                     :126-141
  LayerNormFn  the stacks' final LayerNorm                  deepsvg/model/layers/transformer.py:185-186,239-240
  AddPosFn     PositionalEncodingLUT / ConstEmbedding       positional_encoding.py:40-43, model.py:70-73
  MaskedMeanFn the two masked mean-pools                    deepsvg/model/model.py:137,161
  LinearFn     ResNet / Bottleneck / VAE / heads            basic_blocks.py:15-23,33-39,59-65, model.py:182-197
  MaskedCEFn   the three cross-entropies of SVGLoss         deepsvg/model/loss.py:43,53-54

All arithmetic happens in the HIP kernels; `ops` may be monkey-patched by the CPU test-suite with the
plain-torch restatements in tests/torch_ops_ref.py to exercise this host logic without a GPU.
"""
import math
import contextlib
import functools
import os

import torch

from . import ops

# fused FFN backward (hidden tile recomputed in the kernel) instead of the default: see LayerFn.backward
FFN_BWD_FUSED = os.environ.get("DSVG_FFN_BWD_FUSED", "0") != "0"
# the fused FFN kernels own 256 token rows per workgroup: below ~16k rows they cannot fill the 256 CUs and the three
# unfused launches are faster (measured: 4096 rows 39-50 us fused vs 33 us unfused; 41k rows 56 vs 71 us)
FFN_MIN_ROWS = int(os.environ.get("DSVG_FFN_MIN_ROWS", "16384"))
# the fused attention block owns 8 tiles of <= 32 rows per workgroup (same granularity: unfused launches below this)
ATTN_MIN_ROWS = int(os.environ.get("DSVG_ATTN_MIN_ROWS", "16384"))
# fused-FFN backward: weight-gradient GEMMs right behind the producers of their operands (1) or at the end (0)
FFN_BWD_ORDER = os.environ.get("DSVG_FFN_BWD_ORDER", "1") != "0"
# dx and its dropout-masked copy from one ffn_bwd_dx launch instead of a drop_apply launch: measured SLOWER (8.52 vs 8.43
# ms/step: the extra pass sits on the tail of a one-workgroup-per-CU kernel), so it is opt-in
FFN_BWD_MASKED = int(os.environ.get("DSVG_FFN_BWD_MASKED", "0"))    # 1: every layer; 2: only layers without a conditioning row
# (their bcast_add_bwd launch writes the masked copy anyway, BCAST_MASKED)

# the weight-gradient GEMMs of a layer of the 4096-row stages as one grouped launch (DSVG_GROUP_WGRAD=0: one by one)
GROUP_WGRAD = os.environ.get("DSVG_GROUP_WGRAD", "1") != "0"
# the decoder layers' bcast_add_bwd also writes the masked copy of dx1 the attention half needs (one read of dx1, no drop_apply launch)
BCAST_MASKED = os.environ.get("DSVG_BCAST_MASKED", "1") != "0"
# round 5: the LayerNorm backward that produces a layer's incoming gradient also writes that gradient with the residual-dropout
# mask of the layer BELOW replayed on it (what that layer's FFN half reads): 9 drop_apply launches per step less
LN_BWD_MASKED = os.environ.get("DSVG_LN_BWD_MASKED", "1") != "0"
# round 5: the position / embedding tables' gradient reductions (add_pos_bwd, embed_scatter: 6 launches) join the deferred queue,
# and the library queues segments of any width (csrc/gemm.hip: the heads' 7- and 2827-row gradients: 2 launches); 0 = as before
DEFER_MORE = os.environ.get("DSVG_DEFER_MORE", "1") != "0"
# round 5: ONE grouped weight-gradient launch per group-stage STACK (4 layers x 4 products) instead of one per layer
STACK_GROUP = os.environ.get("DSVG_STACK_GROUP", "1") != "0"
STACK_GROUP_SLICES = int(os.environ.get("DSVG_STACK_GROUP_SLICES", "8"))    # token slices per output tile of a product in that launch (8 =
# the per-layer launches' partition: the bf16 slice sums, hence the gradients, are then the same numbers; 4 measured 0.1 % faster)
# round 5: the group-stage backward kernel also emits the conditioning term's gradient (4 bcast_add_bwd launches + 4 dx1 stores less)
GS_BWD_DG = os.environ.get("DSVG_GS_BWD_DG", "1") != "0"
# round 5: the argument head's input-gradient product with its reduced dimension padded to the LDS-DMA GEMM's K step
HEAD_KPAD = os.environ.get("DSVG_HEAD_KPAD", "1") != "0"
# round 6: the layers of a decoder stack write the gradients of their conditioning rows side by side into ONE buffer (the column
# blocks GlobalCondFn's concatenated products read): no concatenation launch; 0 = one tensor per layer + torch.cat
COND_GRAD_SHARED = os.environ.get("DSVG_COND_GRAD_SHARED", "1") != "0"
# round 6: the group-stage stacks as ONE launch per stack and direction (csrc/group_stage.hip gs_stack_*; 0: one launch per layer)
GS_STACK = os.environ.get("DSVG_GS_STACK", "1") != "0"
# attention backward of the large stages with the out_proj backward inside (no `dao = dx1m @ Wo` GEMM launch)
ATTN_BWD_OUTPROJ = os.environ.get("DSVG_ATTN_BWD_OUTPROJ", "1") != "0"
# ... which only exists on the MFMA attention kernels: the library's A/B knobs that route attention to the VALU kernels
# (dsvg_attention_mfma_ok, csrc/attention_mfma.hip) must switch it off too instead of failing the backward pass
_ATTN_VALU = os.environ.get("DSVG_ATTN_VALU") is not None
_ATTN_MFMA_MIN_S = int(os.environ.get("DSVG_ATTN_MFMA_MIN_S", "2"))
# round 6: the attention half's input gradient (dqkv . W_in + LayerNorm backward + residual) as one launch, csrc/attn_bwd_dx.hip
# (0: the GEMM + dsvg_layernorm_bwd pair of round 5); below ATTN_BWD_DX_MIN_ROWS rows the pair stays
ATTN_BWD_DX = os.environ.get("DSVG_ATTN_BWD_DX", "0") != "0"
ATTN_BWD_DX_MIN_ROWS = int(os.environ.get("DSVG_ATTN_BWD_DX_MIN_ROWS", "8192"))
# training forward of a large dense stage: sequences beyond a multiple of SEQ_ROUND (one round of the chip for the fused
# attention kernel: 256 CUs x 8 sequences) run on the group-stage layer kernel when there are at most this many (0: never)
GS_REMAINDER = int(os.environ.get("DSVG_GS_REMAINDER", "512"))
SEQ_ROUND = 2048
# argument head + masked CE with the logit tile on chip, forward and backward (csrc/head_fused.hip) instead of head GEMM ->
# stored compact logits -> masked-CE kernels.  Opt-in: on the compact token list the stored logits are only ~120 MB, and
# recomputing the tile in the backward pass costs as much as reading them - measured 7.02 vs 6.99 ms/step (same box, three
# alternating runs).  Decoding (greedy_sample at temperature 0) always uses the fused head + arg-max kernel.
HEAD_FUSED = os.environ.get("DSVG_HEAD_FUSED", "0") != "0"

_NULL_CTX = contextlib.nullcontext()

class Runtime:
    """Per-forward execution context shared by the Functions."""

    def __init__(self, dtype=torch.float32, seed=None, store=None, training=False, defer=False):
        self.dtype = dtype
        # queue the partial-sum reductions of the parameter gradients (ops.DEFER) instead of launching ~130 of them one
        # by one: only a caller that flushes before anything reads a gradient may set it (TrainStep)
        self.defer = bool(defer)
        self.seed = seed          # int64[1] device tensor holding the dropout seed of this step
        self.store = store        # ParamStore or None
        self.training = training
        # gradient tensor (data_ptr) -> (masked copy, p, site): handed from the launch that produced the gradient of a layer's
        # OUTPUT to that layer's backward, which would otherwise re-read it once more only to apply the mask (LN_BWD_MASKED)
        self.masked = {}
        self.last_layer_gs = False
        self.stack_group = None
        # base address of a stack's projected conditioning rows (GlobalCondFn's [n_seq, n * 256] product) -> the buffer its layers'
        # backward passes write their gradients into, side by side (COND_GRAD_SHARED)
        self.cond_grad = {}

    def cond_grad_block(self, z, rows):
        """z: the conditioning rows a layer received.  If they are column block i of a row-major [*, n * w] product (GlobalCondFn),
        -> column block i of the stack's shared [rows, n * w] gradient buffer (created by the first layer that asks), else None"""
        if not COND_GRAD_SHARED or z is None or z.dim() != 2 or z.stride(1) != 1 or z.dtype != torch.bfloat16:
            return None
        w, ld = z.shape[1], z.stride(0)
        if ld <= w or ld % w or w % 8 or w > 512:
            return None
        i = (z.storage_offset() % ld) // w
        if (z.storage_offset() % ld) % w:
            return None
        base = z.data_ptr() - i * w * z.element_size()
        key = (base, rows, ld)
        buf = self.cond_grad.get(key)
        if buf is None:
            if len(self.cond_grad) > 8:         # (never consumed: do not grow)
                self.cond_grad.clear()
            buf = self.cond_grad[key] = torch.empty((rows, ld), dtype=z.dtype, device=z.device)
        return buf[:, i * w:(i + 1) * w]

    def deferring(self):
        """context manager around launches whose reductions write parameter gradients"""
        return ops.DEFER if self.defer else _NULL_CTX

    def stack_group_begin(self):
        """open ONE grouped weight-gradient launch over the group-stage layers of a stack (closed by the stack's first layer,
        the last one of the backward pass) -> the list that keeps the queued products' operands alive until the launch"""
        if self.stack_group is None:
            ops.GROUP.__enter__()
            self.stack_group = []
        return self.stack_group

    def stack_group_end(self):
        if self.stack_group is not None:
            ops.GROUP.__exit__(None, None, None)        # the launch; the operands may go now
            self.stack_group = None

    def deferring_tables(self):
        """the same around the embedding / position tables' gradients (round 5; DSVG_DEFER_MORE=0: reduced on the spot)"""
        return ops.DEFER if self.defer and DEFER_MORE else _NULL_CTX

    def grouping(self):
        """context manager around a run of independent weight-gradient GEMMs: one launch for all of them (ops.GROUP)"""
        return ops.GROUP if GROUP_WGRAD else _NULL_CTX

    def p(self, rate):
        """effective dropout probability"""
        return float(rate) if (self.training and rate > 0.0) else 0.0

    def w(self, param):
        """weight matrix in the compute dtype (bf16 image of the fp32 master, or the master itself)"""
        if self.dtype == torch.float32:
            return param.detach()
        if self.store is not None:
            v = self.store.lp(param)
            if v is not None:
                return v
        return param.detach().to(self.dtype)

    def grad_out(self, param):
        """fp32 tensor the gradient of `param` is written into"""
        if self.store is not None:
            v = self.store.grad_view(param)
            if v is not None:
                return v
        return torch.empty(param.shape, dtype=torch.float32, device=param.device)


def _wgrad_tag():
    """profiling label of the weight-gradient GEMMs outside the FFN sub-block (bench.py's roofline leg)"""
    return ops.tag_default("wgrad") if ops.PROFILE_ON else _NULL_CTX


def _wgrad(rt, param, dy, x, *, a_drop_p=0.0, a_drop_site=0):
    """dW[n_out, k_in] = sum_t drop(dy)[t, n_out] * x[t, k_in]   (both operands token-major: a TN GEMM)"""
    out = rt.grad_out(param)
    n_out, k_in = param.shape
    T = dy.shape[0]
    with rt.deferring(), _wgrad_tag():
        ops.gemm(dy, x, a_kc=False, b_kc=False, out=out.view(n_out, k_in), a_drop_p=a_drop_p, a_drop_site=a_drop_site,
                 seed=rt.seed, split_k=ops.split_k_for(n_out, k_in, T))
    return out


def _wbgrad(rt, weight, bias, dy, x, blocks=None):
    """(dW, db) of y = x W^T + b from dy: db[n] = sum_t dy[t, n] is the row sum of the GEMM's A operand, so it
    rides on the weight-gradient GEMM (an extra MFMA against ones in a few workgroups) whenever that GEMM is
    split over tokens; otherwise a separate column-sum launch.  blocks: target workgroup count of this GEMM (default: a
    launch of its own, one workgroup per CU; members of a grouped launch share the chip)."""
    n_out, k_in = weight.shape
    split = ops.split_k_for(n_out, k_in, dy.shape[0], target_blocks=blocks)
    dw = rt.grad_out(weight)
    db = rt.grad_out(bias)

    with rt.deferring(), _wgrad_tag():
        if split > 1:
            ops.gemm(dy, x, a_kc=False, b_kc=False, out=dw.view(n_out, k_in), split_k=split, rowsum=db)
        else:
            ops.gemm(dy, x, a_kc=False, b_kc=False, out=dw.view(n_out, k_in))
            ops.colsum(dy, out=db)
    return dw, db


def _bgrad(rt, param, dy, *, drop_p=0.0, drop_site=0):
    out = rt.grad_out(param)
    with rt.deferring():
        ops.colsum(dy, out=out, drop_p=drop_p, drop_site=drop_site, seed=rt.seed)
    return out


def _hand_masked(rt, dx, dxm, p, site):
    """producer side of the masked-gradient hand-off"""
    if len(rt.masked) > 8:          # (never consumed: a caller that differentiates something else; do not grow)
        rt.masked.clear()
    rt.masked[dx.data_ptr()] = (dxm, p, site, dx._version)


def _take_masked(rt, dx, p, site, rows=None):
    """consumer side: the masked copy of `dx` for exactly this (p, site), or None"""
    hit = rt.masked.pop(dx.data_ptr(), None)
    if hit is None:
        return None
    dxm, hp, hsite, ver = hit
    n = dx.shape[0] if rows is None else rows
    # (an in-place change of the gradient since the hand-off - autograd accumulating a second contribution into the same buffer -
    # shows in the version counter: the masked copy would be stale)
    if (hp != p or hsite != site or dxm.dtype != dx.dtype or dxm.shape[0] < n or dxm.shape[1:] != dx.shape[1:]
            or dx._version != ver):
        return None
    return dxm[:n]


def _aligned2d(t, mult):
    """row-major 2-D tensor whose row stride is a multiple of `mult` elements (copy into a padded buffer if not)"""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % mult == 0 and t.data_ptr() % 16 == 0:
        return t
    rows, cols = t.shape
    ld = (cols + mult - 1) // mult * mult
    buf = torch.zeros((rows, ld), dtype=t.dtype, device=t.device)
    buf[:, :cols].copy_(t)
    return buf[:, :cols]


# --------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = [res +] drop(act(x W^T + b))   (x: [rows, k_in] in the compute dtype)"""

    @staticmethod
    def forward(ctx, rt, x, weight, bias, act, res, drop_rate, site, out_dtype):
        p = rt.p(drop_rate)
        # rows of the output are padded to a 16-byte multiple (e.g. the 2827-wide args logits -> 2832) so that
        # the GEMM epilogue, the loss kernels and the backward GEMMs all use 16-byte accesses
        dt = out_dtype or x.dtype
        n_out = weight.shape[0]
        mult = 4 if dt == torch.float32 else 8
        out = None
        if n_out % mult:
            ld = (n_out + mult - 1) // mult * mult
            out = torch.empty((x.shape[0], ld), dtype=dt, device=x.device)[:, :n_out]
        y = ops.gemm(x, rt.w(weight), bias=bias.detach() if bias is not None else None, act=act,
                     res=res, drop_p=p, drop_site=site, seed=rt.seed, out_dtype=out_dtype, out=out)
        ctx.rt, ctx.act, ctx.p, ctx.site = rt, act, p, site
        ctx.has_res = res is not None
        ctx.save_for_backward(x, weight, bias, y if act == ops.RELU else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        rt = ctx.rt
        x, weight, bias, y = ctx.saved_tensors
        mult = 4 if rt.dtype == torch.float32 else 8
        dy = _aligned2d(dy.to(rt.dtype) if dy.dtype != rt.dtype else dy, mult)
        dres = dy if ctx.has_res else None
        if ctx.act == ops.RELU:
            # y = drop(relu(pre)): (y > 0) <=> relu passed and the element was kept
            assert not ctx.has_res, "relu with a fused residual cannot be differentiated from the saved output"
            scale = ops.keep_scale(ctx.p)
            dy_eff, a_drop_p_eff, a_site = ops.gate_mul(dy.contiguous(), y, scale), 0.0, 0
        else:
            dy_eff, a_drop_p_eff, a_site = dy, ctx.p, ctx.site
        dx = None
        if ctx.needs_input_grad[1]:
            dx = ops.gemm(dy_eff, rt.w(weight), b_kc=False, a_drop_p=a_drop_p_eff, a_drop_site=a_site, seed=rt.seed)
        if a_drop_p_eff == 0.0 and bias is not None:
            dw, db = _wbgrad(rt, weight, bias, dy_eff, x)
        else:
            dw = _wgrad(rt, weight, dy_eff, x, a_drop_p=a_drop_p_eff, a_drop_site=a_site)
            db = _bgrad(rt, bias, dy_eff, drop_p=a_drop_p_eff, drop_site=a_site) if bias is not None else None
        return None, dx, dw, db, None, dres, None, None, None


# --------------------------------------------------------------------------------------------------
class ResBlockFn(torch.autograd.Function):
    """z + relu(z W^T + b): one block of the latent ResNet (deepsvg/model/basic_blocks.py:59-65)"""

    @staticmethod
    def forward(ctx, rt, z, weight, bias):
        r = ops.gemm(z, rt.w(weight), bias=bias.detach(), act=ops.RELU)
        ctx.rt = rt
        ctx.save_for_backward(z, r, weight, bias)
        return ops.add(z, r)

    @staticmethod
    def backward(ctx, dout):
        rt = ctx.rt
        z, r, weight, bias = ctx.saved_tensors
        dout = dout.contiguous()
        dpre = ops.gate_mul(dout, r, 1.0)
        dw, db = _wbgrad(rt, weight, bias, dpre, z)
        dz = ops.gemm(dpre, rt.w(weight), b_kc=False, res=dout)
        return None, dz, dw, db


# the latent ResNet + the bottleneck linear as one launch per direction (csrc/group_stage.hip latent_chain_*; DSVG_LATENT_FUSED=0:
# a GEMM + an add per block forward, gate + two GEMMs per block backward)
LATENT_FUSED = os.environ.get("DSVG_LATENT_FUSED", "1") != "0"


class LatentChainFn(torch.autograd.Function):
    """z_i = z_{i-1} + relu(W_i z_{i-1} + b_i) for the n residual blocks, then out = W_out z_n + b_out (deepsvg/model/
    basic_blocks.py:59-65, model.py:193-198) - bf16, 256 features.  wb = (W_1, b_1, ..., W_n, b_n, W_out, b_out)."""

    @staticmethod
    def forward(ctx, rt, z, *wb):
        ws = [rt.w(w) for w in wb[0::2]]
        bs = [b.detach() for b in wb[1::2]]
        ctx.rt, ctx.n = rt, len(ws) - 1
        z = z.contiguous()
        if any(ctx.needs_input_grad):
            out, zs, rs = ops.latent_chain_fwd(z, ws, bs, train=True)
            ctx.save_for_backward(z, *zs, *rs, *wb)
        else:
            out = ops.latent_chain_fwd(z, ws, bs)
        return out

    @staticmethod
    def backward(ctx, dout):
        rt, n = ctx.rt, ctx.n
        saved = ctx.saved_tensors
        zin = [saved[0]] + list(saved[1:1 + n])          # the input of block i + 1; zin[n] feeds the final linear
        rs = list(saved[1 + n:1 + 2 * n])
        wb = saved[1 + 2 * n:]
        dout = dout.contiguous()
        dz0, dpre = ops.latent_chain_bwd(dout, [rt.w(w) for w in wb[0::2]], rs)
        # the n + 1 weight gradients (512 rows each): independent, a handful of workgroups each - one grouped launch
        grads = []
        with rt.grouping():
            for i in range(n + 1):
                w, b = wb[2 * i], wb[2 * i + 1]
                blocks = (8 * -(-w.shape[0] // 128) * -(-w.shape[1] // 128)) if GROUP_WGRAD else None
                grads += list(_wbgrad(rt, w, b, dpre[i] if i < n else dout, zin[i], blocks))
        return (None, dz0, *grads)


# --------------------------------------------------------------------------------------------------
class LivePrefix(tuple):
    """(n_live_sequences, n_live_rows) of a stage whose backward MAY be restricted to that row prefix: the rest of
    every incoming gradient is zero and the rest of every returned gradient is never read.  That holds only under a
    loss that ignores everything outside the prefix - deepsvg_amd.SVGLoss on the visible-first order of the second
    decoder stage (loss.py:36,51-54) - so the restriction is DISARMED (full backward, correct under any loss) until
    such a loss arms it on the very output it consumes (SVGLoss.forward: output["_dsvg_live"].armed = True)."""
    armed = False


def _live_rows(live, full_rows):
    """the LivePrefix to remember at forward time, or None when it would not shorten anything"""
    if live is None or live[1] >= full_rows:
        return None
    return live


def _armed(live):
    """backward-time decision: the remembered prefix if the loss armed it (plain tuples - tests - count as armed)"""
    if live is None or not getattr(live, "armed", True):
        return None
    return live


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rt, x, gamma, beta, eps, live=None, mask_below=None):
        """mask_below = (dropout rate, site): the producer of `x` reads its incoming gradient through this mask"""
        y, mean, rstd = ops.layernorm_fwd(x, gamma.detach(), beta.detach(), eps)
        ctx.rt, ctx.live = rt, _live_rows(live, x.shape[0])
        ctx.mask_below = mask_below if (LN_BWD_MASKED and mask_below is not None and rt.p(mask_below[0]) > 0) else None
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        rt = ctx.rt
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous()
        live = _armed(ctx.live)
        mk = None
        if ctx.mask_below is not None:
            mk = (rt.p(ctx.mask_below[0]), ctx.mask_below[1], rt.seed)
        with rt.deferring():
            if live is None:
                out = ops.layernorm_bwd(dy, x, mean, rstd, gamma.detach(),
                                        dgamma=rt.grad_out(gamma), dbeta=rt.grad_out(beta), masked=mk)
                dx, dg, db = out[:3]
            else:
                R = live[1]
                dx = torch.empty_like(x)
                out = ops.layernorm_bwd(dy[:R], x[:R], mean[:R], rstd[:R], gamma.detach(), dx=dx[:R],
                                        dgamma=rt.grad_out(gamma), dbeta=rt.grad_out(beta), masked=mk)
                dg, db = out[1:3]
        if mk is not None:
            _hand_masked(rt, dx, out[3], mk[0], mk[1])
        return None, dx, dg, db, None, None, None


# --------------------------------------------------------------------------------------------------
class AddPosFn(torch.autograd.Function):
    """y = drop(x + pos[:S]) with x optional (ConstEmbedding feeds zeros)."""

    @staticmethod
    def forward(ctx, rt, x, pos_weight, n_seq, S, drop_rate, site, live=None):
        p = rt.p(drop_rate)
        y = ops.add_pos_fwd(x, pos_weight.detach(), n_seq, S, rt.dtype, p, site, rt.seed)
        ctx.rt, ctx.n_seq, ctx.S, ctx.p, ctx.site = rt, n_seq, S, p, site
        ctx.has_x = x is not None
        ctx.live = _live_rows(live, n_seq * S)
        assert ctx.live is None or x is None, "live-prefix backward is only wired for the constant embedding"
        ctx.save_for_backward(pos_weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        rt = ctx.rt
        pos_weight, = ctx.saved_tensors
        dpos = rt.grad_out(pos_weight)
        if pos_weight.shape[0] > ctx.S:
            dpos[ctx.S:].zero_()
        live = _armed(ctx.live)
        n_seq = live[0] if live is not None else ctx.n_seq
        dy = dy.contiguous()[:n_seq * ctx.S]
        with rt.deferring_tables():     # (the table's column sums are a parameter gradient: reduced with the others at the flush)
            dx = ops.add_pos_bwd(dy, n_seq, ctx.S, dpos[:ctx.S], want_dx=ctx.has_x, drop_p=ctx.p,
                                 drop_site=ctx.site, seed=rt.seed)
        return None, dx, dpos, None, None, None, None, None


class MaskedMeanFn(torch.autograd.Function):
    """mask: 64-bit key bitmasks (dense layout) or None with seq_off (packed layout: every row of a sequence counts)"""

    @staticmethod
    def forward(ctx, rt, x, mask, n_seq, S, seq_off=None):
        ctx.n_seq, ctx.S, ctx.rows = n_seq, S, x.shape[0]
        ctx.save_for_backward(mask, seq_off)
        return ops.masked_mean_fwd(x, mask, n_seq, S, seq_off=seq_off)

    @staticmethod
    def backward(ctx, dout):
        mask, seq_off = ctx.saved_tensors
        dx = ops.masked_mean_bwd(dout.contiguous(), mask, ctx.n_seq, ctx.S, seq_off=seq_off, total_rows=ctx.rows)
        return None, dx, None, None, None, None


# --------------------------------------------------------------------------------------------------
class EmbedFn(torch.autograd.Function):
    """src = drop( embed_fcn(arg_embed[args+1]) + command_embed[cmd] (+ group_embed[grp]) + pos[s] )"""

    @staticmethod
    def forward(ctx, rt, commands, args, groups, n_seq, S, drop_rate, site,
                command_embed, arg_embed, fcn_w, fcn_b, pos_weight, group_embed):
        ge = group_embed.detach() if group_embed is not None else None
        A, R = ops.embed_gather(commands, args, command_embed.detach(), arg_embed.detach(), rt.dtype, ge, groups)
        pre = ops.gemm(A, rt.w(fcn_w), bias=fcn_b.detach(), res=R, res_pre=True)
        p = rt.p(drop_rate)
        src = ops.add_pos_fwd(pre, pos_weight.detach(), n_seq, S, rt.dtype, p, site, rt.seed)
        ctx.rt, ctx.n_seq, ctx.S, ctx.p, ctx.site = rt, n_seq, S, p, site
        ctx.save_for_backward(commands, args, groups, A, command_embed, arg_embed, fcn_w, fcn_b, pos_weight,
                              group_embed)
        return src

    @staticmethod
    def backward(ctx, dsrc):
        rt = ctx.rt
        commands, args, groups, A, command_embed, arg_embed, fcn_w, fcn_b, pos_weight, group_embed = ctx.saved_tensors
        dpos = rt.grad_out(pos_weight)
        if pos_weight.shape[0] > ctx.S:
            dpos[ctx.S:].zero_()
        with rt.deferring_tables():
            dpre = ops.add_pos_bwd(dsrc.contiguous(), ctx.n_seq, ctx.S, dpos[:ctx.S], want_dx=True, drop_p=ctx.p,
                                   drop_site=ctx.site, seed=rt.seed)
        dw, db = _wbgrad(rt, fcn_w, fcn_b, dpre, A)
        dA = ops.gemm(dpre, rt.w(fcn_w), b_kc=False)
        d_arg = rt.grad_out(arg_embed)
        d_cmd = rt.grad_out(command_embed)
        d_grp = rt.grad_out(group_embed) if group_embed is not None else None
        with rt.deferring_tables():     # (three table gradients from per-workgroup partials)
            ops.embed_scatter(commands, args, dA, dpre, d_arg, d_cmd, groups, d_grp)
        return (None, None, None, None, None, None, None, None, d_cmd, d_arg, dw, db, dpos, d_grp)


class GatherGroupsFn(torch.autograd.Function):
    """y[g*S + s] = x[idx[g]*S + s] (whole sequences), g < n_groups; inv = inverse permutation.  x may hold FEWER sequences
    than the permutation has (a stage that ran on the leading sequences only): the others come out as zero rows; and
    n_groups may be smaller than the permutation (only its leading sequences are wanted): the gradient of the sequences
    of x that nobody read is zero.  With `live` = (n_live, rows): the upstream gradient of the sequences inv[g >= n_live]
    is known to be zero, so backward gathers only the sequences that cover the first `rows` rows (the exact zeros
    included) and leaves the rest of dx unwritten (never read)."""

    @staticmethod
    def forward(ctx, x, idx, inv, n_groups, S, live=None):
        ctx.n_groups, ctx.S, ctx.live = n_groups, S, _live_rows(live, n_groups * S)
        ctx.n_src = x.shape[0] // S
        ctx.save_for_backward(inv)
        return ops.gather_groups(x.contiguous(), idx, n_groups, S, n_src=ctx.n_src)

    @staticmethod
    def backward(ctx, dy):
        inv, = ctx.saved_tensors
        dy = dy.contiguous()
        live = _armed(ctx.live)
        n_back = ctx.n_src
        if live is not None:
            n_back = min(n_back, -(-live[1] // ctx.S))
        dx = torch.empty((ctx.n_src * ctx.S, dy.shape[1]), dtype=dy.dtype, device=dy.device)
        ops.gather_groups(dy, inv, n_back, ctx.S, out=dx, n_src=ctx.n_groups)
        return dx, None, None, None, None, None


class LabelEmbedFn(torch.autograd.Function):
    """LabelEmbedding.forward (deepsvg/model/model.py:87-89): one row of the (n_labels, dim_label) table per icon.
    An N x dim_label lookup and its scatter-add are left to torch (not a hot op); the table's gradient is written
    into the flat gradient buffer like every other parameter gradient.  Apply once per table and forward pass."""

    @staticmethod
    def forward(ctx, rt, label, weight):
        idx = label.reshape(-1).long()
        ctx.rt = rt
        ctx.save_for_backward(idx, weight)
        return rt.w(weight).index_select(0, idx).contiguous()

    @staticmethod
    def backward(ctx, dl):
        idx, weight = ctx.saved_tensors
        out = ctx.rt.grad_out(weight)
        out.zero_()
        out.index_add_(0, idx, dl.to(torch.float32))
        return None, None, out


class PackedEmbedFn(torch.autograd.Function):
    """SVGEmbedding on the packed token layout of the first encoder stage (ops.pack_tokens):
         src[r] = drop( embed_fcn(arg_embed[args_r + 1]) + command_embed[cmd_r] + pos[pos_r] )
    The positional table rides in the gather/scatter kernels' per-token index slot (free here: the two-stage encoder
    has no group embedding, deepsvg/model/model.py:105,119-122) and the dropout runs in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, rt, commands, args, pos_idx, S, drop_rate, site, command_embed, arg_embed, fcn_w, fcn_b, pos_weight):
        pos_tab = pos_weight.detach()[:S].contiguous()
        A, R = ops.embed_gather(commands, args, command_embed.detach(), arg_embed.detach(), rt.dtype, pos_tab, pos_idx)
        p = rt.p(drop_rate)
        src = ops.gemm(A, rt.w(fcn_w), bias=fcn_b.detach(), res=R, res_pre=True, drop_p=p, drop_site=site, seed=rt.seed)
        ctx.rt, ctx.S, ctx.p, ctx.site = rt, S, p, site
        ctx.save_for_backward(commands, args, pos_idx, A, command_embed, arg_embed, fcn_w, fcn_b, pos_weight)
        return src

    @staticmethod
    def backward(ctx, dsrc):
        rt = ctx.rt
        commands, args, pos_idx, A, command_embed, arg_embed, fcn_w, fcn_b, pos_weight = ctx.saved_tensors
        dpre = ops.drop_apply(dsrc.contiguous(), ctx.p, ctx.site, rt.seed)
        dw, db = _wbgrad(rt, fcn_w, fcn_b, dpre, A)
        dA = ops.gemm(dpre, rt.w(fcn_w), b_kc=False)
        d_arg = rt.grad_out(arg_embed)
        d_cmd = rt.grad_out(command_embed)
        dpos = rt.grad_out(pos_weight)
        if pos_weight.shape[0] > ctx.S:
            dpos[ctx.S:].zero_()
        with rt.deferring_tables():
            ops.embed_scatter(commands, args, dA, dpre, d_arg, d_cmd, pos_idx, dpos[:ctx.S])
        return (None, None, None, None, None, None, None, d_cmd, d_arg, dw, db, dpos)


# --------------------------------------------------------------------------------------------------
GLOBAL_COND_CAT = os.environ.get("DSVG_GLOBAL_COND_CAT", "1") != "0"


def _adjacent_columns(ts):
    """ts = the column blocks 0 .. n-1 of ONE row-major [rows, n * w] buffer (in order) -> that buffer as a tensor, else None"""
    t0 = ts[0]
    n = len(ts)
    if t0.dim() != 2 or t0.stride(1) != 1:
        return None
    rows, w = t0.shape
    es = t0.element_size()
    for i, t in enumerate(ts):
        if (t.dim() != 2 or tuple(t.shape) != (rows, w) or t.dtype != t0.dtype or t.stride(1) != 1 or t.stride(0) != n * w
                or t.data_ptr() != t0.data_ptr() + i * w * es
                or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr()):
            return None
    return torch.as_strided(t0, (rows, n * w), (n * w, 1), t0.storage_offset())


class GlobalCondFn(torch.autograd.Function):
    """The conditioning rows of a whole decoder stack at once: g_l = linear_global_l(z) for every layer l
    (layers/improved_transformer.py:131-132; z does not change between the layers).  Hoisted out of the layers so that the
    gradient of z is accumulated by the input-gradient GEMMs themselves (no separate sums of per-layer contributions) and
    the layers' weight-gradient GEMMs run as one grouped launch."""

    @staticmethod
    def forward(ctx, rt, z, *wb):
        pairs = list(zip(wb[0::2], wb[1::2]))
        ctx.rt, ctx.n = rt, len(pairs)
        ctx.save_for_backward(z, *wb)
        st = rt.store
        if (GLOBAL_COND_CAT and z.dtype == torch.bfloat16 and len(pairs) > 1 and st is not None
                and all(id(t) in st.index for pr in pairs for t in pr) and len({tuple(w.shape) for w, _b in pairs}) == 1):
            # the stack's weights / biases are adjacent in the flat buffers (ParamStore._grouped_order): ONE product
            # z [rows, 256] x Wcat [n * 256, 256]^T -> [rows, n * 256]; layer l reads its column block (row stride n * 256:
            # the fused kernels take it as seq_add_ld)
            ow = [st.index[id(w)] for w, _b in pairs]
            ob = [st.index[id(b)] for _w, b in pairs]
            ws = [rt.w(w) for w, _b in pairs]
            w0, b0 = ws[0], pairs[0][1].detach()
            # (adjacent master offsets AND adjacent views of the low-precision image: rt.w may hand out a private copy)
            if (all(ow[i + 1][0] == ow[i][0] + ow[i][1] and ob[i + 1][0] == ob[i][0] + ob[i][1] for i in range(len(pairs) - 1))
                    and all(w.is_contiguous() for w in ws) and b0.is_contiguous()
                    and all(ws[i + 1].data_ptr() == ws[i].data_ptr() + ws[i].numel() * ws[i].element_size()
                            for i in range(len(ws) - 1))):
                n_out, k_in = sum(w.shape[0] for w, _b in pairs), w0.shape[1]
                gcat = ops.gemm(z, torch.as_strided(w0, (n_out, k_in), (k_in, 1)), bias=torch.as_strided(b0, (n_out,), (1,)))
                d = w0.shape[0]
                return tuple(gcat[:, i * d:(i + 1) * d] for i in range(len(pairs)))
        return tuple(ops.gemm(z, rt.w(w), bias=b.detach()) for w, b in pairs)

    @staticmethod
    def backward(ctx, *dgs):
        rt = ctx.rt
        z, *wb = ctx.saved_tensors
        pairs = list(zip(wb[0::2], wb[1::2]))
        grads = []
        small = z.shape[0] < FFN_MIN_ROWS
        cat_ok = GLOBAL_COND_CAT and z.dtype == torch.bfloat16 and len(pairs) > 1 and all(dg.dtype == z.dtype for dg in dgs)
        # GsStackFn hands the gradients over as the column blocks of ONE buffer: that buffer is the concatenation (no launch)
        dgcat = _adjacent_columns(dgs) if cat_ok else None
        raw, dgs = dgs, None
        if dgcat is not None:       # (the stack's shared buffer, if that is what this is, has done its job)
            for k in [k for k, v in rt.cond_grad.items() if v.data_ptr() == dgcat.data_ptr()]:
                del rt.cond_grad[k]

        def dg_blocks():            # the per-layer gradients as tensors of their own (the paths that do not use the concatenation)
            return [dg.contiguous() for dg in raw]
        if cat_ok and dgcat is None:
            dgcat = torch.cat(dg_blocks(), 1)
        merged = None
        st = rt.store
        if cat_ok and st is not None and all(id(t) in st.index for pr in pairs for t in pr):
            # the stack's weights (and biases) sit next to each other in the flat buffers (ParamStore._grouped_order): their
            # gradients are ONE [n * 256, 256] weight-gradient product of the concatenated dg with z (+ its row sums)
            ow = [st.index[id(w)] for w, _b in pairs]
            ob = [st.index[id(b)] for _w, b in pairs]
            adj = all(ow[i + 1][0] == ow[i][0] + ow[i][1] and ob[i + 1][0] == ob[i][0] + ob[i][1] for i in range(len(pairs) - 1))
            if adj:
                dws = [rt.grad_out(w) for w, _b in pairs]
                dbs = [rt.grad_out(b) for _w, b in pairs]
                # (grad_out may hand out private tensors in corner cases - a second gradient for the same parameter: checked)
                adj = all(dws[i + 1].data_ptr() == dws[i].data_ptr() + dws[i].numel() * 4 and
                          dbs[i + 1].data_ptr() == dbs[i].data_ptr() + dbs[i].numel() * 4 for i in range(len(pairs) - 1))
                n_out = sum(w.shape[0] for w, _b in pairs)
                k_in = pairs[0][0].shape[1]
                if adj:
                    dwc = torch.as_strided(dws[0], (n_out, k_in), (k_in, 1))
                    dbc = torch.as_strided(dbs[0], (n_out,), (1,))
                else:       # (not reached by any shipped config: one product into a temporary, then n copies)
                    dwc = torch.empty((n_out, k_in), dtype=torch.float32, device=z.device)
                    dbc = torch.empty(n_out, dtype=torch.float32, device=z.device)
                split = ops.split_k_for(n_out, k_in, z.shape[0])
                with (rt.deferring() if adj else _NULL_CTX), _wgrad_tag():
                    if split > 1:
                        ops.gemm(dgcat, z, a_kc=False, b_kc=False, out=dwc, split_k=split, rowsum=dbc)
                    else:
                        ops.gemm(dgcat, z, a_kc=False, b_kc=False, out=dwc)
                        ops.colsum(dgcat, out=dbc)
                if not adj:
                    r0 = 0
                    for (w, _b), dw, db in zip(pairs, dws, dbs):
                        dw.view(w.shape).copy_(dwc[r0:r0 + w.shape[0]])
                        db.copy_(dbc[r0:r0 + w.shape[0]])
                        r0 += w.shape[0]
                merged = [t for dw, db in zip(dws, dbs) for t in (dw, db)]
        if merged is not None:
            grads = merged
        else:
            with (rt.grouping() if small else _NULL_CTX):
                for (w, b), dg in zip(pairs, dg_blocks()):
                    blocks = (8 * -(-w.shape[0] // 128) * -(-w.shape[1] // 128)) if (small and GROUP_WGRAD) else None
                    grads += list(_wbgrad(rt, w, b, dg, z, blocks))
        dz = None
        if ctx.needs_input_grad[1]:
            if cat_ok:
                # dz = sum_l dg_l W_l as ONE product over the concatenated reduction dimension, [rows, n * 256] x [n * 256, 256]:
                # two concatenations + one LDS-DMA GEMM instead of n accumulating launches of the register-staged kernel
                # (13-21 us each on <= 4096 rows: 65 us per stack in the round-4 trace)
                # (the bf16 images of the stack's weights are adjacent like the masters: the [n * 256, 256] operand is a view)
                w0 = rt.w(pairs[0][0])
                wcat = None
                if st is not None and all(id(w) in st.index for w, _b in pairs) and w0.is_contiguous():
                    ow = [st.index[id(w)] for w, _b in pairs]
                    ws = [rt.w(w) for w, _b in pairs]
                    if (all(ow[i + 1][0] == ow[i][0] + ow[i][1] for i in range(len(pairs) - 1))
                            and all(ws[i + 1].data_ptr() == ws[i].data_ptr() + ws[i].numel() * ws[i].element_size()
                                    for i in range(len(pairs) - 1))):
                        wcat = torch.as_strided(w0, (sum(w.shape[0] for w, _b in pairs), w0.shape[1]), (w0.shape[1], 1))
                if wcat is None:
                    wcat = torch.cat([rt.w(w) for w, _b in pairs], 0)
                dz = ops.gemm(dgcat, wcat, b_kc=False)
            else:
                dz = torch.empty_like(z)
                for i, ((w, _b), dg) in enumerate(zip(pairs, dg_blocks())):
                    ops.gemm(dg, rt.w(w), b_kc=False, out=dz, accumulate=i > 0)
        return (None, dz, *grads)


class LayerFn(torch.autograd.Function):
    """One pre-LN transformer block.  With z: the 'global' decoder block (x += linear_global(z) broadcast over
    the sequence); with l: the label-conditioned variant (x += linear_global2(l)).
    sites: 5 consecutive dropout site ids starting at `site0`:
      +0 attention probabilities, +1 attention residual, +2 linear_global, +3 FFN hidden, +4 FFN residual,
      +5 linear_global2
    """

    @staticmethod
    def forward(ctx, rt, x, key_mask, z, l, n_seq, S, n_heads, drop_rate, site0,
                n1w, n1b, win, bin_, wo, bo, n2w, n2b, w1, b1, w2, b2, wg, bg, wg2, bg2, seq_off=None, live=None,
                tiles=None, causal=False, mask_below=None, first_in_stack=True):
        """mask_below: site of the FFN residual dropout of the layer that produced `x` (same rate), or None;
        first_in_stack: layer 0 of its stack (the last one of the backward pass: it closes the stack's grouped launch)"""
        p = rt.p(drop_rate)
        ctx.mask_below = mask_below if (LN_BWD_MASKED and mask_below is not None and p > 0) else None
        ctx.first_in_stack = bool(first_in_stack)
        rt.last_layer_gs = False
        d = x.shape[1]
        scale = float(d // n_heads) ** -0.5
        want_bwd = any(ctx.needs_input_grad)         # (grad mode itself is off inside Function.forward)
        ctx.gs = False
        gs = None
        if (rt.store is not None and x.dtype == torch.bfloat16 and not causal and seq_off is None and l is None
                and n_heads == 8 and d == 256 and 32 % S == 0 and x.shape[0] == n_seq * S
                and x.shape[0] < min(ATTN_MIN_ROWS, FFN_MIN_ROWS) and live is None
                and (key_mask is None or key_mask.dtype == torch.int64)):
            gs = rt.store.gs(win)
        if gs is not None:
            # the short-sequence ("group") stages: the whole block in ONE launch (csrc/group_stage.hip); with a backward pass
            # ahead it stores exactly what the unfused launches below would have saved
            # (wg None: `z` already is the projected conditioning row linear_global(z), see GlobalCondFn)
            g = (z if wg is None else ops.gemm(z, rt.w(wg), bias=bg.detach())) if z is not None else None
            with ops.tag("gs"):
                res = ops.gs_layer_fwd(x, gs[0], bin_.detach(), bo.detach(), b1.detach(), b2.detach(), n1w.detach(),
                                       n1b.detach(), n2w.detach(), n2b.detach(), key_mask, n_seq, S, scale, 1e-5, p, site0,
                                       rt.seed, seq_add=g, train=want_bwd)
            if want_bwd:
                x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h = res
            else:
                x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h = (res,) + (None,) * 10
            ctx.gs, ctx.ffn_fused, ctx.tiles, ctx.causal, ctx.live = True, False, None, False, None
            rt.last_layer_gs = True     # (its backward kernel applies the mask itself: nobody upstream needs to prepare one)
            ctx.rt, ctx.n_seq, ctx.S, ctx.n_heads, ctx.p, ctx.site0, ctx.scale = rt, n_seq, S, n_heads, p, site0, scale
            ctx.save_for_backward(x, key_mask, z, l, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h,
                                  n1w, n1b, win, bin_, wo, bo, n2w, n2b, w1, b1, w2, b2, wg, bg, wg2, bg2, seq_off)
            return x2
        att = None
        # (p <= 0.5: the fused kernel's packed 16-bit dropout code, csrc/attn_fused.hip; a higher rate takes the unfused launches)
        if (rt.store is not None and x.dtype == torch.bfloat16 and x.shape[0] >= ATTN_MIN_ROWS and S <= 32 and not causal
                and n_heads == 8 and d == 256 and (seq_off is None or tiles is not None) and p <= 0.5
                and (key_mask is None or key_mask.dtype == torch.int64)):
            att = rt.store.attn(win)
        z_fused = False
        split = 0
        if (att is not None and want_bwd and GS_REMAINDER > 0 and seq_off is None and key_mask is None and l is None
                and x.shape[0] == n_seq * S and x.shape[0] >= FFN_MIN_ROWS and n_seq > SEQ_ROUND):
            # The fused kernels of the large stages own a CU per workgroup (8 sequences / 256 rows each): a launch of 2048 k + r
            # sequences pays k + 1 full rounds of the chip for any r > 0.  The training forward therefore runs the first
            # 2048 k sequences on them and the remainder on the one-tile-per-workgroup layer kernel of the group stages
            # (csrc/group_stage.hip, one launch for the whole block), both writing row slices of the same saved tensors.
            rem = n_seq % SEQ_ROUND
            if 0 < rem <= GS_REMAINDER and rt.store.gs(win) is not None and rt.store.ffn(w1) is not None:
                split = n_seq - rem
        if split:
            gsr, ffn = rt.store.gs(win), rt.store.ffn(w1)
            T, R0 = x.shape[0], split * S
            bf = lambda w_: torch.empty((T, w_), dtype=x.dtype, device=x.device)
            f32 = lambda n_: torch.empty(n_, dtype=torch.float32, device=x.device)
            x1, xn1, qkv, ao, mean1, rstd1 = bf(256), bf(256), bf(768), bf(256), f32(T), f32(T)
            x2, h, xn2 = bf(256), bf(512), bf(256)
            g = None
            if z is not None:
                g = z if wg is None else ops.gemm(z, rt.w(wg), bias=bg.detach())
                z_fused = True
            with ops.tag("attn"):
                ops.attn_block_fwd(x[:R0], att, bin_.detach(), bo.detach(), n1w.detach(), n1b.detach(), None, split, S, scale,
                                   1e-5, p, site0, site0 + 1, rt.seed, train=True, seq_add=None if g is None else g[:split],
                                   site_seq_add=site0 + 2,
                                   into=(x1[:R0], xn1[:R0], qkv[:R0], ao[:R0], mean1[:R0], rstd1[:R0]))
            with ops.tag("ffn"):
                ops.ffn_fwd(x1[:R0], ffn[0], ffn[2], b2.detach(), 1e-5, p, site0 + 3, site0 + 4, rt.seed, out=x2[:R0],
                            train=True, into=(h[:R0], xn2[:R0]))
            with ops.tag("gs"):
                ops.gs_layer_fwd(x[R0:], gsr[0], bin_.detach(), bo.detach(), b1.detach(), b2.detach(), n1w.detach(),
                                 n1b.detach(), n2w.detach(), n2b.detach(), None, n_seq - split, S, scale, 1e-5, p, site0,
                                 rt.seed, seq_add=None if g is None else g[split:], train=True, seq_base=split, ffn_format=True,
                                 into=(x2[R0:], mean1[R0:], rstd1[R0:], xn1[R0:], qkv[R0:], ao[R0:], x1[R0:], f32(T - R0),
                                       f32(T - R0), xn2[R0:], h[R0:]))
            mean2 = rstd2 = None
        elif att is not None:
            # one launch: LayerNorm, in_proj, the 8 heads, out_proj, dropout, residual (csrc/attn_fused.hip) - and the
            # decoder's per-sequence conditioning add when the layout is dense and every row belongs to a sequence.  With a
            # backward pass ahead it also stores LN(x), q|k|v, the head outputs and the row statistics
            g = None
            if z is not None and seq_off is None and x.shape[0] == n_seq * S:
                g = z if wg is None else ops.gemm(z, rt.w(wg), bias=bg.detach())
                z_fused = True
            with ops.tag("attn"):
                res = ops.attn_block_fwd(x, att, bin_.detach(), bo.detach(), n1w.detach(), n1b.detach(), key_mask, n_seq,
                                         S, scale, 1e-5, p, site0, site0 + 1, rt.seed, seq_off=seq_off, tiles=tiles,
                                         train=want_bwd, seq_add=g, site_seq_add=site0 + 2)
            if want_bwd:
                x1, xn1, qkv, ao, mean1, rstd1 = res
            else:
                x1, xn1, qkv, ao, mean1, rstd1 = res, None, None, None, None, None
        else:
            xn1, mean1, rstd1 = ops.layernorm_fwd(x, n1w.detach(), n1b.detach())
            qkv = ops.gemm(xn1, rt.w(win), bias=bin_.detach())
            if causal:
                ao = ops.attention_fwd(qkv, key_mask, n_seq, S, n_heads, scale, p, site0, rt.seed, causal=True)
            else:
                ao = ops.attention_fwd(qkv, key_mask, n_seq, S, n_heads, scale, p, site0, rt.seed, seq_off=seq_off,
                                       tiles=tiles)
            x1 = ops.gemm(ao, rt.w(wo), bias=bo.detach(), res=x, drop_p=p, drop_site=site0 + 1, seed=rt.seed)
        ctx.tiles, ctx.causal = tiles, causal
        if z is not None and not z_fused:
            g = z.contiguous() if wg is None else ops.gemm(z, rt.w(wg), bias=bg.detach())      # (z: maybe a column block)
            ops.bcast_add_fwd_(x1, g, n_seq, S, p, site0 + 2, rt.seed)
        if l is not None:
            g2 = ops.gemm(l, rt.w(wg2), bias=bg2.detach())
            ops.bcast_add_fwd_(x1, g2, n_seq, S, p, site0 + 5, rt.seed)
        ffn = rt.store.ffn(w1) if (rt.store is not None and x.dtype == torch.bfloat16 and x.shape[0] >= FFN_MIN_ROWS) else None
        ctx.ffn_fused = ffn is not None
        if split:
            pass                    # (x2, h, xn2 are there already)
        elif ffn is not None:
            # one launch: LayerNorm (folded into the packed linear1), linear1, ReLU, dropout, linear2, dropout, residual
            # (csrc/ffn_fused.hip).  With a backward pass ahead it also stores h (fragment-ordered columns) and the
            # normalised rows xh; the inference call stores nothing but the result.
            mean2 = rstd2 = None
            with ops.tag("ffn"):
                if want_bwd and not FFN_BWD_FUSED:
                    x2, h, xn2, _rstd = ops.ffn_fwd(x1, ffn[0], ffn[2], b2.detach(), 1e-5, p, site0 + 3, site0 + 4, rt.seed,
                                                    train=True)
                else:
                    xn2 = h = None
                    x2 = ops.ffn_fwd(x1, ffn[0], ffn[2], b2.detach(), 1e-5, p, site0 + 3, site0 + 4, rt.seed)
        else:
            with ops.tag("ffn"):        # (norm2 belongs to the FFN sub-block: the fused kernel contains it)
                xn2, mean2, rstd2 = ops.layernorm_fwd(x1, n2w.detach(), n2b.detach())
                h = ops.gemm(xn2, rt.w(w1), bias=b1.detach(), act=ops.RELU, drop_p=p, drop_site=site0 + 3, seed=rt.seed)
                x2 = ops.gemm(h, rt.w(w2), bias=b2.detach(), res=x1, drop_p=p, drop_site=site0 + 4, seed=rt.seed)
        ctx.rt, ctx.n_seq, ctx.S, ctx.n_heads, ctx.p, ctx.site0, ctx.scale = rt, n_seq, S, n_heads, p, site0, scale
        assert seq_off is None or (z is None and l is None), "packed layout: no per-sequence conditioning adds"
        ctx.live = _live_rows(live, x.shape[0])
        assert ctx.live is None or (seq_off is None and key_mask is None and l is None)
        ctx.save_for_backward(x, key_mask, z, l, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h,
                              n1w, n1b, win, bin_, wo, bo, n2w, n2b, w1, b1, w2, b2, wg, bg, wg2, bg2, seq_off)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        rt, n_seq, S, H, p, s0 = ctx.rt, ctx.n_seq, ctx.S, ctx.n_heads, ctx.p, ctx.site0
        (x, key_mask, z, l, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h,
         n1w, n1b, win, bin_, wo, bo, n2w, n2b, w1, b1, w2, b2, wg, bg, wg2, bg2, seq_off) = ctx.saved_tensors
        dx2 = dx2.contiguous()
        full_rows, n_seq_full = x.shape[0], n_seq
        live = _armed(ctx.live)
        # the incoming gradient with this layer's FFN-residual mask already on it, if its producer made one (LN_BWD_MASKED)
        dx2_masked = _take_masked(rt, dx2, p, s0 + 4, rows=(live[1] if live is not None else None)) if p > 0 else None
        if live is not None:            # backward over the live row prefix only (visible-first decoder order)
            n_seq, R = live
            (dx2, x, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h) = (
                (t[:R] if t is not None else None) for t in (dx2, x, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h))
        inv_keep = ops.keep_scale(p)
        dx1m = None
        if ctx.gs:
            # one launch for the whole input-gradient chain of the block (csrc/group_stage.hip); it hands over the token-major
            # operands of the four weight-gradient GEMMs and the LayerNorm parameter gradients
            gs = rt.store.gs(win)
            fuse_dg = GS_BWD_DG and z is not None      # the conditioning term's gradient from the same launch (no dx1 round trip)
            with rt.deferring(), ops.tag("gs"):
                (dx, dx1, dym, dpre, dx1m, dqkv, dn2w, dn2b, dn1w, dn1b, *dg_) = ops.gs_layer_bwd(
                    dx2, gs[1], x, mean1, rstd1, qkv, x1, mean2, rstd2, h, n1w.detach(), n2w.detach(), key_mask, n_seq, S,
                    ctx.scale, p, s0, rt.seed, want_dx1=z is not None and not fuse_dg, dgamma2=rt.grad_out(n2w),
                    dbeta2=rt.grad_out(n2b), dgamma1=rt.grad_out(n1w), dbeta1=rt.grad_out(n1b), want_dg=fuse_dg)
            dz = dwg = dbg = dg = None
            if z is not None:
                dg = dg_[0] if fuse_dg else ops.bcast_add_bwd(dx1, n_seq, S, p, s0 + 2, rt.seed)
                if wg is None:
                    dz = dg                 # `z` was the projected row itself: its gradient goes to GlobalCondFn
                elif ctx.needs_input_grad[3]:
                    dz = ops.gemm(dg, rt.w(wg), b_kc=False)
            # the layer's weight-gradient GEMMs: independent of each other, 64-256 workgroups each - one grouped launch
            # (32 - 36 output tiles of 128 x 128 between them: 8 token slices each fill the chip once, together)
            nsl = STACK_GROUP_SLICES if (STACK_GROUP and GROUP_WGRAD and rt.defer) else 8
            gb = (lambda w_: nsl * -(-w_.shape[0] // 128) * -(-w_.shape[1] // 128)) if GROUP_WGRAD else (lambda w_: None)
            # STACK_GROUP (round 5): the products of ALL the stack's layers wait for one launch (16 of them for 4 layers: the
            # launch's table size) instead of one launch per layer.  Their results are only read after the trainer's flush
            # (rt.defer), their operands are kept alive by the list until the launch
            keep = rt.stack_group_begin() if (STACK_GROUP and GROUP_WGRAD and rt.defer) else None
            with (rt.grouping() if keep is None else _NULL_CTX):
                with ops.tag("ffn"):
                    dw2, db2 = _wbgrad(rt, w2, b2, dym, h, gb(w2))
                    dw1, db1 = _wbgrad(rt, w1, b1, dpre, xn2, gb(w1))
                if z is not None and wg is not None:
                    dwg, dbg = _wbgrad(rt, wg, bg, dg, z, gb(wg))
                dwo, dbo = _wbgrad(rt, wo, bo, dx1m, ao, gb(wo))
                dwin, dbin = _wbgrad(rt, win, bin_, dqkv, xn1, gb(win))
            if keep is not None:
                keep += [dym, h, dpre, xn2, dx1m, ao, dqkv, xn1, dg, z]
                if ctx.first_in_stack:
                    rt.stack_group_end()
            return (None, dx, None, dz, None, None, None, None, None, None,
                    dn1w, dn1b, dwin, dbin, dwo, dbo, dn2w, dn2b, dw1, db1, dw2, db2, dwg, dbg, None, None, None, None, None,
                    None, None, None)
        # (round 3 measured, round 4 removed: the layer's four token-reducing weight-gradient GEMMs as one grouped launch at the
        # end of its backward pass - slower, each product right behind the launch that wrote its operand hits the memory-side
        # cache - and the same GEMMs queued onto a second stream beside the group stages - slower inside the step's hipGraph)
        if ctx.ffn_fused:
            pb, b1f, w2p = rt.store.ffn(w1)[1:]
            T = x1.shape[0]
            with ops.tag("ffn"):
                g2p = torch.empty((256, 512), dtype=torch.float32, device=x1.device)
                g1p = torch.empty((512, 256), dtype=torch.float32, device=x1.device)
                db1p = torch.empty(512, dtype=torch.float32, device=x1.device)
                db2 = rt.grad_out(b2)
                s2, s1 = ops.split_k_for(256, 512, T), ops.split_k_for(512, 256, T)

                def wgrad2(dym, hp):        # G2p = dym^T h (fragment-ordered columns), db2 = its row sums
                    with rt.deferring():
                        if s2 > 1:
                            ops.gemm(dym, hp, a_kc=False, b_kc=False, out=g2p, split_k=s2, rowsum=db2)
                        else:
                            ops.gemm(dym, hp, a_kc=False, b_kc=False, out=g2p)
                            ops.colsum(dym, out=db2)

                def wgrad1(dpre, xh):       # G1p = dpre^T xh, db1' = its row sums
                    with rt.deferring():
                        if s1 > 1:
                            ops.gemm(dpre, xh, a_kc=False, b_kc=False, out=g1p, split_k=s1, rowsum=db1p)
                        else:
                            ops.gemm(dpre, xh, a_kc=False, b_kc=False, out=g1p)
                            ops.colsum(dpre, out=db1p)

                if h is None:
                    # fully fused variant (opt-in, DSVG_FFN_BWD_FUSED=1): hidden tile recomputed from x1, both dropout
                    # masks replayed in the kernel; measured slower than the default below (it writes h, dpre, xh AND dym)
                    dx1, hp, dpre, xh, dym = ops.ffn_bwd(x1, dx2, pb, b1f, 1e-5, p, s0 + 3, s0 + 4, rt.seed)
                    wgrad2(dym, hp)
                    wgrad1(dpre, xh)
                else:
                    # default: the forward kernel stored h (fragment order) and xh.  dym = residual mask replayed once;
                    # dpre = (dym . W2p) gated by h (h > 0 <=> ReLU passed AND kept) in one GEMM; dx by the fused kernel
                    # (dpre . W1' with the LayerNorm backward in its epilogue).  Each weight-gradient GEMM runs right
                    # behind the launch that produced its token-major operand (dym, dpre: 65 / 130 MB that are then still
                    # partly in the memory-side cache) instead of at the end
                    hp, xh = h, xn2
                    # (round 4, MI355X: the three launches below as ONE kernel - dsvg_ffn_bwd_one, the mirror image of ffn_fwd -
                    # were bit-identical and 13 % faster in isolation (93 -> 80 us at 63 k rows) but 0.4 % SLOWER inside the
                    # step, where dym already comes from the masked bcast_add_bwd and dpre is read back from the memory-side
                    # cache; removed.  profiles/r04_experimental_ffn_bwd_one.log, r04_experimental_ab.log)
                    dym = dx2_masked if dx2_masked is not None else ops.drop_apply(dx2, p, s0 + 4, rt.seed)
                    if FFN_BWD_ORDER:
                        wgrad2(dym, hp)
                    dpre = ops.gemm(dym, w2p, b_kc=False, gate=hp, gate_scale=inv_keep)
                    if FFN_BWD_ORDER:
                        wgrad1(dpre, xh)
                    # (the same launch also hands over dx1 with the attention residual's dropout mask replayed on it)
                    if FFN_BWD_MASKED == 1 or (FFN_BWD_MASKED == 2 and z is None and p > 0):
                        dx1, dx1m = ops.ffn_bwd_dx(dpre, x1, dx2, pb, masked=(p, s0 + 1, rt.seed))
                    else:
                        dx1 = ops.ffn_bwd_dx(dpre, x1, dx2, pb)
                    if not FFN_BWD_ORDER:
                        wgrad2(dym, hp)
                        wgrad1(dpre, xh)
                dw1, db1, dw2 = rt.grad_out(w1), rt.grad_out(b1), rt.grad_out(w2)
                dn2w, dn2b = rt.grad_out(n2w), rt.grad_out(n2b)
                # (aliases of the gradient tensors: AccumulateGrad adopts a returned gradient only while nobody else
                # holds a reference to that tensor object, otherwise it clones it - here before it is even written)
                fin = (g1p, db1p, g2p, w1.detach(), n2w.detach(), n2b.detach(), dw1.detach(), db1.detach(), dw2.detach(),
                       dn2w.detach(), dn2b.detach())
                if rt.defer:
                    ops.ffn_wgrad_finish_deferred(*fin)     # reads the queued reductions' outputs: runs right after the
                else:                                       # flush, one launch for all the layers of the backward pass
                    ops.ffn_wgrad_finish(*fin)
            del hp, dpre, xh, dym
        else:
            # ---- FFN: x2 = x1 + drop4(h W2^T + b2),  h = drop3(relu(xn2 W1^T + b1)) ----
            # the mask of the residual dropout is replayed ONCE into dx2m; the three consumers read plain data
            with ops.tag("ffn"):
                dx2m = dx2_masked if dx2_masked is not None else ops.drop_apply(dx2, p, s0 + 4, rt.seed)
                dw2, db2 = _wbgrad(rt, w2, b2, dx2m, h)
                dh = ops.gemm(dx2m, rt.w(w2), b_kc=False, gate=h, gate_scale=inv_keep)   # (h > 0) <=> relu passed AND kept
                dw1, db1 = _wbgrad(rt, w1, b1, dh, xn2)
                dxn2 = ops.gemm(dh, rt.w(w1), b_kc=False)
                with rt.deferring():
                    dx1, dn2w, dn2b = ops.layernorm_bwd(dxn2, x1, mean2, rstd2, n2w.detach(), res=dx2,
                                                        dgamma=rt.grad_out(n2w), dbeta=rt.grad_out(n2b))
            del dx2m
        # ---- conditioning adds ----
        dz = dl = dwg = dbg = dwg2 = dbg2 = None
        if l is not None:
            dg2 = ops.bcast_add_bwd(dx1, n_seq, S, p, s0 + 5, rt.seed)
            dwg2, dbg2 = _wbgrad(rt, wg2, bg2, dg2, l)
            if ctx.needs_input_grad[4]:
                dl = ops.gemm(dg2, rt.w(wg2), b_kc=False)
        if z is not None:
            # (sequences past the live prefix: zero gradient rows, written by the same launch; and with dropout on, the launch
            # that reads every element of dx1 anyway also writes dx1m = drop1's mask replayed on dx1 for the attention half)
            vec = dx1.dtype == torch.bfloat16 and dx1.is_contiguous() and dx1.shape[1] % 8 == 0 and dx1.shape[1] <= 512
            # (hoisted conditioning rows: the gradient goes straight into this layer's column block of the stack's shared buffer)
            dg_out = rt.cond_grad_block(z, n_seq_full) if (wg is None and vec and dx1.data_ptr() % 16 == 0) else None
            if BCAST_MASKED and dx1m is None and p > 0 and vec and n_seq * S <= dx1.shape[0] <= n_seq_full * S:
                dg, dx1m = ops.bcast_add_bwd(dx1, n_seq, S, p, s0 + 2, rt.seed, n_seq_out=n_seq_full, mask_site=s0 + 1, out=dg_out)
            else:
                dg = ops.bcast_add_bwd(dx1, n_seq, S, p, s0 + 2, rt.seed, n_seq_out=n_seq_full, out=dg_out)
            if wg is None:
                dz = dg                     # `z` was the projected row itself: its gradient goes to GlobalCondFn
            else:
                dwg, dbg = _wbgrad(rt, wg, bg, dg, z)
                if ctx.needs_input_grad[3]:
                    dz = ops.gemm(dg, rt.w(wg), b_kc=False)
        # ---- attention: x1 = x + drop1(ao Wo^T + bo) ----
        if dx1m is None:
            dx1m = ops.drop_apply(dx1, p, s0 + 1, rt.seed)
        dwo, dbo = _wbgrad(rt, wo, bo, dx1m, ao)
        wob = None
        if (ATTN_BWD_OUTPROJ and not _ATTN_VALU and S >= _ATTN_MFMA_MIN_S
                and rt.store is not None and x.dtype == torch.bfloat16 and H == 8 and x.shape[1] == 256
                and not ctx.causal and x.shape[0] >= ATTN_MIN_ROWS and (key_mask is None or key_mask.dtype == torch.int64)
                and ((seq_off is None and 16 < S <= 32 and x.shape[0] >= n_seq * S) or (seq_off is not None and ctx.tiles is not None))):
            wob = rt.store.attn_bwd(win)
        if wob is not None:
            # the head-output gradient dx1m @ Wo is formed tile by tile inside the attention backward kernel
            dqkv = ops.attention_bwd_outproj(qkv, key_mask, dx1m, wob, n_seq, S, ctx.scale, p, s0, rt.seed, seq_off=seq_off,
                                             tiles=ctx.tiles)
        else:
            dao = ops.gemm(dx1m, rt.w(wo), b_kc=False)
            if ctx.causal:
                dqkv = ops.attention_bwd(qkv, key_mask, dao, n_seq, S, H, ctx.scale, p, s0, rt.seed, causal=True)
            else:
                dqkv = ops.attention_bwd(qkv, key_mask, dao, n_seq, S, H, ctx.scale, p, s0, rt.seed, seq_off=seq_off,
                                         tiles=ctx.tiles)
        del dx1m
        dwin, dbin = _wbgrad(rt, win, bin_, dqkv, xn1)
        dx_out = None
        if live is not None:
            dx_full = torch.empty((full_rows, x.shape[1]), dtype=x.dtype, device=x.device)
            dx_out = dx_full[:x.shape[0]]
        mk = (p, ctx.mask_below, rt.seed) if (ctx.mask_below is not None and p > 0) else None
        wib = None
        if (ATTN_BWD_DX and rt.store is not None and x.dtype == torch.bfloat16 and x.shape[1] == 256 and H == 8
                and x.shape[0] >= ATTN_BWD_DX_MIN_ROWS and mean1 is not None and dx1.is_contiguous()):
            wib = rt.store.attn_bwd(win)
        with rt.deferring():
            if wib is not None:
                # round 6: dxn1 = dqkv . W_in, the LayerNorm backward and the residual add in ONE launch (csrc/attn_bwd_dx.hip):
                # the [rows, 256] intermediate is never written, one launch less per layer
                with ops.tag("attn"):
                    out = ops.attn_bwd_dx(dqkv, x, mean1, rstd1, n1w.detach(), dx1, wib, dx=dx_out,
                                          dgamma=rt.grad_out(n1w), dbeta=rt.grad_out(n1b), masked=mk)
            else:
                dxn1 = ops.gemm(dqkv, rt.w(win), b_kc=False)
                out = ops.layernorm_bwd(dxn1, x, mean1, rstd1, n1w.detach(), res=dx1, dx=dx_out,
                                        dgamma=rt.grad_out(n1w), dbeta=rt.grad_out(n1b), masked=mk)
            dx, dn1w, dn1b = out[:3]
        if live is not None:
            dx = dx_full
        if mk is not None:
            _hand_masked(rt, dx, out[3], p, ctx.mask_below)     # (the layer below takes it by dx's address)
        return (None, dx, None, dz, dl, None, None, None, None, None,
                dn1w, dn1b, dwin, dbin, dwo, dbo, dn2w, dn2b, dw1, db1, dw2, db2, dwg, dbg, dwg2, dbg2, None, None, None,
                None, None, None)


# --------------------------------------------------------------------------------------------------
def gs_stack_eligible(rt, x, key_mask, n_seq, S, n_heads, n_layers, wins):
    """the conditions of LayerFn's one-launch-per-layer route (csrc/group_stage.hip), for every layer of the stack"""
    return (GS_STACK and rt.store is not None and x.dtype == torch.bfloat16 and n_heads == 8 and x.dim() == 2 and x.shape[1] == 256
            and 32 % S == 0 and x.shape[0] == n_seq * S and x.shape[0] < min(ATTN_MIN_ROWS, FFN_MIN_ROWS)
            and 1 <= n_layers <= ops.GS_STACK_MAX and (key_mask is None or key_mask.dtype == torch.int64)
            and all(rt.store.gs(w) is not None for w in wins))


class GsStackFn(torch.autograd.Function):
    """A whole stack of pre-LN blocks of a short-sequence ("group") stage - the layer loops of
    deepsvg/model/layers/transformer.py:168-188 / :214-242 over hierarchical_encoder / hierarchical_decoder
    (deepsvg/model/model.py:153-161, :246-254) - as ONE launch forward and ONE backward (csrc/group_stage.hip: a workgroup
    carries its 32 rows through all the layers).  Same stores, same draws, same gradients as LayerFn layer by layer (its `gs`
    route); the conditioning rows g_l = linear_global_l(z) come projected (GlobalCondFn) and their gradients leave as the column
    blocks of ONE [n_seq, n * 256] buffer - no concatenation launch in GlobalCondFn.backward.
    tensors: per layer norm1.weight, norm1.bias, in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, norm2.weight,
    norm2.bias, linear1.weight, linear1.bias, linear2.weight, linear2.bias [, g_l]; layer l draws at the sites site0 + 8 l + ..."""

    @staticmethod
    def forward(ctx, rt, x, key_mask, n_seq, S, n_heads, drop_rate, site0, n_layers, has_g, *tensors):
        p = rt.p(drop_rate)
        per = 13 if has_g else 12
        assert len(tensors) == per * n_layers
        scale = float(x.shape[1] // n_heads) ** -0.5
        want_bwd = any(ctx.needs_input_grad)
        layers = []
        for i in range(n_layers):
            n1w, n1b, win, bin_, wo, bo, n2w, n2b, w1, b1, w2, b2 = tensors[per * i:per * i + 12]
            layers.append(dict(img=rt.store.gs(win)[0], in_bias=bin_.detach(), out_bias=bo.detach(), b1=b1.detach(), b2=b2.detach(),
                               gamma1=n1w.detach(), beta1=n1b.detach(), gamma2=n2w.detach(), beta2=n2b.detach(),
                               site0=site0 + 8 * i, seq_add=tensors[per * i + 12] if has_g else None))
        with ops.tag("gs"):
            res = ops.gs_stack_fwd(x, layers, key_mask, n_seq, S, scale, 1e-5, p, rt.seed, train=want_bwd)
        rt.last_layer_gs = True
        ctx.rt, ctx.n_seq, ctx.S, ctx.p, ctx.site0, ctx.scale, ctx.n, ctx.has_g = rt, n_seq, S, p, site0, scale, n_layers, has_g
        if not want_bwd:
            return res[-1]
        saved = [x, key_mask]
        for r in res:
            saved += list(r)                # x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h
        ctx.save_for_backward(*saved, *tensors)
        return res[-1][0]

    @staticmethod
    def backward(ctx, dx2):
        rt, n_seq, S, p, n, has_g = ctx.rt, ctx.n_seq, ctx.S, ctx.p, ctx.n, ctx.has_g
        sv = ctx.saved_tensors
        x, key_mask = sv[0], sv[1]
        acts = [sv[2 + 11 * i:2 + 11 * (i + 1)] for i in range(n)]
        tensors = sv[2 + 11 * n:]
        per = 13 if has_g else 12
        prm = [tensors[per * i:per * i + 12] for i in range(n)]
        layers = []
        for i in range(n):
            x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h = acts[i]
            n1w, n1b, win, bin_, wo, bo, n2w, n2b, w1, b1, w2, b2 = prm[i]
            layers.append(dict(img=rt.store.gs(win)[1], x=x if i == 0 else acts[i - 1][0], mean1=mean1, rstd1=rstd1, qkv=qkv, x1=x1,
                               mean2=mean2, rstd2=rstd2, h=h, gamma1=n1w.detach(), gamma2=n2w.detach(), site0=ctx.site0 + 8 * i,
                               dgamma2=rt.grad_out(n2w), dbeta2=rt.grad_out(n2b), dgamma1=rt.grad_out(n1w),
                               dbeta1=rt.grad_out(n1b)))
        with rt.deferring(), ops.tag("gs"):
            dx, ops_, dgcat = ops.gs_stack_bwd(dx2.contiguous(), layers, key_mask, n_seq, S, ctx.scale, p, rt.seed, want_dg=has_g)
        # the weight-gradient products, queued exactly as LayerFn.backward queues them layer by layer (last layer first): ONE grouped
        # launch for the stack (STACK_GROUP) or one per layer
        stack = STACK_GROUP and GROUP_WGRAD and rt.defer
        nsl = STACK_GROUP_SLICES if stack else 8
        gb = (lambda w_: nsl * -(-w_.shape[0] // 128) * -(-w_.shape[1] // 128)) if GROUP_WGRAD else (lambda w_: None)
        grads = [None] * (per * n)
        for i in range(n - 1, -1, -1):
            x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h = acts[i]
            n1w, n1b, win, bin_, wo, bo, n2w, n2b, w1, b1, w2, b2 = prm[i]
            dym, dpre, dx1m, dqkv = ops_[i]
            keep = rt.stack_group_begin() if stack else None
            with (rt.grouping() if keep is None else _NULL_CTX):
                with ops.tag("ffn"):
                    dw2, db2 = _wbgrad(rt, w2, b2, dym, h, gb(w2))
                    dw1, db1 = _wbgrad(rt, w1, b1, dpre, xn2, gb(w1))
                dwo, dbo = _wbgrad(rt, wo, bo, dx1m, ao, gb(wo))
                dwin, dbin = _wbgrad(rt, win, bin_, dqkv, xn1, gb(win))
            if keep is not None:
                keep += [dym, h, dpre, xn2, dx1m, ao, dqkv, xn1]
                if i == 0:
                    rt.stack_group_end()
            L = layers[i]
            grads[per * i:per * i + 12] = [L["dgamma1"], L["dbeta1"], dwin, dbin, dwo, dbo, L["dgamma2"], L["dbeta2"], dw1, db1, dw2, db2]
            if has_g:
                grads[per * i + 12] = dgcat[:, 256 * i:256 * (i + 1)]
        return (None, dx, None, None, None, None, None, None, None, None, *grads)


class MaskedCEFn(torch.autograd.Function):
    """(sum, count) of CE(logits[row], target[row]) over the rows with w != 0 (fp32 [2]; the mean and the weighted total are
    taken by LossCombineFn); logits2d: [n_tok, group*C] row-major view."""

    @staticmethod
    def forward(ctx, logits2d, target, w, C_, group, count_fn):
        lse, sc = ops.masked_ce_fwd(logits2d, target, w, C_, group)
        if count_fn is not None:
            # data-parallel: replace the local count by (global count / world) so that the rank-averaged
            # gradient equals the gradient of the global mean (deepsvg_amd/trainer.py)
            sc = torch.stack([sc[0], count_fn(sc[1].clone()).to(sc.dtype).reshape(())])
        ctx.C_, ctx.group = C_, group
        ctx.save_for_backward(logits2d, target, w, lse, sc)
        return sc

    @staticmethod
    def backward(ctx, dsc):
        """dsc[0] = dL / d(mean CE) (LossCombineFn): the kernel divides by the count itself"""
        logits2d, target, w, lse, sc = ctx.saved_tensors
        g = dsc[0:1].to(torch.float32).contiguous()
        mult = 4 if logits2d.dtype == torch.float32 else 8
        dlogits = ops.masked_ce_bwd(logits2d, target, w, lse, sc, g, 1.0, ctx.C_, ctx.group, pad_to=mult)
        return dlogits, None, None, None, None, None


class LossCombineFn(torch.autograd.Function):
    """(total, term_0, term_1, ...) with term_i = sum_i / count_i and total = sum_i weights[i] * term_i from the (sum, count)
    pairs of the cross-entropies (deepsvg/model/loss.py:43-57): one launch forward, one backward, instead of a dozen scalar
    elementwise launches each way."""

    @staticmethod
    def forward(ctx, weights, *scs):
        out = ops.loss_combine_fwd([sc.contiguous() for sc in scs], weights)
        ctx.weights, ctx.device = tuple(float(w) for w in weights), out.device
        ctx.set_materialize_grads(False)
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, dtotal, *dterms):
        f32 = lambda t: None if t is None else t.reshape(1).to(torch.float32).contiguous()
        dsc = ops.loss_combine_bwd(f32(dtotal), [f32(t) for t in dterms], ctx.weights, ctx.device)
        return (None,) + tuple(dsc.unbind(0))


def _tail_is_own_bias_or_slack(st, weight, bias, n_read):
    """HEAD_KPAD reads `n_read` elements from the start of `weight` in the bf16 flat image, i.e. past its end: allowed only when
    everything behind the weight up to there is this head's own bias or the store's zero slack (never trained) - `0 x value`
    must not meet another, possibly diverged (Inf / NaN) parameter.  True for every shipped config (args_fcn.weight and
    args_fcn.bias are the last two parameters); anything else takes the register-staged GEMM."""
    ent, entb = st.index.get(id(weight)), st.index.get(id(bias))
    if ent is None:
        return False
    end_read = ent[0] + n_read
    for p in st.params:
        o, n, _shape = st.index[id(p)]
        if o >= ent[0] + ent[1] and o < end_read and p is not bias:
            return False
    return entb is None or entb[0] >= ent[0] + ent[1]


class ArgsHeadLossFn(torch.autograd.Function):
    """(sum, count) of the masked CE over the argument logits (-> loss_args by LossCombineFn), with the argument head (args_fcn, deepsvg/model/model.py:228-246)
    folded in: forward AND backward run on the tokens that carry argument loss only.  Every other token's logits do
    not enter the loss and their dlogits are exact zeros (loss.py:51-54), so logits, dX, dW and db come from a compact
    [n_live, 2827] problem instead of the dense [T, 2827] one (about 30 % of the decoder tokens on the synthetic
    distribution).  The full `args_logits` of the model's result dict is materialised only if somebody reads it.
    `live` = (token list int32 padded with -1, number of rows to process >= number of listed tokens)."""

    @staticmethod
    def forward(ctx, rt, x, weight, bias, target, w, C_, group, count_fn, live, slot_lo=0, cmd_logits=None,
                cmd_weight=None, cmd_bias=None, cmd_target=None, cmd_w=None, cmd_count_fn=None):
        """target / w: [n_tok * group] for the `group` argument slots slot_lo .. slot_lo + group - 1 (the slots that
        carry loss in this batch: the head's other output rows get exact zero gradients).
        cmd_logits [n_tok, n_cmd] (detached: the model computed them from the same x) with command_fcn's weight / bias, the
        command targets / weights and their count_fn: the command head's cross-entropy joins this node - both heads read
        the same x, so ONE backward node writes dX once (the command head's product, the argument head's rows added into
        it) instead of two nodes + a zero-filled scatter target + autograd's sum.  -> sc_args, or (sc_cmd, sc_args)"""
        R = min(int(live[1]), x.shape[0])
        idx = live[0][:R]
        xc = ops.gather_groups(x, idx, R, 1)                       # rows of list padding read token 0 (weight 0)
        r0, r1 = slot_lo * C_, (slot_lo + group) * C_              # the head's output rows in use
        n_out = r1 - r0
        w_used = rt.w(weight)[r0:r1]
        b_used = bias.detach()[r0:r1]
        if HEAD_FUSED and b_used.data_ptr() % 16:
            b_used = b_used.clone()                                # (csrc/head_fused.hip reads the bias in 16-byte pieces; the GEMM
                                                                   # epilogues take any 4-byte-aligned row range - round 6: no copy launch)
        ctx.head_img = None
        if HEAD_FUSED and xc.dtype == torch.bfloat16 and xc.shape[1] == 256 and C_ >= 64 and n_out <= 3008:
            # the logit tile stays on chip (csrc/head_fused.hip): log-sum-exp, target logit and the loss sums in one launch;
            # backward recomputes the tile and emits dlogits directly - the [R, n_out] logits are never stored
            ctx.head_img = ops.head_pack(w_used if w_used.is_contiguous() else w_used.contiguous())
            logits_c = None
            lse, sc = ops.head_lse(xc, ctx.head_img, b_used, n_out, C_, target, w, tok_idx=idx)
        else:
            mult = 4 if xc.dtype == torch.float32 else (64 if HEAD_KPAD else 8)    # (bf16 rows start on 128-byte lines)
            ld = (n_out + mult - 1) // mult * mult
            buf = torch.empty((R, ld), dtype=xc.dtype, device=xc.device)
            logits_c = buf[:, :n_out]
            ops.gemm(xc, w_used, bias=b_used, out=logits_c)
            lse, sc = ops.masked_ce_fwd(logits_c, target, w, C_, group, tok_idx=idx)
        ctx.b_used = b_used
        if count_fn is not None:
            sc = torch.stack([sc[0], count_fn(sc[1].clone()).to(sc.dtype).reshape(())])
        ctx.rt, ctx.C_, ctx.group, ctx.rows_full, ctx.rows_used = rt, C_, group, x.shape[0], (r0, r1)
        ctx.has_cmd = cmd_logits is not None
        if cmd_logits is None:
            ctx.save_for_backward(xc, weight, bias, logits_c, target, w, lse, sc, idx)
            return sc
        cl, cw, cb, ctgt, cwt, ccount = cmd_logits, cmd_weight, cmd_bias, cmd_target, cmd_w, cmd_count_fn
        ctx.n_cmd = cl.shape[1]
        clse, csc = ops.masked_ce_fwd(cl, ctgt, cwt, ctx.n_cmd, 1)
        if ccount is not None:
            csc = torch.stack([csc[0], ccount(csc[1].clone()).to(csc.dtype).reshape(())])
        ctx.save_for_backward(xc, weight, bias, logits_c, target, w, lse, sc, idx, x, cl, cw, cb, ctgt, cwt, clse, csc)
        ctx.set_materialize_grads(False)
        return csc, sc

    @staticmethod
    def backward(ctx, *dscs):
        rt = ctx.rt
        dx = dwc = dbc = None
        if ctx.has_cmd:
            (xc, weight, bias, logits_c, target, w, lse, sc, idx, x, cl, cw, cb, ctgt, cwt, clse, csc) = ctx.saved_tensors
            dcsc, dsc = dscs
            zero1 = lambda: torch.zeros(1, dtype=torch.float32, device=xc.device)
            gc = dcsc[0:1].to(torch.float32).contiguous() if dcsc is not None else zero1()
            if dsc is None:
                dsc = torch.zeros(2, dtype=torch.float32, device=xc.device)
            mult = 4 if cl.dtype == torch.float32 else 8
            dcl = ops.masked_ce_bwd(cl, ctgt, cwt, clse, csc, gc, 1.0, ctx.n_cmd, 1, pad_to=mult)
            dwc, dbc = _wbgrad(rt, cw, cb, dcl, x)
            dx = ops.gemm(dcl, rt.w(cw), b_kc=False)       # dense [rows, d_model]: the argument head's rows join it below
        else:
            xc, weight, bias, logits_c, target, w, lse, sc, idx = ctx.saved_tensors
            dsc, = dscs
        g = dsc[0:1].to(torch.float32).contiguous()
        r0, r1 = ctx.rows_used
        w_ext = None
        if ctx.head_img is not None:
            dl = ops.head_dlogits(xc, ctx.head_img, ctx.b_used, r1 - r0, ctx.C_, target, w, lse, sc, g, 1.0, tok_idx=idx)
        else:
            mult = 4 if logits_c.dtype == torch.float32 else 8
            # The input-gradient product dl [R, n_out] x W [n_out, 256] reduces over n_out = 257 * slots, no multiple of the LDS-DMA
            # GEMM's 64-wide K step (it then runs on the register-staged kernel: 67 us of the step).  With the dlogits rows padded to
            # a multiple of 64 - the pad columns are written as exact zeros - and the weight view extended over the rows that follow
            # it in the bf16 image (finite numbers times zeros), K is a multiple of 64 and the product is the same sum.
            wl = rt.w(weight)
            kp = (r1 - r0 + 63) // 64 * 64
            st = rt.store
            # (only the wave-per-token kernel of dsvg_masked_ce_bwd zero-fills the pad columns: its conditions, csrc/loss.hip)
            tok_kernel = (1 < ctx.group <= 64 and ctx.C_ >= 8 and logits_c.stride(0) % 8 == 0
                          and logits_c.stride(0) >= r1 - r0 and logits_c.data_ptr() % 16 == 0)
            if (HEAD_KPAD and tok_kernel and mult == 8 and kp != r1 - r0 and st is not None and st.flat_lp is not None and wl.is_contiguous()
                    and wl.untyped_storage().data_ptr() == st.flat_lp.untyped_storage().data_ptr()
                    and wl.storage_offset() + (r0 + kp) * wl.shape[1] <= st.flat_lp.numel()
                    and _tail_is_own_bias_or_slack(st, weight, bias, (r0 + kp) * wl.shape[1])):
                mult = 64
                w_ext = torch.as_strided(wl, (kp, wl.shape[1]), (wl.shape[1], 1), wl.storage_offset() + r0 * wl.shape[1])
            dl = ops.masked_ce_bwd(logits_c, target, w, lse, sc, g, 1.0, ctx.C_, ctx.group, pad_to=mult, tok_idx=idx,
                                   logits_compact=True)
        if (r0, r1) == (0, weight.shape[0]):
            dw, db = _wbgrad(rt, weight, bias, dl, xc)
            if w_ext is not None:
                dxc = ops.gemm(torch.as_strided(dl, (dl.shape[0], w_ext.shape[0]), (dl.stride(0), 1)), w_ext, b_kc=False)
            else:
                dxc = ops.gemm(dl, rt.w(weight), b_kc=False)
        else:
            # only the output rows [r0, r1) of the head saw a loss term: their gradient comes from the GEMMs, the rest is 0
            dw, db = rt.grad_out(weight), rt.grad_out(bias)
            split = ops.split_k_for(r1 - r0, weight.shape[1], dl.shape[0])
            with rt.deferring(), _wgrad_tag():
                for t in (dw[:r0], dw[r1:], db[:r0], db[r1:]):       # (inside the scope: the fills ride on the flush)
                    ops.zero_(t) if DEFER_MORE else t.zero_()
                if split > 1:
                    ops.gemm(dl, xc, a_kc=False, b_kc=False, out=dw[r0:r1], split_k=split, rowsum=db[r0:r1])
                else:
                    ops.gemm(dl, xc, a_kc=False, b_kc=False, out=dw[r0:r1])
                    ops.colsum(dl, out=db[r0:r1])
            if w_ext is not None:       # (dl's buffer over its whole padded width)
                dxc = ops.gemm(torch.as_strided(dl, (dl.shape[0], w_ext.shape[0]), (dl.stride(0), 1)), w_ext, b_kc=False)
            else:
                dxc = ops.gemm(dl, rt.w(weight)[r0:r1], b_kc=False)
        if dx is None:
            dx = torch.zeros((ctx.rows_full, xc.shape[1]), dtype=xc.dtype, device=xc.device)
            ops.scatter_rows(dxc, idx, dx)
        else:
            ops.scatter_rows(dxc, idx, dx, accumulate=True)
        return (None, dx, dw, db, None, None, None, None, None, None, None, None, dwc, dbc, None, None, None)
