"""MI355X-native SVGTransformer: same constructor, forward signature, result dict and state_dict layout as
deepsvg.model.model.SVGTransformer (deepsvg/model/model.py:288-479), so cfg.make_model() can return it and the
reference trainer (deepsvg/train.py) and checkpoints work unchanged.  The sub-modules below are parameter
containers whose names mirror the reference's; all arithmetic runs in the gfx950 kernels behind
deepsvg_amd.functional / deepsvg_amd.ops.

Internal layout: token-major activations [n_seq * S, d] with the batch-first order of the inputs
((n, g, s) -> row (n*G + g)*S + s), i.e. the (N, G, S, d) layout; the reference's seq-first permutes and
g*N+n packing (deepsvg/utils/utils.py:20-49) never happen.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops
from . import functional as Fn
from .svgtensor import CMD_ARGS_MASK, EOS_ID, M_ID, SOS_ID

# bf16 copy + every fused-kernel weight image (+ the step's seed advance) in one launch (ops.pack_images); 0 = the 7 (+ 1)
# stand-alone launches, bit-identical (A/B knob)
PACK_ONE_LAUNCH = os.environ.get("DSVG_PACK_ONE", "1") != "0"
PE_DROPOUT = 0.1  # hard-wired in PositionalEncodingLUT (deepsvg/model/layers/positional_encoding.py:26-28)


class ModelOutput(dict):
    """The result dict of SVGTransformer.forward (deepsvg/model/model.py:396-412).  An entry may be LAZY: a thunk that
    is run on first read.  The training forward registers `args_logits` that way when deepsvg_amd.SVGLoss can take the
    loss from the fused argument head (functional.ArgsHeadLossFn), so the dense (N, G, S, 11, 257) tensor - the largest
    HBM stream of the step - is only written if somebody actually reads it (the reference's SVGLoss, a metric, a test)."""
    _PENDING = object()

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._thunks = {}

    def set_lazy(self, key, fn):
        self._thunks[key] = fn
        dict.__setitem__(self, key, ModelOutput._PENDING)

    def is_pending(self, key):
        return key in self._thunks

    def _force(self, key=None):
        for k in ([key] if key is not None else list(self._thunks)):
            fn = self._thunks.pop(k, None)
            if fn is not None:
                dict.__setitem__(self, k, fn())

    def __getitem__(self, key):
        self._force(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._force(key)
        return dict.get(self, key, default)

    def __iter__(self):         # (overriding __iter__ also sends dict(x) / {**x} through keys() + __getitem__)
        return dict.__iter__(self)

    def items(self):
        self._force()
        return dict.items(self)

    def values(self):
        self._force()
        return dict.values(self)

    def pop(self, key, *default):
        self._force(key)
        return dict.pop(self, key, *default)

    def copy(self):
        self._force()
        return dict(dict.items(self))


# ----------------------------------------------------------------------------------------------------
# parameter containers (names == reference state_dict keys)
# ----------------------------------------------------------------------------------------------------
class _PosLUT(nn.Module):
    """PositionalEncodingLUT (positional_encoding.py:24-43): `pos_embed.weight` + `position` buffer"""

    def __init__(self, d_model, max_len):
        super().__init__()
        self.register_buffer("position", torch.arange(0, max_len, dtype=torch.long).unsqueeze(1))
        self.pos_embed = nn.Embedding(max_len, d_model)
        nn.init.kaiming_normal_(self.pos_embed.weight, mode="fan_in")


class _SVGEmbedding(nn.Module):
    """SVGEmbedding (model.py:16-44)"""

    def __init__(self, cfg, seq_len, rel_args=False, use_group=True, group_len=None):
        super().__init__()
        self.command_embed = nn.Embedding(cfg.n_commands, cfg.d_model)
        args_dim = 2 * cfg.args_dim if rel_args else cfg.args_dim + 1
        self.arg_embed = nn.Embedding(args_dim, 64)
        self.embed_fcn = nn.Linear(64 * cfg.n_args, cfg.d_model)
        self.use_group = use_group
        if use_group:
            if group_len is None:
                group_len = cfg.max_num_groups
            self.group_embed = nn.Embedding(group_len + 2, cfg.d_model)
        self.pos_encoding = _PosLUT(cfg.d_model, max_len=seq_len + 2)
        nn.init.kaiming_normal_(self.command_embed.weight, mode="fan_in")
        nn.init.kaiming_normal_(self.arg_embed.weight, mode="fan_in")
        nn.init.kaiming_normal_(self.embed_fcn.weight, mode="fan_in")
        if use_group:
            nn.init.kaiming_normal_(self.group_embed.weight, mode="fan_in")


class _ConstEmbedding(nn.Module):
    """ConstEmbedding (model.py:60-73)"""

    def __init__(self, cfg, seq_len):
        super().__init__()
        self.seq_len = seq_len
        self.PE = _PosLUT(cfg.d_model, max_len=seq_len)


class _SelfAttn(nn.Module):
    """MultiheadAttention parameters (layers/attention.py:46-99)"""

    def __init__(self, d_model):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class _Layer(nn.Module):
    """TransformerEncoderLayerImproved / TransformerDecoderLayerGlobalImproved parameters
    (layers/improved_transformer.py:16-34,97-119)"""

    def __init__(self, d_model, dim_ff, d_global=None, d_global2=None):
        super().__init__()
        self.self_attn = _SelfAttn(d_model)
        if d_global is not None:
            self.linear_global = nn.Linear(d_global, d_model)
        if d_global2 is not None:
            self.linear_global2 = nn.Linear(d_global2, d_model)
        self.linear1 = nn.Linear(d_model, dim_ff)
        self.linear2 = nn.Linear(dim_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _Stack(nn.Module):
    """TransformerEncoder / TransformerDecoder (layers/transformer.py:146-242): `layers.{i}` + final `norm`.
    Like the reference's _get_clones (transformer.py:383-384) all layers start from identical weights."""

    def __init__(self, n_layers, d_model, dim_ff, d_global=None, d_global2=None):
        super().__init__()
        proto = _Layer(d_model, dim_ff, d_global, d_global2)
        layers = [proto]
        for _ in range(n_layers - 1):
            clone = _Layer(d_model, dim_ff, d_global, d_global2)
            clone.load_state_dict(proto.state_dict())
            layers.append(clone)
        self.layers = nn.ModuleList(layers)
        self.norm = nn.LayerNorm(d_model)


class _FCN(nn.Module):
    def __init__(self, d_model, n_commands, n_args, args_dim):
        super().__init__()
        self.command_fcn = nn.Linear(d_model, n_commands)
        self.args_fcn = nn.Linear(d_model, n_args * args_dim)


class _HierarchFCN(nn.Module):
    def __init__(self, d_model, dim_z):
        super().__init__()
        self.visibility_fcn = nn.Linear(d_model, 2)
        self.z_fcn = nn.Linear(d_model, dim_z)


class _ResNet(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        for i in range(1, 5):
            setattr(self, f"linear{i}", nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU()))


class _VAE(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.enc_mu_fcn = nn.Linear(cfg.d_model, cfg.dim_z)
        self.enc_sigma_fcn = nn.Linear(cfg.d_model, cfg.dim_z)
        nn.init.normal_(self.enc_mu_fcn.weight, std=0.001)
        nn.init.constant_(self.enc_mu_fcn.bias, 0)
        nn.init.normal_(self.enc_sigma_fcn.weight, std=0.001)
        nn.init.constant_(self.enc_sigma_fcn.bias, 0)


class _Bottleneck(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.bottleneck = nn.Linear(cfg.d_model, cfg.dim_z)


class _LabelEmbedding(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.label_embedding = nn.Embedding(cfg.n_labels, cfg.dim_label)
        nn.init.kaiming_normal_(self.label_embedding.weight, mode="fan_in")


class _Encoder(nn.Module):
    """Encoder parameters (model.py:92-119)"""

    def __init__(self, cfg):
        super().__init__()
        seq_len = cfg.max_seq_len if cfg.encode_stages == 2 else cfg.max_total_len
        self.use_group = cfg.encode_stages == 1
        self.embedding = _SVGEmbedding(cfg, seq_len, use_group=self.use_group)
        if cfg.label_condition:
            self.label_embedding = _LabelEmbedding(cfg)
        dim_label = cfg.dim_label if cfg.label_condition else None
        self.encoder = _Stack(cfg.n_layers, cfg.d_model, cfg.dim_feedforward, None, dim_label)
        if cfg.encode_stages == 2:
            if not cfg.self_match:
                self.hierarchical_PE = _PosLUT(cfg.d_model, max_len=cfg.max_num_groups)
            self.hierarchical_encoder = _Stack(cfg.n_layers, cfg.d_model, cfg.dim_feedforward, None, dim_label)


class _Decoder(nn.Module):
    """Decoder parameters (model.py:200-236), one-shot prediction mode"""

    def __init__(self, cfg, args_dim):
        super().__init__()
        if cfg.label_condition:
            self.label_embedding = _LabelEmbedding(cfg)
        dim_label = cfg.dim_label if cfg.label_condition else None
        if cfg.decode_stages == 2:
            self.hierarchical_embedding = _ConstEmbedding(cfg, cfg.num_groups_proposal)
            self.hierarchical_decoder = _Stack(cfg.n_layers_decode, cfg.d_model, cfg.dim_feedforward, cfg.dim_z, dim_label)
            self.hierarchical_fcn = _HierarchFCN(cfg.d_model, cfg.dim_z)
        if cfg.pred_mode == "autoregressive":       # model.py:217-222
            self.embedding = _SVGEmbedding(cfg, cfg.max_total_len, rel_args=cfg.rel_targets, use_group=True,
                                           group_len=cfg.max_total_len)
            sz = cfg.max_total_len + 1
            mask = torch.triu(torch.full((sz, sz), float("-inf")), diagonal=1)      # utils.py:69-72
            self.register_buffer("square_subsequent_mask", mask)
        else:
            seq_len = cfg.max_seq_len + 1 if cfg.decode_stages == 2 else cfg.max_total_len + 1
            self.embedding = _ConstEmbedding(cfg, seq_len)
        self.decoder = _Stack(cfg.n_layers_decode, cfg.d_model, cfg.dim_feedforward, cfg.dim_z, dim_label)
        self.fcn = _FCN(cfg.d_model, cfg.n_commands, cfg.n_args, args_dim)


# ----------------------------------------------------------------------------------------------------
# flat parameter / gradient storage
# ----------------------------------------------------------------------------------------------------
class ParamStore:
    """Keeps every parameter of a module as a view of one contiguous fp32 buffer (8-element aligned slots),
    plus a bf16 image of the same layout and flat gradient buffers.  One RCCL all-reduce, one grad-norm and one
    AdamW launch then cover the whole model (deepsvg_amd/trainer.py)."""

    ALIGN = 8

    def __init__(self, module):
        self.module = module
        self.flat = None
        self.flat_lp = None
        self.gbuf = [None, None]
        self.index = {}          # id(param) -> (offset, numel, shape)
        self.params = []
        self.total = 0
        self._lp_views = {}      # id(param) -> view of the bf16 image (views are re-used: ~2000 lookups per step)
        self._grad_views = [{}, {}]
        self._in_flight = set()  # id(param): its slot of gbuf[0] was handed to a backward node, not yet accumulated
        self._hooks = []
        self._ffn = None         # fused-FFN weight images (see _ffn_setup)
        self._attn = None        # fused attention-block weight images (see _attn_setup)
        self.pending_advance = None     # (step counter, seed) TrainStep wants advanced at the start of the next forward

    # a copied / unpickled module gets an empty store that re-flattens lazily on its first forward: the index is keyed
    # by id(param) and the views point into THIS module's buffers (copy.deepcopy(model) for EMA / best-model snapshots,
    # torch.save(model))
    def __deepcopy__(self, memo):
        return ParamStore(memo.get(id(self.module)))

    def __getstate__(self):
        return {"module": self.module}

    def __setstate__(self, state):
        self.__init__(state["module"])

    @staticmethod
    def _grouped_order(module):
        """module.parameters() with the linear_global weights (then biases) of every decoder stack moved next to each other, at
        the position of the stack's first one: the stack's four weights are then ONE [4 * 256, 256] matrix in the flat
        buffers - their gradients one weight-gradient GEMM (functional.GlobalCondFn).  Everything else keeps its order."""
        params = [p for p in module.parameters()]
        pos = {id(p): i for i, p in enumerate(params)}
        for m in module.modules():
            layers = getattr(m, "layers", None)
            if layers is None or len(layers) < 2 or not all(hasattr(L, "linear_global") for L in layers):
                continue
            ws = [L.linear_global.weight for L in layers]
            bs = [L.linear_global.bias for L in layers]
            if any(b is None for b in bs) or len({tuple(w.shape) for w in ws}) != 1:
                continue
            group = ws + bs
            ids = {id(p) for p in group}
            first = min(pos[id(p)] for p in group)
            rest = [p for p in params if id(p) not in ids]
            n_before = sum(1 for p in params[:first] if id(p) not in ids)
            params = rest[:n_before] + group + rest[n_before:]
            pos = {id(p): i for i, p in enumerate(params)}
        return params

    def _flatten(self, device):
        params = self._grouped_order(self.module)
        off = 0
        index = {}
        for p in params:
            index[id(p)] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = off
        # slack behind the last parameter (zeros, never trained: their gradient is zero): 64 elements for 16-byte tails, and 63
        # rows of 256 so that a [k, 256] weight view may be read up to the next multiple of 64 rows (functional.HEAD_KPAD)
        flat = torch.zeros(total + 64 + 63 * 256, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p in params:
                o, n, shape = index[id(p)]
                view = flat[o:o + n].view(shape)
                view.copy_(p.data)
                p.data = view
        self.flat, self.index, self.params, self.total = flat, index, params, total
        self.generation = getattr(self, "generation", 0) + 1        # (TrainStep drops what it cached per parameter id)
        self.flat_lp = None
        self.gbuf = [None, None]
        self._lp_views = {}
        self._grad_views = [{}, {}]
        self._in_flight = set()
        self._ffn_setup(device)
        self._attn_setup(device)
        self._gs_setup(device)
        for h in self._hooks:
            h.remove()
        # a slot of the flat gradient buffer is free again once AccumulateGrad has consumed it (see grad_view)
        self._hooks = [p.register_post_accumulate_grad_hook(self._accumulated) for p in params if p.requires_grad]

    def _accumulated(self, param):
        self._in_flight.discard(id(param))

    def _ffn_setup(self, device):
        """transformer layers whose FFN fits the fused bf16 kernels (d_model 256, dim_feedforward 512; csrc/ffn_fused.hip):
        offsets of (linear1.weight, linear1.bias, linear2.weight, norm2.weight, norm2.bias) in the flat buffer and the
        buffers of the packed weight images (filled by ensure() on every forward: the weights move every step)"""
        self._ffn = None
        if os.environ.get("DSVG_FFN_FUSED", "1") == "0":
            return
        rows, index = [], {}
        for m in self.module.modules():
            l1, l2, n2 = getattr(m, "linear1", None), getattr(m, "linear2", None), getattr(m, "norm2", None)
            if not (isinstance(l1, nn.Linear) and isinstance(l2, nn.Linear) and isinstance(n2, nn.LayerNorm)):
                continue
            if tuple(l1.weight.shape) != (512, 256) or tuple(l2.weight.shape) != (256, 512) or l1.bias is None:
                continue
            ps = (l1.weight, l1.bias, l2.weight, n2.weight, n2.bias)
            if any(id(p) not in self.index for p in ps):
                continue
            index[id(l1.weight)] = len(rows)
            rows.append([self.index[id(p)][0] for p in ps])
        if not rows:
            return
        n = len(rows)
        self._ffn = dict(n=n, index=index, offs=torch.tensor(rows, dtype=torch.int64, device=device),
                         fwd=torch.empty(n * ops.FFN_FWD_LAYER_ELEMS, dtype=torch.bfloat16, device=device),
                         bwd=torch.empty(n * ops.FFN_BWD_LAYER_ELEMS, dtype=torch.bfloat16, device=device),
                         b1f=torch.empty((n, 512), dtype=torch.float32, device=device),
                         w2p=torch.empty((n, 256, 512), dtype=torch.bfloat16, device=device))

    def _attn_setup(self, device):
        """transformer layers whose attention sub-block fits the fused bf16 kernel (d_model 256, 8 heads;
        csrc/attn_fused.hip): offsets of (in_proj_weight, out_proj.weight) in the flat buffer and the buffer of the packed
        weight images (filled by ensure() on every forward)"""
        self._attn = None
        if os.environ.get("DSVG_ATTN_FUSED", "1") == "0":
            return
        rows, index = [], {}
        for m in self.module.modules():
            w_in, op = getattr(m, "in_proj_weight", None), getattr(m, "out_proj", None)
            if not (isinstance(w_in, nn.Parameter) and isinstance(op, nn.Linear)):
                continue
            if tuple(w_in.shape) != (768, 256) or tuple(op.weight.shape) != (256, 256):
                continue
            if id(w_in) not in self.index or id(op.weight) not in self.index:
                continue
            index[id(w_in)] = len(rows)
            rows.append([self.index[id(w_in)][0], self.index[id(op.weight)][0]])
        if not rows:
            return
        n = len(rows)
        self._attn = dict(n=n, index=index, offs=torch.tensor(rows, dtype=torch.int64, device=device),
                          img=torch.empty(n * ops.ATTN_LAYER_ELEMS, dtype=torch.bfloat16, device=device),
                          bwd=torch.empty(n * ops.ATTN_BWD_LAYER_ELEMS, dtype=torch.bfloat16, device=device))

    def _gs_setup(self, device):
        """layers of the short-sequence ("group") stacks - hierarchical_encoder / hierarchical_decoder, model.py:153-161,
        246-254 - that fit the fused per-layer kernels (d_model 256, dim_ff 512, 8 heads; csrc/group_stage.hip): offsets of
        (in_proj_weight, out_proj.weight, linear1.weight, linear2.weight) in the flat buffer and the buffers of the packed
        forward / backward weight images (filled by ensure() on every forward)"""
        self._gs = None
        if os.environ.get("DSVG_GS_FUSED", "1") == "0":
            return
        rows, index = [], {}
        for name, stack in self.module.named_modules():
            # (... and the second decoder stage: the remainder sequences of its training forward, functional.LayerFn)
            if not (isinstance(stack, _Stack) and (name.rsplit(".", 1)[-1].startswith("hierarchical_")
                                                   or name == "decoder.decoder")):
                continue
            for m in stack.layers:
                sa = getattr(m, "self_attn", None)
                ps = (getattr(sa, "in_proj_weight", None), getattr(getattr(sa, "out_proj", None), "weight", None),
                      m.linear1.weight, m.linear2.weight)
                if any(p is None or id(p) not in self.index for p in ps):
                    continue
                if [tuple(p.shape) for p in ps] != [(768, 256), (256, 256), (512, 256), (256, 512)]:
                    continue
                if hasattr(m, "linear_global2"):        # label-conditioned layers keep the unfused launches
                    continue
                index[id(ps[0])] = len(rows)
                rows.append([self.index[id(p)][0] for p in ps])
        if not rows:
            return
        n = len(rows)
        self._gs = dict(n=n, index=index, offs=torch.tensor(rows, dtype=torch.int64, device=device),
                        fwd=torch.empty(n * ops.GS_LAYER_ELEMS, dtype=torch.bfloat16, device=device),
                        bwd=torch.empty(n * ops.GS_LAYER_ELEMS, dtype=torch.bfloat16, device=device))

    def gs(self, w_in):
        """(packed forward image, packed backward image) of the group-stage layer whose in_proj_weight is w_in, or None when
        that layer does not run on the fused per-layer kernels"""
        g = self._gs
        if g is None or self.flat_lp is None or self.flat_lp.dtype != torch.bfloat16:
            return None
        i = g["index"].get(id(w_in))
        if i is None:
            return None
        sl = slice(i * ops.GS_LAYER_ELEMS, (i + 1) * ops.GS_LAYER_ELEMS)
        return g["fwd"][sl], g["bwd"][sl]

    def attn(self, w_in):
        """packed in_proj / out_proj image of the layer whose in_proj_weight is w_in, or None when that layer does not run
        on the fused attention kernel"""
        a = self._attn
        if a is None or self.flat_lp is None or self.flat_lp.dtype != torch.bfloat16:
            return None
        i = a["index"].get(id(w_in))
        if i is None:
            return None
        return a["img"][i * ops.ATTN_LAYER_ELEMS:(i + 1) * ops.ATTN_LAYER_ELEMS]

    def attn_bwd(self, w_in):
        """packed out_proj^T image (attention backward with the out_proj backward inside) of the layer whose in_proj_weight
        is w_in, or None"""
        a = self._attn
        if a is None or self.flat_lp is None or self.flat_lp.dtype != torch.bfloat16:
            return None
        i = a["index"].get(id(w_in))
        if i is None:
            return None
        return a["bwd"][i * ops.ATTN_BWD_LAYER_ELEMS:(i + 1) * ops.ATTN_BWD_LAYER_ELEMS]

    def ffn(self, w1):
        """(packed forward image, packed backward image, folded linear1 bias, linear2.weight with fragment-ordered
        columns) of the layer whose linear1.weight is w1, or None when that layer does not run on the fused FFN kernels"""
        f = self._ffn
        if f is None or self.flat_lp is None or self.flat_lp.dtype != torch.bfloat16:
            return None
        i = f["index"].get(id(w1))
        if i is None:
            return None
        return (f["fwd"][i * ops.FFN_FWD_LAYER_ELEMS:(i + 1) * ops.FFN_FWD_LAYER_ELEMS],
                f["bwd"][i * ops.FFN_BWD_LAYER_ELEMS:(i + 1) * ops.FFN_BWD_LAYER_ELEMS], f["b1f"][i], f["w2p"][i])

    def ensure(self, device, dtype, advance=None):
        params = self.params
        stale = self.flat is None or self.flat.device != device or not params
        if not stale:
            # every parameter must still be the view of the flat buffer it was given (a re-assigned .data, an added or
            # replaced nn.Parameter, load_state_dict(assign=True) ... all re-flatten)
            base, index, n = self.flat.data_ptr(), self.index, 0
            for p in self.module.parameters():
                ent = index.get(id(p))
                if ent is None or p.data_ptr() != base + 4 * ent[0]:
                    stale = True
                    break
                n += 1
            stale = stale or n != len(params)
        if stale:
            self._flatten(device)
        self._in_flight.clear()     # a forward starts a new graph: nothing handed out earlier can still be pending
        # the step counter / dropout seed advance of a training step (TrainStep hands it over, `pending_advance`; a model that
        # owns its seed passes it): it rides on the one-launch image refresh when there is one
        advance = advance if advance is not None else self.pending_advance
        self.pending_advance = None
        if dtype != torch.float32:
            if self.flat_lp is None or self.flat_lp.dtype != dtype:
                self.flat_lp = torch.empty(self.flat.numel(), dtype=dtype, device=device)
                self._lp_views = {}
            if dtype == torch.bfloat16 and PACK_ONE_LAUNCH and ops.pack_images_ok(self.flat, self.flat_lp):
                # bf16 copy + every fragment image (+ the advance) = ONE launch instead of 7 (+ 1): dsvg_pack_images
                c, sd = advance if advance is not None else (None, None)
                ops.pack_images(self.flat, self.flat_lp, self._ffn, self._attn, self._gs,
                                attn_bwd=torch.is_grad_enabled(), counter=c, seed=sd)
                return
            if advance is not None:
                ops.advance_step_(*advance)
                advance = None
            ops.cast_weights(self.flat, self.flat_lp)
            if dtype == torch.bfloat16 and self._ffn is not None:
                f = self._ffn
                ops.ffn_pack(self.flat, f["offs"], f["n"], f["fwd"], f["bwd"], f["b1f"], f["w2p"])
            if dtype == torch.bfloat16 and self._attn is not None:
                a = self._attn
                ops.attn_pack(self.flat, a["offs"], a["n"], a["img"])
                if torch.is_grad_enabled():     # (only a backward pass reads it)
                    ops.attn_pack_bwd(self.flat, a["offs"], a["n"], a["bwd"])
            if dtype == torch.bfloat16 and self._gs is not None:
                g = self._gs
                ops.gs_pack(self.flat, g["offs"], g["n"], g["fwd"], g["bwd"])
        if advance is not None:
            ops.advance_step_(*advance)

    def lp(self, param):
        v = self._lp_views.get(id(param))
        if v is not None:
            return v
        ent = self.index.get(id(param))
        if ent is None or self.flat_lp is None:
            return None
        o, n, shape = ent
        v = self._lp_views[id(param)] = self.flat_lp[o:o + n].view(shape)
        return v

    def grad_buffer(self, which=0):
        if self.gbuf[which] is None:
            self.gbuf[which] = torch.zeros(self.flat.numel(), dtype=torch.float32, device=self.flat.device)
            self._grad_views[which] = {}
        return self.gbuf[which]

    def _grad_view(self, param, which):
        v = self._grad_views[which].get(id(param))
        if v is None:
            ent = self.index.get(id(param))
            if ent is None:
                return None
            o, n, shape = ent
            buf = self.grad_buffer(which)
            v = self._grad_views[which][id(param)] = buf[o:o + n].view(shape)
        return v

    def grad_view(self, param):
        v = self._grad_view(param, 0)
        if v is None:
            return None
        key = id(param)
        if key in self._in_flight:
            # a second backward node asks for this parameter's gradient before AccumulateGrad has consumed the first
            # one (two forwards + one backward, a weight used by two Functions): autograd SUMS the contributions, so
            # they must not share memory - hand out a private tensor.  (A trainer may have queued the reduction that
            # fills the first one: autograd is about to read it.)
            ops.flush_deferred()
            if getattr(self.module, "_defer_wgrad", False) or ops.defer_active():
                # the reduction that fills this private tensor would be QUEUED, and autograd sums the two contributions
                # as soon as the node returns - before any flush.  No shipped config shares a parameter between two
                # Functions; one that does has to run with immediate reductions
                raise RuntimeError("a parameter used by two autograd nodes received a second gradient while deferred "
                                   "reductions are on: use TrainStep(...).defer_reductions = False (DSVG_DEFER_REDUCE=0)")
            return torch.empty(param.shape, dtype=torch.float32, device=v.device)
        self._in_flight.add(key)
        # autograd accumulates into an existing .grad: never hand it a view that aliases that .grad
        if param.grad is not None and param.grad.data_ptr() == v.data_ptr():
            v = self._grad_view(param, 1)
        # a fresh tensor object per call: AccumulateGrad only adopts ("steals") a gradient nobody else references,
        # otherwise it deep-copies it - the cached view itself would cost one copy launch per parameter per step
        return v.detach() if v is not None else None


def _compute_dtype_default():
    name = os.environ.get("DSVG_DTYPE", "fp32").lower()
    return torch.bfloat16 if name in ("bf16", "bfloat16") else torch.float32


# ----------------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------------
class SVGTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        if cfg.model_type != "transformer":
            raise NotImplementedError("model_type='lstm' (SketchRNN baseline) is outside the MI355X hot path")
        if cfg.pred_mode not in ("one_shot", "autoregressive"):
            raise ValueError(f"unknown pred_mode {cfg.pred_mode!r}")
        if cfg.pred_mode == "autoregressive" and cfg.decode_stages != 1:
            raise NotImplementedError("autoregressive decoding is built for the one-stage decoder")
        longest = (cfg.max_total_len if cfg.encode_stages == 1 or cfg.decode_stages == 1 else cfg.max_seq_len) + 2
        if longest > 256:
            raise NotImplementedError(f"sequences of {longest} tokens: the long-sequence attention kernel holds one "
                                      "sequence per workgroup in LDS, at most 256 tokens")
        if (cfg.encode_stages == 2 or cfg.decode_stages == 2) and cfg.max_seq_len + 2 > 64:
            raise NotImplementedError("two-stage configs are built for groups of at most 62 commands")
        if cfg.d_model // cfg.n_heads != 32 or cfg.d_model % cfg.n_heads:
            raise NotImplementedError("the attention kernel is specialised for head_dim == 32")
        self.args_dim = 2 * cfg.args_dim if cfg.rel_targets else cfg.args_dim + 1

        if cfg.encode_stages > 0:
            self.encoder = _Encoder(cfg)
            if cfg.use_resnet:
                self.resnet = _ResNet(cfg.d_model)
            if cfg.use_vae:
                self.vae = _VAE(cfg)
            else:
                self.bottleneck = _Bottleneck(cfg)
        self.decoder = _Decoder(cfg, self.args_dim)
        self.register_buffer("cmd_args_mask", CMD_ARGS_MASK.clone())

        self.compute_dtype = _compute_dtype_default()
        self._store = ParamStore(self)
        self._seed = None           # int64[1] device tensor (dropout seed of the current step)
        self._own_seed = True       # advance the seed on every training forward unless a trainer drives it
        # first encoder stage on the valid tokens only (exact); DSVG_PACK_ENCODER=0 -> padded layout
        self.pack_encoder = os.environ.get("DSVG_PACK_ENCODER", "1") != "0"
        self.last_packing = None
        # backward of the second decoder stage only over the sequences of visible target groups (exact under SVGLoss)
        self.skip_invisible_backward = os.environ.get("DSVG_SKIP_INVISIBLE", "1") != "0"
        # ... and the same sequences' forward pass in a training call (their logits stay available: lazy result entries)
        self.skip_invisible_forward = os.environ.get("DSVG_SKIP_INVISIBLE_FWD", "1") != "0"
        # ... with the heads and the loss reading that stage's rows in its own (visible-first) order
        self.heads_visible_first = os.environ.get("DSVG_HEADS_VF", "1") != "0"
        self._cmd_logits_live = None
        self.last_live = None
        # backward of the argument head only over the tokens that carry argument loss (exact under SVGLoss)
        self.compact_head_backward = os.environ.get("DSVG_COMPACT_HEAD", "1") != "0"
        # argument head + loss on the argument slots that carry loss in the batch only (see _plan)
        self.head_slot_range = os.environ.get("DSVG_HEAD_SLOT_RANGE", "1") != "0"
        # the decoder stacks' conditioning rows linear_global_l(z) computed for all layers up front (functional.GlobalCondFn)
        self.hoist_global = os.environ.get("DSVG_HOIST_GLOBAL", "1") != "0"
        self.last_head_rows = None
        self.kv_cache = True         # autoregressive sampling: incremental decoding over a per-layer q|k|v cache
        # temperature > 0 sampling as a Gumbel arg-max on the device (False: torch.distributions.Categorical over the dense
        # logits, the reference's own call - deepsvg/model/utils.py:75-79)
        self.device_sampling = True
        self.last_assignment = None  # self-matching configs: (N, Gp) int32 assignment of the last training forward
        self._forced_plan = None
        self._decoder_grads_ready = None    # callback of a data-parallel trainer (TrainStep), see forward()
        # queue the parameter-gradient reductions of the backward pass (ops.DEFER): only a trainer that calls
        # ops.flush_deferred() before anything reads a gradient may set it (TrainStep)
        self._defer_wgrad = False
        self._rt = None
        self._live = None

    # ---- runtime plumbing ------------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        return self

    def decoder_param_range(self):
        """[lo, hi) of the decoder's parameters inside the flat parameter / gradient buffers (they come last)"""
        st = self._store
        dec = [st.index[id(p)] for p in self.decoder.parameters()]
        lo = min(e[0] for e in dec)
        assert all(st.index[id(p)][0] < lo for n, p in self.named_parameters() if not n.startswith("decoder.")), \
            "decoder parameters are expected at the end of the flat buffer"
        return lo, st.flat.numel()

    @property
    def store(self):
        return self._store

    def seed_tensor(self, device):
        if self._seed is None or self._seed.device != device:
            self._seed = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
        return self._seed

    def _runtime(self, device):
        training = self.training
        seed = self.seed_tensor(device) if training else None
        self._store.ensure(device, self.compute_dtype, advance=(None, seed) if training and self._own_seed else None)
        self._rt = Fn.Runtime(self.compute_dtype, seed, self._store, training,
                              defer=self._defer_wgrad and (not ops.PROFILE_ON or ops.PROFILE_KEEP_DEFER))
        return self._rt

    # ---- blocks ----------------------------------------------------------------------------------
    def _run_stack(self, rt, stack, x, key_mask, z, n_seq, S, site, seq_off=None, live=None, tiles=None, l=None,
                   causal=False):
        """l: label embedding rows [n_seq, dim_label] of a label-conditioned config (memory2 of the reference layers,
        layers/improved_transformer.py:47-49,134-136)"""
        cfg = self.cfg
        # the decoder's conditioning rows linear_global_l(z) of all layers up front (z is the same for every layer)
        gl = None
        if z is not None and all(hasattr(L, "linear_global") for L in stack.layers) and self.hoist_global:
            wb = [t for L in stack.layers for t in (L.linear_global.weight, L.linear_global.bias)]
            gl = Fn.GlobalCondFn.apply(rt, z, *wb)
        n = len(stack.layers)
        all_g = all(hasattr(L, "linear_global") for L in stack.layers)
        if (n > 0 and not causal and seq_off is None and live is None and tiles is None and l is None
                and (z is None or (gl is not None and all_g))
                and Fn.gs_stack_eligible(rt, x, key_mask, n_seq, S, cfg.n_heads, n, [L.self_attn.in_proj_weight for L in stack.layers])):
            # a short-sequence ("group") stage: the whole stack in ONE launch per direction (functional.GsStackFn)
            ts = []
            for i, L in enumerate(stack.layers):
                ts += [L.norm1.weight, L.norm1.bias, L.self_attn.in_proj_weight, L.self_attn.in_proj_bias,
                       L.self_attn.out_proj.weight, L.self_attn.out_proj.bias, L.norm2.weight, L.norm2.bias,
                       L.linear1.weight, L.linear1.bias, L.linear2.weight, L.linear2.bias]
                if z is not None:
                    ts.append(gl[i])
            x = Fn.GsStackFn.apply(rt, x, key_mask, n_seq, S, cfg.n_heads, cfg.dropout, site, n, z is not None, *ts)
            return Fn.LayerNormFn.apply(rt, x, stack.norm.weight, stack.norm.bias, stack.norm.eps, live, None)
        for i, L in enumerate(stack.layers):
            has_g = hasattr(L, "linear_global")
            has_l = l is not None and hasattr(L, "linear_global2")
            hoisted = gl is not None
            x = Fn.LayerFn.apply(
                rt, x, key_mask, (gl[i] if hoisted else z) if has_g else None, l if has_l else None, n_seq, S, cfg.n_heads,
                cfg.dropout, site + 8 * i,
                L.norm1.weight, L.norm1.bias, L.self_attn.in_proj_weight, L.self_attn.in_proj_bias,
                L.self_attn.out_proj.weight, L.self_attn.out_proj.bias, L.norm2.weight, L.norm2.bias,
                L.linear1.weight, L.linear1.bias, L.linear2.weight, L.linear2.bias,
                L.linear_global.weight if (has_g and not hoisted) else None,
                L.linear_global.bias if (has_g and not hoisted) else None,
                L.linear_global2.weight if has_l else None, L.linear_global2.bias if has_l else None,
                seq_off, live, tiles, causal,
                # the layer below reads this layer's input gradient through the mask of ITS FFN residual dropout (site + 4)
                (site + 8 * (i - 1) + 4) if i > 0 else None, i == 0)
        n = len(stack.layers)
        return Fn.LayerNormFn.apply(rt, x, stack.norm.weight, stack.norm.bias, stack.norm.eps, live,
                                    (cfg.dropout, site + 8 * (n - 1) + 4) if (n > 0 and not rt.last_layer_gs) else None)

    def make_plan(self, commands_enc, args_enc, commands_dec, want_grad=True, args_dec=None):
        """the data-dependent layout plan of one forward (see _plan), for callers that replay captured hipGraphs"""
        forced, self._forced_plan = self._forced_plan, None
        try:
            return self._plan(commands_enc, args_enc, commands_dec, want_grad, args_dec)
        finally:
            self._forced_plan = forced

    def _plan(self, commands_enc, args_enc, commands_dec, want_grad, args_dec=None):
        """Data-dependent layout decisions of one forward, made up front with ONE device->host read:
          * packed first encoder stage (valid tokens only), see _encode_stage1_packed;
          * visible-first order of the second decoder stage: SVGLoss excludes every position of an invisible target
            group (loss.py:36,51-54) and the stage-2 sequences are independent, so the backward pass of those
            sequences is identically zero and only a row prefix has to be processed.  Only used
            when the targets are known (training call with commands_dec) - gradients are those of SVGLoss.
        Both are disabled while a hipGraph is being captured (shapes must be static there)."""
        cfg = self.cfg
        if self._forced_plan is not None:       # a trainer replaying bucketed hipGraphs supplies the plan (make_plan)
            return self._forced_plan
        plan = {"enc": None, "dec": None, "loss": None}
        ref = commands_enc if commands_enc is not None else commands_dec
        if ref is None or (ref.is_cuda and torch.cuda.is_current_stream_capturing()):
            return plan
        counts = []
        # (label-conditioned configs add a per-sequence term in every layer: they keep the padded / dense layouts)
        if (commands_enc is not None and self.pack_encoder and cfg.encode_stages > 0
                and not self.encoder.use_group and not cfg.label_condition):
            N, G, S = commands_enc.shape
            cmd = commands_enc.to(torch.float32).contiguous().view(N * G, S)
            arg = args_enc.to(torch.float32).contiguous().view(N * G * S, -1)
            key_mask, _vis, group_mask = ops.build_masks(cmd, S, G, EOS_ID, want_group_mask=cfg.encode_stages == 2)
            seq_off, pcmd, parg, ppos = ops.pack_tokens(cmd.view(-1), arg, key_mask, N * G, S)
            tiles = ops.attention_tiles(seq_off, N * G, 32) if S <= 32 else None
            plan["enc"] = dict(key_mask=key_mask, group_mask=group_mask, seq_off=seq_off, pcmd=pcmd, parg=parg, ppos=ppos,
                               tiles=tiles)
            counts.append(seq_off[-1:])
        if (commands_dec is not None and want_grad and self.skip_invisible_backward and cfg.decode_stages == 2
                and commands_dec.shape[1] == cfg.num_groups_proposal and not cfg.label_condition
                and not cfg.self_match):        # (self-matching pairs predictions with targets only after the forward)
            N, G, St = commands_dec.shape
            cmd_t = commands_dec.to(torch.float32).contiguous().view(N * G, St)
            _km, vis, _gm = ops.build_masks(cmd_t, St, G, EOS_ID)
            new_of_old, old_of_new, nvis = ops.visible_first(vis)
            plan["dec"] = dict(new_of_old=new_of_old, old_of_new=old_of_new)
            counts.append(nvis)
        if commands_dec is not None and want_grad and self.compact_head_backward and args_dec is not None:
            # targets / weights of SVGLoss (loss.py:33-54) and the tokens that carry argument loss: the backward of
            # the 2827-wide argument head then runs on those tokens only (functional.ArgsHeadLossFn)
            N, G, S1 = commands_dec.shape
            tc = commands_dec.to(torch.float32).contiguous().view(N * G, S1)
            ta = args_dec.to(torch.float32).contiguous().view(N * G, S1, -1)
            cam = self.cmd_args_mask.to(device=tc.device, dtype=torch.float32).contiguous()
            # with the visible-first order of the second decoder stage known, the heads read that stage's rows where they
            # are: command / argument targets, weights and the token list in THAT order (no gather back to the caller's
            # group order in the training step; the visibility targets - first-stage rows - stay in the caller's order)
            vf = plan["dec"] is not None and self.skip_invisible_forward and self.heads_visible_first
            targets = ops.loss_targets(tc, ta, cam, EOS_ID, seq_perm=plan["dec"]["old_of_new"] if vf else None)
            n_args = ta.shape[-1]
            live, n_live = ops.live_rows(targets[3].view(-1), n_args)
            plan["loss"] = dict(targets=targets, live=live, vf=vf)
            counts.append(n_live)
            # which argument slots carry loss anywhere in this batch (CMD_ARGS_MASK columns of the commands present: real
            # DeepSVG data has no arcs, so slots 0-4 never do): the head then runs on that slot range only
            # (column sums of the 0 / 1 weights by the library's column-sum kernel: torch's `any(0)` over 127 k rows x 11
            # columns is a 350 us launch)
            counts.append((ops.colsum(targets[3].view(-1, n_args)) > 0).to(torch.int32))
        if counts:
            vals = (torch.cat(counts) if len(counts) > 1 else counts[0]).tolist()      # the one host read
            if plan["enc"] is not None:
                plan["enc"]["total"] = int(vals.pop(0))
            if plan["dec"] is not None:
                plan["dec"]["n_visible"] = int(vals.pop(0))
            if plan["loss"] is not None:
                plan["loss"]["n_live"] = int(vals.pop(0))
                used = [i for i, v in enumerate(vals) if v]         # (the slot flags are the tail of the read)
                n_args = len(vals)
                lo, hi = (used[0], used[-1] + 1) if (used and self.head_slot_range) else (0, n_args)
                plan["loss"]["slot_lo"], plan["loss"]["slot_hi"] = lo, hi
                if (lo, hi) != (0, n_args):
                    t = plan["loss"]["targets"]
                    plan["loss"]["targets_r"] = (t[2].view(-1, n_args)[:, lo:hi].contiguous().view(-1),
                                                 t[3].view(-1, n_args)[:, lo:hi].contiguous().view(-1))
        return plan


    def _encode_stage1_packed(self, rt, pe, n_seq, S):
        """First encoder stage on the valid tokens only: rows past a sequence's first EOS are
        masked as keys (layers/functional.py:234-239) and dropped by the mean-pool (model.py:137), so they reach
        neither an output nor a gradient.  Tokens are packed back to back; the row count is rounded up to a
        multiple of 128 with inert rows (finite activations, exactly-zero gradients).  The token count comes from
        the forward's single device->host read (_plan; the reference's own loss syncs too, loss.py:53-54)."""
        enc = self.encoder
        emb = enc.embedding
        seq_off, pcmd, parg, ppos, total = pe["seq_off"], pe["pcmd"], pe["parg"], pe["ppos"], pe["total"]
        rows = min((total + 127) // 128 * 128, n_seq * S)
        rows = max(rows, min(int(pe.get("rows", 0)), n_seq * S))     # a graph bucket may ask for more (inert) rows
        self.last_packing = (total, n_seq * S)      # valid tokens, dense tokens (reported by bench.py)
        src = Fn.PackedEmbedFn.apply(rt, pcmd[:rows], parg[:rows], ppos[:rows], S, PE_DROPOUT, 1,
                                     emb.command_embed.weight, emb.arg_embed.weight, emb.embed_fcn.weight,
                                     emb.embed_fcn.bias, emb.pos_encoding.pos_embed.weight)
        mem = self._run_stack(rt, enc.encoder, src, None, None, n_seq, S, 100, seq_off=seq_off, tiles=pe.get("tiles"))
        return Fn.MaskedMeanFn.apply(rt, mem, None, n_seq, S, seq_off)

    def _label_rows(self, rt, part, label, N):
        """rows of `part`'s own label table, one per icon (model.py:123,155,245)"""
        if label is None:
            raise ValueError("label_condition=True: forward needs `label`")
        label = label.reshape(-1)
        if label.numel() != N:
            raise ValueError(f"label has {label.numel()} entries for {N} icons")
        return Fn.LabelEmbedFn.apply(rt, label, part.label_embedding.label_embedding.weight)

    def _encode(self, rt, commands, args, plan=None, label=None):
        """commands (N, G, S) / args (N, G, S, n_args) float32, batch-first  ->  z [N, d_model]"""
        cfg = self.cfg
        enc = self.encoder
        N, G, S = commands.shape
        two = cfg.encode_stages == 2
        emb = enc.embedding
        pe = plan["enc"] if plan is not None else None
        l = l_seq = None
        if cfg.label_condition:
            l = self._label_rows(rt, enc, label, N)
            l_seq = l.repeat_interleave(G, dim=0) if G > 1 else l         # sequence (n, g) carries icon n's label
        if pe is not None:
            group_mask = pe["group_mask"]
            z = self._encode_stage1_packed(rt, pe, N * G, S)
        else:
            cmd = commands.to(torch.float32).contiguous().view(N * G, S)
            arg = args.to(torch.float32).contiguous().view(N * G * S, -1)
            key_mask, _vis, group_mask = ops.build_masks(cmd, S, G, EOS_ID, want_group_mask=two)
            self.last_packing = None
            groups = ops.group_index(cmd, S, M_ID) if enc.use_group else None
            src = Fn.EmbedFn.apply(rt, cmd.view(-1), arg, groups, N * G, S, PE_DROPOUT, 1,
                                   emb.command_embed.weight, emb.arg_embed.weight, emb.embed_fcn.weight,
                                   emb.embed_fcn.bias, emb.pos_encoding.pos_embed.weight,
                                   emb.group_embed.weight if enc.use_group else None)
            mem = self._run_stack(rt, enc.encoder, src, key_mask, None, N * G, S, 100, l=l_seq)
            z = Fn.MaskedMeanFn.apply(rt, mem, key_mask, N * G, S)          # [N*G, d]
        if two:
            # the self-matching variant has no group positional encoding (model.py:114-115,157-158)
            src2 = z if cfg.self_match else \
                Fn.AddPosFn.apply(rt, z, enc.hierarchical_PE.pos_embed.weight, N, G, PE_DROPOUT, 2)
            mem2 = self._run_stack(rt, enc.hierarchical_encoder, src2, group_mask, None, N, G, 200, l=l)
            z = Fn.MaskedMeanFn.apply(rt, mem2, group_mask, N, G)       # [N, d]
        return z

    def _bottleneck(self, rt, z):
        cfg = self.cfg
        mu = logsigma = None
        if (Fn.LATENT_FUSED and cfg.use_resnet and not cfg.use_vae and z.dtype == torch.bfloat16 and z.shape[1] == 256
                and tuple(self.bottleneck.bottleneck.weight.shape) == (256, 256) and rt.store is not None):
            # the four residual blocks and the bottleneck linear in one launch per direction (csrc/group_stage.hip)
            wb = [t for i in range(1, 5) for t in (getattr(self.resnet, f"linear{i}")[0].weight,
                                                   getattr(self.resnet, f"linear{i}")[0].bias)]
            wb += [self.bottleneck.bottleneck.weight, self.bottleneck.bottleneck.bias]
            return Fn.LatentChainFn.apply(rt, z, *wb), mu, logsigma
        if cfg.use_resnet:
            for i in range(1, 5):
                lin = getattr(self.resnet, f"linear{i}")[0]
                z = Fn.ResBlockFn.apply(rt, z, lin.weight, lin.bias)
        if cfg.use_vae:
            mu = Fn.LinearFn.apply(rt, z, self.vae.enc_mu_fcn.weight, self.vae.enc_mu_fcn.bias, 0, None, 0.0, 0,
                                   torch.float32)
            logsigma = Fn.LinearFn.apply(rt, z, self.vae.enc_sigma_fcn.weight, self.vae.enc_sigma_fcn.bias, 0, None,
                                         0.0, 0, torch.float32)
            # reparametrisation (model.py:184-185): N x dim_z elementwise + randn, left to torch (not a hot op)
            sigma = torch.exp(logsigma / 2.0)
            z = (mu + sigma * torch.randn_like(sigma)).to(rt.dtype)
        else:
            z = Fn.LinearFn.apply(rt, z, self.bottleneck.bottleneck.weight, self.bottleneck.bottleneck.bias, 0, None,
                                  0.0, 0, None)
        return z, mu, logsigma

    def _decode(self, rt, z, plan=None, lazy_args=False, label=None, hierarch_logits=None, return_hierarch=False,
                match=None, prefix=None):
        """z [N, dim_z] -> command_logits (N,G,S,n_cmd), args_logits (N,G,S,n_args,args_dim)[, visibility (N,G,1,2)];
        with lazy_args the second result is a thunk that computes args_logits when called.
        hierarch_logits [N*G, 2] given: z is the per-group latent [N*G, dim_z] and the first decoder stage is skipped
        (model.py:249-253); return_hierarch: stop after the first stage and return (visibility logits, per-group z),
        both [N*G, .] (model.py:260-261)."""
        cfg = self.cfg
        dec = self.decoder
        vis_logits = None
        l = l_seq = None
        self._head_vf = False       # (set below by the one-shot path only: an autoregressive call must not inherit the last call's)
        if cfg.decode_stages == 2:
            G = cfg.num_groups_proposal
            N = z.shape[0] if hierarch_logits is None else z.shape[0] // G
            if cfg.label_condition:
                l = self._label_rows(rt, dec, label, N)
                l_seq = l.repeat_interleave(G, dim=0)
            if hierarch_logits is None:
                src = Fn.AddPosFn.apply(rt, None, dec.hierarchical_embedding.PE.pos_embed.weight, N, G, PE_DROPOUT, 3)
                out = self._run_stack(rt, dec.hierarchical_decoder, src, None, z, N, G, 300, l=l)
                hf = dec.hierarchical_fcn
                vis_logits = Fn.LinearFn.apply(rt, out, hf.visibility_fcn.weight, hf.visibility_fcn.bias, 0, None, 0.0,
                                               0, None)
                z = Fn.LinearFn.apply(rt, out, hf.z_fcn.weight, hf.z_fcn.bias, 0, None, 0.0, 0, None)  # [N*G, dim_z]
            else:
                vis_logits = hierarch_logits
            if return_hierarch:
                return vis_logits, z
            n_seq = N * G
        else:
            N = z.shape[0]
            G = 1
            n_seq = N
            if cfg.label_condition:
                l_seq = self._label_rows(rt, dec, label, N)
        if cfg.pred_mode == "autoregressive":
            return self._decode_autoregressive(rt, z, prefix, l_seq, lazy_args)
        S = dec.embedding.seq_len
        pd = plan["dec"] if (plan is not None and cfg.decode_stages == 2) else None
        live = None
        n_run = n_seq               # sequences the stage runs forward
        vf_plan = bool(pd is not None and plan.get("loss") and plan["loss"].get("vf"))     # head targets in that order
        if pd is not None and (max(pd["n_visible"], pd.get("n_live", 0)) < n_seq or vf_plan):
            # visible-first order: sequence `new` of the stage is group old_of_new[new]; backward covers the prefix
            # (any prefix that contains every visible sequence is exact; a graph bucket rounds it up)
            nv = min(max(pd["n_visible"], pd.get("n_live", 0)), n_seq)
            live = Fn.LivePrefix((nv, min((nv * S + 127) // 128 * 128, n_seq * S)))
            if self.skip_invisible_forward and match is None and l_seq is None:
                # ... and so does the forward pass: SVGLoss reads no logit of an invisible target group (loss.py:36,51-54)
                # and the sequences are independent, so the training step runs the whole sequences that cover that prefix.
                # The other groups' logits are LAZY entries of the result (see `complete` below)
                n_run = min(n_seq, -(-live[1] // S))
            z_all = z
            z = Fn.GatherGroupsFn.apply(z, pd["old_of_new"], pd["new_of_old"], n_run, 1, None)
        self.last_live = (live[0], n_seq) if live is not None else None
        self._live = live
        src = Fn.AddPosFn.apply(rt, None, dec.embedding.PE.pos_embed.weight, n_run, S, PE_DROPOUT, 4, live)
        out = self._run_stack(rt, dec.decoder, src, None, z, n_run, S, 400, live=live, l=l_seq)
        # the heads read the stage's rows in visible-first order when the loss plan's targets are in that order (_plan): the
        # training step then never gathers back to the caller's group order; anything else that reads a dense logit tensor
        # gets it through `complete` (lazy)
        vf_heads = bool(live is not None and plan is not None and plan.get("loss") and plan["loss"].get("vf")
                        and match is None and lazy_args)
        self._head_vf = vf_heads    # the head input's rows are in the stage's visible-first order (what SVGLoss checks)
        complete = None
        if live is not None and (n_run < n_seq or vf_heads):
            out_vf = out
            done = []

            def complete():
                """the stage's output with the rows of EVERY group, in the caller's group order: runs the sequences the
                training pass left out (first read of a dense logit tensor by anything but deepsvg_amd.SVGLoss: the
                reference's SVGLoss, a metric, a test).  Differentiable; off the hot path, so the row bookkeeping is
                plain torch.  Dropout: those rows draw from their own sites."""
                if not done:
                    full = out_vf
                    if n_run < n_seq:
                        rest = pd["old_of_new"][n_run:].long()
                        n_rest = n_seq - n_run
                        z_r = z_all.index_select(0, rest)
                        src_r = Fn.AddPosFn.apply(rt, None, dec.embedding.PE.pos_embed.weight, n_rest, S, PE_DROPOUT, 6, None)
                        out_r = self._run_stack(rt, dec.decoder, src_r, None, z_r, n_rest, S, 464)
                        full = torch.cat([out_vf, out_r])
                    done.append(Fn.GatherGroupsFn.apply(full, pd["new_of_old"], pd["old_of_new"], n_seq, S, None))
                return done[0]
            if not vf_heads:        # (heads on caller-order rows: zeros for the sequences that did not run)
                out = Fn.GatherGroupsFn.apply(out, pd["new_of_old"], pd["old_of_new"], n_seq, S, live)
        elif live is not None:
            out = Fn.GatherGroupsFn.apply(out, pd["new_of_old"], pd["old_of_new"], n_seq, S, live)
        if match is not None:
            # Hungarian self-matching (model.py:384-395): cost of every (target group, predicted group) pair from the
            # dense logits (no gradient), exact assignment, then output slot j takes predicted group assign[j].  The
            # heads are per-token linears, so the rows of their INPUT are permuted instead of the three logit tensors.
            tgt_c, tgt_a = match
            fcn = dec.fcn
            with torch.no_grad():
                cl = Fn.LinearFn.apply(rt, out, fcn.command_fcn.weight, fcn.command_fcn.bias, 0, None, 0.0, 0, None)
                al = Fn.LinearFn.apply(rt, out, fcn.args_fcn.weight, fcn.args_fcn.bias, 0, None, 0.0, 0, None)
                cam = self.cmd_args_mask.to(device=out.device, dtype=torch.float32).contiguous()
                cost, vis = ops.match_costs(cl, al, vis_logits.detach(),
                                            tgt_c.to(torch.float32).contiguous(), tgt_a.to(torch.float32).contiguous(),
                                            cam, N, tgt_c.shape[1], G, cfg.n_args, self.args_dim, cfg.n_commands, EOS_ID)
                assign, idx, inv = ops.match_assign(cost, vis)
                del cl, al
            self.last_assignment = assign
            out = Fn.GatherGroupsFn.apply(out, idx, inv, n_seq, S, None)
            vis_logits = vis_logits.index_select(0, idx.long())         # (N*G, 2): a tiny gather, left to torch
        cf = dec.fcn.command_fcn
        cmd_logits = Fn.LinearFn.apply(rt, out, cf.weight, cf.bias, 0, None, 0.0, 0, None)
        if complete is None:
            cmd_logits = cmd_logits.view(N, G, S, cfg.n_commands)
        n_args, args_dim, fcn = cfg.n_args, self.args_dim, dec.fcn.args_fcn

        def make_args_logits():
            al = Fn.LinearFn.apply(rt, out if complete is None else complete(), fcn.weight, fcn.bias, 0, None, 0.0, 0, None)
            return al.view(N, G, S, n_args, args_dim)
        args_logits = make_args_logits if lazy_args else make_args_logits()
        self._head_in = out         # input of the heads (for the fused argument head + loss)
        self._head_argmax = None
        if out.dtype == torch.bfloat16 and out.shape[1] == 256 and args_dim >= 64 and n_args * args_dim <= 3008:
            # decoding at temperature 0: arg-max per (token, argument slot) with the head's logit tile on chip
            # (csrc/head_fused.hip) instead of the dense (N, G, S, n_args, args_dim) logits + an arg-max pass over them
            def head_argmax(temperature=0.0, seed=None):
                src = out if complete is None else complete()
                img = ops.head_pack(rt.w(fcn.weight))
                if temperature > 0:     # the reference's categorical draw, on the on-chip logit tile (Gumbel arg-max)
                    return ops.head_sample(src.detach().contiguous(), img, fcn.bias.detach().float().contiguous(),
                                           n_args * args_dim, args_dim, temperature, seed, ops.SITE_SAMPLE_ARGS) \
                        .view(N, G, S, n_args)
                return ops.head_argmax(src.detach().contiguous(), img, fcn.bias.detach().float().contiguous(),
                                       n_args * args_dim, args_dim).view(N, G, S, n_args)
            self._head_argmax = head_argmax
        self._cmd_logits_live = None
        if complete is not None:
            # valid on the groups that ran (every loss-carrying row is among them), [rows of the head input, n_commands] in
            # the head input's row order: what deepsvg_amd.SVGLoss reads
            self._cmd_logits_live = cmd_logits
            cmd_logits = lambda: Fn.LinearFn.apply(rt, complete(), cf.weight, cf.bias, 0, None, 0.0, 0, None) \
                .view(N, G, S, cfg.n_commands)
        if vis_logits is not None:
            vis_logits = vis_logits.view(N, G, 1, 2)
        return cmd_logits, args_logits, vis_logits

    def _decode_autoregressive(self, rt, z, prefix, l_seq, lazy_args):
        """pred_mode = "autoregressive" (model.py:263-277): the decoder embeds the shifted targets (teacher forcing) or
        the prefix sampled so far and runs causal self-attention.  prefix = (commands (N, 1, S), args (N, 1, S, n_args))
        batch-first float tensors; S <= max_total_len + 1."""
        cfg = self.cfg
        dec = self.decoder
        emb = dec.embedding
        commands, args = prefix
        N, S = commands.shape[0], commands.shape[-1]
        if commands.shape[1] != 1 or z.shape[0] != N:
            raise ValueError("autoregressive decoding takes grouped sequences (N, 1, S) and one latent per icon")
        if S > cfg.max_total_len + 1:
            raise ValueError(f"decoder prefix of {S} tokens exceeds max_total_len + 1 = {cfg.max_total_len + 1}")
        cmd = commands.to(torch.float32).contiguous().view(N, S)
        arg = args.to(torch.float32).contiguous().view(N * S, -1)
        key_mask, _vis, _gm = ops.build_masks(cmd, S, 0, EOS_ID)           # _get_key_padding_mask (model.py:269)
        groups = ops.group_index(cmd, S, M_ID)                               # _get_group_mask (:264)
        src = Fn.EmbedFn.apply(rt, cmd.view(-1), arg, groups, N, S, PE_DROPOUT, 4,
                               emb.command_embed.weight, emb.arg_embed.weight, emb.embed_fcn.weight, emb.embed_fcn.bias,
                               emb.pos_encoding.pos_embed.weight, emb.group_embed.weight)
        out = self._run_stack(rt, dec.decoder, src, key_mask, z, N, S, 400, l=l_seq, causal=True)
        cmd_logits = Fn.LinearFn.apply(rt, out, dec.fcn.command_fcn.weight, dec.fcn.command_fcn.bias, 0, None, 0.0, 0,
                                       None)
        n_args, args_dim, fcn = cfg.n_args, self.args_dim, dec.fcn.args_fcn

        def make_args_logits():
            al = Fn.LinearFn.apply(rt, out, fcn.weight, fcn.bias, 0, None, 0.0, 0, None)
            return al.view(N, 1, S, n_args, args_dim)
        args_logits = make_args_logits if lazy_args else make_args_logits()
        self._head_in = out
        self.last_live = None
        return cmd_logits.view(N, 1, S, cfg.n_commands), args_logits, None

    # ---- public surface (model.py:352-412) ---------------------------------------------------------
    def forward(self, commands_enc, args_enc, commands_dec, args_dec, label=None, z=None, hierarch_logits=None,
                return_tgt=True, params=None, encode_mode=False, return_hierarch=False):
        cfg = self.cfg
        if (hierarch_logits is not None or return_hierarch) and cfg.decode_stages != 2:
            raise ValueError("hierarch_logits / return_hierarch need a two-stage decoder")
        ref = commands_enc if commands_enc is not None else z
        device = ref.device
        ops.require_device(device)
        rt = self._runtime(device)
        mu = logsigma = None
        plan = self._plan(commands_enc if z is None else None, args_enc, commands_dec if return_tgt else None,
                          torch.is_grad_enabled() and not encode_mode, args_dec)
        if z is None:
            zz = self._encode(rt, commands_enc, args_enc, plan, label)
            zz, mu, logsigma = self._bottleneck(rt, zz)
            if self._decoder_grads_ready is not None and zz.requires_grad:
                # data-parallel trainer: the gradient of the bottleneck output is final exactly when every decoder
                # parameter gradient is (the decoder is the only consumer of zz) -> its bucket can be all-reduced
                # while the encoder's backward still runs
                cb = self._decoder_grads_ready
                zz.register_hook(lambda g: (cb(), None)[1])
        else:
            # externally supplied z is batch-first (N, 1, 1, dim_z), or (N, G, 1, dim_z) per-group latents together
            # with hierarch_logits  (model.py:369, 249-253)
            if (hierarch_logits is not None and z.dim() == 4 and z.shape[0] == 1 and z.shape[2] > 1
                    and tuple(z.shape[1:3]) == tuple(hierarch_logits.shape[-3:-1])):
                # per-group latents handed back exactly as return_hierarch produced them, seq-first (1, G, N, dim_z) like
                # hierarch_logits (the reference's notebooks do this with N = 1, where both layouts coincide; for N > 1
                # the reference's own _make_seq_first would scramble them): take them as what they are
                z = z.permute(2, 1, 0, 3)
            elif hierarch_logits is not None and z.dim() == 4 and z.shape[0] != hierarch_logits.shape[-2]:
                raise ValueError(f"z {tuple(z.shape)} does not match hierarch_logits {tuple(hierarch_logits.shape)}: "
                                 "expected batch-first per-group latents (N, G, 1, dim_z) with (1, G, N, 2) logits")
            zz = z.reshape(-1, z.shape[-1]).to(rt.dtype).contiguous()
        if encode_mode:
            return zz.to(torch.float32).view(1, 1, zz.shape[0], zz.shape[1])   # seq-first, as model.py:371
        hl = None
        if hierarch_logits is not None:
            # seq-first (1, G, N, 2), exactly what return_hierarch hands out (the reference does not permute it,
            # model.py:379-380) -> rows n*G + g
            hl = hierarch_logits.reshape(hierarch_logits.shape[-3], hierarch_logits.shape[-2], 2).permute(1, 0, 2) \
                .reshape(-1, 2).to(rt.dtype).contiguous()
            if hl.shape[0] != zz.shape[0]:
                raise ValueError("hierarch_logits (1, G, N, 2) needs the per-group latents z (N, G, 1, dim_z)")
        if return_hierarch:
            vis, zg = self._decode(rt, zz, None, label=label, hierarch_logits=hl, return_hierarch=True)
            G = cfg.num_groups_proposal
            N = vis.shape[0] // G
            # seq-first (1, G, N, .), as the reference returns them (model.py:260-261,382-383)
            return (vis.to(torch.float32).view(N, G, 2).permute(1, 0, 2).unsqueeze(0),
                    zg.to(torch.float32).view(N, G, -1).permute(1, 0, 2).unsqueeze(0))
        lazy_args = bool(return_tgt and plan is not None and plan.get("loss") is not None)
        # a sampling call (greedy_sample): the dense argument logits are only built if somebody reads them - at temperature 0
        # the arg-max comes from the fused head kernel
        sampling = not return_tgt and not torch.is_grad_enabled() and cfg.pred_mode != "autoregressive" and not cfg.self_match
        lazy_args = lazy_args or sampling
        match = None
        if cfg.self_match and return_tgt and commands_dec is not None:      # train-mode call (model.py:384)
            if cfg.decode_stages != 2:
                raise ValueError("self-matching expects a two-stage decoder (model.py:385)")
            match = (commands_dec, args_dec)
        prefix = None
        if cfg.pred_mode == "autoregressive":
            if commands_dec is None:
                raise ValueError("autoregressive decoding needs commands_dec / args_dec (the prefix decoded so far)")
            # train mode feeds the targets shifted by one: everything but the last token (model.py:376-377)
            prefix = (commands_dec[..., :-1], args_dec[..., :-1, :]) if return_tgt else (commands_dec, args_dec)
        cmd_logits, args_logits, vis_logits = self._decode(rt, zz, plan, lazy_args=lazy_args, label=label,
                                                           hierarch_logits=hl, match=match, prefix=prefix)
        if sampling and callable(args_logits):
            # (a sampling call's lazy tensors are built without an autograd graph, whatever the grad mode at read time)
            def _no_grad(fn):
                def run():
                    with torch.no_grad():
                        return fn()
                return run
            args_logits = _no_grad(args_logits)
        res = ModelOutput()
        if callable(cmd_logits):        # (the training pass ran the visible groups only, see _decode)
            res.set_lazy("command_logits", cmd_logits)
        else:
            res["command_logits"] = cmd_logits
        if lazy_args:
            res.set_lazy("args_logits", args_logits)
        else:
            res["args_logits"] = args_logits
        if cfg.decode_stages == 2:
            res["visibility_logits"] = vis_logits
        if sampling and getattr(self, "_head_argmax", None) is not None:
            res["_dsvg_head_argmax"] = dict(fn=self._head_argmax)
        self._head_argmax = None
        if return_tgt:
            res["tgt_commands"] = commands_dec
            res["tgt_args"] = args_dec
            pl = plan["loss"] if plan is not None else None
            self.last_head_rows = None
            if pl is not None:
                # hand SVGLoss what it needs to run the argument head on the loss-carrying tokens only
                T_dec = self._head_in.shape[0]
                n_rows = min(max((pl["n_live"] + 127) // 128 * 128, int(pl.get("rows", 0))), T_dec)
                fcn = self.decoder.fcn.args_fcn
                cf = self.decoder.fcn.command_fcn
                res["_dsvg_head"] = dict(rt=rt, x=self._head_in, weight=fcn.weight, bias=fcn.bias,
                                         cmd_weight=cf.weight, cmd_bias=cf.bias, cmd_logits=self._cmd_logits_live,
                                         targets=pl["targets"], live=(pl["live"], n_rows),
                                         tgt_commands=commands_dec, tgt_args=args_dec,
                                         slots=(pl.get("slot_lo", 0), pl.get("slot_hi", args_dec.shape[-1])),
                                         targets_r=pl.get("targets_r"),
                                         # row order of `targets`: the stage's visible-first order (then `x` / `cmd_logits`
                                         # may cover only the sequences that ran) or the caller's group order
                                         vf=bool(pl.get("vf")), x_vf=bool(getattr(self, "_head_vf", False)))
                self.last_head_rows = (pl["n_live"], T_dec)
            if getattr(self, "_live", None) is not None:
                # the live-prefix backward of the second decoder stage is exact under SVGLoss only: deepsvg_amd.SVGLoss
                # arms it when it consumes THIS output; any other loss gets the full backward (functional.LivePrefix)
                res["_dsvg_live"] = dict(live=self._live, tgt_commands=commands_dec)
            if cfg.use_vae and mu is not None:
                res["mu"] = mu.view(mu.shape[0], 1, 1, -1)
                res["logsigma"] = logsigma.view(logsigma.shape[0], 1, 1, -1)
        self._head_in = None
        self._cmd_logits_live = None
        self._live = None
        return res

    # ---- sampling (model.py:414-479; inference-only host glue on the logits) -------------------------
    @torch.no_grad()
    def greedy_sample(self, commands_enc=None, args_enc=None, commands_dec=None, args_dec=None, label=None,
                      z=None, hierarch_logits=None, concat_groups=True, temperature=0.0001):
        if self.cfg.pred_mode == "autoregressive":
            commands_y, args_y = self._sample_autoregressive(commands_enc, args_enc, label, z, temperature)
            visibility_y = None
        else:
            res = self.forward(commands_enc, args_enc, commands_dec, args_dec, label=label, z=z,
                               hierarch_logits=hierarch_logits, return_tgt=False)
            # restated from /root/reference/deepsvg/model/model.py:414-423,442-448 (one-shot decoding)
            arg_src = None
            if res.is_pending("args_logits") and (temperature == 0 or self.device_sampling):
                arg_src = (dict.get(res, "_dsvg_head_argmax") or {}).get("fn")    # fused head + arg-max / draw of this forward
            commands_y, args_y = self._sample(res["command_logits"], arg_src if arg_src is not None else res["args_logits"],
                                              temperature)
            args_y -= 1   # shift due to -1 PAD_VAL
            visibility_y = None
            if self.cfg.decode_stages == 2:
                scores = torch.softmax(res["visibility_logits"].float(), dim=-1)[..., 1]
                visibility_y = (scores > 0.7).squeeze(-1)
            commands_y, args_y = self._make_valid(commands_y, args_y, visibility_y)
        if self.cfg.rel_targets:
            args_y = self._make_absolute(commands_y, args_y)
        if concat_groups:
            N = commands_y.size(0)
            pm = ((commands_y == EOS_ID).cumsum(dim=-1) == 0)
            commands_y = commands_y[pm].reshape(N, -1)
            args_y = args_y[pm].reshape(N, -1, self.cfg.n_args)
        return commands_y, args_y

    def _sample(self, command_logits, args_logits, temperature):
        """_sample_categorical (deepsvg/model/utils.py:75-80): a draw from softmax(logits / temperature) per slot.
        temperature == 0 (an extension: the reference would divide by zero) takes the limit exactly - the arg-max
        kernel reads the logits in their storage dtype, no fp32 copy / softmax / multinomial over the 2827-wide rows."""
        if temperature == 0 and callable(args_logits):
            # (N, G, S, n_args) arg-max straight from the head's input: the logit tile never leaves the chip
            cs = command_logits.shape
            cl = command_logits.reshape(-1, cs[-1])
            cmd = ops.argmax_rows(cl if cl.stride(-1) == 1 else cl.contiguous(), cs[-1]).long().view(cs[:-1])
            return cmd, args_logits().long().view(*cs[:-1], -1)
        if temperature > 0 and self.device_sampling and (command_logits.is_cuda or ops.sample_rows.__module__ != ops.__name__):
            # the categorical draw on the device: arg-max of logits + temperature * Gumbel noise (counter hash, seeded from
            # torch's generator - torch.manual_seed makes it reproducible), fused into the argument head where the logits
            # are still pending: the (N, G, S, n_args, args_dim) tensor (23 GB at 8192 icons) is never built
            seed = torch.randint(-(1 << 62), 1 << 62, (1,), dtype=torch.int64).to(command_logits.device)
            cs = command_logits.shape
            cl = command_logits.reshape(-1, cs[-1])
            cmd = ops.sample_rows(cl if cl.stride(-1) == 1 else cl.contiguous(), cs[-1], temperature, seed,
                                  ops.SITE_SAMPLE_CMD).long().view(cs[:-1])
            if callable(args_logits):
                return cmd, args_logits(temperature, seed).long().view(*cs[:-1], -1)
            as_ = args_logits.shape
            al = args_logits.reshape(-1, as_[-2] * as_[-1])
            arg = ops.sample_rows(al if al.stride(-1) == 1 else al.contiguous(), as_[-1], temperature, seed,
                                  ops.SITE_SAMPLE_ARGS, group=as_[-2]).long().view(as_[:-1])
            return cmd, arg
        if temperature == 0:
            cs, as_ = command_logits.shape, args_logits.shape
            cl = command_logits.reshape(-1, cs[-1])
            al = args_logits.reshape(-1, as_[-2] * as_[-1])
            cmd = ops.argmax_rows(cl if cl.stride(-1) == 1 else cl.contiguous(), cs[-1]).long().view(cs[:-1])
            arg = ops.argmax_rows(al if al.stride(-1) == 1 else al.contiguous(), as_[-1], as_[-2]).long().view(as_[:-1])
            return cmd, arg
        commands_y = torch.distributions.Categorical(logits=command_logits.float() / temperature).sample()
        args_y = torch.distributions.Categorical(logits=args_logits.float() / temperature).sample()
        return commands_y, args_y

    def _sample_autoregressive(self, commands_enc, args_enc, label, z, temperature):
        """model.py:424-441, for a whole batch.  The reference decodes one icon per call and re-runs the decoder on the
        growing prefix for every new token (O(T^2) token-layers).  Here N icons decode side by side and - causal
        attention makes the earlier tokens' activations final - every step computes ONE new token per icon: its q|k|v
        row goes into a per-layer cache the attention kernel reads (`kv_cache = False`: the reference's re-computation,
        kept as the cross-check).  z: batch-first (N, 1, 1, dim_z) like the one-shot path takes it, or None to encode
        commands_enc / args_enc."""
        cfg = self.cfg
        if z is None:
            z = self.forward(commands_enc, args_enc, None, None, label=label, encode_mode=True).permute(2, 1, 0, 3)
        if self.kv_cache:
            return self._sample_autoregressive_cached(z, label, temperature)
        N = z.shape[0]
        dev = z.device
        commands_y = torch.full((N, 1, 1), float(SOS_ID), device=dev)
        args_y = torch.full((N, 1, 1, cfg.n_args), -1.0, device=dev)
        cam = self.cmd_args_mask.to(dev).bool()
        for _ in range(cfg.max_total_len):
            res = self.forward(None, None, commands_y, args_y, label=label, z=z, return_tgt=False)
            # only the newest position is sampled (:434)
            cmd_new, arg_new = self._sample(res["command_logits"][:, :, -1].contiguous(),
                                            res["args_logits"][:, :, -1].contiguous(), temperature)   # (N, 1[, n_args])
            arg_new = arg_new - 1                                                                 # shift due to PAD_VAL
            arg_new[~cam[cmd_new]] = -1                                                           # _make_valid (:432)
            commands_y = torch.cat([commands_y, cmd_new.unsqueeze(-1).float()], dim=-1)
            args_y = torch.cat([args_y, arg_new.unsqueeze(-2).float()], dim=-2)
        return commands_y[..., 1:].long(), args_y[..., 1:, :].long()       # discard SOS (:436)

    def _sample_autoregressive_cached(self, z, label, temperature):
        """incremental decoding: per step one token row per icon through embedding, every layer and the heads; each
        layer keeps the q|k|v rows of the tokens so far ([N, T + 1, 3 d], zero beyond the prefix) and the causal
        attention kernel reads that cache - row s of its output is the new token's context"""
        cfg, dec = self.cfg, self.decoder
        emb = dec.embedding
        dev = z.device
        ops.require_device(dev)
        training = self.training
        self.eval()
        try:
            rt = self._runtime(dev)
        finally:
            self.train(training)
        N, T, d, H = z.shape[0], cfg.max_total_len, cfg.d_model, cfg.n_heads
        S = T + 1
        dt = rt.dtype
        zz = z.reshape(N, -1).to(dt).contiguous()
        layers = list(dec.decoder.layers)
        gz = [ops.gemm(zz, rt.w(L.linear_global.weight), bias=L.linear_global.bias.detach()) for L in layers]
        g2 = None
        if cfg.label_condition:
            l_rows = Fn.LabelEmbedFn.apply(rt, label.reshape(-1), dec.label_embedding.label_embedding.weight)
            g2 = [ops.gemm(l_rows, rt.w(L.linear_global2.weight), bias=L.linear_global2.bias.detach()) for L in layers]
        cache = [torch.zeros(N * S, 3 * d, dtype=dt, device=dev) for _ in layers]
        ctx_buf = torch.zeros(N * S, d, dtype=dt, device=dev) if S > 64 else None
        # the prefix so far, padded with SOS (any non-EOS value): the key-padding mask is "before the first EOS" (:269)
        cmd_buf = torch.full((N, S), float(SOS_ID), device=dev)
        commands_y = torch.empty(N, 1, T, dtype=torch.long, device=dev)
        args_y = torch.empty(N, 1, T, cfg.n_args, dtype=torch.long, device=dev)
        cam = self.cmd_args_mask.to(dev).bool()
        pos = emb.pos_encoding.pos_embed.weight.detach()
        cmd_s = torch.full((N,), float(SOS_ID), device=dev)
        arg_s = torch.full((N, cfg.n_args), -1.0, device=dev)
        n_m = torch.zeros(N, dtype=torch.int32, device=dev)                 # group index = number of `m` so far (:264)
        scale = float(d // H) ** -0.5
        fcn = dec.fcn
        for s in range(T):
            n_m = n_m + (cmd_s == M_ID).to(torch.int32)
            A, R = ops.embed_gather(cmd_s.contiguous(), arg_s.contiguous(), emb.command_embed.weight.detach(),
                                    emb.arg_embed.weight.detach(), dt, emb.group_embed.weight.detach(), n_m.contiguous())
            pre = ops.gemm(A, rt.w(emb.embed_fcn.weight), bias=emb.embed_fcn.bias.detach(), res=R, res_pre=True)
            x = ops.add_pos_fwd(pre, pos[s:s + 1].contiguous(), N, 1, dt)
            key_mask, _v, _g = ops.build_masks(cmd_buf, S, 0, EOS_ID)
            for li, L in enumerate(layers):
                xn1, _, _ = ops.layernorm_fwd(x, L.norm1.weight.detach(), L.norm1.bias.detach())
                qkv = ops.gemm(xn1, rt.w(L.self_attn.in_proj_weight), bias=L.self_attn.in_proj_bias.detach())
                cache[li].view(N, S, 3 * d)[:, s] = qkv
                if S > 64:      # long sequences: the newest query row alone (O(S) per step)
                    ops.attention_fwd(cache[li], key_mask, N, S, H, scale, causal=True, only_row=s, out=ctx_buf)
                    ao = ctx_buf.view(N, S, d)[:, s].contiguous()
                else:
                    ao = ops.attention_fwd(cache[li], key_mask, N, S, H, scale, causal=True).view(N, S, d)[:, s].contiguous()
                x1 = ops.gemm(ao, rt.w(L.self_attn.out_proj.weight), bias=L.self_attn.out_proj.bias.detach(), res=x)
                ops.bcast_add_fwd_(x1, gz[li], N, 1)
                if g2 is not None:
                    ops.bcast_add_fwd_(x1, g2[li], N, 1)
                xn2, _, _ = ops.layernorm_fwd(x1, L.norm2.weight.detach(), L.norm2.bias.detach())
                h = ops.gemm(xn2, rt.w(L.linear1.weight), bias=L.linear1.bias.detach(), act=ops.RELU)
                x = ops.gemm(h, rt.w(L.linear2.weight), bias=L.linear2.bias.detach(), res=x1)
            xo, _, _ = ops.layernorm_fwd(x, dec.decoder.norm.weight.detach(), dec.decoder.norm.bias.detach())
            cl = ops.gemm(xo, rt.w(fcn.command_fcn.weight), bias=fcn.command_fcn.bias.detach())
            al = ops.gemm(xo, rt.w(fcn.args_fcn.weight), bias=fcn.args_fcn.bias.detach()) \
                .view(N, cfg.n_args, self.args_dim)
            cmd_new, arg_new = self._sample(cl, al, temperature)                                 # (N,), (N, n_args)
            arg_new = arg_new - 1
            arg_new[~cam[cmd_new]] = -1
            commands_y[:, 0, s] = cmd_new
            args_y[:, 0, s] = arg_new
            cmd_s, arg_s = cmd_new.float(), arg_new.float()
            cmd_buf[:, s + 1] = cmd_s
        return commands_y, args_y

    def _make_absolute(self, commands_y, args_y):
        """model.py:461-479, per sequence (the reference flattens the batch, which is only meaningful for one icon):
        relative targets -> absolute coordinates by a running sum of the end positions of the real commands"""
        cfg = self.cfg
        cam = self.cmd_args_mask.to(commands_y.device).bool()
        args_y = args_y.clone()
        mask = cam[commands_y.long()]
        args_y[mask] -= cfg.args_dim - 1
        real = commands_y < EOS_ID                                           # (..., T)
        end = torch.where(real.unsqueeze(-1), args_y[..., 9:11], torch.zeros_like(args_y[..., 9:11]))
        before = end.cumsum(dim=-2) - end                                    # end positions of the earlier real commands
        # columns 5..10 = control1 (x, y), control2 (x, y), end_pos (x, y): each pair shifted by (sum_x, sum_y)
        shift = torch.stack([before[..., 0], before[..., 1]] * 3, dim=-1) * real.unsqueeze(-1).to(before.dtype)
        args_y[..., 5:11] += shift.to(args_y.dtype)
        _, args_y = self._make_valid(commands_y, args_y)
        return args_y

    def _make_valid(self, commands_y, args_y, visibility_y=None, PAD_VAL=-1):
        """restated from /root/reference/deepsvg/model/model.py:450-459 (output-defining host glue of the sampling path: invisible
        groups become [m, EOS, ...] with padded arguments, argument slots a command does not use become PAD_VAL)"""
        if visibility_y is not None:
            S = commands_y.size(-1)
            commands_y[~visibility_y] = commands_y.new_tensor([M_ID, *[EOS_ID] * (S - 1)])
            args_y[~visibility_y] = PAD_VAL
        mask = self.cmd_args_mask[commands_y.long()].bool()
        args_y[~mask] = PAD_VAL
        return commands_y, args_y
