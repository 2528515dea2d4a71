"""Tensor-level wrappers over the C ABI (include/dsvg.h).  PyTorch is used for device memory and streams
only: every op takes/returns torch tensors that live on a HIP device and enqueues hand-written gfx950
kernels on the current stream.  No op has a CPU or aten fallback.
"""
import ctypes as C
import os

import torch

from . import lib as _l

F32, BF16 = _l.DSVG_F32, _l.DSVG_BF16
RELU = 1

# optional per-launch timing of tagged GEMMs (bench.py roofline leg): HIP events on the launch stream
PROFILE_ON = False
PROFILE = []          # (tag, start_event, end_event, flops of the launch, algorithmic bytes of the launch, spec)
# keep the deferred reductions of the train step on while profiling (the profiled sequence = the timed one)
PROFILE_KEEP_DEFER = False
_TAG = None


def _event():
    return torch.cuda.Event(enable_timing=True)


class tag:
    """with ops.tag("ffn"): ...  labels the GEMM launches inside for the profiler"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _TAG
        self.prev, _TAG = _TAG, self.name

    def __exit__(self, *exc):
        global _TAG
        _TAG = self.prev
        return False


class tag_default(tag):
    """like tag, but keeps an enclosing tag (the weight-gradient GEMMs inside the FFN sub-block stay "ffn")"""

    def __enter__(self):
        global _TAG
        self.prev = _TAG
        if _TAG is None:
            _TAG = self.name


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise _l.DsvgError(f"unsupported dtype {t.dtype}")


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _l.DsvgError("deepsvg_amd ops need HIP device tensors (no CPU fallback)")


def require_device(device):
    if torch.device(device).type != "cuda":
        raise _l.DsvgError("deepsvg_amd runs on a HIP device only (no CPU fallback): move the model and its "
                           "inputs to 'cuda'")
    _l.load()


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ws(nbytes, device):
    t = torch.empty(max(int(nbytes), 4) // 4 + 1, dtype=torch.float32, device=device)
    st = _DEFER.state.get(_stream_key())
    if st is not None and st.depth:
        st.keep.append(t)           # a queued reduction reads it at flush_deferred()
        if PROFILE_ON:
            st.bytes_by_tag[_TAG] = st.bytes_by_tag.get(_TAG, 0) + t.numel() * 4
    return t


# ------------------------------------------------------------------------------------------------
# deferred parameter-gradient reductions (include/dsvg.h: dsvg_defer_scope / dsvg_flush_deferred)
# ------------------------------------------------------------------------------------------------
def _stream_key():
    """identity of the stream the ops launch on (0 on the CPU emulation path of the test-suite)"""
    return _stream() if torch.cuda.is_available() else 0


class _DeferState:
    __slots__ = ("depth", "keep", "post", "bytes_by_tag", "finish")

    def __init__(self):
        self.depth, self.keep, self.post, self.bytes_by_tag, self.finish = 0, [], [], {}, []


class _DeferScope:
    """`with ops.DEFER:` - partial-sum reductions launched inside ON THE CURRENT STREAM (split-K slices, LayerNorm
    gamma/beta partials, bias column sums, embedding-table gradients) are queued; their outputs are valid after
    flush_deferred() on that stream.  Only a caller that owns the whole backward pass may open it (TrainStep): nothing
    may read those outputs in between.  All state is per stream - the library's queue (csrc/gemm.hip) and the
    workspaces / follow-up work kept here - so two models or trainers on different streams or devices do not interact."""
    state = {}          # stream -> _DeferState

    def __enter__(self):
        key = _stream_key()
        st = _DeferScope.state.get(key)
        if st is None:
            st = _DeferScope.state[key] = _DeferState()
        if st.depth == 0 and torch.cuda.is_available():
            _l.load().dsvg_defer_scope(1, key)
        st.depth += 1
        self._keys = getattr(self, "_keys", [])
        self._keys.append(key)

    def __exit__(self, *exc):
        key = self._keys.pop()
        st = _DeferScope.state[key]
        st.depth -= 1
        if st.depth == 0 and torch.cuda.is_available():
            _l.load().dsvg_defer_scope(0, key)
        return False


_DEFER = _DeferScope
DEFER = _DeferScope()


def defer_active():
    """True while a deferral scope is open on the current stream"""
    st = _DEFER.state.get(_stream_key())
    return st is not None and st.depth > 0


def zero_(t):
    """t.zero_() for a contiguous fp32 tensor; inside an open deferral scope on this stream the fill joins the queued reductions
    (dsvg_defer_zero: no launch of its own, valid after flush_deferred() like their outputs)"""
    if t.numel() == 0:
        return t
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and defer_active():
        _l.check(_l.load().dsvg_defer_zero(t.data_ptr(), t.numel(), _stream()), "dsvg_defer_zero")
        return t
    return t.zero_()


def defer_post(fn):
    """run fn() right after the queued reductions at the next flush_deferred() on this stream (work that reads their
    outputs); without an open scope on this stream nothing is queued, so fn runs now"""
    st = _DEFER.state.get(_stream_key())
    if st is None:
        fn()
    else:
        st.post.append((fn, _TAG))


def flush_deferred():
    """perform every reduction queued on the current stream (one launch per 64), then the work registered with defer_post"""
    key = _stream_key()
    st = _DEFER.state.get(key)
    if st is None or not (st.keep or st.post or st.finish):
        if st is not None and st.depth == 0:
            _DEFER.state.pop(key, None)
        return
    if torch.cuda.is_available():
        ev = None
        if PROFILE_ON and st.keep:
            ev = _event()
            ev.record()
        _l.check(_l.load().dsvg_flush_deferred(key), "dsvg_flush_deferred")
        if ev is not None:
            # one record for the batched reduction launches; `by_tag` = workspace bytes each tag queued (the share of the
            # launch time a sub-block is charged with)
            ev1 = _event()
            ev1.record()
            PROFILE.append(("reduce", ev, ev1, 0.0, float(sum(st.bytes_by_tag.values())),
                            dict(op="reduce_deferred", by_tag=dict(st.bytes_by_tag))))
    st.bytes_by_tag = {}
    if st.finish:           # the fused-FFN layers' gradient finishes queued behind the reductions: one launch for all of them
        fin, st.finish = st.finish, []
        with tag("ffn"):
            ffn_wgrad_finish_many(fin)
    post, st.post = st.post, []
    for fn, t in post:
        if t is None:
            fn()
        else:
            with tag(t):
                fn()
    st.keep.clear()
    if st.depth == 0:
        _DEFER.state.pop(key, None)


class _GroupScope:
    """`with ops.GROUP:` - the split-K weight-gradient GEMMs launched inside on the current stream run as ONE launch when
    the block is left (include/dsvg.h: dsvg_gemm_group_scope).  Their operands must stay alive until then - they do when
    they are locals of the code inside the block."""
    active = None       # profiling only: {tag: [flops, bytes]} of the members queued in the open scope
    depth = {}          # stream key -> nesting depth: only the outermost block opens / closes the library's scope (a block
                        # inside a stack-level scope - functional.STACK_GROUP - must not launch the stack's queue early)

    def __enter__(self):
        self._key = key = _stream_key()
        d = _GroupScope.depth.get(key, 0)
        _GroupScope.depth[key] = d + 1
        if d:
            return
        if torch.cuda.is_available():
            _l.check(_l.load().dsvg_gemm_group_scope(1, key), "dsvg_gemm_group_scope")
        if PROFILE_ON:
            _GroupScope.active = {}

    def __exit__(self, *exc):
        key = _stream_key()
        d = _GroupScope.depth.get(key, 1) - 1
        if d > 0:
            _GroupScope.depth[key] = d
            return False
        _GroupScope.depth.pop(key, None)
        self._key = key
        members, _GroupScope.active = _GroupScope.active, None
        ev = None
        if PROFILE_ON and members:
            ev = _event()
            ev.record()
        if torch.cuda.is_available():
            _l.check(_l.load().dsvg_gemm_group_scope(0, self._key), "dsvg_gemm_group_scope")
        if ev is not None:
            # one record for the grouped launch; `by_tag` = FLOPs each tag queued (the share of the launch a sub-block is
            # charged with), the members' own records carry no time
            ev1 = _event()
            ev1.record()
            PROFILE.append(("group", ev, ev1, float(sum(v[0] for v in members.values())),
                            float(sum(v[1] for v in members.values())),
                            dict(op="wgrad_group", by_tag={k: v[0] for k, v in members.items()},
                                 bytes_by_tag={k: v[1] for k, v in members.items()})))
        return False


GROUP = _GroupScope()


def _rowmajor(t):
    assert t.dim() == 2 and t.stride(1) == 1, f"need a row-major 2-D tensor, got strides {t.stride()}"
    return t


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
_GEMM_CACHE = {}        # call signature (everything but the pointers) -> (validated descriptor, M, N, out dtype, ws bytes)


def _gemm_build(a, b, a_kc, b_kc, bias, res, res_pre, act, gate, gate_scale, drop_p, drop_site, a_drop_p, a_drop_site,
                seed, out, out_dtype, accumulate, split_k, impl, rowsum):
    """full validation + a descriptor with every non-pointer field filled (cached per call signature by gemm)"""
    _rowmajor(a), _rowmajor(b)
    assert a.dtype == b.dtype
    M, K = (a.shape if a_kc else (a.shape[1], a.shape[0]))
    N, Kb = (b.shape if b_kc else (b.shape[1], b.shape[0]))
    assert K == Kb, f"gemm: inner dims differ ({K} vs {Kb})"
    odt = out.dtype if out is not None else (out_dtype or a.dtype)
    assert odt in (a.dtype, torch.float32)
    d = _l.GemmDesc()
    d.dtype = _dt(a)
    d.M, d.N, d.K = M, N, K
    d.lda, d.a_kc = a.stride(0), int(a_kc)
    d.ldb, d.b_kc = b.stride(0), int(b_kc)
    d.c_f32 = int(odt == torch.float32)
    if out is not None:
        _rowmajor(out)
        assert tuple(out.shape) == (M, N)
        d.ldc = out.stride(0)
    else:
        d.ldc = N
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    if res is not None:
        _rowmajor(res)
        assert res.dtype == a.dtype and tuple(res.shape) == (M, N)
        d.ldres, d.res_pre = res.stride(0), int(res_pre)
    d.act = act
    if gate is not None:
        _rowmajor(gate)
        assert gate.dtype == a.dtype and tuple(gate.shape) == (M, N)
        d.ldgate, d.gate_scale = gate.stride(0), float(gate_scale)
    d.drop_p, d.drop_site = float(drop_p), int(drop_site)
    d.a_drop_p, d.a_drop_site, d.a_drop_ld = float(a_drop_p), int(a_drop_site), a.shape[1]
    if drop_p > 0 or a_drop_p > 0:
        assert seed is not None and seed.dtype == torch.int64
    d.accumulate = int(accumulate)
    d.impl = impl
    ws_bytes = 0
    if split_k > 1:
        assert odt == torch.float32 and (out is None or out.is_contiguous())
        ws_bytes = _l.load().dsvg_gemm_workspace_bytes(M, N, split_k)
        d.split_k = split_k
        if rowsum is not None:
            assert rowsum.dtype == torch.float32 and rowsum.is_contiguous() and rowsum.numel() == M
    else:
        assert rowsum is None, "rowsum needs split_k > 1"
    return d, M, N, odt, ws_bytes


def gemm(a, b, *, a_kc=True, b_kc=True, bias=None, res=None, res_pre=False, act=0, gate=None, gate_scale=1.0,
         drop_p=0.0, drop_site=0, a_drop_p=0.0, a_drop_site=0, seed=None, out=None, out_dtype=None,
         accumulate=False, split_k=1, impl=0, rowsum=None):
    """C[M,N] = epi(sum_k A(m,k) B(n,k)).  a: [M,K] if a_kc else [K,M]; b: [N,K] if b_kc else [K,N].
    rowsum (fp32 [M], only with split_k > 1): also rowsum[m] = sum_k A(m,k) (bias gradient of a weight-grad GEMM).
    The descriptor of a call signature (shapes, strides, options - everything but the pointers) is validated and
    built once and re-used: the Python dispatch of the ~250 GEMMs of a train step is otherwise as long as the step."""
    _chk(a, b, bias, res, gate, seed, out)
    key = (a.dtype, a.shape, a.stride(0), b.shape, b.stride(0), a_kc, b_kc, bias is not None,
           None if res is None else (res.shape, res.stride(0)), res_pre, act,
           None if gate is None else (gate.shape, gate.stride(0)), gate_scale, drop_p, drop_site, a_drop_p,
           a_drop_site, out_dtype if out is None else (out.dtype, out.shape, out.stride(0), out.stride(1)),
           accumulate, split_k, impl, rowsum is not None)
    ent = _GEMM_CACHE.get(key)
    if ent is None:
        ent = _gemm_build(a, b, a_kc, b_kc, bias, res, res_pre, act, gate, gate_scale, drop_p, drop_site, a_drop_p,
                          a_drop_site, seed, out, out_dtype, accumulate, split_k, impl, rowsum)
        if len(_GEMM_CACHE) < 4096:
            _GEMM_CACHE[key] = ent
    d, M, N, odt, ws_bytes = ent
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=a.device)
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.gate = gate.data_ptr() if gate is not None else None
    d.seed = seed.data_ptr() if (seed is not None and (drop_p > 0 or a_drop_p > 0)) else None
    d.rowsum = rowsum.data_ptr() if rowsum is not None else None
    ws = None
    if ws_bytes:
        ws = _ws(ws_bytes, a.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L = _l.load()
    if PROFILE_ON and _TAG is not None:
        esz = 2 if a.dtype == torch.bfloat16 else 4
        extra = (esz * d.M * d.N if (res is not None or gate is not None) else 0)    # residual / gate operand
        if _GroupScope.active is not None and split_k > 1 and not a_kc and not b_kc:
            # queued in an open group: it runs with the group's one launch (timed at the end of the scope)
            _l.check(L.dsvg_gemm(C.byref(d), _stream()), "dsvg_gemm")
            m = _GroupScope.active.setdefault(_TAG, [0.0, 0.0])
            m[0] += 2.0 * d.M * d.N * d.K
            m[1] += float(esz * (d.M * d.K + d.N * d.K) + out.element_size() * d.M * d.N)
            return out
        ev0, ev1 = _event(), _event()
        ev0.record()
        _l.check(L.dsvg_gemm(C.byref(d), _stream()), "dsvg_gemm")
        ev1.record()
        spec = dict(a=tuple(a.shape), b=tuple(b.shape), a_kc=a_kc, b_kc=b_kc, bias=bias is not None, res=res is not None,
                    act=act, gate=gate is not None, gate_scale=gate_scale, drop_p=drop_p, split_k=split_k,
                    rowsum=rowsum is not None, out_f32=out.dtype == torch.float32, dtype=str(a.dtype))
        PROFILE.append((_TAG, ev0, ev1, 2.0 * d.M * d.N * d.K,
                        float(esz * (d.M * d.K + d.N * d.K) + out.element_size() * d.M * d.N + extra), spec))
        return out
    _l.check(L.dsvg_gemm(C.byref(d), _stream()), "dsvg_gemm")
    return out


# target workgroup count of a split-K weight-gradient GEMM.  256 = one per CU: such launches take the 4-stage LDS-DMA variant
# (three K steps in flight per workgroup), which keeps the HBM rate of 512 single-stage workgroups with half the slices to
# write and reduce (step 8.39 -> 8.32 ms; before the 4-stage variant existed 512 was the better setting)
_SPLITK_BLOCKS = int(os.environ.get("DSVG_SPLITK_BLOCKS", "256"))
_SPLITK_REFILL = int(os.environ.get("DSVG_SPLITK_REFILL", "480"))     # workgroups for products whose tiles under-fill the chip (0: off)


def split_k_for(M, N, K, target_blocks=None):
    """split factor for the weight-gradient GEMMs (small M x N output, K = #tokens): a multiple of 8 so that the
    K slices are grouped per XCD (see gemm_bf16.hip), about `target_blocks` workgroups in total."""
    default_target = target_blocks is None
    target_blocks = target_blocks or _SPLITK_BLOCKS
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    s = max(1, target_blocks // tiles)
    s = min(s, max(1, K // 128))
    if s >= 8:
        s = s // 8 * 8
        # 12 tiles (the 768 x 256 in_proj product): 16 slices = 192 workgroups leave a quarter of the CUs without one on the
        # one-workgroup-per-CU variant, 24 = 288 do not fit it.  ~480 workgroups on the high-occupancy variant are 20 % faster
        # (profiles/r04_wgrad_split_probe.log: 54.5 -> 43.8 us at 63 k rows, 40.0 -> 35.7 at 41 k)
        if default_target and _SPLITK_REFILL > 0 and K >= 16384 and tiles * s < 0.8 * target_blocks:
            s2 = min((_SPLITK_REFILL // tiles) // 8 * 8, max(1, K // 128) // 8 * 8)
            if s2 > s:
                s = s2
    return s


def colsum(a, *, out=None, accumulate=False, drop_p=0.0, drop_site=0, seed=None):
    """out[n] (+)= sum_m drop(a[m,n])  (fp32)"""
    _chk(a, out, seed)
    _rowmajor(a)
    M, N = a.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=a.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == N
    L = _l.load()
    ws = _ws(L.dsvg_colsum_workspace_bytes(M, N), a.device)
    _l.check(L.dsvg_colsum(_dt(a), a.data_ptr(), a.stride(0), M, N, out.data_ptr(), int(accumulate), float(drop_p),
                           int(drop_site), _p(seed) if drop_p > 0 else None, ws.data_ptr(), ws.numel() * 4,
                           _stream()), "dsvg_colsum")
    return out


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps=1e-5):
    _chk(x, gamma, beta)
    assert x.is_contiguous() and x.dim() == 2
    rows, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    ev = _prof_begin()
    _l.check(_l.load().dsvg_layernorm_fwd(_dt(x), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                          mean.data_ptr(), rstd.data_ptr(), rows, d, float(eps), _stream()),
             "dsvg_layernorm_fwd")
    _prof_end(ev, 0.0, 0.0, dict(op="layernorm_fwd"))
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, *, res=None, dgamma=None, dbeta=None, accumulate=False, dx=None, masked=None):
    """returns dx = [res +] LN'(dy), dgamma, dbeta (fp32).  masked = (p, site, seed): -> (dx, dgamma, dbeta, dxm) with
    dxm = drop_apply(dx, p, site, seed) from the same launch."""
    _chk(dy, x, mean, rstd, gamma, res, dgamma, dbeta)
    assert dy.is_contiguous() and x.is_contiguous() and dy.shape == x.shape
    rows, d = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = torch.empty(d, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(d, dtype=torch.float32, device=x.device)
    if res is not None:
        assert res.is_contiguous() and res.shape == x.shape and res.dtype == x.dtype
    L = _l.load()
    ws = _ws(L.dsvg_layernorm_bwd_workspace_bytes(rows, d), x.device)
    ev = _prof_begin()
    if masked is not None:
        mp, msite, mseed = masked
        dxm = torch.empty_like(dx)
        _l.check(L.dsvg_layernorm_bwd_masked(_dt(x), dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             gamma.data_ptr(), _p(res), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                             int(accumulate), rows, d, ws.data_ptr(), ws.numel() * 4, dxm.data_ptr(),
                                             float(mp), int(msite), _p(mseed) if mp > 0 else None, _stream()),
                 "dsvg_layernorm_bwd_masked")
        _prof_end(ev, 0.0, 0.0, dict(op="layernorm_bwd"))
        return dx, dgamma, dbeta, dxm
    _l.check(L.dsvg_layernorm_bwd(_dt(x), dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                  gamma.data_ptr(), _p(res), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                  int(accumulate), rows, d, ws.data_ptr(), ws.numel() * 4, _stream()),
             "dsvg_layernorm_bwd")
    _prof_end(ev, 0.0, 0.0, dict(op="layernorm_bwd"))
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def attention_tiles(seq_off, n_seq, max_rows=32):
    """groups of consecutive packed sequences with <= max_rows rows in total (include/dsvg.h): int32 [n_seq + 2]"""
    _chk(seq_off)
    assert seq_off.dtype == torch.int32 and seq_off.numel() == n_seq + 1
    tiles = torch.empty(n_seq + 2, dtype=torch.int32, device=seq_off.device)
    scratch = torch.empty((n_seq + 63) // 64 * 64, dtype=torch.int32, device=seq_off.device)
    _l.check(_l.load().dsvg_attention_tiles(seq_off.data_ptr(), n_seq, max_rows, tiles.data_ptr(), scratch.data_ptr(),
                                            _stream()), "dsvg_attention_tiles")
    return tiles


def attention_fwd(qkv, key_mask, n_seq, S, n_heads, scale, drop_p=0.0, drop_site=0, seed=None, seq_off=None,
                  tiles=None, causal=False, only_row=None, out=None):
    """seq_off (int32 [n_seq+1], device): packed layout, sequence b = rows seq_off[b]..seq_off[b+1]-1 (<= S rows, all
    keys visible, key_mask must be None); rows past seq_off[n_seq] are zero-filled.
    causal: query i attends keys j <= i (autoregressive decoder; dense layout only)."""
    _chk(qkv, key_mask, seed, seq_off)
    rows = qkv.shape[0]
    assert qkv.is_contiguous() and qkv.shape[1] == 3 * 32 * n_heads, "attention needs head_dim == 32"
    assert (rows >= n_seq * S) if seq_off is None else (key_mask is None and seq_off.numel() == n_seq + 1)
    if out is None:
        out = torch.empty((rows, 32 * n_heads), dtype=qkv.dtype, device=qkv.device)
    if S > 64:      # long sequences: key_mask holds valid-prefix lengths (build_masks)
        # only_row (with causal): the incremental decoding step - that query row alone is computed into `out`
        assert seq_off is None and rows == n_seq * S and (key_mask is None or key_mask.dtype == torch.int32)
        _l.check(_l.load().dsvg_attention_long_fwd(_dt(qkv), qkv.data_ptr(), _p(key_mask), out.data_ptr(), n_seq, S,
                                                   n_heads, float(scale), 1 if causal else 0,
                                                   -1 if only_row is None else int(only_row), float(drop_p),
                                                   int(drop_site), _p(seed) if drop_p > 0 else None, _stream()),
                 "dsvg_attention_long_fwd")
        return out
    if causal:
        assert seq_off is None and rows == n_seq * S
        _l.check(_l.load().dsvg_attention_causal_fwd(_dt(qkv), qkv.data_ptr(), _p(key_mask), out.data_ptr(), n_seq, S,
                                                     n_heads, float(scale), float(drop_p), int(drop_site),
                                                     _p(seed) if drop_p > 0 else None, _stream()),
                 "dsvg_attention_causal_fwd")
        return out
    _l.check(_l.load().dsvg_attention_fwd(_dt(qkv), qkv.data_ptr(), _p(key_mask), _p(seq_off), rows, _p(tiles),
                                          out.data_ptr(), n_seq, S, n_heads, float(scale), float(drop_p), int(drop_site),
                                          _p(seed) if drop_p > 0 else None, _stream()), "dsvg_attention_fwd")
    return out


def attention_bwd(qkv, key_mask, dout, n_seq, S, n_heads, scale, drop_p=0.0, drop_site=0, seed=None, seq_off=None,
                  tiles=None, causal=False):
    _chk(qkv, key_mask, dout, seed, seq_off)
    assert qkv.is_contiguous() and dout.is_contiguous() and dout.dtype == qkv.dtype
    assert seq_off is None or (key_mask is None and seq_off.numel() == n_seq + 1)
    dqkv = torch.empty_like(qkv)
    if S > 64:
        assert seq_off is None and qkv.shape[0] == n_seq * S and (key_mask is None or key_mask.dtype == torch.int32)
        _l.check(_l.load().dsvg_attention_long_bwd(_dt(qkv), qkv.data_ptr(), _p(key_mask), dout.data_ptr(),
                                                   dqkv.data_ptr(), n_seq, S, n_heads, float(scale), 1 if causal else 0,
                                                   float(drop_p), int(drop_site), _p(seed) if drop_p > 0 else None,
                                                   _stream()), "dsvg_attention_long_bwd")
        return dqkv
    if causal:
        assert seq_off is None and qkv.shape[0] == n_seq * S
        _l.check(_l.load().dsvg_attention_causal_bwd(_dt(qkv), qkv.data_ptr(), _p(key_mask), dout.data_ptr(),
                                                     dqkv.data_ptr(), n_seq, S, n_heads, float(scale), float(drop_p),
                                                     int(drop_site), _p(seed) if drop_p > 0 else None, _stream()),
                 "dsvg_attention_causal_bwd")
        return dqkv
    _l.check(_l.load().dsvg_attention_bwd(_dt(qkv), qkv.data_ptr(), _p(key_mask), _p(seq_off), qkv.shape[0],
                                          _p(tiles), dout.data_ptr(), dqkv.data_ptr(), n_seq, S, n_heads, float(scale),
                                          float(drop_p), int(drop_site), _p(seed) if drop_p > 0 else None, _stream()),
             "dsvg_attention_bwd")
    return dqkv


ATTN_BWD_LAYER_ELEMS = 512 * 512    # bf16 elements of one layer's backward image (attn_pack_bwd): out_proj^T | in_proj^T fragments


def attn_pack_bwd(flat, offs, n_layers, packed=None):
    """fp32 flat parameters + the [n_layers, 2] offset table of attn_pack -> per layer the A fragments of Wo^T per head, for
    attention_bwd_outproj"""
    _chk(flat, offs, packed)
    assert flat.dtype == torch.float32 and offs.dtype == torch.int64 and offs.is_contiguous() and tuple(offs.shape) == (n_layers, 2)
    if packed is None:
        packed = torch.empty(n_layers * ATTN_BWD_LAYER_ELEMS, dtype=torch.bfloat16, device=flat.device)
    assert packed.numel() == n_layers * ATTN_BWD_LAYER_ELEMS and packed.dtype == torch.bfloat16
    _l.check(_l.load().dsvg_attn_pack_bwd(flat.data_ptr(), offs.data_ptr(), n_layers, packed.data_ptr(), _stream()),
             "dsvg_attn_pack_bwd")
    return packed


def attention_bwd_outproj(qkv, key_mask, dx1m, wo_packed_bwd, n_seq, S, scale, drop_p=0.0, drop_site=0, seed=None, seq_off=None,
                          tiles=None):
    """attention_bwd of 8 heads of 32 with the out_proj backward inside: dx1m [rows, 256] is the gradient of the projected
    output (residual dropout mask applied), the head-output gradient dx1m @ Wo never leaves the chip -> dqkv"""
    _chk(qkv, key_mask, dx1m, wo_packed_bwd, seed, seq_off, tiles)
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.shape[1] == 768
    assert dx1m.dtype == qkv.dtype and dx1m.is_contiguous() and tuple(dx1m.shape) == (qkv.shape[0], 256)
    assert wo_packed_bwd.numel() == ATTN_BWD_LAYER_ELEMS and wo_packed_bwd.is_contiguous()
    assert (seq_off is None) == (tiles is None) and (seq_off is None or (key_mask is None and seq_off.numel() == n_seq + 1))
    dqkv = torch.empty_like(qkv)
    _l.check(_l.load().dsvg_attention_bwd_outproj(qkv.data_ptr(), _p(key_mask), _p(seq_off), qkv.shape[0], _p(tiles),
                                                  dx1m.data_ptr(), wo_packed_bwd.data_ptr(), dqkv.data_ptr(), n_seq, S,
                                                  float(scale), float(drop_p), int(drop_site),
                                                  _p(seed) if drop_p > 0 else None, _stream()),
             "dsvg_attention_bwd_outproj")
    return dqkv


# ------------------------------------------------------------------------------------------------
# masks / embedding / positional / pooling
# ------------------------------------------------------------------------------------------------
def attn_bwd_dx(dqkv, x, mean, rstd, gamma, res, packed_bwd_layer, *, dx=None, dgamma=None, dbeta=None, accumulate=False,
                masked=None):
    """dx = res + LayerNorm'(dqkv . in_proj_weight), dgamma, dbeta in ONE launch (csrc/attn_bwd_dx.hip): what
    gemm(dqkv, W_in, b_kc=False) + layernorm_bwd(..., res=res) compute, without the [rows, 256] intermediate.
    masked = (p, site, seed): -> (dx, dgamma, dbeta, dxm) with dxm = drop_apply(dx, p, site, seed) from the same launch."""
    _chk(dqkv, x, mean, rstd, gamma, res, packed_bwd_layer, dx, dgamma, dbeta)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and x.is_contiguous()
    rows = x.shape[0]
    assert dqkv.dtype == x.dtype and tuple(dqkv.shape) == (rows, 768) and dqkv.is_contiguous()
    assert res.dtype == x.dtype and res.shape == x.shape and res.is_contiguous()
    assert mean.dtype == torch.float32 and rstd.dtype == torch.float32 and mean.numel() >= rows and rstd.numel() >= rows
    assert gamma.dtype == torch.float32 and gamma.numel() == 256
    assert packed_bwd_layer.numel() == ATTN_BWD_LAYER_ELEMS and packed_bwd_layer.is_contiguous()
    if dx is None:
        dx = torch.empty_like(x)
    assert dx.dtype == x.dtype and dx.shape == x.shape and dx.is_contiguous()
    if dgamma is None:
        dgamma = torch.empty(256, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(256, dtype=torch.float32, device=x.device)
    dxm, mp, msite, mseed = None, 0.0, 0, None
    if masked is not None:
        mp, msite, mseed = float(masked[0]), int(masked[1]), masked[2]
        _chk(mseed)
        dxm = torch.empty_like(dx)
    L = _l.load()
    ws = _ws(L.dsvg_attn_bwd_dx_workspace_bytes(rows), x.device)
    ev = _prof_begin()
    _l.check(L.dsvg_attn_bwd_dx(dqkv.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                res.data_ptr(), packed_bwd_layer.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                int(accumulate), rows, ws.data_ptr(), ws.numel() * 4, _p(dxm), mp, msite,
                                _p(mseed) if mp > 0 else None, _stream()), "dsvg_attn_bwd_dx")
    _prof_end(ev, 2.0 * 768 * 256 * rows, (1536.0 + 3 * 512 + (512 if dxm is not None else 0)) * rows,
              dict(op="attn_bwd_dx", rows=rows))
    if masked is not None:
        return dx, dgamma, dbeta, dxm
    return dx, dgamma, dbeta


def seq_lens(commands, S, eos_id=4):
    """commands float32 [n_seq, S] -> int32 [n_seq]: index of the first EOS (S if none) = number of valid keys"""
    _chk(commands)
    assert commands.dtype == torch.float32 and commands.is_contiguous()
    n_seq = commands.numel() // S
    lens = torch.empty(n_seq, dtype=torch.int32, device=commands.device)
    _l.check(_l.load().dsvg_seq_lens(commands.data_ptr(), n_seq, S, eos_id, lens.data_ptr(), _stream()), "dsvg_seq_lens")
    return lens


def build_masks(commands, S, G=0, eos_id=4, want_group_mask=False):
    """commands: float32 [n_seq, S] -> key_mask int64[n_seq], seq_visible int32[n_seq], group_mask int64[n_seq/G].
    S > 64 (one-stage / autoregressive sequences): the key mask is returned as valid-prefix LENGTHS (int32 [n_seq]),
    which attention_fwd/bwd and masked_mean_fwd/bwd accept in its place; no group masks at that length."""
    _chk(commands)
    assert commands.dtype == torch.float32 and commands.is_contiguous()
    if S > 64:
        assert not want_group_mask
        return seq_lens(commands, S, eos_id), None, None
    n_seq = commands.numel() // S
    dev = commands.device
    key_mask = torch.empty(n_seq, dtype=torch.int64, device=dev)
    seq_visible = torch.empty(n_seq, dtype=torch.int32, device=dev)
    group_mask = torch.empty(n_seq // G, dtype=torch.int64, device=dev) if want_group_mask else None
    _l.check(_l.load().dsvg_build_masks(commands.data_ptr(), n_seq, S, G if want_group_mask else 1, eos_id,
                                        key_mask.data_ptr(), seq_visible.data_ptr(), _p(group_mask), _stream()),
             "dsvg_build_masks")
    return key_mask, seq_visible, group_mask


def group_index(commands, S, m_id=0):
    _chk(commands)
    assert commands.dtype == torch.float32 and commands.is_contiguous()
    n_seq = commands.numel() // S
    groups = torch.empty(n_seq * S, dtype=torch.int32, device=commands.device)
    _l.check(_l.load().dsvg_group_index(commands.data_ptr(), n_seq, S, m_id, groups.data_ptr(), _stream()),
             "dsvg_group_index")
    return groups


def embed_gather(commands, args, command_embed, arg_embed, dtype, group_embed=None, groups=None):
    """commands float32 [T], args float32 [T, n_args] -> A [T, n_args*E], R [T, d] in `dtype`."""
    _chk(commands, args, command_embed, arg_embed, group_embed, groups)
    T = commands.numel()
    n_args = args.numel() // T
    n_argvals, E = arg_embed.shape
    n_cmd, d = command_embed.shape
    A = torch.empty((T, n_args * E), dtype=dtype, device=commands.device)
    R = torch.empty((T, d), dtype=dtype, device=commands.device)
    _l.check(_l.load().dsvg_embed_gather(_dt(A), commands.data_ptr(), args.data_ptr(), command_embed.data_ptr(),
                                         arg_embed.data_ptr(), _p(group_embed), _p(groups), A.data_ptr(),
                                         R.data_ptr(), T, n_args, E, d, n_cmd, n_argvals, _stream()),
             "dsvg_embed_gather")
    return A, R


def embed_scatter(commands, args, dA, dR, d_arg_embed, d_command_embed, groups=None, d_group_embed=None):
    """overwrites the fp32 gradient tables d_arg_embed [n_argvals,E], d_command_embed [n_cmd,d] (, d_group_embed)."""
    _chk(commands, args, dA, dR, d_arg_embed, d_command_embed, groups, d_group_embed)
    T = commands.numel()
    n_args = args.numel() // T
    n_argvals, E = d_arg_embed.shape
    n_cmd, d = d_command_embed.shape
    n_groups = d_group_embed.shape[0] if d_group_embed is not None else 0
    assert dA.is_contiguous() and dR.is_contiguous()
    L = _l.load()
    ws = _ws(L.dsvg_embed_scatter_workspace_bytes(T, n_args, E, d, n_cmd, n_argvals, n_groups), dA.device)
    _l.check(L.dsvg_embed_scatter(_dt(dA), commands.data_ptr(), args.data_ptr(), _p(groups), dA.data_ptr(),
                                  dR.data_ptr(), d_arg_embed.data_ptr(), d_command_embed.data_ptr(),
                                  _p(d_group_embed), T, n_args, E, d, n_cmd, n_argvals, n_groups, ws.data_ptr(),
                                  ws.numel() * 4, _stream()), "dsvg_embed_scatter")


def add_pos_fwd(x, pos, n_seq, S, dtype, drop_p=0.0, drop_site=0, seed=None):
    """y[t] = drop((x[t] if x is not None else 0) + pos[t % S]); pos fp32 [>=S, d]"""
    _chk(x, pos, seed)
    d = pos.shape[1]
    assert pos.shape[0] >= S and pos.is_contiguous()
    y = torch.empty((n_seq * S, d), dtype=dtype, device=pos.device)
    if x is not None:
        assert x.is_contiguous() and x.dtype == dtype and tuple(x.shape) == (n_seq * S, d)
    _l.check(_l.load().dsvg_add_pos_fwd(_dt(y), _p(x), pos.data_ptr(), y.data_ptr(), n_seq, S, d, float(drop_p),
                                        int(drop_site), _p(seed) if drop_p > 0 else None, _stream()),
             "dsvg_add_pos_fwd")
    return y


def add_pos_bwd(dy, n_seq, S, d_pos, *, want_dx=True, accumulate=False, drop_p=0.0, drop_site=0, seed=None):
    """d_pos (fp32 [S, d] contiguous view) (+)= sum_b (dy*mask)[b*S+s]; returns dx = dy*mask (or None)."""
    _chk(dy, d_pos, seed)
    assert dy.is_contiguous()
    d = dy.shape[1]
    assert d_pos.is_contiguous() and d_pos.numel() == S * d
    dx = torch.empty_like(dy) if want_dx else None
    L = _l.load()
    ws = _ws(L.dsvg_add_pos_bwd_workspace_bytes(n_seq, S, d), dy.device)
    _l.check(L.dsvg_add_pos_bwd(_dt(dy), dy.data_ptr(), _p(dx), d_pos.data_ptr(), int(accumulate), n_seq, S, d,
                                float(drop_p), int(drop_site), _p(seed) if drop_p > 0 else None, ws.data_ptr(),
                                ws.numel() * 4, _stream()), "dsvg_add_pos_bwd")
    return dx


def masked_mean_fwd(x, mask, n_seq, S, seq_off=None):
    _chk(x, mask, seq_off)
    assert x.is_contiguous() and (mask is not None or seq_off is not None)
    d = x.shape[1]
    out = torch.empty((n_seq, d), dtype=x.dtype, device=x.device)
    if S > 64:      # long sequences: `mask` holds valid-prefix lengths (build_masks)
        assert seq_off is None and mask.dtype == torch.int32
        _l.check(_l.load().dsvg_prefix_mean_fwd(_dt(x), x.data_ptr(), mask.data_ptr(), out.data_ptr(), n_seq, S, d,
                                                _stream()), "dsvg_prefix_mean_fwd")
        return out
    _l.check(_l.load().dsvg_masked_mean_fwd(_dt(x), x.data_ptr(), _p(mask), _p(seq_off), out.data_ptr(), n_seq, S, d,
                                            _stream()), "dsvg_masked_mean_fwd")
    return out


def masked_mean_bwd(dout, mask, n_seq, S, seq_off=None, total_rows=None):
    _chk(dout, mask, seq_off)
    assert dout.is_contiguous() and (mask is not None or seq_off is not None)
    d = dout.shape[1]
    rows = n_seq * S if seq_off is None else int(total_rows)
    dx = torch.empty((rows, d), dtype=dout.dtype, device=dout.device)
    if S > 64:
        assert seq_off is None and mask.dtype == torch.int32
        _l.check(_l.load().dsvg_prefix_mean_bwd(_dt(dout), dout.data_ptr(), mask.data_ptr(), dx.data_ptr(), n_seq, S, d,
                                                _stream()), "dsvg_prefix_mean_bwd")
        return dx
    _l.check(_l.load().dsvg_masked_mean_bwd(_dt(dout), dout.data_ptr(), _p(mask), _p(seq_off), rows, dx.data_ptr(),
                                            n_seq, S, d, _stream()), "dsvg_masked_mean_bwd")
    return dx


def visible_first(visible):
    """visible int32 [n] -> (new_of_old int32 [n], old_of_new int32 [n], n_visible int32 [1]); include/dsvg.h"""
    _chk(visible)
    assert visible.dtype == torch.int32 and visible.is_contiguous()
    n = visible.numel()
    new_of_old = torch.empty(n, dtype=torch.int32, device=visible.device)
    old_of_new = torch.empty(n, dtype=torch.int32, device=visible.device)
    nvis = torch.empty(1, dtype=torch.int32, device=visible.device)
    _l.check(_l.load().dsvg_visible_first(visible.data_ptr(), n, new_of_old.data_ptr(), old_of_new.data_ptr(),
                                          nvis.data_ptr(), _stream()), "dsvg_visible_first")
    return new_of_old, old_of_new, nvis


def gather_groups(src, idx, n_groups, S, out=None, n_src=None):
    """out[g*S + s] = src[idx[g]*S + s] for g < n_groups (rows past n_groups*S of `out` are left untouched); n_src: `src`
    holds that many groups and an index beyond them gives a zero sequence (default: every index is taken to be valid)"""
    _chk(src, idx, out)
    assert src.is_contiguous() and idx.dtype == torch.int32 and src.dim() == 2
    if n_src is None:
        n_src = 1 << 40
    else:
        assert 0 < n_src * S <= src.shape[0]
    if out is None:
        out = torch.empty((n_groups * S, src.shape[1]), dtype=src.dtype, device=src.device)
    assert out.is_contiguous() and out.shape[0] >= n_groups * S and out.shape[1] == src.shape[1] and out.dtype == src.dtype
    _l.check(_l.load().dsvg_gather_groups(_dt(src), src.data_ptr(), idx.data_ptr(), out.data_ptr(), n_groups, S,
                                          src.shape[1], n_src, _stream()), "dsvg_gather_groups")
    return out


def pack_tokens(commands, args, key_mask, n_seq, S):
    """packed token layout of the first encoder stage (include/dsvg.h): returns seq_off int32 [n_seq+1] and the packed
    commands [n_seq*S], args [n_seq*S, n_args], positions int32 [n_seq*S] (rows past seq_off[-1] replicate token 0)"""
    _chk(commands, args, key_mask)
    assert commands.dtype == torch.float32 and args.dtype == torch.float32 and commands.is_contiguous() and \
        args.is_contiguous() and commands.numel() == n_seq * S
    n_args = args.numel() // (n_seq * S)
    dev = commands.device
    seq_off = torch.empty(n_seq + 1, dtype=torch.int32, device=dev)
    pcmd = torch.empty(n_seq * S, dtype=torch.float32, device=dev)
    parg = torch.empty((n_seq * S, n_args), dtype=torch.float32, device=dev)
    ppos = torch.empty(n_seq * S, dtype=torch.int32, device=dev)
    _l.check(_l.load().dsvg_pack_tokens(commands.data_ptr(), args.data_ptr(), key_mask.data_ptr(), n_seq, S, n_args,
                                        seq_off.data_ptr(), pcmd.data_ptr(), parg.data_ptr(), ppos.data_ptr(),
                                        _stream()), "dsvg_pack_tokens")
    return seq_off, pcmd, parg, ppos


def bcast_add_fwd_(x, g, n_seq, S, drop_p=0.0, drop_site=0, seed=None):
    """in place: x[t] += drop(g[t // S])"""
    _chk(x, g, seed)
    assert x.is_contiguous() and g.is_contiguous() and g.dtype == x.dtype
    d = x.shape[1]
    _l.check(_l.load().dsvg_bcast_add_fwd(_dt(x), x.data_ptr(), g.data_ptr(), n_seq, S, d, float(drop_p),
                                          int(drop_site), _p(seed) if drop_p > 0 else None, _stream()),
             "dsvg_bcast_add_fwd")
    return x


def bcast_add_bwd(dx, n_seq, S, drop_p=0.0, drop_site=0, seed=None, n_seq_out=None, mask_site=None, out=None):
    """dg[b] = mask[b] * sum_s dx[b * S + s] for b < n_seq; with n_seq_out > n_seq the result has n_seq_out rows, the extra
    ones zero (sequences past a live prefix).
    mask_site (bf16, drop_p > 0): -> (dg, dxm) with dxm = drop_apply(dx, drop_p, mask_site, seed) from the same launch (one
    read of dx; dx may have rows past the n_seq summed sequences - a rounded-up live prefix - up to n_seq_out * S)"""
    _chk(dx, seed)
    assert dx.is_contiguous()
    d = dx.shape[1]
    n_out = n_seq if n_seq_out is None else int(n_seq_out)
    if out is None:
        dg = torch.empty((n_out, d), dtype=dx.dtype, device=dx.device)
    else:
        # a column block of a wider row-major buffer (the conditioning gradients of a stack's layers side by side)
        assert (out.dtype == dx.dtype == torch.bfloat16 and tuple(out.shape) == (n_out, d) and out.stride(1) == 1
                and out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0 and d % 8 == 0 and d <= 512 and dx.data_ptr() % 16 == 0)
        _chk(out)
        dg = out
    ld = dg.stride(0)
    if mask_site is not None:
        assert dx.dtype == torch.bfloat16 and drop_p > 0 and n_seq * S <= dx.shape[0] <= n_out * S
        dxm = torch.empty_like(dx)
        _l.check(_l.load().dsvg_bcast_add_bwd_masked(dx.data_ptr(), dg.data_ptr(), dxm.data_ptr(), n_seq, n_out, S, d,
                                                     dx.shape[0], float(drop_p), int(drop_site), int(mask_site), _p(seed),
                                                     int(ld), _stream()),
                 "dsvg_bcast_add_bwd_masked")
        return dg, dxm
    _l.check(_l.load().dsvg_bcast_add_bwd(_dt(dx), dx.data_ptr(), dg.data_ptr(), n_seq, n_out, S, d, float(drop_p),
                                          int(drop_site), _p(seed) if drop_p > 0 else None, int(ld), _stream()),
             "dsvg_bcast_add_bwd")
    return dg


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------
def loss_targets(tgt_commands, tgt_args, cmd_args_mask, eos_id=4, seq_perm=None):
    """tgt_commands float32 [n_seq, S1], tgt_args float32 [n_seq, S1, n_args], cmd_args_mask float32 [n_cmd,n_args].
    seq_perm (int32 [n_seq]): the token-level results of sequence b are those of source sequence seq_perm[b]; vis_tgt stays in
    source order"""
    _chk(tgt_commands, tgt_args, cmd_args_mask, seq_perm)
    assert seq_perm is None or (seq_perm.dtype == torch.int32 and seq_perm.numel() >= tgt_commands.shape[0])
    assert tgt_commands.is_contiguous() and tgt_args.is_contiguous() and cmd_args_mask.is_contiguous()
    assert tgt_commands.dtype == torch.float32 and tgt_args.dtype == torch.float32
    assert cmd_args_mask.dtype == torch.float32
    n_seq, S1 = tgt_commands.shape
    n_args = tgt_args.shape[-1]
    n_cmd = cmd_args_mask.shape[0]
    S = S1 - 1
    dev = tgt_commands.device
    cmd_tgt = torch.empty((n_seq, S), dtype=torch.int32, device=dev)
    cmd_w = torch.empty((n_seq, S), dtype=torch.float32, device=dev)
    arg_tgt = torch.empty((n_seq, S, n_args), dtype=torch.int32, device=dev)
    arg_w = torch.empty((n_seq, S, n_args), dtype=torch.float32, device=dev)
    vis_tgt = torch.empty(n_seq, dtype=torch.int32, device=dev)
    _l.check(_l.load().dsvg_loss_targets(tgt_commands.data_ptr(), tgt_args.data_ptr(), cmd_args_mask.data_ptr(),
                                         n_seq, S1, n_args, n_cmd, eos_id, cmd_tgt.data_ptr(), cmd_w.data_ptr(),
                                         arg_tgt.data_ptr(), arg_w.data_ptr(), vis_tgt.data_ptr(), _p(seq_perm), _stream()),
             "dsvg_loss_targets")
    return cmd_tgt, cmd_w, arg_tgt, arg_w, vis_tgt


def masked_ce_fwd(logits2d, target, w, C_, group=1, tok_idx=None):
    """logits2d: [n_tok, >= group*C] row-major (stride(0) = token stride); logical rows = n_tok*group.
    returns lse [rows] fp32, sum_count fp32[2].  tok_idx (int32 [n_tok]): the logits (and lse) are COMPACT - row i
    belongs to source token tok_idx[i] (negative: padding) - while target / w stay indexed by the source token."""
    _chk(logits2d, target, w, tok_idx)
    _rowmajor(logits2d)
    rows = logits2d.shape[0] * group
    assert target.dtype == torch.int32 and target.is_contiguous()
    if tok_idx is None:
        assert target.numel() == rows and (w is None or w.numel() == rows)
    else:
        assert tok_idx.dtype == torch.int32 and tok_idx.is_contiguous() and tok_idx.numel() == logits2d.shape[0]
    if w is not None:
        assert w.dtype == torch.float32 and w.is_contiguous()
    dev = logits2d.device
    lse = torch.empty(rows, dtype=torch.float32, device=dev)
    sc = torch.empty(2, dtype=torch.float32, device=dev)
    L = _l.load()
    ws = _ws(L.dsvg_masked_ce_workspace_bytes(rows), dev)
    _l.check(L.dsvg_masked_ce_fwd(_dt(logits2d), logits2d.data_ptr(), logits2d.stride(0), group, target.data_ptr(),
                                  _p(w), rows, C_, lse.data_ptr(), sc.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                  _p(tok_idx), _stream()), "dsvg_masked_ce_fwd")
    return lse, sc


def masked_ce_bwd(logits2d, target, w, lse, sum_count, gscale, coef, C_, group=1, pad_to=8, tok_idx=None,
                  logits_compact=False):
    """returns dlogits as a [n_tok, group*C] view of a buffer whose token stride is padded to `pad_to` elements.
    tok_idx (int32 [n_out]): compact backward - output token i is source token tok_idx[i], negative -> zero row;
    logits_compact: logits2d / lse are compact too (as produced by masked_ce_fwd(..., tok_idx=))."""
    _chk(logits2d, target, w, lse, sum_count, gscale, tok_idx)
    n_tok = logits2d.shape[0] if tok_idx is None else tok_idx.numel()
    assert tok_idx is None or (tok_idx.dtype == torch.int32 and tok_idx.is_contiguous())
    rows = n_tok * group
    width = group * C_
    ld_d = (width + pad_to - 1) // pad_to * pad_to
    buf = torch.empty((n_tok, ld_d), dtype=logits2d.dtype, device=logits2d.device)
    assert gscale is None or (gscale.dtype == torch.float32 and gscale.numel() == 1)
    _l.check(_l.load().dsvg_masked_ce_bwd(_dt(logits2d), logits2d.data_ptr(), logits2d.stride(0), group,
                                          target.data_ptr(), _p(w), lse.data_ptr(), sum_count.data_ptr(), _p(gscale),
                                          float(coef), buf.data_ptr(), ld_d, rows, C_, _p(tok_idx),
                                          int(bool(logits_compact)), _stream()), "dsvg_masked_ce_bwd")
    return buf[:, :width]


def loss_combine_fwd(scs, weights):
    """scs: fp32 [2] (sum, count) device tensors of n <= 4 cross-entropies -> fp32 [1 + n]: weighted total, then the terms"""
    _chk(*scs)
    n = len(scs)
    assert 1 <= n <= 4 and len(weights) == n and all(t.dtype == torch.float32 and t.numel() == 2 and t.is_contiguous() for t in scs)
    out = torch.empty(1 + n, dtype=torch.float32, device=scs[0].device)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in scs])
    w = (C.c_float * n)(*[float(x) for x in weights])
    _l.check(_l.load().dsvg_loss_combine_fwd(ptrs, w, n, out.data_ptr(), _stream()), "dsvg_loss_combine_fwd")
    return out


def loss_combine_bwd(dtotal, dterms, weights, device):
    """-> fp32 [n, 2]: row i = (dtotal * weights[i] + dterms[i], 0); dtotal / dterms[i]: fp32 scalar device tensors or None"""
    n = len(weights)
    _chk(dtotal, *[t for t in dterms if t is not None])
    for t in [dtotal] + list(dterms):
        assert t is None or (t.dtype == torch.float32 and t.numel() == 1)
    dsc = torch.empty((n, 2), dtype=torch.float32, device=device)
    ptrs = (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in dterms])
    w = (C.c_float * n)(*[float(x) for x in weights])
    _l.check(_l.load().dsvg_loss_combine_bwd(_p(dtotal), ptrs, w, n, dsc.data_ptr(), _stream()), "dsvg_loss_combine_bwd")
    return dsc


def live_rows(w, group):
    """w float32 [n_tok * group] -> (live int32 [n_tok]: ascending tokens with any non-zero weight, -1 padded;
    count int32 [1])"""
    _chk(w)
    assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() % group == 0
    n_tok = w.numel() // group
    live = torch.empty(n_tok, dtype=torch.int32, device=w.device)
    count = torch.empty(1, dtype=torch.int32, device=w.device)
    L = _l.load()
    ws = _ws(L.dsvg_live_rows_workspace_bytes(n_tok), w.device)
    _l.check(L.dsvg_live_rows(w.data_ptr(), n_tok, group, live.data_ptr(), count.data_ptr(), ws.data_ptr(),
                              ws.numel() * 4, _stream()), "dsvg_live_rows")
    return live, count


def scatter_rows(src, idx, dst, accumulate=False):
    """dst[idx[i]] = src[i] (accumulate: +=) for idx[i] >= 0"""
    _chk(src, idx, dst)
    assert src.is_contiguous() and dst.is_contiguous() and idx.dtype == torch.int32 and src.dtype == dst.dtype
    assert src.shape[1] == dst.shape[1] and idx.numel() >= src.shape[0]
    _l.check(_l.load().dsvg_scatter_rows(_dt(src), src.data_ptr(), idx.data_ptr(), dst.data_ptr(), src.shape[0],
                                         src.shape[1], int(bool(accumulate)), _stream()), "dsvg_scatter_rows")
    return dst


# ------------------------------------------------------------------------------------------------
# Hungarian self-matching (model.py:311-350)
# ------------------------------------------------------------------------------------------------
def match_costs(cmd_logits, args_logits, vis_logits, tgt_commands, tgt_args, cam, N, G, Gp, n_args, args_dim, n_cmd,
                eos_id, weights=(2.0, 1.0, 1.0)):
    """cmd_logits [N*Gp*S, n_cmd] / args_logits [N*Gp*S, n_args*args_dim] / vis_logits [N*Gp, 2] (row-strided 2-D views),
    tgt_commands [N, G, S+1] / tgt_args [N, G, S+1, n_args] float32  ->  cost f32 [N, G, Gp], visible int32 [N, G]"""
    _chk(cmd_logits, args_logits, vis_logits, tgt_commands, tgt_args, cam)
    assert cmd_logits.dtype == args_logits.dtype == vis_logits.dtype
    for t in (cmd_logits, args_logits, vis_logits):
        assert t.dim() == 2 and t.stride(1) == 1
    assert tgt_commands.dtype == torch.float32 and tgt_args.dtype == torch.float32 and cam.dtype == torch.float32
    assert tgt_commands.is_contiguous() and tgt_args.is_contiguous() and cam.is_contiguous()
    S1 = tgt_commands.shape[-1]
    assert cmd_logits.shape[0] == N * Gp * (S1 - 1) and vis_logits.shape[0] == N * Gp
    cost = torch.empty(N, G, Gp, dtype=torch.float32, device=cmd_logits.device)
    vis = torch.empty(N, G, dtype=torch.int32, device=cmd_logits.device)
    _l.check(_l.load().dsvg_match_costs(_dt(cmd_logits), cmd_logits.data_ptr(), cmd_logits.stride(0),
                                        args_logits.data_ptr(), args_logits.stride(0), vis_logits.data_ptr(),
                                        vis_logits.stride(0), tgt_commands.data_ptr(), tgt_args.data_ptr(),
                                        cam.data_ptr(), N, G, Gp, S1, n_args, args_dim, n_cmd, eos_id,
                                        float(weights[0]), float(weights[1]), float(weights[2]), cost.data_ptr(),
                                        vis.data_ptr(), _stream()), "dsvg_match_costs")
    return cost, vis


def argmax_rows(logits2d, C, group=1):
    """logits2d [n_tok, >= group*C] (row-strided view) -> int32 [n_tok * group]: arg-max of every C-wide class slot"""
    _chk(logits2d)
    assert logits2d.dim() == 2 and logits2d.stride(1) == 1 and logits2d.shape[1] >= group * C
    rows = logits2d.shape[0] * group
    out = torch.empty(rows, dtype=torch.int32, device=logits2d.device)
    _l.check(_l.load().dsvg_argmax_rows(_dt(logits2d), logits2d.data_ptr(), logits2d.stride(0), group, rows, C,
                                        out.data_ptr(), _stream()), "dsvg_argmax_rows")
    return out


SITE_SAMPLE_CMD, SITE_SAMPLE_ARGS = 7001, 7002     # noise streams of the two categorical draws of greedy_sample


def sample_rows(logits2d, C, temperature, seed, site, group=1):
    """a draw from softmax(row / temperature) per C-wide class slot (Gumbel arg-max on the device, include/dsvg.h) ->
    int32 [n_tok * group]; seed: int64 [1] device tensor"""
    _chk(logits2d, seed)
    assert logits2d.dim() == 2 and logits2d.stride(1) == 1 and logits2d.shape[1] >= group * C and temperature > 0
    rows = logits2d.shape[0] * group
    out = torch.empty(rows, dtype=torch.int32, device=logits2d.device)
    _l.check(_l.load().dsvg_sample_rows(_dt(logits2d), logits2d.data_ptr(), logits2d.stride(0), group, rows, C,
                                        float(temperature), seed.data_ptr(), int(site), out.data_ptr(), _stream()),
             "dsvg_sample_rows")
    return out


# ---- the argument head fused with its consumers (csrc/head_fused.hip) -------------------------------------------------------
def head_pack(weight_lp):
    """bf16 [n_out, 256] rows of the head in use (contiguous) -> packed MFMA fragment image for the three head_* kernels"""
    _chk(weight_lp)
    assert weight_lp.dtype == torch.bfloat16 and weight_lp.dim() == 2 and weight_lp.shape[1] == 256 and weight_lp.is_contiguous()
    L = _l.load()
    img = torch.empty(L.dsvg_head_pack_elems(weight_lp.shape[0]), dtype=torch.bfloat16, device=weight_lp.device)
    _l.check(L.dsvg_head_pack(weight_lp.data_ptr(), weight_lp.shape[0], img.data_ptr(), _stream()), "dsvg_head_pack")
    return img


def _head_args(x, packed, bias, n_out, C):
    _chk(x, packed, bias)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and x.is_contiguous()
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == n_out and n_out % C == 0


def head_argmax(x, packed, bias, n_out, C):
    """-> int32 [rows * (n_out // C)]: arg-max of every C-wide slot of x @ W^T + bias, the logits never stored"""
    _head_args(x, packed, bias, n_out, C)
    out = torch.empty(x.shape[0] * (n_out // C), dtype=torch.int32, device=x.device)
    _l.check(_l.load().dsvg_head_argmax(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), x.shape[0], n_out, C,
                                        out.data_ptr(), _stream()), "dsvg_head_argmax")
    return out


def head_sample(x, packed, bias, n_out, C, temperature, seed, site):
    """-> int32 [rows * (n_out // C)]: a draw from softmax(slot logits / temperature) per C-wide slot of x @ W^T + bias, the
    logits never stored (same draws as sample_rows on the dense logits)"""
    _head_args(x, packed, bias, n_out, C)
    _chk(seed)
    out = torch.empty(x.shape[0] * (n_out // C), dtype=torch.int32, device=x.device)
    _l.check(_l.load().dsvg_head_sample(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), x.shape[0], n_out, C,
                                        float(temperature), seed.data_ptr(), int(site), out.data_ptr(), _stream()),
             "dsvg_head_sample")
    return out


def head_lse(x, packed, bias, n_out, C, target, w, tok_idx=None):
    """masked CE forward on x @ W^T + bias without the logits -> (lse [rows * group], (sum, count) [2]); target / w / tok_idx
    as masked_ce_fwd takes them (compact logits: row i of x is token tok_idx[i])"""
    _head_args(x, packed, bias, n_out, C)
    _chk(target, w, tok_idx)
    rows, group = x.shape[0], n_out // C
    assert target.dtype == torch.int32 and (w is None or w.dtype == torch.float32)
    L = _l.load()
    lse = torch.empty(rows * group, dtype=torch.float32, device=x.device)
    sc = torch.empty(2, dtype=torch.float32, device=x.device)
    nb = L.dsvg_head_lse_workspace_bytes(rows)
    ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
    _l.check(L.dsvg_head_lse(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), rows, n_out, C, target.data_ptr(),
                             w.data_ptr() if w is not None else None, tok_idx.data_ptr() if tok_idx is not None else None,
                             lse.data_ptr(), sc.data_ptr(), ws.data_ptr(), nb, _stream()), "dsvg_head_lse")
    return lse, sc


def head_dlogits(x, packed, bias, n_out, C, target, w, lse, sum_count, gscale, coef, tok_idx=None):
    """-> bf16 [rows, n_out rounded up to 8] (view of its first n_out columns): w g (softmax - onehot), logits recomputed"""
    _head_args(x, packed, bias, n_out, C)
    _chk(target, w, tok_idx, lse, sum_count, gscale)
    ld = (n_out + 7) // 8 * 8
    buf = torch.empty((x.shape[0], ld), dtype=torch.bfloat16, device=x.device)
    _l.check(_l.load().dsvg_head_dlogits(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), x.shape[0], n_out, C,
                                         target.data_ptr(), w.data_ptr() if w is not None else None,
                                         tok_idx.data_ptr() if tok_idx is not None else None, lse.data_ptr(),
                                         sum_count.data_ptr(), gscale.data_ptr() if gscale is not None else None,
                                         float(coef), buf.data_ptr(), ld, _stream()), "dsvg_head_dlogits")
    return buf[:, :n_out]


def match_assign(cost, vis):
    """cost f32 [N, G, Gp], visible int32 [N, G] -> assign [N, Gp], idx [N*Gp], inv [N*Gp] (all int32)"""
    _chk(cost, vis)
    assert cost.dtype == torch.float32 and cost.is_contiguous() and vis.dtype == torch.int32 and vis.is_contiguous()
    N, G, Gp = cost.shape
    assign = torch.empty(N, Gp, dtype=torch.int32, device=cost.device)
    idx = torch.empty(N * Gp, dtype=torch.int32, device=cost.device)
    inv = torch.empty(N * Gp, dtype=torch.int32, device=cost.device)
    _l.check(_l.load().dsvg_match_assign(cost.data_ptr(), vis.data_ptr(), N, G, Gp, assign.data_ptr(), idx.data_ptr(),
                                         inv.data_ptr(), _stream()), "dsvg_match_assign")
    return assign, idx, inv


# ------------------------------------------------------------------------------------------------
# device-side batch assembly (svgtensor_dataset.py:164-205)
# ------------------------------------------------------------------------------------------------
def assemble_batch(rows, slot_off, variant, G, L, grouped, want_args=True, want_rel=False, pad_val=-1.0,
                   args_dim=256):
    """rows int16 [R, 12], slot_off int32 [n_variants*G + 1], variant int32 [N]  ->
    commands f32 [N, G or 1, L], args / args_rel f32 [N, G or 1, L, 11] (None when not wanted)"""
    _chk(rows, slot_off, variant)
    assert rows.dtype == torch.int16 and rows.dim() == 2 and rows.shape[1] == 12 and rows.is_contiguous()
    assert slot_off.dtype == torch.int32 and variant.dtype == torch.int32
    assert slot_off.is_contiguous() and variant.is_contiguous()
    N = variant.numel()
    Gs = 1 if grouped else G
    dev = rows.device
    commands = torch.empty(N, Gs, L, dtype=torch.float32, device=dev)
    args = torch.empty(N, Gs, L, 11, dtype=torch.float32, device=dev) if want_args else None
    rel = torch.empty(N, Gs, L, 11, dtype=torch.float32, device=dev) if want_rel else None
    _l.check(_l.load().dsvg_assemble_batch(rows.data_ptr(), rows.shape[0], slot_off.data_ptr(), slot_off.numel() - 1,
                                           variant.data_ptr(), N, G, 1 if grouped else 0, L,
                                           float(pad_val), int(args_dim), commands.data_ptr(),
                                           args.data_ptr() if want_args else None,
                                           rel.data_ptr() if want_rel else None, _stream()), "dsvg_assemble_batch")
    return commands, args, rel


# ------------------------------------------------------------------------------------------------
# optimizer / housekeeping
# ------------------------------------------------------------------------------------------------
def sumsq(x, out=None):
    _chk(x, out)
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    L = _l.load()
    ws = _ws(L.dsvg_sumsq_workspace_bytes(x.numel()), x.device)
    _l.check(L.dsvg_sumsq(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()),
             "dsvg_sumsq")
    return out


def adamw_step_(p, g, m, v, lr, step, *, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, gnorm_sq=None,
                max_norm=0.0, grad_scale=1.0):
    """in-place AdamW (+ optional global-norm clip) on flat fp32 buffers; lr: float32[1], step: int64[1] on device."""
    _chk(p, g, m, v, lr, step, gnorm_sq)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    assert lr.dtype == torch.float32 and step.dtype == torch.int64
    _l.check(_l.load().dsvg_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                       lr.data_ptr(), beta1, beta2, eps, weight_decay, step.data_ptr(),
                                       _p(gnorm_sq), float(max_norm), float(grad_scale), _stream()),
             "dsvg_adamw_step")


def cast_weights(src, dst=None, dst_t=None):
    """dst = cast(src [rows, cols]); dst_t = cast(src)^T"""
    _chk(src, dst, dst_t)
    assert src.dtype == torch.float32 and src.is_contiguous()
    if src.dim() == 1:
        rows, cols = 1, src.numel()
    else:
        rows, cols = src.shape
    ref = dst if dst is not None else dst_t
    _l.check(_l.load().dsvg_cast_weights(_dt(ref), src.data_ptr(), _p(dst), _p(dst_t), rows, cols, _stream()),
             "dsvg_cast_weights")
    return dst, dst_t


# ------------------------------------------------------------------------------------------------
# fused FFN sub-block (bf16, d_model 256 / dim_ff 512)
# ------------------------------------------------------------------------------------------------
FFN_FWD_CHUNKS, FFN_FWD_LAYER_ELEMS, FFN_BWD_LAYER_ELEMS = 16, 16 * 32 * 512, 16 * 48 * 512
_FFN_STAGES = int(os.environ.get("DSVG_FFN_STAGES", "0"))     # LDS ring depth of the fused kernels (0 = default), tuning knob


def ffn_pack(flat, offs, n_layers, packed_fwd=None, packed_bwd=None, b1f=None, w2p=None):
    """fragment-major bf16 images of linear1 / linear2 of n_layers layers, straight from the fp32 master buffer `flat`,
    with the LayerNorm affine folded into linear1 (include/dsvg.h).  offs: int64 device tensor [n_layers, 5] of element
    offsets (linear1.weight, linear1.bias, linear2.weight, norm.weight, norm.bias).  w2p (optional bf16
    [n_layers, 256, 512]): linear2.weight with fragment-ordered columns.  -> (packed_fwd, packed_bwd, b1f)"""
    _chk(flat, offs, packed_fwd, packed_bwd, b1f)
    assert flat.dtype == torch.float32 and offs.dtype == torch.int64 and tuple(offs.shape) == (n_layers, 5)
    assert offs.is_contiguous()
    dev = flat.device
    if packed_fwd is None:
        packed_fwd = torch.empty(n_layers * FFN_FWD_LAYER_ELEMS, dtype=torch.bfloat16, device=dev)
    if packed_bwd is None:
        packed_bwd = torch.empty(n_layers * FFN_BWD_LAYER_ELEMS, dtype=torch.bfloat16, device=dev)
    if b1f is None:
        b1f = torch.empty((n_layers, 512), dtype=torch.float32, device=dev)
    assert packed_fwd.numel() == n_layers * FFN_FWD_LAYER_ELEMS and packed_bwd.numel() == n_layers * FFN_BWD_LAYER_ELEMS
    assert b1f.numel() == n_layers * 512 and b1f.dtype == torch.float32
    assert w2p is None or (w2p.dtype == torch.bfloat16 and w2p.numel() == n_layers * 131072 and w2p.is_contiguous())
    _l.check(_l.load().dsvg_ffn_pack(flat.data_ptr(), offs.data_ptr(), n_layers, 256, 512, packed_fwd.data_ptr(),
                                     packed_bwd.data_ptr(), b1f.data_ptr(), _p(w2p), _stream()), "dsvg_ffn_pack")
    return packed_fwd, packed_bwd, b1f


def ffn_fwd(x, packed_fwd_layer, b1f, b2, eps=1e-5, drop_p=0.0, site_hidden=0, site_res=0, seed=None, out=None,
            train=False, into=None, stages=None):
    """y = x + drop_r(linear2(drop_h(relu(linear1(LayerNorm(x))))))  (x bf16 [rows, 256]; the LayerNorm's gamma / beta
    are inside packed_fwd_layer / b1f, see ffn_pack).  train: -> (y, h, xh, rstd) with h bf16 [rows, 512] in fragment
    order, xh = (x - mean) * rstd bf16, rstd fp32 [rows] (what the backward pass needs).
    stages: None = the library's choice (half-size workgroups up to 32,768 rows, packed activation code), 2 = half-size
    workgroups, 3 / 4 = the 256-row workgroups with that many weight-ring slots, all three with the scalar activation code and
    bit-identical; 7 / 6 = half-size / 256-row workgroups with the packed activation code (bit-identical to each other and to
    the default; rounding-level differences to 2 / 3 / 4), 5 = the role-specialised 128-row kernel (tests and probes)"""
    _chk(x, packed_fwd_layer, b1f, b2, seed, out)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and x.is_contiguous()
    assert packed_fwd_layer.numel() == FFN_FWD_LAYER_ELEMS and packed_fwd_layer.is_contiguous()
    assert b1f.numel() == 512 and b2.numel() == 256 and b1f.dtype == torch.float32 and b2.dtype == torch.float32
    rows = x.shape[0]
    if out is None:
        out = torch.empty_like(x)
    h = xh = rstd = None
    if train:
        if into is not None:    # (h, xh) given: row slices of longer buffers
            h, xh = into
            assert h.is_contiguous() and xh.is_contiguous() and tuple(h.shape) == (rows, 512) and xh.shape == x.shape
        else:
            h = torch.empty((rows, 512), dtype=x.dtype, device=x.device)
            xh = torch.empty_like(x)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    ev = _prof_begin()
    _l.check(_l.load().dsvg_ffn_fwd(x.data_ptr(), packed_fwd_layer.data_ptr(), b1f.data_ptr(), b2.data_ptr(),
                                    out.data_ptr(), _p(h), _p(xh), _p(rstd), rows, float(eps), float(drop_p),
                                    int(site_hidden), int(site_res), _p(seed) if drop_p > 0 else None,
                                    _FFN_STAGES if stages is None else int(stages), _stream()), "dsvg_ffn_fwd")
    # 2 GEMMs of 2 * 256 * 512 FLOP per row; algorithmic bytes: the row in, the row out, plus - in
    # training - h and xh for the backward pass
    _prof_end(ev, 4.0 * 256 * 512 * rows, (2.0 * 512 + (1536.0 if train else 0.0)) * rows, dict(op="ffn_fwd", rows=rows))
    return (out, h, xh, rstd) if train else out


def ffn_bwd(x, dy, packed_bwd_layer, b1f, eps=1e-5, drop_p=0.0, site_hidden=0, site_res=0, seed=None):
    """backward of ffn_fwd with respect to x, plus the operands of the weight-gradient GEMMs (include/dsvg.h):
    -> (dx, h, dpre, xh, dym); h / dpre [rows, 512] have their hidden columns in fragment order; dym is dy itself
    when there is no dropout"""
    _chk(x, dy, packed_bwd_layer, b1f, seed)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and x.is_contiguous()
    assert dy.dtype == x.dtype and dy.shape == x.shape and dy.is_contiguous()
    assert packed_bwd_layer.numel() == FFN_BWD_LAYER_ELEMS and packed_bwd_layer.is_contiguous() and b1f.numel() == 512
    rows = x.shape[0]
    dx, xh = torch.empty_like(x), torch.empty_like(x)
    dym = torch.empty_like(x) if drop_p > 0 else dy
    h = torch.empty((rows, 512), dtype=x.dtype, device=x.device)
    dpre = torch.empty((rows, 512), dtype=x.dtype, device=x.device)
    ev = _prof_begin()
    _l.check(_l.load().dsvg_ffn_bwd(x.data_ptr(), dy.data_ptr(), packed_bwd_layer.data_ptr(), b1f.data_ptr(),
                                    h.data_ptr(), dpre.data_ptr(), xh.data_ptr(), dym.data_ptr() if drop_p > 0 else None,
                                    dx.data_ptr(), rows, float(eps), float(drop_p), int(site_hidden), int(site_res),
                                    _p(seed) if drop_p > 0 else None, _stream()), "dsvg_ffn_bwd")
    # algorithmic work of the two launches: dh and dxh (2 GEMMs); the recomputed pre-activation is not counted.
    # bytes: x, dy in; dx out; h, dpre, xh (, dym) out for the weight-gradient GEMMs; dpre read back by kernel 2
    _prof_end(ev, 4.0 * 256 * 512 * rows, (3 * 512 + 3 * 1024 + 512 + (512 if drop_p > 0 else 0)) * float(rows),
              dict(op="ffn_bwd", rows=rows))
    return dx, h, dpre, xh, dym


def ffn_bwd_dx(dpre, x, dy, packed_bwd_layer, eps=1e-5, masked=None):
    """dx = dy + LayerNorm'(dpre . W1')  (dpre bf16 [rows, 512] in fragment order; include/dsvg.h).
    masked = (drop_p, drop_site, seed): also return drop_apply(dx, drop_p, drop_site, seed) from the same launch -> (dx, dxm)"""
    _chk(dpre, x, dy, packed_bwd_layer)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and x.is_contiguous()
    assert dy.dtype == x.dtype and dy.shape == x.shape and dy.is_contiguous()
    assert dpre.dtype == x.dtype and tuple(dpre.shape) == (x.shape[0], 512) and dpre.is_contiguous()
    assert packed_bwd_layer.numel() == FFN_BWD_LAYER_ELEMS and packed_bwd_layer.is_contiguous()
    dx = torch.empty_like(x)
    dxm, mp, msite, mseed = None, 0.0, 0, None
    if masked is not None and masked[0] > 0:
        mp, msite, mseed = float(masked[0]), int(masked[1]), masked[2]
        _chk(mseed)
        dxm = torch.empty_like(x)
    ev = _prof_begin()
    _l.check(_l.load().dsvg_ffn_bwd_dx(dpre.data_ptr(), x.data_ptr(), dy.data_ptr(), packed_bwd_layer.data_ptr(),
                                       dx.data_ptr(), x.shape[0], float(eps), _p(dxm), mp, msite, _p(mseed), _stream()),
             "dsvg_ffn_bwd_dx")
    _prof_end(ev, 2.0 * 256 * 512 * x.shape[0], (1024.0 + 3 * 512) * x.shape[0], dict(op="ffn_bwd_dx", rows=x.shape[0]))
    if masked is not None:
        return dx, (dxm if dxm is not None else dx)
    return dx


def ffn_wgrad_finish(g1p, db1p, g2p, w1, gamma, beta, dw1, db1, dw2, dgamma, dbeta):
    """(G1p = dpre^T xh, its row sums, G2p = dym^T h) in fragment order -> gradients of linear1.weight / bias,
    linear2.weight, norm.weight, norm.bias (include/dsvg.h)"""
    ts = (g1p, db1p, g2p, w1, gamma, beta, dw1, db1, dw2, dgamma, dbeta)
    _chk(*ts)
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in ts)
    assert g1p.numel() == 131072 and g2p.numel() == 131072 and db1p.numel() == 512 and w1.numel() == 131072
    assert dw1.numel() == 131072 and dw2.numel() == 131072 and db1.numel() == 512 and dgamma.numel() == 256
    assert dbeta.numel() == 256 and gamma.numel() == 256 and beta.numel() == 256
    ev = _prof_begin()
    _l.check(_l.load().dsvg_ffn_wgrad_finish(*(t.data_ptr() for t in ts), _stream()), "dsvg_ffn_wgrad_finish")
    _prof_end(ev, 0.0, 0.0, dict(op="ffn_wgrad_finish"))


def ffn_wgrad_finish_many(layers):
    """ffn_wgrad_finish for a list of layers (each the 11 tensors of ffn_wgrad_finish) in one launch per 16"""
    if not layers:
        return
    flat = [t for ts in layers for t in ts]
    _chk(*flat)
    assert all(len(ts) == 11 for ts in layers) and all(t.dtype == torch.float32 and t.is_contiguous() for t in flat)
    ptrs = (C.c_void_p * len(flat))(*[t.data_ptr() for t in flat])
    ev = _prof_begin()
    _l.check(_l.load().dsvg_ffn_wgrad_finish_many(ptrs, len(layers), _stream()), "dsvg_ffn_wgrad_finish_many")
    _prof_end(ev, 0.0, 0.0, dict(op="ffn_wgrad_finish"))


def ffn_wgrad_finish_deferred(*ts):
    """ffn_wgrad_finish behind the queued reductions of the current stream's open deferral scope (all such layers of a
    backward pass in ONE launch at the flush); without an open scope: now"""
    st = _DEFER.state.get(_stream_key())
    if st is None:
        ffn_wgrad_finish(*ts)
    else:
        st.finish.append(ts)


# ------------------------------------------------------------------------------------------------
# fused attention sub-block (csrc/attn_fused.hip)
# ------------------------------------------------------------------------------------------------
ATTN_LAYER_ELEMS = 512 * 512        # bf16 elements of one layer's packed in_proj + out_proj image (512 KiB)


def attn_pack(flat, offs, n_layers, packed=None):
    """bf16 MFMA-fragment images of in_proj_weight / out_proj.weight of n_layers layers from the fp32 master buffer
    (include/dsvg.h).  offs: int64 device tensor [n_layers, 2] of element offsets (in_proj_weight, out_proj.weight)."""
    _chk(flat, offs, packed)
    assert flat.dtype == torch.float32 and offs.dtype == torch.int64 and tuple(offs.shape) == (n_layers, 2)
    assert offs.is_contiguous()
    if packed is None:
        packed = torch.empty(n_layers * ATTN_LAYER_ELEMS, dtype=torch.bfloat16, device=flat.device)
    assert packed.numel() == n_layers * ATTN_LAYER_ELEMS and packed.dtype == torch.bfloat16
    _l.check(_l.load().dsvg_attn_pack(flat.data_ptr(), offs.data_ptr(), n_layers, 256, 8, packed.data_ptr(), _stream()),
             "dsvg_attn_pack")
    return packed


def attn_block_fwd(x, packed_layer, in_bias, out_bias, gamma, beta, key_mask, n_seq, S, scale, eps=1e-5, drop_p=0.0,
                   site_probs=0, site_res=0, seed=None, seq_off=None, tiles=None, train=False, seq_add=None, site_seq_add=0,
                   into=None):
    """x1 = x + drop_r(out_proj(MHA(LayerNorm(x)))) [+ drop(seq_add[sequence])]  (x bf16 [rows, 256], 8 heads, S <= 32)
    in one launch.  seq_add (bf16 [n_seq, 256], dense layouts): the decoder's per-sequence conditioning term.
    train=False -> x1;  train=True -> (x1, xn, qkv, ao, mean, rstd): what the unfused backward reads."""
    _chk(x, packed_layer, in_bias, out_bias, gamma, beta, key_mask, seed, seq_off, tiles)
    rows = x.shape[0]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[1] == 256
    assert packed_layer.numel() == ATTN_LAYER_ELEMS and packed_layer.is_contiguous()
    assert in_bias.numel() == 768 and out_bias.numel() == 256 and gamma.numel() == 256 and beta.numel() == 256
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (in_bias, out_bias, gamma, beta))
    assert (seq_off is None) == (tiles is None) and (seq_off is None or key_mask is None)
    if seq_add is not None:
        _chk(seq_add)
        # (row stride free: a column block of the stack's [n_seq, layers * 256] conditioning matrix, functional.GlobalCondFn)
        assert seq_off is None and seq_add.dtype == x.dtype and tuple(seq_add.shape) == (n_seq, 256)
        assert seq_add.stride(1) == 1 and seq_add.stride(0) % 8 == 0 and seq_add.data_ptr() % 16 == 0
    xn = qkv = ao = mean = rstd = None
    if into is not None:        # (x1, xn, qkv, ao, mean, rstd) given: row slices of longer buffers
        assert train and len(into) == 6 and all(t.is_contiguous() and t.shape[0] == rows for t in into)
        x1, xn, qkv, ao, mean, rstd = into
    else:
        x1 = torch.empty_like(x)
    if train and into is None:
        xn = torch.empty_like(x)
        qkv = torch.empty((rows, 768), dtype=x.dtype, device=x.device)
        ao = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    ev = _prof_begin()
    _l.check(_l.load().dsvg_attn_block_fwd(x.data_ptr(), packed_layer.data_ptr(), in_bias.data_ptr(), out_bias.data_ptr(),
                                           gamma.data_ptr(), beta.data_ptr(), _p(key_mask), _p(seq_off), _p(tiles), n_seq,
                                           S, rows, x1.data_ptr(), _p(xn), _p(qkv), _p(ao), _p(mean), _p(rstd), float(eps),
                                           float(scale), float(drop_p), int(site_probs), int(site_res),
                                           _p(seed) if drop_p > 0 else None, _p(seq_add),
                                           int(seq_add.stride(0)) if seq_add is not None else 256, int(site_seq_add), _stream()),
             "dsvg_attn_block_fwd")
    _prof_end(ev, 2.0 * rows * 256 * 1024 + 4.0 * rows * 32 * 256, 1024.0 * rows + (2560.0 * rows if train else 0.0),
              dict(op="attn_block_fwd", rows=rows, train=bool(train)))
    if train:
        return x1, xn, qkv, ao, mean, rstd
    return x1


# ------------------------------------------------------------------------------------------------
# fused layer of the short-sequence ("group") stages (csrc/group_stage.hip)
# ------------------------------------------------------------------------------------------------
GS_LAYER_ELEMS = 8 * 128 * 512      # bf16 elements of one layer's packed image (1 MiB), forward and backward each


def gs_pack(flat, offs, n_layers, packed_fwd=None, packed_bwd=None):
    """wave-major bf16 MFMA-fragment images of (in_proj_weight, out_proj.weight, linear1.weight, linear2.weight) of n_layers
    layers for the forward and the backward kernel, from the fp32 master buffer (include/dsvg.h).  offs: int64 device tensor
    [n_layers, 4] of element offsets.  -> (packed_fwd, packed_bwd)"""
    _chk(flat, offs, packed_fwd, packed_bwd)
    assert flat.dtype == torch.float32 and offs.dtype == torch.int64 and tuple(offs.shape) == (n_layers, 4)
    assert offs.is_contiguous()
    if packed_fwd is None:
        packed_fwd = torch.empty(n_layers * GS_LAYER_ELEMS, dtype=torch.bfloat16, device=flat.device)
    if packed_bwd is None:
        packed_bwd = torch.empty(n_layers * GS_LAYER_ELEMS, dtype=torch.bfloat16, device=flat.device)
    assert packed_fwd.numel() == n_layers * GS_LAYER_ELEMS and packed_bwd.numel() == n_layers * GS_LAYER_ELEMS
    _l.check(_l.load().dsvg_gs_pack(flat.data_ptr(), offs.data_ptr(), n_layers, 256, 512, 8, packed_fwd.data_ptr(),
                                    packed_bwd.data_ptr(), _stream()), "dsvg_gs_pack")
    return packed_fwd, packed_bwd


def gs_layer_fwd(x, packed_fwd_layer, in_bias, out_bias, b1, b2, gamma1, beta1, gamma2, beta2, key_mask, n_seq, S, scale,
                 eps=1e-5, drop_p=0.0, site0=0, seed=None, seq_add=None, train=False, seq_base=0, ffn_format=False, into=None):
    """one pre-LN transformer block in one launch (x bf16 [n_seq * S, 256], S <= 32; include/dsvg.h).
    train=False -> x2;  train=True -> (x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h): the tensors the unfused
    launches of the same block save for the backward pass, in the same layouts.
    seq_base: x (key_mask, seq_add, the outputs) are the rows of the sequences seq_base .. of a longer buffer - dropout draws
    are indexed from that buffer's first row.  ffn_format: xn2 / h as ffn_fwd(train=True) hands them over (affine-free rows,
    fragment-ordered hidden columns).  into: the 11 output tensors (row slices of longer buffers) instead of fresh ones."""
    _chk(x, packed_fwd_layer, in_bias, out_bias, b1, b2, gamma1, beta1, gamma2, beta2, key_mask, seed, seq_add)
    rows = n_seq * S
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and tuple(x.shape) == (rows, 256) and 1 <= S <= 32
    assert packed_fwd_layer.numel() == GS_LAYER_ELEMS and packed_fwd_layer.is_contiguous()
    for t, n in ((in_bias, 768), (out_bias, 256), (b1, 512), (b2, 256), (gamma1, 256), (beta1, 256), (gamma2, 256), (beta2, 256)):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    assert key_mask is None or (key_mask.dtype == torch.int64 and key_mask.numel() >= n_seq)
    if seq_add is not None:
        assert seq_add.dtype == x.dtype and tuple(seq_add.shape) == (n_seq, 256)
        assert seq_add.stride(1) == 1 and seq_add.stride(0) % 8 == 0 and seq_add.data_ptr() % 16 == 0
    dev = x.device
    if into is not None:
        assert train and len(into) == 11
        for t, w_ in zip(into, (256, None, None, 256, 768, 256, 256, None, None, 256, 512)):
            assert t.is_contiguous() and t.shape[0] == rows and (t.dtype == torch.float32 if w_ is None else
                                                                 (t.dtype == x.dtype and t.shape[1] == w_))
        x2, sv = into[0], list(into[1:])
    else:
        x2 = torch.empty_like(x)
        sv = [None] * 10
        if train:
            f32 = lambda: torch.empty(rows, dtype=torch.float32, device=dev)
            sv = [f32(), f32(), torch.empty_like(x), torch.empty((rows, 768), dtype=x.dtype, device=dev), torch.empty_like(x),
                  torch.empty_like(x), f32(), f32(), torch.empty_like(x), torch.empty((rows, 512), dtype=x.dtype, device=dev)]
    ev = _prof_begin()
    _l.check(_l.load().dsvg_gs_layer_fwd(x.data_ptr(), packed_fwd_layer.data_ptr(), in_bias.data_ptr(), out_bias.data_ptr(),
                                         b1.data_ptr(), b2.data_ptr(), gamma1.data_ptr(), beta1.data_ptr(),
                                         gamma2.data_ptr(), beta2.data_ptr(), _p(key_mask), _p(seq_add),
                                         int(seq_add.stride(0)) if seq_add is not None else 256, n_seq, S,
                                         x2.data_ptr(), *[_p(t) for t in sv], float(eps), float(scale), float(drop_p),
                                         int(site0), _p(seed) if drop_p > 0 else None, int(seq_base), int(bool(ffn_format)),
                                         _stream()), "dsvg_gs_layer_fwd")
    # algorithmic FLOPs of the block: in_proj + out_proj + attention (2 x 2 S 32 per head and row) + the two FFN products
    _prof_end(ev, 2.0 * rows * 256 * (768 + 256 + 1024) + 4.0 * rows * S * 256, 1024.0 * rows,
              dict(op="gs_layer_fwd", rows=rows, train=bool(train), ffn_flops=4.0 * 256 * 512 * rows))
    if train:
        return (x2, *sv)
    return x2


def gs_layer_bwd(dx2, packed_bwd_layer, x, mean1, rstd1, qkv, x1, mean2, rstd2, h, gamma1, gamma2, key_mask, n_seq, S,
                 scale, drop_p=0.0, site0=0, seed=None, want_dx1=False, dgamma2=None, dbeta2=None, dgamma1=None,
                 dbeta1=None, want_dg=False):
    """backward of gs_layer_fwd with respect to x (include/dsvg.h) -> (dx, dx1 or None, dym, dpre, dx1m, dqkv, dgamma2,
    dbeta2, dgamma1, dbeta1): dym / dpre / dx1m / dqkv are the token-major operands of the four weight-gradient GEMMs.
    want_dg: one more result at the end, dg [n_seq, 256] = bcast_add_bwd(dx1, n_seq, S, drop_p, site0 + 2, seed) bit for bit
    (the per-sequence term's gradient, formed in the same launch)."""
    _chk(dx2, packed_bwd_layer, x, mean1, rstd1, qkv, x1, mean2, rstd2, h, gamma1, gamma2, key_mask, seed)
    rows = n_seq * S
    assert dx2.dtype == torch.bfloat16 and dx2.is_contiguous() and tuple(dx2.shape) == (rows, 256) and 32 % S == 0
    assert packed_bwd_layer.numel() == GS_LAYER_ELEMS and packed_bwd_layer.is_contiguous()
    for t, shape in ((x, (rows, 256)), (x1, (rows, 256)), (qkv, (rows, 768)), (h, (rows, 512))):
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and tuple(t.shape) == shape
    for t in (mean1, rstd1, mean2, rstd2):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == rows
    dev = dx2.device
    dx, dym, dx1m = torch.empty_like(dx2), torch.empty_like(dx2), torch.empty_like(dx2)
    dx1 = torch.empty_like(dx2) if want_dx1 else None
    dpre = torch.empty((rows, 512), dtype=dx2.dtype, device=dev)
    dqkv = torch.empty((rows, 768), dtype=dx2.dtype, device=dev)
    dg = torch.empty((n_seq, 256), dtype=dx2.dtype, device=dev) if want_dg else None
    outs = []
    for t in (dgamma2, dbeta2, dgamma1, dbeta1):
        if t is None:
            t = torch.empty(256, dtype=torch.float32, device=dev)
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 256
        outs.append(t)
    L = _l.load()
    ws = _ws(L.dsvg_gs_bwd_workspace_bytes(n_seq, S), dev)
    ev = _prof_begin()
    _l.check(L.dsvg_gs_layer_bwd(dx2.data_ptr(), packed_bwd_layer.data_ptr(), x.data_ptr(), mean1.data_ptr(),
                                 rstd1.data_ptr(), qkv.data_ptr(), x1.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(),
                                 h.data_ptr(), gamma1.data_ptr(), gamma2.data_ptr(), _p(key_mask), n_seq, S, dx.data_ptr(),
                                 _p(dx1), dym.data_ptr(), dpre.data_ptr(), dx1m.data_ptr(), dqkv.data_ptr(),
                                 *[t.data_ptr() for t in outs], float(scale), float(drop_p), int(site0),
                                 _p(seed) if drop_p > 0 else None, ws.data_ptr(), ws.numel() * 4, _p(dg), _stream()),
             "dsvg_gs_layer_bwd")
    _prof_end(ev, 2.0 * rows * 256 * (768 + 256 + 1024) + 8.0 * rows * S * 256, 1536.0 * rows,
              dict(op="gs_layer_bwd", rows=rows, ffn_flops=4.0 * 256 * 512 * rows))
    return (dx, dx1, dym, dpre, dx1m, dqkv, *outs, dg) if want_dg else (dx, dx1, dym, dpre, dx1m, dqkv, *outs)


GS_STACK_MAX = 4        # layers per dsvg_gs_stack_fwd / dsvg_gs_stack_bwd launch (csrc/group_stage.hip)


def gs_stack_fwd(x, layers, key_mask, n_seq, S, scale, eps=1e-5, drop_p=0.0, seed=None, train=False):
    """a whole stack of pre-LN blocks in ONE launch (include/dsvg.h: dsvg_gs_stack_fwd; x bf16 [n_seq * S, 256]).
    layers: per layer a dict(img=packed_fwd_layer, in_bias, out_bias, b1, b2, gamma1, beta1, gamma2, beta2, site0, seq_add=None);
    every seq_add is a [n_seq, 256] view with the same row stride.  -> per layer what gs_layer_fwd returns: x2 (train=False;
    the inner layers' x2 are not materialised: None) or the tuple (x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h)."""
    n = len(layers)
    assert 1 <= n <= GS_STACK_MAX
    rows = n_seq * S
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and tuple(x.shape) == (rows, 256) and 1 <= S <= 32
    assert key_mask is None or (key_mask.dtype == torch.int64 and key_mask.numel() >= n_seq)
    _chk(x, key_mask, seed)
    dev = x.device
    arr = (_l.GsFwdLayer * n)()
    outs, keep = [], []
    ld = None
    for i, Ld in enumerate(layers):
        img = Ld["img"]
        assert img.numel() == GS_LAYER_ELEMS and img.is_contiguous()
        small = [Ld[k] for k in ("in_bias", "out_bias", "b1", "b2", "gamma1", "beta1", "gamma2", "beta2")]
        for t, nn in zip(small, (768, 256, 512, 256, 256, 256, 256, 256)):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == nn
        sa = Ld.get("seq_add")
        _chk(img, *small, sa)
        if sa is not None:
            assert sa.dtype == x.dtype and tuple(sa.shape) == (n_seq, 256) and sa.stride(1) == 1
            assert sa.stride(0) % 8 == 0 and sa.data_ptr() % 16 == 0 and ld in (None, sa.stride(0))
            ld = sa.stride(0)
        last = i == n - 1
        x2 = torch.empty_like(x) if (train or last) else None
        sv = [None] * 10
        if train:
            f32 = lambda: torch.empty(rows, dtype=torch.float32, device=dev)
            sv = [f32(), f32(), torch.empty_like(x), torch.empty((rows, 768), dtype=x.dtype, device=dev), torch.empty_like(x),
                  torch.empty_like(x), f32(), f32(), torch.empty_like(x), torch.empty((rows, 512), dtype=x.dtype, device=dev)]
        e = arr[i]
        e.packed_fwd_layer = img.data_ptr()
        for k, t in zip(("in_bias", "out_bias", "b1", "b2", "gamma1", "beta1", "gamma2", "beta2"), small):
            setattr(e, k, t.data_ptr())
        e.seq_add = _p(sa)
        e.x2 = _p(x2)
        for k, t in zip(("mean1", "rstd1", "xn1", "qkv", "ao", "x1", "mean2", "rstd2", "xn2", "h"), sv):
            setattr(e, k, _p(t))
        e.site0 = int(Ld["site0"])
        keep.append((img, small, sa))
        outs.append((x2, *sv) if train else x2)
    ev = _prof_begin()
    _l.check(_l.load().dsvg_gs_stack_fwd(x.data_ptr(), C.addressof(arr), n, _p(key_mask), int(ld) if ld is not None else 256,
                                         n_seq, S, float(eps), float(scale), float(drop_p),
                                         _p(seed) if drop_p > 0 else None, _stream()), "dsvg_gs_stack_fwd")
    _prof_end(ev, n * (2.0 * rows * 256 * (768 + 256 + 1024) + 4.0 * rows * S * 256), n * 1024.0 * rows,
              dict(op="gs_stack_fwd", rows=rows, layers=n, train=bool(train), ffn_flops=n * 4.0 * 256 * 512 * rows))
    return outs


def gs_stack_bwd(dx2, layers, key_mask, n_seq, S, scale, drop_p=0.0, seed=None, want_dg=False):
    """backward of gs_stack_fwd with respect to x in ONE launch (include/dsvg.h: dsvg_gs_stack_bwd).  layers (forward order): per
    layer a dict(img=packed_bwd_layer, x, mean1, rstd1, qkv, x1, mean2, rstd2, h, gamma1, gamma2, site0, dgamma2, dbeta2, dgamma1,
    dbeta1) - the last four: fp32 [256] outputs.  -> (dx, per_layer, dgcat): per_layer[i] = (dym, dpre, dx1m, dqkv); dgcat
    (want_dg) = bf16 [n_seq, n * 256], column block i = layer i's conditioning-term gradient (gs_layer_bwd's dg), else None."""
    n = len(layers)
    assert 1 <= n <= GS_STACK_MAX
    rows = n_seq * S
    assert dx2.dtype == torch.bfloat16 and dx2.is_contiguous() and tuple(dx2.shape) == (rows, 256) and 32 % S == 0
    _chk(dx2, key_mask, seed)
    dev = dx2.device
    L = _l.load()
    wsb = L.dsvg_gs_bwd_workspace_bytes(n_seq, S)
    dx = torch.empty_like(dx2)
    dgcat = torch.empty((n_seq, n * 256), dtype=dx2.dtype, device=dev) if want_dg else None
    arr = (_l.GsBwdLayer * n)()
    per, keep = [], []
    for i, Ld in enumerate(layers):
        img = Ld["img"]
        assert img.numel() == GS_LAYER_ELEMS and img.is_contiguous()
        for k, shape in (("x", (rows, 256)), ("x1", (rows, 256)), ("qkv", (rows, 768)), ("h", (rows, 512))):
            t = Ld[k]
            assert t.dtype == torch.bfloat16 and t.is_contiguous() and tuple(t.shape) == shape
        for k in ("mean1", "rstd1", "mean2", "rstd2"):
            t = Ld[k]
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == rows
        for k in ("gamma1", "gamma2", "dgamma2", "dbeta2", "dgamma1", "dbeta1"):
            t = Ld[k]
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 256
        _chk(img, *[Ld[k] for k in ("x", "x1", "qkv", "h", "mean1", "rstd1", "mean2", "rstd2", "gamma1", "gamma2", "dgamma2",
                                    "dbeta2", "dgamma1", "dbeta1")])
        dym, dx1m = torch.empty_like(dx2), torch.empty_like(dx2)
        dpre = torch.empty((rows, 512), dtype=dx2.dtype, device=dev)
        dqkv = torch.empty((rows, 768), dtype=dx2.dtype, device=dev)
        ws = _ws(wsb, dev)
        e = arr[i]
        e.packed_bwd_layer = img.data_ptr()
        for k in ("x", "mean1", "rstd1", "qkv", "x1", "mean2", "rstd2", "h", "gamma1", "gamma2", "dgamma2", "dbeta2", "dgamma1",
                  "dbeta1"):
            setattr(e, k, Ld[k].data_ptr())
        e.dx = dx.data_ptr() if i == 0 else None
        e.dx1 = None
        e.dym, e.dpre, e.dx1m, e.dqkv = dym.data_ptr(), dpre.data_ptr(), dx1m.data_ptr(), dqkv.data_ptr()
        e.dg = (dgcat.data_ptr() + i * 256 * dgcat.element_size()) if want_dg else None
        e.workspace = ws.data_ptr()
        e.site0 = int(Ld["site0"])
        per.append((dym, dpre, dx1m, dqkv))
        keep.append(ws)
    ev = _prof_begin()
    _l.check(L.dsvg_gs_stack_bwd(dx2.data_ptr(), C.addressof(arr), n, _p(key_mask), n_seq, S, float(scale), float(drop_p),
                                 _p(seed) if drop_p > 0 else None, (keep[0].numel() - 1) * 4, n * 256 if want_dg else 256,
                                 _stream()), "dsvg_gs_stack_bwd")
    _prof_end(ev, n * (2.0 * rows * 256 * (768 + 256 + 1024) + 8.0 * rows * S * 256), n * 1536.0 * rows,
              dict(op="gs_stack_bwd", rows=rows, layers=n, ffn_flops=n * 4.0 * 256 * 512 * rows))
    return dx, per, dgcat


def _prof_begin():
    if not (PROFILE_ON and _TAG is not None):
        return None
    ev0 = _event()
    ev0.record()
    return ev0


def _prof_end(ev0, flops, alg_bytes, spec):
    if ev0 is None:
        return
    ev1 = _event()
    ev1.record()
    PROFILE.append((_TAG, ev0, ev1, float(flops), float(alg_bytes), spec))


def keep_scale(p):
    """the factor the kernels scale kept elements by at dropout probability p: 65536 / (65536 - round(65536 p)) - the
    thresholds are 16-bit (csrc/dsvg_common.h drop_make), so this is the exact gradient scale of a dropped activation"""
    if p <= 0:
        return 1.0
    t = min(int(p * 65536.0 + 0.5), 65535)
    return 65536.0 / (65536 - t)


def _ptr_array(tensors):
    """host array of device pointers (ctypes void*[n]) for the C entry points that take pointer lists"""
    import ctypes
    arr = (ctypes.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def latent_chain_fwd(z, weights, biases, train=False):
    """the latent ResNet + the final linear in ONE launch (include/dsvg.h dsvg_latent_chain_fwd): z bf16 [rows, 256]; weights =
    n_res + 1 row-major bf16 [256, 256] matrices (the final linear last), biases fp32 [256] each.
    -> out, or (out, [z_1 .. z_n], [r_1 .. r_n]) with train=True (r_i = relu(W_i z_{i-1} + b_i))"""
    _chk(z, *weights, *biases)
    n_res = len(weights) - 1
    assert z.dtype == torch.bfloat16 and z.dim() == 2 and z.shape[1] == 256 and z.is_contiguous()
    assert len(biases) == len(weights) and 0 <= n_res <= 4
    for w, b in zip(weights, biases):
        assert w.dtype == torch.bfloat16 and tuple(w.shape) == (256, 256) and w.is_contiguous()
        assert b.dtype == torch.float32 and b.numel() == 256 and b.is_contiguous()
    out = torch.empty_like(z)
    zs = [torch.empty_like(z) for _ in range(n_res)] if train else []
    rs = [torch.empty_like(z) for _ in range(n_res)] if train else []
    _l.check(_l.load().dsvg_latent_chain_fwd(z.data_ptr(), _ptr_array(weights), _ptr_array(biases), n_res,
                                             _ptr_array(zs) if train and n_res else None,
                                             _ptr_array(rs) if train and n_res else None, out.data_ptr(), z.shape[0],
                                             _stream()), "dsvg_latent_chain_fwd")
    return (out, zs, rs) if train else out


def latent_chain_bwd(dout, weights, rs):
    """backward of latent_chain_fwd's input path -> (dz0, [dpre_1 .. dpre_n]); dpre_i = dz_i where r_i > 0"""
    _chk(dout, *weights, *rs)
    n_res = len(weights) - 1
    assert dout.dtype == torch.bfloat16 and dout.dim() == 2 and dout.shape[1] == 256 and dout.is_contiguous()
    assert len(rs) == n_res and all(r.shape == dout.shape and r.dtype == dout.dtype and r.is_contiguous() for r in rs)
    dz0 = torch.empty_like(dout)
    dpre = [torch.empty_like(dout) for _ in range(n_res)]
    _l.check(_l.load().dsvg_latent_chain_bwd(dout.data_ptr(), _ptr_array(weights), _ptr_array(rs) if n_res else None, n_res,
                                             _ptr_array(dpre) if n_res else None, dz0.data_ptr(), dout.shape[0], _stream()),
             "dsvg_latent_chain_bwd")
    return dz0, dpre


def gate_mul(dy, y, scale=1.0):
    """out = dy * scale where y > 0 else 0   (backward of relu [+ dropout] from the saved output)"""
    _chk(dy, y)
    assert dy.is_contiguous() and y.is_contiguous() and dy.shape == y.shape and dy.dtype == y.dtype
    out = torch.empty_like(dy)
    _l.check(_l.load().dsvg_gate_mul(_dt(dy), dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), float(scale),
                                     _stream()), "dsvg_gate_mul")
    return out


def drop_apply(x, drop_p, drop_site, seed):
    """y = x * dropmask(seed, site, flat index); returns x itself when drop_p == 0"""
    if drop_p <= 0:
        return x
    _chk(x, seed)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    ev = _prof_begin()
    _l.check(_l.load().dsvg_drop_apply(_dt(x), x.data_ptr(), y.data_ptr(), x.numel(), float(drop_p), int(drop_site),
                                       seed.data_ptr(), _stream()), "dsvg_drop_apply")
    _prof_end(ev, 0.0, 0.0, dict(op="drop_apply"))
    return y


def add(a, b):
    _chk(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype
    out = torch.empty_like(a)
    _l.check(_l.load().dsvg_add(_dt(a), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "dsvg_add")
    return out


def copy_many(pairs):
    """dst.copy_(src) for every (dst, src) pair of same-sized contiguous device tensors, in one launch per 32 pairs"""
    pairs = [(d, s) for d, s in pairs if d.numel() > 0]
    if not pairs:
        return
    n = len(pairs)
    for d, s in pairs:
        _chk(d, s)
        assert d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype and d.numel() == s.numel(), \
            (d.shape, s.shape, d.dtype, s.dtype)
    src = (C.c_void_p * n)(*[s.data_ptr() for _, s in pairs])
    dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    nb = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in pairs])
    _l.check(_l.load().dsvg_copy_many(src, dst, nb, n, _stream()), "dsvg_copy_many")


def advance_step_(counter, seed):
    _chk(counter, seed)
    _l.check(_l.load().dsvg_advance_step(_p(counter), _p(seed), _stream()), "dsvg_advance_step")


def pack_images(flat, flat_lp, ffn=None, attn=None, gs=None, attn_bwd=True, counter=None, seed=None):
    """every per-step weight image of a bf16 model in ONE launch (dsvg_pack_images): the bf16 copy of the flat buffer
    (cast_weights), ffn = dict(offs, n, fwd, bwd, b1f, w2p) (ffn_pack), attn = dict(offs, n, img, bwd) (attn_pack and, with
    attn_bwd, attn_pack_bwd), gs = dict(offs, n, fwd, bwd) (gs_pack) - the stores' own image tables - and, when given, the
    step counter / dropout seed advance (advance_step_).  Bit-identical to those launches."""
    _chk(flat, flat_lp, counter, seed)
    assert flat.dtype == torch.float32 and flat_lp.dtype == torch.bfloat16 and flat.numel() == flat_lp.numel()
    assert flat.is_contiguous() and flat_lp.is_contiguous()
    f, a, g = ffn or {}, attn or {}, gs or {}
    _l.check(_l.load().dsvg_pack_images(
        flat.data_ptr(), flat_lp.data_ptr(), flat.numel(),
        _p(f.get("offs")), f.get("n", 0), _p(f.get("fwd")), _p(f.get("bwd")), _p(f.get("b1f")), _p(f.get("w2p")),
        _p(a.get("offs")), a.get("n", 0), _p(a.get("img")), _p(a.get("bwd")) if attn_bwd else None,
        _p(g.get("offs")), g.get("n", 0), _p(g.get("fwd")), _p(g.get("bwd")),
        _p(counter), _p(seed), _stream()), "dsvg_pack_images")


def pack_images_ok(flat, flat_lp):
    """the one-launch refresh needs 16-byte aligned flat buffers with a multiple of 8 elements"""
    return (flat.is_cuda and flat_lp is not None and flat_lp.dtype == torch.bfloat16 and flat.numel() % 8 == 0
            and flat.data_ptr() % 16 == 0 and flat_lp.data_ptr() % 16 == 0)


def probe_trread(off):
    _chk(off)
    assert off.dtype == torch.int32 and off.numel() == 64
    out = torch.empty(256, dtype=torch.int16, device=off.device)
    _l.check(_l.load().dsvg_probe_trread(off.data_ptr(), out.data_ptr(), _stream()), "dsvg_probe_trread")
    return out
