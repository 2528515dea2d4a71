"""The fused per-layer kernels of the short-sequence ("group") stages (csrc/group_stage.hip) against
  * the fp32 torch restatement of the same block (tests/torch_ops_ref.py: the unfused reference ops composed, with the bf16
    rounding points of the unfused launches), tensor by tensor - every saved tensor of the forward pass, every operand the
    backward pass hands to the weight-gradient GEMMs, the LayerNorm parameter gradients;
  * the unfused HIP launches they replace (same dropout draws: fused forward + unfused backward and the reverse must agree).
Each check collects ALL mismatching tensors of a case before failing, so that one GPU run localises a defect to a phase."""
import pytest
import torch

from deepsvg_amd import ops
from tests import torch_ops_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
N_W = 196608 + 65536 + 131072 + 131072       # in_proj | out_proj | linear1 | linear2


def _seed_tensor(v=0x1234567887654321):
    return torch.tensor([v if v < (1 << 63) else v - (1 << 64)], dtype=torch.int64, device=DEV)


def _setup(n_seq, S, seed=0, n_layers=2, masked=False, with_add=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    flat = torch.zeros(8 + n_layers * N_W)
    offs = []
    for i in range(n_layers):
        o = 8 + i * N_W
        ent = [o, o + 196608, o + 262144, o + 393216]
        flat[o:o + N_W] = torch.randn(N_W, generator=g) * 0.06
        offs.append(ent)
    rows = n_seq * S
    p = dict(
        in_bias=0.2 * torch.randn(768, generator=g), out_bias=0.2 * torch.randn(256, generator=g),
        b1=0.3 * torch.randn(512, generator=g), b2=0.3 * torch.randn(256, generator=g),
        gamma1=1 + 0.2 * torch.randn(256, generator=g), beta1=0.2 * torch.randn(256, generator=g),
        gamma2=1 + 0.2 * torch.randn(256, generator=g), beta2=0.2 * torch.randn(256, generator=g))
    p = {k: v.to(DEV) for k, v in p.items()}
    x = (torch.randn(rows, 256, generator=g) * 1.5 + 0.3).to(DEV).to(torch.bfloat16)
    key_mask = None
    if masked:      # arbitrary visibility masks with at least one visible key per sequence
        bits = torch.randint(0, 2, (n_seq, S), generator=g)
        bits[torch.arange(n_seq), torch.randint(0, S, (n_seq,), generator=g)] = 1
        key_mask = (bits.to(torch.int64) << torch.arange(S, dtype=torch.int64)).sum(1).to(DEV)
    seq_add = (torch.randn(n_seq, 256, generator=g) * 0.5).to(DEV).to(torch.bfloat16) if with_add else None
    dx2 = (torch.randn(rows, 256, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    return flat.to(DEV), torch.tensor(offs, dtype=torch.int64, device=DEV), p, x, key_mask, seq_add, dx2


def _diff(name, got, want, tol, bad, mean_tol=None):
    got, want = got.float(), want.float()
    if got.shape != want.shape:
        bad.append(f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}")
        return
    if not torch.isfinite(got).all():
        bad.append(f"{name}: {int((~torch.isfinite(got)).sum())} non-finite values")
        return
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs()
    if err.max().item() > tol * scale:
        i = int(err.argmax())
        r, c = (i // got.shape[-1], i % got.shape[-1]) if got.dim() == 2 else (i, 0)
        frac = (err > tol * scale).float().mean().item()
        bad.append(f"{name}: max err {err.max().item():.3e} at (row {r}, col {c}) vs scale {scale:.3e} (tol {tol:.0e}); "
                   f"{100 * frac:.2f} % of the elements out of tolerance")
    elif mean_tol is not None and err.mean().item() > mean_tol * want.abs().mean().item():
        bad.append(f"{name}: mean err {err.mean().item():.3e} vs mean |want| {want.abs().mean().item():.3e}")


FWD_NAMES = ("x2", "mean1", "rstd1", "xn1", "qkv", "ao", "x1", "mean2", "rstd2", "xn2", "h")
BWD_NAMES = ("dx", "dx1", "dym", "dpre", "dx1m", "dqkv", "dgamma2", "dbeta2", "dgamma1", "dbeta1")


def _params(p):
    return (p["in_bias"], p["out_bias"], p["b1"], p["b2"], p["gamma1"], p["beta1"], p["gamma2"], p["beta2"])


def test_gs_pack_layout(gpu_device):
    """every fragment of both images against the index formulas in the header of csrc/group_stage.hip"""
    flat, offs, *_ = _setup(4, 8, seed=3)
    pf, pb = ops.gs_pack(flat, offs, 2)
    pf = pf.view(2, 8, 128, 64, 8).cpu().float()
    pb = pb.view(2, 8, 128, 64, 8).cpu().float()
    lane = torch.arange(64)
    row, half = (lane & 31).view(64, 1), (lane >> 5).view(64, 1)
    e = torch.arange(8).view(1, 8)
    for layer in range(2):
        oi, oo, o1, o2 = (int(v) for v in offs[layer])
        bf = lambda t: t.to(torch.bfloat16).float().cpu()
        Win, Wo = bf(flat[oi:oi + 196608].view(768, 256)), bf(flat[oo:oo + 65536].view(256, 256))
        W1, W2 = bf(flat[o1:o1 + 131072].view(512, 256)), bf(flat[o2:o2 + 131072].view(256, 512))
        for w in range(8):
            for i in range(128):
                k = lambda ks: 16 * ks + 8 * half + e
                if i < 48:
                    want = Win[256 * (i % 3) + 32 * w + row, k(i // 3)]
                elif i < 64:
                    want = Wo[32 * w + row, k(i - 48)]
                elif i < 96:
                    want = W1[64 * w + 32 * (i & 1) + row, k((i - 64) >> 1)]
                else:
                    want = W2[32 * w + row, k(i - 96)]
                assert torch.equal(pf[layer, w, i], want), f"forward image: layer {layer} wave {w} fragment {i}"
                if i < 32:
                    want = W2[k(i >> 1), 64 * w + 32 * (i & 1) + row]
                elif i < 64:
                    want = W1[k(i - 32), 32 * w + row]
                elif i < 80:
                    want = Wo[k(i - 64), 32 * w + row]
                else:
                    want = Win[k(i - 80), 32 * w + row]
                assert torch.equal(pb[layer, w, i], want), f"backward image: layer {layer} wave {w} fragment {i}"


CASES = [  # n_seq, S, key masks, per-sequence add, dropout
    (512, 8, True, False, 0.1),      # encoder group stage at the benchmark size
    (512, 8, False, True, 0.1),      # decoder group stage
    (37, 8, True, True, 0.0),        # ragged last tile (one sequence in it), no dropout
    (5, 32, False, False, 0.1),      # one sequence per tile
    (9, 16, True, True, 0.1),        # two per tile, ragged
    (3, 4, False, True, 0.0),        # a single partly filled tile
]


@pytest.mark.parametrize("n_seq,S,masked,with_add,drop_p", CASES)
def test_gs_layer_fwd_matches_reference(gpu_device, n_seq, S, masked, with_add, drop_p):
    flat, offs, p, x, key_mask, seq_add, _ = _setup(n_seq, S, seed=n_seq + S, masked=masked, with_add=with_add)
    pf, _pb = ops.gs_pack(flat, offs, 2)
    ef, _eb = R.gs_pack(flat, offs, 2)
    seed = _seed_tensor(0x0123456789ABCDEF)
    scale = 32 ** -0.5
    bad = []
    for layer in (0, 1):
        sl = slice(layer * ops.GS_LAYER_ELEMS, (layer + 1) * ops.GS_LAYER_ELEMS)
        got = ops.gs_layer_fwd(x, pf[sl], *_params(p), key_mask, n_seq, S, scale, 1e-5, drop_p, 300 + 8 * layer, seed,
                               seq_add=seq_add, train=True)
        want = R.gs_layer_fwd(x, ef[sl], *_params(p), key_mask, n_seq, S, scale, 1e-5, drop_p, 300 + 8 * layer, seed,
                              seq_add=seq_add, train=True)
        torch.cuda.synchronize()
        for name, a, b in zip(FWD_NAMES, got, want):
            if name == "h":     # ReLU gates within rounding of zero may differ (other summation order): compare where both agree
                same = (a != 0) == (b != 0)
                if (~same).float().mean().item() > 3e-3:
                    bad.append(f"layer {layer} h: {100 * (~same).float().mean().item():.2f} % of the gates differ")
                a, b = torch.where(same, a, torch.zeros_like(a)), torch.where(same, b, torch.zeros_like(b))
            # (mean2 / rstd2 are statistics of x1, which carries bf16 rounding differences of its own: ~1e-3)
            tol = 1e-4 if name in ("mean1", "rstd1") else (4e-3 if name in ("mean2", "rstd2") else 2e-2)
            _diff(f"layer {layer} {name}", a, b, tol, bad, mean_tol=None if name.startswith(("mean", "rstd")) else 4e-3)
        # the inference call: the same x2, nothing else written
        y = ops.gs_layer_fwd(x, pf[sl], *_params(p), key_mask, n_seq, S, scale, 1e-5, drop_p, 300 + 8 * layer, seed,
                             seq_add=seq_add)
        if not torch.equal(y, got[0]):
            bad.append(f"layer {layer}: inference x2 differs from the training call's x2")
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("n_total,n_tail,S,with_add,drop_p", [(300, 257, 31, True, 0.1), (40, 13, 31, False, 0.0),
                                                              (90, 33, 10, True, 0.1), (64, 64, 31, True, 0.1)])
def test_gs_layer_fwd_on_the_tail_sequences_of_a_longer_buffer(gpu_device, n_total, n_tail, S, with_add, drop_p):
    """the forward kernel on sequences of any length <= 32 (here 31: one sequence and one padding row per tile) and on the
    LAST n_tail sequences of a longer buffer (seq_base = n_total - n_tail: pointers at their first row, dropout draws
    indexed from the buffer's first row) - against the restatement run on the whole buffer; then with the FFN half's
    training outputs in the fused FFN kernels' format (affine-free LayerNorm rows, fragment-ordered hidden columns), written
    into row slices of preallocated buffers"""
    flat, offs, p, x, _km, seq_add, _ = _setup(n_total, S, seed=n_total + S, with_add=with_add)
    pf, _pb = ops.gs_pack(flat, offs, 2)
    ef, _eb = R.gs_pack(flat, offs, 2)
    seed = _seed_tensor(0x0123456789ABCDE1)
    scale = 32 ** -0.5
    base = n_total - n_tail
    r0 = base * S
    sl = slice(0, ops.GS_LAYER_ELEMS)
    want = R.gs_layer_fwd(x, ef[sl], *_params(p), None, n_total, S, scale, 1e-5, drop_p, 400, seed, seq_add=seq_add, train=True)
    sa = seq_add[base:].contiguous() if seq_add is not None else None
    got = ops.gs_layer_fwd(x[r0:], pf[sl], *_params(p), None, n_tail, S, scale, 1e-5, drop_p, 400, seed, seq_add=sa,
                           train=True, seq_base=base)
    torch.cuda.synchronize()
    bad = []

    def check(got_t, what):
        for name, a, b in zip(FWD_NAMES, got_t, want):
            b = b[r0:]
            if name == "h":
                same = (a != 0) == (b != 0)
                if (~same).float().mean().item() > 3e-3:
                    bad.append(f"{what} h: {100 * (~same).float().mean().item():.2f} % of the gates differ")
                a, b = torch.where(same, a, torch.zeros_like(a)), torch.where(same, b, torch.zeros_like(b))
            tol = 1e-4 if name in ("mean1", "rstd1") else (4e-3 if name in ("mean2", "rstd2") else 2e-2)
            _diff(f"{what} {name}", a, b, tol, bad, mean_tol=None if name.startswith(("mean", "rstd")) else 4e-3)
    check(got, "tail")
    if base:        # the draws really are those of the longer buffer: the same call with seq_base = 0 gives other masks
        other = ops.gs_layer_fwd(x[r0:], pf[sl], *_params(p), None, n_tail, S, scale, 1e-5, drop_p, 400, seed, seq_add=sa,
                                 train=True)
        assert (drop_p == 0) == torch.equal(other[0], got[0])
    # fused-FFN format, into row slices of buffers that cover the whole batch
    rows = n_total * S
    bf = lambda w_: torch.full((rows, w_), 7.0, dtype=torch.bfloat16, device=DEV)
    f32 = lambda: torch.full((rows,), 7.0, dtype=torch.float32, device=DEV)
    full = [bf(256), f32(), f32(), bf(256), bf(768), bf(256), bf(256), f32(), f32(), bf(256), bf(512)]
    res = ops.gs_layer_fwd(x[r0:], pf[sl], *_params(p), None, n_tail, S, scale, 1e-5, drop_p, 400, seed, seq_add=sa,
                           train=True, seq_base=base, ffn_format=True, into=tuple(t[r0:] for t in full))
    torch.cuda.synchronize()
    for t in full:
        assert bool((t[:r0] == 7.0).all()), "rows in front of the slice were written"
    for i, name in enumerate(FWD_NAMES):
        if name not in ("xn2", "h"):
            assert torch.equal(res[i], got[i]), name
    xh = ((got[6].float() - got[7].unsqueeze(1)) * got[8].unsqueeze(1))
    _diff("ffn_format xn2 = (x1 - mean2) rstd2", res[9], xh, 8e-3, bad)
    perm = R._ffn_frag_perm(DEV)
    assert torch.equal(res[10][:, perm], got[10]), "h is not the fragment-order permutation of the natural-order h"
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("n_seq,S,masked,with_add,drop_p", CASES)
def test_gs_layer_bwd_matches_reference(gpu_device, n_seq, S, masked, with_add, drop_p):
    flat, offs, p, x, key_mask, seq_add, dx2 = _setup(n_seq, S, seed=100 + n_seq + S, masked=masked, with_add=with_add)
    _pf, pb = ops.gs_pack(flat, offs, 2)
    ef, eb = R.gs_pack(flat, offs, 2)
    seed = _seed_tensor(0x00C0FFEE12345678)
    scale = 32 ** -0.5
    bad = []
    for layer in (0, 1):
        sl = slice(layer * ops.GS_LAYER_ELEMS, (layer + 1) * ops.GS_LAYER_ELEMS)
        # inputs of the backward pass from the REFERENCE forward: this test does not depend on the fused forward kernel
        (_x2, mean1, rstd1, _xn1, qkv, _ao, x1, mean2, rstd2, _xn2, h) = R.gs_layer_fwd(
            x, ef[sl], *_params(p), key_mask, n_seq, S, scale, 1e-5, drop_p, 300 + 8 * layer, seed, seq_add=seq_add, train=True)
        args = (x, mean1, rstd1, qkv, x1, mean2, rstd2, h, p["gamma1"], p["gamma2"], key_mask, n_seq, S, scale, drop_p,
                300 + 8 * layer, seed)
        got = ops.gs_layer_bwd(dx2, pb[sl], *args, want_dx1=True)
        want = R.gs_layer_bwd(dx2, eb[sl], *args, want_dx1=True)
        # round 5: the per-sequence term's gradient from the same launch = the bcast_add_bwd launch on the kernel's own dx1, bit
        # for bit (same summation order, same draws), with and without the dx1 store; nothing else changes
        for want_dx1 in (True, False):
            both = ops.gs_layer_bwd(dx2, pb[sl], *args, want_dx1=want_dx1, want_dg=True)
            assert len(both) == len(got) + 1 and (both[1] is None) == (not want_dx1)
            assert torch.equal(both[-1], ops.bcast_add_bwd(got[1], n_seq, S, drop_p, 300 + 8 * layer + 2, seed)), "dg"
            for name, a, b in zip(BWD_NAMES, both, got):
                assert (a is None and name == "dx1") or torch.equal(a, b), name
        torch.cuda.synchronize()
        for name, a, b in zip(BWD_NAMES, got, want):
            _diff(f"layer {layer} {name}", a, b, 2e-2 if not name.startswith(("dgamma", "dbeta")) else 1e-2, bad,
                  mean_tol=5e-3 if not name.startswith(("dgamma", "dbeta")) else None)
    assert not bad, "\n".join(bad)


def _unfused_fwd(x, W, p, key_mask, n_seq, S, scale, drop_p, s0, seed, seq_add):
    """the launches LayerFn.forward issues for a small stage (functional.py), same order"""
    win, wo, w1, w2 = W
    xn1, mean1, rstd1 = ops.layernorm_fwd(x, p["gamma1"], p["beta1"])
    qkv = ops.gemm(xn1, win, bias=p["in_bias"])
    ao = ops.attention_fwd(qkv, key_mask, n_seq, S, 8, scale, drop_p, s0, seed)
    x1 = ops.gemm(ao, wo, bias=p["out_bias"], res=x, drop_p=drop_p, drop_site=s0 + 1, seed=seed)
    if seq_add is not None:
        ops.bcast_add_fwd_(x1, seq_add, n_seq, S, drop_p, s0 + 2, seed)
    xn2, mean2, rstd2 = ops.layernorm_fwd(x1, p["gamma2"], p["beta2"])
    h = ops.gemm(xn2, w1, bias=p["b1"], act=ops.RELU, drop_p=drop_p, drop_site=s0 + 3, seed=seed)
    x2 = ops.gemm(h, w2, bias=p["b2"], res=x1, drop_p=drop_p, drop_site=s0 + 4, seed=seed)
    return x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h


def _unfused_bwd(dx2, W, p, sv, key_mask, n_seq, S, scale, drop_p, s0, seed):
    """the input-gradient chain of LayerFn.backward's unfused branch"""
    win, wo, w1, w2 = W
    (_x2, mean1, rstd1, _xn1, qkv, _ao, x1, mean2, rstd2, _xn2, h), x = sv
    inv_keep = ops.keep_scale(drop_p)
    dym = ops.drop_apply(dx2, drop_p, s0 + 4, seed)
    dpre = ops.gemm(dym, w2, b_kc=False, gate=h, gate_scale=inv_keep)
    dxn2 = ops.gemm(dpre, w1, b_kc=False)
    dx1, dg2, db2 = ops.layernorm_bwd(dxn2, x1, mean2, rstd2, p["gamma2"], res=dx2)
    dx1m = ops.drop_apply(dx1, drop_p, s0 + 1, seed)
    dao = ops.gemm(dx1m, wo, b_kc=False)
    dqkv = ops.attention_bwd(qkv, key_mask, dao, n_seq, S, 8, scale, drop_p, s0, seed)
    dxn1 = ops.gemm(dqkv, win, b_kc=False)
    dx, dg1, db1 = ops.layernorm_bwd(dxn1, x, mean1, rstd1, p["gamma1"], res=dx1)
    return dx, dx1, dym, dpre, dx1m, dqkv, dg2, db2, dg1, db1


@pytest.mark.parametrize("n_seq,S,masked,with_add", [(512, 8, True, False), (130, 8, False, True), (6, 32, False, True)])
def test_gs_layer_is_interchangeable_with_the_unfused_launches(gpu_device, n_seq, S, masked, with_add):
    """dropout 0.1 at all five sites: the fused kernels draw the same masks as the launches they replace, so every tensor
    of (fused forward, fused backward) agrees with (unfused forward, unfused backward), and a fused backward pass fed with
    the UNFUSED forward's saved tensors (and the reverse) reproduces the same input gradient"""
    drop_p = 0.1
    flat, offs, p, x, key_mask, seq_add, dx2 = _setup(n_seq, S, seed=7 + n_seq, n_layers=1, masked=masked, with_add=with_add)
    pf, pb = ops.gs_pack(flat, offs, 1)
    oi, oo, o1, o2 = (int(v) for v in offs[0])
    bf = lambda t: t.to(torch.bfloat16).contiguous()
    W = (bf(flat[oi:oi + 196608].view(768, 256)), bf(flat[oo:oo + 65536].view(256, 256)),
         bf(flat[o1:o1 + 131072].view(512, 256)), bf(flat[o2:o2 + 131072].view(256, 512)))
    seed = _seed_tensor(0x0F1E2D3C4B5A6978)
    scale, s0 = 32 ** -0.5, 208
    bad = []
    fu = ops.gs_layer_fwd(x, pf, *_params(p), key_mask, n_seq, S, scale, 1e-5, drop_p, s0, seed, seq_add=seq_add, train=True)
    if seq_add is not None:
        # conditioning rows as a column block of a wider tensor (row stride 1024, GlobalCondFn's layout): bit-identical
        wide = torch.zeros(seq_add.shape[0], 1024, dtype=seq_add.dtype, device=seq_add.device)
        wide[:, 256:512] = seq_add
        fs = ops.gs_layer_fwd(x, pf, *_params(p), key_mask, n_seq, S, scale, 1e-5, drop_p, s0, seed, seq_add=wide[:, 256:512],
                              train=True)
        assert all(torch.equal(a, b) for a, b in zip(fs, fu) if torch.is_tensor(a)), "strided conditioning rows"
    un = _unfused_fwd(x, W, p, key_mask, n_seq, S, scale, drop_p, s0, seed, seq_add)
    for name, a, b in zip(FWD_NAMES, fu, un):
        if name == "h":
            same = (a != 0) == (b != 0)
            if (~same).float().mean().item() > 3e-3:
                bad.append(f"h: {100 * (~same).float().mean().item():.2f} % of the gates differ between fused and unfused")
            a, b = torch.where(same, a, torch.zeros_like(a)), torch.where(same, b, torch.zeros_like(b))
        tol = 1e-4 if name in ("mean1", "rstd1") else (4e-3 if name in ("mean2", "rstd2") else 2e-2)
        _diff(f"forward {name} (fused vs unfused)", a, b, tol, bad)

    def fused_bwd(sv):
        (_x2, mean1, rstd1, _xn1, qkv, _ao, x1, mean2, rstd2, _xn2, h) = sv
        return ops.gs_layer_bwd(dx2, pb, x, mean1, rstd1, qkv, x1, mean2, rstd2, h, p["gamma1"], p["gamma2"], key_mask, n_seq,
                                S, scale, drop_p, s0, seed, want_dx1=True)
    # the same saved tensors through both backward implementations: tight agreement, whichever forward produced them (the
    # two forwards themselves differ where a pre-activation is within rounding of zero - 0.06 % of the ReLU gates - so their
    # backward passes are compared with each other only through the forward tensors above)
    for label, sv in (("fused forward", fu), ("unfused forward", un)):
        fb = fused_bwd(sv)
        ub = _unfused_bwd(dx2, W, p, (sv, x), key_mask, n_seq, S, scale, drop_p, s0, seed)
        torch.cuda.synchronize()
        for name, a, b in zip(BWD_NAMES, fb, ub):
            tol = 2.5e-2 if not name.startswith(("dgamma", "dbeta")) else 1e-2
            _diff(f"backward {name} ({label}: fused vs unfused backward)", a, b, tol, bad)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("rows", [512, 37, 1000, 1, 4096])
@pytest.mark.parametrize("n_res", [4, 0, 1])
def test_latent_chain_kernels_match_the_unfused_launches(gpu_device, rows, n_res):
    """dsvg_latent_chain_fwd / bwd (the latent ResNet + the bottleneck linear, one launch per direction) against the launches
    they replace - GEMM + add per block forward; gate, input-gradient GEMM per block backward - and against the fp32 torch
    restatement of the same chain (tests/torch_ops_ref.py)"""
    from deepsvg_amd import ops
    import tests.torch_ops_ref as R
    g = torch.Generator().manual_seed(rows + n_res)
    dev = gpu_device
    z = (torch.randn(rows, 256, generator=g) * 0.8).to(dev).to(torch.bfloat16)
    ws = [(torch.randn(256, 256, generator=g) * 0.06).to(dev).to(torch.bfloat16) for _ in range(n_res + 1)]
    bs = [(torch.randn(256, generator=g) * 0.1).to(dev) for _ in range(n_res + 1)]
    dout = torch.randn(rows, 256, generator=g).to(dev).to(torch.bfloat16)
    out, zs, rs = ops.latent_chain_fwd(z, ws, bs, train=True) if n_res else (ops.latent_chain_fwd(z, ws, bs), [], [])
    assert torch.equal(out, ops.latent_chain_fwd(z, ws, bs)), "inference and training variants differ"
    # the unfused launches
    cur, zs_u, rs_u = z, [], []
    for i in range(n_res):
        r = ops.gemm(cur, ws[i], bias=bs[i], act=ops.RELU)
        cur = ops.add(cur, r)
        zs_u.append(cur)
        rs_u.append(r)
    out_u = ops.gemm(cur, ws[n_res], bias=bs[n_res])
    tol = dict(rtol=2e-2, atol=2e-2)
    for a_, b_, what in [(out, out_u, "out")] + [(x, y, f"z{i + 1}") for i, (x, y) in enumerate(zip(zs, zs_u))] + \
            [(x, y, f"r{i + 1}") for i, (x, y) in enumerate(zip(rs, rs_u))]:
        assert torch.allclose(a_.float(), b_.float(), **tol), (what, (a_.float() - b_.float()).abs().max().item())
    ro, rzs, rrs = R.latent_chain_fwd(z.cpu(), [w.cpu() for w in ws], [b.cpu() for b in bs], train=True)
    assert torch.allclose(out.float().cpu(), ro.float(), **tol)
    # backward on the FUSED forward's r (gates must be the same for a meaningful comparison)
    dz0, dpre = ops.latent_chain_bwd(dout, ws, rs)
    gcur = ops.gemm(dout, ws[n_res], b_kc=False)
    dpre_u = [None] * n_res
    for i in range(n_res - 1, -1, -1):
        dp = ops.gate_mul(gcur, rs[i], 1.0)
        dpre_u[i] = dp
        gcur = ops.gemm(dp, ws[i], b_kc=False, res=gcur)
    assert torch.allclose(dz0.float(), gcur.float(), **tol), (dz0.float() - gcur.float()).abs().max().item()
    for i in range(n_res):
        assert torch.allclose(dpre[i].float(), dpre_u[i].float(), **tol), (i, (dpre[i].float() - dpre_u[i].float()).abs().max().item())
        assert torch.equal(dpre[i] == 0, (rs[i] <= 0) | (dpre[i] == 0))
    rdz, rdp = R.latent_chain_bwd(dout.cpu(), [w.cpu() for w in ws], [r.cpu() for r in rs])
    assert torch.allclose(dz0.float().cpu(), rdz.float(), **tol)
    same = [torch.equal(out, out_u), torch.equal(dz0, gcur)]
    print(f"latent chain rows {rows} n_res {n_res}: bit-identical to the unfused launches (out, dz0): {same}")


def _stack_setup(n_seq, S, n_layers, masked, with_add, seed):
    """per-layer parameters (different for every layer), packed images, inputs"""
    flat, offs, _p, x, key_mask, _sa, dx2 = _setup(n_seq, S, seed=seed, n_layers=n_layers, masked=masked)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    ps = []
    for _i in range(n_layers):
        p = dict(
            in_bias=0.2 * torch.randn(768, generator=g), out_bias=0.2 * torch.randn(256, generator=g),
            b1=0.3 * torch.randn(512, generator=g), b2=0.3 * torch.randn(256, generator=g),
            gamma1=1 + 0.2 * torch.randn(256, generator=g), beta1=0.2 * torch.randn(256, generator=g),
            gamma2=1 + 0.2 * torch.randn(256, generator=g), beta2=0.2 * torch.randn(256, generator=g))
        ps.append({k: v.to(DEV) for k, v in p.items()})
    gcat = (torch.randn(n_seq, n_layers * 256, generator=g) * 0.5).to(DEV).to(torch.bfloat16) if with_add else None
    return flat, offs, ps, x, key_mask, gcat, dx2


STACK_CASES = [
    # n_seq, S, layers, key masks, conditioning rows, dropout
    (512, 8, 4, True, False, 0.1),       # hierarchical_encoder at batch 512 (deepsvg/model/model.py:153-161)
    (512, 8, 4, False, True, 0.1),       # hierarchical_decoder: conditioning rows as column blocks of one product
    (509, 8, 4, True, True, 0.1),        # a last tile with one sequence
    (130, 8, 3, False, True, 0.0),
    (6, 32, 2, False, False, 0.1),       # one sequence per tile
    (3, 16, 1, True, True, 0.1),
]


@pytest.mark.parametrize("n_seq,S,n_layers,masked,with_add,drop_p", STACK_CASES)
def test_gs_stack_launch_is_bit_identical_to_the_layer_launches(gpu_device, n_seq, S, n_layers, masked, with_add, drop_p):
    """dsvg_gs_stack_fwd / dsvg_gs_stack_bwd (round 6: one launch per stack and direction, the rows stay on chip between the
    layers) against the same layers launched one by one: every stored tensor of the forward pass, every operand of the
    weight-gradient products, the LayerNorm parameter gradients, the conditioning rows' gradients and dx - bit for bit."""
    flat, offs, ps, x, key_mask, gcat, dx2 = _stack_setup(n_seq, S, n_layers, masked, with_add, seed=40 + n_seq + n_layers)
    pf, pb = ops.gs_pack(flat, offs, n_layers)
    seed = _seed_tensor(0x0BADC0DE0DDBA11 + n_seq)
    scale, s0 = (256 // 8) ** -0.5, 520
    E = ops.GS_LAYER_ELEMS
    sa = lambda i: None if gcat is None else gcat[:, 256 * i:256 * (i + 1)]
    # ---- forward: layer by layer
    want, cur = [], x
    for i, p in enumerate(ps):
        r = ops.gs_layer_fwd(cur, pf[i * E:(i + 1) * E], *_params(p), key_mask, n_seq, S, scale, 1e-5, drop_p, s0 + 8 * i, seed,
                             seq_add=sa(i), train=True)
        want.append(r)
        cur = r[0]
    layers = [dict(img=pf[i * E:(i + 1) * E], site0=s0 + 8 * i, seq_add=sa(i), **p) for i, p in enumerate(ps)]
    got = ops.gs_stack_fwd(x, layers, key_mask, n_seq, S, scale, 1e-5, drop_p, seed, train=True)
    torch.cuda.synchronize()
    bad = []
    for i in range(n_layers):
        for name, a, b in zip(FWD_NAMES, got[i], want[i]):
            if not torch.equal(a, b):
                bad.append(f"forward layer {i} {name}: {int((a != b).sum())} elements differ")
    inf = ops.gs_stack_fwd(x, layers, key_mask, n_seq, S, scale, 1e-5, drop_p, seed, train=False)
    assert all(t is None for t in inf[:-1])
    if not torch.equal(inf[-1], want[-1][0]):
        bad.append("inference stack launch: x2 differs")
    assert not bad, "\n".join(bad)
    # ---- backward: layer by layer from the last one
    def grads():
        return {k: torch.full((256,), float("nan"), device=DEV) for k in ("dgamma2", "dbeta2", "dgamma1", "dbeta1")}
    wl, g = [None] * n_layers, dx2
    for i in range(n_layers - 1, -1, -1):
        (_x2, mean1, rstd1, _xn1, qkv, _ao, x1, mean2, rstd2, _xn2, h) = want[i]
        xin = x if i == 0 else want[i - 1][0]
        wl[i] = ops.gs_layer_bwd(g, pb[i * E:(i + 1) * E], xin, mean1, rstd1, qkv, x1, mean2, rstd2, h, ps[i]["gamma1"],
                                 ps[i]["gamma2"], key_mask, n_seq, S, scale, drop_p, s0 + 8 * i, seed, want_dg=with_add)
        g = wl[i][0]
    bl = []
    for i in range(n_layers):
        (_x2, mean1, rstd1, _xn1, qkv, _ao, x1, mean2, rstd2, _xn2, h) = want[i]
        bl.append(dict(img=pb[i * E:(i + 1) * E], x=x if i == 0 else want[i - 1][0], mean1=mean1, rstd1=rstd1, qkv=qkv, x1=x1,
                       mean2=mean2, rstd2=rstd2, h=h, gamma1=ps[i]["gamma1"], gamma2=ps[i]["gamma2"], site0=s0 + 8 * i, **grads()))
    dx, per, dgcat = ops.gs_stack_bwd(dx2, bl, key_mask, n_seq, S, scale, drop_p, seed, want_dg=with_add)
    torch.cuda.synchronize()
    if not torch.equal(dx, g):
        bad.append(f"dx: {int((dx != g).sum())} elements differ")
    for i in range(n_layers):
        for name, a, b in zip(("dym", "dpre", "dx1m", "dqkv"), per[i], wl[i][2:6]):
            if not torch.equal(a, b):
                bad.append(f"backward layer {i} {name}: {int((a != b).sum())} elements differ")
        for k, name in enumerate(("dgamma2", "dbeta2", "dgamma1", "dbeta1")):
            if not torch.equal(bl[i][name], wl[i][6 + k]):
                bad.append(f"backward layer {i} {name} differs")
        if with_add and not torch.equal(dgcat[:, 256 * i:256 * (i + 1)], wl[i][10]):
            bad.append(f"backward layer {i} dg (column block of the concatenated buffer) differs")
    assert not bad, "\n".join(bad)


def test_gs_stack_launch_is_bit_reproducible(gpu_device):
    flat, offs, ps, x, key_mask, gcat, dx2 = _stack_setup(512, 8, 4, True, True, seed=77)
    pf, _pb = ops.gs_pack(flat, offs, 4)
    seed = _seed_tensor(0x1234)
    E = ops.GS_LAYER_ELEMS
    layers = [dict(img=pf[i * E:(i + 1) * E], site0=8 * i, seq_add=gcat[:, 256 * i:256 * (i + 1)], **p) for i, p in enumerate(ps)]
    a = ops.gs_stack_fwd(x, layers, key_mask, 512, 8, 32 ** -0.5, 1e-5, 0.1, seed, train=True)
    b = ops.gs_stack_fwd(x, layers, key_mask, 512, 8, 32 ** -0.5, 1e-5, 0.1, seed, train=True)
    torch.cuda.synchronize()
    assert all(torch.equal(s, t) for ra, rb in zip(a, b) for s, t in zip(ra, rb))
