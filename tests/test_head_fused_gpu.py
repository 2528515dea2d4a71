"""The argument head fused with its consumers (csrc/head_fused.hip) against the plain-torch fp32
restatements of tests/torch_ops_ref.py: arg-max decode, masked-CE forward, dlogits - dense rows and the compact token list,
all 11 slots and the 6-slot range without arcs, row counts that do not fill the last workgroup."""
import pytest
import torch

from tests import torch_ops_ref as R
from deepsvg_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
C_ = 257


def _rand(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def _setup(rows, group, seed):
    n_out = group * C_
    x = _rand(rows, 256, seed=seed, dtype=torch.bfloat16)
    w = _rand(n_out, 256, seed=seed + 1, scale=0.08, dtype=torch.bfloat16)
    b = _rand(n_out, seed=seed + 2, scale=0.5)
    return x, w, b, n_out, ops.head_pack(w)


@pytest.mark.parametrize("rows,group", [(37, 11), (256, 11), (1500, 6), (4099, 11), (513, 1)])
def test_head_argmax_equals_argmax_of_the_logits(gpu_device, rows, group):
    x, w, b, n_out, img = _setup(rows, group, seed=rows)
    got = ops.head_argmax(x, img, b, n_out, C_)
    lg = R._head_logits(x, w, b, n_out).view(rows, group, C_)
    want = lg.argmax(-1).to(torch.int32).reshape(-1)
    same = got == want
    if not bool(same.all()):
        # fp32 summation order differs: where the two disagree the two candidates' logits must be numerically tied
        r = (~same).nonzero().squeeze(1)
        flat = lg.view(-1, C_)
        a = flat[r, got[r].long()]
        bb = flat[r, want[r].long()]
        assert (a - bb).abs().max().item() <= 2e-5 * lg.abs().max().item(), "arg-max differs beyond rounding ties"
        assert r.numel() <= max(2, rows * group // 1000)
    assert int(got.min()) >= 0 and int(got.max()) < C_
    # ties -> lowest class: duplicate output rows (same weights, same bias) inside one slot
    w2 = w.clone()
    b2 = b.clone()
    w2[5] = w2[3]
    b2[5] = b2[3]
    w2[C_ * (group - 1) + 200] = w2[C_ * (group - 1) + 100]
    b2[C_ * (group - 1) + 200] = b2[C_ * (group - 1) + 100]
    got2 = ops.head_argmax(x, ops.head_pack(w2), b2, n_out, C_).view(rows, group)
    assert not bool((got2[:, 0] == 5).any()) and not bool((got2[:, group - 1] == 200).any())


def _targets(n_tok, group, seed, live_frac=0.35):
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand(n_tok, group, generator=g) < 0.5).float()
    w[torch.rand(n_tok, generator=g) > live_frac] = 0.0
    t = torch.randint(-1, C_, (n_tok * group,), generator=g).to(torch.int32)      # (-1: PAD_VAL + 1 clamps to class 0)
    return t.to(DEV), w.view(-1).contiguous().to(DEV)


@pytest.mark.parametrize("rows,group", [(300, 11), (2048, 6), (1111, 11)])
def test_head_lse_and_dlogits_dense_rows(gpu_device, rows, group):
    x, w, b, n_out, img = _setup(rows, group, seed=7 * rows)
    tgt, wt = _targets(rows, group, seed=rows)
    lse, sc = ops.head_lse(x, img, b, n_out, C_, tgt, wt)
    lse_r, sc_r = R.head_lse(x, w, b, n_out, C_, tgt, wt)
    assert torch.allclose(lse, lse_r, rtol=0, atol=2e-5 * lse_r.abs().max().item())
    assert torch.allclose(sc, sc_r, rtol=2e-6, atol=0)
    gs = torch.tensor([0.7], device=DEV)
    d = ops.head_dlogits(x, img, b, n_out, C_, tgt, wt, lse, sc, gs, 2.0)
    d_r = R.head_dlogits(x, w, b, n_out, C_, tgt, wt, lse_r, sc_r, gs, 2.0)
    assert d.shape == (rows, n_out) and d.stride(0) % 8 == 0
    assert torch.equal(d.float() == 0, d_r[:, :n_out].float() == 0) or \
        ((d.float() - d_r[:, :n_out].float()).abs().max() <= 1e-2 * d_r.float().abs().max())
    assert (d.float() - d_r[:, :n_out].float()).abs().max().item() <= 1e-2 * d_r.float().abs().max().item()
    # rows without any loss term are exact zeros, and so are the padding columns of the buffer behind the view
    dead = (wt.view(rows, group).sum(1) == 0)
    assert torch.count_nonzero(d[dead]) == 0
    full = d.as_strided((rows, d.stride(0)), (d.stride(0), 1))
    assert torch.count_nonzero(full[:, n_out:]) == 0
    # w = None: every row counts
    lse1, sc1 = ops.head_lse(x, img, b, n_out, C_, tgt, None)
    lse1_r, sc1_r = R.head_lse(x, w, b, n_out, C_, tgt, None)
    assert torch.allclose(lse1, lse1_r, rtol=0, atol=2e-5 * lse1_r.abs().max().item()) and torch.allclose(sc1, sc1_r, rtol=2e-6)


@pytest.mark.parametrize("group", [11, 6])
def test_head_lse_and_dlogits_on_the_compact_token_list(gpu_device, group):
    """the training layout: x holds the gathered rows of the tokens that carry loss (padded with -1 entries), targets and
    weights stay indexed by the source token"""
    n_tok = 3000
    n_out = group * C_
    tgt, wt = _targets(n_tok, group, seed=group)
    live, count = ops.live_rows(wt, group)
    n_live = int(count)
    rows = (n_live + 127) // 128 * 128
    idx = live[:rows].contiguous()
    assert 0 < n_live < rows and int(idx[-1]) == -1
    xs = _rand(n_tok, 256, seed=3, dtype=torch.bfloat16)
    x = ops.gather_groups(xs, idx, rows, 1)
    w = _rand(n_out, 256, seed=4, scale=0.08, dtype=torch.bfloat16)
    b = _rand(n_out, seed=5, scale=0.5)
    img = ops.head_pack(w)
    lse, sc = ops.head_lse(x, img, b, n_out, C_, tgt, wt, tok_idx=idx)
    lse_r, sc_r = R.head_lse(x, w, b, n_out, C_, tgt, wt, tok_idx=idx)
    assert torch.allclose(lse, lse_r, rtol=0, atol=2e-5 * lse_r.abs().max().item())
    assert torch.allclose(sc, sc_r, rtol=2e-6, atol=0)
    assert abs(float(sc[1]) - float(wt.sum())) < 0.5
    gs = torch.tensor([1.3], device=DEV)
    d = ops.head_dlogits(x, img, b, n_out, C_, tgt, wt, lse, sc, gs, 1.0, tok_idx=idx)
    d_r = R.head_dlogits(x, w, b, n_out, C_, tgt, wt, lse_r, sc_r, gs, 1.0, tok_idx=idx)
    assert (d.float() - d_r[:, :n_out].float()).abs().max().item() <= 1e-2 * d_r.float().abs().max().item()
    assert torch.count_nonzero(d[n_live:]) == 0
    # the same numbers as the unfused pair (head GEMM -> bf16 logits -> masked CE) up to the logits' bf16 rounding
    lg = ops.gemm(x, w, bias=b)
    lse_u, sc_u = ops.masked_ce_fwd(lg, tgt, wt, C_, group, tok_idx=idx)
    assert abs(float(sc[0] / sc[1]) - float(sc_u[0] / sc_u[1])) <= 2e-3 * abs(float(sc_u[0] / sc_u[1]))


# ---- categorical sampling on the device (round 5): Gumbel arg-max, deepsvg/model/utils.py:75-79 -----------------------------
def _seed(v):
    return torch.tensor([v], dtype=torch.int64, device=DEV)


def _agree_up_to_ties(got, noisy, what, max_frac=2e-3):
    """got: int32 [n]; noisy: fp32 [n, C] perturbed logits of the restatement.  Where the kernel picked another class than the
    restatement's arg-max the two candidates must be tied within the rounding of the two implementations' logarithms / sums"""
    want = noisy.argmax(-1)
    diff = (got.long() != want).nonzero().squeeze(1)
    if diff.numel():
        a = noisy[diff, got[diff].long()]
        b = noisy[diff, want[diff]]
        scale = noisy.abs().max().item()
        assert (a - b).abs().max().item() <= 3e-5 * scale, what
        assert diff.numel() <= max(2, int(max_frac * got.numel())), (what, diff.numel(), got.numel())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,group,C,temperature", [(1000, 1, 7, 1.0), (333, 11, 257, 1.0), (257, 11, 257, 1e-4), (4100, 3, 70, 0.3)])
def test_sample_rows_is_the_gumbel_argmax_of_the_restated_noise(gpu_device, dtype, rows, group, C, temperature):
    lg = _rand(rows, group * C + 5, seed=rows + group, scale=2.0, dtype=dtype)
    sd = _seed(0x1234ABCD5678 + rows)
    got = ops.sample_rows(lg, C, temperature, sd, 7001, group=group)
    noise = R.gumbel_noise(sd, 7001, rows, group * C, DEV)
    assert torch.isfinite(noise).all() and noise.min().item() > -2.82 and noise.max().item() < 16.7
    noisy = (lg[:, :group * C].float() + temperature * noise).reshape(rows * group, C)
    _agree_up_to_ties(got, noisy, "sample_rows")
    assert int(got.min()) >= 0 and int(got.max()) < C
    # another seed / another site: other draws (at a temperature where the noise decides)
    if temperature >= 0.3:
        assert not torch.equal(got, ops.sample_rows(lg, C, temperature, _seed(99), 7001, group=group))
        assert not torch.equal(got, ops.sample_rows(lg, C, temperature, sd, 7002, group=group))
    assert torch.equal(got, ops.sample_rows(lg, C, temperature, sd, 7001, group=group))


def test_sample_rows_draws_follow_the_softmax(gpu_device):
    """2^18 independent draws from ONE 7-class and one 257-class distribution at T = 1 and T = 0.5: Pearson chi-square against
    softmax(logits / T) (statistic below the 1 - 1e-5 quantile), and independence of the draws of neighbouring rows"""
    import math
    n = 1 << 18
    for C, T, seed in ((7, 1.0, 1), (7, 0.5, 2), (257, 1.0, 3)):
        g = torch.Generator().manual_seed(seed)
        row = torch.randn(C, generator=g) * 1.5
        lg = row.to(DEV).repeat(n, 1).contiguous()
        got = ops.sample_rows(lg, C, T, _seed(4242 + seed), 7001).long()
        p = torch.softmax(row.double() / T, 0)
        cnt = torch.bincount(got.cpu(), minlength=C).double()
        keep = p * n >= 5          # (classes with an expected count below 5 are pooled)
        exp = torch.cat([p[keep] * n, (p[~keep] * n).sum().view(1)])
        obs = torch.cat([cnt[keep], cnt[~keep].sum().view(1)])
        if exp[-1] < 1e-9:
            exp, obs = exp[:-1], obs[:-1]
        chi = ((obs - exp) ** 2 / exp).sum().item()
        dof = exp.numel() - 1
        # Wilson-Hilferty bound of the chi-square quantile at 1 - 1e-5 (z = 4.265)
        bound = dof * (1 - 2 / (9 * dof) + 4.265 * math.sqrt(2 / (9 * dof))) ** 3
        assert chi < bound, (C, T, chi, bound)
        if C == 7:      # neighbouring rows: 7 x 7 contingency table, independence
            a, b = got[0::2].cpu(), got[1::2].cpu()
            tab = torch.zeros(C, C, dtype=torch.float64)
            tab.view(-1).index_add_(0, a * C + b, torch.ones(a.numel(), dtype=torch.float64))
            e = tab.sum(1, keepdim=True) * tab.sum(0, keepdim=True) / tab.sum()
            ok = e >= 5
            chi2 = (((tab - e) ** 2 / e)[ok]).sum().item()
            d2 = (C - 1) * (C - 1)
            assert chi2 < d2 * (1 - 2 / (9 * d2) + 4.265 * math.sqrt(2 / (9 * d2))) ** 3, (chi2, d2)


@pytest.mark.parametrize("rows,group,temperature", [(300, 11, 1.0), (1500, 6, 0.2), (4099, 11, 1e-4)])
def test_head_sample_draws_what_sample_rows_draws_on_the_dense_logits(gpu_device, rows, group, temperature):
    """the fused head perturbs element (row, slot * C + c) with the same noise as sample_rows on the [rows, group * C] logit
    matrix: identical samples (up to numerical ties of the two logit computations), the logits never stored"""
    x, w, b, n_out, img = _setup(rows, group, seed=rows + 3)
    sd = _seed(0x777 + rows)
    got = ops.head_sample(x, img, b, n_out, C_, temperature, sd, 7002)
    lg = R._head_logits(x, w, b, n_out)
    noisy = (lg + temperature * R.gumbel_noise(sd, 7002, rows, n_out, DEV)).reshape(rows * group, C_)
    _agree_up_to_ties(got, noisy, "head_sample")
    dense = ops.sample_rows(lg.contiguous(), C_, temperature, sd, 7002, group=group)
    assert (dense != got).float().mean().item() < 2e-3
    if temperature <= 1e-4:     # the reference's default temperature: the arg-max wherever the two best logits are 1e-3 apart
        top2 = lg.view(rows * group, C_).topk(2, -1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2e-3
        am = ops.head_argmax(x, img, b, n_out, C_)
        assert torch.equal(got[clear], am[clear]) and clear.float().mean().item() > 0.9
