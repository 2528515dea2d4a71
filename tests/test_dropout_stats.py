"""Statistical evidence for the counter-hash dropout the benchmark runs with (the reference draws its masks
from torch's Philox stream, deepsvg/model/layers/improved_transformer.py:26-33,45,52-53 and positional_encoding.py:26-28;
a counter hash cannot reproduce those bits, so "statistically equivalent Bernoulli(1 - p) masks" is the claim to back).

The three draw schemes of the kernels - the library's standard draws (groups of 8 ids, dsvg_common.h drop_mult), the fused
FFN's hidden-site draws (groups of 16, ffn_fused.hip drop2_*) and the attention probabilities' row draws (attn_drop_*) - are
restated bit for bit in tests/torch_ops_ref.py, and tests/test_kernels_gpu.py proves kernel == restatement.  Here the
restatements (CPU) and, under -m gpu, the masks the real kernels produce are tested on 2^24 elements at p = 0.1 for
  * the keep rate (within 4 sigma of 1 - thresh16 / 65536),
  * independence (2 x 2 chi-square, p > 1e-4) of an element and its neighbour at distance 1, 2, 8, 16 and one row (256) away,
  * independence of the SAME element under two sites and under two consecutive seeds of the trainer's seed recurrence,
  * the distribution of the number of dropped elements per hash group (8 / 16 / 32 elements share one counter hash) and per
    row of 256 / 512 against Binomial(L, p) (chi-square goodness of fit: structure inside a group that pairwise tests miss -
    e.g. multipliers in arithmetic progression - shows here)."""
import numpy as np
import pytest
import torch
from scipy import stats

from tests import torch_ops_ref as R

P = 0.1
N = 1 << 24
P_DROP = 6554.0 / 65536.0            # round(0.1 * 65536) / 65536: what the 16-bit threshold realises
ALPHA = 1e-4
SEED0 = 0x1234567ABCDEF


def _next_seed(s):
    """the trainer's seed recurrence (csrc/optim.hip advance_step_kernel)"""
    m = (1 << 64) - 1
    s = (s + 0x9E3779B97F4A7C15) & m
    z = s
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def _seed_t(s):
    return torch.tensor([s if s < (1 << 63) else s - (1 << 64)], dtype=torch.int64)


def _mask(scheme, seed, site, n=N):
    """bool tensor [n]: True = dropped"""
    if scheme == "attn":
        rows, keys = n // 32, 32
        row = torch.arange(rows, dtype=torch.int64).unsqueeze(1).expand(rows, keys)
        key = torch.arange(keys, dtype=torch.int64).unsqueeze(0).expand(rows, keys)
        m = R.attn_drop_mult(P, _seed_t(seed), site, row, key).reshape(-1)
    else:
        fn = R.drop_mult if scheme == "std" else R.drop2_mult
        m = fn(P, _seed_t(seed), site, torch.arange(n, dtype=torch.int64))
    return m == 0


def _chi2_2x2(a, b):
    a, b = a.to(torch.int64), b.to(torch.int64)
    n11 = int((a & b).sum()); n10 = int((a & (1 - b)).sum()); n01 = int(((1 - a) & b).sum())
    n = a.numel()
    n00 = n - n11 - n10 - n01
    obs = np.array([[n00, n01], [n10, n11]], dtype=np.float64)
    exp = obs.sum(1, keepdims=True) * obs.sum(0, keepdims=True) / n
    return float(((obs - exp) ** 2 / exp).sum())


def _check_rate(d, what):
    n = d.numel()
    rate = float(d.sum()) / n
    sigma = (P_DROP * (1 - P_DROP) / n) ** 0.5
    assert abs(rate - P_DROP) <= 4 * sigma, f"{what}: drop rate {rate:.6f} vs {P_DROP:.6f} (sigma {sigma:.2e})"
    return rate


def _check_lags(d, lags, what):
    crit = stats.chi2.ppf(1 - ALPHA, 1)
    for lag in lags:
        # disjoint pairs (i, i + lag) with i in the first half of every block of 2 * lag elements
        n = d.numel() // (2 * lag) * (2 * lag)
        v = d[:n].view(-1, 2, lag)
        c = _chi2_2x2(v[:, 0].reshape(-1), v[:, 1].reshape(-1))
        assert c < crit, f"{what}: elements {lag} apart are not independent (chi2 = {c:.1f} >= {crit:.1f})"


def _check_row_counts(d, L, what):
    k = d.view(-1, L).sum(1).numpy()
    rows = k.size
    lo, hi = max(0, int(stats.binom.ppf(1e-4, L, P_DROP))), min(L, int(stats.binom.ppf(1 - 1e-4, L, P_DROP)))
    edges = np.arange(lo, hi + 2)
    obs = np.array([np.sum(k < lo)] + [np.sum(k == v) for v in range(lo, hi + 1)] + [np.sum(k > hi)], dtype=np.float64)
    pm = stats.binom.pmf(np.arange(lo, hi + 1), L, P_DROP)
    exp = rows * np.concatenate([[stats.binom.cdf(lo - 1, L, P_DROP)], pm, [stats.binom.sf(hi, L, P_DROP)]])
    keep = exp >= 5
    obs_k = np.concatenate([obs[keep], [obs[~keep].sum()]]) if (~keep).any() else obs
    exp_k = np.concatenate([exp[keep], [exp[~keep].sum()]]) if (~keep).any() else exp
    sel = exp_k > 0
    c = float(((obs_k[sel] - exp_k[sel]) ** 2 / exp_k[sel]).sum())
    crit = stats.chi2.ppf(1 - ALPHA, int(sel.sum()) - 1)
    assert c < crit, f"{what}: dropped elements per row of {L} do not follow Binomial({L}, p) (chi2 = {c:.1f} >= {crit:.1f})"
    # dispersion index: variance of the row counts over the binomial variance (1 when the draws of a row are independent)
    return float(k.var() / (L * P_DROP * (1 - P_DROP)))


@pytest.mark.parametrize("scheme,group,row_len", [("std", 8, 256), ("drop2", 16, 512), ("attn", 32, 32)])
def test_dropout_masks_are_independent_bernoulli_draws(scheme, group, row_len):
    for seed in (SEED0, 0xFFFFFFFF, 3):
        d = _mask(scheme, seed, 404)
        _check_rate(d, scheme)
        _check_lags(d, (1, 2, 3, 4, 8, 16, 32, 256), scheme)
        # the draws of one hash group (8 / 16 / 32 elements share a counter hash) and of a whole row
        _check_row_counts(d, group, f"{scheme}, hash group")
        disp = _check_row_counts(d, row_len, scheme)
        assert 0.97 < disp < 1.03, (scheme, disp)


@pytest.mark.parametrize("scheme", ["std", "drop2", "attn"])
def test_dropout_masks_of_two_sites_and_of_two_consecutive_steps_are_independent(scheme):
    crit = stats.chi2.ppf(1 - ALPHA, 1)
    d0 = _mask(scheme, SEED0, 404)
    for what, d1 in (("next site", _mask(scheme, SEED0, 405)), ("site + 8 (next layer)", _mask(scheme, SEED0, 412)),
                     ("next step's seed", _mask(scheme, _next_seed(SEED0), 404))):
        _check_rate(d1, f"{scheme}, {what}")
        c = _chi2_2x2(d0, d1)
        assert c < crit, f"{scheme}: the same element under {what} is not independent (chi2 = {c:.1f})"
        assert not torch.equal(d0, d1)


def test_keep_rate_over_many_steps_and_sites():
    """the drop rate of 64 (seed, site) combinations (2^18 elements each) scatters like 64 independent binomial samples"""
    n = 1 << 18
    s = SEED0
    z = []
    for step in range(16):
        s = _next_seed(s)
        for site in (100, 101, 404, 408):
            d = _mask("std", s, site, n)
            z.append((float(d.sum()) / n - P_DROP) / (P_DROP * (1 - P_DROP) / n) ** 0.5)
    z = np.array(z)
    assert np.abs(z).max() < 4.5, z
    # the z-scores themselves ~ N(0, 1): a Kolmogorov-Smirnov test against it
    assert stats.kstest(z, "norm").pvalue > 1e-3, stats.kstest(z, "norm")


@pytest.mark.gpu
def test_dropout_masks_from_the_kernels_pass_the_same_tests(gpu_device):
    """the masks as the KERNELS apply them (dsvg_drop_apply on a tensor of ones: standard draws; dsvg_ffn_fwd's training output
    h with a huge positive bias so that every unit passes the ReLU: the hidden site's draws), 2^24 elements each"""
    from deepsvg_amd import ops
    dev = "cuda"
    seed = _seed_t(SEED0).to(dev)
    x = torch.ones((N // 256, 256), dtype=torch.bfloat16, device=dev)
    d = (ops.drop_apply(x, P, 404, seed) == 0).reshape(-1).cpu()
    assert torch.equal(d, _mask("std", SEED0, 404)), "kernel mask != restatement"
    _check_rate(d, "drop_apply kernel")
    _check_lags(d, (1, 2, 8, 16, 256), "drop_apply kernel")
    _check_row_counts(d, 256, "drop_apply kernel")
    # hidden site of the fused FFN: W1 = 0, b1 = 1 -> relu(1) = 1 everywhere, h = drop(1) in fragment order
    L = 131072 + 512 + 131072 + 256 + 256 + 8
    flat = torch.zeros(8 + L, device=dev)
    o = 8
    offs = torch.tensor([[o, o + 131072, o + 131072 + 512, o + 262144 + 512, o + 262144 + 768]], dtype=torch.int64, device=dev)
    flat[o + 131072:o + 131072 + 512] = 1.0
    flat[o + 262144 + 512:o + 262144 + 768] = 1.0
    pf, pb, b1f = ops.ffn_pack(flat, offs, 1)
    rows = N // 512
    xin = torch.randn((rows, 256), device=dev).to(torch.bfloat16)
    y, h, xh, rstd = ops.ffn_fwd(xin, pf[:ops.FFN_FWD_LAYER_ELEMS], b1f[0], torch.zeros(256, device=dev), 1e-5, P, 403, 404,
                                 seed, train=True)
    dh = (h == 0).reshape(-1).cpu()
    _check_rate(dh, "ffn_fwd hidden site")
    _check_lags(dh, (1, 2, 8, 16, 512), "ffn_fwd hidden site")
    _check_row_counts(dh, 512, "ffn_fwd hidden site")
