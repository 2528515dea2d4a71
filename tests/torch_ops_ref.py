"""Plain-PyTorch restatement of every op in deepsvg_amd/ops.py (same names, same signatures).

Two uses, both test-only:
  * `-m gpu` tests compare each HIP kernel with the function of the same name here (fp32 reference);
  * `-m "not gpu"` tests monkey-patch deepsvg_amd.ops with this module (see conftest.emulated_ops) to exercise
    the host logic (autograd wiring, flat parameter store, trainer, gloo data-parallel path) on CPU.
The product never imports this file.

The dropout masks reproduce the counter-based hash of deepsvg_amd/csrc/dsvg_common.h bit for bit.
"""
import numpy as np
import torch
import torch.nn.functional as F

RELU = 1
M32 = 0xFFFFFFFF


# ----------------------------------------------------------------------------------------------------
# dropout hash (dsvg_common.h: dsvg_hash32 / drop_make / drop_mult)
# ----------------------------------------------------------------------------------------------------
def _hash32_int(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def _hash32_t(x):
    x = x & M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & M32
    x = x ^ (x >> 16)
    return x


def _mulhi32(a, b):
    """high 32 bits of the 64-bit product of two tensors / ints < 2^32 (int64 arithmetic without overflow)"""
    a0, a1 = a & 0xFFFF, a >> 16
    b0, b1 = b & 0xFFFF, b >> 16
    lo_lo, hi_lo, lo_hi, hi_hi = a0 * b0, a1 * b0, a0 * b1, a1 * b1
    cross = (lo_lo >> 16) + (hi_lo & 0xFFFF) + lo_hi
    return ((hi_lo >> 16) + (cross >> 16) + hi_hi) & M32


def _drop_word(h, i):
    """word i (< 16) of a group / row hash (dsvg_common.h drop_word): fold(h * C_i) = lo32 ^ hi32 of the 64-bit product,
    C_i = ((0x7feb352d * (i + 1)) ^ (0x846ca68b >> i)) | 1"""
    c = (((0x7FEB352D * (i + 1)) & M32) ^ (0x846CA68B >> i)) | 1
    lo = ((h & 0xFFFF) * c + (((h >> 16) * c) & 0xFFFF) * 65536) & M32
    return lo ^ _mulhi32(h, c)


def drop_mult(p, seed, site, idx):
    """multiplier tensor (0 or 65536/(65536-thresh16)) for int64 element ids `idx` (dsvg_common.h drop_mult)"""
    if p <= 0 or seed is None:
        return torch.ones(idx.shape, dtype=torch.float32, device=idx.device)
    s = int(seed.reshape(-1)[0].item()) & 0xFFFFFFFFFFFFFFFF
    s0 = _hash32_int((s & M32) ^ ((site * 0x9E3779B1) & M32))
    s1 = _hash32_int(((s >> 32) + site * 0x85EBCA77 + 0x165667B1) & M32)
    thresh = min(65535, int(np.float32(np.float32(p) * np.float32(65536.0) + np.float32(0.5))))
    scale = float(np.float32(65536.0) / np.float32(65536 - thresh))
    idx = idx.to(torch.int64)
    g = idx >> 3
    slot = idx & 7
    lo, hi = g & M32, (g >> 32) & M32
    h = _hash32_t(lo ^ s0)
    h = ((h ^ s1) + hi * 0x9E3779B1) & M32
    w = _drop_word(h, slot >> 1)
    draw = torch.where((slot & 1) == 1, w >> 16, w & 0xFFFF)
    return torch.where(draw < thresh, torch.zeros((), dtype=torch.float32, device=idx.device),
                       torch.full((), scale, dtype=torch.float32, device=idx.device))


def drop2_mult(p, seed, site, idx):
    """draw scheme "v2" of the fused kernels (csrc/ffn_fused.hip drop2_*): one counter hash per 16 consecutive ids, a
    one-multiply finaliser per pair of 16-bit draws; same thresholds / scale / seed mixing as drop_mult"""
    if p <= 0 or seed is None:
        return torch.ones(idx.shape, dtype=torch.float32, device=idx.device)
    s = int(seed.reshape(-1)[0].item()) & 0xFFFFFFFFFFFFFFFF
    s0 = _hash32_int((s & M32) ^ ((site * 0x9E3779B1) & M32))
    s1 = _hash32_int(((s >> 32) + site * 0x85EBCA77 + 0x165667B1) & M32)
    thresh = min(65535, int(np.float32(np.float32(p) * np.float32(65536.0) + np.float32(0.5))))
    scale = float(np.float32(65536.0) / np.float32(65536 - thresh))
    idx = idx.to(torch.int64)
    g = idx >> 4
    slot = idx & 15
    lo, hi = g & M32, (g >> 32) & M32
    h = _hash32_t(lo ^ s0)
    h = ((h ^ s1) + hi * 0x9E3779B1) & M32
    w = _drop_word(h, slot >> 1)
    draw = torch.where((slot & 1) == 1, w >> 16, w & 0xFFFF)
    return torch.where(draw < thresh, torch.zeros((), dtype=torch.float32, device=idx.device),
                       torch.full((), scale, dtype=torch.float32, device=idx.device))


def _ids(rows, cols, device, ld=None):
    ld = cols if ld is None else ld
    return torch.arange(rows, device=device, dtype=torch.int64).unsqueeze(1) * ld + \
        torch.arange(cols, device=device, dtype=torch.int64).unsqueeze(0)


def _f(t):
    return t.to(torch.float32)


# ----------------------------------------------------------------------------------------------------
def gemm(a, b, *, a_kc=True, b_kc=True, bias=None, res=None, res_pre=False, act=0, gate=None, gate_scale=1.0,
         drop_p=0.0, drop_site=0, a_drop_p=0.0, a_drop_site=0, seed=None, out=None, out_dtype=None,
         accumulate=False, split_k=1, impl=0, rowsum=None):
    af = _f(a)
    if a_drop_p > 0:
        af = af * drop_mult(a_drop_p, seed, a_drop_site, _ids(a.shape[0], a.shape[1], a.device))
        af = _f(af.to(a.dtype))      # the kernel re-rounds the dropped operand to the storage type
    A = af if a_kc else af.t()
    B = _f(b) if b_kc else _f(b).t()
    v = A @ B.t()
    if rowsum is not None:
        rowsum.copy_(rowsum + A.sum(1) if accumulate else A.sum(1))
    M, N = v.shape
    if bias is not None:
        v = v + _f(bias)
    if res is not None and res_pre:
        v = v + _f(res)
    if act == RELU:
        v = torch.relu(v)
    if gate is not None:
        v = torch.where(_f(gate) > 0, v * gate_scale, torch.zeros_like(v))
    if drop_p > 0:
        v = v * drop_mult(drop_p, seed, drop_site, _ids(M, N, v.device))
    if res is not None and not res_pre:
        v = v + _f(res)
    dt = out.dtype if out is not None else (out_dtype or a.dtype)
    if out is None:
        return v.to(dt)
    if accumulate:
        v = v + _f(out)
    out.copy_(v.to(dt))
    return out


def split_k_for(M, N, K, target_blocks=None):
    """same policy as deepsvg_amd.ops.split_k_for (the emulated gemm ignores the value, but the host logic that
    decides whether the bias gradient can ride on the weight-gradient GEMM depends on it)"""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    s = max(1, (target_blocks or 256) // tiles)
    s = min(s, max(1, K // 128))
    if s >= 8:
        s = s // 8 * 8
    return s


def colsum(a, *, out=None, accumulate=False, drop_p=0.0, drop_site=0, seed=None):
    af = _f(a)
    if drop_p > 0:
        af = af * drop_mult(drop_p, seed, drop_site, _ids(a.shape[0], a.shape[1], a.device))
    s = af.sum(0)
    if out is None:
        return s
    out.copy_(s + out if accumulate else s)
    return out


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    xf = _f(x)
    mean = xf.mean(-1)
    var = ((xf - mean.unsqueeze(-1)) ** 2).mean(-1)
    rstd = torch.rsqrt(var + eps)
    y = (xf - mean.unsqueeze(-1)) * rstd.unsqueeze(-1) * gamma + beta
    return y.to(x.dtype), mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, *, res=None, dgamma=None, dbeta=None, accumulate=False, dx=None, masked=None):
    if masked is not None:
        o, dg, db = layernorm_bwd(dy, x, mean, rstd, gamma, res=res, dgamma=dgamma, dbeta=dbeta, accumulate=accumulate, dx=dx)
        return o, dg, db, drop_apply(o, masked[0], masked[1], masked[2])
    dyf, xf = _f(dy), _f(x)
    xh = (xf - mean.unsqueeze(-1)) * rstd.unsqueeze(-1)
    gd = dyf * gamma
    c1 = gd.mean(-1, keepdim=True)
    c2 = (gd * xh).mean(-1, keepdim=True)
    o = rstd.unsqueeze(-1) * (gd - c1 - xh * c2)
    if res is not None:
        o = o + _f(res)
    dg, db = (dyf * xh).sum(0), dyf.sum(0)
    if dgamma is not None:
        dgamma.copy_(dgamma + dg if accumulate else dg)
        dg = dgamma
    if dbeta is not None:
        dbeta.copy_(dbeta + db if accumulate else db)
        db = dbeta
    o = o.to(x.dtype)
    if dx is not None:
        dx.copy_(o)
        o = dx
    return o, dg, db


def _mask_bits(mask, S):
    """int64 [n] bitmask (S <= 64) or int32 [n] valid-prefix lengths (longer sequences) -> bool [n, S]"""
    bits = torch.arange(S, device=mask.device, dtype=torch.int64)
    if mask.dtype == torch.int32:
        return bits.unsqueeze(0) < mask.long().unsqueeze(1)
    return ((mask.unsqueeze(1) >> bits) & 1).bool()


def _attn_probs(qkv, key_mask, n_seq, S, H, scale, causal=False):
    d = 32 * H
    q, k, v = _f(qkv).view(n_seq, S, 3, H, 32).permute(2, 0, 3, 1, 4)      # each (n_seq, H, S, 32)
    s = (q * scale) @ k.transpose(-1, -2)
    if causal:          # square_subsequent_mask: query i sees keys j <= i
        s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=s.device), diagonal=1), float("-inf"))
    if key_mask is not None:
        valid = _mask_bits(key_mask, S)
        s = s.masked_fill(~valid.view(n_seq, 1, 1, S), float("-inf"))
    return q, k, v, torch.softmax(s, dim=-1)


def attn_drop_mult(p, seed, site, row, key):
    """dropout of the attention probabilities (dsvg_common.h attn_drop_row / attn_drop_key): one counter hash per
    (row = (sequence * heads + head) * S + query, block of 32 keys), a multiply-fold per pair of keys"""
    shape = torch.broadcast_shapes(row.shape, key.shape)
    if p <= 0 or seed is None:
        return torch.ones(shape, dtype=torch.float32, device=row.device)
    s = int(seed.reshape(-1)[0].item()) & 0xFFFFFFFFFFFFFFFF
    s0 = _hash32_int((s & M32) ^ ((site * 0x9E3779B1) & M32))
    s1 = _hash32_int(((s >> 32) + site * 0x85EBCA77 + 0x165667B1) & M32)
    thresh = min(65535, int(np.float32(np.float32(p) * np.float32(65536.0) + np.float32(0.5))))
    scale = float(np.float32(65536.0) / np.float32(65536 - thresh))
    row, key = row.to(torch.int64), key.to(torch.int64)
    g = row * 8 + (key >> 5)
    lo, hi = g & M32, (g >> 32) & M32
    h = _hash32_t(lo ^ s0)
    h = _hash32_t((h + hi * 0x9E3779B1 + s1) & M32)
    w = _drop_word(h, (key & 31) >> 1)
    draw = torch.where((key & 1) == 1, w >> 16, w & 0xFFFF)
    return torch.where(draw < thresh, torch.zeros((), dtype=torch.float32, device=row.device),
                       torch.full((), scale, dtype=torch.float32, device=row.device))


def _attn_drop(p, seed, site, n_seq, S, H, device):
    row = torch.arange(n_seq * H * S, device=device, dtype=torch.int64).view(n_seq, H, S, 1)
    key = torch.arange(S, device=device, dtype=torch.int64).view(1, 1, 1, S)
    return attn_drop_mult(p, seed, site, row, key)


# packed layout helpers: seq_off int32 [n_seq+1]; sequence b = rows seq_off[b]..seq_off[b+1]-1
def _unpack_rows(x, seq_off, n_seq, S):
    """packed [rows, w] -> dense zero-padded [n_seq * S, w] plus the dense row index of every packed row"""
    off = seq_off.long()
    lens = off[1:] - off[:-1]
    total = int(off[-1])
    seq = torch.repeat_interleave(torch.arange(n_seq, device=x.device), lens)
    pos = torch.arange(total, device=x.device) - off[:-1][seq]
    idx = seq * S + pos
    dense = torch.zeros((n_seq * S, x.shape[1]), dtype=x.dtype, device=x.device)
    dense[idx] = x[:total]
    return dense, idx, lens


def _len_mask(lens):
    return ((torch.ones_like(lens) << lens) - 1).to(torch.int64)


def _repack_rows(dense, idx, rows):
    out = torch.zeros((rows, dense.shape[1]), dtype=dense.dtype, device=dense.device)
    out[:idx.numel()] = dense[idx]
    return out


def attention_tiles(seq_off, n_seq, max_rows=32):
    off = seq_off.tolist()
    first, start = [], None
    for i in range(n_seq):
        if i % 64 == 0:                 # tiles never cross a segment of 64 sequences (device kernel: a wave per segment)
            start = None
        if start is None or off[i + 1] - start > max_rows:
            first.append(i)
            start = off[i]
    t = torch.zeros(n_seq + 2, dtype=torch.int32, device=seq_off.device)
    t[:len(first)] = torch.tensor(first, dtype=torch.int32)
    t[len(first)] = n_seq
    t[n_seq + 1] = len(first)
    return t


def attention_fwd(qkv, key_mask, n_seq, S, n_heads, scale, drop_p=0.0, drop_site=0, seed=None, seq_off=None,
                  tiles=None, causal=False, only_row=None, out=None):
    if only_row is not None or out is not None:     # incremental decoding step: one row of a caller-owned buffer
        full = attention_fwd(qkv, key_mask, n_seq, S, n_heads, scale, drop_p, drop_site, seed, causal=causal)
        if out is None:
            return full
        if only_row is None:
            out.copy_(full)
        else:
            out.view(n_seq, S, -1)[:, only_row] = full.view(n_seq, S, -1)[:, only_row]
        return out
    if seq_off is not None:
        dense, idx, lens = _unpack_rows(qkv, seq_off, n_seq, S)
        o = attention_fwd(dense, _len_mask(lens), n_seq, S, n_heads, scale, drop_p, drop_site, seed)
        return _repack_rows(o, idx, qkv.shape[0])
    if qkv.shape[0] > n_seq * S:        # dense layout, rows past the last sequence are zero-filled
        o = attention_fwd(qkv[:n_seq * S], key_mask, n_seq, S, n_heads, scale, drop_p, drop_site, seed)
        return torch.cat([o, torch.zeros((qkv.shape[0] - n_seq * S, o.shape[1]), dtype=o.dtype, device=o.device)])
    H = n_heads
    q, k, v, P = _attn_probs(qkv, key_mask, n_seq, S, H, scale, causal)
    Pd = P * _attn_drop(drop_p, seed, drop_site, n_seq, S, H, qkv.device)
    o = Pd @ v                                                               # (n_seq, H, S, 32)
    return o.permute(0, 2, 1, 3).reshape(n_seq * S, H * 32).to(qkv.dtype)


def attention_bwd(qkv, key_mask, dout, n_seq, S, n_heads, scale, drop_p=0.0, drop_site=0, seed=None, seq_off=None,
                  tiles=None, causal=False):
    if seq_off is not None:
        dense, idx, lens = _unpack_rows(qkv, seq_off, n_seq, S)
        ddense, _, _ = _unpack_rows(dout, seq_off, n_seq, S)
        g = attention_bwd(dense, _len_mask(lens), ddense, n_seq, S, n_heads, scale, drop_p, drop_site, seed)
        return _repack_rows(g, idx, qkv.shape[0])
    if qkv.shape[0] > n_seq * S:
        g = attention_bwd(qkv[:n_seq * S], key_mask, dout[:n_seq * S], n_seq, S, n_heads, scale, drop_p, drop_site, seed)
        return torch.cat([g, torch.zeros((qkv.shape[0] - n_seq * S, g.shape[1]), dtype=g.dtype, device=g.device)])
    H = n_heads
    q, k, v, P = _attn_probs(qkv, key_mask, n_seq, S, H, scale, causal)
    mult = _attn_drop(drop_p, seed, drop_site, n_seq, S, H, qkv.device)
    do = _f(dout).view(n_seq, S, H, 32).permute(0, 2, 1, 3)
    dv = (P * mult).transpose(-1, -2) @ do
    dP = (do @ v.transpose(-1, -2)) * mult
    D = (P * dP).sum(-1, keepdim=True)
    dS = P * (dP - D)
    dq = (dS @ k) * scale
    dk = dS.transpose(-1, -2) @ (q * scale)
    out = torch.stack([dq, dk, dv], dim=0).permute(1, 3, 0, 2, 4).reshape(n_seq * S, 3 * H * 32)
    return out.to(qkv.dtype)


def seq_lens(commands, S, eos_id=4):
    valid = (commands.view(-1, S).long() == eos_id).cumsum(1) == 0
    return valid.sum(1).to(torch.int32)


def build_masks(commands, S, G=0, eos_id=4, want_group_mask=False):
    if S > 64:
        assert not want_group_mask
        return seq_lens(commands, S, eos_id), None, None
    cmd = commands.view(-1, S)
    n_seq = cmd.shape[0]
    is_eos = (cmd.long() == eos_id)
    valid = is_eos.cumsum(1) == 0
    w = (1 << torch.arange(S, dtype=torch.int64, device=cmd.device))
    key_mask = (valid.long() * w).sum(1)
    seq_visible = (is_eos.sum(1) < S - 1).to(torch.int32)
    group_mask = None
    if want_group_mask:
        wg = (1 << torch.arange(G, dtype=torch.int64, device=cmd.device))
        group_mask = (seq_visible.view(-1, G).long() * wg).sum(1)
    return key_mask, seq_visible, group_mask


def group_index(commands, S, m_id=0):
    return (commands.view(-1, S).long() == m_id).cumsum(1).to(torch.int32).reshape(-1)


def visible_first(visible):
    v = visible.bool()
    n = v.numel()
    idx = torch.arange(n, device=visible.device)
    old_of_new = torch.cat([idx[v], idx[~v]]).to(torch.int32)
    new_of_old = torch.empty(n, dtype=torch.int32, device=visible.device)
    new_of_old[old_of_new.long()] = idx.to(torch.int32)
    return new_of_old, old_of_new, v.sum().to(torch.int32).reshape(1)


def gather_groups(src, idx, n_groups, S, out=None, n_src=None):
    g = idx[:n_groups].long().clamp(min=0)
    ok = torch.ones_like(g, dtype=torch.bool) if n_src is None else g < n_src
    rows = (torch.where(ok, g, torch.zeros_like(g)).unsqueeze(1) * S + torch.arange(S, device=src.device)).reshape(-1)
    val = src[rows] * ok.repeat_interleave(S).unsqueeze(1).to(src.dtype)
    if out is None:
        return val.clone()
    out[:n_groups * S] = val
    return out


def pack_tokens(commands, args, key_mask, n_seq, S):
    valid = _mask_bits(key_mask, S)                                      # [n_seq, S], a prefix per row
    lens = valid.sum(1)
    seq_off = torch.zeros(n_seq + 1, dtype=torch.int32, device=commands.device)
    seq_off[1:] = torch.cumsum(lens, 0).to(torch.int32)
    total = int(seq_off[-1])
    cap = n_seq * S
    a = args.reshape(cap, -1)
    sel = valid.reshape(-1).nonzero().squeeze(1)
    pcmd = commands.reshape(-1)[0].repeat(cap).clone()
    parg = a[0:1].repeat(cap, 1).clone()
    ppos = torch.zeros(cap, dtype=torch.int32, device=commands.device)
    pcmd[:total] = commands.reshape(-1)[sel]
    parg[:total] = a[sel]
    ppos[:total] = (sel % S).to(torch.int32)
    return seq_off, pcmd, parg, ppos


def embed_gather(commands, args, command_embed, arg_embed, dtype, group_embed=None, groups=None):
    T = commands.numel()
    a = args.view(T, -1)
    iv = (a.long() + 1).clamp(0, arg_embed.shape[0] - 1)
    A = arg_embed[iv].reshape(T, -1)
    ic = commands.view(-1).long().clamp(0, command_embed.shape[0] - 1)
    R = command_embed[ic]
    if group_embed is not None:
        R = R + group_embed[groups.long()]
    return A.to(dtype), R.to(dtype)


def embed_scatter(commands, args, dA, dR, d_arg_embed, d_command_embed, groups=None, d_group_embed=None):
    T = commands.numel()
    n_argvals, E = d_arg_embed.shape
    a = args.view(T, -1)
    iv = (a.long() + 1).clamp(0, n_argvals - 1).reshape(-1)
    d_arg_embed.zero_().index_add_(0, iv, _f(dA).reshape(-1, E))
    ic = commands.view(-1).long().clamp(0, d_command_embed.shape[0] - 1)
    d_command_embed.zero_().index_add_(0, ic, _f(dR))
    if d_group_embed is not None:
        d_group_embed.zero_().index_add_(0, groups.long(), _f(dR))


def add_pos_fwd(x, pos, n_seq, S, dtype, drop_p=0.0, drop_site=0, seed=None):
    d = pos.shape[1]
    v = pos[:S].unsqueeze(0).expand(n_seq, S, d).reshape(n_seq * S, d)
    if x is not None:
        v = v + _f(x)
    v = v * drop_mult(drop_p, seed, drop_site, _ids(n_seq * S, d, pos.device))
    return v.to(dtype)


def add_pos_bwd(dy, n_seq, S, d_pos, *, want_dx=True, accumulate=False, drop_p=0.0, drop_site=0, seed=None):
    d = dy.shape[1]
    g = _f(dy) * drop_mult(drop_p, seed, drop_site, _ids(n_seq * S, d, dy.device))
    s = g.view(n_seq, S, d).sum(0)
    d_pos.copy_(d_pos + s if accumulate else s)
    return g.to(dy.dtype) if want_dx else None


def masked_mean_fwd(x, mask, n_seq, S, seq_off=None):
    if seq_off is not None:
        dense, _, lens = _unpack_rows(x, seq_off, n_seq, S)
        return masked_mean_fwd(dense, _len_mask(lens), n_seq, S)
    d = x.shape[1]
    valid = _mask_bits(mask, S).to(torch.float32)
    xf = _f(x).view(n_seq, S, d)
    return ((xf * valid.unsqueeze(-1)).sum(1) / valid.sum(1, keepdim=True)).to(x.dtype)


def masked_mean_bwd(dout, mask, n_seq, S, seq_off=None, total_rows=None):
    if seq_off is not None:
        probe = torch.zeros((int(total_rows), 1), dtype=dout.dtype, device=dout.device)
        _, idx, lens = _unpack_rows(probe, seq_off, n_seq, S)
        return _repack_rows(masked_mean_bwd(dout, _len_mask(lens), n_seq, S), idx, int(total_rows))
    valid = _mask_bits(mask, S).to(torch.float32)
    g = _f(dout) / valid.sum(1, keepdim=True)
    return (g.unsqueeze(1) * valid.unsqueeze(-1)).reshape(n_seq * S, -1).to(dout.dtype)


def bcast_add_fwd_(x, g, n_seq, S, drop_p=0.0, drop_site=0, seed=None):
    """x[t] += drop(g)[t // S]: the dropout acts on the per-sequence row before the broadcast (mask id = b * d + c)"""
    d = x.shape[1]
    gd = _f(g) * drop_mult(drop_p, seed, drop_site, _ids(n_seq, d, x.device))
    gv = gd.unsqueeze(1).expand(n_seq, S, d).reshape(n_seq * S, d)
    x.copy_((_f(x) + gv).to(x.dtype))
    return x


def bcast_add_bwd(dx, n_seq, S, drop_p=0.0, drop_site=0, seed=None, n_seq_out=None, mask_site=None, out=None):
    d = dx.shape[1]
    g = _f(dx[:n_seq * S]).view(n_seq, S, d).sum(1) * drop_mult(drop_p, seed, drop_site, _ids(n_seq, d, dx.device))
    g = g.to(dx.dtype)
    if n_seq_out is not None and n_seq_out > n_seq:
        g = torch.cat([g, g.new_zeros((n_seq_out - n_seq, d))])
    if out is not None:
        out.copy_(g)
        g = out
    if mask_site is not None:
        return g, drop_apply(dx, drop_p, mask_site, seed)
    return g


def copy_many(pairs):
    for dst, src in pairs:
        dst.copy_(src)


def loss_targets(tgt_commands, tgt_args, cmd_args_mask, eos_id=4, seq_perm=None):
    if seq_perm is not None:
        perm = seq_perm[:tgt_commands.shape[0]].long()
        res = loss_targets(tgt_commands[perm], tgt_args[perm], cmd_args_mask, eos_id)
        vis = torch.empty_like(res[4])
        vis[perm] = res[4]
        return res[0], res[1], res[2], res[3], vis
    n_seq, S1 = tgt_commands.shape
    c = tgt_commands.long()
    is_eos = c == eos_id
    pm = (is_eos.cumsum(1) == 0)
    ext = pm.clone()
    ext[:, 3:] |= pm[:, :S1 - 3]
    vis = (is_eos.sum(1) < S1 - 1)
    cmd_tgt = c[:, 1:].clamp(0, cmd_args_mask.shape[0] - 1).to(torch.int32).contiguous()
    cmd_w = (ext[:, 1:] & vis.unsqueeze(1)).to(torch.float32).contiguous()
    arg_tgt = (tgt_args[:, 1:].long() + 1).to(torch.int32).contiguous()
    arg_w = cmd_args_mask[cmd_tgt.long()].to(torch.float32).contiguous()
    return cmd_tgt, cmd_w, arg_tgt, arg_w, vis.to(torch.int32)


def _ce_rows(logits2d, C_, group):
    n_tok = logits2d.shape[0]
    return _f(logits2d)[:, :group * C_].reshape(n_tok * group, C_)


def _compact_tw(target, w, tok_idx, group):
    """targets / weights of the listed tokens (negative index -> weight 0), compact order"""
    sel = tok_idx.long()
    rows = (sel.clamp(min=0).unsqueeze(1) * group + torch.arange(group, device=sel.device)).reshape(-1)
    t = target.reshape(-1)[rows]
    ww = (torch.ones(target.numel(), device=target.device) if w is None else w.reshape(-1))[rows].clone()
    ww[(sel < 0).repeat_interleave(group)] = 0
    return t, ww


def masked_ce_fwd(logits2d, target, w, C_, group=1, tok_idx=None):
    if tok_idx is not None:     # compact logits / lse, source-indexed targets and weights
        t, ww = _compact_tw(target, w, tok_idx, group)
        return masked_ce_fwd(logits2d, t, ww, C_, group)
    rows = _ce_rows(logits2d, C_, group)
    lse = torch.logsumexp(rows, dim=-1)
    t = target.long().clamp(0, C_ - 1)
    nll = lse - rows.gather(1, t.unsqueeze(1)).squeeze(1)
    ww = torch.ones_like(lse) if w is None else w
    lse = torch.where(ww != 0, lse, torch.zeros_like(lse))
    sc = torch.stack([(ww * torch.where(ww != 0, nll, torch.zeros_like(nll))).sum(), ww.sum()])
    return lse, sc


def loss_combine_fwd(scs, weights):
    terms = [sc[0] / sc[1] for sc in scs]
    total = sum(float(w) * t for w, t in zip(weights, terms))
    return torch.stack([total] + terms).to(torch.float32)


def loss_combine_bwd(dtotal, dterms, weights, device):
    rows = []
    for w, dt in zip(weights, dterms):
        g = torch.zeros((), dtype=torch.float32, device=device)
        if dtotal is not None:
            g = g + dtotal.reshape(()).to(torch.float32) * float(w)
        if dt is not None:
            g = g + dt.reshape(()).to(torch.float32)
        rows.append(torch.stack([g, torch.zeros_like(g)]))
    return torch.stack(rows)


def live_rows(w, group):
    live_tok = (w.reshape(-1, group) != 0).any(1)
    idx = live_tok.nonzero().squeeze(1).to(torch.int32)
    live = torch.full((live_tok.numel(),), -1, dtype=torch.int32, device=w.device)
    live[:idx.numel()] = idx
    return live, torch.tensor([idx.numel()], dtype=torch.int32, device=w.device)


def scatter_rows(src, idx, dst, accumulate=False):
    sel = idx[:src.shape[0]].long()
    ok = sel >= 0
    if accumulate:
        dst[sel[ok]] = (_f(dst[sel[ok]]) + _f(src[ok])).to(dst.dtype)
    else:
        dst[sel[ok]] = src[ok]
    return dst


def masked_ce_bwd(logits2d, target, w, lse, sum_count, gscale, coef, C_, group=1, pad_to=8, tok_idx=None,
                  logits_compact=False):
    if tok_idx is not None and logits_compact:
        t, ww = _compact_tw(target, w, tok_idx, group)
        return masked_ce_bwd(logits2d, t, ww, lse, sum_count, gscale, coef, C_, group, pad_to)
    if tok_idx is not None:     # compact backward: dense result, then pick the listed tokens (negative -> zero row)
        dense = masked_ce_bwd(logits2d, target, w, lse, sum_count, gscale, coef, C_, group, pad_to)
        sel = tok_idx.long()
        out = dense[sel.clamp(min=0)].clone()
        out[sel < 0] = 0
        width = group * C_
        ld = (width + pad_to - 1) // pad_to * pad_to
        buf = torch.zeros((sel.numel(), ld), dtype=logits2d.dtype, device=logits2d.device)
        buf[:, :width] = out
        return buf[:, :width]
    rows = _ce_rows(logits2d, C_, group)
    n_tok = logits2d.shape[0]
    ww = torch.ones(rows.shape[0], device=rows.device) if w is None else w
    g = coef * (gscale.reshape(()) if gscale is not None else 1.0) / sum_count[1]
    sm = torch.exp(rows - lse.unsqueeze(1))
    onehot = F.one_hot(target.long().clamp(0, C_ - 1), C_).to(torch.float32)
    d = (ww * g).unsqueeze(1) * (sm - onehot)
    d = torch.where((ww != 0).unsqueeze(1), d, torch.zeros_like(d))
    width = group * C_
    ld = (width + pad_to - 1) // pad_to * pad_to
    buf = torch.zeros((n_tok, ld), dtype=logits2d.dtype, device=logits2d.device)
    buf[:, :width] = d.reshape(n_tok, width).to(logits2d.dtype)
    return buf[:, :width]


def assemble_batch(rows, slot_off, variant, G, L, grouped, want_args=True, want_rel=False, pad_val=-1.0, args_dim=256):
    """restatement of dsvg_assemble_batch on the packed store (per sequence: SOS, rows, EOS, EOS padding)"""
    from oracle import batch_assembly_oracle as B
    N = variant.numel()
    Gs = 1 if grouped else G
    cmds = torch.empty(N, Gs, L)
    args = torch.empty(N, Gs, L, 11) if want_args else None
    rel = torch.empty(N, Gs, L, 11) if want_rel else None
    r16 = rows.numpy().astype("float32")
    off = slot_off.numpy()
    for n in range(N):
        v = int(variant[n])
        for g in range(Gs):
            lo = off[v * G + (0 if grouped else g)]
            hi = off[v * G + (G if grouped else g + 1)]
            c = torch.full((L,), 4.0)
            a = torch.full((L, 11), float(pad_val))
            c[0] = 5.0
            k = min(int(hi - lo), L - 2)
            c[1:1 + k] = torch.from_numpy(r16[lo:lo + k, 0])
            a[1:1 + k] = torch.from_numpy(r16[lo:lo + k, 1:])
            cmds[n, g] = c
            if want_args:
                args[n, g] = a
            if want_rel:
                rel[n, g] = torch.from_numpy(B.relative_args(c.numpy(), a.numpy(), pad_val, args_dim))
    return cmds, args, rel


def match_costs(cmd_logits, args_logits, vis_logits, tgt_commands, tgt_args, cam, N, G, Gp, n_args, args_dim, n_cmd,
                eos_id, weights=(2.0, 1.0, 1.0)):
    """plain-torch restatement of dsvg_match_costs (deepsvg/model/model.py:311-339), written per (g, p) pair"""
    S1 = tgt_commands.shape[-1]
    S = S1 - 1
    cl = cmd_logits[:, :n_cmd].float().reshape(N, Gp, S, n_cmd)
    al = args_logits[:, :n_args * args_dim].float().reshape(N, Gp, S, n_args, args_dim)
    vl = vis_logits[:, :2].float().reshape(N, Gp, 2)
    tc = tgt_commands[..., 1:].long()                            # (N, G, S)  the sequence without SOS
    ta = (tgt_args[..., 1:, :].long() + 1).clamp(0, args_dim - 1)
    is_eos = tc == eos_id
    vis = is_eos.sum(-1) < S - 1                                 # (N, G)
    pm = (is_eos.cumsum(-1) == 0).float()
    ext = pm.clone()
    ext[..., 3:] = (pm[..., 3:] + pm[..., :-3]).clamp(max=1)
    ext = ext * vis.unsqueeze(-1).float()
    mask = cam[tc.clamp(0, n_cmd - 1)]                           # (N, G, S, n_args)
    lpa, lpc, lpv = al.log_softmax(-1), cl.log_softmax(-1), vl.log_softmax(-1)
    cost = torch.empty(N, G, Gp)
    for g in range(G):
        for p in range(Gp):
            ce_a = -lpa[:, p].gather(-1, ta[:, g].unsqueeze(-1)).squeeze(-1)           # (N, S, n_args)
            ce_c = -lpc[:, p].gather(-1, tc[:, g].clamp(0, n_cmd - 1).unsqueeze(-1)).squeeze(-1)
            ce_v = -lpv[:, p].gather(-1, vis[:, g].long().unsqueeze(-1)).squeeze(-1)
            la = (ce_a * mask[:, g]).sum((-1, -2)) / mask[:, g].sum((-1, -2))
            lc = (ce_c * ext[:, g]).sum(-1) / ext[:, g].sum(-1)
            cost[:, g, p] = weights[0] * la + weights[1] * lc + weights[2] * ce_v
    return cost, vis.to(torch.int32)


def argmax_rows(logits2d, C, group=1):
    n = logits2d.shape[0]
    return logits2d[:, :group * C].float().reshape(n * group, C).argmax(-1).to(torch.int32)


def gumbel_noise(seed, site, rows, n_cols, device):
    """[rows, n_cols] fp32 Gumbel noise of the device sampler (dsvg_common.h dsvg_gumbel): element (row, col) takes word
    (col & 3) of the hash of key = row * ceil(n_cols / 4) + (col >> 2); u = (word >> 9 + 1/2) 2^-23, g = -log(-log(u)).  The
    bits are restated exactly; the two logarithms are torch's (the kernel's are v_log_f32: equal to ~1e-6 relative)"""
    s = int(seed.reshape(-1)[0].item()) & 0xFFFFFFFFFFFFFFFF
    s0 = _hash32_int((s & M32) ^ ((site * 0x9E3779B1) & M32))
    s1 = _hash32_int(((s >> 32) + site * 0x85EBCA77 + 0x165667B1) & M32)
    k4 = (n_cols + 3) // 4
    row = torch.arange(rows, dtype=torch.int64, device=device).view(-1, 1)
    col = torch.arange(n_cols, dtype=torch.int64, device=device).view(1, -1)
    key = row * k4 + (col >> 2)
    lo, hi = key & M32, (key >> 32) & M32
    h = _hash32_t(lo ^ s0)
    h = ((h ^ s1) + hi * 0x9E3779B1) & M32
    w = torch.zeros_like(key)
    for i in range(4):
        w = torch.where((col & 3) == i, _drop_word(h, i), w)
    u = ((w >> 9).to(torch.float32) + 0.5) * (1.0 / 8388608.0)
    return -torch.log(-torch.log(u))


def sample_rows(logits2d, C, temperature, seed, site, group=1):
    n = logits2d.shape[0]
    g = gumbel_noise(seed, site, n, group * C, logits2d.device)
    noisy = logits2d[:, :group * C].float() + float(temperature) * g
    return noisy.reshape(n * group, C).argmax(-1).to(torch.int32)


def head_sample(x, packed, bias, n_out, C, temperature, seed, site):
    return sample_rows(_head_logits(x, packed, bias, n_out), C, temperature, seed, site, n_out // C)


def match_assign(cost, vis):
    """scipy's Hungarian solver, as the reference (deepsvg/model/model.py:341-348)"""
    from scipy.optimize import linear_sum_assignment
    N, G, Gp = cost.shape
    assign = torch.empty(N, Gp, dtype=torch.int32)
    for n in range(N):
        rows = vis[n].bool()
        a = linear_sum_assignment(cost[n][rows].double().numpy())[1].tolist() if bool(rows.any()) else []
        assign[n] = torch.tensor(a + sorted(set(range(Gp)) - set(a)), dtype=torch.int32)
    base = (torch.arange(N, dtype=torch.int32) * Gp).unsqueeze(1)
    idx = (base + assign).reshape(-1)
    inv = torch.empty_like(idx)
    inv[idx.long()] = torch.arange(N * Gp, dtype=torch.int32)
    return assign, idx, inv


def sumsq(x, out=None):
    s = (x.double() ** 2).sum().to(torch.float32).reshape(1)
    if out is not None:
        out.copy_(s)
        return out
    return s


def adamw_step_(p, g, m, v, lr, step, *, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, gnorm_sq=None,
                max_norm=0.0, grad_scale=1.0):
    coef = grad_scale
    if gnorm_sq is not None and max_norm > 0:
        norm = torch.sqrt(gnorm_sq.reshape(())) * grad_scale
        coef = coef * torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    t = float(step.item())
    lr_ = float(lr.item())
    gi = g * coef
    m.mul_(beta1).add_(gi, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
    bc1 = 1 - beta1 ** t
    bc2s = (1 - beta2 ** t) ** 0.5
    p.mul_(1 - lr_ * weight_decay).addcdiv_(m, v.sqrt() / bc2s + eps, value=-lr_ / bc1)


def cast_weights(src, dst=None, dst_t=None):
    if dst is not None:
        dst.view(-1)[:src.numel()].copy_(src.reshape(-1).to(dst.dtype))
    if dst_t is not None:
        dst_t.view(-1)[:src.numel()].copy_(src.t().reshape(-1).to(dst_t.dtype))
    return dst, dst_t


def advance_step_(counter, seed):
    if counter is not None:
        counter += 1
    if seed is not None:
        s = (int(seed.item()) + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z = z ^ (z >> 31)
        if z >= 1 << 63:
            z -= 1 << 64
        seed.fill_(z)


# ----------------------------------------------------------------------------------------------------
# fused FFN sub-block.  The emulated "packed" images simply carry the two weight matrices in bf16 with the LayerNorm
# affine folded into linear1 exactly as csrc/ffn_fused.hip does: per layer packed_fwd = [W1' (512 x 256) | W2 (256 x 512)],
# packed_bwd = [W1' | W2 | unused]; b1f = b1 + W1 beta
# ----------------------------------------------------------------------------------------------------
FFN_FWD_LAYER_ELEMS, FFN_BWD_LAYER_ELEMS = 16 * 32 * 512, 16 * 48 * 512


def ffn_pack(flat, offs, n_layers, packed_fwd=None, packed_bwd=None, b1f=None, w2p=None):
    dev = flat.device
    if packed_fwd is None:
        packed_fwd = torch.empty(n_layers * FFN_FWD_LAYER_ELEMS, dtype=torch.bfloat16, device=dev)
    if packed_bwd is None:
        packed_bwd = torch.zeros(n_layers * FFN_BWD_LAYER_ELEMS, dtype=torch.bfloat16, device=dev)
    if b1f is None:
        b1f = torch.empty((n_layers, 512), dtype=torch.float32, device=dev)
    for i in range(n_layers):
        o1, ob1, o2, og, ob = (int(v) for v in offs[i])
        W1 = flat[o1:o1 + 131072].view(512, 256)
        gamma, beta = flat[og:og + 256], flat[ob:ob + 256]
        w = torch.cat([(W1 * gamma).reshape(-1), flat[o2:o2 + 131072]]).to(torch.bfloat16)
        packed_fwd[i * FFN_FWD_LAYER_ELEMS:(i + 1) * FFN_FWD_LAYER_ELEMS] = w
        packed_bwd[i * FFN_BWD_LAYER_ELEMS:i * FFN_BWD_LAYER_ELEMS + 262144] = w
        b1f.view(n_layers, 512)[i] = flat[ob1:ob1 + 512] + W1 @ beta
        if w2p is not None:
            w2p.view(n_layers, 256, 512)[i] = flat[o2:o2 + 131072].view(256, 512)[:, _ffn_frag_perm(dev)].to(torch.bfloat16)
    return packed_fwd, packed_bwd, b1f


def _ffn_weights(packed_layer):
    return packed_layer[:131072].view(512, 256), packed_layer[131072:262144].view(256, 512)


def _ffn_normalise(x, eps):
    """xh = (x - mean) * rstd rounded to the storage dtype (no affine: it is folded into the packed linear1)"""
    xf = _f(x)
    mean = xf.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((xf - mean) ** 2).mean(-1, keepdim=True) + eps)
    return ((xf - mean) * rstd).to(x.dtype), mean.squeeze(-1), rstd.squeeze(-1)


def _ffn_hidden_ids(rows, device):
    """dropout ids of the hidden site in the order the fused kernels' lanes own them: unit j = 32 c + 8 q + 4 half + e
    (q < 4, e < 4) of token t has id 512 t + 32 c + 16 half + 4 q + e"""
    j = torch.arange(512, device=device, dtype=torch.int64)
    jl = j & 31
    perm = (j - jl) + 16 * ((jl >> 2) & 1) + 4 * (jl >> 3) + (jl & 3)
    return torch.arange(rows, device=device, dtype=torch.int64).unsqueeze(1) * 512 + perm.unsqueeze(0)


def _ffn_hidden(xn, W1, b1, drop_p, site_hidden, seed):
    """drop_h(relu(linear1(xn))) rounded to the storage dtype, and the multiplier mask"""
    pre = _f(xn) @ _f(W1).t() + b1
    m = drop2_mult(drop_p, seed, site_hidden, _ffn_hidden_ids(xn.shape[0], xn.device))
    return (torch.relu(pre) * m).to(xn.dtype), pre, m


def ffn_fwd(x, packed_fwd_layer, b1f, b2, eps=1e-5, drop_p=0.0, site_hidden=0, site_res=0, seed=None, out=None,
            train=False, into=None):
    W1, W2 = _ffn_weights(packed_fwd_layer)
    xh, _, rstd = _ffn_normalise(x, eps)
    h, _, _ = _ffn_hidden(xh, W1, b1f, drop_p, site_hidden, seed)
    y = _f(h) @ _f(W2).t() + b2
    y = y * drop_mult(drop_p, seed, site_res, _ids(x.shape[0], 256, x.device)) + _f(x)
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        y = out
    if not train:
        return y
    hp = torch.empty_like(h)
    hp[:, _ffn_frag_perm(x.device)] = h
    if into is not None:
        into[0].copy_(hp)
        into[1].copy_(xh)
        hp, xh = into
    return y, hp, xh, rstd


def ffn_wgrad_finish_many(layers):
    for ts in layers:
        ffn_wgrad_finish(*ts)


def _ffn_frag_perm(device):
    """position p(j) of hidden unit j in the fragment-ordered h / dpre matrices: bits 2 and 3 of j swapped"""
    j = torch.arange(512, device=device)
    return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1)


def ffn_bwd(x, dy, packed_bwd_layer, b1f, eps=1e-5, drop_p=0.0, site_hidden=0, site_res=0, seed=None):
    W1, W2 = _ffn_weights(packed_bwd_layer)
    rows = x.shape[0]
    xh, mean, rstd = _ffn_normalise(x, eps)
    h, pre, m_h = _ffn_hidden(xh, W1, b1f, drop_p, site_hidden, seed)
    if drop_p > 0:
        dym = (_f(dy) * drop_mult(drop_p, seed, site_res, _ids(rows, 256, x.device))).to(x.dtype)
    else:
        dym = dy
    dh = _f(dym) @ _f(W2)
    dpre = torch.where(pre > 0, dh * m_h, torch.zeros_like(dh)).to(x.dtype)
    g = _f(dpre) @ _f(W1)                                                    # gradient wrt the normalised rows
    xhf = (_f(x) - mean.unsqueeze(-1)) * rstd.unsqueeze(-1)
    dx = _f(dy) + rstd.unsqueeze(-1) * (g - g.mean(-1, keepdim=True) - xhf * (g * xhf).mean(-1, keepdim=True))
    perm = _ffn_frag_perm(x.device)
    hp, dp = torch.empty_like(h), torch.empty_like(dpre)
    hp[:, perm] = h
    dp[:, perm] = dpre
    return dx.to(x.dtype), hp, dp, xh, dym


def ffn_bwd_dx(dpre, x, dy, packed_bwd_layer, eps=1e-5, masked=None):
    if masked is not None:
        dx = ffn_bwd_dx(dpre, x, dy, packed_bwd_layer, eps)
        return dx, drop_apply(dx, masked[0], masked[1], masked[2])
    W1, _ = _ffn_weights(packed_bwd_layer)
    _, mean, rstd = _ffn_normalise(x, eps)
    g = _f(dpre)[:, _ffn_frag_perm(x.device)] @ _f(W1)          # dpre[:, p(j)] is unit j
    xhf = (_f(x) - mean.unsqueeze(-1)) * rstd.unsqueeze(-1)
    dx = _f(dy) + rstd.unsqueeze(-1) * (g - g.mean(-1, keepdim=True) - xhf * (g * xhf).mean(-1, keepdim=True))
    return dx.to(x.dtype)


def ffn_wgrad_finish(g1p, db1p, g2p, w1, gamma, beta, dw1, db1, dw2, dgamma, dbeta):
    perm = _ffn_frag_perm(g1p.device)
    G1 = g1p.view(512, 256)[perm]                   # G1[j] = G1p[p(j)]
    b = db1p.view(512)[perm]
    W1 = w1.view(512, 256)
    dw1.view(512, 256).copy_(G1 * gamma + b.unsqueeze(1) * beta)
    db1.view(512).copy_(b)
    dw2.view(256, 512).copy_(g2p.view(256, 512)[:, perm])
    dgamma.view(256).copy_((W1 * G1).sum(0))
    dbeta.view(256).copy_(W1.t() @ b)


# ------------------------------------------------------------------------------------------------
# fused attention sub-block.  The emulated "packed" image carries in_proj_weight | out_proj.weight in bf16.
# ------------------------------------------------------------------------------------------------
ATTN_LAYER_ELEMS = 512 * 512


def attn_pack(flat, offs, n_layers, packed=None):
    if packed is None:
        packed = torch.empty(n_layers * ATTN_LAYER_ELEMS, dtype=torch.bfloat16, device=flat.device)
    for i in range(n_layers):
        oi, oo = (int(v) for v in offs[i])
        packed[i * ATTN_LAYER_ELEMS:(i + 1) * ATTN_LAYER_ELEMS] = torch.cat(
            [flat[oi:oi + 196608], flat[oo:oo + 65536]]).to(torch.bfloat16)
    return packed


ATTN_BWD_LAYER_ELEMS = 512 * 512    # emulated image: out_proj.weight [256, 256] | in_proj_weight [768, 256], bf16


def attn_pack_bwd(flat, offs, n_layers, packed=None):
    """emulated image of a layer: out_proj.weight (bf16, row-major [256, 256]) followed by in_proj_weight ([768, 256])"""
    if packed is None:
        packed = torch.empty(n_layers * ATTN_BWD_LAYER_ELEMS, dtype=torch.bfloat16, device=flat.device)
    for i in range(n_layers):
        oi, oo = int(offs[i][0]), int(offs[i][1])
        lay = packed[i * ATTN_BWD_LAYER_ELEMS:(i + 1) * ATTN_BWD_LAYER_ELEMS]
        lay[:65536] = flat[oo:oo + 65536].to(torch.bfloat16)
        lay[65536:] = flat[oi:oi + 768 * 256].to(torch.bfloat16)
    return packed


def attention_bwd_outproj(qkv, key_mask, dx1m, wo_packed_bwd, n_seq, S, scale, drop_p=0.0, drop_site=0, seed=None, seq_off=None,
                          tiles=None):
    """the two launches it replaces: dao = dx1m @ Wo (rounded to the storage dtype), then attention_bwd"""
    dao = (_f(dx1m) @ _f(wo_packed_bwd[:65536].view(256, 256))).to(qkv.dtype)
    return attention_bwd(qkv, key_mask, dao, n_seq, S, 8, scale, drop_p, drop_site, seed, seq_off=seq_off, tiles=tiles)


def attn_bwd_dx(dqkv, x, mean, rstd, gamma, res, packed_bwd_layer, *, dx=None, dgamma=None, dbeta=None, accumulate=False,
                masked=None):
    """the two launches it replaces, without the bf16 rounding of the intermediate: dxn1 = dqkv @ W_in (fp32), then layernorm_bwd"""
    rows = x.shape[0]
    dxn1 = _f(dqkv) @ _f(packed_bwd_layer[65536:].view(768, 256))
    return layernorm_bwd(dxn1, _f(x), mean[:rows], rstd[:rows], gamma, res=_f(res), dgamma=dgamma, dbeta=dbeta,
                         accumulate=accumulate, dx=dx if dx is not None else torch.empty_like(x), masked=masked)


# ------------------------------------------------------------------------------------------------
# the argument head fused with its consumers (csrc/head_fused.hip).  The emulated image is the bf16 weight itself; the
# restatement is head GEMM (fp32 accumulation, logits NOT rounded to bf16 - they never leave the chip) + the masked-CE math.
# ------------------------------------------------------------------------------------------------
def head_pack(weight_lp):
    return weight_lp.clone()


def _head_logits(x, packed, bias, n_out):
    return x.float() @ packed[:n_out].float().t() + bias.float()


def head_argmax(x, packed, bias, n_out, C):
    lg = _head_logits(x, packed, bias, n_out).view(x.shape[0], n_out // C, C)
    return lg.argmax(-1).to(torch.int32).reshape(-1)


def head_lse(x, packed, bias, n_out, C, target, w, tok_idx=None):
    return masked_ce_fwd(_head_logits(x, packed, bias, n_out), target, w, C, n_out // C, tok_idx=tok_idx)


def head_dlogits(x, packed, bias, n_out, C, target, w, lse, sum_count, gscale, coef, tok_idx=None):
    lg = _head_logits(x, packed, bias, n_out)
    d = masked_ce_bwd(lg, target, w, lse, sum_count, gscale, coef, C, n_out // C, pad_to=8, tok_idx=tok_idx,
                      logits_compact=tok_idx is not None)
    return d.to(torch.bfloat16)


def attn_block_fwd(x, packed_layer, in_bias, out_bias, gamma, beta, key_mask, n_seq, S, scale, eps=1e-5, drop_p=0.0,
                   site_probs=0, site_res=0, seed=None, seq_off=None, tiles=None, train=False, seq_add=None, site_seq_add=0,
                   into=None):
    """the unfused launches, composed (LayerNorm -> in_proj -> attention -> out_proj + dropout + residual [-> bcast add])"""
    win, wo = packed_layer[:196608].view(768, 256), packed_layer[196608:].view(256, 256)
    xn, mean, rstd = layernorm_fwd(x, gamma, beta, eps)
    qkv = gemm(xn, win, bias=in_bias)
    ao = attention_fwd(qkv, key_mask, n_seq, S, 8, scale, drop_p, site_probs, seed, seq_off=seq_off, tiles=tiles)
    x1 = gemm(ao, wo, bias=out_bias, res=x, drop_p=drop_p, drop_site=site_res, seed=seed)
    if seq_add is not None:
        bcast_add_fwd_(x1, seq_add, n_seq, S, drop_p, site_seq_add, seed)
    if train:
        res = (x1, xn, qkv, ao, mean, rstd)
        if into is not None:
            for dst, src in zip(into, res):
                dst.copy_(src)
            res = tuple(into)
        return res
    return x1


# ------------------------------------------------------------------------------------------------
# fused group-stage layer (csrc/group_stage.hip).  The emulated images carry in_proj | out_proj | linear1 | linear2 in bf16;
# the restatement is the composition of the unfused launches in their own order.
# ------------------------------------------------------------------------------------------------
GS_LAYER_ELEMS = 8 * 128 * 512


def gs_pack(flat, offs, n_layers, packed_fwd=None, packed_bwd=None):
    dev = flat.device
    if packed_fwd is None:
        packed_fwd = torch.empty(n_layers * GS_LAYER_ELEMS, dtype=torch.bfloat16, device=dev)
    if packed_bwd is None:
        packed_bwd = torch.empty(n_layers * GS_LAYER_ELEMS, dtype=torch.bfloat16, device=dev)
    for i in range(n_layers):
        oi, oo, o1, o2 = (int(v) for v in offs[i])
        w = torch.cat([flat[oi:oi + 196608], flat[oo:oo + 65536], flat[o1:o1 + 131072], flat[o2:o2 + 131072]]).to(torch.bfloat16)
        packed_fwd[i * GS_LAYER_ELEMS:(i + 1) * GS_LAYER_ELEMS] = w
        packed_bwd[i * GS_LAYER_ELEMS:(i + 1) * GS_LAYER_ELEMS] = w
    return packed_fwd, packed_bwd


def _gs_weights(img):
    return (img[:196608].view(768, 256), img[196608:262144].view(256, 256), img[262144:393216].view(512, 256),
            img[393216:524288].view(256, 512))


def gs_layer_fwd(x, packed_fwd_layer, in_bias, out_bias, b1, b2, gamma1, beta1, gamma2, beta2, key_mask, n_seq, S, scale,
                 eps=1e-5, drop_p=0.0, site0=0, seed=None, seq_add=None, train=False, seq_base=0, ffn_format=False, into=None):
    if seq_base or ffn_format or into is not None:
        # the sequences seq_base .. of a longer buffer: the draws are indexed from that buffer's first row, so the restatement
        # runs on a buffer of that length (zeros in front) and keeps its tail
        pad = lambda t, w_: None if t is None else torch.cat([torch.zeros((seq_base * w_,) + tuple(t.shape[1:]), dtype=t.dtype,
                                                                               device=t.device), t])
        km = None if key_mask is None else torch.cat([torch.ones(seq_base, dtype=key_mask.dtype, device=key_mask.device),
                                                      key_mask[:n_seq]])
        res = gs_layer_fwd(pad(x, S), packed_fwd_layer, in_bias, out_bias, b1, b2, gamma1, beta1, gamma2, beta2, km,
                           seq_base + n_seq, S, scale, eps, drop_p, site0, seed, seq_add=pad(seq_add, 1), train=train)
        if not train:
            return res[seq_base * S:]
        res = [t[seq_base * S:] for t in res]
        if ffn_format:
            x1f = _f(res[6])
            res[9] = ((x1f - res[7].unsqueeze(-1)) * res[8].unsqueeze(-1)).to(x.dtype)
            hp = torch.empty_like(res[10])
            hp[:, _ffn_frag_perm(x.device)] = res[10]
            res[10] = hp
        if into is not None:
            for dst, src in zip(into, res):
                dst.copy_(src)
            res = list(into)
        return tuple(res)
    win, wo, w1, w2 = _gs_weights(packed_fwd_layer)
    xn1, mean1, rstd1 = layernorm_fwd(x, gamma1, beta1, eps)
    qkv = gemm(xn1, win, bias=in_bias)
    ao = attention_fwd(qkv, key_mask, n_seq, S, 8, scale, drop_p, site0, seed)
    x1 = gemm(ao, wo, bias=out_bias, res=x, drop_p=drop_p, drop_site=site0 + 1, seed=seed)
    if seq_add is not None:
        bcast_add_fwd_(x1, seq_add, n_seq, S, drop_p, site0 + 2, seed)
    xn2, mean2, rstd2 = layernorm_fwd(x1, gamma2, beta2, eps)
    h = gemm(xn2, w1, bias=b1, act=RELU, drop_p=drop_p, drop_site=site0 + 3, seed=seed)
    x2 = gemm(h, w2, bias=b2, res=x1, drop_p=drop_p, drop_site=site0 + 4, seed=seed)
    if train:
        return x2, mean1, rstd1, xn1, qkv, ao, x1, mean2, rstd2, xn2, h
    return x2


def gs_layer_bwd(dx2, packed_bwd_layer, x, mean1, rstd1, qkv, x1, mean2, rstd2, h, gamma1, gamma2, key_mask, n_seq, S,
                 scale, drop_p=0.0, site0=0, seed=None, want_dx1=False, dgamma2=None, dbeta2=None, dgamma1=None,
                 dbeta1=None, want_dg=False):
    win, wo, w1, w2 = _gs_weights(packed_bwd_layer)
    inv_keep = keep_scale(drop_p)
    dym = drop_apply(dx2, drop_p, site0 + 4, seed)
    dpre = gemm(dym, w2, b_kc=False, gate=h, gate_scale=inv_keep)
    dxn2 = gemm(dpre, w1, b_kc=False)
    dx1, dg2, db2 = layernorm_bwd(dxn2, x1, mean2, rstd2, gamma2, res=dx2, dgamma=dgamma2, dbeta=dbeta2)
    dx1m = drop_apply(dx1, drop_p, site0 + 1, seed)
    dao = gemm(dx1m, wo, b_kc=False)
    dqkv = attention_bwd(qkv, key_mask, dao, n_seq, S, 8, scale, drop_p, site0, seed)
    dxn1 = gemm(dqkv, win, b_kc=False)
    dx, dg1, db1 = layernorm_bwd(dxn1, x, mean1, rstd1, gamma1, res=dx1, dgamma=dgamma1, dbeta=dbeta1)
    res = (dx, (dx1 if want_dx1 else None), dym, dpre, dx1m, dqkv, dg2, db2, dg1, db1)
    return res + (bcast_add_bwd(dx1, n_seq, S, drop_p, site0 + 2, seed),) if want_dg else res


def gs_stack_fwd(x, layers, key_mask, n_seq, S, scale, eps=1e-5, drop_p=0.0, seed=None, train=False):
    """dsvg_gs_stack_fwd = its layers one after the other (csrc/group_stage.hip gs_stack_fwd_kernel)"""
    outs, cur = [], x
    for i, L in enumerate(layers):
        res = gs_layer_fwd(cur, L["img"], L["in_bias"], L["out_bias"], L["b1"], L["b2"], L["gamma1"], L["beta1"], L["gamma2"],
                           L["beta2"], key_mask, n_seq, S, scale, eps, drop_p, L["site0"], seed, seq_add=L.get("seq_add"),
                           train=train)
        cur = res[0] if train else res
        outs.append(res if (train or i == len(layers) - 1) else None)
    return outs


def gs_stack_bwd(dx2, layers, key_mask, n_seq, S, scale, drop_p=0.0, seed=None, want_dg=False):
    n = len(layers)
    per, dgs, g = [None] * n, [None] * n, dx2
    for i in range(n - 1, -1, -1):
        L = layers[i]
        res = gs_layer_bwd(g, L["img"], L["x"], L["mean1"], L["rstd1"], L["qkv"], L["x1"], L["mean2"], L["rstd2"], L["h"],
                           L["gamma1"], L["gamma2"], key_mask, n_seq, S, scale, drop_p, L["site0"], seed, dgamma2=L["dgamma2"],
                           dbeta2=L["dbeta2"], dgamma1=L["dgamma1"], dbeta1=L["dbeta1"], want_dg=want_dg)
        g = res[0]
        per[i] = (res[2], res[3], res[4], res[5])
        if want_dg:
            dgs[i] = res[10]
    return g, per, (torch.cat(dgs, 1) if want_dg else None)


def keep_scale(p):
    if p <= 0:
        return 1.0
    t = min(int(p * 65536.0 + 0.5), 65535)
    return 65536.0 / (65536 - t)


def latent_chain_fwd(z, weights, biases, train=False):
    """csrc/group_stage.hip latent_chain_fwd_kernel: the unfused launches it replaces, with their bf16 roundings"""
    n_res = len(weights) - 1
    zs, rs = [], []
    cur = z
    for i in range(n_res):
        r = torch.relu(_f(cur) @ _f(weights[i]).t() + biases[i]).to(z.dtype)
        cur = (_f(cur) + _f(r)).to(z.dtype)
        zs.append(cur)
        rs.append(r)
    out = (_f(cur) @ _f(weights[n_res]).t() + biases[n_res]).to(z.dtype)
    return (out, zs, rs) if train else out


def latent_chain_bwd(dout, weights, rs):
    n_res = len(weights) - 1
    g = (_f(dout) @ _f(weights[n_res])).to(dout.dtype)
    dpre = [None] * n_res
    for i in range(n_res - 1, -1, -1):
        dp = torch.where(_f(rs[i]) > 0, _f(g), torch.zeros_like(_f(g))).to(dout.dtype)
        dpre[i] = dp
        g = (_f(g) + _f(dp) @ _f(weights[i])).to(dout.dtype)
    return g, dpre


def gate_mul(dy, y, scale=1.0):
    return torch.where(_f(y) > 0, _f(dy) * scale, torch.zeros_like(_f(dy))).to(dy.dtype)


def drop_apply(x, drop_p, drop_site, seed):
    if drop_p <= 0:
        return x
    idx = torch.arange(x.numel(), device=x.device, dtype=torch.int64).view(x.shape)
    return (_f(x) * drop_mult(drop_p, seed, drop_site, idx)).to(x.dtype)


def add(a, b):
    return (_f(a) + _f(b)).to(a.dtype)


def require_device(device):
    return None
