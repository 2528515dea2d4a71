"""Microbenchmark of the fused attention sub-block (csrc/attn_fused.hip) against the four launches it replaces, at the
benchmark's shapes: 4096 sequences x 31 rows dense (decoder stage 2) and a packed encoder layout of ~110k rows."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    flat = (torch.randn(8 + 262144, generator=g) * 0.06).to(DEV)
    offs = torch.tensor([[8, 8 + 196608]], dtype=torch.int64, device=DEV)
    img = ops.attn_pack(flat, offs, 1)
    win = flat[8:8 + 196608].view(768, 256).to(torch.bfloat16)
    wo = flat[8 + 196608:8 + 262144].view(256, 256).to(torch.bfloat16)
    bi, bo = torch.zeros(768, device=DEV), torch.zeros(256, device=DEV)
    ga, be = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    seed = torch.tensor([12345], dtype=torch.int64, device=DEV)
    scale = 32 ** -0.5
    if "--pmc-unfused" in sys.argv:     # the four launches the fused kernel replaces, same shape (FETCH / WRITE passes)
        n_seq, S = 4096, 31
        x = (torch.randn(n_seq * S, 256, generator=g) * 1.5).to(DEV).to(torch.bfloat16)
        for _ in range(3):
            xn, m, r = ops.layernorm_fwd(x, ga, be, 1e-5)
            qkv = ops.gemm(xn, win, bias=bi)
            ao = ops.attention_fwd(qkv, None, n_seq, S, 8, scale, 0.1, 7, seed)
            ops.gemm(ao, wo, bias=bo, res=x, drop_p=0.1, drop_site=8, seed=seed)
        torch.cuda.synchronize()
        return
    if "--pmc" in sys.argv:         # counter collection (scripts/gpu_attn_pmc.sh): three launches of each variant, no timing
        n_seq, S = 4096, 31
        x = (torch.randn(n_seq * S, 256, generator=g) * 1.5).to(DEV).to(torch.bfloat16)
        for _ in range(3):
            ops.attn_block_fwd(x, img, bi, bo, ga, be, None, n_seq, S, scale, 1e-5, 0.1, 7, 8, seed, train=True)
            ops.attn_block_fwd(x, img, bi, bo, ga, be, None, n_seq, S, scale, 1e-5, 0.1, 7, 8, seed, train=False)
        torch.cuda.synchronize()
        return
    for name in ("dense31", "packed"):
        if name == "dense31":
            n_seq, S = 4096, 31
            rows, seq_off, tiles = n_seq * S, None, None
        else:
            n_seq, S = 4096, 30
            lens = torch.randint(8, 31, (n_seq,), generator=g)
            off = torch.zeros(n_seq + 1, dtype=torch.int32)
            off[1:] = lens.cumsum(0)
            rows = (int(off[-1]) + 1023) // 1024 * 1024
            seq_off = off.to(DEV)
            tiles = ops.attention_tiles(seq_off, n_seq, 32)
        x = (torch.randn(rows, 256, generator=g) * 1.5).to(DEV).to(torch.bfloat16)
        for p in (0.0, 0.1):
            def unfused():
                xn, m, r = ops.layernorm_fwd(x, ga, be, 1e-5)
                qkv = ops.gemm(xn, win, bias=bi)
                ao = ops.attention_fwd(qkv, None, n_seq, S, 8, scale, p, 7, seed, seq_off=seq_off, tiles=tiles)
                return ops.gemm(ao, wo, bias=bo, res=x, drop_p=p, drop_site=8, seed=seed)

            def fused(train):
                return ops.attn_block_fwd(x, img, bi, bo, ga, be, None, n_seq, S, scale, 1e-5, p, 7, 8, seed,
                                          seq_off=seq_off, tiles=tiles, train=train)
            tu, tt, ti = timeit(unfused), timeit(lambda: fused(True)), timeit(lambda: fused(False))
            fl = 2.0 * rows * 256 * 1024 + 4.0 * rows * 32 * 256
            print(f"{name} rows={rows} p={p}: unfused 4 launches {tu:7.1f} us | fused train {tt:7.1f} us "
                  f"({fl / tt * 1e-6:6.1f} TFLOP/s) | fused inference {ti:7.1f} us ({fl / ti * 1e-6:6.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
