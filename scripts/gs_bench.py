#!/usr/bin/env python
"""Micro-benchmark of the fused group-stage layer kernels (csrc/group_stage.hip) against the launches they replace.
    python scripts/gs_bench.py [n_seq] [S]      (default 512 sequences of 8: the benchmark's 4096-row stages)
DSVG_GS_PF=8|12|16|24 selects the prefetch depth of the weight stream."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402
from tests.test_group_stage_gpu import _setup, _params, _unfused_fwd, _unfused_bwd, _seed_tensor  # noqa: E402


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps // 10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps // 10 * 10) * 1e6


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    flat, offs, p, x, key_mask, seq_add, dx2 = _setup(n_seq, S, seed=1, n_layers=1, masked=True, with_add=True)
    pf, pb = ops.gs_pack(flat, offs, 1)
    oi, oo, o1, o2 = (int(v) for v in offs[0])
    bf = lambda t: t.to(torch.bfloat16).contiguous()
    W = (bf(flat[oi:oi + 196608].view(768, 256)), bf(flat[oo:oo + 65536].view(256, 256)),
         bf(flat[o1:o1 + 131072].view(512, 256)), bf(flat[o2:o2 + 131072].view(256, 512)))
    seed = _seed_tensor(77)
    scale, s0, dp = 32 ** -0.5, 208, 0.1
    fwd = lambda train: ops.gs_layer_fwd(x, pf, *_params(p), key_mask, n_seq, S, scale, 1e-5, dp, s0, seed, seq_add=seq_add,
                                         train=train)
    sv = fwd(True)
    (_x2, mean1, rstd1, _xn1, qkv, _ao, x1, mean2, rstd2, _xn2, h) = sv
    bwd = lambda: ops.gs_layer_bwd(dx2, pb, x, mean1, rstd1, qkv, x1, mean2, rstd2, h, p["gamma1"], p["gamma2"], key_mask, n_seq,
                                   S, scale, dp, s0, seed, want_dx1=True)
    flops = 2.0 * n_seq * S * 256 * 2048
    t = {"fused fwd (inference)": timeit(lambda: fwd(False)), "fused fwd (training)": timeit(lambda: fwd(True)),
         "fused bwd": timeit(bwd),
         "unfused fwd (10 launches)": timeit(lambda: _unfused_fwd(x, W, p, key_mask, n_seq, S, scale, dp, s0, seed, seq_add)),
         "unfused bwd (11 launches, no dW)": timeit(lambda: _unfused_bwd(dx2, W, p, (sv, x), key_mask, n_seq, S, scale, dp, s0, seed))}
    print(f"rows {n_seq * S} (S = {S}), DSVG_GS_PF = {os.environ.get('DSVG_GS_PF', 'default')}")
    for k, v in t.items():
        print(f"  {k:36s} {v:8.1f} us   ({flops * (2 if 'bwd' in k else 1) / v * 1e-6:7.1f} TFLOP/s, weights "
              f"{1.048576 / v * 1e3 * 128:6.1f} GB/s summed over 128 workgroups)")


if __name__ == "__main__":
    main()
