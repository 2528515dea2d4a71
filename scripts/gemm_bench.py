#!/usr/bin/env python
"""GEMM micro-benchmark on the hot-path shapes (hierarchical_ordered, 512 icons/GPU): times every layout/epilogue
variant of dsvg_gemm with HIP events and prints achieved TFLOP/s and the HBM-bytes lower bound."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--T", type=int, default=131072)
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    T = a.T
    B = 2 if dt == torch.bfloat16 else 4
    seed = torch.tensor([12345], dtype=torch.int64, device=DEV)
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*s):
        return (torch.randn(*s, generator=g) * 0.5).to(DEV).to(dt)

    cases = []
    # name, M, N, K, kwargs builder
    x256, x512, x768 = rnd(T, 256), rnd(T, 512), rnd(T, 768)
    w = {(n, k): rnd(n, k) for (n, k) in [(768, 256), (256, 256), (512, 256), (256, 512)]}
    bias = {n: torch.randn(n, device=DEV) for n in (256, 512, 768)}
    res = rnd(T, 256)
    cases += [
        ("fwd QKV   NT  T x768 x256", lambda: ops.gemm(x256, w[(768, 256)], bias=bias[768]), T, 768, 256),
        ("fwd out   NT+res+drop T x256x256", lambda: ops.gemm(x256, w[(256, 256)], bias=bias[256], res=res, drop_p=0.1, drop_site=1, seed=seed), T, 256, 256),
        ("fwd FFN1  NT+relu+drop T x512x256", lambda: ops.gemm(x256, w[(512, 256)], bias=bias[512], act=1, drop_p=0.1, drop_site=2, seed=seed), T, 512, 256),
        ("fwd FFN2  NT+res+drop T x256x512", lambda: ops.gemm(x512, w[(256, 512)], bias=bias[256], res=res, drop_p=0.1, drop_site=3, seed=seed), T, 256, 512),
        ("bwd dh    NN+adrop+gate T x512x256", lambda: ops.gemm(x256, w[(256, 512)], b_kc=False, a_drop_p=0.1, a_drop_site=3, seed=seed, gate=x512, gate_scale=1.1), T, 512, 256),
        ("bwd dxn2  NN  T x256x512", lambda: ops.gemm(x512, w[(512, 256)], b_kc=False), T, 256, 512),
        ("bwd dxn1  NN  T x256x768", lambda: ops.gemm(x768, w[(768, 256)], b_kc=False), T, 256, 768),
        ("bwd dW1   TN  512x256 xT", lambda: ops.gemm(x512, x256, a_kc=False, b_kc=False, out_dtype=torch.float32, split_k=ops.split_k_for(512, 256, T)), 512, 256, T),
        ("bwd dW2   TN+adrop 256x512 xT", lambda: ops.gemm(x256, x512, a_kc=False, b_kc=False, out_dtype=torch.float32, a_drop_p=0.1, a_drop_site=3, seed=seed, split_k=ops.split_k_for(256, 512, T)), 256, 512, T),
        ("bwd dWin  TN  768x256 xT", lambda: ops.gemm(x768, x256, a_kc=False, b_kc=False, out_dtype=torch.float32, split_k=ops.split_k_for(768, 256, T)), 768, 256, T),
    ] + [
        (f"bwd dW1   TN split={s:4d}", (lambda s=s: ops.gemm(x512, x256, a_kc=False, b_kc=False, out_dtype=torch.float32, split_k=s)), 512, 256, T)
        for s in (32, 64, 128, 256, 512)
    ] + [
        ("drop_apply T x256", lambda: ops.drop_apply(x256, 0.1, 5, seed), T, 256, 0),
        ("colsum T x512", lambda: ops.colsum(x512), T, 512, 0),
        ("colsum+drop T x256", lambda: ops.colsum(x256, drop_p=0.1, drop_site=3, seed=seed), T, 256, 0),
    ]
    print(f"dtype={a.dtype} T={T}")
    for name, fn, M, N, K in cases:
        if a.only and a.only not in name:
            continue
        us = timeit(fn, a.iters)
        if K:
            flop = 2.0 * M * N * K
            byts = (M * K + N * K + M * N) * B
            print(f"{name:40s} {us:9.1f} us  {flop / us / 1e6:8.1f} TF   min-bytes {byts / 1e6:7.1f} MB -> {byts / us / 1e6:6.2f} TB/s")
        else:
            byts = M * N * B
            print(f"{name:40s} {us:9.1f} us  {byts / us / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
