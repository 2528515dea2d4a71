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
    ap.add_argument("--impls", default="0", help="comma list of dsvg_gemm_desc.impl values to time side by side "
                                                 "(0 best, 2 register-staged, 4 / 3 LDS-DMA with 1 / 2 stages)")
    ap.add_argument("--vendor", type=int, default=0, help="1: also time torch.matmul (hipBLASLt) on the plain shapes")
    a = ap.parse_args()
    impls = [int(v) for v in a.impls.split(",")]
    cur = {"impl": 0}
    _gemm = ops.gemm

    def gemm(*args, **kw):
        return _gemm(*args, impl=cur["impl"], **kw)
    ops_gemm = gemm
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
    # weight gradient + bias gradient live back to back (as in the flat gradient buffer)
    dwb = {}
    for n, k in ((512, 256), (256, 512), (768, 256)):
        flat = torch.empty(n * k + n, device=DEV, dtype=torch.float32)
        dwb[n] = (flat[:n * k].view(n, k), flat[n * k:])
    cases += [
        ("fwd QKV   NT  T x768 x256", lambda: ops_gemm(x256, w[(768, 256)], bias=bias[768]), T, 768, 256),
        ("fwd out   NT+res+drop T x256x256", lambda: ops_gemm(x256, w[(256, 256)], bias=bias[256], res=res, drop_p=0.1, drop_site=1, seed=seed), T, 256, 256),
        ("fwd FFN1  NT+relu+drop T x512x256", lambda: ops_gemm(x256, w[(512, 256)], bias=bias[512], act=1, drop_p=0.1, drop_site=2, seed=seed), T, 512, 256),
        ("fwd FFN2  NT+res+drop T x256x512", lambda: ops_gemm(x512, w[(256, 512)], bias=bias[256], res=res, drop_p=0.1, drop_site=3, seed=seed), T, 256, 512),
        ("bwd dh    NN+gate T x512x256", lambda: ops_gemm(x256, w[(256, 512)], b_kc=False, gate=x512, gate_scale=1.1), T, 512, 256),
        ("bwd dxn2  NN  T x256x512", lambda: ops_gemm(x512, w[(512, 256)], b_kc=False), T, 256, 512),
        ("bwd dxn1  NN  T x256x768", lambda: ops_gemm(x768, w[(768, 256)], b_kc=False), T, 256, 768),
        ("bwd dW1+db TN  512x256 xT", lambda: ops_gemm(x512, x256, a_kc=False, b_kc=False, out=dwb[512][0], rowsum=dwb[512][1], split_k=ops.split_k_for(512, 256, T)), 512, 256, T),
        ("bwd dW2+db TN  256x512 xT", lambda: ops_gemm(x256, x512, a_kc=False, b_kc=False, out=dwb[256][0], rowsum=dwb[256][1], split_k=ops.split_k_for(256, 512, T)), 256, 512, T),
        ("bwd dWin+db TN 768x256 xT", lambda: ops_gemm(x768, x256, a_kc=False, b_kc=False, out=dwb[768][0], rowsum=dwb[768][1], split_k=ops.split_k_for(768, 256, T)), 768, 256, T),
    ] + [
        (f"bwd dW1   TN split={s:4d}", (lambda s=s: ops_gemm(x512, x256, a_kc=False, b_kc=False, out_dtype=torch.float32, split_k=s)), 512, 256, T)
        for s in (32, 64, 128, 256, 512)
    ] + [
        ("drop_apply T x256", lambda: ops.drop_apply(x256, 0.1, 5, seed), T, 256, 0),
        ("colsum T x512", lambda: ops.colsum(x512), T, 512, 0),
        ("colsum+drop T x256", lambda: ops.colsum(x256, drop_p=0.1, drop_site=3, seed=seed), T, 256, 0),
    ]
    print(f"dtype={a.dtype} T={T} impls={impls}")
    for name, fn, M, N, K in cases:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        if K:
            flop = 2.0 * M * N * K
            byts = (M * K + N * K + M * N) * B
            cols = []
            for im in impls:
                cur["impl"] = im
                us = timeit(fn, a.iters)
                cols.append(f"impl{im}: {us:8.1f} us {flop / us / 1e6:7.1f} TF {byts / us / 1e6:5.2f} TB/s")
            cur["impl"] = 0
            print(f"{name:36s} " + " | ".join(cols) + f"   (min-bytes {byts / 1e6:6.1f} MB)")
        else:
            us = timeit(fn, a.iters)
            byts = M * N * B
            print(f"{name:36s} {us:9.1f} us  {byts / us / 1e6:6.2f} TB/s")
    if a.vendor:
        # calibration only: what the vendor library (hipBLASLt through torch.matmul) reaches on the bare products
        for name, x, wt in (("vendor T x768x256", x256, w[(768, 256)]), ("vendor T x512x256", x256, w[(512, 256)]),
                            ("vendor T x256x512", x512, w[(256, 512)]), ("vendor T x256x256", x256, w[(256, 256)])):
            N, K = wt.shape
            out = torch.empty(T, N, device=DEV, dtype=dt)
            us = timeit(lambda: torch.matmul(x, wt.t(), out=out), a.iters)
            byts = (T * K + N * K + T * N) * B
            print(f"{name:36s} {us:9.1f} us {2.0 * T * N * K / us / 1e6:7.1f} TF {byts / us / 1e6:5.2f} TB/s")
        xt = x512.t()
        out = torch.empty(512, 256, device=DEV, dtype=dt)
        us = timeit(lambda: torch.matmul(xt, x256, out=out), a.iters)
        print(f"{'vendor dW 512x256xT':36s} {us:9.1f} us {2.0 * T * 512 * 256 / us / 1e6:7.1f} TF")


if __name__ == "__main__":
    main()
