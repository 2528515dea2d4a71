"""Is a split-K weight-gradient product bound by HBM bytes or by the bytes its tiles pull through L2 -> LDS?  dW[a, b] = A^T B over
T = 63,488 token rows with a + b = 1024 fixed (the same 130 MB from HBM) and different tile multiplicities: a 128 x 128 tile
schedule reads A ceil(b / 128) times and B ceil(a / 128) times.  (round 5)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


T = 63488
g = torch.Generator().manual_seed(0)
junk = torch.empty(160 * 1024 * 1024, dtype=torch.float32, device="cuda")
for a, b in ((768, 256), (512, 512), (896, 128), (128, 896), (256, 768), (1024, 128) if False else (640, 384)):
    A = (torch.randn(T, a, generator=g) * 0.5).cuda().bfloat16()
    B = (torch.randn(T, b, generator=g) * 0.5).cuda().bfloat16()
    out = torch.empty(a, b, device="cuda")
    sk = ops.split_k_for(a, b, T)
    tiles = ((a + 127) // 128) * ((b + 127) // 128)

    def run():
        junk.fill_(0.0)         # operands out of L2 / MALL: every product reads them from HBM
        ops.gemm(A, B, a_kc=False, b_kc=False, out=out, split_k=sk)
    t_fill = timeit(lambda: junk.fill_(0.0))
    us = timeit(run) - t_fill
    hbm = T * 2 * (a + b) / 1e6
    l2 = T * 2 * (a * ((b + 127) // 128) + b * ((a + 127) // 128)) / 1e6
    print(f"{a:4d} x {b:4d}: {us:6.1f} us (product + reduction, split_k {sk}, {tiles} tiles) | HBM {hbm:5.0f} MB -> {hbm / us:5.2f} TB/s | "
          f"L2->LDS {l2:5.0f} MB -> {l2 / us:5.2f} TB/s")
