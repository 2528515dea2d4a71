"""Round 6 probe: is the input-gradient GEMM of the attention half (dxn1 = dqkv . W_in: M = 63,488 rows, N = 256, K = 768; 128 x 128 tiles,
every workgroup streams its A strip AND its B strip through L2 -> LDS) bound by HBM (A read once: 97.5 MB) or by what the workgroups
pull through L2 -> LDS (992 workgroups x 393 KB = 390 MB)?  The same product with N = 128 (half the workgroups, same A traffic from HBM,
half the L2 -> LDS bytes) and with K = 384 (half of everything) tells.  Rows rotated over buffers larger than every cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
NB = 6


def timeit(fn, reps=30):
    for i in range(3):
        fn(i % NB)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % NB)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M in (63488, 41216):
    for (N, K) in ((256, 768), (128, 768), (256, 384), (512, 256), (256, 256), (128, 256)):
        a = [(torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NB)]
        w = (torch.randn(K, N, device=dev) * 0.05).to(torch.bfloat16)      # B mn-contiguous (b_kc=False), as the input-gradient products
        out = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
        us = timeit(lambda i: ops.gemm(a[i], w, b_kc=False, out=out[i]))
        hbm = (M * K + M * N) * 2 / 1e6
        wgs = -(-M // 128) * -(-N // 128)
        l2 = wgs * (128 * K + 128 * K) * 2 / 1e6
        print(f"M {M:6d} N {N:4d} K {K:4d}: {us:6.1f} us   HBM {hbm:6.1f} MB = {hbm / us:5.2f} TB/s   L2->LDS {l2:6.1f} MB = {l2 / us:5.2f} TB/s   "
              f"{2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s   ({wgs} workgroups)")
        del a, out
