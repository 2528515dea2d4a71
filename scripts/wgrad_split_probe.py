"""in_proj weight-gradient product (768 x 256 over T token rows): time against the split-K factor.  split_k_for picks 16 (12 tiles
x 16 = 192 workgroups, multiples of 8 keep the K slices grouped per XCD); 21 fills 252 of the 256 CUs but runs the 2-D schedule."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops

def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

g = torch.Generator().manual_seed(0)
shapes = ((768, 256), (512, 256), (256, 512), (256, 256))
for T in (63488, 40960):
    x = {n: (torch.randn(T, n, generator=g) * 0.5).cuda().bfloat16() for n in (256, 512, 768)}
    x2 = (torch.randn(T, 256, generator=g) * 0.5).cuda().bfloat16()
    for M, N in shapes:
        a, b = x[M], (x2 if N == 256 else x[512])
        out = torch.empty(M, N, device="cuda"); rs = torch.empty(M, device="cuda")
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        cur = ops.split_k_for(M, N, T)
        row = []
        for wgs in (192, 256, 384, 512, 640, 768, 1024):
            s_ = max(8, wgs // tiles // 8 * 8)
            us = timeit(lambda: ops.gemm(a, b, a_kc=False, b_kc=False, out=out, rowsum=rs, split_k=s_))
            row.append(f"{s_}x{tiles}={s_ * tiles}: {us:5.1f}")
        print(f"T={T} {M}x{N} (split_k_for -> {cur}): " + " | ".join(row) + "  us (product + reduction)")
