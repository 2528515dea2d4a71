"""The fused-FFN backward's input path: dsvg_ffn_bwd_one (one launch) against the default three launches (drop_apply, gated
GEMM on W2p, ffn_bwd_dx) - launch times and the largest deviations of dpre / dx, dropout 0.1 and 0."""
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
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator(device="cpu").manual_seed(0)
    L = 131072 + 512 + 131072 + 256 + 256 + 8
    flat = torch.zeros(8 + L)
    o = 8
    offs = [[o, o + 131072, o + 131072 + 512, o + 262144 + 512, o + 262144 + 768]]
    flat[o:o + 131072] = torch.randn(131072, generator=g) * 0.06
    flat[o + 131072 + 512:o + 262144 + 512] = torch.randn(131072, generator=g) * 0.06
    flat[o + 262144 + 512:o + 262144 + 768] = 1.0
    flat = flat.to(DEV)
    offs = torch.tensor(offs, dtype=torch.int64, device=DEV)
    w2p = torch.empty((1, 256, 512), dtype=torch.bfloat16, device=DEV)
    pf, pb, b1f = ops.ffn_pack(flat, offs, 1, w2p=w2p)
    pl, pbl = pf[:ops.FFN_FWD_LAYER_ELEMS], pb[:ops.FFN_BWD_LAYER_ELEMS]
    b2 = torch.zeros(256, device=DEV)
    seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
    for rows in (4096, 40960, 63488, 126976):
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        dy = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        for p in (0.1, 0.0):
            ks = ops.keep_scale(p)
            y, h, xh, rstd = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, p, 3, 4, seed, train=True)

            def three():
                dym = ops.drop_apply(dy, p, 4, seed)
                dpre = ops.gemm(dym, w2p[0], b_kc=False, gate=h, gate_scale=ks)
                return ops.ffn_bwd_dx(dpre, x, dy, pbl), dpre, dym

            def one():
                return ops.ffn_bwd_one(dy, h, x, pbl, ks, 1e-5, p, 4, seed)
            a, b = three(), one()
            torch.cuda.synchronize()
            err = [((u.float() - v.float()).abs().max() / (u.float().abs().max() + 1e-12)).item() for u, v in zip(a, b)]
            t3, t1 = timeit(three), timeit(one)
            mb3 = rows * (512 * 2 + 512 * 2 + 1024 * 2 + 1024 + 512 * 3) / 1e6 if p > 0 else rows * (512 + 1024 * 2 + 1024 + 512 * 3) / 1e6
            mb1 = rows * (512 * 3 + 1024 * 2 + (512 if p > 0 else 0)) / 1e6
            print(f"rows {rows:6d} p {p}: three launches {t3:6.1f} us ({mb3 / t3:5.2f} TB/s of {mb3:5.0f} MB) | one launch "
                  f"{t1:6.1f} us ({mb1 / t1:5.2f} TB/s of {mb1:5.0f} MB) | max deviation / max magnitude: dx {err[0]:.2e} "
                  f"dpre {err[1]:.2e} dym {err[2]:.2e}", flush=True)


if __name__ == "__main__":
    main()
