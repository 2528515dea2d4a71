"""Microbenchmark of the fused FFN kernels against the launches they replace (one MI355X, bf16, dropout 0.1).
Rows = the token counts of the BASELINE C2 step: 4096 (group stages), ~41k (packed encoder stage 1), ~71k (decoder
stage-2 backward prefix), 126,976 / 131,072 (decoder stage 2 / padded encoder)."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops          # noqa: E402

DEV = "cuda"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3       # us


def main():
    g = torch.Generator(device="cpu").manual_seed(0)
    L = 131072 + 512 + 131072 + 256 + 256 + 8
    flat = torch.zeros(8 + 2 * L)
    offs = []
    for i in range(2):
        o = 8 + i * L
        offs.append([o, o + 131072, o + 131072 + 512, o + 262144 + 512, o + 262144 + 768])
        flat[o:o + 131072] = torch.randn(131072, generator=g) * 0.06
        flat[o + 131072 + 512:o + 262144 + 512] = torch.randn(131072, generator=g) * 0.06
        flat[o + 262144 + 512:o + 262144 + 768] = 1.0
    flat = flat.to(DEV)
    offs = torch.tensor(offs, dtype=torch.int64, device=DEV)
    pf, pb, b1f = ops.ffn_pack(flat, offs, 2)
    t_pack = timeit(lambda: ops.ffn_pack(flat, offs, 2, pf, pb, b1f))
    print(f"ffn_pack (2 layers): {t_pack:.1f} us")
    W1 = flat[8:8 + 131072].view(512, 256).to(torch.bfloat16)
    W2 = flat[8 + 131072 + 512:8 + 262144 + 512].view(256, 512).to(torch.bfloat16)
    gamma = torch.ones(256, device=DEV)
    beta = torch.zeros(256, device=DEV)
    b1 = torch.zeros(512, device=DEV)
    b2 = torch.zeros(256, device=DEV)
    seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
    pl = pf[:ops.FFN_FWD_LAYER_ELEMS]
    quick = '--quick' in sys.argv
    if '--pmc' in sys.argv:      # a few plain launches for rocprofv3 --pmc (no timing loop): BASELINE C2's decoder stage 2
        rows = 126976
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        dy = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        w2p = torch.empty((2, 256, 512), dtype=torch.bfloat16, device=DEV)
        ops.ffn_pack(flat, offs, 2, pf, pb, b1f, w2p)
        for _ in range(3):
            ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed)
            y, h, xh, rstd = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=True)
            dym = ops.drop_apply(dy, 0.1, 4, seed)
            dpre = ops.gemm(dym, w2p[0], b_kc=False, gate=h, gate_scale=1.0 / 0.9)
            ops.ffn_bwd_dx(dpre, x, dy, pb[:ops.FFN_BWD_LAYER_ELEMS])
        torch.cuda.synchronize()
        return
    for rows in ((4096, 131072) if quick else (4096, 40960, 71680, 126976, 131072, 262144)):
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        y = torch.empty_like(x)
        flops = 4.0 * 256 * 512 * rows

        def unfused(p=0.1):
            xn, _, _ = ops.layernorm_fwd(x, gamma, beta)
            h = ops.gemm(xn, W1, bias=b1, act=ops.RELU, drop_p=p, drop_site=3, seed=seed)
            return ops.gemm(h, W2, bias=b2, res=x, drop_p=p, drop_site=4, seed=seed)
        tu = 1.0 if quick else timeit(unfused)
        line = f"rows {rows:7d}: unfused (LN + 2 GEMM) {tu:7.1f} us = {flops / tu * 1e-6:6.0f} TF/s |"
        for st in ((3,) if quick else (3, 4)):
            ops._FFN_STAGES = st
            for p in (0.1, 0.0):
                tf = timeit(lambda: ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, p, 3, 4, seed, out=y))
                line += f" fused s{st} p{p}: {tf:6.1f} us {flops / tf * 1e-6:5.0f} TF/s ({flops / tf * 1e-6 / 2500 * 100:4.1f} %) |"
        ops._FFN_STAGES = 0
        tt = timeit(lambda: ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, out=y, train=True))
        line += f" TRAIN (writes h, xh): {tt:6.1f} us {flops / tt * 1e-6:5.0f} TF/s |"
        print(line, flush=True)
        if quick:
            continue
        # ---- backward: fused (2 launches + 2 weight-gradient GEMMs + finish) vs the unfused sequence ----
        dy = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        pbl = pb[:ops.FFN_BWD_LAYER_ELEMS]
        W1m = flat[8:8 + 131072].view(512, 256)
        outs = [torch.empty(n, device=DEV) for n in (131072, 512, 131072, 256, 256)]
        g2p, g1p = torch.empty(256, 512, device=DEV), torch.empty(512, 256, device=DEV)
        db1p, db2 = torch.empty(512, device=DEV), torch.empty(256, device=DEV)
        s2, s1 = ops.split_k_for(256, 512, rows), ops.split_k_for(512, 256, rows)

        def fused_bwd_kernels(p=0.1):
            return ops.ffn_bwd(x, dy, pbl, b1f[0], 1e-5, p, 3, 4, seed)
        dx, hp, dpre, xh, dym = fused_bwd_kernels()

        def fused_wgrad():
            ops.gemm(dym, hp, a_kc=False, b_kc=False, out=g2p, split_k=s2, rowsum=db2)
            ops.gemm(dpre, xh, a_kc=False, b_kc=False, out=g1p, split_k=s1, rowsum=db1p)
            ops.ffn_wgrad_finish(g1p, db1p, g2p, W1m, gamma, beta, *outs)
        xn, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
        h = ops.gemm(xn, W1, bias=b1, act=ops.RELU, drop_p=0.1, drop_site=3, seed=seed)
        dw2, dw1 = torch.empty(256, 512, device=DEV), torch.empty(512, 256, device=DEV)
        dg, dbt = torch.empty(256, device=DEV), torch.empty(256, device=DEV)

        def unfused_bwd():
            dm = ops.drop_apply(dy, 0.1, 4, seed)
            ops.gemm(dm, h, a_kc=False, b_kc=False, out=dw2, split_k=s2, rowsum=db2)
            dh = ops.gemm(dm, W2, b_kc=False, gate=h, gate_scale=1.0 / 0.9)
            ops.gemm(dh, xn, a_kc=False, b_kc=False, out=dw1, split_k=s1, rowsum=db1p)
            dxn = ops.gemm(dh, W1, b_kc=False)
            ops.layernorm_bwd(dxn, x, mean, rstd, gamma, res=dy, dgamma=dg, dbeta=dbt)
        t1, t2, t3 = timeit(fused_bwd_kernels), timeit(fused_wgrad), timeit(unfused_bwd)
        fl = 8.0 * 256 * 512 * rows
        print(f"   backward rows {rows:7d}: fused dX kernels {t1:7.1f} us + wgrad GEMMs/finish {t2:7.1f} us = {t1 + t2:7.1f} us "
              f"({fl / (t1 + t2) * 1e-6:5.0f} TF/s algorithmic) | unfused {t3:7.1f} us ({fl / t3 * 1e-6:5.0f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
