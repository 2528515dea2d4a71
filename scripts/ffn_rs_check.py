"""Role-specialised fused FFN forward (dsvg_ffn_fwd stages = 5) against the 256-row kernel (stages = 4) and the fp32
restatement (tests/torch_ops_ref.py), then a timing table of both at the row counts of the BASELINE C2 step.
Run on the GPU box:  timeout 300 python scripts/ffn_rs_check.py [--quick]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepsvg_amd import ops                      # noqa: E402
from tests import torch_ops_ref as R             # noqa: E402
from tests.test_kernels_gpu import _ffn_setup, _seed_tensor   # noqa: E402

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


def ulp_report(name, a, b):
    """a, b bf16 tensors: fraction of elements that differ, and by how many bf16 steps at most"""
    ai, bi = a.view(torch.int16).to(torch.int32), b.view(torch.int16).to(torch.int32)
    # order-preserving integer image of a bf16 pattern
    ai = torch.where(ai < 0, -32768 - ai, ai)
    bi = torch.where(bi < 0, -32768 - bi, bi)
    d = (ai - bi).abs()
    print(f"    {name}: {100.0 * (d != 0).float().mean().item():.4f} % of the elements differ, max {int(d.max().item())} bf16 steps, "
          f"max abs {float((a.float() - b.float()).abs().max()):.4e}")
    return d


def check(rows, drop_p):
    flat, offs, x, b2 = _ffn_setup(rows, seed=rows)
    pf, pb, b1f = ops.ffn_pack(flat, offs, 2)
    epf, _, eb1f = R.ffn_pack(flat, offs, 2)
    seed = _seed_tensor(0x0123456789ABCDEF)
    pl = pf[:ops.FFN_FWD_LAYER_ELEMS]
    ok = True
    y4 = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, drop_p, 403, 404, seed, stages=4)
    y5 = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, drop_p, 403, 404, seed, stages=5)
    torch.cuda.synchronize()
    want = R.ffn_fwd(x, epf[:ops.FFN_FWD_LAYER_ELEMS], eb1f[0], b2, 1e-5, drop_p, 403, 404, seed)
    print(f"rows {rows} p {drop_p}: inference")
    d = ulp_report("y rs vs 256-row kernel", y5, y4)
    e4 = (y4.float() - want.float()).abs().max().item()
    e5 = (y5.float() - want.float()).abs().max().item()
    scale = want.float().abs().max().item()
    print(f"    max err vs restatement: rs {e5:.3e}, 256-row {e4:.3e} (scale {scale:.3e})")
    ok &= e5 <= 1.5e-2 * scale
    t4 = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, drop_p, 403, 404, seed, train=True, stages=4)
    t5 = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, drop_p, 403, 404, seed, train=True, stages=5)
    torch.cuda.synchronize()
    print(f"rows {rows} p {drop_p}: training")
    ok &= bool(torch.equal(t5[0], y5))
    print(f"    y(train) == y(inference): {torch.equal(t5[0], y5)}")
    ulp_report("h", t5[1], t4[1])
    gate_same = ((t5[1] != 0) == (t4[1] != 0)).float().mean().item()
    print(f"    gates equal on {100 * gate_same:.4f} %")
    ok &= gate_same > 0.998
    print(f"    xh bit-equal {torch.equal(t5[2], t4[2])}, rstd bit-equal {torch.equal(t5[3], t4[3])}")
    ok &= bool(torch.equal(t5[2], t4[2])) and bool(torch.equal(t5[3], t4[3]))
    if drop_p > 0:      # the dropped positions of h are the same elements (same draws)
        both_pos = (t5[1] != 0) | (t4[1] != 0)
        mism = ((t5[1] != 0) != (t4[1] != 0)) & both_pos
        print(f"    gate mismatches {int(mism.sum().item())} of {both_pos.numel()}")
    return ok


def main():
    quick = "--quick" in sys.argv
    ok = True
    for rows in (() if "--timing-only" in sys.argv else (256, 1000) if quick else (100, 128, 256, 1000, 4096 + 37, 40000)):
        for p in (0.0, 0.1):
            ok &= check(rows, p)
    print("CORRECTNESS", "OK" if ok else "FAILED")
    flat, offs, _, b2 = _ffn_setup(8, seed=1)
    pf, pb, b1f = ops.ffn_pack(flat, offs, 2)
    pl = pf[:ops.FFN_FWD_LAYER_ELEMS]
    seed = _seed_tensor(77)
    g = torch.Generator(device="cpu").manual_seed(0)
    for rows in ((63488,) if quick else (4096, 16384, 32768, 40960, 63488, 126976, 131072)):
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        y = torch.empty_like(x)
        h = torch.empty((rows, 512), dtype=torch.bfloat16, device=DEV)
        xh = torch.empty_like(x)
        flops = 4.0 * 256 * 512 * rows
        line = f"rows {rows:7d}:"
        for train in (False, True):
            for p in (0.1, 0.0):
                for st in (0, 4, 6, 5):
                    if train:
                        t = timeit(lambda: ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, p, 3, 4, seed, out=y, train=True, into=(h, xh), stages=st))
                    else:
                        t = timeit(lambda: ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, p, 3, 4, seed, out=y, stages=st))
                    line += f" {'train' if train else 'infer'} p={p} st={st}: {t:6.1f} us ({flops / t * 1e-6 / 2500:.3f})"
            line += "\n             "
        print(line)


if __name__ == "__main__":
    main()
