"""ffn_fwd launch variants (`stages`: 2 half-size workgroups, 3 / 4 ring slots of the 256-row kernel, 5 the software-
pipelined chunk loop) against stages = 4: bit-equality of every output and the launch time, inference and training variants,
with and without dropout.  usage: python scripts/ffn_variant_probe.py [stages ...]   (default: 5)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402

DEV = "cuda"
VARIANTS = [int(a) for a in sys.argv[1:]] or [5]


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
    flat[o + 131072:o + 131072 + 512] = torch.randn(512, generator=g) * 0.1
    flat[o + 131072 + 512:o + 262144 + 512] = torch.randn(131072, generator=g) * 0.06
    flat[o + 262144 + 512:o + 262144 + 768] = 1.0 + 0.1 * torch.randn(256, generator=g)
    flat[o + 262144 + 768:o + 262144 + 1024] = 0.1 * torch.randn(256, generator=g)
    flat = flat.to(DEV)
    offs = torch.tensor(offs, dtype=torch.int64, device=DEV)
    pf, pb, b1f = ops.ffn_pack(flat, offs, 1)
    pl = pf[:ops.FFN_FWD_LAYER_ELEMS]
    b2 = torch.randn(256, generator=g).to(DEV) * 0.1
    seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
    for rows in (1000, 40960, 63488, 126976):
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        flops = 4.0 * 256 * 512 * rows
        for p in (0.1, 0.0):
            for train in (False, True):
                ref = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, p, 3, 4, seed, train=train, stages=4)
                ref = ref if train else (ref,)
                t4 = timeit(lambda: ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, p, 3, 4, seed, train=train, stages=4))
                line = (f"rows {rows:6d} p {p} {'train' if train else 'infer'}: stages 4 {t4:6.1f} us "
                        f"({flops / t4 * 1e-6 / 2500 * 100:4.1f} % of 2.5 PF)")
                for st in VARIANTS:
                    got = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, p, 3, 4, seed, train=train, stages=st)
                    torch.cuda.synchronize()
                    got = got if train else (got,)
                    same = all(torch.equal(a, b) for a, b in zip(ref, got))
                    worst = max((a.float() - b.float()).abs().max().item() for a, b in zip(ref, got))
                    ts = timeit(lambda: ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, p, 3, 4, seed, train=train, stages=st))
                    line += (f" | stages {st} {ts:6.1f} us ({flops / ts * 1e-6 / 2500 * 100:4.1f} %) "
                             f"equal {same}" + ("" if same else f" (max abs diff {worst:.3e})"))
                print(line, flush=True)


if __name__ == "__main__":
    main()
