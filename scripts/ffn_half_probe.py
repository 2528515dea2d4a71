"""ffn_fwd with half-size workgroups (stages=2: 128 rows, two workgroups per CU, the second one of a CU delayed) against the
256-row kernel: bit-equality of every output and the launch time, inference and training variants."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402

DEV = "cuda"


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
    pf, pb, b1f = ops.ffn_pack(flat, offs, 1)
    pl = pf[:ops.FFN_FWD_LAYER_ELEMS]
    b2 = torch.randn(256, generator=g).to(DEV) * 0.1
    seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("DSVG_FFN_HALF"))
    print(f"[{tag}]")
    for rows in (1000, 65536, 63488):
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        for train in (False, True):
            ref = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=4)
            got = ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=2)
            torch.cuda.synchronize()
            ref, got = (ref, got) if train else ((ref,), (got,))
            same = all(torch.equal(a, b) for a, b in zip(ref, got))
            ts = []
            for st in (4, 2):
                for _ in range(3):
                    ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=st)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            print(f"rows {rows:6d} {'train' if train else 'infer'}: equal {same} | 256-row kernel {ts[0]:6.1f} us, half-size {ts[1]:6.1f} us")


if __name__ == "__main__":
    main()
