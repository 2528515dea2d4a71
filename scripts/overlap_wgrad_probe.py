"""Would the weight-gradient GEMMs gain from running on a second stream next to the token-stationary kernels of the
backward chain?  Main chain: 4 ffn_fwd launches (training variant, 65,536 rows: one workgroup per CU, prologue / MFMA loop /
epilogue in lockstep); side chain: 8 split-K weight-gradient GEMMs over the same rows (HBM-bound, one 128 KiB workgroup per
CU).  Measures each alone, both on one stream, both on two streams (eager and inside one hipGraph)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops          # noqa: E402

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
    b2 = torch.zeros(256, device=DEV)
    seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    x = torch.randn(T, 256, generator=g).to(DEV).to(torch.bfloat16)
    dh = torch.randn(T, 512, generator=g).to(DEV).to(torch.bfloat16)
    outs = [torch.empty_like(x) for _ in range(2)]
    dws = [torch.empty(512, 256, dtype=torch.float32, device=DEV) for _ in range(2)]
    side = torch.cuda.Stream()
    sk = ops.split_k_for(512, 256, T)

    def main_chain():
        cur = x
        for i in range(4):
            ops.ffn_fwd(cur, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, out=outs[i % 2], train=True)
            cur = outs[i % 2]

    def side_chain():
        for i in range(8):
            ops.gemm(dh, x, a_kc=False, b_kc=False, split_k=sk, out=dws[i % 2])

    def serial():
        main_chain()
        side_chain()

    def two_streams():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            side_chain()
        main_chain()
        cur.wait_stream(side)

    def timeit(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    print(f"{T} rows: 4 ffn_fwd (train) launches | 8 weight-gradient GEMMs 512 x 256 x T (split {sk})")
    print(f"  ffn chain alone              : {timeit(main_chain):7.1f} us")
    print(f"  weight-gradient chain alone  : {timeit(side_chain):7.1f} us")
    print(f"  both, one stream             : {timeit(serial):7.1f} us")
    print(f"  both, two streams            : {timeit(two_streams):7.1f} us")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for fn, name in ((serial, "one stream"), (two_streams, "two streams")):
            fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                fn()
            torch.cuda.synchronize()
            print(f"  hipGraph replay, {name:12s}: {timeit(gr.replay):7.1f} us")


if __name__ == "__main__":
    main()
