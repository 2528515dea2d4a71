"""Can the weight-gradient GEMMs of the large stages hide beside the group stages' layer kernels?  A group-stage launch is
128 workgroups (one 32-row tile each, bound by the weight stream per CU): half of the CUs and most of the HBM bandwidth idle
for ~35 us per layer and direction.  Main chain: 4 x gs_layer_bwd (4096 rows); side chain: N split-K weight-gradient GEMMs
over 65,536 rows (256 workgroups, one per CU, HBM-bound).  Each alone, both on one stream, both on two streams - eager and
inside one hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402
from tests.test_group_stage_gpu import _setup, _params, _seed_tensor  # noqa: E402

DEV = "cuda"


def main():
    n_w = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n_seq, S = 512, 8
    flat, offs, p, x, key_mask, seq_add, dx2 = _setup(n_seq, S, seed=1, n_layers=1, masked=True, with_add=True)
    pf, pb = ops.gs_pack(flat, offs, 1)
    seed = _seed_tensor(77)
    scale, s0, dp = 32 ** -0.5, 208, 0.1
    sv = ops.gs_layer_fwd(x, pf, *_params(p), key_mask, n_seq, S, scale, 1e-5, dp, s0, seed, seq_add=seq_add, train=True)
    (_x2, mean1, rstd1, _xn1, qkv, _ao, x1, mean2, rstd2, _xn2, h) = sv
    g = torch.Generator(device="cpu").manual_seed(0)
    T = 65536
    xa = torch.randn(T, 256, generator=g).to(DEV).to(torch.bfloat16)
    dh = torch.randn(T, 512, generator=g).to(DEV).to(torch.bfloat16)
    dws = [torch.empty(512, 256, dtype=torch.float32, device=DEV) for _ in range(2)]
    sk = ops.split_k_for(512, 256, T)
    side = torch.cuda.Stream()

    def main_chain():
        for _ in range(4):
            ops.gs_layer_bwd(dx2, pb, x, mean1, rstd1, qkv, x1, mean2, rstd2, h, p["gamma1"], p["gamma2"], key_mask, n_seq, S,
                             scale, dp, s0, seed, want_dx1=True)

    def side_chain():
        for i in range(n_w):
            ops.gemm(dh, xa, a_kc=False, b_kc=False, split_k=sk, out=dws[i % 2])

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

    print(f"4 gs_layer_bwd launches (4096 rows) | {n_w} weight-gradient GEMMs 512 x 256 x {T} (split {sk})")
    print(f"  group-stage chain alone      : {timeit(main_chain):7.1f} us")
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
