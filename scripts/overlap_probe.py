"""Do two independent chains of fused-kernel launches on two HIP streams overlap on the device?  The question behind it: the
training forward of the second decoder stage runs ~71k rows = 279 workgroups of ffn_fwd (1.09 "rounds" of the 256 CUs, one
workgroup per CU), so every launch pays two rounds.  Split into a 65,536-row chain (exactly one round per launch) and a
5,9xx-row remainder chain on a second stream, the remainder's workgroups could fill the CUs the main chain's launches leave
idle at their edges - if the hardware interleaves the two queues.  Measures: one stream (main then remainder, per layer),
two streams eager, two streams inside one hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops          # noqa: E402

DEV = "cuda"
LAYERS = 4


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
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 71424
    main_rows = 65536
    x = torch.randn(total, 256, generator=g).to(DEV).to(torch.bfloat16)
    bufs = [[torch.empty_like(x) for _ in range(2)] for _ in range(2)]
    side = torch.cuda.Stream()

    def chain(lo, hi):
        cur = x[lo:hi]
        for i in range(LAYERS):
            out = bufs[0][i % 2][lo:hi]
            ops.ffn_fwd(cur, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, out=out)
            cur = out

    def whole():
        chain(0, total)

    def one_stream():
        cur_m, cur_r = x[:main_rows], x[main_rows:]
        for i in range(LAYERS):
            om, orr = bufs[0][i % 2][:main_rows], bufs[0][i % 2][main_rows:]
            ops.ffn_fwd(cur_m, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, out=om)
            ops.ffn_fwd(cur_r, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, out=orr)
            cur_m, cur_r = om, orr

    def two_streams():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            chain(main_rows, total)
        chain(0, main_rows)
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

    print(f"{LAYERS} chained ffn_fwd launches (inference variant), {total} rows = {-(-total // 256)} workgroups")
    print(f"  one launch per layer over all rows            : {timeit(whole):7.1f} us")
    print(f"  main {main_rows} + remainder, one stream        : {timeit(one_stream):7.1f} us")
    print(f"  main chain | remainder chain on two streams     : {timeit(two_streams):7.1f} us")
    print(f"  main chain alone                              : {timeit(lambda: chain(0, main_rows)):7.1f} us")
    print(f"  remainder chain alone                         : {timeit(lambda: chain(main_rows, total)):7.1f} us")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for fn, name in ((whole, "one launch per layer"), (two_streams, "two streams")):
            fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                fn()
            torch.cuda.synchronize()
            print(f"  hipGraph replay, {name:22s}        : {timeit(gr.replay):7.1f} us")


if __name__ == "__main__":
    main()
