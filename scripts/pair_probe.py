"""Round 6 probe: what would ONE launch for a weight-gradient product and its independent sibling buy?  In a layer's backward pass four pairs
of launches read the same freshly written operand and do not depend on each other (dW_in | dxn1 = dqkv . W_in, dW2 | the gated dpre product,
dW1 | ffn_bwd_dx, dWo | attention backward).  Upper bound: the two launches back to back on one stream against the same two on two
streams (free to overlap), rows rotated over buffers larger than every cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
NB = 6
M = int(os.environ.get("PROBE_ROWS", "63488"))
dqkv = [(torch.randn(M, 768, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NB)]
xn1 = [(torch.randn(M, 256, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NB)]
win = (torch.randn(768, 256, device=dev) * 0.05).to(torch.bfloat16)
dxn1 = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
dw = torch.empty(768, 256, device=dev)
db = torch.empty(768, device=dev)
split = ops.split_k_for(768, 256, M)
s2 = torch.cuda.Stream()


def wgrad(i):
    ops.gemm(dqkv[i], xn1[i], a_kc=False, b_kc=False, out=dw, split_k=split, rowsum=db)


def dx(i):
    ops.gemm(dqkv[i], win, b_kc=False, out=dxn1[i])


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


def both_two_streams(i):
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        wgrad(i)
    dx(i)
    cur.wait_stream(s2)


a, b = timeit(wgrad), timeit(dx)
c = timeit(lambda i: (wgrad(i), dx(i)))
d = timeit(both_two_streams)
print(f"rows {M}: dW_in (split {split}) {a:5.1f} us, dxn1 {b:5.1f} us, back to back {c:5.1f} us, on two streams {d:5.1f} us")
