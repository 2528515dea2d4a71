"""Round 6: dsvg_attn_bwd_dx (one launch) against the two launches it replaces (input-gradient GEMM + LayerNorm backward) at the
row counts of the training step (encoder stage ~41 k packed rows, decoder stage 63,488 visible rows).  Rows are rotated over
buffers larger than every cache; times are HIP-event means over `reps` launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsvg_amd import ops

dev = "cuda"
torch.manual_seed(0)
per = 768 * 256 + 256 * 256
flat = torch.zeros(8 + per, device=dev)
flat[8:] = torch.randn(per, device=dev) * 0.06
offs = torch.tensor([[8, 8 + 196608]], dtype=torch.int64, device=dev)
img = ops.attn_pack_bwd(flat, offs, 1)
win = flat[8:8 + 196608].view(768, 256).to(torch.bfloat16)
gamma = (1 + 0.1 * torch.randn(256, device=dev)).contiguous()
seed = torch.tensor([0x1234567], dtype=torch.int64, device=dev)
NB = 6


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


for rows in (16384, 32768, 41216, 63488, 126976):
    xs = [torch.randn(rows, 256, device=dev).to(torch.bfloat16) for _ in range(NB)]
    dq = [(torch.randn(rows, 768, device=dev) * 0.3).to(torch.bfloat16) for _ in range(NB)]
    rs = [torch.randn(rows, 256, device=dev).to(torch.bfloat16) for _ in range(NB)]
    st = [ops.layernorm_fwd(x, gamma, torch.zeros_like(gamma))[1:] for x in xs]
    dxo = torch.empty(rows, 256, device=dev, dtype=torch.bfloat16)
    for masked in (None, (0.1, 5, seed)):
        def fused(i):
            ops.attn_bwd_dx(dq[i], xs[i], st[i][0], st[i][1], gamma, rs[i], img, dx=dxo, masked=masked)

        def pair(i):
            d = ops.gemm(dq[i], win, b_kc=False)
            ops.layernorm_bwd(d, xs[i], st[i][0], st[i][1], gamma, res=rs[i], dx=dxo, masked=masked)

        def gemm_only(i):
            ops.gemm(dq[i], win, b_kc=False)

        tf, tp, tg = timeit(fused), timeit(pair), timeit(gemm_only)
        mb = rows * (1536 + 3 * 512 + (512 if masked else 0)) / 1e6
        print(f"rows {rows:7d} masked {masked is not None!s:5}: fused {tf:6.1f} us ({mb / tf * 1e-6 * 1e6 / 1e3:5.2f} TB/s of {mb:5.0f} MB)"
              f"   gemm + ln_bwd {tp:6.1f} us (gemm alone {tg:5.1f})   ratio {tf / tp:.2f}", flush=True)
