"""Where does a workgroup of attn_bwd_dx spend its time?  Stamps of the chip-wide 100 MHz counter (dsvg_attn_bwd_dx_debug_clock)
at wave start, first rows staged, K loop done, pass 1 done, sums published, stores issued; per launch size the mean phase lengths
over all waves, the spread of the wave START times (dispatch ramp) and of the END times (tail)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsvg_amd import ops, lib

dev = "cuda"
torch.manual_seed(0)
per = 768 * 256 + 256 * 256
flat = torch.zeros(8 + per, device=dev); flat[8:] = torch.randn(per, device=dev) * 0.06
offs = torch.tensor([[8, 8 + 196608]], dtype=torch.int64, device=dev)
img = ops.attn_pack_bwd(flat, offs, 1)
gamma = (1 + 0.1 * torch.randn(256, device=dev)).contiguous()
seed = torch.tensor([0x1234567], dtype=torch.int64, device=dev)
L = lib.load()
names = ["prologue (first rows landed)", "K loop (24 steps)", "barrier + pass 1", "publish sums", "pass 2 + stores"]
for rows in (16384, 32768, 41216, 63488):
    for masked in (None, (0.1, 5, seed)):
        x = torch.randn(rows, 256, device=dev).to(torch.bfloat16)
        dq = (torch.randn(rows, 768, device=dev) * 0.3).to(torch.bfloat16)
        rs = torch.randn(rows, 256, device=dev).to(torch.bfloat16)
        _, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros_like(gamma))
        nb = (rows + 127) // 128
        buf = torch.zeros(nb * 4 * 8, dtype=torch.int64, device=dev)
        junk = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        for _ in range(2):
            junk.fill_(1.0)         # evict the caches
            lib.check(L.dsvg_attn_bwd_dx_debug_clock(buf.data_ptr()), "dbg")
            ops.attn_bwd_dx(dq, x, mean, rstd, gamma, rs, img, masked=masked)
            lib.check(L.dsvg_attn_bwd_dx_debug_clock(None), "dbg")
            torch.cuda.synchronize()
        t = buf.view(nb, 4, 8)[:, :, :6].double() * 0.01        # us
        t0 = t[:, :, 0].min()
        ph = (t[:, :, 1:] - t[:, :, :-1]).mean((0, 1))
        print(f"rows {rows} masked {masked is not None}: launch span {float(t[:, :, 5].max() - t0):.1f} us; wave starts spread "
              f"{float(t[:, :, 0].max() - t0):.1f} us; wave life mean {float((t[:, :, 5] - t[:, :, 0]).mean()):.1f} us", flush=True)
        for n, v in zip(names, ph.tolist()):
            print(f"     {n:32s} {v:6.1f} us")
        # second-half workgroups (the co-resident ones) vs first half
        if nb > 256:
            for nm, sl in (("blocks < 256", slice(0, 256)), ("blocks >= 256", slice(256, nb))):
                tt = t[sl]
                print(f"     {nm}: start {float(tt[:, :, 0].mean() - t0):.1f}, loop done {float(tt[:, :, 2].mean() - t0):.1f}, end {float(tt[:, :, 5].mean() - t0):.1f} us after the first wave")
