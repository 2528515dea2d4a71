"""Where does a wave of ffn_bwd_dx spend its time?  8 stamps of the chip-wide 100 MHz counter per wave (dsvg_ffn_debug_clock with
bit 1 of the buffer address set, csrc/ffn_fused.hip bwd_stamp): start, first chunk ready, K loop done, x rows landed + statistics,
LayerNorm math done, residual rows landed, stores issued, masked pass done."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsvg_amd import ops, lib

dev = "cuda"
torch.manual_seed(0)
n = 512 * 256 + 512 + 256 * 512 + 256 + 256
flat = torch.randn(8 + n, device=dev) * 0.05
o = 8
offs = torch.tensor([[o, o + 131072, o + 131072 + 512, o + 131072 + 512 + 131072, o + 131072 + 512 + 131072 + 256]], dtype=torch.int64, device=dev)
flat[int(offs[0, 3]):int(offs[0, 3]) + 256] = 1.0
_, pb, _ = ops.ffn_pack(flat, offs, 1)[:3]
seed = torch.tensor([0x1234567], dtype=torch.int64, device=dev)
L = lib.load()
names = ["prologue -> first chunk ready", "K loop (16 chunks)", "x rows landed + statistics", "LayerNorm math", "residual rows landed",
         "stores issued", "masked pass"]
for rows in (41216, 63488):
    for masked in (None, (0.1, 5, seed)):
        x = torch.randn(rows, 256, device=dev).to(torch.bfloat16)
        dpre = (torch.randn(rows, 512, device=dev) * 0.3).to(torch.bfloat16)
        dy = torch.randn(rows, 256, device=dev).to(torch.bfloat16)
        nb = (rows + 255) // 256
        buf = torch.zeros(nb * 8 * 8 + 8, dtype=torch.int64, device=dev)
        junk = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        for rep in range(3):
            junk.fill_(1.0)
            ops.ffn_bwd_dx(dpre, x, dy, pb[:ops.FFN_BWD_LAYER_ELEMS], masked=masked)      # (code warm, data cold)
            junk.fill_(2.0)
            lib.check(L.dsvg_ffn_debug_clock(buf.data_ptr() | 2), "dbg")
            ops.ffn_bwd_dx(dpre, x, dy, pb[:ops.FFN_BWD_LAYER_ELEMS], masked=masked)
            lib.check(L.dsvg_ffn_debug_clock(None), "dbg")
            torch.cuda.synchronize()
        t = buf[:nb * 64].view(nb, 8, 8).double() * 0.01        # us
        t0 = t[:, :, 0].min()
        ph = (t[:, :, 1:] - t[:, :, :-1]).mean((0, 1))
        print(f"rows {rows} masked {masked is not None}: launch span {float(t[:, :, 7].max() - t0):.1f} us; wave starts spread "
              f"{float(t[:, :, 0].max() - t0):.1f} us; wave life mean {float((t[:, :, 7] - t[:, :, 0]).mean()):.1f} us", flush=True)
        for nm, v in zip(names, ph.tolist()):
            print(f"     {nm:34s} {v:6.1f} us")
