"""Where does a workgroup of ffn_fwd spend its time?  s_memtime stamps (dsvg_ffn_debug_clock) at kernel start, LayerNorm done,
chunk loop done and stores issued, per wave; printed as the median and the spread over the workgroups of one launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops, lib  # noqa: E402

DEV = "cuda"
STAGES = int(os.environ.get("PROBE_STAGES", "0")) or None      # 2: the half-size workgroups (4 waves each)


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
    L_ = lib.load()
    for rows in (63488, 71424, 126976):
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        nwg = (rows + 255) // 256
        for train in (False, True):
            for _ in range(3):
                ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=STAGES)
            buf = torch.zeros(nwg * 8 * 4, dtype=torch.int64, device=DEV)
            lib.check(L_.dsvg_ffn_debug_clock(buf.data_ptr()), "dbg")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=STAGES)
            e1.record()
            torch.cuda.synchronize()
            lib.check(L_.dsvg_ffn_debug_clock(None), "dbg")
            t = buf.view(nwg * 8, 4).double().cpu()
            t = t[(t > 0).all(1)].view(-1, 1, 4)            # (stamps of waves that ran the whole kernel)
            if os.environ.get("PROBE_RAW"):
                print(buf.view(nwg, 8, 4)[0].tolist())
            # (every XCD has a counter of its own: only differences inside one wave mean anything)
            ph = torch.stack([t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]], -1).view(-1, 3)
            total = (t[:, :, 3] - t[:, :, 0]).view(-1)
            us = e0.elapsed_time(e1) * 1e3
            if nwg <= 256 and not train:        # one round: a workgroup's life ~ the launch -> ticks per microsecond
                main.tick = total.median().item() / us
            if os.environ.get("PROBE_TICK"):        # ticks per microsecond measured by an earlier run of the 256-row kernel
                main.tick = float(os.environ["PROBE_TICK"])
            k = 1.0 / getattr(main, "tick", 1.0)
            med = ph.median(0).values * k
            p90 = ph.quantile(0.9, 0) * k
            print(f"rows {rows:6d} ({nwg} workgroups) {'train' if train else 'infer'}: launch {us:6.1f} us | per wave, median "
                  f"(90th pct): prologue {med[0]:5.1f} ({p90[0]:5.1f}) us, chunk loop {med[1]:5.1f} ({p90[1]:5.1f}) us, epilogue "
                  f"{med[2]:5.1f} ({p90[2]:5.1f}) us, total {total.median().item() * k:5.1f} us")
            if os.environ.get("PROBE_SPLIT"):       # the faster and the slower half of the waves, by the length of the prologue
                order = torch.argsort(ph[:, 0])
                for name, idx in (("early half", order[:order.numel() // 2]), ("late half ", order[order.numel() // 2:])):
                    m = ph[idx].median(0).values * k
                    print(f"        {name}: prologue {m[0]:5.1f} us, chunk loop {m[1]:5.1f} us, epilogue {m[2]:5.1f} us")


if __name__ == "__main__":
    main()
