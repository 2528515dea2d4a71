"""Where does a LAUNCH of ffn_fwd spend its time, on one chip-wide time base?  s_memrealtime stamps (100 MHz, the same counter
for every XCD; dsvg_ffn_debug_clock with bit 0 of the buffer address set) at wave start, LayerNorm done, chunk loop done and
stores issued: the spread of the start stamps is the dispatch ramp of the grid, first start -> last end the span the waves
cover, and the launch period of back-to-back launches minus that span what the launch costs outside any wave's life (launch
gap, end-of-kernel write-back)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops, lib  # noqa: E402

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
    L_ = lib.load()
    rows_list = [int(r) for r in os.environ.get("PROBE_ROWS", "4096,16384,32768,40960,63488,65536,126976").split(",")]
    for rows in rows_list:
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        nwg = (rows + 255) // 256
        for train in (False, True):
            run = lambda: ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=4)
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            period = e0.elapsed_time(e1) / 50 * 1e3
            buf = torch.zeros(nwg * 8 * 4, dtype=torch.int64, device=DEV)
            lib.check(L_.dsvg_ffn_debug_clock(buf.data_ptr() | 1), "dbg")
            for _ in range(3):          # the third launch's stamps stay (back-to-back launches: the steady state)
                run()
            torch.cuda.synchronize()
            lib.check(L_.dsvg_ffn_debug_clock(None), "dbg")
            t = buf.view(nwg, 8, 4).double().cpu() * 0.01         # microseconds
            t0 = t[:, :, 0].min().item()
            start = t[:, :, 0] - t0
            end = t[:, :, 3] - t0
            wg_start = start.min(1).values
            wg_end = end.max(1).values
            life = (t[:, :, 3] - t[:, :, 0]).view(-1)
            ph = torch.stack([t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]], -1).view(-1, 3)
            q = lambda v, p: v.quantile(p).item()
            # the same launch stamped with s_memtime (the shader clock): ticks per phase / microseconds per phase = the clock the
            # waves actually ran at in that phase
            buf2 = torch.zeros(nwg * 8 * 4, dtype=torch.int64, device=DEV)
            lib.check(L_.dsvg_ffn_debug_clock(buf2.data_ptr()), "dbg")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            lib.check(L_.dsvg_ffn_debug_clock(None), "dbg")
            t2 = buf2.view(nwg, 8, 4).double().cpu()
            ph2 = torch.stack([t2[:, :, 1] - t2[:, :, 0], t2[:, :, 2] - t2[:, :, 1], t2[:, :, 3] - t2[:, :, 2]], -1).view(-1, 3)
            mhz = [ph2[:, i].median().item() / max(ph[:, i].median().item(), 1e-9) for i in range(3)]
            print(f"rows {rows:6d} {'train' if train else 'infer'}: shader clock by phase (s_memtime ticks / s_memrealtime us, medians): "
                  f"prologue {mhz[0]:5.0f} MHz, loop {mhz[1]:5.0f} MHz ({ph2[:, 1].median().item() / 16:5.0f} ticks per chunk), epilogue {mhz[2]:5.0f} MHz")
            print(f"rows {rows:6d} ({nwg} workgroups) {'train' if train else 'infer'}: launch period {period:5.1f} us | workgroup starts: "
                  f"median {q(wg_start, 0.5):4.1f}, 90 % {q(wg_start, 0.9):4.1f}, last {wg_start.max().item():4.1f} us after the first | "
                  f"a wave lives {q(life, 0.5):4.1f} us (prologue {q(ph[:, 0], 0.5):4.1f}, loop {q(ph[:, 1], 0.5):4.1f}, epilogue "
                  f"{q(ph[:, 2], 0.5):4.1f}) | last wave ends {wg_end.max().item():5.1f} us after the first start "
                  f"(median workgroup end {q(wg_end, 0.5):5.1f})")


if __name__ == "__main__":
    main()
