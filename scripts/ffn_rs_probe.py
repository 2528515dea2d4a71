"""Where does a 128-row workgroup of the role-specialised ffn_fwd (stages = 5) spend its time?  8 stamps per wave through
dsvg_ffn_debug_clock (see rs_stamp in csrc/ffn_fused.hip), once in shader cycles (s_memtime) and once on the chip-wide
100 MHz counter (s_memrealtime: comparable between waves and workgroups)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops, lib  # noqa: E402

DEV = "cuda"
NAMES = ["prologue", "iter 0-1", "iter 2-8", "iter 9-15", "iter 16-17", "dump + E_B", "epilogue"]


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
    for rows in (4096, 32768, 63488):
        x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
        nwg = (rows + 127) // 128
        for train in (False, True):
            for real in (0, 1):
                for _ in range(3):
                    ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=5)
                buf = torch.zeros(nwg * 8 * 16 + 2, dtype=torch.int64, device=DEV)
                lib.check(L_.dsvg_ffn_debug_clock(buf.data_ptr() + real), "dbg")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=5)
                e1.record()
                torch.cuda.synchronize()
                lib.check(L_.dsvg_ffn_debug_clock(None), "dbg")
                us = e0.elapsed_time(e1) * 1e3
                t16 = buf[:nwg * 128].view(nwg, 8, 16).double().cpu()
                t = t16[:, :, :8]
                unit = "x10 ns" if real else "cycles"
                for role, sl in (("matrix", slice(0, 4)), ("vector", slice(4, 8))):
                    tt = t[:, sl, :].reshape(-1, 8)
                    ph = tt[:, 1:] - tt[:, :-1]
                    med = ph.median(0).values
                    p90 = ph.quantile(0.9, 0)
                    tot = (tt[:, 7] - tt[:, 0]).median().item()
                    print(f"rows {rows:6d} ({nwg:4d} wg) {'train' if train else 'infer'} {role} [{unit}] launch {us:6.1f} us | total {tot:8.0f} | "
                          + " | ".join(f"{n} {m:7.0f} (p90 {q:7.0f})" for n, m, q in zip(NAMES, med.tolist(), p90.tolist())))
                if not real:    # inside iteration 8
                    for role, sl in (("matrix", slice(0, 4)), ("vector", slice(4, 8))):
                        tt = t16[:, sl, 8:14].reshape(-1, 6)
                        ph = (tt[:, 1:] - tt[:, :-1]).median(0).values.tolist()
                        print(f"        iteration 8, {role}: A -> at B {ph[0]:6.0f} | in B {ph[1]:6.0f} | B -> work issued {ph[2]:6.0f} | final waits {ph[3]:6.0f} | in A {ph[4]:6.0f}")
                if real:        # spread of the workgroups' start and end times over the launch
                    st, en = t[:, 0, 0], t[:, :, 7].max(1).values
                    t0 = st.min()
                    print(f"        workgroup start (us after the first): median {(st - t0).median().item() / 100:6.2f} max {(st - t0).max().item() / 100:6.2f};"
                          f" end: median {(en - t0).median().item() / 100:6.2f} max {(en - t0).max().item() / 100:6.2f}")


if __name__ == "__main__":
    main()
