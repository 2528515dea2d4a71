"""Round 6: where does a workgroup of gs_stack_fwd (ONE launch for the 4 layers of a group stage, 128 workgroups for 4096 rows) spend
a layer?  s_memtime stamps (100 MHz) per wave at the phase boundaries of ONE layer of the stack (DSVG_GS_DBG_LAYER), medians over
the launch's waves; the per-layer launch beside it.  Usage: DSVG_GS_DBG_LAYER=k python scripts/gs_stack_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops, lib  # noqa: E402
from tests.test_group_stage_gpu import _stack_setup, _params, _seed_tensor  # noqa: E402

NAMES = ["loads + LayerNorm 1", "in_proj + attention", "out_proj + LayerNorm 2", "linear1", "linear2", "x2 store"]


def report(label, buf, nwg, us):
    t = buf.view(nwg * 8, 8)[:, :7].double().cpu()
    d = (t[:, 1:] - t[:, :-1]) / 100.0
    med, p90 = d.median(0).values, d.quantile(0.9, 0)
    tot = ((t[:, 6] - t[:, 0]) / 100.0).median().item()
    print(f"{label}: launch {us:6.1f} us; the stamped layer per wave {tot:5.1f} us")
    for i, nme in enumerate(NAMES):
        print(f"    {nme:24s} {med[i].item():5.1f} us (90th pct {p90[i].item():5.1f})")


def main():
    n_seq, S, n = 512, 8, 4
    flat, offs, ps, x, key_mask, gcat, dx2 = _stack_setup(n_seq, S, n, True, True, seed=3)
    pf, _pb = ops.gs_pack(flat, offs, n)
    seed = _seed_tensor(77)
    scale, dp = 32 ** -0.5, 0.1
    E = ops.GS_LAYER_ELEMS
    L_ = lib.load()
    nwg = n_seq * S // 32
    layers = [dict(img=pf[i * E:(i + 1) * E], site0=8 * i, seq_add=gcat[:, 256 * i:256 * (i + 1)], **p) for i, p in enumerate(ps)]
    evict = torch.empty(96 << 20, dtype=torch.float32, device="cuda")      # 384 MB: every cache
    stack = lambda: ops.gs_stack_fwd(x, layers, key_mask, n_seq, S, scale, 1e-5, dp, seed, train=True)
    k = int(os.environ.get("DSVG_GS_DBG_LAYER", str(n - 1)))
    one = lambda: ops.gs_layer_fwd(x, pf[k * E:(k + 1) * E], *_params(ps[k]), key_mask, n_seq, S, scale, 1e-5, dp, 8 * k, seed,
                                   seq_add=gcat[:, 256 * k:256 * (k + 1)], train=True)
    for label, fn in ((f"gs_stack_fwd, layer {k} of {n}", stack), (f"gs_layer_fwd (layer {k} alone)", one)):
        for _ in range(2):
            fn()
        buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device="cuda")
        evict.zero_()
        lib.check(L_.dsvg_gs_debug_clock(buf.data_ptr()), "dbg")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        lib.check(L_.dsvg_gs_debug_clock(None), "dbg")
        report(label + " (weights and rows cold)", buf, nwg, e0.elapsed_time(e1) * 1e3)


if __name__ == "__main__":
    main()
