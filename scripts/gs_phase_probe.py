"""Where does a workgroup of gs_layer_fwd (one launch per group-stage layer, 128 workgroups for 4096 rows) spend its ~25-40 us?
s_memtime stamps per wave at the phase boundaries (development hook dsvg_gs_debug_clock), median over the launch's waves."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops, lib  # noqa: E402
from tests.test_group_stage_gpu import _setup, _params, _seed_tensor  # noqa: E402

NAMES = ["loads + LayerNorm 1", "in_proj + attention", "out_proj + LayerNorm 2", "linear1", "linear2", "x2 store"]


def main():
    n_seq, S = 512, 8
    flat, offs, p, x, key_mask, seq_add, dx2 = _setup(n_seq, S, seed=1, n_layers=1, masked=True, with_add=True)
    pf, pb = ops.gs_pack(flat, offs, 1)
    seed = _seed_tensor(77)
    scale, s0, dp = 32 ** -0.5, 208, 0.1
    L_ = lib.load()
    nwg = n_seq * S // 32
    tick = float(os.environ.get("PROBE_TICK", "1580"))         # shader cycles per microsecond under load (ffn_phase_probe)
    for train in (False, True):
        fwd = lambda: ops.gs_layer_fwd(x, pf, *_params(p), key_mask, n_seq, S, scale, 1e-5, dp, s0, seed, seq_add=seq_add, train=train)
        for _ in range(3):
            fwd()
        buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device="cuda")
        lib.check(L_.dsvg_gs_debug_clock(buf.data_ptr()), "dbg")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fwd()
        e1.record()
        torch.cuda.synchronize()
        lib.check(L_.dsvg_gs_debug_clock(None), "dbg")
        t = buf.view(nwg * 8, 8)[:, :7].double().cpu()
        d = (t[:, 1:] - t[:, :-1])
        med, p90 = d.median(0).values, d.quantile(0.9, 0)
        tot = (t[:, 6] - t[:, 0]).median().item()
        print(f"gs_layer_fwd {'train' if train else 'infer'}, {nwg} workgroups: launch {e0.elapsed_time(e1) * 1e3:5.1f} us; per wave "
              f"{tot / tick:5.1f} us = {tot:6.0f} cycles")
        for i, nme in enumerate(NAMES):
            print(f"    {nme:24s} {med[i].item():7.0f} cycles ({med[i].item() / tick:5.1f} us; 90th pct {p90[i].item() / tick:5.1f})")


if __name__ == "__main__":
    main()
