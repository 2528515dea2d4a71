"""Micro-timing of the device-side batch assembly (dsvg_assemble_batch) at BASELINE's batch: 512 icons, G=8, S=30.
Prints icons/s and achieved HBM GB/s against the algorithmic bytes (kernel table)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import dataset as D       # noqa: E402


def main():
    rng = np.random.default_rng(0)
    G, S, n_icons, N = 8, 30, 4096, 512
    icons = []
    for _ in range(n_icons):
        ng = int(rng.integers(1, G + 1))
        groups = []
        for _g in range(ng):
            ln = int(rng.integers(2, S + 1))
            t = np.full((ln, 14), -1.0, np.float32)
            t[:, 0] = rng.integers(1, 3, size=ln)
            t[0, 0] = 0
            t[:, 8:14] = rng.integers(0, 256, size=(ln, 6))
            groups.append(t)
        icons.append([groups])
    store = D.PackedSVGStore.from_icons(icons, max_num_groups=G)
    ds = D.SVGTensorDataset(store=store, model_args=["commands", "args"], max_num_groups=G, max_seq_len=S)
    idx = torch.randint(0, n_icons, (N,), device="cuda")
    for keys in (["commands", "args"], ["commands", "args", "args_rel"]):
        for _ in range(5):
            ds.batch(idx, keys, random_aug=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for _ in range(reps):
            ds.batch(idx, keys, random_aug=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tokens = N * G * (S + 2)
        out_b = tokens * 4 * (1 + 11 * (len(keys) - 1))
        in_b = int(store.slot_off[-1]) / n_icons * N * 24
        print(f"{'+'.join(keys)}: {ms * 1e3:.1f} us per 512-icon batch (host + kernel), {N / ms * 1e3:,.0f} icons/s, "
              f"{(out_b + in_b) / ms / 1e6:.1f} GB/s of {out_b + in_b:,.0f} algorithmic bytes")


if __name__ == "__main__":
    main()
