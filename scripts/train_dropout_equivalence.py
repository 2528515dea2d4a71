#!/usr/bin/env python
"""Does training with the kernels' counter-hash dropout behave like training with the reference's torch-RNG dropout?
(Round-4 review item 4b; SURVEY.md 7.3-1: bit equality with torch's Philox masks is impossible, statistical equivalence is the
claim.)  300 steps of the full train step (forward + SVGLoss + backward + clip 1.0 + AdamW, lr 3e-4) at dropout 0.1 on the
same 16 rotating synthetic batches from the same initial weights:
  oracle A / B   oracle/svg_transformer_oracle.py (the reference restated with stock torch ops, fp32, torch's dropout) on this
                 GPU, torch seeds 0 and 1: their difference is the run-to-run noise of the reference's own dropout
  hip fp32 / bf16   deepsvg_amd TrainStep (hipGraph), seeds of its own
Loss curves are compared as means over windows of 20 steps: the product's deviation from the oracle mean must stay within
2 x the oracle's own A-B spread (floored at 1.5 % of the loss).  Prints the table; exit code 1 on failure."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deepsvg_amd  # noqa: E402
from deepsvg_amd.synthetic import make_batch, det_state_dict  # noqa: E402
from deepsvg_amd.trainer import TrainStep  # noqa: E402
from oracle import svg_transformer_oracle as O  # noqa: E402  (the checker; this script is test infrastructure)

STEPS = int(os.environ.get("EQ_STEPS", "300"))
BATCH = int(os.environ.get("EQ_BATCH", "64"))
WIN = 20
LR = 3e-4
dev = torch.device("cuda:0")
cfg = deepsvg_amd.HierarchicalOrdered()
cfg.dropout = 0.1
batches_cpu = [make_batch(BATCH, seed=100 + s) for s in range(16)]
batches = [(c.to(dev), a.to(dev)) for c, a in batches_cpu]
ref_model = deepsvg_amd.SVGTransformer(cfg)
sd = det_state_dict(ref_model, seed=5)


def oracle_run(seed):
    torch.manual_seed(seed)
    leaves = {k: v.detach().clone().to(dev).requires_grad_(torch.is_floating_point(v)) for k, v in sd.items()}
    params = [v for v in leaves.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=LR)
    losses = []
    O.TRAIN_DROPOUT = 0.1
    try:
        for it in range(STEPS):
            c, a = batches[it % len(batches)]
            opt.zero_grad()
            out = O.forward(leaves, cfg, c, a, c, a)
            ld = O.svg_loss(cfg, out, O.DEFAULT_WEIGHTS)
            ld["loss"].backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            losses.append(float(ld["loss"].detach()))
    finally:
        O.TRAIN_DROPOUT = 0.0
    return losses


def hip_run(dtype, seed):
    torch.manual_seed(seed)
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(sd)
    model.to(dev).set_compute_dtype(dtype).train()
    ts = TrainStep(model, deepsvg_amd.SVGLoss(cfg).to(dev), lr=LR, use_graph=True)
    losses = []
    for it in range(STEPS):
        ld = ts.step(*batches[it % len(batches)])
        losses.append(float(ld["loss"]))        # (read NOW: a hipGraph step returns its static result tensors)
    return losses


def windows(v):
    return [sum(v[i:i + WIN]) / WIN for i in range(0, len(v) - WIN + 1, WIN)]


runs = {"oracle A": oracle_run(0), "oracle B": oracle_run(1), "hip fp32": hip_run(torch.float32, 10),
        "hip bf16": hip_run(torch.bfloat16, 11), "hip bf16 (2nd seed)": hip_run(torch.bfloat16, 12)}
w = {k: windows(v) for k, v in runs.items()}
n = len(w["oracle A"])
print(f"{STEPS} steps, batch {BATCH}, 16 rotating batches, dropout 0.1, lr {LR}; mean loss per window of {WIN} steps")
print("window " + " ".join(f"{k:>20s}" for k in w))
for i in range(n):
    print(f"{i:6d} " + " ".join(f"{w[k][i]:20.4f}" for k in w))
ok = True
spread = [abs(a - b) for a, b in zip(w["oracle A"], w["oracle B"])]
mean_o = [(a + b) / 2 for a, b in zip(w["oracle A"], w["oracle B"])]
for k in ("hip fp32", "hip bf16", "hip bf16 (2nd seed)"):
    dev_k = [abs(x - m) for x, m in zip(w[k], mean_o)]
    lim = [2.0 * max(s, 0.015 * m) for s, m in zip(spread, mean_o)]
    worst = max(d / l for d, l in zip(dev_k, lim))
    print(f"{k}: largest deviation from the oracle mean {max(dev_k):.4f} (oracle A-B spread: max {max(spread):.4f}, mean "
          f"{sum(spread) / n:.4f}); worst deviation / limit = {worst:.2f}")
    ok = ok and worst <= 1.0
first, last = mean_o[0], mean_o[-1]
for k in w:
    assert w[k][-1] < 0.9 * w[k][0], f"{k}: the loss did not go down"
print(f"oracle mean loss {first:.3f} -> {last:.3f}")
print("dropout training equivalence: " + ("OK" if ok else "FAILED"))
sys.exit(0 if ok else 1)
