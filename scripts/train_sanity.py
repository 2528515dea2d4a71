#!/usr/bin/env python
"""Sanity of the whole train step: repeated steps on a few fixed synthetic batches must drive the loss down, in bf16
(graph replay and eager) and fp32, with all exact work-skipping layouts on."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepsvg_amd  # noqa: E402
from deepsvg_amd.synthetic import make_batch, det_state_dict  # noqa: E402
from deepsvg_amd.trainer import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
cfg = deepsvg_amd.HierarchicalOrdered()
cfg.dropout = 0.1
batches = [tuple(t.to(dev) for t in make_batch(128, seed=s)) for s in (1, 2, 3, 4)]
for dtype, graph in ((torch.bfloat16, True), (torch.bfloat16, False), (torch.float32, False)):
    torch.manual_seed(0)
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(det_state_dict(model, seed=5))
    model.to(dev).set_compute_dtype(dtype).train()
    ts = TrainStep(model, deepsvg_amd.SVGLoss(cfg).to(dev), lr=3e-4, use_graph=graph)
    every = []
    for it in range(80):
        ld = ts.step(*batches[it % 4])
        every.append(float(ld["loss"]))
    first, last = sum(every[:4]) / 4, sum(every[-8:]) / 8
    hist = [round(v, 3) for v in every[::10]]
    print(f"dtype={dtype} graph={graph} loss every 10 steps: {hist}  mean first 4 = {first:.3f}, mean last 8 = {last:.3f}"
          f"  graphs captured: {len(ts._graphs)}")
    assert last < 0.85 * first, "loss did not go down"
print("train sanity ok")
