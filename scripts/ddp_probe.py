"""Where does the data-parallel hipGraph step spend its extra time?  One-rank RCCL group, force_ddp: wall time of the phases
of TrainStep.step with a device synchronisation between them (so the numbers do not add up to the pipelined step time)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepsvg_amd  # noqa: E402
from deepsvg_amd.synthetic import make_batch, det_state_dict  # noqa: E402
from deepsvg_amd.trainer import TrainStep  # noqa: E402

dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29733", rank=0, world_size=1)
cfg = deepsvg_amd.HierarchicalOrdered()
c, a = make_batch(512, seed=1)
c, a = c.cuda(), a.cuda()


def build(force):
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(det_state_dict(model, seed=1))
    model.to("cuda").set_compute_dtype(torch.bfloat16).train()
    ts = TrainStep(model, deepsvg_amd.SVGLoss(cfg).cuda(), lr=1e-3, use_graph=True, force_ddp=force)
    ts.inputs_resident = True
    for _ in range(5):
        ts.step(c, a)
    torch.cuda.synchronize()
    return ts


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for force in (False, True):
    ts = build(force)
    print(f"force_ddp={force}: pipelined step {timed(lambda: ts.step(c, a)):.3f} ms")
    if force:
        entry = next(iter(ts._graphs.values()))
        plan = ts.model.make_plan(c, a, c, True, a)
        print(f"  _global_counts alone      {timed(lambda: ts._global_counts(c, a, plan)):.3f} ms")
        print(f"  graph replay alone        {timed(lambda: entry[0].replay()):.3f} ms")
        print(f"  _step_back alone          {timed(lambda: ts._step_back()):.3f} ms")
        flat_g = ts.model.store.grad_buffer(0)
        print(f"  all_reduce(flat_g) alone  {timed(lambda: dist.all_reduce(flat_g)):.3f} ms")
        print(f"  make_plan alone           {timed(lambda: ts.model.make_plan(c, a, c, True, a)):.3f} ms")
dist.destroy_process_group()
