"""Where do the small device-to-device copies of a train step come from?  One eager step under torch.profiler, aten::copy_ /
clone / cat / fill calls grouped by the innermost deepsvg_amd frame."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepsvg_amd  # noqa: E402
from deepsvg_amd.synthetic import make_batch, det_state_dict  # noqa: E402
from deepsvg_amd.trainer import TrainStep  # noqa: E402

cfg = deepsvg_amd.HierarchicalOrdered()
model = deepsvg_amd.SVGTransformer(cfg)
model.load_state_dict(det_state_dict(model, seed=1))
model.to("cuda").set_compute_dtype(torch.bfloat16).train()
c, a = make_batch(512, seed=1)
c, a = c.cuda(), a.cuda()
ts = TrainStep(model, deepsvg_amd.SVGLoss(cfg).cuda(), lr=1e-3, use_graph=False)
for _ in range(3):
    ts.step(c, a)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    ts.step(c, a)
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::cat", "aten::fill_", "aten::zero_", "aten::add", "aten::add_",
                   "aten::mul", "aten::div", "aten::stack", "aten::sum", "aten::ne", "aten::_to_copy", "aten::contiguous",
                   "aten::index_select", "aten::zeros", "aten::full"):
        frame = next((f for f in ev.stack if "deepsvg_amd" in f), "?")
        by[(ev.name, frame.strip()[-90:])] += 1
for (name, frame), n in sorted(by.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{n:4d}  {name:18s} {frame}")
