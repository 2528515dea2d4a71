"""Run-to-run determinism of the bf16 training forward + backward: two TrainSteps from the same seed on the same batches must
produce bit-identical losses and flat gradients.  Knobs come from the environment (DSVG_FFN_PACKED, DSVG_FFN_FUSED, ...)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepsvg_amd                                    # noqa: E402
from deepsvg_amd.synthetic import make_batch, det_state_dict   # noqa: E402
from deepsvg_amd.trainer import TrainStep            # noqa: E402

DEV = "cuda"
cfg = deepsvg_amd.HierarchicalOrdered()
cfg.dropout = 0.1
m0 = deepsvg_amd.SVGTransformer(cfg)
sd = det_state_dict(m0, seed=77)
batches = [tuple(t.to(DEV) for t in make_batch(640, seed=s)) for s in (21, 22)]
runs = []
for rep in range(3):
    torch.manual_seed(99)
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(sd)
    model.to(DEV).set_compute_dtype(torch.bfloat16).train()
    ts = TrainStep(model, deepsvg_amd.SVGLoss(cfg).to(DEV), lr=0.0, use_graph=False)
    losses = []
    for c, a in batches:
        ld = ts.step(c, a)
        losses.append({k: float(v) for k, v in ld.items()})
    torch.cuda.synchronize()
    runs.append((losses, model.store.grad_buffer(0).detach().clone()))
for r in runs[1:]:
    same_l = r[0] == runs[0][0]
    d = (r[1] - runs[0][1]).abs().max().item()
    print("losses identical:", same_l, "| max |grad diff|:", d, "|", [x["loss"] for x in r[0]], [x["loss"] for x in runs[0][0]])
