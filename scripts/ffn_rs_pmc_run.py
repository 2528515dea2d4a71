"""Launch set for the PMC comparison of the role-specialised ffn_fwd (stages = 5) with the default kernel: 32,768 and 63,488
rows, inference and training variants, p = 0.1, 3 launches each after 2 warm-ups (scripts/gpu_ffn_rs_pmc.sh wraps it)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops                      # noqa: E402
from tests.test_kernels_gpu import _ffn_setup, _seed_tensor   # noqa: E402

flat, offs, _, b2 = _ffn_setup(8, seed=1)
pf, pb, b1f = ops.ffn_pack(flat, offs, 2)
pl = pf[:ops.FFN_FWD_LAYER_ELEMS]
seed = _seed_tensor(77)
g = torch.Generator(device="cpu").manual_seed(0)
for rows in (32768, 63488):
    x = torch.randn(rows, 256, generator=g).cuda().to(torch.bfloat16)
    for st in (6, 5):
        for train in (False, True):
            for _ in range(5):
                ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train, stages=st)
torch.cuda.synchronize()
print("done")
