"""Launch sequence for the HBM-traffic measurement of ffn_fwd (scripts/gpu_ffn_traffic.sh wraps it in rocprofv3 --pmc passes):
for rows in (63488, 40960): 3 x inference variant, 3 x training variant, p = 0.1, after 2 warm-up launches of each.  The
inference variant's read bytes are known (the rows once: rows x 512 B, + 512 KiB of packed weights per launch from L2), which
calibrates what FETCH_SIZE reports for THIS kernel's access pattern (16-byte pieces at a 512-byte lane stride)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvg_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
flat = torch.zeros(8 + 131072 + 512 + 131072 + 256 + 256 + 8)
o = 8
offs = [[o, o + 131072, o + 131072 + 512, o + 262144 + 512, o + 262144 + 768]]
flat[o:o + 131072] = torch.randn(131072, generator=g) * 0.06
flat[o + 131072 + 512:o + 262144 + 512] = torch.randn(131072, generator=g) * 0.06
flat[o + 262144 + 512:o + 262144 + 768] = 1.0
flat = flat.to(DEV)
pf, pb, b1f = ops.ffn_pack(flat, torch.tensor(offs, dtype=torch.int64, device=DEV), 1)
pl = pf[:ops.FFN_FWD_LAYER_ELEMS]
b2 = torch.zeros(256, device=DEV)
seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
for rows in (63488, 40960):
    x = torch.randn(rows, 256, generator=g).to(DEV).to(torch.bfloat16)
    # push x out of the caches between launches with a 512 MB fill (MALL is 256 MB): every launch reads its rows from HBM
    junk = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=DEV)
    for train in (False, True):
        for _ in range(5):
            junk.fill_(1.0)
            ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=train)
    torch.cuda.synchronize()
print("done")
