import os, sys, torch
sys.path.insert(0, "/root/repo")
from deepsvg_amd import ops
DEV = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
n_seq, S = 4096, 31
qkv = (torch.randn(n_seq * S, 768, generator=g)).to(DEV).to(torch.bfloat16)
do = (torch.randn(n_seq * S, 256, generator=g)).to(DEV).to(torch.bfloat16)
seed = torch.tensor([12345], dtype=torch.int64, device=DEV)
for p in (0.0, 0.1):
    f = timeit(lambda: ops.attention_fwd(qkv, None, n_seq, S, 8, 32 ** -0.5, p, 7, seed))
    b = timeit(lambda: ops.attention_bwd(qkv, None, do, n_seq, S, 8, 32 ** -0.5, p, 7, seed))
    print(f"dense31 4096 seq: p={p}: attention fwd {f:.1f} us, bwd {b:.1f} us")
n_seq, S = 512, 8
qkv = (torch.randn(n_seq * S, 768, generator=g)).to(DEV).to(torch.bfloat16)
do = (torch.randn(n_seq * S, 256, generator=g)).to(DEV).to(torch.bfloat16)
km = torch.full((n_seq,), 255, dtype=torch.int64, device=DEV)
for p in (0.0, 0.1):
    f = timeit(lambda: ops.attention_fwd(qkv, km, n_seq, S, 8, 32 ** -0.5, p, 7, seed))
    b = timeit(lambda: ops.attention_bwd(qkv, km, do, n_seq, S, 8, 32 ** -0.5, p, 7, seed))
    print(f"group stage 512 seq x 8: p={p}: attention fwd {f:.1f} us, bwd {b:.1f} us")
