import torch, deepsvg_amd, time
from deepsvg_amd import ops
import deepsvg_amd.functional as Fn
from deepsvg_amd.synthetic import make_batch
from deepsvg_amd.trainer import TrainStep
from tests import helpers as H
cfg = H.build_cfg("hier"); cfg.dropout = 0.1
model = deepsvg_amd.SVGTransformer(cfg).cuda(); model.set_compute_dtype(torch.bfloat16); model.train()
ts = TrainStep(model, deepsvg_amd.SVGLoss(cfg).cuda(), lr=1e-4, use_graph=False)
c, a = (t.cuda() for t in make_batch(512, seed=1))
orig = ops.gemm
def spy(A, B, **kw):
    if max(A.shape[1], B.shape[0], B.shape[1]) > 1100 and A.shape[0] > 1000:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig(A, B, **kw)
        torch.cuda.synchronize(); print("gemm", tuple(A.shape), A.stride(), tuple(B.shape), {k: v for k, v in kw.items() if k in ("a_kc", "b_kc", "split_k")}, f"{(time.perf_counter()-t0)*1e6:.0f} us")
        return r
    return orig(A, B, **kw)
ops.gemm = spy
for kp in (True, False):
    Fn.HEAD_KPAD = kp
    print("HEAD_KPAD", kp); ts.step(c, a); torch.cuda.synchronize()
st = model.store
for n, p in model.named_parameters():
    if "args_fcn" in n or "args" in n and "weight" in n:
        wl = st.lp(p)
        print(n, tuple(p.shape), "lp", None if wl is None else (wl.is_contiguous(), wl.storage_offset(), wl.untyped_storage().data_ptr() == st.flat_lp.untyped_storage().data_ptr(), st.flat_lp.numel()))
