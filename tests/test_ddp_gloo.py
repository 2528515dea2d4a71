"""The N>1 path on CPU: two gloo ranks, each with half of a global batch, must produce exactly the update of one
process that sees the whole batch (global-count loss normalisation + one flat gradient all-reduce + identical
clip/AdamW on every rank).  Ops are the plain-torch restatements (host-logic test)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(kind="hier"):
    import deepsvg_amd
    cfg = H.build_cfg(kind)
    cfg.use_vae = False       # (a sampled latent would differ between the two-rank and the one-process run)
    cfg.n_layers = cfg.n_layers_decode = 1
    torch.manual_seed(0)
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(H.weights_for(model, 5))
    model.eval()      # dropout off: ranks would otherwise draw masks for different element ids than the full batch
    return cfg, model, deepsvg_amd.SVGLoss(cfg)


def _labels(cfg, n):
    if not cfg.label_condition:
        return None
    return torch.randint(0, cfg.n_labels, (n,), generator=torch.Generator().manual_seed(3))


def _worker(rank, world, port, ret, kind):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.conftest import install_emulated_ops
        install_emulated_ops()
        from deepsvg_amd.trainer import TrainStep
        from deepsvg_amd.synthetic import make_batch
        cfg, model, loss_fn = _make(kind)
        commands, args = make_batch(8, seed=21)
        label = _labels(cfg, 8)
        per = commands.shape[0] // world
        c, a = commands[rank * per:(rank + 1) * per], args[rank * per:(rank + 1) * per]
        ts = TrainStep(model, loss_fn, lr=1e-2)
        assert ts.overlap_allreduce
        ld = ts.step(c, a, label=label[rank * per:(rank + 1) * per] if label is not None else None)
        # the decoder bucket went out from inside the backward pass (hook on the bottleneck output's gradient)
        assert ts._pending is not None and 0 < ts._pending[0] < model.store.flat.numel()
        ret[rank] = (model.store.flat.clone(), ts.grad_norm(), {k: v.item() for k, v in ld.items()},
                     model.store.grad_buffer(0).clone() / world)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["hier", "fonts", "selfmatch"])
def test_two_rank_gloo_step_equals_single_process_step(kind):
    """hier: the north-star config; fonts: label conditioning (label tables + linear_global2); selfmatch: Hungarian
    assignment per icon before the loss"""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret, kind), nprocs=world, join=True)

    from tests.conftest import install_emulated_ops, restore_ops
    saved = install_emulated_ops()
    try:
        from deepsvg_amd.trainer import TrainStep
        from deepsvg_amd.synthetic import make_batch
        cfg, model, loss_fn = _make(kind)
        commands, args = make_batch(8, seed=21)
        ts = TrainStep(model, loss_fn, lr=1e-2)
        ld = ts.step(commands, args, label=_labels(cfg, 8))
        flat_ref, gn_ref = model.store.flat.clone(), ts.grad_norm()
        grad_ref = model.store.grad_buffer(0).clone()
    finally:
        restore_ops(saved)

    f0, gn0, ld0, g0 = ret[0]
    f1, gn1, ld1, g1 = ret[1]
    assert torch.equal(f0, f1), "ranks diverged after the step"
    assert abs(gn0 - gn_ref) <= 1e-4 * gn_ref, (gn0, gn_ref)
    assert torch.equal(g0, g1)
    assert torch.allclose(g0, grad_ref, rtol=1e-4, atol=1e-7), (g0 - grad_ref).abs().max().item()
    # Adam's first update is lr * g/|g|: elements whose gradient is fp32 noise may flip sign, hence the lr-scaled atol
    assert torch.allclose(f0, flat_ref, rtol=1e-4, atol=0.05 * 1e-2), (f0 - flat_ref).abs().max().item()
    # rank-average of the locally normalised losses == global loss
    for k in ("loss", "loss_cmd", "loss_args", "loss_visibility"):
        avg = 0.5 * (ld0[k] + ld1[k])
        assert abs(avg - ld[k].item()) <= 1e-5 * max(1.0, abs(ld[k].item())), (k, avg, ld[k].item())
