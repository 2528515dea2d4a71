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


def _worker(rank, world, port, ret, kind, bf16_reduce=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if bf16_reduce:
        os.environ["DSVG_DDP_BF16"] = "1"
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
        assert ts.overlap_allreduce and ts.allreduce_bf16 == bf16_reduce
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


def test_two_rank_gloo_step_with_the_gradient_reduced_in_bf16():
    """DSVG_DDP_BF16=1 (20.6 MB instead of 41.2 MB on the wire): both buckets - the decoder's, sent from inside
    the backward pass, and the rest - are cast to bf16, summed, cast back.  The ranks still end bit-identical to each other; the
    averaged gradient equals the single-process one to bf16 rounding (2^-8 relative per element, far less in the norm)"""
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret, "hier", True), nprocs=world, join=True)
    from tests.conftest import install_emulated_ops, restore_ops
    saved = install_emulated_ops()
    try:
        from deepsvg_amd.trainer import TrainStep
        from deepsvg_amd.synthetic import make_batch
        cfg, model, loss_fn = _make("hier")
        commands, args = make_batch(8, seed=21)
        ts = TrainStep(model, loss_fn, lr=1e-2)
        assert not ts.allreduce_bf16
        ts.step(commands, args)
        gn_ref, grad_ref = ts.grad_norm(), model.store.grad_buffer(0).clone()
    finally:
        restore_ops(saved)
    f0, gn0, _, g0 = ret[0]
    f1, gn1, _, g1 = ret[1]
    assert torch.equal(f0, f1) and torch.equal(g0, g1), "ranks diverged"
    assert abs(gn0 - gn_ref) <= 2e-3 * gn_ref, (gn0, gn_ref)
    rel = ((g0 - grad_ref).norm() / grad_ref.norm()).item()
    assert 1e-5 < rel < 4e-3, rel              # bf16 rounding is there (the switch took effect) and is all there is
    assert (g0 - grad_ref).abs().max().item() <= 2.0 ** -7 * grad_ref.abs().max().item()


# ---------------------------------------------------------------------------------------------------------------------
# four ranks, unequal per-rank icon counts, ranks in different graph buckets
# ---------------------------------------------------------------------------------------------------------------------
_SPLIT4 = (1, 4, 2, 3)          # icons per rank (10 in total)


def _worker4(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.conftest import install_emulated_ops
        install_emulated_ops()
        from deepsvg_amd.trainer import TrainStep
        from deepsvg_amd.synthetic import make_batch
        cfg, model, loss_fn = _make("hier")
        commands, args = make_batch(sum(_SPLIT4), seed=33)
        lo = sum(_SPLIT4[:rank])
        c, a = commands[lo:lo + _SPLIT4[rank]], args[lo:lo + _SPLIT4[rank]]
        ts = TrainStep(model, loss_fn, lr=1e-2)
        # the graph bucket this rank's batch would replay on a GPU (packed encoder rows, visible sequences, loss rows, slot
        # range): ranks differ, yet every rank issues the same collectives - one count all-reduce, the gradient all-reduce
        ts.row_bucket, ts.seq_bucket = 16, 2
        key, _ = ts._bucketed(model.make_plan(c, a, c, True, a), c)
        assert ts.rccl_ranks() == world
        ld = ts.step(c, a)
        ret[rank] = (model.store.flat.clone(), ts.grad_norm(), {k: v.item() for k, v in ld.items()},
                     model.store.grad_buffer(0).clone() / world, key)
    finally:
        dist.destroy_process_group()


def test_four_rank_gloo_step_with_unequal_batches_equals_single_process_step():
    """Ranks hold 1 / 4 / 2 / 3 icons: the loss normalisers are the GLOBAL selected-element counts / world, so the
    rank-averaged gradient is still the gradient of the global-batch mean (deepsvg/train.py:74,100 semantics), whatever
    the per-rank sizes and whatever layout bucket each rank lands in."""
    world = 4
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker4, args=(world, port, ret), nprocs=world, join=True)

    from tests.conftest import install_emulated_ops, restore_ops
    saved = install_emulated_ops()
    try:
        from deepsvg_amd.trainer import TrainStep
        from deepsvg_amd.synthetic import make_batch
        cfg, model, loss_fn = _make("hier")
        commands, args = make_batch(sum(_SPLIT4), seed=33)
        ts = TrainStep(model, loss_fn, lr=1e-2)
        assert ts.rccl_ranks() == 1
        ld = ts.step(commands, args)
        flat_ref, gn_ref = model.store.flat.clone(), ts.grad_norm()
        grad_ref = model.store.grad_buffer(0).clone()
    finally:
        restore_ops(saved)

    flats = [ret[r][0] for r in range(world)]
    for r in range(1, world):
        assert torch.equal(flats[0], flats[r]), f"rank {r} diverged from rank 0 after the step"
        assert torch.equal(ret[0][3], ret[r][3])
    g0 = ret[0][3]
    assert torch.allclose(g0, grad_ref, rtol=1e-4, atol=1e-7), (g0 - grad_ref).abs().max().item()
    assert abs(ret[0][1] - gn_ref) <= 1e-4 * gn_ref
    assert torch.allclose(flats[0], flat_ref, rtol=1e-4, atol=0.05 * 1e-2)
    keys = {ret[r][4] for r in range(world)}
    assert len(keys) >= 2, f"the four ranks were expected to land in different layout buckets, got {keys}"
    for k in ("loss", "loss_cmd", "loss_args", "loss_visibility"):
        avg = sum(ret[r][2][k] for r in range(world)) / world
        assert abs(avg - ld[k].item()) <= 1e-5 * max(1.0, abs(ld[k].item())), (k, avg, ld[k].item())


def test_graph_cache_is_bounded_lru():
    """TrainStep keeps at most `max_graphs` captured graphs, least recently used out first (host logic of the cache: the
    capture itself is a callback here; tests/test_model_gpu.py runs the real thing)"""
    from deepsvg_amd.trainer import TrainStep
    cfg, model, loss_fn = _make("hier")
    ts = TrainStep(model, loss_fn)
    ts.max_graphs = 3
    made = []

    def cap(k):
        made.append(k)
        return ("graph", k)

    for k in (1, 2, 3):
        assert ts._graph_entry(k, lambda k=k: cap(k)) == (("graph", k), True)
    assert ts._graph_entry(1, lambda: cap(1)) == (("graph", 1), False)      # hit: 1 becomes the most recent
    assert ts._graph_entry(4, lambda: cap(4))[1] and list(ts._graphs) == [3, 1, 4]      # 2 (the oldest) is gone
    assert ts._graph_entry(2, lambda: cap(2))[1] and list(ts._graphs) == [1, 4, 2]      # re-captured, 3 evicted
    assert made == [1, 2, 3, 4, 2] and ts.graphs_captured == 5 and ts.graphs_evicted == 2
