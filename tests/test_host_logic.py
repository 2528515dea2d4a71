"""Host logic of deepsvg_amd (autograd wiring, layouts, flat parameter store, loss masks) on CPU, with
deepsvg_amd.ops replaced by the plain-torch restatements of tests/torch_ops_ref.py.  These tests say nothing
about the HIP kernels (the -m gpu tests do); they prove that IF every op computes what its restatement computes,
the model reproduces the reference's logits / loss / gradients."""
import pytest
import torch

import deepsvg_amd
from deepsvg_amd import config as C
from oracle import svg_transformer_oracle as O
from tests import helpers as H


def _run_model(cfg, sd, commands, args, eps=None, dtype=torch.float32, label=None, args_dec=None):
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(sd)
    model.set_compute_dtype(dtype)
    loss_fn = deepsvg_amd.SVGLoss(cfg)
    model.eval()
    if eps is not None:
        torch.randn_like_orig = torch.randn_like
    model.zero_grad()
    if eps is not None:
        import deepsvg_amd.model as M
        orig = torch.randn_like
        M.torch.randn_like = lambda t: eps.reshape(t.shape).to(t.dtype)
    try:
        out = model(commands, args, commands, args if args_dec is None else args_dec, label=label, params={})
        ld = loss_fn(out, None, weights=O.DEFAULT_WEIGHTS)
        ld["loss"].backward()
    finally:
        if eps is not None:
            M.torch.randn_like = orig
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    return model, {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}, ld, grads


@pytest.mark.parametrize("name", H.golden_cases())
def test_model_matches_golden_with_emulated_ops(name, emulated_ops):
    g, cfg, commands, args, eps = H.golden_setup(name)
    ref_model = deepsvg_amd.SVGTransformer(cfg)
    sd = H.weights_for(ref_model, g["wseed"])
    label = H.golden_label(g)
    model, out, ld, grads = _run_model(cfg, sd, commands, args, eps, label=label, args_dec=H.golden_args_dec(g, args))
    H.check_against_golden(g, out, {k: v.item() for k, v in ld.items()}, grads, logit_rtol=1e-4, logit_atol=1e-5,
                           loss_tol=1e-5, grad_norm_rtol=2e-4)
    if "assignment" in g:       # Hungarian self-matching: the assignment the reference's perfect_matching returned
        assert torch.equal(model.last_assignment.long(), torch.from_numpy(g["assignment"]))
    if "sample_commands" in g:  # autoregressive sampling, the whole batch at once vs the reference's icon-by-icon loop
        cy, ay = model.greedy_sample(commands, args, None, None, concat_groups=False)
        H.check_sampled_sequences(cy, ay, g)
    z = model(commands, args, commands, args, label=label, encode_mode=True) if eps is None else None
    if z is not None:
        assert z.shape == tuple(g["z"].shape)
        assert torch.allclose(z, torch.from_numpy(g["z"]), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["hier_ordered_n5", "fonts_label_n4"])
def test_hierarch_path_with_emulated_ops(name, emulated_ops):
    """return_hierarch hands out the first decoder stage's (visibility logits, per-group latents), seq-first; fed
    back as hierarch_logits + z they reproduce the full forward (deepsvg/model/model.py:246-261,379-383)"""
    g, cfg, commands, args, eps = H.golden_setup(name)
    label = H.golden_label(g)
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(H.weights_for(model, g["wseed"]))
    model.eval()
    import deepsvg_amd.model as M
    orig = torch.randn_like
    if eps is not None:
        M.torch.randn_like = lambda t: eps.reshape(t.shape).to(t.dtype)
    try:
        with torch.no_grad():
            hl, zg = model(commands, args, commands, args, label=label, return_hierarch=True)
            full = model(commands, args, commands, args, label=label)
    finally:
        M.torch.randn_like = orig
    assert hl.shape == tuple(g["hier_logits"].shape) and zg.shape == tuple(g["hier_z"].shape)
    assert torch.allclose(hl, torch.from_numpy(g["hier_logits"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(zg, torch.from_numpy(g["hier_z"]), rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        again = model(None, None, None, None, label=label, z=zg.permute(2, 1, 0, 3).contiguous(), hierarch_logits=hl,
                      return_tgt=False)
    for k in ("command_logits", "args_logits", "visibility_logits"):
        assert torch.allclose(again[k], full[k], rtol=1e-5, atol=1e-6), k


def test_packed_encoder_equals_padded_encoder(emulated_ops):
    """the first encoder stage on the valid tokens only (default) vs on the reference's padded layout: same logits,
    losses and gradients (padded rows reach neither)"""
    g, cfg, commands, args, eps = H.golden_setup("hier_ordered_n5")
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), g["wseed"])
    res = {}
    for packed in (True, False):
        model = deepsvg_amd.SVGTransformer(cfg)
        model.load_state_dict(sd)
        model.pack_encoder = packed
        model.skip_invisible_backward = packed      # second exact skip: decoder stage-2 backward of invisible groups
        model.compact_head_backward = packed        # third: argument-head backward on the loss-carrying tokens only
        model.eval()
        out = model(commands, args, commands, args, params={})
        ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
        ld["loss"].backward()
        res[packed] = (out, ld, {n: p.grad.clone() for n, p in model.named_parameters()}, model.last_packing)
        assert (model.last_live is not None) == packed and (model.last_head_rows is not None) == packed
        if packed:
            assert 0 < model.last_live[0] < model.last_live[1]
            assert 0 < model.last_head_rows[0] < model.last_head_rows[1]
    total, dense = res[True][3]
    assert res[False][3] is None and 0 < total < dense
    for k in ("command_logits", "args_logits", "visibility_logits"):
        assert torch.allclose(res[True][0][k], res[False][0][k], rtol=1e-4, atol=2e-5), k
    assert abs(res[True][1]["loss"].item() - res[False][1]["loss"].item()) < 1e-5
    for n in res[True][2]:
        assert H.rel_l2(res[True][2][n], res[False][2][n]) < 1e-4, n


def test_args_logits_is_lazy_with_the_fused_head_loss(emulated_ops):
    """training forward + deepsvg_amd.SVGLoss: the dense args_logits is never built (the fused argument head + loss
    works on the loss-carrying tokens); reading it later still gives the reference's tensor"""
    g, cfg, commands, args, eps = H.golden_setup("hier_ordered_n2")
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), g["wseed"])
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(sd)
    model.eval()
    out = model(commands, args, commands, args, params={})
    assert out.is_pending("args_logits") and "args_logits" in out and set(out) >= {"command_logits", "args_logits"}
    ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
    ld["loss"].backward()
    assert out.is_pending("args_logits"), "the loss materialised the dense argument logits"
    al = out["args_logits"]                      # first read computes it
    assert not out.is_pending("args_logits") and tuple(al.shape) == (2, 8, 31, 11, 257)
    stride = int(g["args_logits_stride"])
    assert torch.allclose(al.detach().float().reshape(-1)[::stride], torch.from_numpy(g["args_logits_sample"]),
                          rtol=1e-4, atol=1e-5)
    assert abs(ld["loss_args"].item() - float(g["loss_args"])) < 1e-5
    # dict(out) / out.items() see real tensors, never the placeholder
    assert all(torch.is_tensor(v) or v is None or isinstance(v, dict) for v in dict(out).values())
    model.compact_head_backward = False
    out2 = model(commands, args, commands, args, params={})
    assert not out2.is_pending("args_logits")


def test_state_dict_layout_matches_reference_names(emulated_ops):
    g = H.load_golden("hier_ordered_n2")
    cfg = H.build_cfg("hier")
    model = deepsvg_amd.SVGTransformer(cfg)
    names = [n for n, _ in model.named_parameters()]
    assert names == [str(n) for n in g["grad_names"]], "parameter names/order differ from the reference"
    assert sum(p.numel() for p in model.parameters()) == 10304596
    bufs = sorted(n for n, _ in model.named_buffers())
    assert bufs == sorted(["cmd_args_mask", "encoder.embedding.pos_encoding.position", "encoder.hierarchical_PE.position",
                           "decoder.hierarchical_embedding.PE.position", "decoder.embedding.PE.position"])


@pytest.mark.parametrize("kind", ["hier", "fonts"])
def test_param_store_flat_views_and_grad_aliasing(emulated_ops, kind):
    cfg = H.build_cfg(kind)
    cfg.n_layers = cfg.n_layers_decode = 1
    cfg.use_vae = False                                     # (a sampled latent would differ between the steps below)
    model = deepsvg_amd.SVGTransformer(cfg)
    from deepsvg_amd.synthetic import make_batch
    commands, args = make_batch(2, seed=3)
    label = torch.tensor([7, 7]) if cfg.label_condition else None       # a repeated label: scatter-ADD of its rows
    loss_fn = deepsvg_amd.SVGLoss(cfg)
    model.eval()

    def step():
        out = model(commands, args, commands, args, label=label)
        loss_fn(out, None, weights=O.DEFAULT_WEIGHTS)["loss"].backward()

    step()
    store = model.store
    base = store.flat.data_ptr()
    for p in model.parameters():
        off = store.index[id(p)][0]
        assert p.data_ptr() == base + 4 * off and off % 8 == 0
        # autograd adopted the view of the flat gradient buffer (a deep copy would cost one launch per parameter)
        assert p.grad.data_ptr() == store.grad_buffer(0).data_ptr() + 4 * off, "p.grad is not a view of the flat buffer"
    g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    # second backward WITHOUT zeroing: autograd must accumulate (2x), not alias-and-double (4x) or overwrite (1x)
    step()
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[n], rtol=1e-5, atol=1e-7), n
    # zero_grad(set_to_none=False) then backward -> exactly 1x again
    model.zero_grad(set_to_none=False)
    step()
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, g1[n], rtol=1e-5, atol=1e-7), n
    # load_state_dict keeps the flat views; .to() style re-materialisation is detected and re-flattened
    model.load_state_dict(model.state_dict())
    for p in model.parameters():
        p.data = p.data.clone()
    model.zero_grad()
    step()
    assert model.store.flat.data_ptr() != base
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, g1[n], rtol=1e-5, atol=1e-7), n


def test_dropout_training_mode_runs_and_is_reproducible_in_backward(emulated_ops):
    """with p > 0 the backward must replay exactly the forward masks: check d(loss)/d(param) by finite differences
    on one scalar parameter while the seed is frozen."""
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    torch.manual_seed(0)
    model = deepsvg_amd.SVGTransformer(cfg)
    from deepsvg_amd.synthetic import make_batch
    commands, args = make_batch(2, seed=5)
    loss_fn = deepsvg_amd.SVGLoss(cfg)
    model.train()
    model._own_seed = False      # freeze the dropout seed so the function is deterministic
    model.seed_tensor(commands.device)

    def loss():
        return loss_fn(model(commands, args, commands, args), None, weights=O.DEFAULT_WEIGHTS)["loss"]

    l0 = loss()
    l0.backward()
    assert abs(loss().item() - l0.item()) < 1e-6          # same seed -> same masks
    p = model.decoder.decoder.layers[0].linear2.bias
    g = p.grad[3].item()
    h = 1e-2
    with torch.no_grad():
        p[3] += h
    lp = loss().item()
    with torch.no_grad():
        p[3] -= 2 * h
    lm = loss().item()
    fd = (lp - lm) / (2 * h)
    assert abs(fd - g) < 5e-3 * max(1.0, abs(g)), (fd, g)
    # and eval mode differs from train mode (dropout actually on)
    model.eval()
    assert abs(loss().item() - l0.item()) > 1e-4


def test_no_cpu_fallback_without_emulation():
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    model = deepsvg_amd.SVGTransformer(cfg)
    from deepsvg_amd.synthetic import make_batch
    commands, args = make_batch(1, seed=1)
    with pytest.raises(Exception, match="HIP device"):
        model(commands, args, commands, args)


def test_greedy_sample_shapes(emulated_ops):
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    model = deepsvg_amd.SVGTransformer(cfg).eval()
    z = torch.randn(3, 1, 1, cfg.dim_z) * 0.3
    commands_y, args_y = model.greedy_sample(z=z, concat_groups=False)
    assert commands_y.shape == (3, 8, 31) and args_y.shape == (3, 8, 31, 11)
    assert args_y.min().item() >= -1 and args_y.max().item() <= 255


def test_autoregressive_cached_sampling_equals_recompute(emulated_ops):
    """incremental decoding over the per-layer q|k|v cache vs the reference's scheme (decoder re-run on the whole prefix
    for every new token): same token sequences, on weights that make the sequences vary"""
    from deepsvg_amd.synthetic import make_batch_onestage
    cfg = H.build_cfg("sketchformer")
    cfg.max_total_len = 24
    model = deepsvg_amd.SVGTransformer(cfg)
    sd = H.weights_for(model, 11)
    sd["decoder.fcn.command_fcn.weight"] = sd["decoder.fcn.command_fcn.weight"] * 8      # spread the command logits
    model.load_state_dict(sd)
    model.eval()
    commands, args = make_batch_onestage(5, total_len=cfg.max_total_len, seed=3)
    outs = {}
    for kv in (True, False):
        model.kv_cache = kv
        torch.manual_seed(0)            # same categorical draws for both schemes
        outs[kv] = model.greedy_sample(commands, args, None, None, concat_groups=False)
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    assert outs[True][0].unique().numel() > 1, "degenerate sample: the test would not see an ordering bug"
