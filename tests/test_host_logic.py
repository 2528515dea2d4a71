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


def test_invisible_groups_run_forward_only_when_their_logits_are_read(emulated_ops):
    """training call + deepsvg_amd.SVGLoss: the second decoder stage runs the visible groups' sequences only (forward and
    backward) and both dense logit tensors stay lazy; loss and gradients are those of the dense computation, and a later
    read of the logits gives the dense tensors"""
    from deepsvg_amd.synthetic import make_batch
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 2
    c, a = make_batch(12, seed=5)
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), 7)
    res = {}
    for skip in (True, False):
        model = deepsvg_amd.SVGTransformer(cfg).eval()
        model.load_state_dict(sd)
        model.skip_invisible_forward = skip
        out = model(c, a, c, a, params={})
        assert out.is_pending("command_logits") == skip and out.is_pending("args_logits")
        ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
        ld["loss"].backward()
        assert out.is_pending("command_logits") == skip, "the loss materialised the dense command logits"
        res[skip] = (ld["loss"].item(), {n: p.grad.clone() for n, p in model.named_parameters()},
                     out["command_logits"].detach().float(), out["args_logits"].detach().float())
        assert not out.is_pending("command_logits")
    assert model.last_live is not None and model.last_live[0] < model.last_live[1], "the batch has no invisible group"
    assert abs(res[True][0] - res[False][0]) < 1e-6
    for n in res[True][1]:
        assert H.rel_l2(res[True][1][n], res[False][1][n]) < 1e-5, n
    assert torch.allclose(res[True][2], res[False][2], rtol=1e-5, atol=1e-6)
    assert torch.allclose(res[True][3], res[False][3], rtol=1e-5, atol=1e-6)


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


@pytest.mark.parametrize("tag", H.sample_cases())
def test_greedy_sample_matches_reference_golden_with_emulated_ops(tag, emulated_ops):
    """the product's one-shot greedy_sample host glue (draw, visibility threshold, _make_valid, concat_groups) against
    the reference's own samples, every leg of tests/golden/make_golden_sample.py"""
    import deepsvg_amd.model as M
    t, cfg = H.sample_fixture(tag)
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(H.weights_for(model, int(t["wseed"])))
    model.eval()
    eps = t.get("eps")

    def sample(c, a, label, z, hl, concat, icon):
        orig = torch.randn_like
        if eps is not None:
            e = eps if icon is None else eps[:, :, icon:icon + 1]
            M.torch.randn_like = lambda x: e.reshape(x.shape).to(x.dtype)
        try:
            torch.manual_seed(0)
            return model.greedy_sample(c, a, None, None, label=label, z=z, hierarch_logits=hl, concat_groups=concat)
        finally:
            M.torch.randn_like = orig
    H.run_sample_checks(sample, t, cfg)


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


# ---- round-1 advisor findings ---------------------------------------------------------------------------------------
def _loss_of(model, lf, c, a):
    return lf(model(c, a, c, a, params={}), None, weights=O.DEFAULT_WEIGHTS)["loss"]


def test_two_forwards_one_backward_sum_their_gradients(emulated_ops):
    """(L(model(x1)) + L(model(x2))).backward(): two backward nodes ask for the same parameter's gradient before
    AccumulateGrad runs - they must not be handed the same slot of the flat gradient buffer (ParamStore.grad_view)"""
    from deepsvg_amd.synthetic import make_batch
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    model = deepsvg_amd.SVGTransformer(cfg).eval()
    lf = deepsvg_amd.SVGLoss(cfg)
    b1, b2 = make_batch(3, seed=1), make_batch(3, seed=2)
    model.zero_grad()
    for c, a in (b1, b2):
        _loss_of(model, lf, c, a).backward()
    want = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    (_loss_of(model, lf, *b1) + _loss_of(model, lf, *b2)).backward()
    for n, p in model.named_parameters():
        assert H.rel_l2(p.grad, want[n]) < 1e-5, n


def test_other_losses_get_the_full_backward(emulated_ops):
    """the live-prefix backward of the second decoder stage is exact under SVGLoss only: a loss that touches the logits
    of invisible groups must see the same gradients with the skip on (default) and off"""
    from deepsvg_amd.synthetic import make_batch
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    c, a = make_batch(6, seed=3)
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), 4)
    grads = {}
    for skip in (True, False):
        model = deepsvg_amd.SVGTransformer(cfg).eval()
        model.load_state_dict(sd)
        model.skip_invisible_backward = skip
        out = model(c, a, c, a, params={})
        assert (model.last_live is not None) == skip
        (out["command_logits"].float().pow(2).sum() + out["visibility_logits"].float().sum()).backward()
        grads[skip] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert grads[True].keys() == grads[False].keys()
    for n in grads[True]:
        assert H.rel_l2(grads[True][n], grads[False][n]) < 1e-5, n
    # ... while SVGLoss arms the restriction (and stays exact: test_packed_encoder_equals_padded_encoder)
    model = deepsvg_amd.SVGTransformer(cfg).eval()
    out = model(c, a, c, a, params={})
    live = dict.get(out, "_dsvg_live")["live"]
    assert not live.armed
    deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
    assert live.armed


def test_model_survives_deepcopy_pickle_and_parameter_reassignment(emulated_ops):
    import copy
    import io
    from deepsvg_amd.synthetic import make_batch
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    model = deepsvg_amd.SVGTransformer(cfg).eval()
    lf = deepsvg_amd.SVGLoss(cfg)
    c, a = make_batch(3, seed=1)

    def grads(m):
        m.zero_grad()
        _loss_of(m, lf, c, a).backward()
        return {n: p.grad.clone() for n, p in m.named_parameters()}
    g0 = grads(model)                        # the flat store now exists
    clone = copy.deepcopy(model)             # EMA / best-model snapshot
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    for m in (clone, loaded):
        g = grads(m)
        assert all(torch.equal(g[n], g0[n]) for n in g0)
        assert m.store is not model.store and m.store.flat.data_ptr() != model.store.flat.data_ptr()
    # a re-assigned .data of a parameter in the MIDDLE of the flat buffer is noticed (the buffer is rebuilt)
    name = "encoder.encoder.layers.0.linear1.weight"
    p = dict(model.named_parameters())[name]
    p.data = p.data.clone() * 0.5
    g1 = grads(model)
    other = "encoder.encoder.layers.0.linear2.weight"
    assert not torch.equal(g1[other], g0[other])
    assert p.data_ptr() == model.store.flat.data_ptr() + 4 * model.store.index[id(p)][0]


def test_hierarch_outputs_feed_straight_back_for_a_batch(emulated_ops):
    """return_hierarch hands out (1, G, N, 2) and (1, G, N, dim_z), both seq-first; feeding them straight back (as the
    reference's notebooks do with N = 1) must give the full forward's logits for N > 1 too, and the documented
    batch-first per-group latents keep working"""
    from deepsvg_amd.synthetic import make_batch
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    model = deepsvg_amd.SVGTransformer(cfg).eval()
    c, a = make_batch(3, seed=5)
    with torch.no_grad():
        full = model(c, a, c, a, return_tgt=False)
        hl, zg = model(c, a, None, None, return_hierarch=True, return_tgt=False)
        assert hl.shape == (1, 8, 3, 2) and zg.shape == (1, 8, 3, cfg.dim_z)
        back = model(None, None, None, None, z=zg, hierarch_logits=hl, return_tgt=False)
        back2 = model(None, None, None, None, z=zg.permute(2, 1, 0, 3).contiguous(), hierarch_logits=hl, return_tgt=False)
    for k in ("command_logits", "args_logits"):
        assert torch.allclose(back[k], full[k], atol=1e-5) and torch.allclose(back2[k], full[k], atol=1e-5), k
    with pytest.raises(ValueError):
        model(None, None, None, None, z=zg[:, :, :2].permute(2, 1, 0, 3).contiguous(), hierarch_logits=hl, return_tgt=False)


def test_fused_ffn_path_matches_unfused_with_emulated_ops(emulated_ops):
    """bf16 compute: every layer's FFN runs through ffn_fwd / ffn_bwd / wgrad_finish (LayerNorm folded into the packed
    linear1, fragment-ordered weight gradients) - the wiring must reproduce the unfused layer's loss and gradients up
    to bf16 rounding, including the live-prefix backward of the second decoder stage"""
    from deepsvg_amd.synthetic import make_batch
    import deepsvg_amd.functional as Fn
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 2
    c, a = make_batch(6, seed=3)
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), 4)
    res = {}
    monkey_min_rows, Fn.FFN_MIN_ROWS = Fn.FFN_MIN_ROWS, 0       # (the production threshold is far above this batch)
    for fused in (True, False):
        model = deepsvg_amd.SVGTransformer(cfg).eval()
        model.load_state_dict(sd)
        model.set_compute_dtype(torch.bfloat16)
        if not fused:
            model.store._ffn_setup = lambda device: None
        out = model(c, a, c, a, params={})
        assert (model.store._ffn is not None) == fused
        ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
        ld["loss"].backward()
        res[fused] = (float(ld["loss"]), {n: p.grad.clone() for n, p in model.named_parameters()})
    Fn.FFN_MIN_ROWS = monkey_min_rows
    assert abs(res[True][0] - res[False][0]) < 2e-2 * abs(res[False][0])
    worst, name = max((H.rel_l2(res[True][1][n], res[False][1][n]), n) for n in res[True][1])
    assert worst < 0.12, (worst, name)
    med = sorted(H.rel_l2(res[True][1][n], res[False][1][n]) for n in res[True][1])[len(res[True][1]) // 2]
    assert med < 0.05, med


def test_latent_chain_wiring_matches_the_unfused_blocks_with_emulated_ops(emulated_ops, monkeypatch):
    """bf16 compute: the latent ResNet + bottleneck through functional.LatentChainFn (one launch per direction on the GPU) -
    with the emulated ops the fused call restates the unfused launches with their bf16 roundings, so loss and gradients must be
    IDENTICAL to the per-block path (DSVG_LATENT_FUSED=0): this pins the wiring (which z / r / dpre feeds which weight gradient)"""
    from deepsvg_amd.synthetic import make_batch
    import deepsvg_amd.functional as Fn
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    c, a = make_batch(6, seed=3)
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), 4)
    res, used = {}, {}
    for fused in (True, False):
        monkeypatch.setattr(Fn, "LATENT_FUSED", fused)
        calls = []
        real = Fn.LatentChainFn.apply
        monkeypatch.setattr(Fn.LatentChainFn, "apply", lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1])
        model = deepsvg_amd.SVGTransformer(cfg).eval()
        model.load_state_dict(sd)
        model.set_compute_dtype(torch.bfloat16)
        out = model(c, a, c, a, params={})
        ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
        ld["loss"].backward()
        monkeypatch.setattr(Fn.LatentChainFn, "apply", real)
        used[fused] = len(calls)
        res[fused] = (float(ld["loss"].detach()), {n: p.grad.clone() for n, p in model.named_parameters()})
    assert used == {True: 1, False: 0}
    assert res[True][0] == res[False][0]
    for n in res[True][1]:
        assert torch.equal(res[True][1][n], res[False][1][n]), n


def test_decoder_conditioning_gradients_as_one_product_with_emulated_ops(emulated_ops, monkeypatch):
    """functional.GlobalCondFn: the four linear_global weights of a decoder stack sit next to each other in the flat buffers
    (ParamStore._grouped_order), so their gradients come from ONE weight-gradient product of the concatenated dg and dz from ONE
    product over the concatenated reduction dimension (DSVG_GLOBAL_COND_CAT=0: one launch per layer).  Same numbers up to the
    bf16 rounding of dz (one rounding instead of one per accumulating launch); state_dict order and values are untouched."""
    from deepsvg_amd.synthetic import make_batch
    import deepsvg_amd.functional as Fn
    cfg = H.build_cfg("hier")
    c, a = make_batch(6, seed=3)
    ref = deepsvg_amd.SVGTransformer(cfg)
    sd = H.weights_for(ref, 4)
    res = {}
    for cat in (True, False):
        monkeypatch.setattr(Fn, "GLOBAL_COND_CAT", cat)
        model = deepsvg_amd.SVGTransformer(cfg).eval()
        model.load_state_dict(sd)
        model.set_compute_dtype(torch.bfloat16)
        out = model(c, a, c, a, params={})
        ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
        ld["loss"].backward()
        res[cat] = (float(ld["loss"].detach()), {n: p.grad.clone() for n, p in model.named_parameters()})
        st = model.store
        for stack in (model.decoder.decoder, model.decoder.hierarchical_decoder):
            ws = [st.index[id(L.linear_global.weight)] for L in stack.layers]
            bs = [st.index[id(L.linear_global.bias)] for L in stack.layers]
            assert all(ws[i + 1][0] == ws[i][0] + ws[i][1] for i in range(len(ws) - 1)), "weights not adjacent"
            assert all(bs[i + 1][0] == bs[i][0] + bs[i][1] for i in range(len(bs) - 1)), "biases not adjacent"
        assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
        for k, v in model.state_dict().items():
            assert torch.equal(v.cpu().float(), sd[k].float()), k
    assert res[True][0] == res[False][0]
    for n in res[True][1]:
        if "linear_global" in n:        # the conditioning linears themselves: same products, fp32 sums (their dg inputs differ by dz's rounding upstream)
            assert H.rel_l2(res[True][1][n], res[False][1][n]) < 1e-2, n
    worst, name = max((H.rel_l2(res[True][1][n], res[False][1][n]), n) for n in res[True][1])
    assert worst < 3e-2, (worst, name)


def test_fused_attention_path_matches_unfused_with_emulated_ops(emulated_ops):
    """bf16 compute, every layer's attention sub-block through attn_pack / attn_block_fwd (packed tiles of the first
    encoder stage, dense key-masked group stages, the live-prefix decoder stage): with the emulated ops the fused call is
    the composition of the unfused ones, so loss and gradients must be IDENTICAL - this pins the wiring (which tensors
    the forward hands the unfused backward, dropout sites, masks / tiles, the per-layer weight image)"""
    from deepsvg_amd.synthetic import make_batch
    import deepsvg_amd.functional as Fn
    from deepsvg_amd import ops
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 2
    cfg.dropout = 0.1
    c, a = make_batch(6, seed=5)
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), 8)
    res, calls = {}, {}
    saved = (Fn.ATTN_MIN_ROWS, Fn.FFN_MIN_ROWS, ops.attn_block_fwd)
    Fn.ATTN_MIN_ROWS, Fn.FFN_MIN_ROWS = 0, 1 << 40
    try:
        for fused in (True, False):
            n_calls = [0]

            def counted(*args, _f=saved[2], **kw):
                n_calls[0] += 1
                return _f(*args, **kw)
            ops.attn_block_fwd = counted
            torch.manual_seed(3)
            model = deepsvg_amd.SVGTransformer(cfg).train()
            model.load_state_dict(sd)
            model.set_compute_dtype(torch.bfloat16)
            if not fused:
                model.store._attn_setup = lambda device: None
            out = model(c, a, c, a, params={})
            assert (model.store._attn is not None) == fused
            ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
            ld["loss"].backward()
            res[fused] = (float(ld["loss"]), {n: p.grad.clone() for n, p in model.named_parameters()})
            calls[fused] = n_calls[0]
    finally:
        Fn.ATTN_MIN_ROWS, Fn.FFN_MIN_ROWS, ops.attn_block_fwd = saved
    assert calls[True] == 8 and calls[False] == 0        # 4 stacks x 2 layers
    assert res[True][0] == res[False][0]
    for n in res[True][1]:
        assert torch.equal(res[True][1][n], res[False][1][n]), n


def test_fused_group_stage_path_matches_unfused_with_emulated_ops(emulated_ops):
    """bf16 compute: the layers of the two short-sequence stacks (hierarchical_encoder with its visibility key masks,
    hierarchical_decoder with the per-icon conditioning row) run through gs_pack / gs_layer_fwd / gs_layer_bwd.  With the
    emulated ops the fused calls are the composition of the unfused ones, so loss and gradients must be IDENTICAL - this
    pins the wiring: saved tensors, dropout sites, key masks, the conditioning row and its gradient, which operand pairs
    reach the four weight-gradient GEMMs, the LayerNorm parameter gradients"""
    from deepsvg_amd.synthetic import make_batch
    from deepsvg_amd import ops
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 2
    cfg.dropout = 0.1
    c, a = make_batch(6, seed=5)
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), 8)
    import deepsvg_amd.functional as Fn
    res, calls = {}, {}
    names = ("gs_layer_fwd", "gs_layer_bwd", "gs_stack_fwd", "gs_stack_bwd")
    saved = tuple(getattr(ops, n) for n in names)
    saved_knob = Fn.GS_STACK
    try:
        # "stack": ONE launch per stack and direction (round 6, functional.GsStackFn); "layer": one per layer (LayerFn's route)
        for fused in ("stack", "layer", False):
            n_calls = [0, 0, 0, 0]

            def counted(k, f):
                def g(*args, **kw):
                    n_calls[k] += 1
                    return f(*args, **kw)
                return g
            for k, n in enumerate(names):
                setattr(ops, n, counted(k, saved[k]))
            Fn.GS_STACK = fused == "stack"
            torch.manual_seed(3)
            model = deepsvg_amd.SVGTransformer(cfg).train()
            model.load_state_dict(sd)
            model.set_compute_dtype(torch.bfloat16)
            if not fused:
                model.store._gs_setup = lambda device: setattr(model.store, "_gs", None)
            out = model(c, a, c, a, params={})
            assert (model.store._gs is not None) == bool(fused)
            ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
            ld["loss"].backward()
            res[fused] = (float(ld["loss"]), {n: p.grad.clone() for n, p in model.named_parameters()})
            calls[fused] = tuple(n_calls)
    finally:
        for n, f in zip(names, saved):
            setattr(ops, n, f)
        Fn.GS_STACK = saved_knob
    # 2 group stacks x 2 layers, forward and backward
    assert calls["stack"] == (0, 0, 2, 2) and calls["layer"] == (4, 4, 0, 0) and calls[False] == (0, 0, 0, 0)
    for fused in ("stack", "layer"):
        assert abs(res[fused][0] - res[False][0]) <= 1e-6 * abs(res[False][0])
        assert all(g is not None for g in res[fused][1].values())
        for n in res[fused][1]:
            assert torch.equal(res[fused][1][n], res[False][1][n]), (fused, n)
    # inference call: one launch per layer as well, nothing saved
    model.eval()
    with torch.no_grad():
        model(c, a, c, a, params={})


def test_fused_argument_head_path_with_emulated_ops(emulated_ops):
    """bf16 compute: SVGLoss takes loss_args from head_lse / head_dlogits (the logit tile never stored) and
    greedy_sample(temperature=0) its arguments from head_argmax.  With the emulated ops the only difference to the unfused
    pair is that the logits are not rounded to bf16 in between: loss and gradients agree to that rounding, the decoded
    arguments to logit ties."""
    from deepsvg_amd.synthetic import make_batch
    from deepsvg_amd import ops
    import deepsvg_amd.functional as Fn
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1
    c, a = make_batch(6, seed=11)
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), 9)
    res, calls = {}, {}
    saved = (ops.head_lse, ops.head_dlogits, Fn.HEAD_FUSED)
    try:
        for fused in (True, False):
            n_calls = [0, 0]

            def counted_f(*args, _f=saved[0], **kw):
                n_calls[0] += 1
                return _f(*args, **kw)

            def counted_b(*args, _f=saved[1], **kw):
                n_calls[1] += 1
                return _f(*args, **kw)
            ops.head_lse, ops.head_dlogits, Fn.HEAD_FUSED = counted_f, counted_b, fused
            model = deepsvg_amd.SVGTransformer(cfg).eval()
            model.load_state_dict(sd)
            model.set_compute_dtype(torch.bfloat16)
            out = model(c, a, c, a, params={})
            ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
            ld["loss"].backward()
            assert out.is_pending("args_logits")
            res[fused] = ({k: float(v) for k, v in ld.items()}, {n: p.grad.clone() for n, p in model.named_parameters()})
            calls[fused] = tuple(n_calls)
    finally:
        ops.head_lse, ops.head_dlogits, Fn.HEAD_FUSED = saved
    assert calls[True] == (1, 1) and calls[False] == (0, 0)
    assert abs(res[True][0]["loss_args"] - res[False][0]["loss_args"]) <= 3e-3 * abs(res[False][0]["loss_args"])
    assert res[True][0]["loss_cmd"] == res[False][0]["loss_cmd"]
    for n in res[True][1]:
        assert H.rel_l2(res[True][1][n], res[False][1][n]) < 2e-2, n
    # decoding
    model = deepsvg_amd.SVGTransformer(cfg).eval()
    model.load_state_dict(sd)
    model.set_compute_dtype(torch.bfloat16)
    n_arg = [0]
    saved_am = ops.head_argmax

    def counted_am(*args, **kw):
        n_arg[0] += 1
        return saved_am(*args, **kw)
    ops.head_argmax = counted_am
    try:
        cy, ay = model.greedy_sample(c, a, temperature=0, concat_groups=False)
    finally:
        ops.head_argmax = saved_am
    assert n_arg[0] == 1
    with torch.no_grad():
        out = model(c, a, None, None, return_tgt=False)
        assert out.is_pending("args_logits")
        cy2, ay2 = model._sample(out["command_logits"], out["args_logits"], 0)
        ay2 = ay2 - 1
        vis = (torch.softmax(out["visibility_logits"].float(), dim=-1)[..., 1] > 0.7).squeeze(-1)
        cy2, ay2 = model._make_valid(cy2, ay2, vis)
    assert torch.equal(cy, cy2)
    assert (ay != ay2).float().mean().item() < 5e-3


def test_remainder_sequences_on_the_group_stage_kernel_with_emulated_ops(emulated_ops, monkeypatch):
    """bf16 training forward of the second decoder stage: the sequences beyond a multiple of SEQ_ROUND run on gs_layer_fwd
    (seq_base, fused-FFN output format, written into row slices of the saved tensors), the others on attn_block_fwd +
    ffn_fwd; the backward pass reads the joint buffers.  Shrunk thresholds so that a 6-icon batch takes the path.  Against
    the unsplit run: same loss and gradients up to the bf16 rounding differences of the two formulations."""
    from deepsvg_amd.synthetic import make_batch
    from deepsvg_amd import ops
    import deepsvg_amd.functional as Fn
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 2
    cfg.dropout = 0.0           # (the two paths draw the hidden-site dropout masks differently: compare without)
    c, a = make_batch(6, seed=21)
    sd = H.weights_for(deepsvg_amd.SVGTransformer(cfg), 10)
    monkeypatch.setattr(Fn, "FFN_MIN_ROWS", 64)
    monkeypatch.setattr(Fn, "ATTN_MIN_ROWS", 64)
    monkeypatch.setattr(Fn, "SEQ_ROUND", 8)
    res, calls = {}, {}
    saved = ops.gs_layer_fwd
    try:
        for rem_max in (7, 0):
            n_calls = []

            def counted(*args, _f=saved, **kw):
                n_calls.append((kw.get("seq_base", 0), bool(kw.get("ffn_format", False))))
                return _f(*args, **kw)
            ops.gs_layer_fwd = counted
            monkeypatch.setattr(Fn, "GS_REMAINDER", rem_max)
            torch.manual_seed(3)
            model = deepsvg_amd.SVGTransformer(cfg).train()
            model.load_state_dict(sd)
            model.set_compute_dtype(torch.bfloat16)
            out = model(c, a, c, a, params={})
            ld = deepsvg_amd.SVGLoss(cfg)(out, None, weights=O.DEFAULT_WEIGHTS)
            ld["loss"].backward()
            res[rem_max] = (float(ld["loss"].detach()), {n: p.grad.clone() for n, p in model.named_parameters()})
            calls[rem_max] = [cc for cc in n_calls if cc[1]]
            n_run = model.last_live
    finally:
        ops.gs_layer_fwd = saved
    assert len(calls[7]) == 2 and all(b > 0 and b % 8 == 0 for b, _ in calls[7]), calls     # the stage's two layers
    assert calls[0] == []
    assert abs(res[7][0] - res[0][0]) <= 2e-3 * abs(res[0][0])
    for n in res[7][1]:
        assert H.rel_l2(res[7][1][n], res[0][1][n]) < 3e-2, n


def test_split_k_policy_fills_the_chip():
    """ops.split_k_for: ~256 workgroups (one per CU, slices in multiples of 8 = grouped per XCD); products whose tile count
    leaves more than a fifth of the CUs without a workgroup at that target (12 tiles: 16 slices = 192) go to ~480 workgroups on
    the high-occupancy variant instead (large stages only; an explicit target is taken as given)."""
    from deepsvg_amd import ops
    assert ops.split_k_for(512, 256, 63488) == 32 and ops.split_k_for(256, 512, 40960) == 32
    assert ops.split_k_for(256, 256, 63488) == 64
    assert ops.split_k_for(768, 256, 63488) == 40 and ops.split_k_for(768, 256, 40960) == 40
    assert ops.split_k_for(768, 256, 63488, target_blocks=256) == 16
    assert ops.split_k_for(768, 256, 4096) == 16           # the 4096-row stages keep the one-workgroup-per-CU schedule
    assert ops.split_k_for(1024, 256, 4096) == 16 and ops.split_k_for(256, 256, 512) == 4


def test_trainer_gradless_slots_follow_call_shapes_and_reflattening(emulated_ops):
    """TrainStep zeroes the flat-gradient slots of parameters that received no gradient (the norm / AdamW / all-reduce read the
    whole buffer).  The bookkeeping is per parameter id and per view of the gradient buffer: it must survive alternating call
    shapes (with / without separate decoder-side tensors) and must be rebuilt when the ParamStore re-flattens (a re-assigned
    .data): a poisoned slot is zero after every step, and the buffer equals a fresh trainer's on the same call."""
    from deepsvg_amd.synthetic import make_batch
    from deepsvg_amd.trainer import TrainStep
    cfg = H.build_cfg("hier")
    cfg.n_layers = cfg.n_layers_decode = 1

    def build():
        torch.manual_seed(0)
        m = deepsvg_amd.SVGTransformer(cfg)
        m.load_state_dict(H.weights_for(m, 11))
        m.unused_probe = torch.nn.Parameter(torch.ones(24))       # never reached by forward: gradless in every call shape
        return m.eval()
    c, a = make_batch(3, seed=4)
    shapes = [dict(), dict(commands_dec=c.clone(), args_dec=a.clone()), dict(), dict(commands_dec=c.clone(), args_dec=a.clone())]
    model = build()
    ts = TrainStep(model, deepsvg_amd.SVGLoss(cfg), lr=0.0)

    def slot(m):
        st = m.store
        o, n, _ = st.index[id(m.unused_probe)]
        return st.grad_buffer(0)[o:o + n]
    bufs = []
    for k, kw in enumerate(shapes):
        if k:
            slot(model).fill_(7.0)                      # a stale gradient of "an earlier step"
        ts.step(c, a, **kw)
        assert id(model.unused_probe) in ts._gradless_ids
        assert float(slot(model).abs().max()) == 0.0, k
        bufs.append(model.store.grad_buffer(0).clone())
    fresh = build()
    tf = TrainStep(fresh, deepsvg_amd.SVGLoss(cfg), lr=0.0)
    tf.step(c, a, **shapes[3])
    assert torch.equal(bufs[3], fresh.store.grad_buffer(0))
    assert torch.equal(bufs[1], bufs[3]) and torch.equal(bufs[0], bufs[2])
    # re-flatten: the ids and views cached by the trainer are stale and must be dropped, not reused
    gen, old_ids = model.store.generation, set(ts._gradless_ids)
    p = dict(model.named_parameters())["encoder.encoder.layers.0.linear1.weight"]
    p.data = p.data.clone()
    ts.step(c, a)
    assert model.store.generation == gen + 1 and ts._store_generation == gen + 1
    slot(model).fill_(7.0)
    ts.step(c, a)
    assert float(slot(model).abs().max()) == 0.0
    assert ts._gradless_ids == old_ids                  # (same parameter objects; the VIEWS were rebuilt on the new buffer)
    assert all(v.data_ptr() >= model.store.grad_buffer(0).data_ptr() for v in ts._gradless_slots)
    assert torch.equal(model.store.grad_buffer(0), bufs[0])


def test_group_scope_nests():
    """ops.GROUP (one grouped weight-gradient launch per block): a block inside an open block - a per-layer scope inside the
    stack-level scope of functional.STACK_GROUP - must not close the outer one; only the outermost exit launches"""
    from deepsvg_amd import ops
    key = ops._stream_key()
    assert ops._GroupScope.depth.get(key, 0) == 0
    with ops.GROUP:
        assert ops._GroupScope.depth[key] == 1
        with ops.GROUP:
            assert ops._GroupScope.depth[key] == 2
        assert ops._GroupScope.depth[key] == 1          # still open
    assert ops._GroupScope.depth.get(key, 0) == 0


def test_head_kpad_tail_guard():
    """functional._tail_is_own_bias_or_slack: the extended head-weight view may only run over the head's own bias and the zero
    slack behind the last parameter"""
    import torch
    from deepsvg_amd import functional as Fn

    class St:
        pass
    w, b, other = torch.nn.Parameter(torch.zeros(10, 4)), torch.nn.Parameter(torch.zeros(10)), torch.nn.Parameter(torch.zeros(64))
    st = St()
    st.params = [other, w, b]
    st.index = {id(other): (0, 64, (64,)), id(w): (64, 40, (10, 4)), id(b): (104, 10, (10,))}
    assert Fn._tail_is_own_bias_or_slack(st, w, b, 40 + 24)          # runs over its own bias and the slack
    st.params = [w, b, other]
    st.index = {id(w): (0, 40, (10, 4)), id(b): (40, 10, (10,)), id(other): (56, 64, (64,))}
    assert Fn._tail_is_own_bias_or_slack(st, w, b, 40 + 12)          # ends inside the bias
    assert not Fn._tail_is_own_bias_or_slack(st, w, b, 40 + 24)      # would read another parameter
