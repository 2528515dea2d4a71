"""Drop-in on the device: the calling sequence of the reference's training script (deepsvg/train.py), reproduced call by
call on the HIP path WITHOUT deepsvg_amd's own trainer -

    model = cfg.make_model().to(device)                       train.py:29
    optimizer = AdamW(model.parameters(), lr)                 train.py:54
    model = nn.DataParallel(model)                            train.py:74
    output = model(*model_args, params=params)                train.py:94
    loss_dict = loss_fns[i](output, labels, weights=...)      train.py:95   (here: the REFERENCE's SVGLoss restatement,
    loss.backward(); clip_grad_norm_; optimizer.step()        train.py:98-102        reading the lazy args_logits)
    save / load through state_dict()                          train.py:135, train_utils.py:147-152

- and checked against the oracle doing the same on the CPU (dropout off on both sides - cfg.dropout = 0 and the positional
encodings' hard-wired 0.1, positional_encoding.py:26 - because dropout masks are not comparable across implementations,
SURVEY.md 7.3-1)."""
import copy

import pytest
import torch
import torch.nn as nn

import deepsvg_amd
from deepsvg_amd.synthetic import make_batch
from oracle import svg_transformer_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_reference_training_sequence_drives_the_hip_kernels(gpu_device, monkeypatch):
    import deepsvg_amd.model as M
    monkeypatch.setattr(M, "PE_DROPOUT", 0.0)       # (train mode is kept: the calling sequence is train.py's)
    cfg = H.build_cfg("hier")
    cfg.use_vae = False
    cfg.dropout = 0.0
    model = deepsvg_amd.SVGTransformer(cfg)
    sd0 = H.weights_for(model, 11)
    model.load_state_dict(sd0)
    model = model.to(DEV)                                                 # train.py:29
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)   # utils.count_parameters, train.py:40
    assert n_params == sum(v.numel() for k, v in sd0.items() if torch.is_floating_point(v) and "position" not in k
                           and k != "cmd_args_mask")
    lr = 1e-3
    optimizer = torch.optim.AdamW(model.parameters(), lr)                 # train.py:54 (stock optimiser, 244 tensors)
    wrapped = nn.DataParallel(model)                                      # train.py:74 (one visible device)
    wrapped.train()                                                       # train.py:86
    commands, args = make_batch(6, seed=5)
    c, a = commands.to(DEV), args.to(DEV)
    model_args = [c, a, c, a]                                             # model/config.py:50-55
    weights = dict(O.DEFAULT_WEIGHTS)

    # ---- the same two steps on the CPU with stock PyTorch (oracle + torch.optim.AdamW) -----------------------------------
    leaves = {k: v.detach().clone().requires_grad_(torch.is_floating_point(v)) for k, v in sd0.items()}
    ref_params = [v for v in leaves.values() if v.requires_grad]
    ref_opt = torch.optim.AdamW(ref_params, lr)
    ref_losses = []
    for _ in range(2):
        ref_opt.zero_grad()
        ld = O.svg_loss(cfg, O.forward(leaves, cfg, commands, args, commands, args), weights)
        ld["loss"].backward()
        torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
        ref_opt.step()
        ref_losses.append(float(ld["loss"]))

    # ---- train.py:92-106 on the device ------------------------------------------------------------------------------------
    losses = []
    for _ in range(2):
        optimizer.zero_grad()                                             # train.py:92
        output = wrapped(*model_args, params={})                          # train.py:94
        assert output["command_logits"].is_cuda
        # the reference's own loss (restated line by line in the oracle) on the model's result dict: it indexes the dense
        # args_logits with a boolean mask (loss.py:54), i.e. it reads the LAZY entry of the result dict
        ld = O.svg_loss(cfg, output, weights)                             # train.py:95
        ld["loss"].backward()                                             # train.py:98
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)           # train.py:100
        optimizer.step()                                                  # train.py:102
        losses.append(float(ld["loss"]))
    torch.cuda.synchronize()
    for got, want in zip(losses, ref_losses):
        assert abs(got - want) <= 1e-4 * abs(want), (losses, ref_losses)
    assert losses[1] < losses[0], "the second step must see the weights the stock optimiser updated in place"
    sd1 = {k: v.detach().cpu() for k, v in wrapped.module.state_dict().items()}       # train.py:135 (save_ckpt_list)
    assert list(sd1) == list(sd0), "state_dict keys / order must be the reference's"
    for k, v in leaves.items():
        if v.requires_grad:
            # Adam's first updates are lr * g / |g| per element whatever |g| is: an element whose gradient is fp32 summation
            # noise may move the other way (2 lr apart per step), every other element agrees to gradient accuracy
            d = (sd1[k] - v.detach()).abs()
            assert d.max().item() <= 2.05 * lr * 2, (k, d.max().item())
            close = d <= 1e-3 * v.detach().abs() + 0.05 * lr
            assert close.float().mean().item() >= 0.99, (k, close.float().mean().item())

    # ---- checkpoint round trip (train_utils.py:147-152: load_state_dict(strict=False)) into a fresh model ------------------
    fresh = deepsvg_amd.SVGTransformer(cfg)
    missing = fresh.load_state_dict(copy.deepcopy(sd1), strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys
    fresh = fresh.to(DEV).eval()
    wrapped.eval()                                                        # train.py:125
    with torch.no_grad():
        o1 = wrapped(*model_args, params={})
        o2 = fresh(*model_args, params={})
    for k in ("command_logits", "args_logits", "visibility_logits"):
        assert torch.equal(o1[k], o2[k]), f"{k}: a model rebuilt from the checkpoint must reproduce the logits bit for bit"
    # visualisation hook of the training loop (configs/deepsvg/default_icons.py:85): model.module.greedy_sample
    cy, ay = wrapped.module.greedy_sample(c, a, None, None)
    assert cy.dtype == torch.int64 and ay.dtype == torch.int64 and cy.shape[0] == 6
