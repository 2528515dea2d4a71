"""Generates tests/golden/greedy_sample.npz: outputs of the REAL reference's one-shot `SVGTransformer.greedy_sample`
(+ `_make_valid`, `_threshold_sample`, the concat_groups squeeze; deepsvg/model/model.py:414-459, model/utils.py:75-84)
on the inputs / weights of two existing fixtures.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sample.py

Per case (prefix `<case>/`):
  commands, args, label, wseed, kind          inputs (weights = det_state_dict(wseed))
  cy, ay                                      greedy_sample(commands, args, None, None[, label], concat_groups=False)
  cy_z, ay_z                                  greedy_sample(z=<encode_mode output, batch-first>[, label=...])
  cmd_gap, args_gap                           top-2 logit gaps (the reference draws from Categorical(logits / 1e-4):
                                              the arg-max except on near-ties; checkers skip slots with gap < 2e-3)
  cat_c, cat_a, cat_len                       concat_groups=True, one icon per call (the reference's reshape needs equal
                                              token counts across the batch), concatenated; cat_len[i] tokens each
  hz, hl                                      per-group latents (batch-first) + FORCED visibility logits: icon 0 keeps
                                              one visible group, icon 1 none, the others the model's own
  cy_h, ay_h                                  greedy_sample(z=hz, hierarch_logits=hl[, label], concat_groups=False)
The oracle's greedy_sample restatement is checked against every one of these here (outside the near-ties).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from deepsvg.model.model import SVGTransformer as RefModel          # noqa: E402
import deepsvg.model.model as ref_model_mod                          # noqa: E402
from deepsvg_amd.synthetic import make_batch, det_state_dict       # noqa: E402
from oracle import svg_transformer_oracle as O                        # noqa: E402
from tests.golden.make_golden import build_cfg                        # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
TIE = 2e-3


def _same_outside_ties(got_c, got_a, want_c, want_a, cmd_gap, args_gap):
    okc = cmd_gap > TIE
    assert torch.equal(got_c[okc], want_c[okc])
    # a flipped command changes which argument slots are valid: compare arguments where the command is safe too
    oka = (args_gap > TIE) & okc.unsqueeze(-1)
    assert torch.equal(got_a[oka], want_a[oka])


def run_case(tag, kind, n, seed, wseed, rec):
    torch.manual_seed(0)
    cfg = build_cfg(kind)
    model = RefModel(cfg)
    sd = det_state_dict(model, seed=wseed)
    model.load_state_dict(sd)
    model.eval()
    commands, args = make_batch(n, G=cfg.max_num_groups, S=cfg.max_seq_len, seed=seed)
    label = None
    if cfg.label_condition:
        label = torch.randint(0, cfg.n_labels, (n,), generator=torch.Generator().manual_seed(seed + 5))
    kw = {"label": label} if label is not None else {}
    eps = None
    if cfg.use_vae:     # fixed reparametrisation noise (model.py:185), as tests/golden/make_golden.py does
        eps = torch.randn(1, 1, n, cfg.dim_z, generator=torch.Generator().manual_seed(seed + 77))
        ref_model_mod.torch.randn_like = lambda t: (eps if t.shape[-2] == n else eps[:, :, run_case.icon:run_case.icon + 1]).to(t.dtype)
    try:
        _run(tag, kind, n, wseed, rec, cfg, model, sd, commands, args, label, kw, eps)
    finally:
        ref_model_mod.torch.randn_like = torch.randn_like


def _run(tag, kind, n, wseed, rec, cfg, model, sd, commands, args, label, kw, eps):
    with torch.no_grad():
        torch.manual_seed(1)
        cy, ay = model.greedy_sample(commands, args, None, None, concat_groups=False, **kw)
        z = model(commands, args, None, None, encode_mode=True, **kw)                 # (1, 1, N, dz)
        zb = z.permute(2, 1, 0, 3).contiguous()                                       # batch-first for `z=`
        torch.manual_seed(2)
        cy_z, ay_z = model.greedy_sample(None, None, None, None, z=zb, concat_groups=False, **kw)
        out = model(commands, args, None, None, return_tgt=False, **kw)
        t2c, t2a = out["command_logits"].topk(2, -1).values, out["args_logits"].topk(2, -1).values
        cmd_gap, args_gap = t2c[..., 0] - t2c[..., 1], t2a[..., 0] - t2a[..., 1]
        cat_c, cat_a, cat_len = [], [], []
        for i in range(n):
            torch.manual_seed(3 + i)
            run_case.icon = i
            kwi = {"label": label[i:i + 1]} if label is not None else {}
            c1, a1 = model.greedy_sample(commands[i:i + 1], args[i:i + 1], None, None, concat_groups=True, **kwi)
            cat_c.append(c1[0])
            cat_a.append(a1[0])
            cat_len.append(c1.shape[1])
        hl, hz = model(commands, args, None, None, return_hierarch=True, return_tgt=False, **kw)       # (1, G, N, 2), (1, G, N, dz)
        hl = hl.clone()
        hl[0, :, 0, 0], hl[0, :, 0, 1] = 4.0, -4.0                                    # icon 0: only group 3 visible
        hl[0, 3, 0, 0], hl[0, 3, 0, 1] = -4.0, 4.0
        hl[0, :, 1, 0], hl[0, :, 1, 1] = 4.0, -4.0                                    # icon 1: nothing visible
        hzb = hz.permute(2, 1, 0, 3).contiguous()                                     # (N, G, 1, dz)
        torch.manual_seed(4)
        cy_h, ay_h = model.greedy_sample(None, None, None, None, z=hzb, hierarch_logits=hl, concat_groups=False, **kw)
        out_h = model(None, None, None, None, z=hzb, hierarch_logits=hl, return_tgt=False, **kw)
        t2c_h, t2a_h = out_h["command_logits"].topk(2, -1).values, out_h["args_logits"].topk(2, -1).values
    # the same z must sample the same icons
    _same_outside_ties(cy_z, ay_z, cy, ay, cmd_gap, args_gap)
    inv = torch.tensor([0] + [4] * 30)
    assert all(torch.equal(cy_h[0, g], inv) for g in range(8) if g != 3) and not torch.equal(cy_h[0, 3], inv)
    assert torch.equal(cy_h[1], inv.expand(8, 31)) and bool((ay_h[1] == -1).all())

    # ---- the oracle's restatement against the live reference ----
    o_c, o_a, o_cg, o_ag = O.greedy_sample(sd, cfg, commands, args, label=label, concat_groups=False, eps=eps)
    assert (o_cg - cmd_gap).abs().max().item() < 1e-4 and (o_ag - args_gap).abs().max().item() < 1e-4
    _same_outside_ties(o_c, o_a, cy, ay, cmd_gap, args_gap)
    o_c, o_a, _, _ = O.greedy_sample(sd, cfg, label=label, z=zb, concat_groups=False)
    _same_outside_ties(o_c, o_a, cy_z, ay_z, cmd_gap, args_gap)
    o_c, o_a, _, _ = O.greedy_sample(sd, cfg, label=label, z=hzb, hierarch_logits=hl, concat_groups=False)
    _same_outside_ties(o_c, o_a, cy_h, ay_h, t2c_h[..., 0] - t2c_h[..., 1], t2a_h[..., 0] - t2a_h[..., 1])
    for i in range(n):
        o_c, o_a, _, _ = O.greedy_sample(sd, cfg, commands[i:i + 1], args[i:i + 1],
                                         label=label[i:i + 1] if label is not None else None, concat_groups=True,
                                         eps=eps[:, :, i:i + 1] if eps is not None else None)
        assert o_c.shape[1] == cat_len[i]
        if bool((cmd_gap[i] > TIE).all()) and bool((args_gap[i] > TIE).all()):
            assert torch.equal(o_c[0], cat_c[i]) and torch.equal(o_a[0], cat_a[i])

    rec.update({
        f"{tag}/kind": kind, f"{tag}/wseed": wseed,
        f"{tag}/commands": commands.numpy().astype(np.float32), f"{tag}/args": args.numpy().astype(np.float32),
        f"{tag}/cy": cy.numpy().astype(np.int16), f"{tag}/ay": ay.numpy().astype(np.int16),
        f"{tag}/cy_z": cy_z.numpy().astype(np.int16), f"{tag}/ay_z": ay_z.numpy().astype(np.int16),
        f"{tag}/cmd_gap": cmd_gap.numpy().astype(np.float32), f"{tag}/args_gap": args_gap.numpy().astype(np.float16),
        f"{tag}/cat_c": torch.cat(cat_c).numpy().astype(np.int16), f"{tag}/cat_a": torch.cat(cat_a).numpy().astype(np.int16),
        f"{tag}/cat_len": np.array(cat_len, dtype=np.int32),
        f"{tag}/hz": hzb.numpy(), f"{tag}/hl": hl.numpy(),
        f"{tag}/cy_h": cy_h.numpy().astype(np.int16), f"{tag}/ay_h": ay_h.numpy().astype(np.int16),
        f"{tag}/cmd_gap_h": (t2c_h[..., 0] - t2c_h[..., 1]).numpy().astype(np.float32),
        f"{tag}/args_gap_h": (t2a_h[..., 0] - t2a_h[..., 1]).numpy().astype(np.float16),
        f"{tag}/z": zb.numpy(),
    })
    if label is not None:
        rec[f"{tag}/label"] = label.numpy()
    if eps is not None:
        rec[f"{tag}/eps"] = eps.numpy()
    n_tie = int((args_gap <= TIE).sum()) + int((cmd_gap <= TIE).sum())
    print(f"{tag}: {n} icons, visible groups per icon {[(cy[i, :, 1] != 4).sum().item() for i in range(n)]}, "
          f"concat lengths {cat_len}, near-tie slots {n_tie}")


if __name__ == "__main__":
    rec = {}
    run_case("hier5", "hier", 5, 12, 4321, rec)           # = hier_ordered_n5's inputs and weights
    run_case("fonts4", "fonts", 4, 15, 1234, rec)         # = fonts_label_n4's (label_condition, dim_z = 128)
    np.savez_compressed(os.path.join(OUT, "greedy_sample.npz"), **rec)
