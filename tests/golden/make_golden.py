"""Generates tests/golden/*.npz by running the REAL reference (imported read-only from /root/reference) on
seeded synthetic inputs with deterministic weights.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is stored (small enough to commit): the inputs, the weight seed, full command/visibility logits, a strided
sample + checksums of args_logits, the latent z, every loss term, and per-parameter gradient norms + a 16-value
sample of each gradient.  Weights are NOT stored: deepsvg_amd.synthetic.det_state_dict(seed) regenerates them.

loss_cmd: the reference's `_get_padding_mask(extended=True)` adds overlapping views in place
(deepsvg/model/utils.py:28), whose result is implementation-defined.  Two values are
recorded: `loss_cmd_ref_aliased` (the unmodified reference on this CPU) and `loss_cmd` / `loss` / the
gradients with `_get_padding_mask` patched to the non-aliased semantics (clone before add).
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
from deepsvg.model.loss import SVGLoss as RefLoss                    # noqa: E402
import deepsvg.model.loss as ref_loss_mod                            # noqa: E402
import deepsvg.model.model as ref_model_mod                          # noqa: E402
from deepsvg.model import config as ref_cfg                          # noqa: E402
from deepsvg.difflib.tensor import SVGTensor                         # noqa: E402

from deepsvg_amd.synthetic import make_batch, make_batch_onestage, det_state_dict   # noqa: E402
from oracle import svg_transformer_oracle as O                        # noqa: E402

WEIGHTS = dict(O.DEFAULT_WEIGHTS)
OUT = os.path.dirname(os.path.abspath(__file__))


def _canonical_padding_mask(commands, seq_dim=0, extended=False):
    with torch.no_grad():
        pm = (commands == SVGTensor.COMMANDS_SIMPLIFIED.index("EOS")).cumsum(dim=seq_dim) == 0
        pm = pm.float()
        if extended:
            S = commands.size(seq_dim)
            src = torch.narrow(pm, seq_dim, 0, S - 3).clone()
            torch.narrow(pm, seq_dim, 3, S - 3).add_(src).clamp_(max=1)
        return pm.unsqueeze(-1) if seq_dim == 0 else pm


def build_cfg(kind):
    if kind == "hier":
        cfg = ref_cfg.Hierarchical()
        cfg.use_vae = False
    elif kind == "hier_vae":
        cfg = ref_cfg.Hierarchical()
    elif kind == "onestage":
        cfg = ref_cfg.OneStageOneShot()
        cfg.max_total_len = 50
        cfg.use_vae = False
    elif kind == "onestage240":     # OneStageOneShot as the reference defines it: max_total_len = 240 (242-token sequences)
        cfg = ref_cfg.OneStageOneShot()
        cfg.use_vae = False
    elif kind == "sketchformer240":  # Sketchformer as the reference defines it (241-token causal decoder)
        cfg = ref_cfg.Sketchformer()
        cfg.use_vae = False
    elif kind == "onestage_label":  # SURVEY.md C4 with the optional label conditioning (one-stage, VAE, n_labels = 100)
        cfg = ref_cfg.OneStageOneShot()
        cfg.max_total_len = 50
        cfg.label_condition = True
    elif kind == "hier_rel":        # two-stage one-shot model on relative targets (args_rel, 2 * args_dim classes)
        cfg = ref_cfg.Hierarchical()
        cfg.rel_targets = True
        cfg.use_vae = False
    elif kind == "sketchformer":    # deepsvg/model/config.py:74-80 (transformer, autoregressive, one-stage, rel. targets)
        cfg = ref_cfg.Sketchformer()
        cfg.max_total_len = 50
        cfg.use_vae = False
    elif kind == "selfmatch":       # deepsvg/model/config.py:101-108
        cfg = ref_cfg.HierarchicalSelfMatching()
        cfg.use_vae = False
    elif kind == "fonts":           # ModelConfig of configs/deepsvg/hierarchical_ordered_fonts.py:4-9
        cfg = ref_cfg.Hierarchical()
        cfg.label_condition = True
        cfg.dim_z = 128
    else:
        raise ValueError(kind)
    return cfg


def run_case(name, kind, n, seed, wseed):
    torch.manual_seed(0)
    cfg = build_cfg(kind)
    model = RefModel(cfg)
    sd = det_state_dict(model, seed=wseed)
    model.load_state_dict(sd)
    if kind in ("onestage", "sketchformer", "onestage_label", "onestage240", "sketchformer240"):
        commands, args = make_batch_onestage(n, total_len=cfg.max_total_len, seed=seed)
    else:
        commands, args = make_batch(n, G=cfg.max_num_groups, S=cfg.max_seq_len, seed=seed)
    args_dec = args
    if cfg.rel_targets:             # decoder side takes args_rel(_grouped) (model/config.py:52-53): SVGTensor.get_relative_args
        args_dec = torch.stack([torch.stack([SVGTensor.from_cmd_args(commands[i, g], args[i, g]).get_relative_args()
                                             for g in range(commands.shape[1])]) for i in range(n)])
    label = None
    if cfg.label_condition:
        label = torch.randint(0, cfg.n_labels, (n,), generator=torch.Generator().manual_seed(seed + 5))
    eps = None
    if cfg.use_vae:
        g = torch.Generator().manual_seed(seed + 77)
        eps = torch.randn(1, 1, n, cfg.dim_z, generator=g)
        ref_model_mod.torch.randn_like = lambda t: eps.to(t.dtype)      # model.py:185
    captured = {}
    if cfg.self_match:
        # perfect_matching builds its command mask with the aliasing in-place add too (model.py:315 -> utils.py:28):
        # canonical semantics for the whole case; the assignment it returns is recorded
        ref_model_mod._get_padding_mask = _canonical_padding_mask
        orig_pm = model.perfect_matching

        def _pm(*a, **k):
            r = orig_pm(*a, **k)
            captured["assignment"] = r.squeeze(-1).squeeze(-1).clone()
            return r
        model.perfect_matching = _pm
    try:
        # ---- eval forward: logits ----
        model.eval()
        with torch.no_grad():
            out = model(commands, args, commands, args_dec, label=label, params={})
            z = model(commands, args, commands, args_dec, label=label, encode_mode=True)
            hier = None
            if cfg.decode_stages == 2:
                # GUI path (model.py:246-261,382-383): per-group latents out, then back in with hierarch_logits
                hier = model(commands, args, commands, args_dec, label=label, return_hierarch=True)
                z_groups = hier[1].permute(2, 1, 0, 3).contiguous()     # batch-first for `z=` (model.py:369)
                out2 = model(None, None, commands, args, label=label, z=z_groups, hierarch_logits=hier[0],
                             return_tgt=False)       # (with a VAE, return_tgt=True needs mu: model.py:408-410)
                plain = model(commands, args, commands, args, label=label, return_tgt=False) if cfg.self_match else out
                for k in plain:
                    if k.endswith("logits"):
                        assert torch.allclose(plain[k], out2[k], atol=1e-6), k
        # ---- train mode, every dropout p = 0: loss + grads ----
        model.train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
                m.dropout = 0.0
        loss_fn = RefLoss(cfg)
        out_t = model(commands, args, commands, args_dec, label=label, params={})
        ld_alias = loss_fn(out_t, None, weights=WEIGHTS)
        orig = ref_loss_mod._get_padding_mask
        ref_loss_mod._get_padding_mask = _canonical_padding_mask
        try:
            model.zero_grad()
            out_t = model(commands, args, commands, args_dec, label=label, params={})
            ld = loss_fn(out_t, None, weights=WEIGHTS)
            ld["loss"].backward()
        finally:
            ref_loss_mod._get_padding_mask = orig
    finally:
        if cfg.use_vae:
            ref_model_mod.torch.randn_like = torch.randn_like
        if cfg.self_match:
            from deepsvg.model import utils as _ref_utils
            ref_model_mod._get_padding_mask = _ref_utils._get_padding_mask

    # ---- the oracle restatement must agree with the live reference ----
    o_out = O.forward(sd, cfg, commands, args, commands, args_dec, eps=eps, label=label)
    if hier is not None:
        o_hier = O.forward(sd, cfg, commands, args, commands, args, eps=eps, label=label, return_hierarch=True)
        assert (o_hier[0] - hier[0]).abs().max().item() < 2e-5 and (o_hier[1] - hier[1]).abs().max().item() < 2e-5
        o_out2 = O.forward(sd, cfg, None, None, commands, args, z=o_hier[1].permute(2, 1, 0, 3), label=label,
                           hierarch_logits=o_hier[0])
        if not cfg.self_match:
            assert (o_out2["args_logits"] - out["args_logits"]).abs().max().item() < 2e-5
    for k in ("command_logits", "args_logits", "visibility_logits"):
        if k in out:
            err = (o_out[k] - out[k]).abs().max().item()
            assert err < 2e-5, (name, k, err)
    _, o_ld, o_grads = O.loss_and_grads(sd, cfg, commands, args, WEIGHTS, eps=eps, label=label, args_dec=args_dec)
    for k in ld:
        assert abs(o_ld[k].item() - ld[k].item()) < 2e-5 * max(1.0, abs(ld[k].item())), (name, k, o_ld[k].item(), ld[k].item())
    worst = 0.0
    for pname, p in model.named_parameters():
        g_ref, g_o = p.grad, o_grads[pname]
        rel = (g_ref - g_o).norm().item() / max(g_ref.norm().item(), 1e-12)
        worst = max(worst, rel)
    assert worst < 1e-4, (name, "grad rel", worst)

    al = out["args_logits"].reshape(-1)
    stride = 997
    rec = {
        "kind": kind, "n": n, "seed": seed, "wseed": wseed,
        "commands": commands.numpy().astype(np.float32), "args": args.numpy().astype(np.float32),
        "command_logits": out["command_logits"].numpy(),
        "args_logits_sample": al[::stride].numpy(), "args_logits_stride": stride,
        "args_logits_sum": np.float64(al.double().sum().item()),
        "args_logits_abssum": np.float64(al.double().abs().sum().item()),
        "args_argmax": out["args_logits"].argmax(-1).numpy().astype(np.int16),
        "z": z.numpy(),
    }
    if "visibility_logits" in out:
        rec["visibility_logits"] = out["visibility_logits"].numpy()
    if eps is not None:
        rec["eps"] = eps.numpy()
    if label is not None:
        rec["label"] = label.numpy()
    if cfg.rel_targets:
        rec["args_dec"] = args_dec.numpy().astype(np.float32)
    if cfg.pred_mode == "autoregressive":
        # autoregressive sampling (model.py:424-441), one icon at a time as the reference decodes; stored without the
        # concat_groups squeeze so that the icons stack
        model.eval()
        cs, as_ = [], []
        with torch.no_grad():
            for i in range(n):
                cy, ay = model.greedy_sample(commands[i:i + 1], args[i:i + 1], None, None, concat_groups=False)
                cs.append(cy)
                as_.append(ay)
        rec["sample_commands"] = torch.cat(cs).numpy().astype(np.int64)
        rec["sample_args"] = torch.cat(as_).numpy().astype(np.int64)
    if "assignment" in captured:
        rec["assignment"] = captured["assignment"].numpy().astype(np.int64)     # (N, Gp), of the last (train) pass
        assert torch.equal(o_out["_assignment"], captured["assignment"])
    if hier is not None:
        rec["hier_logits"] = hier[0].numpy()        # seq-first (1, G, N, 2)
        rec["hier_z"] = hier[1].numpy()             # seq-first (1, G, N, dim_z)
    for k, v in ld.items():
        rec[k] = np.float64(v.item())
    rec["loss_cmd_ref_aliased"] = np.float64(ld_alias["loss_cmd"].item())
    names, norms, samples = [], [], []
    for pname, p in model.named_parameters():
        names.append(pname)
        norms.append(p.grad.double().norm().item())
        flat = p.grad.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 16).long()
        samples.append(flat[idx].numpy())
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms, dtype=np.float64)
    rec["grad_samples"] = np.stack(samples).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: loss={ld['loss'].item():.6f} loss_cmd={ld['loss_cmd'].item():.6f} "
          f"(aliased {ld_alias['loss_cmd'].item():.6f}) oracle-vs-ref grad rel {worst:.2e}")


if __name__ == "__main__":
    if len(sys.argv) > 1:           # regenerate selected cases only: make_golden.py <name> ...
        ALL = {"hier_ordered_n2": ("hier", 2, 11, 1234), "hier_ordered_n5": ("hier", 5, 12, 4321),
               "hier_vae_n3": ("hier_vae", 3, 13, 1234), "onestage50_n3": ("onestage", 3, 14, 1234),
               "fonts_label_n4": ("fonts", 4, 15, 1234), "selfmatch_n6": ("selfmatch", 6, 16, 1234),
               "sketchformer50_n4": ("sketchformer", 4, 17, 1234), "onestage50_label_n3": ("onestage_label", 3, 18, 1234),
               "hier_rel_n3": ("hier_rel", 3, 19, 1234), "onestage240_n2": ("onestage240", 2, 20, 1234),
               "sketchformer240_n2": ("sketchformer240", 2, 21, 1234)}
        for nm in sys.argv[1:]:
            run_case(nm, *ALL[nm])
        sys.exit(0)
    run_case("hier_ordered_n2", "hier", 2, 11, 1234)          # BASELINE config C1 shape
    run_case("hier_ordered_n5", "hier", 5, 12, 4321)
    run_case("hier_vae_n3", "hier_vae", 3, 13, 1234)
    run_case("onestage50_n3", "onestage", 3, 14, 1234)
    run_case("fonts_label_n4", "fonts", 4, 15, 1234)
    run_case("selfmatch_n6", "selfmatch", 6, 16, 1234)
    run_case("sketchformer50_n4", "sketchformer", 4, 17, 1234)
    run_case("onestage50_label_n3", "onestage_label", 3, 18, 1234)
    run_case("hier_rel_n3", "hier_rel", 3, 19, 1234)
    run_case("onestage240_n2", "onestage240", 2, 20, 1234)
    run_case("sketchformer240_n2", "sketchformer240", 2, 21, 1234)
