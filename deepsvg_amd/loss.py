"""Drop-in for deepsvg.model.loss.SVGLoss (deepsvg/model/loss.py:11-65): same constructor, same
forward(output, labels, weights) -> {"loss", "loss_cmd", "loss_args"[, "loss_visibility"][, "loss_kl"]}.

The three cross-entropies run in the masked-CE HIP kernels (no boolean-mask gathers, no host syncs, static
shapes -> hipGraph-capturable).  Mask semantics: the `extended` padding mask uses the non-aliased reading of
deepsvg/model/utils.py:25-28 (mask | mask shifted by 3), see DESIGN.md section 5, "loss_cmd mask".
"""
import os

import torch
import torch.nn as nn

from . import ops
from . import functional as Fn
from .svgtensor import CMD_ARGS_MASK, EOS_ID


class SVGLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.args_dim = 2 * cfg.args_dim if cfg.rel_targets else cfg.args_dim + 1
        self.register_buffer("cmd_args_mask", CMD_ARGS_MASK.clone())
        self._cam_f32 = None
        # data-parallel hook (deepsvg_amd/trainer.py): count_reducer(float32 [3] local counts of the visibility / command /
        # argument cross-entropies) -> global counts / world, ONE collective per step
        self.count_reducer = None
        # the command head's cross-entropy inside the argument head's autograd node (functional.ArgsHeadLossFn)
        self.joint_heads = os.environ.get("DSVG_JOINT_HEADS", "1") != "0"

    def _cam(self, device):
        if self._cam_f32 is None or self._cam_f32.device != device:
            self._cam_f32 = self.cmd_args_mask.to(device=device, dtype=torch.float32).contiguous()
        return self._cam_f32

    def forward(self, output, labels=None, weights=None):
        cfg = self.cfg
        loss = 0.0
        res = {}
        if cfg.use_vae:
            # KL term (loss.py:24-30): N x dim_z elementwise math on the fp32 mu/logsigma, left to torch
            mu, logsigma = output["mu"].float(), output["logsigma"].float()
            loss_kl = -0.5 * torch.mean(1 + logsigma - mu.pow(2) - torch.exp(logsigma))
            loss_kl = loss_kl.clamp(min=weights["kl_tolerance"])
            loss = loss + weights["loss_kl_weight"] * loss_kl
            res["loss_kl"] = loss_kl

        tgt_commands, tgt_args = output["tgt_commands"], output["tgt_args"]
        N, G, S1 = tgt_commands.shape
        n_args = tgt_args.shape[-1]
        S = S1 - 1
        # the model's forward may have prepared the targets and the list of tokens that carry argument loss
        # (SVGTransformer._plan): the argument head's backward then runs on those tokens only
        lp = output.get("_dsvg_live")
        if lp is not None and lp["tgt_commands"] is tgt_commands:
            # this loss ignores every position of an invisible target group (loss.py:36,51-54): the model's second
            # decoder stage may restrict its backward to the visible-first prefix (functional.LivePrefix)
            lp["live"].armed = True
        head = output.get("_dsvg_head")
        if head is not None and not (head["tgt_commands"] is tgt_commands and head["tgt_args"] is tgt_args
                                     and torch.is_grad_enabled()):
            head = None
        # the dense logit tensors may be LAZY entries of the model's result (only read when they are needed): a training
        # forward that ran the visible groups only hands over the command logits of those groups - every row with a loss
        # term is among them
        command_logits = head.get("cmd_logits") if head is not None else None
        if command_logits is None:
            command_logits = output["command_logits"]
        device = command_logits.device
        if head is not None:
            cmd_tgt, cmd_w, arg_tgt, arg_w, vis_tgt = head["targets"]
        else:
            tc = tgt_commands.to(device=device, dtype=torch.float32).contiguous().view(N * G, S1)
            ta = tgt_args.to(device=device, dtype=torch.float32).contiguous().view(N * G, S1, n_args)
            cmd_tgt, cmd_w, arg_tgt, arg_w, vis_tgt = ops.loss_targets(tc, ta, self._cam(device), EOS_ID)

        # data-parallel: the three means are taken over the GLOBAL selected-element counts.  The counts depend on the
        # targets only, so they are known before any logit is read: one 3-element all-reduce up front
        red = self.count_reducer
        cnt = None
        if red is not None:
            local = torch.stack([torch.full((), float(N * G), device=device),
                                 (cmd_w != 0).sum().float(), (arg_w != 0).sum().float()])
            cnt = red(local)
        scs, ws, names = [], [], []
        if cfg.decode_stages == 2:
            vl = output["visibility_logits"].reshape(N * G, 2)
            scs.append(Fn.MaskedCEFn.apply(vl, vis_tgt, None, 2, 1, (lambda c: cnt[0]) if red else None))
            ws.append(weights["loss_visibility_weight"])
            names.append("loss_visibility")

        cl = command_logits.reshape(-1, cfg.n_commands)      # (the head input's rows: all N G S, or the sequences that ran)
        if head is not None:
            # the targets' row order and the head input's row order must be the SAME one (both the stage's visible-first order,
            # or both the caller's group order) - the plan guarantees it, checked here on every call rather than assumed
            assert bool(head.get("vf")) == bool(head.get("x_vf")), \
                f"targets in {'visible-first' if head.get('vf') else 'caller'} order, head rows in " \
                f"{'visible-first' if head.get('x_vf') else 'caller'} order"
        if cl.shape[0] != cmd_tgt.numel():
            # logits of a row prefix (the visible sequences that ran): only meaningful in the visible-first order
            assert head is not None and head.get("vf") and head.get("x_vf"), \
                "command logits cover a row prefix but the targets are not in the stage's visible-first order"
        cmd_t, cmd_wt = cmd_tgt.view(-1)[:cl.shape[0]], cmd_w.view(-1)[:cl.shape[0]]
        joint = head is not None and head.get("cmd_weight") is not None and self.joint_heads
        if not joint:
            scs.append(Fn.MaskedCEFn.apply(cl, cmd_t, cmd_wt, cfg.n_commands, 1,
                                           (lambda c: cnt[1]) if red else None))
            ws.append(weights["loss_cmd_weight"])
            names.append("loss_cmd")
        if head is not None:
            # fused argument head + loss on the loss-carrying tokens (forward and backward); the dense args_logits of
            # the result dict stays unmaterialised
            lo, hi = head.get("slots", (0, n_args))
            tr = head.get("targets_r")
            if tr is not None:      # the slots [lo, hi) are the only ones that carry loss in this batch
                a_t, a_w, slots = tr[0], tr[1], (lo, hi)
            else:
                a_t, a_w, slots = arg_tgt.view(-1), arg_w.view(-1), (0, n_args)
            if joint:
                # ... together with the command head's cross-entropy: one backward node for both heads (their input's
                # gradient is written once: the command head's product, the argument head's rows added into it)
                sc_c, sc_a = Fn.ArgsHeadLossFn.apply(
                    head["rt"], head["x"], head["weight"], head["bias"], a_t, a_w, self.args_dim, slots[1] - slots[0],
                    (lambda c: cnt[2]) if red else None, head["live"], slots[0], cl.detach(), head["cmd_weight"],
                    head["cmd_bias"], cmd_t, cmd_wt, (lambda c: cnt[1]) if red else None)
                scs.append(sc_c)
                ws.append(weights["loss_cmd_weight"])
                names.append("loss_cmd")
            else:
                sc_a = Fn.ArgsHeadLossFn.apply(head["rt"], head["x"], head["weight"], head["bias"], a_t, a_w, self.args_dim,
                                               slots[1] - slots[0], (lambda c: cnt[2]) if red else None, head["live"],
                                               slots[0])
        else:
            al = output["args_logits"].reshape(N * G * S, n_args * self.args_dim)
            sc_a = Fn.MaskedCEFn.apply(al, arg_tgt.view(-1), arg_w.view(-1), self.args_dim, n_args,
                                       (lambda c: cnt[2]) if red else None)
        scs.append(sc_a)
        ws.append(weights["loss_args_weight"])
        names.append("loss_args")
        # the three means and their weighted sum (loss.py:43-57) in one launch (one more in the backward pass)
        total, *terms = Fn.LossCombineFn.apply(tuple(ws), *scs)
        res.update(dict(zip(names, terms)))
        res["loss"] = total if not cfg.use_vae else loss + total
        return res
