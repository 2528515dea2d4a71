"""Secondary measurements next to the headline train step ("Secondary: C4 training icons/s; C5 decoded
icons/s"), bf16, synthetic inputs, one MI355X:
  C4  one-stage one-shot model (OneStageOneShot, max_total_len = 50): full train step, 512 icons
  C5a decode-only, one-shot: hierarchical_ordered greedy_sample from latents, 8192 icons
  C5b decode-only, autoregressive command sampling: Sketchformer (max_total_len = 50), 8192 icons, incremental decoding
      over the per-layer q|k|v cache; the reference's scheme (decoder re-run on the growing prefix) beside it at 1024 icons
Prints one line per item."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepsvg_amd                                  # noqa: E402
from deepsvg_amd import config as C                 # noqa: E402
from deepsvg_amd.synthetic import make_batch, make_batch_onestage, det_state_dict      # noqa: E402
from deepsvg_amd.trainer import TrainStep           # noqa: E402

DEV = "cuda"


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps


def model_for(cfg, seed=1):
    m = deepsvg_amd.SVGTransformer(cfg)
    m.load_state_dict(det_state_dict(m, seed=seed))
    return m.to(DEV).set_compute_dtype(torch.bfloat16)


def leg_c4p():
    """C4 at the reference's default length"""
    cfg = C.OneStageOneShot()
    cfg.use_vae = False                         # max_total_len = 240: 242-token encoder / 241-token decoder sequences
    model = model_for(cfg).train()
    commands, args = make_batch_onestage(256, total_len=240, seed=1)
    commands, args = commands.to(DEV), args.to(DEV)
    step = TrainStep(model, deepsvg_amd.SVGLoss(cfg).to(DEV), lr=1e-3, use_graph=True)
    for _ in range(3):
        step.step(commands, args)
    sec = timed(lambda: step.step(commands, args), 10)
    print(f"C4' one-stage train step at max_total_len = 240 (256 icons x 242 tokens, bf16, hipGraph, streaming attention): "
          f"{sec * 1e3:.2f} ms/step, {256 / sec:,.0f} icons/s, {256 * 242 / sec:,.0f} tokens/s")


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None          # "c4p": the 242-token training leg alone (A/B of knobs)
    if only == "c4p":
        return leg_c4p()
    # ---- C4 --------------------------------------------------------------------------------------------------
    cfg = C.OneStageOneShot()
    cfg.max_total_len = 50
    cfg.use_vae = False
    model = model_for(cfg).train()
    commands, args = make_batch_onestage(512, total_len=50, seed=1)
    commands, args = commands.to(DEV), args.to(DEV)
    step = TrainStep(model, deepsvg_amd.SVGLoss(cfg).to(DEV), lr=1e-3, use_graph=True)
    for _ in range(4):
        step.step(commands, args)
    sec = timed(lambda: step.step(commands, args), 20)
    print(f"C4  one-stage train step (512 icons x 52 tokens, bf16, hipGraph): {sec * 1e3:.2f} ms/step, "
          f"{512 / sec:,.0f} icons/s")
    del step, model

    leg_c4p()

    # ---- C5a -------------------------------------------------------------------------------------------------
    cfg = C.HierarchicalOrdered()
    model = model_for(cfg).eval()
    N = 8192
    z = torch.randn(N, 1, 1, cfg.dim_z, device=DEV)

    def one_shot():
        out = []
        for i in range(0, N, 1024):       # bounded logits footprint: 1024 icons = 2.9 GB of fp32 argument logits
            out.append(model.greedy_sample(z=z[i:i + 1024], concat_groups=False))
        return out
    sec = timed(one_shot, 2)
    print(f"C5a one-shot decode from latents (8192 icons, G=8, S=30, bf16): {sec * 1e3:.0f} ms, {N / sec:,.0f} icons/s")

    def one_shot_argmax():
        return [model.greedy_sample(z=z[i:i + 1024], concat_groups=False, temperature=0) for i in range(0, N, 1024)]
    sec = timed(one_shot_argmax, 2)
    print(f"C5a' same with temperature = 0 (argument head + arg-max in one kernel, the logits never stored): "
          f"{sec * 1e3:.0f} ms, {N / sec:,.0f} icons/s")
    def one_shot_argmax_unfused():      # round 2's temperature-0 path: dense bf16 argument logits + an arg-max pass over them
        out = []
        with torch.no_grad():
            for i in range(0, N, 1024):
                res = model.forward(None, None, None, None, z=z[i:i + 1024], return_tgt=False)
                out.append(model._sample(res["command_logits"], res["args_logits"], 0))
        return out
    sec_u = timed(one_shot_argmax_unfused, 2)
    print(f"C5a'' forward + dense bf16 argument logits + arg-max kernel over them (round 2's temperature-0 path, without the "
          f"validity fix-ups of greedy_sample): {sec_u * 1e3:.0f} ms")
    del model

    # ---- C5b -------------------------------------------------------------------------------------------------
    cfg = C.Sketchformer()
    cfg.max_total_len = 50
    cfg.use_vae = False
    model = model_for(cfg).eval()
    z = torch.randn(N, 1, 1, cfg.dim_z, device=DEV)
    model.kv_cache = True
    sec = timed(lambda: model.greedy_sample(z=z, concat_groups=False), 2)
    print(f"C5b autoregressive sampling, q|k|v cache (8192 icons x 50 tokens, bf16): {sec * 1e3:.0f} ms, "
          f"{N / sec:,.0f} icons/s, {N * 50 / sec:,.0f} tokens/s")
    sec = timed(lambda: model.greedy_sample(z=z, concat_groups=False, temperature=0), 2)
    print(f"C5b'' cache + temperature = 0: {sec * 1e3:.0f} ms, {N / sec:,.0f} icons/s, {N * 50 / sec:,.0f} tokens/s")
    model.kv_cache = False
    n2 = 1024
    sec2 = timed(lambda: model.greedy_sample(z=z[:n2], concat_groups=False), 1)
    print(f"C5b' same, decoder re-run on the growing prefix as the reference does (1024 icons): {sec2 * 1e3:.0f} ms, "
          f"{n2 / sec2:,.0f} icons/s")


if __name__ == "__main__":
    main()
