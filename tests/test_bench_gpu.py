"""bench.py honours the driver's contract: one JSON line with the required keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract(gpu_device):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "32",
           "--cpu-batch", "2", "--batches", "3"]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["unit"] == "icons/s"
    assert rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert rec["data"] == "synthetic" and rec["dtype"] == "bf16" and "workload" in rec["config"]
    assert abs(rec["value"] - 32 / (rec["ms_per_step"] * 1e-3)) < 0.01 * rec["value"]
    r = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "hbm_view", "fused_fwd_kernel"):
        assert k in r, k
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or str(r["traffic_source"]).startswith("committed")
    if r["traffic"] is not None:
        # the committed PMC measurement must have been taken on THIS kernel source (scripts/gpu_ffn_traffic.sh re-collects it)
        assert r["traffic_collected_at"]["kernel_source_unchanged"], r["traffic_collected_at"]
    for k in ("dense_layout", "secondary", "in_kernel_clock"):
        assert k in rec and rec[k] is not None and "error" not in rec[k], (k, rec.get(k))
    assert rec["dense_layout"]["ms_per_step"] > 0
    assert rec["secondary"]["c4_one_stage_train"]["ms_per_step"] > 0 and rec["secondary"]["c5_one_shot_decode"]["ms"] > 0
    assert rec["secondary"]["c5_autoregressive_decode"]["ms"] > 0
    ck = rec["in_kernel_clock"]["shader_clock_mhz"]
    assert 200 < ck["chunk_loop"] < 3500 and 200 < ck["whole_wave"] < 3500, ck      # (a sanity range, not a power-state claim)
    assert rec["config"]["clock_mhz"]["in_kernel"] == ck
    if r["fused_fwd_kernel"] is not None:       # (32 icons: the stages are below the fused kernels' row threshold)
        assert r["fused_fwd_kernel"]["frac_at_measured_clock"] > 0
    assert rec["fp32"] is not None and rec["fp32"]["ms_per_step"] > 0
    c = rec["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and "by_threads" in c and c["dropout"] == 0.1
    assert c["c1_batch2"]["value"] > 0
    g = rec["graphs"]
    assert g["batches_rotated"] == 3 and g["graphs_captured_inside_timed_region"] == 0 and rec["rccl_ranks"] == 1
    t = rec["torch_rocm_reference"]
    assert t["ms_per_step"] > 0 and t["dtype"] == "fp32" and t["batch"] == 32
    assert "timed_in" in r


def test_secondary_workloads_perf_guard(gpu_device):
    """BASELINE configs C4 (one-stage, 512 icons x 52 tokens, train step) and C5 (one-shot decode of 8192 latents): prints
    ms per step / per call (min of 3 repeats of 10) so that the driver's log carries them (bench.py's `secondary` object carries
    them into the driver-parsed record), and guards against a regression: 1.5 x the round-3 / round-4 measurements (C4 5.41 ms,
    C5 31-34 ms; the boxes of the pool differ by +-3 %)"""
    import time
    import torch
    import deepsvg_amd
    from deepsvg_amd import config as C
    from deepsvg_amd.synthetic import make_batch_onestage, det_state_dict
    from deepsvg_amd.trainer import TrainStep

    def best_of(fn, reps=10, rounds=3):
        fn()
        out = []
        for _ in range(rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / reps)
        return min(out), out

    cfg = C.OneStageOneShot()
    cfg.max_total_len = 50
    cfg.use_vae = False
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(det_state_dict(model, seed=1))
    model.to("cuda").set_compute_dtype(torch.bfloat16).train()
    commands, args = make_batch_onestage(512, total_len=50, seed=1)
    commands, args = commands.to("cuda"), args.to("cuda")
    step = TrainStep(model, deepsvg_amd.SVGLoss(cfg).to("cuda"), lr=1e-3, use_graph=True)
    for _ in range(4):
        step.step(commands, args)
    c4, c4_all = best_of(lambda: step.step(commands, args))
    print(f"C4 one-stage train step (512 icons x 52 tokens, bf16, hipGraph): {c4 * 1e3:.2f} ms/step "
          f"(rounds: {', '.join(f'{t * 1e3:.2f}' for t in c4_all)}), {512 / c4:,.0f} icons/s")
    del step, model

    cfg = deepsvg_amd.HierarchicalOrdered()
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(det_state_dict(model, seed=1))
    model.to("cuda").set_compute_dtype(torch.bfloat16).eval()
    z = (torch.randn(8192, 1, 1, cfg.dim_z, generator=torch.Generator().manual_seed(0)) * 0.3).to("cuda")
    c5, c5_all = best_of(lambda: model.greedy_sample(z=z, concat_groups=False, temperature=0), reps=3)
    print(f"C5 one-shot decode of 8192 latents (hierarchical_ordered, arg-max): {c5 * 1e3:.1f} ms "
          f"(rounds: {', '.join(f'{t * 1e3:.1f}' for t in c5_all)}), {8192 / c5:,.0f} icons/s")
    # hard limits with head room for shared / noisy boxes (round 4 read 5.2-5.4 ms and 25 ms; a 9.3 ms C4 reading with no code
    # change is on record); DSVG_STRICT_PERF=1 asserts the measured level itself
    import os
    lim4, lim5 = (8.2e-3, 50e-3) if os.environ.get("DSVG_STRICT_PERF") == "1" else (16e-3, 100e-3)
    assert c4 < lim4 and c5 < lim5, (c4, c5)
