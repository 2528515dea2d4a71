"""Effective shader clock under the train step (round-4 review item 3).
  1. sclk samples (sysfs hwmon freq1_input / pp_dpm_sclk, whichever the box exposes; `rocm-smi --showclocks` once) taken by a
     background thread every 10 ms while the main thread runs: 2 s idle, the bench's 25 set-up + timed steps, then 10 s of the
     replayed train step; ms/step per window of 25 steps, so that a ramp (a short run timed at a lower clock than a long one)
     shows as a drift of the window means;
  2. the same for a loop of ffn_fwd alone (the kernel whose roofline fraction the bench reports).
GRBM_GUI_ACTIVE / kernel duration per kernel comes from scripts/gpu_clock.sh (rocprofv3 --pmc over this script's step loop)."""
import glob
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import deepsvg_amd  # noqa: E402
from deepsvg_amd import lib, ops  # noqa: E402
from deepsvg_amd.synthetic import make_batch, det_state_dict  # noqa: E402
from deepsvg_amd.trainer import TrainStep  # noqa: E402

SECONDS = float(os.environ.get("CLOCK_PROBE_SECONDS", "10"))


def sclk_sources():
    src = []
    for p in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
        src.append(("hwmon", p))
    for p in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        src.append(("dpm", p))
    return src


def read_sclk(kind, path):
    try:
        txt = open(path).read()
    except OSError:
        return None
    if kind == "hwmon":
        return float(txt) / 1e6
    for line in txt.splitlines():
        if line.strip().endswith("*"):
            return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
    return None


class Sampler(threading.Thread):
    def __init__(self, src):
        super().__init__(daemon=True)
        self.src, self.samples, self.stop, self.mark = src, [], False, "idle"

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            for kind, path in self.src:
                v = read_sclk(kind, path)
                if v is not None:
                    self.samples.append((t, self.mark, kind, v))
            time.sleep(0.01)


def summarize(samples, mark):
    out = {}
    for kind in sorted({s[2] for s in samples}):
        v = sorted(s[3] for s in samples if s[1] == mark and s[2] == kind)
        if v:
            out[kind] = f"n={len(v)} min {v[0]:.0f} median {v[len(v) // 2]:.0f} max {v[-1]:.0f} MHz"
    return out


def main():
    dev = torch.device("cuda", 0)
    lib.load()
    src = sclk_sources()
    print("sclk sources:", src)
    try:
        print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=30).stdout[-1500:])
    except Exception as e:   # noqa: BLE001
        print("rocm-smi unavailable:", e)
    smp = Sampler(src)
    smp.start()
    torch.manual_seed(42)
    cfg = deepsvg_amd.HierarchicalOrdered()
    cfg.dropout = 0.1
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(det_state_dict(model, seed=42))
    model.to(dev).set_compute_dtype(torch.bfloat16)
    model.train()
    loss_fn = deepsvg_amd.SVGLoss(cfg).to(dev)
    batches = []
    for k in range(8):
        c, a = make_batch(512, G=8, S=30, seed=1000 + 97 * k)
        batches.append((c.to(dev), a.to(dev)))
    ts = TrainStep(model, loss_fn, lr=1e-3, grad_clip=1.0, use_graph=True)
    ts.inputs_resident = True
    for k in range(8):
        ts.step(*batches[k])
    torch.cuda.synchronize()
    time.sleep(2.0)
    print("idle:", summarize(smp.samples, "idle"))

    def window(n, tag):
        smp.mark = tag
        t0 = time.perf_counter()
        for i in range(n):
            ts.step(*batches[i % 8])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    # what bench.py times: 5 warm-up + 20 steps straight out of an idle GPU
    w = window(5, "bench_warmup")
    b = window(20, "bench_timed")
    print(f"from idle: 5 warm-up steps {w:.3f} ms/step, then 20 timed steps {b:.3f} ms/step",
          summarize(smp.samples, "bench_timed"))
    t_end = time.perf_counter() + SECONDS
    k = 0
    while time.perf_counter() < t_end:
        ms = window(25, f"w{k}")
        if k < 12 or k % 10 == 0:
            print(f"window {k:3d} (t = {time.perf_counter() - (t_end - SECONDS):5.2f} s): {ms:.3f} ms/step", summarize(smp.samples, f"w{k}"))
        k += 1
    print(f"last window {k - 1}: {ms:.3f} ms/step", summarize(smp.samples, f"w{k - 1}"))

    # ffn_fwd alone (training variant, 63,488 rows, p = 0.1), back to back
    g = torch.Generator(device="cpu").manual_seed(0)
    flat = torch.zeros(8 + 131072 + 512 + 131072 + 256 + 256 + 8)
    o = 8
    offs = [[o, o + 131072, o + 131072 + 512, o + 262144 + 512, o + 262144 + 768]]
    flat[o:o + 131072] = torch.randn(131072, generator=g) * 0.06
    flat[o + 131072 + 512:o + 262144 + 512] = torch.randn(131072, generator=g) * 0.06
    flat[o + 262144 + 512:o + 262144 + 768] = 1.0
    flat = flat.to(dev)
    pf, pb, b1f = ops.ffn_pack(flat, torch.tensor(offs, dtype=torch.int64, device=dev), 1)
    pl = pf[:ops.FFN_FWD_LAYER_ELEMS]
    b2 = torch.zeros(256, device=dev)
    seed = torch.tensor([1234567], dtype=torch.int64, device=dev)
    x = torch.randn(63488, 256, generator=g).to(dev).to(torch.bfloat16)
    time.sleep(1.0)
    for rep in range(6):
        smp.mark = f"ffn{rep}"
        n = 50 if rep == 0 else 2000
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.ffn_fwd(x, pl, b1f[0], b2, 1e-5, 0.1, 3, 4, seed, train=True)
        e1.record()
        torch.cuda.synchronize()
        print(f"ffn_fwd train 63,488 rows x {n}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch", summarize(smp.samples, f"ffn{rep}"))
    smp.stop = True


if __name__ == "__main__":
    main()
