"""BASELINE.json configs[0] exactly as it is stated: configs/deepsvg/hierarchical_ordered.py at batch 2 (G = 8, S = 30, d_model
256) through the UNMODIFIED deepsvg/train.py on the host's cores - the reference's own model, loss, AdamW, grad-clip, its
DataLoader on a synthetic on-disk dataset in its own format, and its own `time` statistic.
Runs only where /root/reference is mounted (the build container; the GPU boxes do not have it - there bench.py times the
line-by-line restatement under oracle/ instead and says so: cpu_baseline.kind = "port").
usage: python scripts/cpu_c1_reference_train.py [steps=60] [threads=all]   -> one JSON line (and the reference's log lines)"""
import contextlib
import io
import json
import os
import pickle
import re
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"


class _Stub:
    def __getattr__(self, k):
        return _Stub()

    def __call__(self, *a, **k):
        return _Stub()

    def __mro_entries__(self, bases):
        return (object,)


def stub_missing_modules():
    """the reference imports its drawing / logging stack at module level (cairosvg, tensorboardX, ...)"""
    for name in ["cairosvg", "IPython", "IPython.display", "moviepy", "moviepy.editor", "shapely", "shapely.geometry",
                 "shapely.ops", "torchvision", "torchvision.utils", "torchvision.transforms",
                 "torchvision.transforms.functional", "tensorboardX", "networkx", "PIL", "PIL.Image", "PIL.ImageOps",
                 "matplotlib", "matplotlib.pyplot", "matplotlib.figure", "matplotlib.colors"]:
        try:
            __import__(name)
        except Exception:
            m = types.ModuleType(name)
            m.__file__ = "/dev/null"
            m.__getattr__ = lambda k: _Stub()
            sys.modules[name] = m


def write_dataset(root, n_icons, G=8, S=30, seed=0):
    """<id>.pkl + meta CSV in the reference's format (svgtensor_dataset.py:33-52,106-109): 1..G groups of 2..S commands (m,
    then l / c), arguments uniform in 0..255, redrawn until the icon passes the config's own filters (at most 50 commands in
    total, configs/deepsvg/default_icons.py:40-41; svgtensor_dataset.py:41-43).  The model's work does not depend on the icon:
    every batch is padded to (2, 8, 32) commands"""
    import pandas as pd
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n_icons):
        while True:
            ng = int(rng.integers(1, G + 1))
            lens = [int(rng.integers(2, S + 1)) for _ in range(ng)]
            if sum(lens) <= 50:
                break
        groups = []
        for ln in lens:
            t = np.full((ln, 14), -1.0, np.float32)
            t[:, 0] = rng.integers(1, 3, size=ln)
            t[0, 0] = 0
            for r in range(ln):
                cols = [12, 13] if t[r, 0] < 2 else [8, 9, 10, 11, 12, 13]
                t[r, cols] = rng.integers(0, 256, size=len(cols))
            groups.append(torch.from_numpy(t))
        with open(os.path.join(root, f"{i}.pkl"), "wb") as f:
            pickle.dump({"tensors": [groups], "fillings": [0] * ng}, f)
        rows.append(dict(id=i, nb_groups=ng, max_len_group=max(lens), total_len=sum(lens), category="arrows"))
    meta = os.path.join(root, "meta.csv")
    pd.DataFrame(rows).to_csv(meta, index=False)
    return meta


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    stub_missing_modules()
    sys.path.insert(0, REF)
    from configs.deepsvg import hierarchical_ordered as ref_config
    from deepsvg import train as ref_train
    with tempfile.TemporaryDirectory() as tmp:
        meta = write_dataset(tmp, n_icons=2 * (steps + 4))

        class Config(ref_config.Config):
            def __init__(self):
                super().__init__(num_gpus=1)
                self.data_dir, self.meta_filepath = tmp, meta
                self.batch_size, self.loader_num_workers = 2, 0        # BASELINE configs[0]: batch = 2, reference plumbing
                self.num_epochs, self.num_steps = 1, steps
                self.log_every = 10
                self.val_every = self.ckpt_every = 10 ** 9
                self.device = "cpu"

            def set_train_vars(self, train_vars, dataloader):          # the drawing hook needs the real svglib / cairosvg
                pass

        cfg = Config()
        times = []
        real_update = None
        # the reference's own per-step wall time (`time` statistic of train.py:118-121), collected as it is produced
        from deepsvg.utils import stats as ref_stats
        real_update = ref_stats.Stats.update

        def spy(self, split, step, epoch, dic):
            if "time" in dic:
                times.append(float(dic["time"]))
            return real_update(self, split, step, epoch, dic)
        ref_stats.Stats.update = spy
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                ref_train.train(cfg, "deepsvg", "c1_cpu_baseline", log_dir=os.path.join(tmp, "logs"), debug=True)
        finally:
            ref_stats.Stats.update = real_update
        log = buf.getvalue()
    for line in log.splitlines():
        if re.match(r"^\[\d+/", line) or line.startswith("#Parameters"):
            print(line)
    warm = 5
    t = np.array(times[warm:])
    print(json.dumps({
        "what": "unmodified /root/reference deepsvg/train.py, configs/deepsvg/hierarchical_ordered.py, batch 2, CPU, fp32, dropout 0.1",
        "steps_timed": int(t.size), "warmup_steps_dropped": warm, "threads": threads, "host_cores": os.cpu_count(),
        "s_per_step_median": round(float(np.median(t)), 4), "s_per_step_mean": round(float(t.mean()), 4),
        "icons_per_s": round(2.0 / float(np.median(t)), 2), "torch": torch.__version__,
    }))


if __name__ == "__main__":
    main()
