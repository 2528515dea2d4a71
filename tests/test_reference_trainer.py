"""The drop-in claim of INTEGRATION.md, exercised end to end: the reference's UNMODIFIED training loop
(`deepsvg/train.py`, imported from /root/reference) drives `deepsvg_amd.SVGTransformer` / `deepsvg_amd.SVGLoss` through
`cfg.make_model()` / `cfg.make_losses()` - first on the reference's own dataset class, then on the packed store +
`device_collate` of `deepsvg_amd.dataset` through the `cfg.dataloader_module` / `cfg.collate_fn` seam.

CPU only, with the ops emulated in plain torch (tests/torch_ops_ref.py): this checks the host surface (constructor,
forward signature, result dict, parameters / optimizer / clip_grad_norm_ / DataParallel-free path, loss dict), not the
kernels.  Skipped where the reference is not mounted (the GPU boxes)."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "deepsvg")), reason="reference not mounted")


class _Stub:
    def __getattr__(self, k):
        return _Stub()

    def __call__(self, *a, **k):
        return _Stub()

    def __mro_entries__(self, bases):
        return (object,)


@pytest.fixture
def reference_on_path():
    """the reference imports its drawing / logging stack at module level (cairosvg, tensorboardX, ...): stubbed"""
    added = []
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
            added.append(name)
    sys.path.insert(0, REF)
    yield
    sys.path.remove(REF)
    for name in added:
        sys.modules.pop(name, None)


def _write_dataset(root, n_icons=24, n_var=2, G=8, S=30, T=50, seed=0):
    """the reference's on-disk format: <id>.pkl + meta CSV (svgtensor_dataset.py:33-52,106-109)"""
    import pandas as pd
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n_icons):
        ng = int(rng.integers(1, 5))
        lens = [int(rng.integers(2, 9)) for _ in range(ng)]
        variants = []
        for _v in range(n_var):
            groups = []
            for ln in lens:
                t = np.full((ln, 14), -1.0, np.float32)
                t[:, 0] = rng.integers(1, 3, size=ln)
                t[0, 0] = 0
                for r in range(ln):
                    cols = [12, 13] if t[r, 0] < 2 else [8, 9, 10, 11, 12, 13]
                    t[r, cols] = rng.integers(0, 256, size=len(cols))
                groups.append(torch.from_numpy(t))
            variants.append(groups)
        with open(os.path.join(root, f"{i}.pkl"), "wb") as f:
            pickle.dump({"tensors": variants, "fillings": [0] * ng}, f)
        rows.append(dict(id=i, nb_groups=ng, max_len_group=max(lens), total_len=sum(lens), category="arrows"))
    meta = os.path.join(root, "meta.csv")
    pd.DataFrame(rows).to_csv(meta, index=False)
    return meta


@pytest.mark.parametrize("data_path", ["reference_dataset", "packed_store"])
def test_reference_train_loop_drives_the_drop_in(tmp_path, emulated_ops, reference_on_path, data_path):
    import deepsvg_amd
    import deepsvg_amd.dataset
    from configs.deepsvg import hierarchical_ordered as ref_config          # the reference's north-star config
    from deepsvg import train as ref_train

    meta = _write_dataset(str(tmp_path))
    holder = {}

    class Config(ref_config.Config):
        def __init__(self):
            super().__init__(num_gpus=1)
            self.data_dir, self.meta_filepath = str(tmp_path), meta
            self.batch_size, self.loader_num_workers = 4, 0
            self.num_epochs, self.num_steps = 1, 3
            self.log_every = self.val_every = self.ckpt_every = 10 ** 9
            self.model_cfg.n_layers = self.model_cfg.n_layers_decode = 1       # keep the CPU emulation quick
            self.device = "cpu"
            if data_path == "packed_store":
                self.dataloader_module = "deepsvg_amd.dataset"
                self.collate_fn = deepsvg_amd.dataset.device_collate

        def make_model(self):
            holder["model"] = deepsvg_amd.SVGTransformer(self.model_cfg)
            holder["before"] = {n: p.detach().clone() for n, p in holder["model"].named_parameters()}
            return holder["model"]

        def make_losses(self):
            return [deepsvg_amd.SVGLoss(self.model_cfg)]

        def set_train_vars(self, train_vars, dataloader):                     # the drawing hook needs the real svglib
            pass

    ref_train.train(Config(), "deepsvg", "drop_in_test", log_dir=str(tmp_path / "logs"), debug=True)
    model = holder["model"]
    moved = 0
    for n, p in model.named_parameters():
        assert torch.isfinite(p).all(), n
        moved += int(not torch.equal(p.detach(), holder["before"][n]))
    assert moved > 0.9 * len(holder["before"]), f"only {moved} of {len(holder['before'])} parameters were updated"
