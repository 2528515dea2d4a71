"""Generates tests/golden/batch_assembly.npz by running the REAL reference batch assembly,
`deepsvg.svgtensor_dataset.SVGTensorDataset.get_data` (imported read-only from /root/reference), on seeded icons.
Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_batch.py

`deepsvg.svgtensor_dataset` imports the reference's drawing stack (cairosvg, IPython, moviepy, shapely ...), which
is not installed here; those modules are stubbed - get_data itself only touches torch and
deepsvg.difflib.tensor.SVGTensor.  Stored: the per-icon group tensors (flattened rows + lengths), fillings, and
every model_args key get_data produces for two (G, S, T) settings.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")


class _Stub:
    def __getattr__(self, k):
        return _Stub()

    def __call__(self, *a, **k):
        return _Stub()

    def __mro_entries__(self, bases):
        return (object,)


for _name in ["cairosvg", "IPython", "IPython.display", "moviepy", "moviepy.editor", "shapely", "shapely.geometry",
              "shapely.ops", "torchvision", "torchvision.utils", "torchvision.transforms",
              "torchvision.transforms.functional", "tensorboardX", "networkx", "PIL", "PIL.Image", "PIL.ImageOps",
              "matplotlib", "matplotlib.pyplot", "matplotlib.figure", "matplotlib.colors"]:
    try:
        __import__(_name)
    except Exception:
        _m = types.ModuleType(_name)
        _m.__file__ = "/dev/null"
        _m.__getattr__ = lambda k: _Stub()
        sys.modules[_name] = _m

from deepsvg.svgtensor_dataset import SVGTensorDataset          # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KEYS = ["commands", "args", "args_rel", "commands_grouped", "args_grouped", "args_rel_grouped", "filling"]
# per-command used argument columns of the 14-wide row: radius 1:3, x_axis_rot 3, large_arc 4, sweep 5,
# start_pos 6:8 (ignored by from_data), control1 8:10, control2 10:12, end_pos 12:14
USED = {0: [12, 13], 1: [12, 13], 2: [8, 9, 10, 11, 12, 13], 3: [1, 2, 3, 4, 5, 12, 13], 6: []}


def make_icon(rng, G, S, T, cmds_pool):
    """a list of <= G group tensors [len, 14] (float32, numericalised values 0..255, unused columns -1)"""
    n_groups = int(rng.integers(0, G + 1))
    lens, left = [], T
    for _ in range(n_groups):
        ln = int(rng.integers(0, min(S, left) + 1))     # empty groups are legal ([0, 14] tensors)
        lens.append(ln)
        left -= ln
    groups = []
    for ln in lens:
        t = np.full((ln, 14), -1.0, dtype=np.float32)
        for r in range(ln):
            c = 0 if r == 0 else int(rng.choice(cmds_pool))
            t[r, 0] = c
            for col in USED[c]:
                t[r, col] = float(rng.integers(0, 256))
            t[r, 6:8] = rng.integers(0, 256, size=2)    # start_pos is stored in the .pkl rows but never read
        groups.append(t)
    fill = [int(rng.integers(0, 3)) for _ in groups]
    return groups, fill


def run(tag, G, S, T, n_icons, seed, cmds_pool, rec):
    rng = np.random.default_rng(seed)

    class _Self:
        pass
    ds = _Self()
    ds.MAX_NUM_GROUPS, ds.MAX_SEQ_LEN, ds.PAD_VAL = G, S, -1
    ds.MAX_TOTAL_LEN = T if T is not None else G * S            # svgtensor_dataset.py:24-28
    ds.model_args = KEYS
    rows, lens, fills, n_groups = [], [], [], []
    outs = {k: [] for k in KEYS}
    for _ in range(n_icons):
        groups, fill = make_icon(rng, G, S, ds.MAX_TOTAL_LEN, cmds_pool)
        n_groups.append(len(groups))
        lens.extend(len(g) for g in groups)
        fills.extend(fill)
        rows.extend(groups)
        res = SVGTensorDataset.get_data(ds, [torch.from_numpy(g.copy()) for g in groups], list(fill),
                                        model_args=KEYS)
        for k in KEYS:
            outs[k].append(res[k].numpy())
    rec[f"{tag}/cfg"] = np.array([G, S, ds.MAX_TOTAL_LEN], dtype=np.int64)
    rec[f"{tag}/rows"] = np.concatenate(rows + [np.zeros((0, 14), np.float32)], axis=0).astype(np.int16)
    rec[f"{tag}/lens"] = np.array(lens, dtype=np.int32)
    rec[f"{tag}/n_groups"] = np.array(n_groups, dtype=np.int32)
    rec[f"{tag}/fills"] = np.array(fills, dtype=np.int32)
    for k in KEYS:
        a = np.stack(outs[k])
        assert np.array_equal(a, np.round(a)) and np.abs(a).max() < 32000
        rec[f"{tag}/{k}"] = a.astype(np.int16)


if __name__ == "__main__":
    rec = {}
    run("icons", 8, 30, 50, 24, 11, [1, 2], rec)                 # default_icons.py:40-41 (m / l / c only)
    run("arcs", 4, 12, None, 12, 12, [0, 1, 2, 3, 6], rec)        # a, z and mid-path m rows; T = G*S
    path = os.path.join(OUT, "batch_assembly.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path), "bytes")
