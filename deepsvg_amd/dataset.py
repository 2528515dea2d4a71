"""Device-side batch assembly: the data format next to the hot path.

Drop-in for `deepsvg.svgtensor_dataset` at the seam `deepsvg/train.py:25-28` uses
(`importlib.import_module(cfg.dataloader_module).load_dataset(cfg)` + `DataLoader(..., collate_fn=cfg.collate_fn)`):

    cfg.dataloader_module  = "deepsvg_amd.dataset"
    cfg.collate_fn         = deepsvg_amd.dataset.device_collate
    cfg.loader_num_workers = 0

The reference assembles every item on a CPU core with a chain of torch.cat calls
(SVGTensorDataset.get_data, svgtensor_dataset.py:164-205: 2.8 ms per icon, SURVEY.md probe A.6) and the DataLoader
stacks them.  Here the `.pkl` tensors are packed ONCE into flat int16 rows + offsets (PackedSVGStore), the store
lives in HBM, and one HIP kernel (dsvg_assemble_batch) writes the whole batch - SOS / rows / EOS / padding,
`args` and `args_rel`, per-group and grouped layouts - straight into the tensors the model consumes.

Same names as the reference where the surface is kept: `SVGTensorDataset(data_dir, meta_filepath, model_args,
max_num_groups, max_seq_len, max_total_len, filter_uni, filter_platform, filter_category, train_ratio, df, PAD_VAL)`
(svgtensor_dataset.py:18-19), `get`, `get_data`-equivalent keys ("commands", "args", "args_rel", their
"_grouped" forms, "filling", "label"), `__len__ = len(df) * nb_augmentations` (:111-112), `load_dataset(cfg)`
(:230-233).  Not carried over: the `svg=` path of `get` and the "tensor" keys (they return drawing objects of the
reference's svglib, a visualisation path outside the hot path).
"""
import os
import pickle
from collections import namedtuple

import numpy as np
import torch
import torch.utils.data

from . import ops
from .lib import DsvgError

# columns of a stored 14-wide row that SVGTensor.from_data keeps (deepsvg/difflib/tensor.py:23-32,85-88): the command
# and the 11 arg_keys columns (radius, x_axis_rot, large_arc_flg, sweep_flg, control1, control2, end_pos); START_POS
# (6:8) is derived data the tensors never read
ROW_COLS = (0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13)
N_ARGS = 11

# svgtensor_dataset.py:77-85
_CATEGORIES = ['characters', 'free-icons', 'logos', 'alphabet', 'animals', 'arrows', 'astrology', 'baby', 'beauty',
               'business', 'cinema', 'city', 'clothing', 'computer-hardware', 'crime', 'cultures', 'data', 'diy',
               'drinks', 'ecommerce', 'editing', 'files', 'finance', 'folders', 'food', 'gaming', 'hands',
               'healthcare', 'holidays', 'household', 'industry', 'maps', 'media-controls', 'messaging', 'military',
               'mobile', 'music', 'nature', 'network', 'photo-video', 'plants', 'printing', 'profile', 'programming',
               'science', 'security', 'shopping', 'social-networks', 'sports', 'time-and-date', 'transport', 'travel',
               'user-interface', 'users', 'weather', 'flags', 'emoji', 'men', 'women']


def _uni_to_label(uni):
    """svgtensor_dataset.py:60-66"""
    if 48 <= uni <= 57:
        return uni - 48
    if 65 <= uni <= 90:
        return uni - 65 + 10
    return uni - 97 + 36


def _as_rows(t):
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    a = a.reshape(-1, 14)
    r = np.rint(a)
    if a.size and (not np.array_equal(r, a) or r.min() < -32768 or r.max() > 32767):
        raise DsvgError("PackedSVGStore: the packed store holds numericalised (integer) tensors only")
    if a.size and (r[:, 0].min() < 0 or r[:, 0].max() > 6):
        raise DsvgError("PackedSVGStore: command index outside 0..6")
    return r[:, ROW_COLS].astype(np.int16)


class PackedSVGStore:
    """All icons of a dataset as flat arrays (host numpy or device torch):
      rows       int16 [R, 12]           (command, 11 arguments) per stored drawing command
      slot_off   int32 [n_variants*G+1]  row range of slot variant*G + group
      var_base   int32 [n_icons+1]       variants (stored augmentations, the list entries of the .pkl "tensors",
                                         svgtensor_dataset.py:106-109,156) of icon i are var_base[i] .. var_base[i+1]-1
      filling    int64 [n_icons, G]      per-group filling, 0 for missing groups (svgtensor_dataset.py:170-173)
      label      int64 [n_icons] or None
    """

    def __init__(self, rows, slot_off, var_base, filling, label, G, ids=None, max_group_len=0, max_total_len=0):
        self.rows, self.slot_off, self.var_base, self.filling, self.label = rows, slot_off, var_base, filling, label
        self.G = int(G)
        self.ids = list(ids) if ids is not None else None
        self.max_group_len, self.max_total_len = int(max_group_len), int(max_total_len)

    # ---- construction -------------------------------------------------------------------------------------------
    @classmethod
    def from_icons(cls, icons, fillings=None, labels=None, max_num_groups=8, ids=None):
        """icons[i][v] = list of <= max_num_groups group tensors [len, 14] (variant v of icon i);
        fillings[i] = per-group filling list of icon i (shared by its variants, as in the .pkl files)"""
        G = int(max_num_groups)
        rows, lens, var_base = [], [], [0]
        max_len = max_tot = 0
        for i, variants in enumerate(icons):
            if len(variants) == 0:
                raise DsvgError(f"PackedSVGStore: icon {i} has no tensors")
            for groups in variants:
                if len(groups) > G:
                    raise DsvgError(f"PackedSVGStore: icon {i} has {len(groups)} groups > max_num_groups={G}")
                g_rows = [_as_rows(g) for g in groups]
                g_lens = [r.shape[0] for r in g_rows] + [0] * (G - len(g_rows))
                rows.extend(g_rows)
                lens.extend(g_lens)
                max_len = max(max_len, max(g_lens))
                max_tot = max(max_tot, sum(g_lens))
            var_base.append(var_base[-1] + len(variants))
        n_icons = len(icons)
        fill = np.zeros((n_icons, G), dtype=np.int64)
        if fillings is not None:
            for i, f in enumerate(fillings):
                f = list(f)[:G]
                fill[i, :len(f)] = f
        rows = np.concatenate(rows + [np.zeros((0, 12), np.int16)], axis=0)
        if rows.shape[0] == 0:
            rows = np.zeros((1, 12), np.int16)      # keep the device array non-empty (never read: every len is 0)
        slot_off = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(np.asarray(lens, dtype=np.int64), out=slot_off[1:])
        if slot_off[-1] >= 2 ** 31:
            raise DsvgError("PackedSVGStore: more than 2^31 rows")
        label = None if labels is None else np.asarray(labels, dtype=np.int64).reshape(n_icons)
        return cls(np.ascontiguousarray(rows), slot_off.astype(np.int32), np.asarray(var_base, dtype=np.int32), fill,
                   label, G, ids, max_len, max_tot)

    @classmethod
    def from_pkl_dir(cls, data_dir, df, max_num_groups):
        """the reference's on-disk format: one `<id>.pkl` per row of the meta table with {"tensors": [variant ->
        [group tensors]], "fillings": [...]} (svgtensor_dataset.py:106-109); labels as get_label (:87-98)"""
        icons, fillings, labels, ids = [], [], [], []
        for entry in df.itertuples():
            with open(os.path.join(data_dir, f"{entry.id}.pkl"), "rb") as f:
                data = pickle.load(f)
            icons.append(data["tensors"])
            fillings.append(data["fillings"])
            ids.append(entry.id)
            if "uni" in df.columns:
                labels.append(_uni_to_label(int(entry.uni)))
            elif "category" in df.columns:
                labels.append(_CATEGORIES.index(entry.category))
        return cls.from_icons(icons, fillings, labels if labels else None, max_num_groups, ids)

    # ---- persistence / placement --------------------------------------------------------------------------------
    def save(self, path):
        np.savez_compressed(path, rows=self._np(self.rows), slot_off=self._np(self.slot_off),
                            var_base=self._np(self.var_base), filling=self._np(self.filling),
                            label=self._np(self.label) if self.label is not None else np.zeros(0, np.int64),
                            has_label=np.array(self.label is not None), G=np.array(self.G),
                            ids=np.array([str(i) for i in self.ids]) if self.ids is not None else np.zeros(0, "U1"),
                            lims=np.array([self.max_group_len, self.max_total_len]))

    @classmethod
    def load(cls, path):
        z = np.load(path)
        return cls(z["rows"], z["slot_off"], z["var_base"], z["filling"], z["label"] if bool(z["has_label"]) else None,
                   int(z["G"]), list(z["ids"]) if z["ids"].size else None, *[int(v) for v in z["lims"]])

    @staticmethod
    def _np(a):
        return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a

    @property
    def n_icons(self):
        return int(self.var_base.shape[0]) - 1

    @property
    def is_device(self):
        return isinstance(self.rows, torch.Tensor)

    def to(self, device):
        device = torch.device(device)
        ops.require_device(device)

        def mv(a):
            return None if a is None else torch.as_tensor(self._np(a)).to(device).contiguous()
        return PackedSVGStore(mv(self.rows), mv(self.slot_off), mv(self.var_base), mv(self.filling), mv(self.label),
                              self.G, self.ids, self.max_group_len, self.max_total_len)

    def n_variants_of(self, icon):
        vb = self._np(self.var_base)
        return int(vb[icon + 1] - vb[icon])


_Deferred = namedtuple("_Deferred", ["dataset", "icon"])      # what __getitem__ hands to device_collate


class SVGTensorDataset(torch.utils.data.Dataset):
    """Same constructor and meaning as deepsvg.svgtensor_dataset.SVGTensorDataset (svgtensor_dataset.py:17-52), plus
    `store=` (a ready PackedSVGStore instead of data_dir / meta_filepath) and `device=`."""

    def __init__(self, data_dir=None, meta_filepath=None, model_args=None, max_num_groups=8, max_seq_len=30,
                 max_total_len=None, filter_uni=None, filter_platform=None, filter_category=None, train_ratio=1.0,
                 df=None, PAD_VAL=-1, store=None, device="cuda", deferred=False):
        self.data_dir = data_dir
        self.MAX_NUM_GROUPS = max_num_groups
        self.MAX_SEQ_LEN = max_seq_len
        self.MAX_TOTAL_LEN = max_total_len if max_total_len is not None else max_num_groups * max_seq_len
        self.model_args = model_args
        self.PAD_VAL = PAD_VAL
        self.deferred = deferred        # True: __getitem__ defers the work to device_collate (one launch per batch)
        if store is None:
            import pandas as pd
            if df is None:
                df = pd.read_csv(meta_filepath)
            df = self.filter_meta(df, max_num_groups, max_seq_len, max_total_len, filter_uni, filter_platform,
                                  filter_category)
            df = df.sample(frac=train_ratio) if train_ratio < 1.0 else df
            store = PackedSVGStore.from_pkl_dir(data_dir, df, max_num_groups)
        self.df = df
        if store.G != max_num_groups:
            raise DsvgError(f"store was packed for {store.G} groups, dataset asks for {max_num_groups}")
        # the reference fails in torch.stack when a tensor is longer than the padded length (pad never truncates,
        # deepsvg/difflib/tensor.py:134-135); refuse such a store up front
        if store.max_group_len > self.MAX_SEQ_LEN or store.max_total_len > self.MAX_TOTAL_LEN:
            raise DsvgError(f"store holds groups of {store.max_group_len} / icons of {store.max_total_len} commands; "
                            f"limits are max_seq_len={self.MAX_SEQ_LEN}, max_total_len={self.MAX_TOTAL_LEN}")
        self.store = store if store.is_device else store.to(device)
        self.device = self.store.rows.device
        self.nb_augmentations = store.n_variants_of(0)          # svgtensor_dataset.py:52
        self._gen = None

    @staticmethod
    def filter_meta(df, max_num_groups, max_seq_len, max_total_len=None, filter_uni=None, filter_platform=None,
                    filter_category=None):
        """svgtensor_dataset.py:33-45"""
        if len(df) > 0:
            if filter_uni is not None:
                df = df[df.uni.isin(filter_uni)]
            if filter_platform is not None:
                df = df[df.platform.isin(filter_platform)]
            if filter_category is not None:
                df = df[df.category.isin(filter_category)]
            df = df[(df.nb_groups <= max_num_groups) & (df.max_len_group <= max_seq_len)]
            if max_total_len is not None:
                df = df[df.total_len <= max_total_len]
        return df

    def __len__(self):
        return self.store.n_icons * self.nb_augmentations

    def idx_to_id(self, idx):
        return self.store.ids[idx] if self.store.ids is not None else idx

    def manual_seed(self, seed):
        """seed of the augmentation choice (the reference draws it from python's `random`, :156)"""
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(seed))
        return self

    # ---- the batch path -----------------------------------------------------------------------------------------
    def batch(self, icon_idx, model_args=None, random_aug=True, aug=None):
        """What DataLoader(default collate) over `get(i, model_args)` returns for the icons `icon_idx`, assembled on
        the device by one kernel launch per layout: {"commands": [N, G, S+2], "args": [N, G, S+2, 11], ...}.
        `aug` (optional, per item) pins the stored variant; otherwise random_aug picks one uniformly per item
        (random.choice(tensors), svgtensor_dataset.py:156) or variant 0."""
        st = self.store
        model_args = list(model_args if model_args is not None else self.model_args)
        icon = torch.as_tensor(icon_idx, dtype=torch.int64).to(self.device).reshape(-1) % st.n_icons   # (:151)
        base = st.var_base[icon]
        if aug is not None:
            off = torch.as_tensor(aug, dtype=torch.int64).to(self.device).reshape(-1)
        elif random_aug:
            cnt = (st.var_base[icon + 1] - base).to(torch.float32)
            u = torch.rand(icon.numel(), device=self.device, generator=self._gen)
            off = torch.minimum((u * cnt).to(torch.int64), (cnt - 1).to(torch.int64))
        else:
            off = torch.zeros_like(icon)
        variant = (base.to(torch.int64) + off).to(torch.int32).contiguous()
        res = {}
        for grouped in (False, True):
            sfx = "_grouped" if grouped else ""
            keys = [k for k in ("commands", "args", "args_rel") if k + sfx in model_args]
            if not keys:
                continue
            L = (self.MAX_TOTAL_LEN if grouped else self.MAX_SEQ_LEN) + 2
            cmds, args, rel = ops.assemble_batch(st.rows, st.slot_off, variant, st.G, L, grouped,
                                                 want_args="args" in keys, want_rel="args_rel" in keys,
                                                 pad_val=float(self.PAD_VAL))
            for k, v in (("commands", cmds), ("args", args), ("args_rel", rel)):
                if k in keys:
                    res[k + sfx] = v
        unknown = [k for k in model_args if k not in res and k not in ("filling", "label")]
        if unknown:
            raise DsvgError(f"model_args {unknown} are not produced by the device batch assembly")
        if "filling" in model_args:
            res["filling"] = st.filling[icon].unsqueeze(-1)         # (N, G, 1) int64, svgtensor_dataset.py:198-199
        if "label" in model_args:
            if st.label is None:
                raise DsvgError("the store has no labels (no 'uni' / 'category' column in the meta table)")
            res["label"] = st.label[icon]
        return res

    # ---- per-item surface of the reference ------------------------------------------------------------------------
    def get(self, idx=0, model_args=None, random_aug=True, id=None, svg=None):
        """svgtensor_dataset.py:149-162 (without the `svg=` path): per-item tensors, e.g. commands [G, S+2]"""
        if svg is not None:
            raise NotImplementedError("get(svg=...) needs the reference's svglib; use deepsvg.svgtensor_dataset for it")
        if id is not None:
            idx = self.store.ids.index(id)
        out = self.batch([idx], model_args, random_aug)
        return {k: v[0] for k, v in out.items()}

    def __getitem__(self, idx):
        if self.deferred:
            return _Deferred(self, int(idx))
        return self.get(idx, self.model_args)

    def random_icon(self):
        return self.get(int(torch.randint(0, len(self), (1,))))


def device_collate(items):
    """collate_fn for DataLoader(num_workers=0): the sampled indices of one batch -> one device-side assembly"""
    if not items or not isinstance(items[0], _Deferred):
        raise DsvgError("device_collate expects items of a deepsvg_amd.dataset.SVGTensorDataset with deferred=True "
                        "(load_dataset sets it when cfg.collate_fn is device_collate)")
    ds = items[0].dataset
    return ds.batch([it.icon for it in items])


def load_dataset(cfg):
    """deepsvg/svgtensor_dataset.py:230-233"""
    return SVGTensorDataset(cfg.data_dir, cfg.meta_filepath, cfg.model_args, cfg.max_num_groups, cfg.max_seq_len,
                            cfg.max_total_len, cfg.filter_uni, cfg.filter_platform, cfg.filter_category,
                            cfg.train_ratio, deferred=getattr(cfg, "collate_fn", None) is device_collate,
                            device=getattr(cfg, "device", "cuda"))
