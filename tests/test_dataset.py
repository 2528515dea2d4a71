"""Device-side batch assembly (deepsvg_amd/dataset.py, dsvg_assemble_batch) against the oracle
(oracle/batch_assembly_oracle.py) and the reference's own outputs (tests/golden/batch_assembly.npz)."""
import os
import pickle

import numpy as np
import pytest
import torch

from deepsvg_amd import dataset as D
from deepsvg_amd.lib import DsvgError
from oracle import batch_assembly_oracle as B
from tests import helpers as H

TAGS = ["icons", "arcs"]


def _oracle_batch(icons, fillings, G, S, T, keys):
    return B.collate([B.get_data(g, f, G, S, T, keys) for g, f in zip(icons, fillings)])


def _random_icons(rng, n_icons, n_var, G, S, T):
    """[icon][variant] -> list of group arrays, m/l/c commands only (what real data holds)"""
    used = {0: [12, 13], 1: [12, 13], 2: [8, 9, 10, 11, 12, 13]}
    icons, fills = [], []
    for _ in range(n_icons):
        variants = []
        ng = int(rng.integers(1, G + 1))
        for _v in range(n_var):
            groups, left = [], T
            for _g in range(ng):
                ln = int(rng.integers(1, max(min(S, left - (ng - len(groups) - 1)), 1) + 1))
                left -= ln
                t = np.full((ln, 14), -1.0, dtype=np.float32)
                t[:, 0] = rng.integers(1, 3, size=ln)
                t[0, 0] = 0
                for r in range(ln):
                    for c in used[int(t[r, 0])]:
                        t[r, c] = rng.integers(0, 256)
                groups.append(t)
            variants.append(groups)
        icons.append(variants)
        fills.append([int(rng.integers(0, 3)) for _ in range(ng)])
    return icons, fills


# ---------------------------------------------------------------------------------------------------------------
# CPU: oracle pinned to the reference, host-side packing
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", TAGS)
def test_batch_oracle_matches_reference_golden(tag):
    icons, fillings, (G, S, T), expected = H.golden_batch(tag)
    out = _oracle_batch(icons, fillings, G, S, T, H.BATCH_KEYS)
    for k in H.BATCH_KEYS:
        assert out[k].shape == expected[k].shape, k
        assert np.array_equal(out[k], expected[k]), k


def test_store_layout_and_roundtrip(tmp_path):
    icons, fillings, (G, S, T), _ = H.golden_batch("icons")
    st = D.PackedSVGStore.from_icons([[g] for g in icons], fillings, labels=list(range(len(icons))), max_num_groups=G)
    assert st.rows.dtype == np.int16 and st.rows.shape[1] == 12 and st.slot_off.dtype == np.int32
    assert st.n_icons == len(icons) and st.slot_off.shape[0] == len(icons) * G + 1
    assert np.array_equal(st.var_base, np.arange(len(icons) + 1))
    for i, groups in enumerate(icons):
        for g in range(G):
            lo, hi = st.slot_off[i * G + g], st.slot_off[i * G + g + 1]
            want = groups[g][:, list(D.ROW_COLS)] if g < len(groups) else np.zeros((0, 12))
            assert np.array_equal(st.rows[lo:hi].astype(np.float32), want.astype(np.float32))
        assert list(st.filling[i]) == fillings[i] + [0] * (G - len(fillings[i]))
    assert st.max_group_len <= S and st.max_total_len <= T
    st.save(tmp_path / "store.npz")
    st2 = D.PackedSVGStore.load(tmp_path / "store.npz")
    for name in ("rows", "slot_off", "var_base", "filling", "label"):
        assert np.array_equal(getattr(st, name), getattr(st2, name)), name
    assert (st2.G, st2.max_group_len, st2.max_total_len) == (st.G, st.max_group_len, st.max_total_len)


def test_store_refuses_what_the_reference_cannot_stack():
    t = np.zeros((3, 14), np.float32)
    with pytest.raises(DsvgError):
        D.PackedSVGStore.from_icons([[[t] * 9]], max_num_groups=8)          # more groups than slots
    bad = t.copy()
    bad[0, 12] = 0.5
    with pytest.raises(DsvgError):
        D.PackedSVGStore.from_icons([[[bad]]], max_num_groups=8)            # not numericalised
    bad = t.copy()
    bad[1, 0] = 7
    with pytest.raises(DsvgError):
        D.PackedSVGStore.from_icons([[[bad]]], max_num_groups=8)            # command outside the vocabulary


def test_store_from_reference_pkl_layout(tmp_path):
    """the on-disk format of the reference: <id>.pkl = {"tensors": [variant][group] -> [len, 14], "fillings"} and a
    meta CSV with id / nb_groups / max_len_group / total_len (svgtensor_dataset.py:33-52,106-109)"""
    import pandas as pd
    rng = np.random.default_rng(5)
    icons, fills = _random_icons(rng, 6, 3, 8, 30, 50)
    meta = []
    for i, (variants, f) in enumerate(zip(icons, fills)):
        with open(tmp_path / f"{100 + i}.pkl", "wb") as fh:
            pickle.dump({"tensors": [[torch.from_numpy(g) for g in v] for v in variants], "fillings": f}, fh)
        lens = [len(g) for g in variants[0]]
        meta.append(dict(id=100 + i, nb_groups=len(lens) + (5 if i == 2 else 0), max_len_group=max(lens),
                         total_len=sum(lens), category="arrows" if i % 2 else "food"))
    df = pd.DataFrame(meta)
    kept = D.SVGTensorDataset.filter_meta(df, 8, 30, 50)
    assert list(kept.id) == [100, 101, 103, 104, 105]                       # icon 102 claims 13 groups
    assert list(D.SVGTensorDataset.filter_meta(df, 8, 30, 50, filter_category=["food"]).id) == [100, 104]
    st = D.PackedSVGStore.from_pkl_dir(str(tmp_path), kept, 8)
    assert st.n_icons == 5 and st.ids == [100, 101, 103, 104, 105]
    assert np.array_equal(st.var_base, 3 * np.arange(6))
    assert list(st.label) == [D._CATEGORIES.index(c) for c in kept.category]
    want = D.PackedSVGStore.from_icons([icons[i] for i in (0, 1, 3, 4, 5)], [fills[i] for i in (0, 1, 3, 4, 5)],
                                       max_num_groups=8)
    assert np.array_equal(st.rows, want.rows) and np.array_equal(st.slot_off, want.slot_off)
    assert np.array_equal(st.filling, want.filling)


def test_dataset_needs_the_device():
    icons, fillings, (G, S, T), _ = H.golden_batch("icons")
    st = D.PackedSVGStore.from_icons([[g] for g in icons], fillings, max_num_groups=G)
    with pytest.raises(DsvgError):
        D.SVGTensorDataset(store=st, model_args=["commands", "args"], max_num_groups=G, max_seq_len=S,
                           max_total_len=T, device="cpu")
    with pytest.raises(DsvgError):                                          # longer than the padded length
        D.SVGTensorDataset(store=st, model_args=["commands"], max_num_groups=G, max_seq_len=3, device="cpu")


# ---------------------------------------------------------------------------------------------------------------
# GPU: the kernel
# ---------------------------------------------------------------------------------------------------------------
def _dataset(icons_variants, fillings, G, S, T, keys, device, labels=None, **kw):
    st = D.PackedSVGStore.from_icons(icons_variants, fillings, labels=labels, max_num_groups=G)
    return D.SVGTensorDataset(store=st, model_args=keys, max_num_groups=G, max_seq_len=S, max_total_len=T,
                              device=device, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_assembled_batch_equals_reference_golden(gpu_device, tag):
    icons, fillings, (G, S, T), expected = H.golden_batch(tag)
    ds = _dataset([[g] for g in icons], fillings, G, S, T, H.BATCH_KEYS, gpu_device)
    out = ds.batch(list(range(len(icons))), random_aug=False)
    assert set(out) == set(H.BATCH_KEYS)
    for k in H.BATCH_KEYS:
        got = out[k].cpu().numpy()
        assert got.dtype == expected[k].dtype and got.shape == expected[k].shape, k
        assert np.array_equal(got, expected[k]), k                          # integers in float32: bit-exact
    # a permuted, repeating index list and the per-item surface
    idx = [5, 0, 5, len(icons) - 1, 2]
    out = ds.batch(idx, ["commands", "args_rel_grouped"], random_aug=False)
    assert set(out) == {"commands", "args_rel_grouped"}
    assert np.array_equal(out["commands"].cpu().numpy(), expected["commands"][idx])
    assert np.array_equal(out["args_rel_grouped"].cpu().numpy(), expected["args_rel_grouped"][idx])
    item = ds.get(len(icons) + 3, random_aug=False)                         # idx % len(df), svgtensor_dataset.py:151
    for k in H.BATCH_KEYS:
        assert np.array_equal(item[k].cpu().numpy(), expected[k][3]), k


@pytest.mark.gpu
def test_variant_choice_and_dataloader_seam(gpu_device):
    rng = np.random.default_rng(9)
    G, S, T, n_var = 8, 30, 50, 4
    icons, fills = _random_icons(rng, 40, n_var, G, S, T)
    keys = ["commands", "args", "commands_grouped", "args_grouped", "filling", "label"]
    labels = [int(v) for v in rng.integers(0, 50, size=len(icons))]
    ds = _dataset(icons, fills, G, S, T, keys, gpu_device, labels=labels, deferred=True)
    assert len(ds) == len(icons) * n_var and ds.nb_augmentations == n_var
    # pinned variants
    idx = [int(v) for v in rng.integers(0, len(icons), size=64)]
    aug = [int(v) for v in rng.integers(0, n_var, size=64)]
    out = ds.batch(idx, aug=aug)
    want = _oracle_batch([icons[i][a] for i, a in zip(idx, aug)], [fills[i] for i in idx], G, S, T, keys)
    for k in keys[:-1]:
        assert np.array_equal(out[k].cpu().numpy(), want[k]), k
    assert out["label"].tolist() == [labels[i] for i in idx]
    # random variants: every item is one of its icon's stored variants, and all variants get drawn
    ds.manual_seed(3)
    idx = list(range(len(icons))) * 8
    out = ds.batch(idx)
    cmds = out["commands"].cpu().numpy()
    per_variant = [[_oracle_batch([icons[i][a]], [fills[i]], G, S, T, ["commands"])["commands"][0] for a in range(n_var)]
                   for i in range(len(icons))]
    seen = set()
    for j, i in enumerate(idx):
        hits = [a for a in range(n_var) if np.array_equal(cmds[j], per_variant[i][a])]
        assert hits, j
        seen.update(hits)
    assert seen == set(range(n_var))
    # the reference's loader seam: DataLoader(dataset, collate_fn=cfg.collate_fn) (deepsvg/train.py:27-28)
    loader = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=True, drop_last=True, num_workers=0,
                                         collate_fn=D.device_collate)
    n = 0
    for data in loader:
        assert data["commands"].shape == (16, G, S + 2) and data["commands"].is_cuda
        assert data["args_grouped"].shape == (16, 1, T + 2, 11) and data["filling"].shape == (16, G, 1)
        assert bool((data["commands"][:, :, 0] == 5).all())
        n += 1
    assert n == len(ds) // 16


@pytest.mark.gpu
def test_assembled_batch_drives_the_model(gpu_device):
    """the assembled tensors are what SVGTransformer.forward takes (deepsvg/train.py:93-97): same logits as the same
    icons assembled by the oracle on the host"""
    import deepsvg_amd
    from deepsvg_amd.synthetic import det_state_dict
    rng = np.random.default_rng(21)
    G, S, T = 8, 30, 50
    icons, fills = _random_icons(rng, 12, 1, G, S, T)
    keys = ["commands", "args"]
    ds = _dataset(icons, fills, G, S, T, keys, gpu_device)
    data = ds.batch(list(range(12)), random_aug=False)
    want = _oracle_batch([v[0] for v in icons], fills, G, S, T, keys)
    cfg = H.build_cfg("hier")
    model = deepsvg_amd.SVGTransformer(cfg)
    model.load_state_dict(det_state_dict(model, seed=4))
    model = model.to(gpu_device).eval()
    with torch.no_grad():
        a = model(data["commands"], data["args"], data["commands"], data["args"])
        hc, ha = (torch.from_numpy(want[k]).to(gpu_device) for k in keys)
        b = model(hc, ha, hc, ha)
    assert torch.equal(a["command_logits"], b["command_logits"])
    assert torch.equal(a["args_logits"], b["args_logits"])


@pytest.mark.gpu
def test_full_size_batch_512_icons(gpu_device):
    """BASELINE's batch (512 icons, G=8, S=30): the whole batch against the oracle, plus size-independent properties"""
    rng = np.random.default_rng(33)
    G, S, T = 8, 30, 240
    icons, fills = _random_icons(rng, 512, 2, G, S, T)
    keys = ["commands", "args", "args_rel", "commands_grouped", "args_grouped"]
    ds = _dataset(icons, fills, G, S, None, keys, gpu_device)
    aug = [int(v) for v in rng.integers(0, 2, size=512)]
    out = ds.batch(list(range(512)), aug=aug)
    want = _oracle_batch([icons[i][a] for i, a in enumerate(aug)], fills, G, S, None, keys)
    for k in keys:
        assert np.array_equal(out[k].cpu().numpy(), want[k]), k
    cmd, cg = out["commands"], out["commands_grouped"]
    assert cmd.shape == (512, G, S + 2) and cg.shape == (512, 1, T + 2)
    # every sequence: SOS first, at least one EOS, EOS-closed; grouped length = sum of the group lengths
    assert bool((cmd[..., 0] == 5).all()) and bool((cmd[..., -1] == 4).all())
    n_real = (cmd < 4).sum(dim=(1, 2))
    assert torch.equal(n_real, (cg < 4).sum(dim=(1, 2)))
    # the multiset of stored argument values is preserved by the grouped layout
    assert float(out["args"].clamp_min(0).double().sum()) == float(out["args_grouped"].clamp_min(0).double().sum())
