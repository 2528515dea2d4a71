"""ORACLE — test infrastructure, not product code.

A numpy restatement of the reference's per-item batch assembly, `SVGTensorDataset.get_data`
(deepsvg/svgtensor_dataset.py:164-205) with the `SVGTensor` helpers it chains
(deepsvg/difflib/tensor.py: from_data :85-88, add_sos :108-116, add_eos :125-132, pad :134-143, cmds :155-156,
args :158-162, get_relative_args :172-189), followed by the default collate of `torch.utils.data.DataLoader`
(stack over items; deepsvg/train.py:27-28).  Only tests/ may import this file; the product's batch assembly
(deepsvg_amd/dataset.py -> dsvg_assemble_batch, HIP) never does.

Parity pinning: the reference has no tests for this path; tests/golden/make_golden_batch.py runs
the real `SVGTensorDataset.get_data` (imported from /root/reference in the build container, viz-only
dependencies stubbed) on seeded icons and commits inputs + outputs as tests/golden/batch_assembly.npz;
tests/test_oracle_golden.py checks this file against that fixture.  All values are small integers stored in
float32, so the comparison is bit-exact.
"""
import numpy as np

EOS, SOS = 4, 5
N_ARGS = 11
# deepsvg/difflib/tensor.py:15-21
CMD_ARGS_MASK = np.array([[0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1],
                          [0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1],
                          [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1],
                          [1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1],
                          [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                          [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                          [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], dtype=bool)
# columns of the 14-wide stored row that survive SVGTensor.from_data + args(): the command (Index.COMMAND) and the
# 11 arg_keys columns; START_POS (6:8) is not kept (tensor.py:23-32,45-46,85-88)
DATA_COLS = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13]


def sequence(rows14, seq_len, pad_val=-1.0):
    """one SVGTensor.from_data(rows).add_eos().add_sos().pad(seq_len): (commands [L], args [L, 11]) float32.
    pad() never truncates (tensor.py:135): longer inputs keep their length, as in the reference."""
    rows14 = np.asarray(rows14, dtype=np.float32).reshape(-1, 14)
    n = rows14.shape[0]
    L = max(seq_len, n + 2)
    cmds = np.full((L,), float(EOS), dtype=np.float32)         # eos_token == pad_token (tensor.py:71)
    args = np.full((L, N_ARGS), pad_val, dtype=np.float32)
    cmds[0] = SOS
    cmds[1:1 + n] = rows14[:, 0]
    args[1:1 + n] = rows14[:, DATA_COLS[1:]]
    return cmds, args


def relative_args(cmds, args, pad_val=-1.0, args_dim=256):
    """SVGTensor.get_relative_args (tensor.py:172-189)"""
    data = args.copy()
    real = cmds < EOS
    d = data[real]
    start = d[:-1, 9:11].copy()
    d[1:, 5:7] -= start
    d[1:, 7:9] -= start
    d[1:, 9:11] -= start
    data[real] = d
    mask = CMD_ARGS_MASK[cmds.astype(np.int64)]
    data[mask] += args_dim - 1
    data[~mask] = pad_val
    return data


def get_data(t_sep, fillings, max_num_groups, max_seq_len, max_total_len=None, model_args=(), pad_val=-1.0):
    """SVGTensorDataset.get_data (svgtensor_dataset.py:164-205) for one icon: t_sep = list of [len, 14] arrays"""
    if max_total_len is None:
        max_total_len = max_num_groups * max_seq_len            # svgtensor_dataset.py:27-28
    t_sep = [np.asarray(t, dtype=np.float32).reshape(-1, 14) for t in t_sep]
    fillings = list(fillings)
    pad = max(max_num_groups - len(t_sep), 0)
    t_sep = t_sep + [np.zeros((0, 14), np.float32)] * pad
    fillings = fillings + [0] * pad
    grouped = [sequence(np.concatenate(t_sep, axis=0), max_total_len + 2, pad_val)]
    sep = [sequence(t, max_seq_len + 2, pad_val) for t in t_sep]
    res = {}
    for arg in set(model_args):
        if arg in ("filling", "label"):
            continue
        lst = grouped if "_grouped" in arg else sep
        key = arg.split("_grouped")[0]
        if key == "commands":
            res[arg] = np.stack([c for c, _ in lst])
        elif key == "args":
            res[arg] = np.stack([a for _, a in lst])
        elif key == "args_rel":
            res[arg] = np.stack([relative_args(c, a, pad_val) for c, a in lst])
        else:
            raise KeyError(arg)
    if "filling" in model_args:
        res["filling"] = np.asarray(fillings, dtype=np.int64).reshape(-1, 1)
    return res


def collate(items):
    """default_collate of a list of get_data dicts: stack along a new batch dimension"""
    return {k: np.stack([it[k] for it in items]) for k in items[0]}
