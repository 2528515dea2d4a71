"""Synthetic SVGTensor-format batches: the same tensors SVGTensorDataset.get_data delivers
(deepsvg/svgtensor_dataset.py:164-205, deepsvg/difflib/tensor.py:53-62): float32 commands (N, G, S+2) and args
(N, G, S+2, 11), SOS + <=S commands + EOS padding per group, PAD_VAL = -1 in unused argument slots.
"""
import torch

from .svgtensor import CMD_ARGS_MASK, EOS_ID, SOS_ID, M_ID, L_ID, C_ID


def make_batch(n, G=8, S=30, seed=0, device="cpu", min_groups=1):
    """Returns (commands (n,G,S+2) float32, args (n,G,S+2,11) float32).

    per icon: n_groups ~ U{min_groups..G} visible groups (a prefix); per visible group len ~ U{2..S} commands
    [m, then l|c ...]; invisible groups are [SOS, EOS, EOS...]; args drawn U{0..255} where CMD_ARGS_MASK enables
    the slot.  Only m/l/c occur (arcs are converted to Béziers and z is dropped: deepsvg/svglib/svg.py:333-349).
    """
    g = torch.Generator().manual_seed(int(seed))
    L = S + 2
    n_groups = torch.randint(min_groups, G + 1, (n,), generator=g)
    lens = torch.randint(2, S + 1, (n, G), generator=g)
    visible = torch.arange(G).unsqueeze(0) < n_groups.unsqueeze(1)            # (n, G)
    lens = torch.where(visible, lens, torch.zeros_like(lens))
    pos = torch.arange(L).view(1, 1, L)
    draw = torch.randint(L_ID, C_ID + 1, (n, G, L), generator=g)              # l or c
    commands = torch.full((n, G, L), EOS_ID, dtype=torch.long)
    body = (pos >= 1) & (pos <= lens.unsqueeze(-1))
    commands = torch.where(body, draw, commands)
    commands = torch.where((pos == 1) & body, torch.full_like(commands, M_ID), commands)
    commands[:, :, 0] = SOS_ID
    vals = torch.randint(0, 256, (n, G, L, 11), generator=g)
    mask = CMD_ARGS_MASK[commands].bool()                                     # (n, G, L, 11)
    args = torch.where(mask, vals, torch.full_like(vals, -1))
    return commands.float().to(device), args.float().to(device)


def make_batch_onestage(n, total_len=50, max_groups=8, seed=0, device="cpu"):
    """One-stage ("grouped") layout: commands_grouped (n, 1, total_len+2): SOS, then up to total_len commands in
    which several `m` mark group boundaries, then EOS padding."""
    g = torch.Generator().manual_seed(int(seed))
    L = total_len + 2
    lens = torch.randint(4, total_len + 1, (n, 1), generator=g)
    pos = torch.arange(L).view(1, 1, L)
    draw = torch.randint(L_ID, C_ID + 1, (n, 1, L), generator=g)
    starts = torch.rand((n, 1, L), generator=g) < (max_groups / float(total_len)) * 0.5
    commands = torch.full((n, 1, L), EOS_ID, dtype=torch.long)
    body = (pos >= 1) & (pos <= lens.unsqueeze(-1))
    commands = torch.where(body, draw, commands)
    commands = torch.where(body & (starts | (pos == 1)), torch.full_like(commands, M_ID), commands)
    # cap the number of groups at max_groups (group_embed has max_groups + 2 rows)
    n_m = (commands == M_ID).cumsum(dim=-1)
    commands = torch.where((commands == M_ID) & (n_m > max_groups), torch.full_like(commands, L_ID), commands)
    commands[:, :, 0] = SOS_ID
    vals = torch.randint(0, 256, (n, 1, L, 11), generator=g)
    mask = CMD_ARGS_MASK[commands].bool()
    args = torch.where(mask, vals, torch.full_like(vals, -1))
    return commands.float().to(device), args.float().to(device)


def det_state_dict(model, seed=1234, scale=1.0):
    """Deterministic, reference-independent weights for parity tests: every tensor of `model.state_dict()` with a
    floating dtype is filled from a numpy-free LCG keyed by (seed, parameter name) so that the reference model
    and this model can be loaded with bit-identical values without shipping a 41 MB checkpoint.  Matrices get
    ~U(-a, a) with a = sqrt(3 / fan_in) (unit-ish gain), LayerNorm weights ~1 +- 0.1, biases ~U(-0.05, 0.05).
    Each of the 4 layers of a stack gets different values (the reference's clones would otherwise be identical,
    hiding layer-order bugs: SURVEY.md §7.3-13)."""
    import zlib
    sd = {}
    for name, t in model.state_dict().items():
        if not torch.is_floating_point(t) or name.endswith(("cmd_args_mask", "square_subsequent_mask")):  # constants
            sd[name] = t.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        u = torch.rand(t.shape, generator=g, dtype=torch.float64) * 2 - 1
        if t.dim() >= 2:
            fan_in = t.shape[1]
            a = (3.0 / fan_in) ** 0.5
            if "embed" in name and "fcn" not in name:
                a = 1.0 * (3.0 / t.shape[1]) ** 0.5 * 4.0     # embedding rows ~ kaiming-normal scale
            v = u * a * scale
        elif "norm" in name and name.endswith("weight"):
            v = 1.0 + 0.1 * u
        else:
            v = 0.05 * u
        sd[name] = v.to(t.dtype)
    return sd
