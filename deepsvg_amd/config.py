"""Model hyper-parameter surface, attribute-for-attribute the same as deepsvg/model/config.py:4-108 so that a
reference `ModelConfig` instance (e.g. configs/deepsvg/hierarchical_ordered.py:4-9) can be handed to
deepsvg_amd.SVGTransformer unchanged, and these classes can be handed to the reference model in tests.
"""
from .svgtensor import COMMANDS_SIMPLIFIED


class _DefaultConfig:
    """Model config (defaults of deepsvg/model/config.py:8-45)."""

    def __init__(self):
        self.args_dim = 256              # coordinate numericalization (8-bit)
        self.n_args = 11                 # rx, ry, phi, fA, fS, qx1, qy1, qx2, qy2, x, y
        self.n_commands = len(COMMANDS_SIMPLIFIED)  # m, l, c, a, EOS, SOS, z

        self.dropout = 0.1

        self.model_type = "transformer"  # the "lstm" variant of the reference is not on the hot path

        self.encode_stages = 1           # 1 | 2
        self.decode_stages = 1           # 1 | 2

        self.use_resnet = True
        self.use_vae = True

        self.pred_mode = "one_shot"      # "one_shot" | "autoregressive"
        self.rel_targets = False

        self.label_condition = False
        self.n_labels = 100
        self.dim_label = 64

        self.self_match = False

        self.n_layers = 4
        self.n_layers_decode = 4
        self.n_heads = 8
        self.dim_feedforward = 512
        self.d_model = 256

        self.dim_z = 256

        self.max_num_groups = 8
        self.max_seq_len = 30
        self.max_total_len = self.max_num_groups * self.max_seq_len

        self.num_groups_proposal = self.max_num_groups

    def get_model_args(self):
        """deepsvg/model/config.py:47-60"""
        model_args = []
        model_args += ["commands_grouped", "args_grouped"] if self.encode_stages <= 1 else ["commands", "args"]
        if self.rel_targets:
            model_args += ["commands_grouped", "args_rel_grouped"] if self.decode_stages == 1 else ["commands", "args_rel"]
        else:
            model_args += ["commands_grouped", "args_grouped"] if self.decode_stages == 1 else ["commands", "args"]
        if self.label_condition:
            model_args.append("label")
        return model_args


class Sketchformer(_DefaultConfig):
    """Transformer - autoregressive - one-stage, relative targets (deepsvg/model/config.py:74-80)"""

    def __init__(self):
        super().__init__()
        self.pred_mode = "autoregressive"
        self.rel_targets = True


class OneStageOneShot(_DefaultConfig):
    """Transformer - one-shot - one-stage (deepsvg/model/config.py:83-89)"""

    def __init__(self):
        super().__init__()
        self.encode_stages = 1
        self.decode_stages = 1


class Hierarchical(_DefaultConfig):
    """Transformer - one-shot - two-stage - ordered (deepsvg/model/config.py:92-98)"""

    def __init__(self):
        super().__init__()
        self.encode_stages = 2
        self.decode_stages = 2


class HierarchicalSelfMatching(_DefaultConfig):
    """Transformer - one-shot - two-stage - Hungarian assignment (deepsvg/model/config.py:101-108)"""

    def __init__(self):
        super().__init__()
        self.encode_stages = 2
        self.decode_stages = 2
        self.self_match = True


class HierarchicalOrdered(Hierarchical):
    """The model config of configs/deepsvg/hierarchical_ordered.py:4-9 (the north-star config)."""

    def __init__(self):
        super().__init__()
        self.label_condition = False
        self.use_vae = False
