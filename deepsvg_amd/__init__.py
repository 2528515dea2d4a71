"""deepsvg_amd — MI355X (gfx950) native implementation of the DeepSVG SVGTransformer train/infer hot path.

Public surface mirrors the reference (alexandre01/deepsvg):
    deepsvg_amd.SVGTransformer  <->  deepsvg.model.model.SVGTransformer
    deepsvg_amd.SVGLoss         <->  deepsvg.model.loss.SVGLoss
    deepsvg_amd.config.*        <->  deepsvg.model.config.*
"""
# (No process-wide side effects on import.  The data-parallel hipGraph step wants GPU_MAX_HW_QUEUES=6 in the environment BEFORE
# the HIP runtime comes up - see trainer.HW_QUEUES_NOTE; bench.py sets it, TrainStep warns when a data-parallel trainer
# finds it unset.)

from .config import _DefaultConfig, Hierarchical, HierarchicalOrdered, OneStageOneShot  # noqa: F401
from .model import SVGTransformer  # noqa: F401
from .loss import SVGLoss  # noqa: F401

__all__ = ["SVGTransformer", "SVGLoss", "_DefaultConfig", "Hierarchical", "HierarchicalOrdered", "OneStageOneShot"]
