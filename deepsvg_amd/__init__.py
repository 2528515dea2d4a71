"""deepsvg_amd — MI355X (gfx950) native implementation of the DeepSVG SVGTransformer train/infer hot path.

Public surface mirrors the reference (alexandre01/deepsvg):
    deepsvg_amd.SVGTransformer  <->  deepsvg.model.model.SVGTransformer
    deepsvg_amd.SVGLoss         <->  deepsvg.model.loss.SVGLoss
    deepsvg_amd.config.*        <->  deepsvg.model.config.*
"""
from .config import _DefaultConfig, Hierarchical, HierarchicalOrdered, OneStageOneShot  # noqa: F401
from .model import SVGTransformer  # noqa: F401
from .loss import SVGLoss  # noqa: F401

__all__ = ["SVGTransformer", "SVGLoss", "_DefaultConfig", "Hierarchical", "HierarchicalOrdered", "OneStageOneShot"]
