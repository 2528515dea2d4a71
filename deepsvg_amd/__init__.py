"""deepsvg_amd — MI355X (gfx950) native implementation of the DeepSVG SVGTransformer train/infer hot path.

Public surface mirrors the reference (alexandre01/deepsvg):
    deepsvg_amd.SVGTransformer  <->  deepsvg.model.model.SVGTransformer
    deepsvg_amd.SVGLoss         <->  deepsvg.model.loss.SVGLoss
    deepsvg_amd.config.*        <->  deepsvg.model.config.*
"""
import os as _os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a stream that waits on an event
# holds up every other stream of its queue.  The data-parallel step uses five (main, layout plan, loss counts, RCCL, torch's
# copy stream): with four queues the plan stream ended up behind RCCL's wait for the previous step's graph, the host's read of
# the plan blocked for a whole step and could never run ahead of the GPU (one-rank RCCL group: 7.19 ms/step, 6.85 with six
# queues against 6.71 for the single-GPU step; with EIGHT the two-graph step of TrainStep(split_graph=True) ran at 15 ms -
# six is what every measured combination likes).  Must be set before the HIP runtime initialises, i.e. before the first device call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")

from .config import _DefaultConfig, Hierarchical, HierarchicalOrdered, OneStageOneShot  # noqa: F401
from .model import SVGTransformer  # noqa: F401
from .loss import SVGLoss  # noqa: F401

__all__ = ["SVGTransformer", "SVGLoss", "_DefaultConfig", "Hierarchical", "HierarchicalOrdered", "OneStageOneShot"]
