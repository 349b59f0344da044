"""vfa_amd -- MI355X-native multiview image-feature -> voxel projection and aggregation (the VFA hot path).

Public surface mirrors the reference's ``vfa.model`` / ``vfa.utils`` names for this path:
``VFA`` (projector), ``aggregate_views`` (the camera loop of ``VFANet.forward``), ``make_grid``, ``project``.
The compute runs in hand-written HIP kernels (``vfa_amd/csrc``) behind the C ABI of ``include/vfa_hip.h``.
"""
from .utils import make_grid, project  # noqa: F401
from .vfa_op import VFA, FrameGeometry, box_parameters  # noqa: F401
from .lazy import DeferredOrtho, materialize  # noqa: F401
from .aggregate import (aggregate_views, all_reduce_ortho, all_reduce_prehead_grads, camera_shard, reduce_ortho,  # noqa: F401
                        reduce_scatter_ortho, row_bands)

__all__ = ["VFA", "FrameGeometry", "aggregate_views", "all_reduce_ortho", "all_reduce_prehead_grads", "camera_shard", "box_parameters", "make_grid", "materialize", "project",
           "reduce_ortho", "reduce_scatter_ortho", "row_bands"]
