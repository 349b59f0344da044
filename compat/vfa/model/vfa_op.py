"""Alias of the MI355X projector under the reference's module name (reference vfa/model/vfa_op.py)."""
from vfa_amd.vfa_op import EPSILON, MAXIMUM_AREA_RATIO, VFA, box_parameters, project  # noqa: F401
