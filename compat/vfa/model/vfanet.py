"""Alias of the MI355X VFANet under the reference's module name (reference vfa/model/vfanet.py)."""
from vfa_amd.vfanet import VFANet  # noqa: F401
