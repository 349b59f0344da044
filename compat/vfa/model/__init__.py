"""``vfa.model``: `vfa_op` and `vfanet` come from the MI355X build (the two alias modules next to this file); the rest
of the reference's ``vfa/model`` directory (resnet.py, loss.py) is appended to the package path when it is present."""
import os

from .. import _reference_dirs

__path__ = [os.path.dirname(os.path.abspath(__file__))] + _reference_dirs("model")
