"""Alias package: lets the reference's scripts (`train.py`, `evaluate.py`: ``from vfa.model.vfanet import VFANet``)
pick up the MI355X build of the hot path WITHOUT editing the reference tree.

    PYTHONPATH=<this repo>/compat:<this repo>:<reference checkout>  python train.py --data MultiviewC

``vfa.model.vfa_op`` and ``vfa.model.vfanet`` resolve to ``vfa_amd`` (same class names, constructor and forward
signatures, ``state_dict`` keys: reference vfa/model/vfa_op.py:46-125, vfa/model/vfanet.py:14-149); every other
``vfa.*`` module (data, trainer, loss, evaluation, utils ...) still resolves to the reference checkout, found on
``sys.path`` or through ``VFA_REFERENCE_ROOT``.  Nothing of the reference is copied here.
"""
import os
import sys

__path__ = [os.path.dirname(os.path.abspath(__file__))]


def _reference_dirs(sub=""):
    roots = [os.environ["VFA_REFERENCE_ROOT"]] if os.environ.get("VFA_REFERENCE_ROOT") else []
    roots += [p for p in sys.path if p]
    out = []
    for root in roots:
        cand = os.path.join(os.path.abspath(root), "vfa", sub) if sub else os.path.join(os.path.abspath(root), "vfa")
        if os.path.isdir(cand) and os.path.abspath(cand) not in map(os.path.abspath, out) \
                and not os.path.abspath(cand).startswith(os.path.dirname(os.path.abspath(__file__))):
            out.append(cand)
    return out


__path__ += _reference_dirs()
