"""Alias for the reference's compiled extension module ``sort_vertices`` (built by
``vfa/evaluation/pyeval/cuda_op/setup.py`` from ``sort_vert.cpp`` / ``sort_vert_kernel.cu``): with ``compat/`` on
``PYTHONPATH`` the reference's ``cuda_op/cuda_ext.py:4`` (``import sort_vertices``) and through it ``IoU.py:3``
(``from .cuda_op.cuda_ext import sort_v``) bind to the HIP kernel ``vfa_sort_vertices_f32`` without building or editing anything
in the reference tree.  One function, the one ``sort_vert.cpp`` exports (``sort_vertices_forward``): same arguments, same
``(b, n, 9)`` int32 result, same requirement that the tensors live on the GPU.
"""
from vfa_amd.eval_ops import sort_vertices as sort_vertices_forward

__all__ = ["sort_vertices_forward"]
