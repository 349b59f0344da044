"""Geometry helpers that sit either side of the projector.

Mirrors the two functions of the reference's ``vfa/utils.py`` that touch the hot path:

* ``make_grid``  (reference ``vfa/utils.py:16-37``)  -- producer of the ``grid`` input.
* ``project``    (reference ``vfa/utils.py:50-59``)  -- 3x4 pinhole projection.  The product
  path never calls this Python version: projection runs inside the HIP box-parameter kernel
  (``vfa_amd/csrc/vfa_kernels.hip``).  It is kept for callers that used ``vfa.utils.project``
  directly (the visualisation side path).
"""
import torch

DATASETS = ("MultiviewC", "MultiviewX", "Wildtrack")


def make_grid(world_size=(3900, 3900), grid_offset=(0, 0, 0), cube_LW=(25, 25), dataset="Wildtrack"):
    """BEV cell *origins* in grid units, shape (L, W, 3), fp32.

    Same signature, defaults and axis conventions as the reference (``vfa/utils.py:16``):
    for Wildtrack ``(length, width) = world_size[::-1]`` and x varies along dim 0; otherwise
    ``(length, width) = world_size`` and x varies along dim 1.  z is ``grid_offset[2]``.
    """
    if dataset == "Wildtrack":
        length, width = world_size[::-1]
    else:
        length, width = world_size
    xoff, yoff, zoff = grid_offset
    xcoords = torch.arange(0.0, width, cube_LW[0]) + xoff
    ycoords = torch.arange(0.0, length, cube_LW[1]) + yoff
    if dataset == "Wildtrack":
        xx, yy = torch.meshgrid(xcoords, ycoords, indexing="ij")
    else:
        yy, xx = torch.meshgrid(ycoords, xcoords, indexing="ij")
    return torch.stack([xx, yy, torch.full_like(xx, zoff)], dim=-1)


def project(vectors, calib):
    """Pinhole projection of (..., 3) points with a broadcastable (..., 3, 4) matrix.

    Reference ``vfa/utils.py:50-59``; no behind-camera test, exactly like the reference.
    """
    vectors = vectors.unsqueeze(-1)
    homography = torch.matmul(calib[..., :-1], vectors) + calib[..., -1:]
    homography = homography.squeeze(-1)
    return homography[..., :-1] / homography[..., -1:]
