"""Consumers of the path on the GPU: BEV decode and the AP/AOS metric's vertex sorter.

* ``sort_vertices`` / ``sort_v``  -- drop-in for the reference's CUDA op (``vfa/evaluation/pyeval/cuda_op/cuda_ext.py:6-17``,
  kernel ``sort_vert_kernel.cu:42-134``): same arguments, same ``(b, n, 9)`` int32 result, on the HIP kernel
  ``vfa_sort_vertices_f32``.  ``evaluateAPAOS.py:79-83`` hard-codes ``device('cuda')``, which IS the MI355X under PyTorch-ROCm.
* ``BEVDecoder``  -- ``ObjectEncoder.nms / decode3d / decode2d`` (``vfa/data/encoder.py:230-305``) with the dataset constants
  passed explicitly: sigmoid + 5x5 max-pool NMS in one HIP kernel (``vfa_bev_nms_f32``), then top-k and the gathers (torch ops on
  the device: a few hundred numbers).
"""
import numpy as np
import torch

from . import _lib


def sort_vertices(vertices, mask, num_valid):
    """vertices (b,n,m,2) f32, mask (b,n,m) bool, num_valid (b,n) int32 -> idx (b,n,9) int32 (reference ``sort_v``)."""
    _lib.require_device(vertices, mask, num_valid)
    assert vertices.dtype == torch.float32 and mask.dtype == torch.bool and num_valid.dtype == torch.int32
    vertices, mask, num_valid = vertices.contiguous(), mask.contiguous(), num_valid.contiguous()
    b, n, m, _ = vertices.shape
    idx = torch.zeros((b, n, 9), dtype=torch.int32, device=vertices.device)
    _lib.call("vfa_sort_vertices_f32", _lib.ptr(vertices), _lib.ptr(mask.view(torch.uint8)), _lib.ptr(num_valid), _lib.ptr(idx),
              b, n, m, _lib.current_stream_handle())
    return idx


sort_v = sort_vertices  # the name the reference imports (IoU.py:3)


def bev_nms(heatmap):
    """heatmap (1,1,L,W) logits -> (1,1,L,W): sigmoid where it is the 5x5 maximum, else 0 (encoder.py:230-232, :238)."""
    _lib.require_device(heatmap)
    h = heatmap.to(torch.float32).contiguous()
    L, W = h.shape[-2:]
    assert h.numel() == L * W, "batch 1, one class, like the reference"
    conf = torch.empty_like(h)
    _lib.call("vfa_bev_nms_f32", _lib.ptr(h), _lib.ptr(conf), L, W, _lib.current_stream_handle())
    return conf


class BEVDecoder:
    """``ObjectEncoder``'s decode half (encoder.py:230-305).  ``base`` is the dataset class name, ``world_size`` / ``cube_LWH``
    as in the dataset configs, ``dimension_mean`` = ``classAverage.get_mean(...)`` (3D only)."""

    def __init__(self, base, world_size, cube_LWH, dimension_mean=None, topk=100):
        self.base, self.topk = base, topk
        self.world_size = np.array(world_size)
        self.grid_size = self.world_size / np.array(cube_LWH)[:2]
        self.dimension_mean = dimension_mean

    def nms(self, heatmap):
        return bev_nms(heatmap)

    def _peaks(self, pred):
        heatmap, tytx = pred["heatmap"], pred["loc_offset"]
        device, dtype = heatmap.device, heatmap.dtype
        conf = self.nms(heatmap).flatten(start_dim=2).transpose(1, 2)            # (1, L*W, 1)
        conf, _ = torch.max(conf, dim=-1)
        L, W = heatmap.shape[2:]
        grid_y, grid_x = torch.meshgrid(torch.arange(L, dtype=dtype, device=device), torch.arange(W, dtype=dtype, device=device),
                                        indexing="ij")
        tytx = torch.sigmoid(tytx)
        cy = (grid_y[None, ...] + tytx[..., 0]).flatten(start_dim=1) / self.grid_size[0] * self.world_size[0]
        cx = (grid_x[None, ...] + tytx[..., 1]).flatten(start_dim=1) / self.grid_size[1] * self.world_size[1]
        _, topk_index = torch.topk(conf, k=min(self.topk, conf.shape[1]), dim=1)
        return conf, cy, cx, topk_index

    def decode3d(self, pred, cls_thresh):
        conf, cy, cx, topk_index = self._peaks(pred)
        thtwtl, orient = pred["dim_offset"], pred["rotation"]
        mean = self.dimension_mean
        dims = [torch.exp(thtwtl[..., k]).flatten(start_dim=1) * mean[k] for k in range(3)]
        _, orient_idx = torch.max(torch.sigmoid(orient), dim=-1)
        orient_idx = orient_idx.flatten(start_dim=1)
        out = [torch.gather(x, dim=1, index=topk_index) for x in [conf, cy, cx, *dims, orient_idx]]
        mask = out[0] > cls_thresh
        return {"conf": out[0][mask],
                "location": torch.stack([out[2][mask], out[1][mask], torch.zeros_like(out[1][mask])], dim=-1),
                "dimension": torch.stack([out[3][mask], out[4][mask], out[5][mask]], dim=-1),
                "rotation": torch.deg2rad(out[6][mask].to(torch.float32))}

    def decode2d(self, pred, cls_thresh):
        conf, cy, cx, topk_index = self._peaks(pred)
        out = [torch.gather(x, dim=1, index=topk_index) for x in [conf, cy, cx]]
        mask = out[0] > cls_thresh
        first, second = (out[1], out[2]) if self.base == "Wildtrack" else (out[2], out[1])
        return {"conf": out[0][mask],
                "location": torch.stack([first[mask], second[mask], torch.zeros_like(out[1][mask])], dim=-1)}

    def batch_decode(self, pred, cls_thresh):
        return self.decode3d(pred, cls_thresh) if self.base in ("MultiviewC", "MVM3D") else self.decode2d(pred, cls_thresh)
