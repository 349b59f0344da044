"""Torch-op CPU restatement of the path -- TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; nothing
under ``vfa_amd/`` does.  It restates the reference's ``VFA.forward`` (``/root/reference/vfa/model/vfa_op.py:61-125``;
``project``: ``vfa/utils.py:50-59``) and the camera loop of ``VFANet.forward`` (``vfa/model/vfanet.py:64-82``) with the
SAME sequence of stock torch ops on explicit tensors instead of module state, so that

  * in fp32 on the CPU every pre-GEMM tensor is bit-identical to the reference's own run (pinned against the fixtures
    generated from the reference: tests/test_oracle_golden.py::test_torch_restatement_bitwise) -- this is the CPU
    baseline ``bench.py`` times on the GPU box's host cores, where /root/reference does
    not exist;
  * in float64 it is the differentiable gradient reference of the HIP backward kernels (tests/test_hip_backward.py).
"""
import torch
import torch.nn.functional as F

EPSILON = 1e-6
MAXIMUM_AREA_RATIO = 0.3


def world_coords(pts, data):
    """grid units -> world units (reference vfa_op.py:23-44).  ``pts`` is a temporary and may be modified in place."""
    if data == "MultiviewC":
        return pts / 1.
    if data == "MultiviewX":
        return pts / 40.
    if data == "Wildtrack":
        pts[..., 0] = pts[..., 0] * 2.5 - 300
        pts[..., 1] = pts[..., 1] * 2.5 - 900
        pts[..., 2] = pts[..., 2] * 2.5
        return pts
    raise UnboundLocalError(f"local variable 'coord' referenced before assignment ({data!r})")


def project(vectors, calib):
    """reference vfa/utils.py:50-59 (broadcast matmul, no behind-camera test)."""
    vectors = vectors.unsqueeze(-1)
    hom = torch.matmul(calib[..., :-1], vectors) + calib[..., -1:]
    hom = hom.squeeze(-1)
    return hom[..., :-1] / hom[..., -1:]


def vfa_stages(feature, calib, grid, z_layers, corner_off, data, image_size, crange=(-1, 0.95)):
    """feature (1,C,Hf,Wf), calib (3,4), grid (L,W,3), z_layers (nl), corner_off (8,3)
    -> dict(box (1,nl,L*W,4), area (1,1,nl,L*W), visible, integral (1,C,Hf,Wf), vox (L*W, C*nl; column c*nl + layer))."""
    dt = feature.dtype
    nl = z_layers.numel()
    z_corners = torch.zeros(nl, 1, 1, 3, dtype=z_layers.dtype, device=grid.device)
    z_corners[:, 0, 0, 2] = z_layers
    corners = grid.to(dt)[None].unsqueeze(0) + z_corners.view(-1, 1, 1, 3)                  # vfa_op.py:64
    corners = corners.unsqueeze(-2)
    corners3d = corners.repeat((1, 1, 1, 1, 8, 1)) + corner_off.to(dt).view(1, 1, 1, 1, 8, 3)  # :66
    corners3d = world_coords(corners3d, data)                                                # :68
    img = project(corners3d, calib.to(dt).view(-1, 1, 1, 1, 1, 3, 4))                        # :70-71
    Hf, Wf = feature.shape[2:]
    img_size = corners.new_tensor(list(image_size[::-1]))                                    # :75
    norm = (2 * img / img_size - 1).clamp(crange[0], crange[1])                              # :76
    box = torch.cat([torch.min(norm[..., 0], dim=-1, keepdim=True)[0], torch.min(norm[..., 1], dim=-1, keepdim=True)[0],
                     torch.max(norm[..., 0], dim=-1, keepdim=True)[0], torch.max(norm[..., 1], dim=-1, keepdim=True)[0]],
                    dim=-1)                                                                  # :81-86
    box = box.flatten(2, 3)                                                                  # (1,nl,L*W,4)
    area = (((box[..., 2:] - box[..., :2]).prod(dim=-1)) * Hf * Wf + EPSILON).unsqueeze(1)   # :104-105
    visible = torch.logical_and(area > EPSILON, area < (Hf * Wf * MAXIMUM_AREA_RATIO))       # :106
    integral = torch.cumsum(torch.cumsum(feature, dim=-1), dim=-2)                           # :110, 172-173
    lt = F.grid_sample(integral, box[..., [0, 1]], align_corners=False)                      # :112-115
    rb = F.grid_sample(integral, box[..., [2, 3]], align_corners=False)
    rt = F.grid_sample(integral, box[..., [2, 1]], align_corners=False)
    lb = F.grid_sample(integral, box[..., [0, 3]], align_corners=False)
    vox = (lt + rb - rt - lb) / area                                                         # :118
    vox = vox * visible                                                                      # :119
    vox = vox.permute(0, 3, 1, 2).flatten(0, 1).flatten(1, 2)                                # :120
    return dict(box=box, area=area, visible=visible, integral=integral, vox=vox)


def vfa_forward(feature, calib, grid, weight, bias, z_layers, corner_off, data, image_size, crange=(-1, 0.95)):
    """-> (1,Co,L,W): the stages above + ``collapse`` (weight (Co, C*nl) in the REFERENCE column order) + ReLU (:123-125)."""
    L, W = grid.shape[:2]
    st = vfa_stages(feature, calib, grid, z_layers, corner_off, data, image_size, crange)
    out = F.linear(st["vox"], weight.to(feature.dtype), bias.to(feature.dtype)).view(1, L, W, -1)
    return F.relu(out.permute(0, 3, 1, 2))


def vfanet_aggregate(lats, calibs, grid, weights, biases, z_layers, corner_off, data, image_size, cameras=None):
    """The camera loop of ``VFANet.forward`` (vfanet.py:64-82) given the lateral maps: lats / weights / biases are dicts
    {8,16,32}; ``cameras`` restricts the loop (the bounded sample of bench.py).  -> (1,Co,L,W)."""
    ortho = 0
    for cam in (range(calibs.shape[0]) if cameras is None else cameras):
        f = [vfa_forward(lats[s][cam:cam + 1], calibs[cam], grid, weights[s], biases[s], z_layers, corner_off, data,
                         image_size) for s in (8, 16, 32)]
        ortho = ortho + (f[0] + f[1] + f[2])                                                 # vfanet.py:79, 82
    return ortho
