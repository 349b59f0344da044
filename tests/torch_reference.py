"""Plain-PyTorch restatement of the projector (test infrastructure, differentiable, any dtype/device).

Same op sequence as the reference's ``VFA.forward`` (vfa/model/vfa_op.py:61-125) written against explicit
tensors instead of module state; used (a) in float64 as the gradient reference of the HIP backward kernels and
(b) as a second, independent forward check.  Never imported by the product.
"""
import torch
import torch.nn.functional as F


def world_coords(pts, data):
    """grid units -> world units (reference vfa_op.py:23-44)."""
    if data == "MultiviewC":
        return pts / 1.0
    if data == "MultiviewX":
        return pts / 40.0
    if data == "Wildtrack":
        return torch.stack([pts[..., 0] * 2.5 - 300, pts[..., 1] * 2.5 - 900, pts[..., 2] * 2.5], dim=-1)
    raise ValueError(data)


def vfa_forward(feature, calib, grid, weight, bias, z_layers, corner_off, data, image_size, crange=(-1, 0.95)):
    """feature (1,C,Hf,Wf), calib (3,4), grid (L,W,3), weight (Co, C*nl) in the REFERENCE column order (c*nl + layer),
    z_layers (nl), corner_off (8,3) -> (1,Co,L,W)."""
    dt = feature.dtype
    L, W = grid.shape[:2]
    nl = z_layers.numel()
    Hf, Wf = feature.shape[-2:]
    base = grid.to(dt).view(1, L, W, 1, 3) + torch.stack([torch.zeros_like(z_layers), torch.zeros_like(z_layers),
                                                          z_layers]).t().to(dt).view(nl, 1, 1, 1, 3)
    pts = world_coords(base + corner_off.to(dt).view(1, 1, 1, 8, 3), data)                # (nl,L,W,8,3)
    P = calib.to(dt)
    hom = pts @ P[:, :3].t() + P[:, 3]
    uv = hom[..., :2] / hom[..., 2:]
    size = torch.tensor([image_size[1], image_size[0]], dtype=dt, device=feature.device)
    norm = (2 * uv / size - 1).clamp(crange[0], crange[1])                                   # (nl,L,W,8,2)
    lo, hi = norm.min(dim=-2)[0], norm.max(dim=-2)[0]
    box = torch.cat([lo, hi], dim=-1).view(1, nl, L * W, 4)                                  # l,t,r,b
    area = ((box[..., 2] - box[..., 0]) * (box[..., 3] - box[..., 1]) * Hf * Wf + 1e-6).unsqueeze(1)
    visible = (area > 1e-6) & (area < Hf * Wf * 0.3)
    integral = feature.cumsum(-1).cumsum(-2)
    samp = lambda a, b: F.grid_sample(integral, box[..., [a, b]], align_corners=False)       # noqa: E731
    vox = (samp(0, 1) + samp(2, 3) - samp(2, 1) - samp(0, 3)) / area * visible               # (1,C,nl,L*W)
    vox = vox.permute(0, 3, 1, 2).flatten(0, 1).flatten(1, 2)                               # (L*W, C*nl)
    out = F.linear(vox, weight.to(dt), bias.to(dt)).view(1, L, W, -1)
    return F.relu(out.permute(0, 3, 1, 2))
