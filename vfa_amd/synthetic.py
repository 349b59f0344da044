"""Synthetic cameras, grids and feature maps for the bench and the parity tests.

There are no datasets on the build or GPU boxes, so every workload is generated: pinhole
cameras on a ring around the ground-plane grid, looking at its centre.
World units follow the reference's per-dataset conventions (``vfa/model/vfa_op.py:23-44``):
MultiviewC world = grid units (cm); MultiviewX world = grid / 40 (m); Wildtrack world =
grid * 2.5 + (-300, -900) (cm).
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

from .utils import make_grid


def look_at_camera(pos, target, focal, image_wh):
    """3x4 projection matrix K.[R | -R.pos] (float64 numpy) of a camera at ``pos`` looking at ``target``."""
    pos = np.asarray(pos, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    z = target - pos
    z /= np.linalg.norm(z)
    x = np.cross(z, np.array([0.0, 0.0, 1.0]))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    W, H = image_wh
    K = np.array([[focal, 0.0, W / 2.0], [0.0, focal, H / 2.0], [0.0, 0.0, 1.0]])
    return K @ np.concatenate([R, (-R @ pos)[:, None]], axis=1)


def ring_cameras(n, centre, radius, height, focal, image_wh, phase=0.0):
    """``n`` cameras on a circle of ``radius`` around ``centre`` at ``height``, looking at the centre.

    Returns a (n, 3, 4) fp32 tensor, cast from float64 the way the reference's ``collate`` does
    (``vfa/utils.py:44``: ``torch.Tensor(calib)``).
    """
    cx, cy, cz = centre
    mats = []
    for i in range(n):
        a = 2.0 * math.pi * i / n + phase
        pos = (cx + radius * math.cos(a), cy + radius * math.sin(a), cz + height)
        mats.append(look_at_camera(pos, centre, focal, image_wh))
    return torch.tensor(np.stack(mats), dtype=torch.float32)


def feature_sizes(image_hw, strides=(8, 16, 32)):
    """ResNet spatial sizes: every stride-2 stage maps n -> floor((n-1)/2)+1 (reference ``resnet.py:138-147``)."""
    out = []
    for s in strides:
        h, w = image_hw
        for _ in range(int(round(math.log2(s)))):
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out.append((h, w))
    return out


# Workloads of BASELINE.json (restated in SURVEY.md section 8d / Appendix C).
#   data        : dataset kind -> world-unit conversion inside the projector
#   image_size  : ORIGINAL image (H, W) used for normalisation (reference vfa_op.py:75)
#   feat_image  : image (H, W) the backbone actually sees -> feature map sizes
WORKLOADS = {
    # C1/C2 shipped MultiviewC config (reference config.py:5-28)
    "multiviewc_156x156x5": dict(data="MultiviewC", n_cam=7, image_size=(720, 1280), feat_image=(720, 1280),
                                 world_size=(3900, 3900), cube_size=(25, 25, 32), grid_height=160),
    # C2 BASELINE: 37.5 m x 37.5 m field, 200x200x1 grid
    "multiviewc_200x200x1": dict(data="MultiviewC", n_cam=7, image_size=(720, 1280), feat_image=(720, 1280),
                                 world_size=(3750, 3750), cube_size=(18.75, 18.75, 160), grid_height=160),
    # C3 shipped Wildtrack config (reference config.py:60-85), images resized to 720x1280
    "wildtrack_120x360x8": dict(data="Wildtrack", n_cam=7, image_size=(1080, 1920), feat_image=(720, 1280),
                                world_size=(480, 1440), cube_size=(4, 4, 4), grid_height=32),
    # C3 BASELINE: native 1080p, 480x1440x1 grid
    "wildtrack_480x1440x1": dict(data="Wildtrack", n_cam=7, image_size=(1080, 1920), feat_image=(1080, 1920),
                                 world_size=(480, 1440), cube_size=(1, 1, 4), grid_height=4),
    # C4 shipped MultiviewX config (reference config.py:32-57)
    "multiviewx_160x250x8": dict(data="MultiviewX", n_cam=6, image_size=(1080, 1920), feat_image=(720, 1280),
                                 world_size=(640, 1000), cube_size=(4, 4, 8), grid_height=64),
    # C5 synthetic 8 cameras x 4K -> 512x512x32
    "synthetic4k_512x512x32": dict(data="MultiviewC", n_cam=8, image_size=(2160, 3840), feat_image=(2160, 3840),
                                   world_size=(3840, 3840), cube_size=(7.5, 7.5, 5), grid_height=160),
}


def _cameras_for(cfg):
    data, n = cfg["data"], cfg["n_cam"]
    H, W = cfg["image_size"]
    ws = cfg["world_size"]
    if data == "MultiviewC":
        # grid units == world cm; field centre, ring radius 0.72 x field, 6 m high, f = 900 px @ 1280
        cx, cy = ws[0] / 2.0, ws[1] / 2.0
        return ring_cameras(n, (cx, cy, 0.0), 0.72 * ws[0], 600.0, 900.0 * W / 1280.0, (W, H))
    if data == "Wildtrack":
        # world cm = grid*2.5 + (-300,-900); grid x spans world_size[0], y spans world_size[1]
        cx = ws[0] * 2.5 / 2.0 - 300.0
        cy = ws[1] * 2.5 / 2.0 - 900.0
        return ring_cameras(n, (cx, cy, 0.0), 0.45 * ws[1] * 2.5, 400.0, 1100.0 * W / 1920.0, (W, H))
    if data == "MultiviewX":
        # world m = grid/40
        cx, cy = ws[1] / 40.0 / 2.0, ws[0] / 40.0 / 2.0
        return ring_cameras(n, (cx, cy, 0.0), 0.8 * ws[1] / 40.0, 3.0, 1700.0 * W / 1920.0, (W, H))
    raise ValueError(data)


def make_workload(name, channels=256, seed=0, device="cpu", n_cam=None, cameras=None):
    """Build the synthetic inputs of one named workload.

    Returns a dict with ``args`` (namespace with .data/.image_size, what the projector reads from
    the reference's argparse bag), ``calibs`` (N,3,4), ``grid`` (1,L,W,3), ``features`` (list over
    cameras of the three (1,C,Hf,Wf) lateral maps, ``relu(randn)``), and the constructor kwargs.
    ``cameras``: only these cameras get feature maps (the others ``None``) -- a rank of a camera-sharded run needs
    its own cameras only; every camera has its own generator, so the maps do not depend on who builds them.
    """
    cfg = dict(WORKLOADS[name])
    if n_cam is not None:
        cfg["n_cam"] = n_cam
    args = SimpleNamespace(data=cfg["data"], image_size=tuple(cfg["image_size"]))
    cs = cfg["cube_size"]
    grid = make_grid(world_size=cfg["world_size"], cube_LW=cs[:2], dataset=cfg["data"]).unsqueeze(0)
    calibs = _cameras_for(cfg)
    sizes = feature_sizes(cfg["feat_image"])
    feats = []
    for cam in range(cfg["n_cam"]):
        if cameras is not None and cam not in cameras:
            feats.append(None)
            continue
        gen = torch.Generator().manual_seed(seed * 1000 + cam)
        feats.append([torch.relu(torch.randn(1, channels, h, w, generator=gen)).to(device) for (h, w) in sizes])
    return dict(name=name, args=args, calibs=calibs.to(device), grid=grid.to(device), features=feats,
                cube_size=cs, grid_height=cfg["grid_height"], n_cam=cfg["n_cam"], feat_sizes=sizes,
                channels=channels)
