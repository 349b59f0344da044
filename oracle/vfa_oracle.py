"""ctypes front-end of the CPU oracle (oracle/vfa_oracle.c) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  It restates the reference's ``VFA.forward`` (``/root/reference/vfa/model/vfa_op.py:61-125``)
and the ``VFANet`` scale/view sums (``vfa/model/vfanet.py:64-82``) on numpy arrays in the
reference's own layouts (NCHW integral image, ``vox[cell, c*nl + layer]``).

Parity pinning: see the header of vfa_oracle.c -- pinned bitwise (pre-GEMM) against fixtures generated
from the reference by tests/golden/make_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvfa_oracle.so")
_lib = None

CONV_KIND = {"MultiviewC": 0, "MultiviewX": 1, "Wildtrack": 2}


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "vfa_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvfa_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def integral_image(feature):
    """feature (C,H,W) -> integral (C,H,W).  reference vfa_op.py:172-173."""
    f = _f32(feature)
    C, H, W = f.shape
    out = np.empty_like(f)
    lib().vfa_oracle_integral_image(_fp(f), _fp(out), C, H, W)
    return out


def z_layers_of(grid_height, cube_size):
    """Heights of the z-layers: arange(0, grid_height, cube_h) (reference vfa_op.py:50)."""
    return np.arange(0, grid_height, cube_size[2]).astype(np.float32)


def corner_offsets(cube_size):
    """The 8 cube-corner offsets in the reference's ``generate_cube`` order (vfa_op.py:127-133)."""
    l, w, h = (float(v) for v in cube_size)
    x = [-l / 2, l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2]
    y = [-w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2, w / 2]
    z = [0, 0, 0, 0, h, h, h, h]
    return np.stack([x, y, z], axis=1).astype(np.float32)


def box_params(calib, grid, z_layers, corner_off, data, image_size, Hf, Wf, crange=(-1, 0.95)):
    """-> box (nl, n_cells, 4), area (nl, n_cells), visible (nl, n_cells) bool.  vfa_op.py:64-106."""
    calib = _f32(calib).reshape(12)
    g = _f32(grid).reshape(-1, 3)
    zl = _f32(z_layers).reshape(-1)
    co = _f32(corner_off).reshape(8, 3)
    n_cells, nl = g.shape[0], zl.shape[0]
    box = np.empty((nl, n_cells, 4), np.float32)
    area = np.empty((nl, n_cells), np.float32)
    vis = np.empty((nl, n_cells), np.uint8)
    img_h, img_w = image_size  # args.image_size = (H, W); the path uses image_size[::-1]
    lib().vfa_oracle_box_params(_fp(calib), _fp(g), n_cells, _fp(zl), nl, _fp(co), CONV_KIND[data],
                                ctypes.c_float(img_w), ctypes.c_float(img_h), Hf, Wf,
                                ctypes.c_float(crange[0]), ctypes.c_float(crange[1]), _fp(box), _fp(area), _u8(vis))
    return box, area, vis.astype(bool)


def gather(integral, box, area, visible):
    """integral (C,Hf,Wf), box (nl,n,4) -> vox (n, C*nl), column = c*nl + layer.  vfa_op.py:112-120."""
    I = _f32(integral)
    C, Hf, Wf = I.shape
    box = _f32(box)
    nl, n_cells = box.shape[:2]
    area = _f32(area)
    vis = np.ascontiguousarray(visible, dtype=np.uint8)
    vox = np.empty((n_cells, C * nl), np.float32)
    lib().vfa_oracle_gather(_fp(I), _fp(box), _fp(area), _u8(vis), C, Hf, Wf, nl, n_cells, _fp(vox))
    return vox


def lateral(feat, weight, bias, gamma, beta, eps=1e-5, groups=16):
    """The lateral branch in front of the path, restated in float64 numpy: relu(GroupNorm(conv1x1(feat)))   (reference
    vfa/model/vfanet.py:37-42, 72-74: nn.Conv2d(K, 256, 1), nn.GroupNorm(16, 256), F.relu).  feat (K, h, w), weight (256, K)
    -> (y (256, h, w) = the convolution, lat (256, h, w) = the normalised, rectified map), both float64.  Pinned by
    tests/golden/laterals_*.npz (a real VFANet.forward of the reference)."""
    f = np.asarray(feat, np.float64)
    K, h, w = f.shape
    y = np.asarray(weight, np.float64).reshape(-1, K) @ f.reshape(K, h * w) + np.asarray(bias, np.float64)[:, None]
    co = y.shape[0]
    g = y.reshape(groups, -1)
    mean, var = g.mean(1, keepdims=True), g.var(1, keepdims=True)  # biased, like nn.GroupNorm
    norm = ((g - mean) / np.sqrt(var + eps)).reshape(co, h * w)
    out = norm * np.asarray(gamma, np.float64)[:, None] + np.asarray(beta, np.float64)[:, None]
    return y.reshape(co, h, w), np.maximum(out, 0.0).reshape(co, h, w)


def collapse_relu(vox, weight, bias):
    """relu(vox @ weight.T + bias) -> (M, N).  vfa_op.py:123-125."""
    vox, weight, bias = _f32(vox), _f32(weight), _f32(bias)
    M, K = vox.shape
    N = weight.shape[0]
    out = np.empty((M, N), np.float32)
    scratch = np.empty((K, N), np.float32)
    lib().vfa_oracle_collapse_relu(_fp(vox), _fp(weight), _fp(bias), M, K, N, _fp(scratch), _fp(out))
    return out


def vfa_forward(feature, calib, grid, weight, bias, data, image_size, cube_size, grid_height,
                crange=(-1, 0.95), stages=False):
    """Whole ``VFA.forward``: feature (C,Hf,Wf), calib (3,4), grid (L,W,3) -> ortho (C_out, L, W)."""
    feature = _f32(feature)
    C, Hf, Wf = feature.shape
    L, W = grid.shape[:2]
    zl = z_layers_of(grid_height, cube_size)
    co = corner_offsets(cube_size)
    I = integral_image(feature)
    box, area, vis = box_params(calib, grid, zl, co, data, image_size, Hf, Wf, crange)
    vox = gather(I, box, area, vis)
    mn = collapse_relu(vox, weight, bias)
    N = mn.shape[1]
    ortho = np.empty((N, L * W), np.float32)
    lib().vfa_oracle_to_nchw(_fp(mn), L * W, N, _fp(ortho))
    ortho = ortho.reshape(N, L, W)
    if stages:
        return dict(integral=I, box=box, area=area, visible=vis, vox=vox, ortho=ortho)
    return ortho


def aggregate(per_camera_scale_maps):
    """``ortho += (f8 + f16) + f32`` over cameras in order.  vfanet.py:79, 82."""
    ortho = None
    for f8, f16, f32 in per_camera_scale_maps:
        f8, f16, f32 = _f32(f8), _f32(f16), _f32(f32)
        if ortho is None:
            ortho = np.zeros_like(f8)
        lib().vfa_oracle_accumulate(_fp(ortho), _fp(f8), _fp(f16), _fp(f32), ctypes.c_size_t(ortho.size))
    return ortho


def vfanet_aggregate(lats, calibs, grid, weights, biases, data, image_size, cube_size, grid_height):
    """The camera loop of ``VFANet.forward`` (vfanet.py:64-82) given the lateral maps.

    lats: dict {8,16,32} -> (N_cam, C, h, w); weights/biases: dict {8,16,32}.
    """
    n_cam = calibs.shape[0]
    maps = []
    for cam in range(n_cam):
        maps.append(tuple(
            vfa_forward(lats[s][cam], calibs[cam], grid, weights[s], biases[s], data, image_size, cube_size,
                        grid_height) for s in (8, 16, 32)))
    return aggregate(maps)


def make_grid(world_size=(3900, 3900), grid_offset=(0, 0, 0), cube_LW=(25, 25), dataset="Wildtrack"):
    """numpy restatement of the reference's make_grid (vfa/utils.py:16-37)."""
    if dataset == "Wildtrack":
        length, width = world_size[::-1]
    else:
        length, width = world_size
    xoff, yoff, zoff = grid_offset
    # torch.arange(0., end, step) for fp32: start + i*step evaluated in double then cast
    nx = int(np.ceil(width / cube_LW[0]))
    ny = int(np.ceil(length / cube_LW[1]))
    xs = (np.arange(nx, dtype=np.float64) * cube_LW[0]).astype(np.float32) + np.float32(xoff)
    ys = (np.arange(ny, dtype=np.float64) * cube_LW[1]).astype(np.float32) + np.float32(yoff)
    if dataset == "Wildtrack":
        xx, yy = np.meshgrid(xs, ys, indexing="ij")
    else:
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
    return np.stack([xx, yy, np.full_like(xx, np.float32(zoff))], axis=-1).astype(np.float32)
