"""``VFANet`` -- the caller of the projector, with the reference's interface (``vfa/model/vfanet.py:14-149``).

Only the camera loop (:64-82) is the hot path; it is replaced by one batched ``aggregate_views`` call (HIP
kernels).  Backbone, laterals and BEV heads are stock convolutions / GroupNorm and stay PyTorch (MIOpen on ROCm):
they are laid out with the reference's sub-module names so that its checkpoints (``model_state_dict``) load
key-for-key.  GroupNorm is per sample, so running the laterals on all cameras at once is numerically the
per-camera computation of the reference.

Multi-GPU (``distributed=True``): each rank runs backbone + laterals + projection for ITS cameras
(``camera_shard``) and the partial BEV maps are summed with one RCCL all-reduce; heads run replicated.
"""
import glob
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, vfa_op
from .aggregate import aggregate_views, camera_shard
from .vfa_op import VFA

# Producer fusion, inference on the fused frame path: the GroupNorm affine + ReLU of the lateral branch
# is applied inside the integral-image row scan (`vfa_affine_relu_integral_image_f32`); the lateral maps are never written.
FUSE_PRODUCER = os.environ.get("VFA_AMD_FUSE_PRODUCER", "1") == "1"

# ``pretrained=True`` (the default of the reference's train.py:90): the reference downloads the torchvision ImageNet
# checkpoint (resnet.py:155-159, 170-172).  There is no network on the MI355X boxes, so the file is looked up locally.
PRETRAINED_ENV = "VFA_AMD_PRETRAINED"


def _find_pretrained(base):
    cand = [os.environ.get(PRETRAINED_ENV)] if os.environ.get(PRETRAINED_ENV) else []
    hub = os.path.join(os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch")), "hub",
                       "checkpoints")
    cand += sorted(glob.glob(os.path.join(hub, f"{base}-*.pth")))
    for path in cand:
        if path and os.path.isdir(path):
            hits = sorted(glob.glob(os.path.join(path, f"{base}*.pth")))
            path = hits[0] if hits else None
        if path and os.path.isfile(path):
            return path
    return None


def load_pretrained_trunk(trunk, base):
    """Reference ``_load_pretrained`` (resnet.py:170-175): copy every checkpoint entry whose key exists in the trunk
    (conv weights and the affine parameters the GroupNorm layers share with torchvision's BatchNorm by NAME; running
    statistics have no counterpart).  Returns the number of tensors loaded.  The reference downloads the checkpoint; a box
    without network cannot, and training from scratch where ImageNet weights were asked for must not happen silently: without
    a local checkpoint this RAISES, unless random initialisation is chosen explicitly with ``VFA_AMD_PRETRAINED=none``."""
    if os.environ.get(PRETRAINED_ENV, "").strip().lower() == "none":
        warnings.warn(f"VFANet(pretrained=True) with {PRETRAINED_ENV}=none: the trunk keeps its random initialisation")
        return 0
    path = _find_pretrained(base)
    if path is None:
        raise FileNotFoundError(
            f"VFANet(pretrained=True): no local {base} ImageNet checkpoint and no network to download it like the reference "
            f"does (resnet.py:159).  Set {PRETRAINED_ENV} to the .pth file or its directory, or place it in the torch hub cache; "
            f"{PRETRAINED_ENV}=none starts from random weights on purpose")
    ckpt = torch.load(path, map_location="cpu")
    own = trunk.state_dict()
    hit = {k: v for k, v in ckpt.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    own.update(hit)
    trunk.load_state_dict(own)
    print(f"VFANet: loaded {len(hit)} of {len(own)} trunk tensors from {path}")
    return len(hit)


def _gn(ch):
    return nn.GroupNorm(16, ch)


class _Block(nn.Module):
    """Two 3x3 convs with GroupNorm(16) and an identity / 1x1 projection shortcut (reference resnet.py:26-57)."""
    expansion = 1

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = _gn(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = _gn(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), _gn(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)), inplace=True)


class _Trunk(nn.Module):
    """ResNet-18/34 trunk returning the stride-8/16/32 maps (reference resnet.py:95-147)."""

    def __init__(self, depths):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = _gn(64)
        widths, cin = (64, 128, 256, 512), 64
        for i, (w, d) in enumerate(zip(widths, depths)):
            blocks = [_Block(cin, w, 1 if i == 0 else 2)] + [_Block(w, w) for _ in range(d - 1)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
            cin = w
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x)), inplace=True), 3, stride=2, padding=1)
        f8 = self.layer2(self.layer1(x))
        f16 = self.layer3(f8)
        return f8, f16, self.layer4(f16)


_DEPTHS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


def _head(cout):
    return nn.Sequential(nn.Conv2d(256, 256, 3, padding=1), _gn(256), nn.ReLU(True),
                         nn.Conv2d(256, cout, 3, padding=1, bias=False))


class VFANet(nn.Module):
    def __init__(self, args, base="resnet18", grid_height=160, cube_size=(25, 25, 32), angle_range=360, mode="3D",
                 pretrained=False):
        super().__init__()
        assert base in _DEPTHS, f"Unrecognized model, expect `resnet18` or `resnet34`, got {base}."
        assert mode in ("2D", "3D"), f"mode error, expect `2D` or `3D`, got{mode}"
        self.mode = mode
        self.base = _Trunk(_DEPTHS[base])
        if pretrained:
            load_pretrained_trunk(self.base, base)
        for s in (8, 16, 32):
            setattr(self, f"vfa{s}", VFA(channel=256, grid_height=grid_height, cube_size=cube_size, feat_scale=1. / s,
                                         args=args))
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]))
        self.lat8, self.lat16, self.lat32 = nn.Conv2d(128, 256, 1), nn.Conv2d(256, 256, 1), nn.Conv2d(512, 256, 1)
        self.bn8, self.bn16, self.bn32 = _gn(256), _gn(256), _gn(256)
        self.fuse = nn.Sequential(nn.Conv2d(256, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU(True),
                                  nn.Conv2d(256, 256, 3, padding=2, dilation=2), nn.BatchNorm2d(256), nn.ReLU(True))
        self.map_classifier = nn.Sequential(nn.Conv2d(256, 1, 3, padding=4, dilation=4, bias=False))
        self.tytx_pred = _head(2)
        if mode == "3D":
            self.orient_pred = nn.Sequential(nn.Conv2d(256, angle_range, 3, padding=4, dilation=4, bias=False))
            self.thtwtl_pred = _head(3)

    def laterals(self, images):
        """images (n,3,iH,iW) -> the three (n,256,h,w) non-negative lateral maps (reference vfanet.py:59-62, 72-74)."""
        x = (images - self.mean.view(3, 1, 1)) / self.std.view(3, 1, 1)
        f8, f16, f32 = self.base(x)
        return (F.relu(self.bn8(self.lat8(f8))), F.relu(self.bn16(self.lat16(f16))), F.relu(self.bn32(self.lat32(f32))))

    def lateral_integrals(self, images):
        """images (n,3,iH,iW) -> the three zero-bordered channels-last integral images of the lateral maps, without
        materialising those maps (reference vfanet.py:72-74 + vfa_op.py:110, 172-173).  ONE hand-written kernel for the three
        1x1 convolutions (``ops.lateral_convs``: six bf16 MFMA products of a three-piece split, channels-last output, GroupNorm
        statistics in its epilogue) + a tiny statistics kernel; then one launch pair for the three integral images, whose row scan applies the
        GroupNorm affine and the ReLU (``vfa_integral_images_hwc_f32``).  No NCHW lateral tensor exists."""
        x = (images - self.mean.view(3, 1, 1)) / self.std.view(3, 1, 1)
        parts = ops.lateral_convs([(feat, conv.weight, conv.bias, gn.weight, gn.bias, gn.eps) for feat, conv, gn in
                                   zip(self.base(x), (self.lat8, self.lat16, self.lat32), (self.bn8, self.bn16, self.bn32))])
        return ops.integral_images([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts], channels_last=True)

    def ortho_features(self, images, calibs, grid, distributed=False):
        """The fused BEV map (1,256,L,W) entering the heads (reference vfanet.py:64-82, 131)."""
        if distributed:
            mine = camera_shard(images.shape[0])
            idx = torch.tensor(mine, dtype=torch.long, device=images.device)
            images, calibs = images[idx], calibs[idx]
        mods3 = [self.vfa8, self.vfa16, self.vfa32]
        if (FUSE_PRODUCER and not torch.is_grad_enabled() and images.is_cuda and images.shape[0]
                and (vfa_op.fused_frame_ok(mods3, images.shape[0]) or vfa_op.pipe_frame_ok(mods3, images.shape[0]))):
            return aggregate_views(self.vfa8, self.vfa16, self.vfa32, None, None, None, calibs, grid, (-1, 0.95),
                                   distributed=distributed, integrals=self.lateral_integrals(images))
        lat8, lat16, lat32 = self.laterals(images) if images.shape[0] else (images.new_zeros(0, 256, 1, 1),) * 3
        return aggregate_views(self.vfa8, self.vfa16, self.vfa32, lat8, lat16, lat32, calibs, grid, (-1, 0.95),
                               distributed=distributed)

    def forward(self, images, calibs, grid, visualize=False, visualize_ortho=False, distributed=False):
        """images (N,3,iH,iW), calibs (N,3,4), grid (1,L,W,3) -> dict like the reference (vfanet.py:141-149)."""
        if visualize or visualize_ortho:
            self._visualize(images, calibs, grid, boxes=visualize_ortho)
        topdown = self.ortho_features(images, calibs, grid, distributed)
        return self.heads(topdown)

    def _visualize(self, images, calibs, grid, boxes=False):
        """Slow matplotlib side path (reference vfanet.py:84-128): per camera, the norms of the three lateral maps, of
        that camera's BEV contribution and of the running fused map; ``boxes`` also draws the projected cubes
        (``VFA.visualize_cube``, reference vfa_op.py:90-101)."""
        import matplotlib.pyplot as plt
        with torch.no_grad():
            lats = self.laterals(images)
            fused = 0
            for cam in range(images.shape[0]):
                one = [l[cam:cam + 1] for l in lats]
                part = aggregate_views(self.vfa8, self.vfa16, self.vfa32, *one, calibs[cam:cam + 1], grid)
                fused = fused + part
                fig, axes = plt.subplots(1, 5, figsize=(15, 3))
                for ax, t, title in zip(axes, one + [part, fused], ("feat8", "feat16", "feat32", "ortho", "fused ortho")):
                    ax.imshow(torch.norm(t, dim=1)[0].cpu().numpy())
                    ax.set_title(f"C{cam + 1} {title}")
                    ax.axis("off")
                plt.show()
                plt.close(fig)
                if boxes:
                    self.vfa8.visualize_cube(one[0], calibs[cam], grid)

    def heads(self, topdown):
        """The BEV heads on the fused map (reference vfanet.py:131-149)."""
        fused = self.fuse(topdown)
        out = {"heatmap": self.map_classifier(fused), "loc_offset": self.tytx_pred(topdown).permute(0, 2, 3, 1)}
        if self.mode == "3D":
            out["dim_offset"] = self.thtwtl_pred(topdown).permute(0, 2, 3, 1)
            out["rotation"] = self.orient_pred(fused).permute(0, 2, 3, 1)
        return out
