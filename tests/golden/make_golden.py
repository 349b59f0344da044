#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE (CPU, fp32).

Runs only in the build container, where the reference lives at /root/reference.  Nothing from the
reference is copied: the script imports ``vfa.model.vfa_op.VFA`` / ``vfa.model.vfanet.VFANet``
(with empty stand-in modules for ``cv2``/``torchvision``, which the arithmetic never touches --
SURVEY.md Appendix B), feeds them seeded synthetic inputs and records inputs + outputs as ``.npz``.

Stage tensors are captured from inside ``VFA.forward`` (reference ``vfa/model/vfa_op.py:61-125``):
  * ``box``      : the grid argument of the four ``F.grid_sample`` calls (:112-115)
  * ``lt/rb/rt/lb`` are not stored; ``vox`` (input of ``collapse``, :123) is, via a forward-pre-hook
  * ``visible``  : the output of ``torch.logical_and`` (:106)
  * ``integral`` : ``VFA.integral_image(feature)`` (:172-173)
  * ``area``     : re-evaluated with the reference's own expression (:104-105) on the captured box
  * ``ortho``    : the module output (:125)
The VFANet fixture captures the lateral maps entering ``vfa8/16/32`` (vfanet.py:76-78) and the
summed ``ortho`` entering ``fuse`` (vfanet.py:131-134) from a real ``VFANet.forward``.

Usage:  python tests/golden/make_golden.py      (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

# --- stand-ins for modules the reference imports but the path never uses -------------------------
sys.modules["cv2"] = types.ModuleType("cv2")
_tv, _tvd, _tvv, _tvt = (types.ModuleType(n) for n in (
    "torchvision", "torchvision.datasets", "torchvision.datasets.vision", "torchvision.transforms"))


class _VisionDataset:
    def __init__(self, root=None, transform=None, **kw):
        self.root, self.transform = root, transform


_tvd.VisionDataset = _tvv.VisionDataset = _VisionDataset
_tvt.ToTensor = lambda: None
_tv.datasets, _tv.transforms = _tvd, _tvt
sys.modules.update({m.__name__: m for m in (_tv, _tvd, _tvv, _tvt)})
import matplotlib  # noqa: E402

matplotlib.use("agg")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import vfa.model.vfa_op as ref_op  # noqa: E402  (the reference)
from vfa.model.vfa_op import VFA as RefVFA  # noqa: E402
from vfa.utils import make_grid as ref_make_grid  # noqa: E402

from vfa_amd.synthetic import look_at_camera, ring_cameras  # noqa: E402

torch.set_num_threads(4)


def run_ref_vfa(vfa, feature, calib, grid):
    """One reference VFA.forward with stage capture."""
    cap = {"gs_grids": []}
    orig_gs, orig_land = ref_op.F.grid_sample, ref_op.torch.logical_and

    def gs(inp, g, *a, **k):
        cap["gs_grids"].append(g.detach().clone())
        return orig_gs(inp, g, *a, **k)

    def land(a, b, *r, **k):
        out = orig_land(a, b, *r, **k)
        cap["visible"] = out.detach().clone()
        return out

    h = vfa.collapse.register_forward_pre_hook(lambda m, i: cap.__setitem__("vox", i[0].detach().clone()))
    ref_op.F.grid_sample, ref_op.torch.logical_and = gs, land
    try:
        with torch.no_grad():
            ortho = vfa(feature, calib, grid)
    finally:
        ref_op.F.grid_sample, ref_op.torch.logical_and = orig_gs, orig_land
        h.remove()
    g_lt, g_rb, g_rt, g_lb = cap["gs_grids"]
    box = torch.cat([g_lt, g_rb], dim=-1)  # (1, nl, L*W, 4) = left, top, right, bottom
    assert torch.equal(g_rt, box[..., [2, 1]]) and torch.equal(g_lb, box[..., [0, 3]])
    Hf, Wf = feature.shape[2:]
    # the reference's own expression (vfa_op.py:104-105) on the captured box
    area = (((box[..., 2:] - box[..., :2]).prod(dim=-1)) * Hf * Wf + ref_op.EPSILON).unsqueeze(1)
    with torch.no_grad():
        integral = vfa.integral_image(feature)
    return dict(box=box[0].numpy(), area=area[0, 0].numpy(), visible=cap["visible"][0, 0].numpy(),
                integral=integral[0].numpy(), vox=cap["vox"].numpy(),
                ortho=ortho.contiguous()[0].numpy())


def vfa_case(fname, data, image_size, cube_size, grid_height, grid, calib, C, feat_hw, seed, signed=False):
    torch.manual_seed(seed)
    args = types.SimpleNamespace(data=data, image_size=tuple(image_size))
    vfa = RefVFA(channel=C, grid_height=grid_height, cube_size=cube_size, args=args).eval()
    feat = torch.randn(1, C, *feat_hw)
    if not signed:
        feat = torch.relu(feat)
    out = run_ref_vfa(vfa, feat, calib, grid)
    meta = dict(data=data, image_size=np.array(image_size), cube_size=np.array(cube_size, dtype=np.float64),
                grid_height=np.array(grid_height), seed=np.array(seed))
    np.savez_compressed(os.path.join(HERE, fname), feature=feat[0].numpy(), calib=calib.numpy(),
                        grid=grid[0].numpy(), weight=vfa.collapse.weight.detach().numpy(),
                        bias=vfa.collapse.bias.detach().numpy(),
                        z_corners=vfa.z_corners.numpy(), corners_offset=vfa.corners_offset.numpy(),
                        **meta, **out)
    vis = out["visible"].mean()
    print(f"{fname}: box {out['box'].shape} vox {out['vox'].shape} visible {vis:.2f} "
          f"|ortho|max {np.abs(out['ortho']).max():.4f}")


def make_grid_cases():
    """make_grid outputs for the three dataset kinds (reference vfa/utils.py:16-37)."""
    out = {}
    for name, ws, cl, off in [("MultiviewC", (3900, 3900), (25, 25), (0, 0, 0)),
                              ("MultiviewX", (640, 1000), (4, 4), (0, 0, 0)),
                              ("Wildtrack", (480, 1440), (4, 4), (0, 0, 0)),
                              ("Wildtrack", (100, 60), (7, 3), (1.5, -2.0, 0.25)),
                              ("MultiviewC", (110, 70), (9, 4), (0.5, 2.0, 1.0))]:
        g = ref_make_grid(world_size=ws, grid_offset=off, cube_LW=list(cl), dataset=name)
        key = f"{name}_{ws[0]}x{ws[1]}_{cl[0]}x{cl[1]}"
        out[key] = g.numpy()
        out[key + "_args"] = np.array(list(ws) + list(cl) + list(off), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "make_grid.npz"), **out)
    print("make_grid.npz:", {k: v.shape for k, v in out.items() if not k.endswith("_args")})


def vfanet_case(fname, data, image_size, world_size, cube_size, grid_height, cube_LW, calibs, img_hw, seed):
    """Real VFANet.forward on tiny images; capture laterals and the summed ortho (vfanet.py:64-82)."""
    from vfa.model.vfanet import VFANet as RefVFANet
    torch.manual_seed(seed)
    args = types.SimpleNamespace(data=data, image_size=tuple(image_size))
    net = RefVFANet(args, grid_height=grid_height, cube_size=cube_size, mode="2D", pretrained=False).eval()
    # default-init collapse weights give tiny outputs after 3 GroupNorm'd laterals; keep them as they are
    N = calibs.shape[0]
    images = torch.rand(N, 3, *img_hw)
    grid = ref_make_grid(world_size=world_size, cube_LW=list(cube_LW), dataset=data).unsqueeze(0)
    lats = {8: [], 16: [], 32: []}
    hooks = []
    for s in (8, 16, 32):
        hooks.append(getattr(net, f"vfa{s}").register_forward_pre_hook(
            lambda m, i, s=s: lats[s].append(i[0].detach().clone())))
    cap = {}
    hooks.append(net.fuse.register_forward_pre_hook(lambda m, i: cap.__setitem__("ortho", i[0].detach().clone())))
    with torch.no_grad():
        net(images, calibs, grid)
    for h in hooks:
        h.remove()
    sd = {}
    for s in (8, 16, 32):
        m = getattr(net, f"vfa{s}")
        sd[f"weight{s}"] = m.collapse.weight.detach().numpy()
        sd[f"bias{s}"] = m.collapse.bias.detach().numpy()
        sd[f"lat{s}"] = torch.cat(lats[s], 0).numpy()
    np.savez_compressed(os.path.join(HERE, fname), data=data, image_size=np.array(image_size),
                        cube_size=np.array(cube_size, dtype=np.float64), grid_height=np.array(grid_height),
                        calibs=calibs.numpy(), grid=grid[0].numpy(), ortho=cap["ortho"].contiguous()[0].numpy(),
                        seed=np.array(seed), **sd)
    print(f"{fname}: lat8 {sd['lat8'].shape} ortho {cap['ortho'].shape} max {cap['ortho'].abs().max():.4f}")


def main():
    make_grid_cases()

    # ---- MultiviewC (grid units = cm), coarse grids spanning the 39 m field --------------------
    g = ref_make_grid((3900, 3900), cube_LW=[300, 260], dataset="MultiviewC").unsqueeze(0)  # (1,15,13,3)
    cams = ring_cameras(3, (1950.0, 1950.0, 0.0), 2800.0, 600.0, 900.0, (1280, 720))
    vfa_case("mc_cam0_s32.npz", "MultiviewC", (720, 1280), (25, 25, 32), 160, g, cams[0], 8, (23, 40), 1)
    vfa_case("mc_cam1_s16.npz", "MultiviewC", (720, 1280), (25, 25, 32), 160, g, cams[1], 8, (45, 80), 2)
    # camera INSIDE the field, low: cells behind the camera (h2 < 0) and huge near boxes
    inside = torch.tensor(look_at_camera((1500.0, 1700.0, 250.0), (2600.0, 2300.0, 0.0), 700.0, (1280, 720)),
                          dtype=torch.float32)
    g2 = ref_make_grid((3900, 3900), cube_LW=[150, 175], dataset="MultiviewC").unsqueeze(0)  # (1,23,26,3)
    vfa_case("mc_inside_s8.npz", "MultiviewC", (720, 1280), (25, 25, 32), 160, g2, inside, 4, (90, 160), 3)
    # signed features (not post-ReLU), big cubes -> many multi-pixel boxes, odd feature size
    vfa_case("mc_signed_bigcube.npz", "MultiviewC", (720, 1280), (300, 260, 90), 400, g, cams[2], 8, (31, 37), 4,
             signed=True)
    # dense patch of the shipped grid (25 cm cells) near the field centre: sub-pixel boxes at stride 32
    gd = ref_make_grid((3900, 3900), cube_LW=[25, 25], dataset="MultiviewC")[60:84, 70:90].unsqueeze(0).contiguous()
    vfa_case("mc_dense_s32.npz", "MultiviewC", (720, 1280), (25, 25, 32), 160, gd, cams[0], 8, (23, 40), 5)
    # single layer, 200x200-style cubes (BASELINE config 2 geometry, coarse)
    g3 = ref_make_grid((3750, 3750), cube_LW=[250, 375], dataset="MultiviewC").unsqueeze(0)
    vfa_case("mc_nl1.npz", "MultiviewC", (720, 1280), (18.75, 18.75, 160), 160, g3, cams[1], 8, (45, 80), 6)

    # ---- Wildtrack (world = grid*2.5 - (300, 900) cm) ------------------------------------------
    gw = ref_make_grid((480, 1440), cube_LW=[32, 60], dataset="Wildtrack").unsqueeze(0)  # (1,15,24,3)
    wt_c = (480 * 2.5 / 2 - 300.0, 1440 * 2.5 / 2 - 900.0, 0.0)
    wcams = ring_cameras(3, wt_c, 0.45 * 1440 * 2.5, 400.0, 1100.0, (1920, 1080))
    vfa_case("wt_cam0_s8.npz", "Wildtrack", (1080, 1920), (4, 4, 4), 32, gw, wcams[0], 4, (90, 160), 11)
    vfa_case("wt_cam1_s32.npz", "Wildtrack", (1080, 1920), (4, 4, 4), 32, gw, wcams[1], 8, (23, 40), 12)
    wt_side = torch.tensor(look_at_camera((-700.0, 900.0, 400.0), (300.0, 900.0, 0.0), 1100.0, (1920, 1080)),
                           dtype=torch.float32)
    gw2 = ref_make_grid((480, 1440), cube_LW=[4, 4], dataset="Wildtrack")[40:57, 150:173].unsqueeze(0).contiguous()
    vfa_case("wt_side_dense_s16.npz", "Wildtrack", (1080, 1920), (4, 4, 4), 32, gw2, wt_side, 8, (45, 80), 13)

    # ---- MultiviewX (world = grid/40 m) --------------------------------------------------------
    gx = ref_make_grid((640, 1000), cube_LW=[50, 40], dataset="MultiviewX").unsqueeze(0)  # (1,16,20,3)?
    mx_cam = torch.tensor(look_at_camera((-5.0, 8.0, 3.0), (12.0, 8.0, 0.0), 1700.0, (1920, 1080)),
                          dtype=torch.float32)
    vfa_case("mx_cam0_s16.npz", "MultiviewX", (1080, 1920), (4, 4, 8), 64, gx, mx_cam, 8, (45, 80), 21)
    mx_cam2 = torch.tensor(look_at_camera((30.0, 20.0, 2.5), (12.0, 6.0, 0.0), 1400.0, (1920, 1080)),
                           dtype=torch.float32)
    vfa_case("mx_cam1_s8.npz", "MultiviewX", (1080, 1920), (4, 4, 8), 64, gx, mx_cam2, 4, (90, 160), 22)

    # ---- VFANet loop: 3 cameras x 3 scales, summed ---------------------------------------------
    vfanet_case("vfanet_mc.npz", "MultiviewC", (720, 1280), (3900, 3900), (25, 25, 32), 160, (390, 325),
                ring_cameras(3, (1950.0, 1950.0, 0.0), 2800.0, 600.0, 900.0, (1280, 720)), (96, 160), 31)
    vfanet_case("vfanet_wt.npz", "Wildtrack", (1080, 1920), (480, 1440), (4, 4, 4), 8, (60, 120),
                ring_cameras(2, wt_c, 0.45 * 1440 * 2.5, 400.0, 1100.0, (1920, 1080)), (72, 128), 32)


def main_nl1():
    """Round-2 additions: single-layer grids at C = 256 (K = N = 256), the shape of BASELINE configs[1] / configs[2] --
    the reference's own ``VFANet.forward`` output for the flagship MFMA collapse kernels."""
    vfanet_case("vfanet_mc_nl1.npz", "MultiviewC", (720, 1280), (3750, 3750), (18.75, 18.75, 160), 160, (150, 125),
                ring_cameras(3, (1875.0, 1875.0, 0.0), 2700.0, 600.0, 900.0, (1280, 720)), (96, 160), 41)
    wt_c = (480 * 2.5 / 2 - 300.0, 1440 * 2.5 / 2 - 900.0, 0.0)
    vfanet_case("vfanet_wt_nl1.npz", "Wildtrack", (1080, 1920), (480, 1440), (1, 1, 4), 4, (40, 90),
                ring_cameras(2, wt_c, 0.45 * 1440 * 2.5, 400.0, 1100.0, (1920, 1080)), (72, 128), 42)


def main_decode():
    """The step AFTER the path: the reference's own BEV decode (vfa/data/encoder.py:230-305) on random head outputs.  The
    encoder only reads a few attributes of its dataset object, so a stand-in namespace is enough (no images, no labels)."""
    from vfa.data.encoder import ObjectEncoder
    for fname, base, ws, cube, shape, seed in (("decode_mc.npz", "MultiviewC", (3900, 3900), (25, 25, 32), (24, 30), 51),
                                                ("decode_wt.npz", "Wildtrack", (480, 1440), (4, 4, 4), (30, 90), 52),
                                                ("decode_mx.npz", "MultiviewX", (640, 1000), (4, 4, 8), (40, 63), 53)):
        torch.manual_seed(seed)
        mean = np.array([140.0, 60.0, 230.0], dtype=np.float32)
        base_ns = types.SimpleNamespace(label_names=["Cow"], __name__=base)
        ds = types.SimpleNamespace(base=base_ns, world_size=ws, cube_LWH=cube,
                                   classAverage=types.SimpleNamespace(get_mean=lambda name: mean))
        enc = ObjectEncoder(ds, topk=100)
        L, W = shape
        heat = torch.randn(1, 1, L, W) * 2.0 - 1.0
        heat[0, 0, 3:6, 4:9] = 1.5          # a plateau: ties inside a 5 x 5 window
        heat[0, 0, 0, 0] = 6.0              # a corner peak
        pred = {"heatmap": heat, "loc_offset": torch.randn(1, L, W, 2), "dim_offset": torch.randn(1, L, W, 3) * 0.2,
                "rotation": torch.randn(1, L, W, 360)}
        out = {}
        if base == "MultiviewC":
            d = enc.decode3d(pred, 0.4)
            out = {k: d[k].numpy() for k in ("conf", "location", "dimension", "rotation")}
        else:
            d = enc.decode2d(pred, 0.4)
            out = {k: d[k].numpy() for k in ("conf", "location")}
        with torch.no_grad():
            nms = enc.nms(torch.sigmoid(heat))
        np.savez_compressed(os.path.join(HERE, fname), base=base, world_size=np.array(ws), cube_LWH=np.array(cube),
                            dimension_mean=mean, heatmap=heat.numpy(), loc_offset=pred["loc_offset"].numpy(),
                            nms=nms.numpy(), **({"dim_offset": pred["dim_offset"].numpy(), "rotation_logits": pred["rotation"].numpy()}
                                                if base == "MultiviewC" else {}), **{"out_" + k: v for k, v in out.items()})
        print(f"{fname}: {len(out['conf'])} detections above 0.4, nms peaks {int((nms > 0).sum())}")


def _clip_area(c1, c2):
    """Independent truth for the overlap of two convex quadrilaterals: Sutherland-Hodgman clipping + shoelace in float64.
    Shares nothing with the reference's candidate-vertex / angular-sort scheme."""
    def ccw(poly):
        a = sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[i][1] * poly[(i + 1) % len(poly)][0] for i in range(len(poly)))
        return poly if a > 0 else poly[::-1]
    subject, clip = ccw([tuple(map(float, p)) for p in c1]), ccw([tuple(map(float, p)) for p in c2])
    for i in range(len(clip)):
        a, b = clip[i], clip[(i + 1) % len(clip)]
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        out = []
        for j in range(len(subject)):
            p, q = subject[j], subject[(j + 1) % len(subject)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        subject = out
        if not subject:
            return 0.0
    return abs(sum(subject[i][0] * subject[(i + 1) % len(subject)][1] - subject[i][1] * subject[(i + 1) % len(subject)][0]
                   for i in range(len(subject)))) / 2


def main_iou():
    """Pins the ``sort_vertices`` restatement through the reference's only call site of the kernel, the rotated-box IoU
    (vfa/evaluation/pyeval/IoU.py:139-198).  The reference's CUDA extension module ``sort_vertices`` cannot be built here
    (nvcc), so it is the ONE piece stood in for -- by oracle/eval_oracle.sort_vertices, the thing under test; every other step
    (boxes2corners, boxes_intersection, box1_in_box2, build_vertices, the mean normalisation, calculate_area, IoUs2D) is the
    reference's own torch code, run on CPU.  If the restatement ordered the vertices differently from what the reference's
    pipeline needs, the shoelace area it feeds would not be the overlap: the script checks every case against an independent
    float64 polygon clipper and stores vertices, masks, the index lists and both areas."""
    from oracle import eval_oracle
    captured = []
    stub = types.ModuleType("sort_vertices")

    def sort_vertices_forward(vertices, mask, num_valid):
        idx = eval_oracle.sort_vertices(vertices.numpy(), mask.numpy(), num_valid.numpy())
        captured.append((vertices.numpy().copy(), mask.numpy().copy(), num_valid.numpy().copy(), idx.copy()))
        return torch.from_numpy(idx)
    stub.sort_vertices_forward = sort_vertices_forward
    sys.modules["sort_vertices"] = stub
    from vfa.evaluation.pyeval import IoU as ref_iou

    rng = np.random.default_rng(61)
    pairs, kinds = [], []

    def add(kind, b1, b2):
        pairs.append((np.asarray(b1, np.float32), np.asarray(b2, np.float32)))
        kinds.append(kind)
    for _ in range(192):      # generic rotated pairs, most of them overlapping
        b1 = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.5, 3), rng.uniform(0.5, 3), rng.uniform(-np.pi, np.pi)]
        b2 = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.5, 3), rng.uniform(0.5, 3), rng.uniform(-np.pi, np.pi)]
        add("random", b1, b2)
    for _ in range(16):       # identical boxes: 8 valid corners, the kernel's duplicate rule (sort_vert_kernel.cu:110-129)
        b = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.5, 3), rng.uniform(0.5, 3), rng.uniform(0.1, 1.4)]
        add("identical", b, b)
    for _ in range(16):       # one box strictly inside the other: 4 valid corners, no intersections
        b = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(2, 3), rng.uniform(2, 3), rng.uniform(0.1, 1.4)]
        add("contained", b, [b[0] + 0.1, b[1] - 0.1, 0.6, 0.5, rng.uniform(-np.pi, np.pi)])
    for _ in range(16):       # far apart: no valid vertex at all (num_valid = 0, everything padding)
        b = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.5, 1), rng.uniform(0.5, 1), rng.uniform(-np.pi, np.pi)]
        add("disjoint", b, [b[0] + 5, b[1] + 5, 1, 1, rng.uniform(-np.pi, np.pi)])
    for _ in range(16):       # MultiviewC-sized cows in centimetres (the evaluation's real magnitudes)
        b1 = [rng.uniform(500, 3400), rng.uniform(500, 3400), rng.uniform(180, 260), rng.uniform(60, 110), rng.uniform(-np.pi, np.pi)]
        b2 = [b1[0] + rng.uniform(-60, 60), b1[1] + rng.uniform(-60, 60), rng.uniform(180, 260), rng.uniform(60, 110),
              b1[4] + rng.uniform(-0.5, 0.5)]
        add("cows_cm", b1, b2)

    rows = []
    for kind, (b1, b2) in zip(kinds, pairs):
        t1, t2 = torch.from_numpy(b1).view(1, 1, 5), torch.from_numpy(b2).view(1, 1, 5)
        with np.errstate(all="ignore"):
            iou, c1, c2, union = ref_iou.IoUs2D(t1, t2)
        verts, mask, nv, idx = captured[-1]
        # the un-normalised vertices calculate_area reads (IoU.py:189-191)
        inters, mi = ref_iou.boxes_intersection(c1, c2)
        raw, _ = ref_iou.build_vertices(c1, c2, inters, ref_iou.box1_in_box2(c1, c2), ref_iou.box1_in_box2(c2, c1), mi)
        overlap = float(b1[2] * b1[3] + b2[2] * b2[3] - float(union))
        truth = _clip_area(c1[0, 0].numpy().astype(np.float64), c2[0, 0].numpy().astype(np.float64))
        rows.append((kind, b1, b2, verts[0, 0], mask[0, 0], int(nv[0, 0]), idx[0, 0], raw[0, 0].numpy(), overlap, truth, float(iou)))
    scale = np.array([max(r[1][2] * r[1][3], r[2][2] * r[2][3]) for r in rows])
    err = np.abs(np.array([r[8] for r in rows]) - np.array([r[9] for r in rows])) / scale
    for kind in sorted(set(kinds)):
        sel = np.array([k == kind for k in kinds])
        print(f"iou_pairs {kind:10s}: {sel.sum():3d} pairs, max |overlap - clipped| / box area = {err[sel].max():.2e}, "
              f"vertex counts {sorted(set(r[5] for r, s in zip(rows, sel) if s))}")
    np.savez_compressed(os.path.join(HERE, "iou_pairs.npz"), kind=np.array(kinds), box1=np.stack([r[1] for r in rows]),
                        box2=np.stack([r[2] for r in rows]), vertices=np.stack([r[3] for r in rows]),
                        mask=np.stack([r[4] for r in rows]), num_valid=np.array([r[5] for r in rows], np.int32),
                        idx=np.stack([r[6] for r in rows]).astype(np.int32), raw_vertices=np.stack([r[7] for r in rows]),
                        overlap=np.array([r[8] for r in rows]), clipped=np.array([r[9] for r in rows]),
                        iou=np.array([r[10] for r in rows]), rel_err=err)


if __name__ == "__main__":
    if "--decode" in sys.argv:
        main_decode()
    elif "--nl1" in sys.argv:
        main_nl1()
    elif "--iou" in sys.argv:
        main_iou()
    else:
        main()
        main_nl1()
        main_decode()
        main_iou()
