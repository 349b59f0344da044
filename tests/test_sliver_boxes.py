"""Noise-dominated sliver boxes through the FUSED frame kernels at full size.   ``-m gpu``.

The reference keeps every box with ``area > 1e-6`` visible (vfa/model/vfa_op.py:104-106) and divides the box sum by that area
(:118-119): on a box a hundred-thousandth of a pixel wide the quotient is the fp32 rounding noise of the integral image times 1e5 --
several times the largest feature value -- and it goes into ``collapse`` like any other voxel feature.  The default product of the
frame kernels splits its operands into fp16 pieces under a power-of-two scale (vfa_amd/csrc/vfa_split.h); round 4 sized that scale for
honest boxes (|vox| <= absmax / 4) and the verdict found Wildtrack cells 1.3 x short of fp16's 65504.  Now the geometry pass bounds the
noise of every box (vfa_geom.h: sliver_shift) and the kernels scale the affected items down by that many binary places.

These tests pick the hard cells with the ORACLE over the WHOLE grid of every BASELINE workload -- visible boxes of area < 1e-3, boxes
clamped at the crange limits, and all 32 cells of the tiles that hold the boxes with the largest noise shift (their honest neighbours
are scaled down with them) -- run the whole frame through the product path (``pool_collapse_kernel`` / ``pipe_kernel``, default
arithmetic) and compare exactly those cells with the float64 product of the oracle's voxel features: rtol 1e-4, atol 1e-5 x the
largest HONEST sum (contributions of boxes with a noise shift are left out of that scale: round-5 verdict).  A synthetic near-constant map (the worst case of the noise bound: every
integral-image entry as large as it can be) drives the same boxes far beyond 65504 in the round-4 scaling.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL_REL = 1e-4, 1e-5
C = 256


def _mods(wl, dev):
    import vfa_amd
    torch.manual_seed(2)
    mods = [vfa_amd.VFA(C, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    with torch.no_grad():
        for m in mods:
            m.collapse.weight.mul_(3.0)
            m.collapse.bias.uniform_(-0.3, 0.1)
    return mods


TILE_L, TILE_W = 4, 8   # the 32-cell tile of the frame kernels (vfa_fused.hip: kTileL x kTileW); an item = (tile, view, scale)


def _sliver_shift(area, Hf, Wf):
    """vfa_amd/csrc/vfa_geom.h: sliver_shift, restated (float32 like the device code)."""
    bound = np.float32(0.25) + (np.float32(Hf) * np.float32(Wf) * np.float32(2.0 ** -18)) / area.astype(np.float32)
    e = np.floor(np.log2(np.maximum(bound, 1.0).astype(np.float64))).astype(np.int64) + 1
    return np.where(bound >= 1.0, np.minimum(e, 48), 0)


def _tile_cells(cell, L, W):
    """All cells of the 32-cell tile that holds `cell` (row-major L x W grid)."""
    l0, w0 = (cell // W) // TILE_L * TILE_L, (cell % W) // TILE_W * TILE_W
    return [l * W + w for l in range(l0, min(L, l0 + TILE_L)) for w in range(w0, min(W, w0 + TILE_W))]


def _hard_cells(oracle, wl, cams, feats, per_pair=24, clamped=6, shifted_tiles=12, crange=(-1, 0.95)):
    """-> (sorted cell indices, report).  Per (camera, scale), over the WHOLE grid and every layer: the visible boxes with the smallest
    areas below 1e-3, a few visible boxes that touch the crange clamp, and -- round-5 verdict -- ALL 32 cells of the tiles that hold the
    boxes with the largest noise shift (sliver_shift > 0): the honest neighbours of a shifted item are scaled down with it (in the
    pipelined kernel: every view and layer of the (tile, scale)) and must keep their accuracy."""
    grid_np = wl["grid"][0].reshape(-1, 3).numpy()
    L, W = wl["grid"].shape[1:3]
    zl = oracle.z_layers_of(wl["grid_height"], wl["cube_size"])
    co = oracle.corner_offsets(wl["cube_size"])
    chosen, smallest = set(), np.inf
    n_small, n_shifted, top_shift = 0, 0, 0
    for si in range(3):
        Hf, Wf = feats[cams[0]][si].shape[-2:]
        for cam in cams:
            box, area, vis = oracle.box_params(wl["calibs"][cam].numpy(), grid_np, zl, co, wl["args"].data, wl["args"].image_size, Hf, Wf,
                                               crange)
            a = np.where(vis, area, np.inf)                       # (nl, cells)
            flat = a.reshape(-1)
            small = np.flatnonzero(flat < 1e-3)
            n_small += small.size
            if small.size:
                order = small[np.argsort(flat[small])][:per_pair]
                chosen.update((order % a.shape[1]).tolist())
                smallest = min(smallest, float(flat[order[0]]))
            lo, hi = np.float32(crange[0]), np.float32(crange[1])
            touch = vis & ((box[..., 0] == lo) | (box[..., 1] == lo) | (box[..., 2] == hi) | (box[..., 3] == hi))
            tflat = np.flatnonzero(touch.reshape(-1))
            if tflat.size:
                order = tflat[np.argsort(flat[tflat])][:clamped]
                chosen.update((order % a.shape[1]).tolist())
            sh = np.where(vis, _sliver_shift(area, Hf, Wf), 0).max(axis=0)   # (cells): the largest shift over the layers
            n_shifted += int((sh > 0).sum())
            top_shift = max(top_shift, int(sh.max()))
            for cell in np.argsort(-sh, kind="stable")[:shifted_tiles]:
                if sh[cell] > 0:
                    chosen.update(_tile_cells(int(cell), L, W))
    return np.array(sorted(chosen), dtype=np.int64), dict(small_boxes=n_small, smallest_area=smallest, shifted_cells=n_shifted,
                                                           top_shift=top_shift)


def _oracle_rows(oracle, wl, cams, feats, mods, cells, crange=(-1, 0.95)):
    """float64 sum_scale sum_camera relu(vox . W^T + b) at `cells` from the oracle's voxel features; the same sum over the HONEST
    contributions only (a (camera, scale, cell) none of whose layers holds a box with a noise shift); the largest |vox| / absmax."""
    grid_np = wl["grid"][0].reshape(-1, 3).numpy()[cells]
    zl = oracle.z_layers_of(wl["grid_height"], wl["cube_size"])
    co = oracle.corner_offsets(wl["cube_size"])
    want = np.zeros((len(cells), C), np.float64)
    honest = np.zeros((len(cells), C), np.float64)
    worst_ratio, worst_scaled = 0.0, 0.0
    for si, m in enumerate(mods):
        w64 = m.collapse.weight.detach().cpu().double().numpy()
        b64 = m.collapse.bias.detach().cpu().double().numpy()
        absmax = max(float(feats[cam][si].abs().max()) for cam in cams)   # (the scale of the split is per feature scale, over the views)
        for cam in cams:
            f = feats[cam][si][0].numpy()
            Hf, Wf = f.shape[1:]
            I = oracle.integral_image(f)
            box, area, vis = oracle.box_params(wl["calibs"][cam].numpy(), grid_np, zl, co, wl["args"].data, wl["args"].image_size, Hf, Wf,
                                               crange)
            ref = oracle.gather(I, box, area, vis)
            peak = float(np.abs(ref).max())
            worst_ratio = max(worst_ratio, peak / absmax)
            # what round 4's scaling (absmax 2^ea in [2^14, 2^15)) would have handed to the fp16 conversion
            worst_scaled = max(worst_scaled, peak * 2.0 ** (14 - int(np.floor(np.log2(absmax)))))
            term = np.maximum(ref.astype(np.float64) @ w64.T + b64, 0.0)
            want += term
            clean = (np.where(vis, _sliver_shift(area, Hf, Wf), 0).max(axis=0) == 0)   # (cells)
            honest += term * clean[:, None]
    return want, honest, worst_ratio, worst_scaled


def _run_and_check(oracle, wl, cams, feats, name, crange=(-1, 0.95), cells=None):
    import vfa_amd
    from vfa_amd import ops
    dev = torch.device("cuda:0")
    mods = _mods(wl, dev)
    nl = mods[0].num_grid_layer
    if cells is None:
        cells, rep = _hard_cells(oracle, wl, cams, feats, crange=crange)
    else:
        rep = dict(small_boxes=-1, smallest_area=np.nan, shifted_cells=-1, top_shift=-1)
    assert len(cells) > 0, f"{name}: the oracle found no hard cell on the whole grid"
    want, honest, ratio, scaled = _oracle_rows(oracle, wl, cams, feats, mods, cells, crange)
    lats = [torch.cat([feats[c][s] for c in cams]).to(dev) for s in range(3)]
    calibs, grid = wl["calibs"][list(cams)].to(dev), wl["grid"].to(dev)
    L, W = grid.shape[1:3]
    with torch.no_grad(), ops.KernelTimer() as kt:
        ortho = vfa_amd.aggregate_views(*mods, *lats, calibs, grid, crange)
    torch.cuda.synchronize()
    ran = set(kt.summary())
    assert ("vfa_pool_collapse_relu_sum_f32" if nl == 1 else "vfa_pipe_collapse_relu_sum_f32") in ran, sorted(ran)
    assert not ran & {"vfa_project_gather_f32", "vfa_collapse_gemm_f32", "vfa_pool_windows_f32"}, sorted(ran)
    assert torch.isfinite(ortho).all()
    got = ortho[0].permute(1, 2, 0).reshape(L * W, C)[torch.from_numpy(cells).to(dev)].cpu().double().numpy()
    # The absolute part of the tolerance is set by the HONEST contributions alone (round-5 verdict: max|want| holds the sliver noise
    # itself, up to 20 x absmax on a near-constant map -- the thing under test inflated the scale it was judged against); a cell that
    # holds a noise term still gets rtol of that term.
    scale = np.abs(honest).max()
    assert scale > 0, f"{name}: no honest contribution among the chosen cells"
    tol = RTOL * np.abs(want) + ATOL_REL * scale
    worst = float((np.abs(got - want) / tol).max())
    print(f"[slivers] {name}: {len(cells)} cells ({rep['small_boxes']} visible boxes of area < 1e-3, smallest {rep['smallest_area']:.2e}; "
          f"{rep['shifted_cells']} (camera, scale, cell) with a noise shift, largest {rep['top_shift']}), largest |vox| / absmax {ratio:.2f} "
          f"(round-4 scaling: {scaled:.0f} of 65504), max|want| / honest scale {np.abs(want).max() / scale:.2f}, worst |err| / tolerance {worst:.3f}")
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL_REL * scale, err_msg=f"{name}: fused map at the hard cells")
    return ratio, scaled, rep


WORKLOADS = ["multiviewc_200x200x1", "multiviewc_156x156x5", "wildtrack_480x1440x1", "wildtrack_120x360x8", "multiviewx_160x250x8",
             "synthetic4k_512x512x32"]


@pytest.mark.parametrize("name", WORKLOADS)
def test_hard_cells_of_every_baseline_workload_through_the_fused_kernels(oracle, name):
    from vfa_amd.synthetic import make_workload
    wl = make_workload(name, channels=C, seed=4)
    cams = tuple(range(wl["n_cam"]))
    _run_and_check(oracle, wl, cams, wl["features"], name)


def test_wildtrack_cameras_4_and_5_where_round_4_had_a_factor_1p3_left(oracle):
    """The cells of the round-4 verdict: wildtrack_120x360x8, seed 4, cameras 4-5 -- visible boxes of area 1e-5 .. 4e-5 whose voxel
    features reach 1.1 - 2.8 x the largest feature value."""
    from vfa_amd.synthetic import make_workload
    wl = make_workload("wildtrack_120x360x8", channels=C, seed=4)
    ratio, scaled, rep = _run_and_check(oracle, wl, (4, 5), wl["features"], "wildtrack_120x360x8 cameras 4-5")
    assert rep["small_boxes"] > 0 and ratio > 1.0, (rep, ratio)  # (the test must be where the verdict was)


def test_near_constant_map_drives_the_same_boxes_past_65504(oracle):
    """The worst case of the noise bound: a map whose values all sit near its maximum (0.02 relu(randn) + 4), so that every
    integral-image entry -- and with it the rounding noise a sliver box divides by its area -- is as large as absmax allows.  In
    round 4's scaling the same Wildtrack boxes then overflow fp16 several times over (asserted: the test is only worth something
    there); with the per-item shift the fused map still matches the reference arithmetic at those cells."""
    from vfa_amd.synthetic import make_workload
    wl = make_workload("wildtrack_120x360x8", channels=C, seed=4)
    feats = [None if f is None else [0.02 * t + 4.0 for t in f] for f in wl["features"]]
    ratio, scaled, rep = _run_and_check(oracle, wl, (4, 5), feats, "wildtrack_120x360x8 cameras 4-5, near-constant map")
    assert scaled > 65504.0, f"largest scaled voxel feature {scaled:.0f}: this map does not cross fp16's range"


def _diagonal_rig(nl):
    """One camera whose two image rows of the calibration are EQUAL, on a square image: u == v bit for bit for every corner, so every box
    is a square on the image diagonal (left == top, right == bottom) -- a legal input like any other 3 x 4 matrix, and the way to steer a
    box onto the smallest area the reference keeps visible.  -> workload dict (C5's 4K feature maps: the finest is 270 x 480, the
    largest map of any BASELINE config), a 16 x 32 sub-grid around the grid centre."""
    from types import SimpleNamespace
    from vfa_amd.synthetic import make_workload
    wl = make_workload("synthetic4k_512x512x32", channels=C, seed=9, n_cam=1)
    wl["args"] = SimpleNamespace(data="MultiviewC", image_size=(2160, 2160))
    calib = wl["calibs"][0].clone()
    calib[1] = calib[0]
    wl["calibs"] = calib[None]
    wl["grid"] = wl["grid"][:, 248:264, 240:272].contiguous()
    wl["grid_height"] = wl["cube_size"][2] * nl          # nl layers of one cube height each
    wl["features"] = [[0.02 * t + 4.0 for t in wl["features"][0]]]   # near-constant: the worst case of the noise bound
    return wl


@pytest.mark.parametrize("nl", [1, 4])
def test_largest_reachable_shift_next_to_honest_boxes(oracle, nl):
    """The largest shift the geometry can ask for, inside a tile of honest boxes (round-5 verdict, weak 1a).  area = fl(w h Hf Wf +
    1e-6) and a box is visible from area > fl(1e-6) (vfa_op.py:104-106); normalised coordinates are multiples of 2^-24 near 0.5, so the
    smallest visible box is ONE such step wide and high: area = 1e-6 (1 + 4.6e-4) on the 270 x 480 map, noise bound 2^18.9, shift 19
    (vfa_geom.h: sliver_shift; 1e-6 (1 + 2^-23) itself is not reachable: the box extent is quantised).  The upper clamp of `crange`
    (an argument of VFA.forward, vfa_op.py:61, 76-77) is put one step above the left edge of a chosen cell's box: that box becomes
    the sliver -- its voxel features are rounding noise / 1e-6, ten thousand times the map's maximum --, the boxes to its left in the same
    32-cell tile stay honest (clamped on the right only), those to its right vanish.  nl = 1: the serial kernel shifts the (tile,
    view, scale) item; nl = 4: the pipelined kernel shifts the (tile, scale) over all layers.  Every cell of the 16 x 32 sub-grid is
    compared; the absolute tolerance comes from the honest contributions alone."""
    wl = _diagonal_rig(nl)
    L, W = wl["grid"].shape[1:3]
    zl = oracle.z_layers_of(wl["grid_height"], wl["cube_size"])
    assert len(zl) == nl
    co = oracle.corner_offsets(wl["cube_size"])
    grid_np = wl["grid"][0].reshape(-1, 3).numpy()
    Hf, Wf = wl["features"][0][0].shape[-2:]
    assert (Hf, Wf) == (270, 480)
    calib = wl["calibs"][0].numpy()
    box, area, vis = oracle.box_params(calib, grid_np, zl, co, "MultiviewC", (2160, 2160), Hf, Wf)
    assert (box[..., 0] == box[..., 1]).all() and (box[..., 2] == box[..., 3]).all(), "diagonal rig: u == v bit for bit"
    assert vis.all() and (_sliver_shift(area, Hf, Wf) == 0).all(), "under the default crange every box of the sub-grid is honest"
    target = (L // 2 + 1) * W + W // 2 + 3               # a cell inside a tile, with neighbours on both sides
    left = box[0, target, 0]
    assert 0.25 < left < 0.9
    hi = float(np.nextafter(left, np.float32(2.0)))     # one step above the left edge of the target's box
    crange = (-1.0, hi)
    box2, area2, vis2 = oracle.box_params(calib, grid_np, zl, co, "MultiviewC", (2160, 2160), Hf, Wf, crange)
    sh2 = np.where(vis2, _sliver_shift(area2, Hf, Wf), 0)
    assert vis2[0, target] and sh2[0, target] == 19 and abs(float(area2[0, target]) / 1e-6 - 1.0) < 1e-3, (area2[0, target], sh2[0, target])
    tile = _tile_cells(target, L, W)
    honest_in_tile = [c for c in tile if vis2[:, c].any() and sh2[:, c].max() == 0]
    assert len(honest_in_tile) >= 4, "the sliver must have honest neighbours in its tile"
    cells = np.arange(L * W, dtype=np.int64)
    ratio, scaled, _ = _run_and_check(oracle, wl, (0,), wl["features"], f"diagonal rig, nl = {nl}, crange hi = {hi!r}", crange=crange,
                                      cells=cells)
    assert ratio > 100.0, f"largest |vox| / absmax {ratio:.1f}: the sliver's noise should dwarf the map"


def test_sliver_shift_bound_and_fp16_headroom_on_the_hard_cells(oracle):
    """The bound behind the shift, checked where it matters: on the hard cells of the Wildtrack rig (ordinary and near-constant
    maps) the oracle's |vox| never exceeds absmax (1/4 + 2^-18 Hf Wf / area) -- the quantity vfa_geom.h turns into binary places --,
    and what the fp16 conversion of the frame kernels is handed, |vox| 2^(ea - shift) with absmax 2^ea in [2^13, 2^14) (vfa_split.h:
    kExpA) and the box's OWN shift (its item's is at least that), stays a factor >= 8 below 65504 on the worst box found."""
    from vfa_amd.synthetic import make_workload
    wl = make_workload("wildtrack_120x360x8", channels=C, seed=4)
    zl = oracle.z_layers_of(wl["grid_height"], wl["cube_size"])
    co = oracle.corner_offsets(wl["cube_size"])
    grid_np = wl["grid"][0].reshape(-1, 3).numpy()
    nl = len(zl)
    largest, shifted = 0.0, 0
    for feats in (wl["features"], [[0.02 * t + 4.0 for t in f] for f in wl["features"]]):
        cells, _ = _hard_cells(oracle, wl, (4, 5), feats)
        for si in range(3):
            absmax = max(float(feats[cam][si].abs().max()) for cam in (4, 5))
            ea = 13 - int(np.floor(np.log2(absmax)))
            for cam in (4, 5):
                f = feats[cam][si][0].numpy()
                Hf, Wf = f.shape[1:]
                box, area, vis = oracle.box_params(wl["calibs"][cam].numpy(), grid_np[cells], zl, co, wl["args"].data, wl["args"].image_size, Hf, Wf)
                ref = oracle.gather(oracle.integral_image(f), box, area, vis)            # (cells, C * nl), column c * nl + layer
                peak = np.abs(ref.reshape(len(cells), C, nl)).max(axis=1).T              # (nl, cells)
                bound = absmax * (0.25 + 2.0 ** -18 * Hf * Wf / area.astype(np.float64))
                assert (peak[vis] <= bound[vis]).all()
                sh = _sliver_shift(area, Hf, Wf)
                shifted += int((sh[vis] > 0).sum())
                if vis.any():
                    largest = max(largest, float((peak[vis] * 2.0 ** (ea - sh[vis])).max()))
    print(f"[slivers] largest scaled voxel feature on the hard cells {largest:.1f} = 65504 / {65504 / largest:.0f}; {shifted} boxes with a shift")
    assert shifted > 0 and largest * 8.0 <= 65504.0
