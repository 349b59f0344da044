"""The pipelined per-frame inference path for ANY number of z-layers (``vfa_pipe_records_f32`` +
``vfa_pipe_collapse_relu_sum_f32``: geometry once per frame; pooling waves and matrix waves side by side in one persistent kernel;
the accumulators of four views in registers across all layers; vox never in HBM) against the older kernels, float64 from
bit-pinned voxel features and the CPU oracle.   ``-m gpu``.

Reference lines covered: vfa/model/vfa_op.py:50-59 (collapse = Linear(C * nl -> C)), :61-125 (all of VFA.forward),
vfa/model/vfanet.py:64-82 (camera loop); shipped layer counts vfa/config.py:22-24, 49-52, 77-80.
"""
from types import SimpleNamespace

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL_REL = 1e-4, 1e-5
PIPE_ENTRY = "vfa_pipe_collapse_relu_sum_f32"


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _mods(wl, dev, seed=1, scale=3.0, n=3):
    import vfa_amd
    torch.manual_seed(seed)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
            for _ in range(n)]
    with torch.no_grad():
        for m in mods:  # bigger weights and a negative-leaning bias: the ReLU cuts a real share of the outputs
            m.collapse.weight.mul_(scale)
            m.collapse.bias.uniform_(-0.3, 0.1)
    return mods


def _float64_reference(mods, lats, calibs, grid, wl, cells=None):
    """sum_scale sum_view relu(vox . W^T + b) in float64 from the BITWISE-pinned voxel features of the direct pooling kernel
    (layer-major vox against the layer-major view of collapse.weight)."""
    from vfa_amd import _lib, ops
    dev = grid.device
    n = calibs.shape[0]
    grid_flat = grid.reshape(-1, 3).contiguous()
    cells = grid_flat.shape[0] if cells is None else cells
    want = torch.zeros(cells, 256, dtype=torch.float64, device=dev)
    for m, lat in zip(mods, lats):
        zl, co = m._kernel_geometry(dev)
        vox = ops.project_gather(ops.integral_image(lat), calibs.reshape(n, 12).contiguous(), grid_flat, zl, co,
                                 _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], cell_count=cells, kernel="direct")
        want += torch.relu(vox.double() @ m.layer_major_weight().double().T + m.collapse.bias.double()).sum(0)
    return want


def _check(name, got, want):
    scale = want.abs().max().item()
    assert scale > 0
    tol = RTOL * want.abs() + ATOL_REL * scale
    worst = ((got.double() - want).abs() / tol).max().item()
    print(f"[margin] {name}: worst |err| / tolerance = {worst:.3f}")
    torch.testing.assert_close(got.double(), want, rtol=RTOL, atol=ATOL_REL * scale, msg=lambda m: f"{name}: {m}")


def _frame(name, n_cam, crop, dev, seed=3, origin=(11, 5)):
    from vfa_amd.synthetic import make_workload
    wl = make_workload(name, channels=256, seed=seed, **({"n_cam": n_cam} if n_cam else {}))
    n = wl["n_cam"]
    grid = wl["grid"] if crop is None else wl["grid"][:, origin[0]:origin[0] + crop[0], origin[1]:origin[1] + crop[1]].contiguous()
    lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
    return wl, grid.to(dev), lats, wl["calibs"].to(dev)


SINGLE = [  # workload, cameras, grid crop (rows, cols) or None
    ("multiviewc_200x200x1", None, None),          # the bench frame: 7 cameras (groups of 4 + 3) x 3 scales, 1250 full tiles
    ("multiviewc_200x200x1", 2, (37, 53)),         # ragged grid, one half-empty group per (tile, scale)
    ("wildtrack_480x1440x1", 3, (100, 1440)),      # Wildtrack conversion, 1080p maps
    ("multiviewc_200x200x1", 11, (45, 64)),        # three groups per (tile, scale)
    ("multiviewc_200x200x1", 8, (16, 24)),         # fewer tiles than workgroups: nearly every tile is shared by two of them
]


@pytest.mark.parametrize("name,n_cam,crop", SINGLE)
def test_pipe_equals_the_serial_fused_kernel_bit_for_bit_on_single_layer_grids(name, n_cam, crop, monkeypatch):
    """Same pooling arithmetic, same product sequence (k ascending, lo.hi / hi.hi / hi.lo per k-step): on nl = 1 the pipelined kernel
    must reproduce ``vfa_pool_collapse_relu_sum_f32`` up to the association of the view / scale sum."""
    import vfa_amd
    from vfa_amd import ops, vfa_op
    dev = _dev()
    wl, grid, lats, calibs = _frame(name, n_cam, crop, dev)
    mods = _mods(wl, dev)
    L, W = grid.shape[1:3]
    with torch.no_grad():
        monkeypatch.setattr(vfa_op, "PIPE", True)
        monkeypatch.setattr(vfa_op, "PIPE_SINGLE_LAYER", True)
        with ops.KernelTimer() as kt:
            piped = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        assert PIPE_ENTRY in kt.summary() and "vfa_pool_collapse_relu_sum_f32" not in kt.summary(), sorted(kt.summary())
        monkeypatch.setattr(vfa_op, "PIPE", False)
        with ops.KernelTimer() as kt:
            serial = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        assert "vfa_pool_collapse_relu_sum_f32" in kt.summary() and PIPE_ENTRY not in kt.summary(), sorted(kt.summary())
        monkeypatch.setattr(vfa_op, "PIPE", True)
        again = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        # Same pooled features, same products; what differs is the association of the sum over views and scales: the serial kernel
        # adds relu(view) one by one to the tile's running sum, the pipelined one adds a GROUP's views first (((r0 + r1) + r2) +
        # r3, the running sum is fetched under the group's last step) and the group to the tile.  Also compared on 8 workgroups
        # (VFA_FLAG_RESERVED_CUS(248): at most 7 tiles cut between workgroups in each kernel).
        p8 = vfa_op.pipe_frame(mods, lats, calibs, grid, reserved_cus=248)
        s8 = vfa_op.fused_frame(mods, lats, calibs, grid, reserved_cus=248)
    assert torch.isfinite(piped).all()
    p = piped[0].permute(1, 2, 0).reshape(L * W, 256)
    s = serial[0].permute(1, 2, 0).reshape(L * W, 256)
    _check(f"{name} pipe vs serial kernel, full launch", p, s.double())
    # <= 21 roundings of partial sums apart (7 views x 3 scales): a few 1e-7 of the largest value
    for a, b, what in ((p8, s8, "8 workgroups"), (p, s, "full launch")):
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err <= 1.5e-6, f"{name} pipe vs serial kernel, {what}: max |diff| / max = {err:.3e}"
    _check(f"{name} pipe vs serial kernel, 8 workgroups", p8, s8.double())
    assert torch.equal(again, piped)  # deterministic, shared tiles included (fixed addition order of the parts)


MULTI = [  # workload, cameras, crop, origin
    ("multiviewc_156x156x5", None, (40, 64), (60, 40)),   # shipped MultiviewC: 5 layers, K = 1280
    ("wildtrack_120x360x8", 7, (24, 96), (50, 130)),      # shipped Wildtrack: 8 layers, many masked boxes
    ("multiviewx_160x250x8", 6, (28, 72), (70, 90)),      # shipped MultiviewX: 6 cameras = groups of 4 + 2
    ("synthetic4k_512x512x32", 3, (12, 40), (250, 240)),  # 32 layers, K = 8192, 4K feature maps
    ("multiviewc_156x156x5", 1, (21, 35), (10, 100)),     # one camera: every group has one sub-tile
]


@pytest.mark.parametrize("name,n_cam,crop,origin", MULTI)
def test_pipe_multi_layer_vs_float64_and_the_vox_through_hbm_path(name, n_cam, crop, origin, monkeypatch):
    import vfa_amd
    from vfa_amd import ops, vfa_op
    dev = _dev()
    wl, grid, lats, calibs = _frame(name, n_cam, crop, dev, origin=origin)
    mods = _mods(wl, dev)
    nl = mods[0].num_grid_layer
    assert nl > 1
    L, W = grid.shape[1:3]
    with torch.no_grad():
        monkeypatch.setattr(vfa_op, "PIPE", True)
        with ops.KernelTimer() as kt:
            piped = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        assert PIPE_ENTRY in kt.summary() and "vfa_project_gather_f32" not in kt.summary(), sorted(kt.summary())
        monkeypatch.setattr(vfa_op, "PIPE", False)
        with ops.KernelTimer() as kt:
            older = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        assert PIPE_ENTRY not in kt.summary() and "vfa_project_gather_f32" in kt.summary(), sorted(kt.summary())
        want = _float64_reference(mods, lats, calibs, grid, wl)
        monkeypatch.setattr(vfa_op, "PIPE", True)
        again = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
    p = piped[0].permute(1, 2, 0).reshape(L * W, 256)
    o = older[0].permute(1, 2, 0).reshape(L * W, 256)
    assert torch.isfinite(p).all()
    _check(f"{name} x{nl} pipe vs float64", p, want)
    _check(f"{name} x{nl} vox-through-HBM path vs float64", o, want)
    assert torch.equal(again, piped)


@pytest.mark.parametrize("name,n_cam,crop,origin", [
    ("multiviewc_156x156x5", 1, (40, 64), (60, 40)),      # one camera per rank: every group has ONE sub-tile
    ("multiviewc_156x156x5", 2, (37, 53), (10, 100)),     # two cameras, ragged grid: groups of one and of two views
    ("multiviewc_200x200x1", 2, None, None),              # single layer, 1250 tiles, tiles cut between workgroups
    ("wildtrack_120x360x8", 2, (24, 96), (50, 130)),      # eight layers, many masked boxes
])
def test_one_and_two_view_frames_fill_their_groups_across_tiles(name, n_cam, crop, origin):
    """A frame of one or two views (a rank's share of a camera-sharded rig): a group takes its four sub-tiles from up to four TILES of a
    run (round 6; rounds 4-5 ran a four-step phase with one or two sub-tiles per pass of the weight).  Against float64, and the
    production build against the diagnostic build (VFA_FLAG_DEBUG 8: "no wave priorities", nothing else) bit for bit, on 256 and
    on 8 workgroups (runs shared between workgroups at other places)."""
    from vfa_amd import _lib, ops
    dev = _dev()
    wl, grid, lats, calibs = _frame(name, n_cam, crop, dev, origin=origin or (11, 5))
    mods = _mods(wl, dev)
    nl = mods[0].num_grid_layer
    L, W = grid.shape[1:3]
    zl, co = mods[0]._kernel_geometry(dev)
    with torch.no_grad():
        integrals = ops.integral_images(lats)
        ws = ops.pipe_records(calibs, grid, zl, co, _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1],
                              [tuple(l.shape[-2:]) for l in lats], weights=[m.collapse.weight for m in mods])
        biases = [m.collapse.bias for m in mods]
        with ops.KernelTimer() as kt:
            small = ops.pipe_collapse(integrals, biases, ws, (L, W), nl, absmax=integrals.absmax)
        eight = ops.pipe_collapse(integrals, biases, ws, (L, W), nl, absmax=integrals.absmax, debug=8)
        small8 = ops.pipe_collapse(integrals, biases, ws, (L, W), nl, absmax=integrals.absmax, reserved_cus=248)
        eight8 = ops.pipe_collapse(integrals, biases, ws, (L, W), nl, absmax=integrals.absmax, reserved_cus=248, debug=8)
        bf = ops.pipe_collapse(integrals, biases, ops.pipe_records(
            calibs, grid, zl, co, _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], [tuple(l.shape[-2:]) for l in lats],
            weights=[m.collapse.weight for m in mods], terms=3), (L, W), nl, terms=3)
        want = _float64_reference(mods, lats, calibs, grid, wl)
    torch.cuda.synchronize()
    assert PIPE_ENTRY in kt.summary()
    assert torch.isfinite(small).all() and small.abs().max() > 0
    assert torch.equal(small, eight), (small - eight).abs().max().item()
    assert torch.equal(small8, eight8), (small8 - eight8).abs().max().item()
    _check(f"{name} x{nl}, {n_cam} views, four-step phase vs float64", small, want)
    _check(f"{name} x{nl}, {n_cam} views, four-step phase (bf16 x 2) vs float64", bf, want)


@pytest.mark.parametrize("data,image_size,cube,gh,world,step,cam", [
    ("MultiviewC", (720, 1280), (75.0, 75.0, 40), 160, (3750, 3750), (75.0, 75.0), "ring"),       # 4 layers
    ("MultiviewX", (1080, 1920), (8, 8, 16), 64, (640, 1000), (8, 8), "mx"),                        # 4 layers
    ("Wildtrack", (1080, 1920), (12, 12, 8), 24, (480, 1440), (12, 12), "wt"),                      # 3 layers
    ("MultiviewC", (720, 1280), (150.0, 150.0, 100), 300, (3750, 3750), (150.0, 150.0), "inside"),  # 3 layers, huge near boxes
])
def test_pipe_vs_oracle_small_multi_layer_scenes(oracle, data, image_size, cube, gh, world, step, cam):
    """Multi-layer scenes of every dataset conversion against the CPU oracle end to end (oracle voxel features in the
    reference's column order c * nl + layer, float64 product with collapse.weight as it is): cameras far, near and INSIDE the
    field (boxes behind the camera, boxes whose tap window does not fit LDS and are pooled from L2, fully masked tiles, views
    and layers)."""
    import vfa_amd
    from vfa_amd.synthetic import look_at_camera, ring_cameras
    from vfa_amd.utils import make_grid
    dev = _dev()
    H, W_img = image_size
    if cam == "ring":
        calibs = ring_cameras(5, (1875.0, 1875.0, 0.0), 2700.0, 600.0, 900.0, (W_img, H))
    elif cam == "inside":
        calibs = torch.tensor(np.stack([look_at_camera((1500.0, 1700.0, 250.0), (2600.0, 2300.0, 0.0), 700.0, (W_img, H)),
                                        look_at_camera((300.0, 300.0, 200.0), (1800.0, 1900.0, 0.0), 500.0, (W_img, H))]),
                              dtype=torch.float32)
    elif cam == "mx":
        calibs = torch.tensor(np.stack([look_at_camera((-5.0, 8.0, 3.0), (12.0, 8.0, 0.0), 1700.0, (W_img, H)),
                                        look_at_camera((30.0, 20.0, 2.5), (12.0, 6.0, 0.0), 1400.0, (W_img, H))]),
                              dtype=torch.float32)
    else:
        wt_c = (480 * 2.5 / 2 - 300.0, 1440 * 2.5 / 2 - 900.0, 0.0)
        calibs = ring_cameras(3, wt_c, 0.45 * 1440 * 2.5, 400.0, 1100.0, (W_img, H))
    grid = make_grid(world_size=world, cube_LW=list(step), dataset=data)
    args = SimpleNamespace(data=data, image_size=image_size)
    n = calibs.shape[0]
    gen = torch.Generator().manual_seed(7)
    sizes = [(45, 80), (23, 40), (12, 20)]
    lats = [torch.relu(torch.randn(n, 256, h, w, generator=gen)) for h, w in sizes]
    torch.manual_seed(3)
    mods = [vfa_amd.VFA(256, grid_height=gh, cube_size=cube, args=args).to(dev) for _ in range(3)]
    nl = mods[0].num_grid_layer
    assert nl > 1
    with torch.no_grad():
        for m in mods:
            m.collapse.weight.mul_(3.0)
            m.collapse.bias.uniform_(-0.3, 0.1)
        with vfa_amd.ops.KernelTimer() as kt:
            out = vfa_amd.aggregate_views(*mods, *(l.to(dev) for l in lats), calibs.to(dev), grid.to(dev)[None])
        torch.cuda.synchronize()
    assert PIPE_ENTRY in kt.summary()
    L, W = grid.shape[:2]
    zl_h, co_h = oracle.z_layers_of(gh, cube), oracle.corner_offsets(cube)
    want = np.zeros((L * W, 256), np.float64)
    vis_total = 0
    for si, m in enumerate(mods):
        w64 = m.collapse.weight.detach().cpu().double().numpy()
        b64 = m.collapse.bias.detach().cpu().double().numpy()
        for c in range(n):
            f = lats[si][c].numpy()
            box, area, vis = oracle.box_params(calibs[c].numpy(), grid.reshape(-1, 3).numpy(), zl_h, co_h, data, image_size,
                                               f.shape[1], f.shape[2])
            vox = oracle.gather(oracle.integral_image(f), box, area, vis)
            vis_total += int(vis.sum())
            want += np.maximum(vox.astype(np.float64) @ w64.T + b64, 0.0)
    assert vis_total > 0
    got = out[0].permute(1, 2, 0).reshape(L * W, 256).cpu()
    _check(f"{data}/{cam} x{nl}", got, torch.from_numpy(want))


def test_pipe_single_scale_accumulate_degenerate_grids_and_bands(monkeypatch):
    """`VFA.forward` (one camera, one scale: the reference's own interface) on a multi-layer grid; accumulate on top of an
    existing map; grids smaller than one 8 x 4 tile; a view that sees nothing; the frame in bands of grid rows."""
    import vfa_amd
    from vfa_amd import ops, vfa_op
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload("multiviewc_156x156x5", channels=256, seed=5, n_cam=2)
    mods = _mods(wl, dev, seed=2)
    lat = wl["features"][0][1].to(dev)
    calib = wl["calibs"][0].to(dev)
    for rows, cols in ((3, 5), (4, 8), (1, 1), (9, 17)):
        grid = wl["grid"][:, 70:70 + rows, 80:80 + cols].contiguous().to(dev)
        with torch.no_grad(), ops.KernelTimer() as kt:
            out = vfa_amd.materialize(mods[1](lat, calib, grid))  # (inference results are deferred: vfa_amd/lazy.py)
        torch.cuda.synchronize()
        assert PIPE_ENTRY in kt.summary()
        want = _float64_reference([mods[1]], [lat], calib[None], grid, wl)
        _check(f"forward {rows}x{cols}", out[0].permute(1, 2, 0).reshape(rows * cols, 256), want)
    # accumulate
    grid = wl["grid"][:, 40:61, 30:70].contiguous().to(dev)
    base = torch.randn(21 * 40, 256, device=dev)
    feats = [wl["features"][1][0].to(dev)]
    with torch.no_grad():
        got = vfa_op.pipe_frame([mods[0]], feats, wl["calibs"][1:2].to(dev), grid, out=base.clone(), accumulate=True)
        want = base.double() + _float64_reference([mods[0]], feats, wl["calibs"][1:2].to(dev), grid, wl)
    _check("accumulate", got, want)
    # a camera looking away: every box masked -> every output row is relu(bias) (vox = 0)
    away = torch.tensor([[900., 0, 640, -1e9], [0, 900., 360, -1e9], [0, 0, 0., 1.]], device=dev)  # every corner clamps to -1
    with torch.no_grad():
        out = mods[2](wl["features"][0][2].to(dev), away, grid)
    want = torch.relu(mods[2].collapse.bias.detach()).expand(21 * 40, 256)
    assert torch.equal(out[0].permute(1, 2, 0).reshape(-1, 256), want)
    # bands: a workspace limit that forces several passes over bands of grid rows gives the same map
    grid = wl["grid"][:, 30:83, 20:101].contiguous().to(dev)
    lats = [torch.cat([wl["features"][c][s] for c in range(2)]).to(dev) for s in range(3)]
    calibs = wl["calibs"].to(dev)
    with torch.no_grad():
        whole = vfa_op.pipe_frame(mods, lats, calibs, grid)
        monkeypatch.setattr(vfa_op, "PIPE_WS_LIMIT", ops.pipe_workspace_bytes(2, 16, 81, 5, 3))
        with ops.KernelTimer() as kt:
            banded = vfa_op.pipe_frame(mods, lats, calibs, grid)
        torch.cuda.synchronize()
    assert kt.summary()[PIPE_ENTRY]["launches"] >= 3
    # (the bands cut tiles between workgroups at other places: another association of the sums of those tiles)
    _check("banded vs whole", banded, whole.double())


@pytest.mark.parametrize("name,n_cam,crop,origin", [
    ("multiviewc_156x156x5", 3, (41, 67), (20, 10)),   # ragged grid, five layers
    ("wildtrack_120x360x8", 2, (24, 96), (50, 130)),   # eight layers, many masked boxes, boxes in front of a camera
])
def test_pipe_box_records_match_the_box_parameter_kernel(name, n_cam, crop, origin):
    """The 48-byte box records and the window headers of the pipelined kernel's geometry pass, decoded on the host, against the
    bit-exact box-parameter entry point (`vfa_box_params_f32`, pinned to the reference's fixtures): visibility, RN(1 / area), the
    upper bilinear fraction of each of the four axes (the frame kernel forms the sixteen tap weights from them with the operations
    of `bilinear_weights`), and the sixteen taps' pixel coordinates rebuilt from the window header -- every layer, view and scale."""
    from vfa_amd import _lib, ops
    dev = _dev()
    wl, grid, lats, calibs = _frame(name, n_cam, crop, dev, origin=origin)
    mods = _mods(wl, dev)
    zl, co = mods[0]._kernel_geometry(dev)
    kind, img_wh = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1]
    sizes = [tuple(l.shape[-2:]) for l in lats]
    n, nl, ns = calibs.shape[0], mods[0].num_grid_layer, 3
    L, W = grid.shape[1:3]
    ws = ops.pipe_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=[m.collapse.weight for m in mods])
    torch.cuda.synchronize()
    host = ws.cpu().numpy()
    lay = ops.pipe_workspace_layout(n, L, W, nl, ns)
    tl_n, tw_n = lay["tiles_l"], lay["tiles_w"]
    tiles = tl_n * tw_n
    seen_direct = seen_shift = 0
    for s, (Hf, Wf) in enumerate(sizes):
        box, area, vis = ops.box_params(calibs, grid.reshape(-1, 3), zl, co, kind, img_wh, (Hf, Wf))
        box = box.cpu().numpy().reshape(n, nl, L, W, 4)
        area = area.cpu().numpy().reshape(n, nl, L, W)
        vis = vis.cpu().numpy().reshape(n, nl, L, W).astype(bool)
        rec = host[lay["recs"][s]:lay["recs"][s] + tiles * nl * n * 32 * 48].view(np.uint32).reshape(tl_n, tw_n, nl, n, 4, 8, 12)
        hdr = host[lay["hdrs"][s]:lay["hdrs"][s] + tiles * nl * n * 32].view(np.uint32).reshape(tl_n, tw_n, nl, n, 8)
        # (tile row, tile col, layer, view, cell row, cell col) -> (view, layer, grid row, grid col)
        g = rec.transpose(3, 2, 0, 4, 1, 5, 6).reshape(n, nl, tl_n * 4, tw_n * 8, 12)[:, :, :L, :W]
        h = np.broadcast_to(hdr.transpose(3, 2, 0, 1, 4)[:, :, :, None, :, None, :], (n, nl, tl_n, 4, tw_n, 8, 8))
        h = h.reshape(n, nl, tl_n * 4, tw_n * 8, 8)[:, :, :L, :W].astype(np.int64)
        assert np.array_equal((g[..., 5] & 1).astype(bool), vis), f"scale {s}: visibility"
        rcp = (np.float32(1) / area).view(np.uint32)
        assert np.array_equal(g[..., 4][vis], rcp[vis]), f"scale {s}: 1 / area"
        assert (g[..., 4][~vis & (area == area)] == 0).all(), f"scale {s}: the factor of a masked box is 0"
        assert np.array_equal(g[..., 10][vis], area.view(np.uint32)[vis]), f"scale {s}: area (the divisor of the exact quotient)"
        # sliver shift of (tile, scale): the largest over views, layers and visible boxes of floor(log2(1/4 + 2^-18 Hf Wf / area)) + 1
        bound = np.float32(0.25) + (np.float32(Hf) * np.float32(Wf) * np.float32(2.0 ** -18)) / area
        sh = np.where(vis & (bound >= 1), np.floor(np.log2(np.maximum(bound, 1).astype(np.float64))).astype(np.int64) + 1, 0)
        pad = np.zeros((n, nl, tl_n * 4, tw_n * 8), np.int64)
        pad[:, :, :L, :W] = sh
        want_shift = pad.reshape(n, nl, tl_n, 4, tw_n, 8).max(axis=(0, 1, 3, 5)).reshape(-1)
        got_shift = host[lay["shifts"][s]:lay["shifts"][s] + tiles * 4].view(np.uint32).astype(np.int64)
        assert np.array_equal(got_shift, want_shift), f"scale {s}: sliver shifts"
        seen_shift += int((want_shift > 0).sum())

        def axis(coord, size):  # vfa_geom.h make_axis: X = fma(g + 1, size / 2, -0.5) (one rounding), i0 = floor(X), hi = X - i0
            x = (np.float32(coord) + np.float32(1)).astype(np.float64) * (size / 2.0) - 0.5
            x = x.astype(np.float32)
            f = np.floor(x)
            return f.astype(np.int64), (x - f).astype(np.float32)

        (xl_i, xl_h), (xr_i, xr_h) = axis(box[..., 0], Wf), axis(box[..., 2], Wf)
        (yt_i, yt_h), (yb_i, yb_h) = axis(box[..., 1], Hf), axis(box[..., 3], Hf)
        for k, (nm, want) in enumerate((("left", xl_h), ("right", xr_h), ("top", yt_h), ("bottom", yb_h))):
            assert np.array_equal(g[..., k][vis], want.view(np.uint32)[vis]), f"scale {s}: upper fraction of the {nm} axis"
        # tap coordinates: the record holds (row part, column part) of the four tap rows / columns, relative to the tile's window
        # (slots) or, for a tile whose window does not fit LDS, as pixels of the padded image
        direct = ((h[..., 0] >> 1) & 1).astype(bool)
        cwid, x0, t0, top, b0 = h[..., 2], h[..., 4].astype(np.int32).astype(np.int64), h[..., 5].astype(np.int32).astype(np.int64), h[..., 6], h[..., 7].astype(np.int32).astype(np.int64)
        rows = np.stack([g[..., 6] & 0xffff, g[..., 6] >> 16, g[..., 7] & 0xffff, g[..., 7] >> 16], -1).astype(np.int64)
        cols = np.stack([g[..., 8] & 0xffff, g[..., 8] >> 16, g[..., 9] & 0xffff, g[..., 9] >> 16], -1).astype(np.int64)
        want_y = np.stack([np.clip(yt_i, -1, Hf), np.clip(yt_i + 1, -1, Hf), np.clip(yb_i, -1, Hf), np.clip(yb_i + 1, -1, Hf)], -1)
        want_x = np.stack([np.clip(xl_i, -1, Wf), np.clip(xl_i + 1, -1, Wf), np.clip(xr_i, -1, Wf), np.clip(xr_i + 1, -1, Wf)], -1)
        sel = vis & direct
        seen_direct += int(sel.sum())
        assert np.array_equal(rows[sel], want_y[sel] + 1) and np.array_equal(cols[sel], want_x[sel] + 1), f"scale {s}: direct taps"
        sel = vis & ~direct
        assert (h[..., 1][sel] <= lay["max_slots"]).all()
        cw = np.maximum(cwid, 1)[..., None]
        srow = rows // cw
        assert np.array_equal((srow * cw)[sel], rows[sel])
        y = np.where(srow < top[..., None], t0[..., None] + srow, b0[..., None] + (srow - top[..., None]))
        assert np.array_equal(y[sel], want_y[sel]) and np.array_equal((cols + x0[..., None])[sel], want_x[sel]), f"scale {s}: window taps"
        assert ((rows + cols)[sel] < h[..., 1][..., None][sel]).all(), f"scale {s}: a tap outside its window"
    print(f"[records] {name}: {seen_direct} visible boxes in tiles pooled straight from L2, {seen_shift} (tile, scale) pairs with a sliver shift")


def test_pipe_balance_minimises_the_heaviest_share(monkeypatch):
    """`vfa_pipe_balance_f32` through `vfa_op.pipe_frame`: the first frame of a geometry leaves bounds of the workgroups' shares in the
    persistent workspace that minimise the heaviest share (estimated cost); every frame of the stream uses them and repeats bit for bit;
    the heaviest share is no heavier than under the uniform split and within one group of the mean; the result stays within the path's
    tolerance of the float64 product; a frame with OTHER cameras does not match the state's signature and still comes out right."""
    from vfa_amd import ops, vfa_op
    dev = _dev()
    monkeypatch.setattr(vfa_op, "PIPE_BALANCE", True)
    vfa_op._pipe_states.clear()
    wl, grid, lats, calibs = _frame("multiviewc_156x156x5", None, None, dev)
    mods = _mods(wl, dev)
    n, nl = calibs.shape[0], mods[0].num_grid_layer
    L, W = grid.shape[1:3]
    lay = ops.pipe_workspace_layout(n, L, W, nl, 3)
    outs = [vfa_op.pipe_frame(mods, lats, calibs, grid).clone() for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    _check("balanced frame", outs[0], _float64_reference(mods, lats, calibs, grid, wl))
    (st,) = vfa_op._pipe_states.values()
    assert st["frames"] == 3
    K = lay["n_chunks"]
    ws = st["ws"]
    state = ws[lay["balance"]:lay["balance"] + 4096].cpu().numpy().view(np.int32)
    chunks = ws[lay["chunks"]:lay["chunks"] + 4 * (K + 3)].cpu().numpy().view(np.int32)
    nblk = int(state[513])
    assert nblk > 0 and nblk % 8 == 0 and tuple(state[514:516]) == tuple(chunks[K + 1:K + 3])
    bounds = state[:nblk + 1].astype(np.int64)
    assert bounds[0] == 0 and bounds[-1] == K and (np.diff(bounds) >= 0).all()
    # estimated cost of a share from the cumulative costs the cuts kernel left per piece (pieces behind the last group: ~0 = the total)
    total = int(np.uint32(chunks[K + 1])) | (int(np.uint32(chunks[K + 2])) << 32)
    costs_off = lay["ranks"] + (4 * (K + 1) + 255) // 256 * 256
    G = ws[costs_off:costs_off + 8 * (K + 1)].cpu().numpy().view(np.uint64).astype(np.float64)
    G[G > 1e19] = total
    assert (np.diff(G) >= 0).all() and G[0] == 0 and G[K] == total
    heaviest = np.diff(G[bounds]).max()
    uniform = np.array([K * k // nblk for k in range(nblk + 1)])
    heaviest_uniform = np.diff(G[uniform]).max()
    mean = total / nblk
    print(f"[balance] {nblk} workgroups: heaviest share / mean {heaviest / mean:.3f} (uniform split: {heaviest_uniform / mean:.3f})")
    assert heaviest <= heaviest_uniform and heaviest / mean < 1.06
    cycles = ws[lay["balance"] + 4096:lay["balance"] + 4096 + 8 * nblk].cpu().numpy().view(np.uint64).astype(np.float64)
    print(f"[balance] last launch: slowest / mean workgroup (cycles) {cycles.max() / cycles[cycles > 0].mean():.3f}")
    # other cameras, same shapes: the state does not fit these cuts
    calibs2 = calibs.flip(0).contiguous()
    lats2 = [l.flip(0).contiguous() for l in lats]
    got = vfa_op.pipe_frame(mods, lats2, calibs2, grid)
    _check("other cameras on a balanced workspace", got, _float64_reference(mods, lats2, calibs2, grid, wl))
    vfa_op._pipe_states.clear()


def test_pipe_work_cuts_match_the_serial_restatement():
    """The chunk tables the device kernel leaves in the workspace against a serial Python restatement of the same rule
    (tests/native/pipe_seq_harness.cpp holds the C++ one and checks the step order on the CPU)."""
    from vfa_amd import _lib, ops
    dev = _dev()
    wl, grid, lats, calibs = _frame("multiviewc_156x156x5", None, (50, 70), dev, origin=(20, 30))
    mods = _mods(wl, dev)
    m0 = mods[0]
    zl, co = m0._kernel_geometry(dev)
    n, nl, ns = calibs.shape[0], m0.num_grid_layer, 3
    L, W = grid.shape[1:3]
    ws = ops.pipe_records(calibs, grid, zl, co, _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1],
                          [tuple(l.shape[-2:]) for l in lats], weights=[m.collapse.weight for m in mods])
    torch.cuda.synchronize()
    lay = ops.pipe_workspace_layout(n, L, W, nl, ns)
    host = ws.cpu().numpy()
    tiles = lay["tiles_l"] * lay["tiles_w"]
    K = lay["n_chunks"]
    live = [host[lay["live"][s]:lay["live"][s] + 4 * tiles].view(np.uint32) for s in range(ns)]
    start = host[lay["chunks"]:lay["chunks"] + 4 * (K + 1)].view(np.int32)
    rank = host[lay["ranks"]:lay["ranks"] + 4 * (K + 1)].view(np.int32)
    assert sum(int(l.sum() > 0) for l in live) == ns

    globs = host[lay["globs"]:lay["globs"] + 4 * tiles].view(np.uint32)

    # tiles of a run (vfa_pipe_seq.h: run_tiles_of, restated): a group takes up to four live (tile, view) sub-tiles of one (run, scale)
    nblk = min(256, tiles)
    nblk = (nblk + 7) // 8 * 8
    steps = 2 * nl * n * ns * tiles // nblk
    RUN = (4 if steps >= 250 else 1) if n <= 2 else (4 if steps >= 1600 else (2 if steps >= 1200 else 1))
    forced = os.environ.get("VFA_AMD_PIPE_RT")
    RUN = int(forced) if forced in ("1", "2", "4") else RUN
    runs = (tiles + RUN - 1) // RUN

    # cost of a sub-tile over its layers, as pipe_records_kernel adds it up from the headers it writes (vfa_pipe_seq.h: sub_layer_cost):
    # live 366 + 2 per window slot (or + 500 when pooled from L2), without a live box 188; units of 16 cycles
    hdrs = [host[lay["hdrs"][s]:lay["hdrs"][s] + tiles * nl * n * 32].view(np.uint32).reshape(tiles, nl, n, 8) for s in range(ns)]

    def subcost(s, t, v):
        h = hdrs[s][t, :, v]
        lv, direct = (h[:, 0] & 1) == 1, (h[:, 0] & 2) == 2
        return int(np.where(lv, 366 + np.where(direct, 500, 2 * h[:, 1].astype(np.int64)), 188).sum())

    def groups_of(r):  # costs of the groups of run r in (scale, tile, view) order (vfa_pipe_seq.h: walk_run)
        out = []
        ts = range(r * RUN, min(tiles, (r + 1) * RUN))
        for s in range(ns):
            subs = [(t, v) for t in ts for v in range(n) if (int(live[s][t]) >> v) & 1]
            for g0 in range(0, len(subs), 4):
                grp = subs[g0:g0 + 4]
                out.append(nl * (625 + (100 if len(grp) <= 2 else 0)) + 9 + sum(subcost(s, t, v) for t, v in grp) + (144 * len(ts) if not out else 0))
        return out

    costs = [groups_of(r) for r in range(runs)]
    weight = [sum(c) if c else 16 * min(RUN, tiles - r * RUN) for r, c in enumerate(costs)]
    before = np.concatenate([[0], np.cumsum(weight)])
    total = int(before[-1])
    exp_start, exp_rank = np.full(K + 1, runs, np.int32), np.zeros(K + 1, np.int32)
    c = 0
    for t in range(runs):
        tb, w0 = int(before[t]), 0
        for k, wi in enumerate(costs[t]):
            while c < K:
                pc = (total * c + K - 1) // K
                if pc >= tb + w0 + wi:
                    break
                kk = k if (pc - tb - w0) * 2 < wi else k + 1
                if kk >= len(costs[t]):
                    exp_start[c], exp_rank[c] = (t + 1 if t + 1 < runs else runs), 0
                else:
                    exp_start[c], exp_rank[c] = t, kk
                c += 1
            w0 += wi
        if not costs[t]:
            while c < K and (total * c + K - 1) // K < tb + weight[t]:
                exp_start[c], exp_rank[c] = t, 0
                c += 1
    assert np.array_equal(start, exp_start) and np.array_equal(rank, exp_rank)


@pytest.mark.parametrize("name,n_cam,crop,origin", [
    ("multiviewc_200x200x1", 7, (48, 96), (70, 50)),      # K = 256
    ("multiviewc_156x156x5", 5, (32, 56), (60, 40)),      # K = 1280
])
def test_three_piece_product_has_the_width_of_the_reference_sgemm(name, n_cam, crop, origin):
    """VFA_FLAG_TERMS 6: three bf16 pieces per operand, six products.  The reference's ``collapse`` is an fp32 ``nn.Linear``
    (vfa_op.py:123); an fp32 sgemm sits ~3e-7 (normwise) from float64 at these shapes.  The three-piece product must be in that
    class -- <= 5e-7 -- an order of magnitude below the default two-piece / three-product arithmetic (~3e-6), on ordinary,
    SIGNED and HEAVY-TAILED feature maps (the split loses nothing to cancellation or dynamic range)."""
    from vfa_amd import vfa_op
    dev = _dev()
    wl, grid, lats, calibs = _frame(name, n_cam, crop, dev, origin=origin)
    mods = _mods(wl, dev)
    gen = torch.Generator().manual_seed(11)
    variants = {
        "relu(randn)": lats,
        "signed": [l - 0.4 for l in lats],
        "heavy-tailed": [l * (torch.exp(2.5 * torch.randn(l.shape, generator=gen)) * torch.sign(torch.randn(l.shape, generator=gen))).to(dev)
                         for l in lats],
    }
    for label, feats in variants.items():
        with torch.no_grad():
            want = _float64_reference(mods, feats, calibs, grid, wl)
            got6 = vfa_op.pipe_frame(mods, feats, calibs, grid, terms=6)
            got3 = vfa_op.pipe_frame(mods, feats, calibs, grid, terms=3)
        scale = want.abs().max().item()
        e6 = ((got6.double() - want).norm() / want.norm()).item()
        e3 = ((got3.double() - want).norm() / want.norm()).item()
        m6 = (got6.double() - want).abs().max().item() / scale
        m3 = (got3.double() - want).abs().max().item() / scale
        print(f"[width] {name} {label}: three pieces normwise {e6:.2e} (max {m6:.2e} of max|out|), two pieces {e3:.2e} (max {m3:.2e})")
        assert torch.isfinite(got6).all()
        assert e6 <= 5e-7, (label, e6)
        assert m6 <= 2e-6, (label, m6)
        _check(f"{name} {label} two pieces", got3, want)


def _sgemm_reference(mods, lats, calibs, grid, wl):
    """The same sum with the product as a plain fp32 library GEMM on the bit-pinned voxel features: what the reference's fp32
    nn.Linear (vfa_op.py:123) is, numerically."""
    from vfa_amd import _lib, ops
    dev = grid.device
    n = calibs.shape[0]
    grid_flat = grid.reshape(-1, 3).contiguous()
    out = torch.zeros(grid_flat.shape[0], 256, dtype=torch.float32, device=dev)
    for m, lat in zip(mods, lats):
        zl, co = m._kernel_geometry(dev)
        vox = ops.project_gather(ops.integral_image(lat), calibs.reshape(n, 12).contiguous(), grid_flat, zl, co,
                                 _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], kernel="direct")
        out += torch.relu(torch.matmul(vox, m.layer_major_weight().T) + m.collapse.bias).sum(0)
    return out


@pytest.mark.parametrize("name,n_cam,crop,origin", [
    ("multiviewc_200x200x1", 7, (48, 96), (70, 50)),      # K = 256
    ("multiviewc_156x156x5", 5, (32, 56), (60, 40)),      # K = 1280
])
def test_fp16_split_product_has_the_width_of_the_reference_sgemm(name, n_cam, crop, origin):
    """VFA_FLAG_TERMS 2, the default: two fp16 pieces per operand with a power-of-two scale, three products (vfa_split.h).  The
    reference's ``collapse`` is an fp32 ``nn.Linear`` (vfa_op.py:123).  Against float64 the default product must sit where an fp32
    library GEMM on the same voxel features sits (<= 1.3 x its normwise error, or 3e-7), on ordinary and signed maps and on maps
    scaled by 1e4 and 1e-6 (the scale follows the feature maximum); on heavy-tailed maps (dynamic range 2^17 inside one map) it stays
    inside 1e-6 normwise and far inside the path's tolerance.  The two-piece bf16 form (terms 3) is an order of magnitude wider."""
    from vfa_amd import vfa_op
    dev = _dev()
    wl, grid, lats, calibs = _frame(name, n_cam, crop, dev, origin=origin)
    mods = _mods(wl, dev)
    gen = torch.Generator().manual_seed(11)
    variants = {
        "relu(randn)": lats,
        "signed": [l - 0.4 for l in lats],
        "x 1e4": [l * 1e4 for l in lats],
        "x 1e-6": [l * 1e-6 for l in lats],
        "heavy-tailed": [l * (torch.exp(2.5 * torch.randn(l.shape, generator=gen)) * torch.sign(torch.randn(l.shape, generator=gen))).to(dev)
                         for l in lats],
    }
    for label, feats in variants.items():
        if label == "x 1e-6":  # (the bias would drown the product: leave it out of this variant)
            saved = [m.collapse.bias.detach().clone() for m in mods]
            with torch.no_grad():
                for m in mods:
                    m.collapse.bias.zero_()
        with torch.no_grad():
            want = _float64_reference(mods, feats, calibs, grid, wl)
            lib32 = _sgemm_reference(mods, feats, calibs, grid, wl)
            got2 = vfa_op.pipe_frame(mods, feats, calibs, grid, terms=2)
            got3 = vfa_op.pipe_frame(mods, feats, calibs, grid, terms=3)
        if label == "x 1e-6":
            with torch.no_grad():
                for m, b in zip(mods, saved):
                    m.collapse.bias.copy_(b)
        scale = want.abs().max().item()
        e2 = ((got2.double() - want).norm() / want.norm()).item()
        e3 = ((got3.double() - want).norm() / want.norm()).item()
        e32 = ((lib32.double() - want).norm() / want.norm()).item()
        m2 = (got2.double() - want).abs().max().item() / scale
        print(f"[width] {name} {label}: fp16 x 2 normwise {e2:.2e} (max {m2:.2e} of max|out|), fp32 library GEMM {e32:.2e}, bf16 x 2 {e3:.2e}")
        assert torch.isfinite(got2).all()
        if label == "heavy-tailed":
            assert e2 <= 1e-6, (label, e2)
        else:
            assert e2 <= max(1.3 * e32, 3e-7), (label, e2, e32)
            assert e3 >= 3 * e2, (label, e2, e3)
        _check(f"{name} {label} fp16 x 2", got2, want)


def test_fp16_split_saturates_instead_of_overflowing_and_keeps_nan():
    """vfa_split.h: a voxel feature beyond fp16's range after scaling (here: feature maps whose statistics are deliberately those of
    a map 1e6 times smaller) must come out finite (MODE.FP16_OVFL: +-65504 instead of Inf - Inf = NaN), a NaN feature stays a NaN in
    the cells that see it and nowhere else."""
    from vfa_amd import ops, vfa_op
    dev = _dev()
    wl, grid, lats, calibs = _frame("multiviewc_156x156x5", 3, (16, 24), dev, origin=(60, 40))
    mods = _mods(wl, dev)
    with torch.no_grad():
        good = vfa_op.pipe_frame(mods, lats, calibs, grid)
        integrals = ops.integral_images(lats)
        tiny = ops.integral_images([l * 1e-6 for l in lats])
        integrals.absmax = tiny.absmax  # the scale of a map a million times smaller: every voxel feature overflows fp16
        sat = vfa_op.pipe_frame(mods, None, calibs, grid, integrals=integrals)
        assert torch.isfinite(sat).all()
        assert not torch.allclose(sat, good, rtol=1e-2, atol=0.0)  # (and it IS wrong: the statistics must belong to the maps)
        poisoned = [l.clone() for l in lats]
        poisoned[0][0, 3, 0, 0] = float("nan")  # (the double cumsum spreads it over the whole of channel 3 of camera 0: as in the reference)
        out = vfa_op.pipe_frame(mods, poisoned, calibs, grid)
        want = _float64_reference(mods, poisoned, calibs, grid, wl)
        bad = torch.isnan(want).any(dim=1)
        assert bad.any(), "camera 0 should see cells of the crop"
        assert torch.equal(torch.isnan(out), torch.isnan(want))
        if not bad.all():
            _check("cells that do not see the NaN", out[~bad], want[~bad])


@pytest.mark.parametrize("name", ["multiviewc_156x156x5", "multiviewc_200x200x1"])
def test_captured_frames_keep_a_workspace_of_their_own(name):
    """hipGraph capture of a frame (``vfa_amd.graph.GraphedAggregate``: the pipelined kernel on the five-layer grid, the serial
    kernel on the single-layer one).  A captured frame replays into the geometry workspace it was captured with, so that workspace
    must never be handed to anybody else: two graphs of the SAME geometry, replayed alternately with eager frames of that geometry in
    between (on the capture streams' handles too: torch recycles them), every result bit for bit the eager one; the states a capture
    pinned are not in the eager cache and no two owners share a workspace."""
    import vfa_amd
    from vfa_amd import vfa_op
    from vfa_amd.graph import GraphedAggregate
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload(name, channels=256, seed=6, n_cam=3, device=dev)
    grid = wl["grid"][:, 8:56, 4:44].contiguous()
    torch.manual_seed(2)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    lats_a = [torch.cat([wl["features"][c][s] for c in range(3)]) for s in range(3)]
    lats_b = [torch.relu(torch.randn_like(l)) for l in lats_a]
    calibs = wl["calibs"]
    vfa_op._pipe_states.clear()
    pinned_before = len(vfa_op._pipe_pinned)
    with torch.no_grad():
        want_a = vfa_amd.aggregate_views(*mods, *lats_a, calibs, grid).clone()
        want_b = vfa_amd.aggregate_views(*mods, *lats_b, calibs, grid).clone()
    g1 = GraphedAggregate(*mods, *lats_a, calibs, grid)
    g2 = GraphedAggregate(*mods, *lats_b, calibs, grid)
    streams = [torch.cuda.Stream(device=dev) for _ in range(40)]  # (more than torch's pool of 32 handles: some repeat the captures')
    with torch.no_grad():
        for rnd in range(3):
            assert torch.equal(g1(*lats_a), want_a), rnd
            with torch.cuda.stream(streams[(7 * rnd) % 40]):
                eager = vfa_amd.aggregate_views(*mods, *lats_b, calibs, grid)
            torch.cuda.synchronize()
            assert torch.equal(eager, want_b), rnd
            assert torch.equal(g2(*lats_b), want_b), rnd
            for st in streams[rnd::3]:
                with torch.cuda.stream(st):
                    eager = vfa_amd.aggregate_views(*mods, *lats_a, calibs, grid)
            torch.cuda.synchronize()
            assert torch.equal(eager, want_a), rnd
            assert torch.equal(g1(*lats_b), want_b) and torch.equal(g2(*lats_a), want_a), rnd
    if mods[0].num_grid_layer > 1:  # (the pipelined path keeps persistent states; the serial path allocates per frame)
        # the workspaces belong to the graph objects (round-5 advisor finding: the process-wide list grew with every capture) ...
        assert len(vfa_op._pipe_pinned) == pinned_before
        pinned = g1._states.states + g2._states.states
        assert len(pinned) == 2 and all(st["pinned"] for st in pinned)
        assert not any(st.get("pinned") for st in vfa_op._pipe_states.values())
        owners = [st["ws"].data_ptr() for st in pinned] + [st["ws"].data_ptr() for st in vfa_op._pipe_states.values()]
        assert len(set(owners)) == len(owners)
        # ... two frames of one geometry inside ONE capture share a workspace (and the balanced shares of the warm-up frame) ...
        g3 = torch.cuda.CUDAGraph()
        with torch.no_grad():
            with vfa_op.owned_capture_states() as keep, torch.cuda.graph(g3):
                o1 = vfa_amd.aggregate_views(*mods, *lats_a, calibs, grid)
                o2 = vfa_amd.aggregate_views(*mods, *lats_b, calibs, grid)
            g3.replay()
            torch.cuda.synchronize()
        assert len(keep.states) == 1 and keep.states[0]["frames"] > 0
        assert torch.equal(o1, want_a) and torch.equal(o2, want_b)
        # ... and a bare capture (no owner named) still gets a workspace nobody else touches, kept by the process
        g4 = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g4):
            o4 = vfa_amd.aggregate_views(*mods, *lats_a, calibs, grid)
        g4.replay()
        torch.cuda.synchronize()
        assert len(vfa_op._pipe_pinned) == pinned_before + 1 and torch.equal(o4, want_a)
        del vfa_op._pipe_pinned[pinned_before:]
