"""The two-launch inference path of single-layer grids (``vfa_frame_records_f32`` + ``vfa_pool_collapse_relu_sum_f32``:
geometry once per frame, pooling + collapse + ReLU + view / scale sums in one persistent kernel, vox never in HBM) against
the unfused kernels, the CPU oracle and float64.   ``-m gpu``.

Reference lines covered: vfa/model/vfa_op.py:61-125 (all of VFA.forward), vfa/model/vfanet.py:64-82 (camera loop).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL_REL = 1e-4, 1e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _mods(wl, dev, seed=1, scale=3.0):
    import vfa_amd
    torch.manual_seed(seed)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
            for _ in range(3)]
    with torch.no_grad():
        for m in mods:  # bigger weights and a negative-leaning bias: the ReLU cuts a real share of the outputs
            m.collapse.weight.mul_(scale)
            m.collapse.bias.uniform_(-0.3, 0.1)
    return mods


def _float64_reference(mods, lats, calibs, grid, wl, cells=None):
    """sum_scale sum_view relu(vox . W^T + b) in float64 from the BITWISE-pinned voxel features of the pooling kernel."""
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
        want += torch.relu(vox.double() @ m.collapse.weight.double().T + m.collapse.bias.double()).sum(0)
    return want


def _check(name, got, want):
    scale = want.abs().max().item()
    assert scale > 0
    tol = RTOL * want.abs() + ATOL_REL * scale
    worst = ((got.double() - want).abs() / tol).max().item()
    print(f"[margin] {name}: worst |err| / tolerance = {worst:.3f}")
    torch.testing.assert_close(got.double(), want, rtol=RTOL, atol=ATOL_REL * scale, msg=lambda m: f"{name}: {m}")


CASES = [  # workload, cameras, grid crop (rows, cols) or None
    ("multiviewc_200x200x1", None, None),          # the bench frame: 7 cameras x 3 scales, 1250 full tiles
    ("multiviewc_200x200x1", 2, (37, 53)),         # ragged grid: partial tiles on both edges
    ("wildtrack_480x1440x1", 3, (100, 1440)),      # Wildtrack conversion, 1080p maps
    ("multiviewc_200x200x1", 11, (45, 64)),        # more than 8 cameras: two view groups per tile in the work-cut kernel
]


@pytest.mark.parametrize("name,n_cam,crop", CASES)
def test_fused_frame_matches_unfused_kernels_and_float64(name, n_cam, crop, monkeypatch):
    import vfa_amd
    from vfa_amd import ops, vfa_op
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload(name, channels=256, seed=3, **({"n_cam": n_cam} if n_cam else {}))
    n = wl["n_cam"]
    grid = wl["grid"] if crop is None else wl["grid"][:, 11:11 + crop[0], 5:5 + crop[1]].contiguous()
    grid = grid.to(dev)
    mods = _mods(wl, dev)
    lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
    calibs = wl["calibs"].to(dev)
    L, W = grid.shape[1:3]
    with torch.no_grad():
        monkeypatch.setattr(vfa_op, "FUSED_POOL", True)
        with ops.KernelTimer() as kt:
            fused = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        assert "vfa_pool_collapse_relu_sum_f32" in kt.summary() and "vfa_project_gather_f32" not in kt.summary(), sorted(kt.summary())
        monkeypatch.setattr(vfa_op, "FUSED_POOL", False)
        monkeypatch.setattr(vfa_op, "WINDOW_POOL", False)
        with ops.KernelTimer() as kt:
            unfused = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        assert "vfa_collapse_relu_sum_f32" in kt.summary() and "vfa_pool_collapse_relu_sum_f32" not in kt.summary()
        cells = min(L * W, 30000)
        want = _float64_reference(mods, lats, calibs, grid, wl, cells)
    f = fused[0].permute(1, 2, 0).reshape(L * W, 256)
    u = unfused[0].permute(1, 2, 0).reshape(L * W, 256)
    assert torch.isfinite(f).all()
    _check(f"{name} fused vs float64", f[:cells], want)
    _check(f"{name} unfused vs float64", u[:cells], want)
    _check(f"{name} fused vs unfused", f, u.double())
    # same frame twice: the persistent kernel writes every output row exactly once, deterministically
    with torch.no_grad():
        monkeypatch.setattr(vfa_op, "FUSED_POOL", True)
        again = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
    assert torch.equal(again, fused)


@pytest.mark.parametrize("data,image_size,cube,gh,world,step,cam", [
    ("MultiviewC", (720, 1280), (18.75, 18.75, 160), 160, (3750, 3750), (75.0, 75.0), "ring"),
    ("MultiviewX", (1080, 1920), (4, 4, 64), 64, (640, 1000), (8, 8), "mx"),
    ("Wildtrack", (1080, 1920), (4, 4, 32), 32, (480, 1440), (12, 12), "wt"),
    ("MultiviewC", (720, 1280), (150.0, 150.0, 300), 300, (3750, 3750), (150.0, 150.0), "inside"),  # huge near boxes: direct tiles
])
def test_fused_frame_vs_oracle_small_scenes(oracle, data, image_size, cube, gh, world, step, cam):
    """Single-layer scenes of every dataset conversion against the CPU oracle end to end (oracle voxel features, float64
    product): cameras far, near and INSIDE the field (boxes behind the camera, tiles whose tap window does not fit LDS and
    take the direct path, fully masked tiles and views)."""
    import vfa_amd
    from vfa_amd.synthetic import look_at_camera, ring_cameras
    from vfa_amd.utils import make_grid
    dev = _dev()
    H, W_img = image_size
    if cam == "ring":
        calibs = ring_cameras(3, (1875.0, 1875.0, 0.0), 2700.0, 600.0, 900.0, (W_img, H))
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
        calibs = ring_cameras(2, wt_c, 0.45 * 1440 * 2.5, 400.0, 1100.0, (W_img, H))
    grid = make_grid(world_size=world, cube_LW=list(step), dataset=data)
    args = SimpleNamespace(data=data, image_size=image_size)
    n = calibs.shape[0]
    gen = torch.Generator().manual_seed(7)
    sizes = [(45, 80), (23, 40), (12, 20)]
    lats = [torch.relu(torch.randn(n, 256, h, w, generator=gen)) for h, w in sizes]
    torch.manual_seed(3)
    mods = [vfa_amd.VFA(256, grid_height=gh, cube_size=cube, args=args).to(dev) for _ in range(3)]
    with torch.no_grad():
        for m in mods:
            m.collapse.weight.mul_(3.0)
            m.collapse.bias.uniform_(-0.3, 0.1)
        with vfa_amd.ops.KernelTimer() as kt:
            out = vfa_amd.aggregate_views(*mods, *(l.to(dev) for l in lats), calibs.to(dev), grid.to(dev)[None])
        torch.cuda.synchronize()
    assert "vfa_pool_collapse_relu_sum_f32" in kt.summary()
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
    _check(f"{data}/{cam}", got, torch.from_numpy(want))


def test_fused_frame_single_scale_accumulate_and_degenerate_grids():
    """`VFA.forward` (one camera, one scale -- the reference's own interface) runs the same two launches; accumulate on top of
    an existing map; grids smaller than one 8 x 4 tile; a view that sees nothing."""
    import vfa_amd
    from vfa_amd import ops, vfa_op
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=5, n_cam=2)
    mods = _mods(wl, dev, seed=2)
    lat = wl["features"][0][1].to(dev)
    calib = wl["calibs"][0].to(dev)
    for rows, cols in ((3, 5), (4, 8), (1, 1), (9, 17)):
        grid = wl["grid"][:, 90:90 + rows, 100:100 + cols].contiguous().to(dev)
        with torch.no_grad(), ops.KernelTimer() as kt:
            out = vfa_amd.materialize(mods[1](lat, calib, grid))  # (inference results are deferred: vfa_amd/lazy.py)
        torch.cuda.synchronize()
        assert "vfa_pool_collapse_relu_sum_f32" in kt.summary()
        want = _float64_reference([mods[1]], [lat], calib[None], grid, wl)
        _check(f"forward {rows}x{cols}", out[0].permute(1, 2, 0).reshape(rows * cols, 256), want)
    # accumulate
    grid = wl["grid"][:, 40:61, 30:70].contiguous().to(dev)
    base = torch.randn(21 * 40, 256, device=dev)
    with torch.no_grad():
        got = vfa_op.fused_frame([mods[0]], [wl["features"][1][0].to(dev)], wl["calibs"][1:2].to(dev), grid, out=base.clone(),
                                 accumulate=True)
        want = base.double() + _float64_reference([mods[0]], [wl["features"][1][0].to(dev)], wl["calibs"][1:2].to(dev), grid, wl)
    _check("accumulate", got, want)
    # a camera looking away: every box masked -> every output row is relu(bias) (vox = 0)
    away = torch.tensor([[900., 0, 640, -1e9], [0, 900., 360, -1e9], [0, 0, 0., 1.]], device=dev)  # every corner clamps to -1
    with torch.no_grad():
        out = mods[2](wl["features"][0][2].to(dev), away, grid)
    want = torch.relu(mods[2].collapse.bias.detach()).expand(21 * 40, 256)
    assert torch.equal(out[0].permute(1, 2, 0).reshape(-1, 256), want)


def test_frame_records_match_the_box_parameter_kernel():
    """The per-frame geometry pass against the bit-exact box-parameter entry point: visibility, area and 1 / area of every
    record, scale by scale (the tap weights and coordinates are covered end to end above)."""
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import make_workload
    import vfa_amd
    dev = _dev()
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=0, n_cam=3)
    grid = wl["grid"][:, 20:61, 10:77].contiguous().to(dev)  # 41 x 67: ragged
    L, W = grid.shape[1:3]
    mod = vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
    zl, co = mod._kernel_geometry(dev)
    kind = _lib.CONV_KIND[wl["args"].data]
    img_wh = wl["args"].image_size[::-1]
    sizes = [tuple(s) for s in wl["feat_sizes"]]
    calibs = wl["calibs"].to(dev)
    ws = ops.frame_records(calibs, grid, zl, co, kind, img_wh, sizes).cpu().numpy()
    n = 3
    lay = ops.frame_workspace_layout(n, L, W, 3)
    tiles_l, tiles_w = lay["tiles_l"], lay["tiles_w"]
    assert (tiles_l, tiles_w) == ((L + 3) // 4, (W + 7) // 8) and lay["total"] <= ws.size
    n_tiles = tiles_l * tiles_w
    for s, (Hf, Wf) in enumerate(sizes):
        live_off, direct_off, hdr_off, rec_off = lay["live"][s], lay["direct"][s], lay["hdrs"][s], lay["recs"][s]
        rec = ws[rec_off:rec_off + n * n_tiles * 32 * 96].view(np.uint32).reshape(n, tiles_l, tiles_w, 4, 8, 24)
        box, area, vis = ops.box_params(calibs, grid.reshape(-1, 3), zl, co, kind, img_wh, (Hf, Wf))
        area = area.cpu().numpy().reshape(n, L, W)
        vis = vis.cpu().numpy().reshape(n, L, W).astype(bool)
        # tile-major -> grid order
        g = rec.transpose(0, 1, 3, 2, 4, 5).reshape(n, tiles_l * 4, tiles_w * 8, 24)[:, :L, :W]
        assert np.array_equal((g[..., 17] & 1).astype(bool), vis), f"scale {s}: visibility"
        assert np.array_equal(g[..., 23], area.view(np.uint32)), f"scale {s}: area"
        rcp = (np.float32(1) / area).view(np.uint32)
        assert np.array_equal(g[..., 16][vis], rcp[vis]), f"scale {s}: 1 / area"
        live = ws[live_off:live_off + n_tiles * 4].view(np.uint32)
        tile_vis = np.zeros((n, tiles_l * 4, tiles_w * 8), bool)
        tile_vis[:, :L, :W] = vis
        tile_any = tile_vis.reshape(n, tiles_l, 4, tiles_w, 8).any(axis=(2, 4)).reshape(n, n_tiles)
        want_live = sum((tile_any[v].astype(np.uint32) << v) for v in range(n))
        assert np.array_equal(live, want_live), f"scale {s}: live-view masks"
        # items whose tap window exceeds the LDS capacity are flagged in the header and in the direct mask, nowhere else
        hdr = ws[hdr_off:hdr_off + n * n_tiles * 32].view(np.uint32).reshape(n, n_tiles, 8)
        is_direct = ((hdr[..., 0] >> 1) & 1).astype(bool) & (hdr[..., 0] & 1).astype(bool)
        assert ((hdr[..., 1][~is_direct]) <= lay["max_slots"]).all()
        direct = ws[direct_off:direct_off + n_tiles * 4].view(np.uint32)
        assert np.array_equal(direct, sum((is_direct[v].astype(np.uint32) << v) for v in range(n))), f"scale {s}: direct masks"
        assert (direct & ~live == 0).all()


@pytest.mark.parametrize("name,n_cam,crop", [("multiviewc_200x200x1", None, None), ("multiviewc_200x200x1", 2, (37, 53)),
                                             ("wildtrack_480x1440x1", 2, (64, 1440))])
def test_window_pooling_kernel_is_bitwise_the_direct_kernel(name, n_cam, crop):
    """`vfa_pool_windows_f32` (LDS tap windows from the per-frame records) against `vfa_project_gather_f32` (which is pinned
    bitwise to the reference's voxel features): every scale, every camera, ragged grids, tiles whose window does not fit."""
    import vfa_amd
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload(name, channels=256, seed=6, **({"n_cam": n_cam} if n_cam else {}))
    n = wl["n_cam"]
    grid = wl["grid"] if crop is None else wl["grid"][:, 7:7 + crop[0], 3:3 + crop[1]].contiguous()
    grid = grid.to(dev)
    L, W = grid.shape[1:3]
    mod = vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
    zl, co = mod._kernel_geometry(dev)
    kind = _lib.CONV_KIND[wl["args"].data]
    img_wh = wl["args"].image_size[::-1]
    calibs = wl["calibs"].to(dev)
    lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
    ws = ops.frame_records(calibs, grid, zl, co, kind, img_wh, [tuple(l.shape[-2:]) for l in lats])
    for k, lat in enumerate(lats):
        integral = ops.integral_image(lat)
        got = ops.pool_windows(integral, ws, (L, W), 3, k)
        want = ops.project_gather(integral, calibs.reshape(n, 12).contiguous(), grid.reshape(-1, 3).contiguous(), zl, co, kind,
                                  img_wh, kernel="direct")
        same = (got.view(torch.int32) == want.view(torch.int32)) | ((got == 0) & (want == 0))
        assert bool(same.all()), f"{name} scale {k}: {int((~same).sum())} voxel features differ"
        assert float(want.abs().max()) > 0


def test_window_path_is_used_when_the_fused_kernel_is_off(monkeypatch):
    import vfa_amd
    from vfa_amd import ops, vfa_op
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=1, n_cam=3)
    grid = wl["grid"][:, 50:90, 20:100].contiguous().to(dev)
    mods = _mods(wl, dev)
    lats = [torch.cat([wl["features"][c][s] for c in range(3)]).to(dev) for s in range(3)]
    calibs = wl["calibs"].to(dev)
    outs = {}
    for fused, window in ((True, True), (False, True), (False, False)):
        monkeypatch.setattr(vfa_op, "FUSED_POOL", fused)
        monkeypatch.setattr(vfa_op, "WINDOW_POOL", window)
        with torch.no_grad(), ops.KernelTimer() as kt:
            outs[(fused, window)] = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        names = set(kt.summary())
        assert ("vfa_pool_collapse_relu_sum_f32" in names) == fused
        assert ("vfa_pool_windows_f32" in names) == (window and not fused)
        assert ("vfa_project_gather_f32" in names) == (not fused and not window)
    # window path and legacy path share the collapse kernel and bit-identical voxel features: identical maps
    assert torch.equal(outs[(False, True)], outs[(False, False)])
    scale = outs[(False, False)].abs().max().item()
    torch.testing.assert_close(outs[(True, True)], outs[(False, False)], rtol=RTOL, atol=2 * ATOL_REL * scale)


def test_producer_fusion_integral_is_bitwise_and_vfanet_agrees(monkeypatch):
    """GroupNorm affine + ReLU applied inside the integral-image row scan.  (1) the kernel against
    the unfused pair (relu(x * scale + shift) with two rounded fp32 operations, then `vfa_integral_image_f32`): BITWISE;
    (2) a whole `VFANet` forward with the fusion on and off: the same maps within the path's tolerance (the statistics are
    formed by other kernels than torch's GroupNorm)."""
    from types import SimpleNamespace
    from vfa_amd import ops, vfanet
    from vfa_amd.synthetic import ring_cameras
    from vfa_amd.utils import make_grid
    dev = _dev()
    gen = torch.Generator().manual_seed(9)
    for n, C, H, W in ((2, 256, 23, 40), (1, 256, 45, 80), (3, 70, 9, 13)):
        x = torch.randn(n, C, H, W, generator=gen).to(dev)
        scale = (torch.rand(n, C, generator=gen) * 2 + 0.1).to(dev)
        shift = (torch.randn(n, C, generator=gen) * 0.5).to(dev)
        got = ops.affine_relu_integral_image(x, scale, shift)
        lat = torch.relu(x * scale[:, :, None, None] + shift[:, :, None, None])
        want = ops.integral_image(lat)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (n, C, H, W)
    args = SimpleNamespace(data="MultiviewC", image_size=(720, 1280))
    torch.manual_seed(4)
    net = vfanet.VFANet(args, grid_height=160, cube_size=(18.75, 18.75, 160), mode="2D").to(dev).eval()
    images = torch.rand(3, 3, 96, 160, generator=gen).to(dev)
    calibs = ring_cameras(3, (1875.0, 1875.0, 0.0), 2700.0, 600.0, 900.0, (1280, 720)).to(dev)
    grid = make_grid(world_size=(3750, 3750), cube_LW=[125, 150], dataset="MultiviewC")[None].to(dev)
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(vfanet, "FUSE_PRODUCER", fuse)
        with torch.no_grad(), ops.KernelTimer() as kt:
            outs.append(net.ortho_features(images, calibs, grid))
        torch.cuda.synchronize()
        ran = kt.summary()
        if fuse:  # hand-written conv (statistics in its epilogue) + the channels-last row scan with affine + ReLU
            assert "vfa_lateral_convs_f32" in ran and all(tag[-1] for tag in ran["vfa_integral_images_hwc_f32"]["by_tag"]), sorted(ran)
        else:     # library conv + GroupNorm + ReLU, plain integral images
            assert "vfa_lateral_convs_f32" not in ran and not any(tag[-1] for tag in ran["vfa_integral_images_f32"]["by_tag"])
    scale = outs[1].abs().max().item()
    assert scale > 0
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-3, atol=1e-4 * scale)


@pytest.mark.parametrize("shapes,n,C", [(((90, 160), (45, 80), (23, 40)), 2, 256),   # the bench frame: tails of 0, 16 and 8 columns
                                        (((7, 44), (3, 4)), 3, 128),                  # a 12-column tail, a one-chunk map
                                        (((9, 13), (5, 8)), 2, 70),                   # odd shapes: the per-map kernels
                                        (((12, 36),), 1, 64)])
def test_batched_integral_images_are_bitwise_the_per_map_ones(shapes, n, C):
    """`vfa_integral_images_f32` (all strides of a frame in one launch pair, loads one chunk ahead) against
    `vfa_integral_image_f32` / `vfa_affine_relu_integral_image_f32` per map -- which the oracle pins: BITWISE, borders included,
    with heavy-tailed inputs (sums that are inexact in double would expose any re-association)."""
    from vfa_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(len(shapes) * 100 + C)
    feats = []
    for H, W in shapes:
        f = torch.randn(n, C, H, W, generator=gen)
        f = f * torch.exp(6 * torch.randn(n, C, H, W, generator=gen))  # magnitudes over ~15 decades
        feats.append(f.to(dev))
    got = ops.integral_images(feats)
    for f, g in zip(feats, got):
        want = ops.integral_image(f)
        assert g.shape == want.shape and torch.equal(g.view(torch.int32), want.view(torch.int32)), tuple(f.shape)
    scales = [(torch.rand(n, C, generator=gen) * 2 + 0.1).to(dev) for _ in shapes]
    shifts = [(torch.randn(n, C, generator=gen) * 0.5).to(dev) for _ in shapes]
    got = ops.integral_images(feats, scales, shifts)
    for f, sc, sh, g in zip(feats, scales, shifts, got):
        want = ops.affine_relu_integral_image(f, sc, sh)
        assert torch.equal(g.view(torch.int32), want.view(torch.int32)), tuple(f.shape)


@pytest.mark.parametrize("shapes,n,C", [(((90, 160), (45, 80), (23, 40)), 7, 256),   # the bench frame: two column parts on the wide map
                                        (((40, 300), (9, 36)), 4, 256),               # ten strips with a 12-column tail, a two-strip map
                                        (((33, 64), (8, 4)), 8, 128)])                # one row past a batch; a map of one narrow strip
def test_one_pass_integral_images_are_bitwise_the_per_map_ones(shapes, n, C):
    """`integral_onepass_kernel` (frames of >= 64 (view, 16-channel block) units: the case above stays on the two-pass kernels) against
    the per-map kernels the oracle pins -- BITWISE, borders included, and the SAME feature statistic; heavy-tailed inputs (double
    sums that round would expose any re-association); plain and with the fused affine + ReLU; twice (nothing carried over)."""
    from vfa_amd import ops
    dev = _dev()
    assert n * (C // 16) >= 64
    gen = torch.Generator().manual_seed(len(shapes) * 1000 + C + n)
    feats = []
    for H, W in shapes:
        f = torch.randn(n, C, H, W, generator=gen)
        f = f * torch.exp(6 * torch.randn(n, C, H, W, generator=gen))
        feats.append(f.to(dev))
    scales = [(torch.rand(n, C, generator=gen) * 2 + 0.1).to(dev) for _ in shapes]
    shifts = [(torch.randn(n, C, generator=gen) * 0.5).to(dev) for _ in shapes]
    for affine in (False, True):
        sc, sh = (scales, shifts) if affine else (None, None)
        want = [ops.affine_relu_integral_image(f, a, b) if affine else ops.integral_image(f) for f, a, b in zip(feats, scales, shifts)]
        for _ in range(2):
            got = ops.integral_images(feats, sc, sh)
            for f, g, w_, st, a, b in zip(feats, got, want, got.absmax, scales, shifts):
                assert g.shape == w_.shape and torch.equal(g.view(torch.int32), w_.view(torch.int32)), (tuple(f.shape), affine)
                x = torch.relu(f * a[:, :, None, None] + b[:, :, None, None]) if affine else f
                assert int(st.max()) == int(x.abs().max().view(torch.int32)), (tuple(f.shape), affine)


@pytest.mark.parametrize("row_slots", [0, 5])
def test_direct_items_without_a_row_slot_take_the_second_launch(row_slots):
    """The workspace may be smaller than recommended: direct items (tap window larger than LDS) that find no pooled-row slot are
    left to the second launch of the persistent kernel.  A camera inside the field (many direct items), with all / some / none of
    them in row slots: the same map within the path's tolerance, and the scene really has direct items."""
    import vfa_amd
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import look_at_camera
    from vfa_amd.utils import make_grid
    dev = _dev()
    image_size = (720, 1280)
    calibs = torch.tensor(np.stack([look_at_camera((1500.0, 1700.0, 250.0), (2600.0, 2300.0, 0.0), 700.0, (1280, 720)),
                                    look_at_camera((300.0, 300.0, 200.0), (1800.0, 1900.0, 0.0), 500.0, (1280, 720))]),
                          dtype=torch.float32).to(dev)
    grid = make_grid(world_size=(3750, 3750), cube_LW=[150.0, 150.0], dataset="MultiviewC").to(dev)
    L, W = grid.shape[:2]
    args = SimpleNamespace(data="MultiviewC", image_size=image_size)
    gen = torch.Generator().manual_seed(11)
    lats = [torch.relu(torch.randn(2, 256, h, w, generator=gen)).to(dev) for h, w in ((90, 160), (45, 80), (23, 40))]
    torch.manual_seed(5)
    mods = [vfa_amd.VFA(256, grid_height=300, cube_size=(150.0, 150.0, 300), args=args).to(dev) for _ in range(3)]
    zl, co = mods[0]._kernel_geometry(dev)
    integrals = ops.integral_images(lats)
    outs = {}
    for slots in (None, row_slots):
        ws = ops.frame_records(calibs, grid, zl, co, _lib.CONV_KIND["MultiviewC"], image_size[::-1], [tuple(l.shape[-2:]) for l in lats],
                               weights=[m.layer_major_weight() for m in mods], row_slots=slots)
        with torch.no_grad():
            outs[slots] = ops.pool_collapse(integrals, [m.collapse.bias for m in mods], ws, (L, W))
        torch.cuda.synchronize()
        lay = ops.frame_workspace_layout(2, L, W, 3)
        n_direct = int(ws[lay["counter"]:lay["counter"] + 4].cpu().numpy().view(np.uint32)[0])
        assert n_direct > 5, "the scene should have direct items"
        assert ws.numel() == (lay["total"] if slots is None else lay["rows"] + slots * 32 * 1024)
    scale = outs[None].abs().max().item()
    torch.testing.assert_close(outs[row_slots], outs[None], rtol=RTOL, atol=2 * ATOL_REL * scale)


def test_two_call_forms_of_the_entry_points_are_bitwise_the_one_call_forms():
    """`vfa_frame_boxes_f32` + `vfa_frame_cuts_f32` leave the workspace `vfa_frame_records_f32` leaves, and the fused entry point
    called as ROWS_ONLY then SKIP_ROWS gives the map of the single call -- the form `fused_frame` uses to wait for the boxes and for
    the work cuts separately.  A scene with direct items (a camera inside the field), so that the pre-pass has work."""
    import vfa_amd
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import look_at_camera
    from vfa_amd.utils import make_grid
    dev = _dev()
    image_size = (720, 1280)
    calibs = torch.tensor(np.stack([look_at_camera((1500.0, 1700.0, 250.0), (2600.0, 2300.0, 0.0), 700.0, (1280, 720)),
                                    look_at_camera((300.0, 300.0, 200.0), (1800.0, 1900.0, 0.0), 500.0, (1280, 720))]),
                          dtype=torch.float32).to(dev)
    grid = make_grid(world_size=(3750, 3750), cube_LW=[150.0, 150.0], dataset="MultiviewC").to(dev)
    L, W = grid.shape[:2]
    args = SimpleNamespace(data="MultiviewC", image_size=image_size)
    gen = torch.Generator().manual_seed(12)
    lats = [torch.relu(torch.randn(2, 256, h, w, generator=gen)).to(dev) for h, w in ((90, 160), (45, 80), (23, 40))]
    torch.manual_seed(6)
    mods = [vfa_amd.VFA(256, grid_height=300, cube_size=(150.0, 150.0, 300), args=args).to(dev) for _ in range(3)]
    zl, co = mods[0]._kernel_geometry(dev)
    sizes = [tuple(l.shape[-2:]) for l in lats]
    weights = [m.layer_major_weight() for m in mods]
    biases = [m.collapse.bias for m in mods]
    kind = _lib.CONV_KIND["MultiviewC"]
    lay = ops.frame_workspace_layout(2, L, W, 3)
    nt, K = lay["tiles_l"] * lay["tiles_w"], lay["n_chunks"]
    one = torch.zeros(lay["total"], dtype=torch.uint8, device=dev)
    two = torch.zeros(lay["total"], dtype=torch.uint8, device=dev)
    ops.frame_records(calibs, grid, zl, co, kind, image_size[::-1], sizes, weights=weights, workspace=one)
    ops.frame_records(calibs, grid, zl, co, kind, image_size[::-1], sizes, workspace=two, cuts=False)
    assert not two[lay["chunks"]:lay["chunks"] + 4 * (K + 1)].any(), "the boxes call must not write the cuts"
    ops.frame_cuts(two, 2, (L, W), 3, weights=weights)
    torch.cuda.synchronize()
    # (the order in which direct items get their row slots is arbitrary: compare what does not name a slot)
    sizes_of = {"live": nt * 4, "direct": nt * 4, "overflow": nt * 4, "recs": 2 * nt * 32 * 96, "wfrag": 8 * 16 * 2 * 64 * 16}
    for key, nbytes in sizes_of.items():
        for s in range(3):
            assert torch.equal(one[lay[key][s]:lay[key][s] + nbytes], two[lay[key][s]:lay[key][s] + nbytes]), (key, s)
    for key in ("chunks", "ranks"):
        assert torch.equal(one[lay[key]:lay[key] + 4 * (K + 1)], two[lay[key]:lay[key] + 4 * (K + 1)]), key
    assert int(one[lay["counter"]:lay["counter"] + 4].cpu().numpy().view(np.uint32)[0]) > 5, "the scene should have direct items"
    integrals = ops.integral_images(lats)
    with torch.no_grad():
        ref = ops.pool_collapse(integrals, biases, one, (L, W))
        assert ops.pool_collapse(integrals, biases, two, (L, W), stage="rows") is None
        got = ops.pool_collapse(integrals, biases, two, (L, W), stage="main")
    assert torch.equal(got, ref)
    with pytest.raises(_lib.VFAHipError):  # both stage flags at once
        _lib.call("vfa_pool_collapse_relu_sum_f32", _lib.ptr_array(list(integrals)), None, _lib.ptr_array([None] * 3), _lib.ptr(two), two.numel(),
                  _lib.ptr(got), 2, L, W, 3, _lib.int_array([v for i in integrals for v in (i.shape[1] - 2, i.shape[2] - 2)]), 0,
                  _lib.FLAG_ROWS_ONLY | _lib.FLAG_SKIP_ROWS, _lib.current_stream_handle())


@pytest.mark.parametrize("name,n_cam,crop", [("multiviewc_200x200x1", 7, None), ("multiviewc_200x200x1", 2, (21, 40)), ("multiviewc_200x200x1", 1, (3, 5)),
                                            ("multiviewc_200x200x1", 13, (60, 90))])
def test_work_cuts_cover_every_item_once(name, n_cam, crop):
    """`tile_chunks_kernel`: the n_chunks + 1 cuts (tile, rank) of the item sequence are monotonic, start at (0, 0), end at (n_tiles, 0),
    never point behind the last item of a tile, and the pieces between them carry equal estimated COST (the kernel's cost model,
    restated here: an item 704 + its window slots, a row item 522, + 294 for the first item of a (tile, scale), + 330 for the first
    of a tile, a tile without items 32) to within one item: cuts fall inside tiles."""
    import vfa_amd
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload(name, channels=256, seed=0, n_cam=n_cam)
    grid = wl["grid"] if crop is None else wl["grid"][:, 40:40 + crop[0], 30:30 + crop[1]].contiguous()
    L, W = grid.shape[1:3]
    mod = vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
    zl, co = mod._kernel_geometry(dev)
    sizes = [tuple(s) for s in wl["feat_sizes"]]
    ws = ops.frame_records(wl["calibs"].to(dev), grid.to(dev), zl, co, _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], sizes)
    host = ws.cpu().numpy()
    lay = ops.frame_workspace_layout(n_cam, L, W, 3)
    nt = lay["tiles_l"] * lay["tiles_w"]
    items = np.zeros(nt, np.int64)
    weights = [[] for _ in range(nt)]               # per tile: the cost of its items in kernel order
    for s in range(3):
        live = host[lay["live"][s]:lay["live"][s] + nt * 4].view(np.uint32)
        ovf = host[lay["overflow"][s]:lay["overflow"][s] + nt * 4].view(np.uint32)
        hdr = host[lay["hdrs"][s]:lay["hdrs"][s] + n_cam * nt * 32].view(np.uint32).reshape(n_cam, nt, 8)
        items += np.array([bin(int(a) & ~int(b)).count("1") for a, b in zip(live, ovf)])
        for t in range(nt):
            first = True
            for v in range(n_cam):
                if (int(live[t]) & ~int(ovf[t])) >> v & 1:
                    h = hdr[v, t]
                    w = 522 if h[0] & 4 else 704 + int(h[1])
                    weights[t].append(w + (294 if first else 0) + (330 if not weights[t] else 0))
                    first = False
    K = lay["n_chunks"]
    cs = host[lay["chunks"]:lay["chunks"] + (K + 1) * 4].view(np.int32).astype(np.int64)
    cr = host[lay["ranks"]:lay["ranks"] + (K + 1) * 4].view(np.int32).astype(np.int64)
    assert (cs[0], cr[0]) == (0, 0) and (cs[-1], cr[-1]) == (nt, 0)
    pos = cs * 4096 + cr
    assert (np.diff(pos) >= 0).all(), "cuts must be monotonic"
    inside = cs < nt
    assert (cr[inside] <= np.maximum(items[cs[inside]] - 1, 0)).all(), "a cut may not point behind the last item of its tile"
    before = np.concatenate([[0], np.cumsum(items)])
    upto = before[np.minimum(cs, nt)] + cr          # items in front of each cut
    assert upto[-1] == items.sum()
    cost_before = np.concatenate([[0], np.cumsum([sum(w) if w else 32 for w in weights])])
    inner = [np.concatenate([[0], np.cumsum(w)]) for w in weights]
    cost_upto = np.array([cost_before[t] + (inner[t][r] if t < nt else 0) for t, r in zip(np.minimum(cs, nt), cr)])
    biggest = max(max(w) for w in weights if w)
    for wgs in (256, 248, 240, 8):                  # pieces of a launch with that many workgroups (240: 16 CUs left to RCCL)
        idx = (np.arange(wgs + 1) * K) // wgs
        per = np.diff(upto[idx])
        assert per.sum() == items.sum()
        cost = np.diff(cost_upto[idx])
        # (a cut sits at the item boundary nearest to its ideal position: at most half an item off, a piece at most one item, two pieces two)
        assert cost.max() - cost.min() <= 2 * biggest + 0.03 * cost.mean(), (cost.min(), cost.max(), biggest)


def test_serial_kernel_default_product_has_the_width_of_the_reference_sgemm(monkeypatch):
    """`pool_collapse_kernel<2>` (the default of single-layer grids): two fp16 pieces per operand with a power-of-two scale, three
    MFMA products (vfa_split.h).  The reference's ``collapse`` is an fp32 ``nn.Linear`` (vfa_op.py:59, :123): against float64
    the fused frame must sit where an fp32 library GEMM on the same (bit-pinned) voxel features sits -- <= 1.3 x its normwise
    error or 3e-7 --, on ordinary, signed, 1e4 x and 1e-6 x feature maps, with or without the statistics of the integral-image call
    (without them the entry point derives the scale from the integral images), and an order of magnitude inside the two-piece bf16
    form (VFA_FLAG_TERMS 3)."""
    from vfa_amd import _lib, ops, vfa_op
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=3, n_cam=5)
    grid = wl["grid"][:, 70:70 + 48, 50:50 + 96].contiguous().to(dev)
    mods = _mods(wl, dev)
    lats0 = [torch.cat([wl["features"][c][s] for c in range(5)]).to(dev) for s in range(3)]
    calibs = wl["calibs"].to(dev)
    L, W = grid.shape[1:3]
    z_layers, corner_off = mods[0]._kernel_geometry(dev)
    kind, img_wh = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1]
    weights = [m.layer_major_weight() for m in mods]

    def run(lats, terms, with_stats=True):
        ws = ops.frame_records(calibs, grid, z_layers, corner_off, kind, img_wh, [tuple(l.shape[-2:]) for l in lats], weights=weights, terms=terms)
        integrals = ops.integral_images(lats)
        return ops.pool_collapse(integrals if with_stats else list(integrals), [m.collapse.bias for m in mods], ws, (L, W), terms=terms)

    for label, factor, shift in (("relu(randn)", 1.0, 0.0), ("signed", 1.0, -0.4), ("x 1e4", 1e4, 0.0), ("x 1e-6", 1e-6, 0.0)):
        lats = [l * factor + shift for l in lats0]
        saved = [m.collapse.bias.detach().clone() for m in mods]
        with torch.no_grad():
            if factor < 1e-3:  # (the bias would drown the product: leave it out of this variant)
                for m in mods:
                    m.collapse.bias.zero_()
            want = _float64_reference(mods, lats, calibs, grid, wl)
            lib32 = torch.zeros(L * W, 256, device=dev)
            grid_flat = grid.reshape(-1, 3).contiguous()
            for m, lat in zip(mods, lats):
                vox = ops.project_gather(ops.integral_image(lat), calibs.reshape(5, 12).contiguous(), grid_flat, z_layers, corner_off, kind, img_wh,
                                         kernel="direct")
                lib32 += torch.relu(torch.matmul(vox, m.collapse.weight.T) + m.collapse.bias).sum(0)
            got2, got2n, got3 = run(lats, 2), run(lats, 2, with_stats=False), run(lats, 3)
            for m, b in zip(mods, saved):
                m.collapse.bias.copy_(b)
        err = lambda t: ((t.double() - want).norm() / want.norm()).item()
        e2, e2n, e3, e32 = err(got2), err(got2n), err(got3), err(lib32)
        print(f"[width] serial kernel, {label}: fp16 x 2 {e2:.2e} (scale from the integral images: {e2n:.2e}), fp32 library GEMM {e32:.2e}, bf16 x 2 {e3:.2e}")
        assert torch.isfinite(got2).all() and torch.isfinite(got2n).all()
        assert e2 <= max(1.3 * e32, 3e-7) and e2n <= max(1.3 * e32, 3e-7), (label, e2, e2n, e32)
        assert e3 >= 3 * e2, (label, e2, e3)
        _check(f"serial kernel {label} fp16 x 2", got2, want)


def _ulp_diff(a, b):
    """Distance in fp32 units in the last place between two float tensors of the same sign pattern (0 where both are zero)."""
    ia = a.contiguous().view(torch.int32).to(torch.int64)
    ib = b.contiguous().view(torch.int32).to(torch.int64)
    ia = torch.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = torch.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return (ia - ib).abs()


@pytest.mark.parametrize("name,n_cam,crop", [("multiviewc_200x200x1", 7, None), ("wildtrack_480x1440x1", 2, (120, 1440))])
def test_shipped_pooling_code_vs_the_reference_voxel_features(name, n_cam, crop):
    """The voxel features AS THE FUSED KERNEL FORMS THEM (diagnostic VFA_FLAG_DUMP_VOX: the pooled fp32 rows in front of the operand
    split, written by the kernel's own pooling code) against the bit-pinned voxel features of ``vfa_project_gather_f32`` (= the
    reference's, tests/test_hip_parity.py).  Tap chains, box sum AND quotient are the reference's: the kernels divide with two
    Markstein corrections on RN(1 / area) (vfa_geom.h: box_quotient_scaled -- the correctly rounded quotient of vfa_op.py:118-119,
    with the power of two of the fp16 split folded in exactly), so every row is the reference's BIT FOR BIT (round 4: v * RN(1 / area),
    <= 1 ulp, 79-87 % identical).  Serial kernel (`pool_collapse_kernel`) and pipelined kernel (`pipe_kernel`, same frame as a
    one-layer grid) alike."""
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload(name, channels=256, seed=3, n_cam=n_cam)
    grid = (wl["grid"] if crop is None else wl["grid"][:, 11:11 + crop[0], 0:crop[1]]).contiguous().to(dev)
    L, W = grid.shape[1:3]
    mods = _mods(wl, dev)
    zl, co = mods[0]._kernel_geometry(dev)
    kind, img_wh = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1]
    grid_flat = grid.reshape(-1, 3).contiguous()
    worst, total, equal = 0, 0, 0
    for cam in range(n_cam):
        calib = wl["calibs"][cam:cam + 1].to(dev)
        for s in range(3):
            lat = wl["features"][cam][s].to(dev)
            with torch.no_grad():
                integrals = ops.integral_images([lat])
                ref = ops.project_gather(integrals[0], calib.reshape(1, 12).contiguous(), grid_flat, zl, co, kind, img_wh, kernel="direct")[0]
                ws = ops.frame_records(calib, grid, zl, co, kind, img_wh, [tuple(lat.shape[-2:])], weights=[mods[s].layer_major_weight()])
                got = ops.pool_collapse(integrals, [mods[s].collapse.bias], ws, (L, W), dump_vox=True)
                wsp = ops.pipe_records(calib, grid, zl, co, kind, img_wh, [tuple(lat.shape[-2:])], weights=[mods[s].collapse.weight])
                gotp = ops.pipe_collapse(integrals, [mods[s].collapse.bias], wsp, (L, W), 1, dump_vox=True)
            for label, g in (("serial", got), ("pipelined", gotp)):
                same_zero = (g == 0) & (ref == 0)  # (the sign of a masked zero is free)
                d = torch.where(same_zero, torch.zeros_like(g, dtype=torch.int64), _ulp_diff(g, ref))
                assert torch.isfinite(g).all()
                worst = max(worst, int(d.max().item()))
                total += d.numel()
                equal += int((d == 0).sum().item())
                assert d.max().item() == 0, (label, cam, s, d.max().item(), int((d != 0).sum().item()))
            assert torch.equal(got, gotp), (cam, s)  # the two kernels pool with the same operations: the same bits
    print(f"[pooled rows] {name}: {equal / total:.4%} of {total} voxel features bit-identical to the reference's, worst {worst} ulp")
    assert equal == total


@pytest.mark.gpu
def test_repeated_launches_on_one_workspace_find_clear_hand_off_tickets():
    """No memset in front of a launch any more: the geometry call zeroes the tile tickets, the workgroup that draws a tile's last
    ticket puts it back.  Launches of 8, 256 and 40 workgroups on ONE workspace (different tiles cut every time), serial and pipelined
    kernel, accumulate on and off: every launch must find its tickets clear -- a stale one would drop or double a shared tile."""
    import vfa_amd
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import make_workload
    dev = torch.device("cuda:0")
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=11, n_cam=5, device=dev)
    grid = wl["grid"][:, 20:92, 30:126].contiguous()
    L, W = grid.shape[1:3]
    torch.manual_seed(4)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    lats = [torch.cat([wl["features"][c][s] for c in range(5)]) for s in range(3)]
    zl, co = mods[0]._kernel_geometry(dev)
    kind, img_wh, sizes = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], [tuple(l.shape[-2:]) for l in lats]
    biases = [m.collapse.bias for m in mods]
    with torch.no_grad():
        integrals = ops.integral_images(lats)
        ws_s = ops.frame_records(wl["calibs"], grid, zl, co, kind, img_wh, sizes, weights=[m.layer_major_weight().contiguous() for m in mods])
        ws_p = ops.pipe_records(wl["calibs"], grid, zl, co, kind, img_wh, sizes, weights=[m.collapse.weight for m in mods])
        for run, ws in ((lambda **kw: ops.pool_collapse(integrals, biases, ws_s, (L, W), absmax=integrals.absmax, **kw), ws_s),
                        (lambda **kw: ops.pipe_collapse(integrals, biases, ws_p, (L, W), 1, absmax=integrals.absmax, **kw), ws_p)):
            first = {r: run(reserved_cus=r) for r in (248, 0, 216)}
            for r in (0, 216, 248, 248, 0):
                assert torch.equal(run(reserved_cus=r), first[r]), r
            acc = first[0].clone()
            run(reserved_cus=248, out=acc, accumulate=True)
            torch.testing.assert_close(acc, first[0] + first[248], rtol=1e-6, atol=0)
        torch.cuda.synchronize()


def test_product_form_of_workspace_and_launch_must_agree():
    """The geometry call splits ``collapse.weight`` for ONE arithmetic (fp16 pieces, or bf16 pieces) and leaves a tag beside the weight
    exponents; a collapse call that asks for the other form would read fp16 fragments as bf16 numbers.  It must fail LOUDLY: the frame
    kernels check the tag and write a map of NaNs; diagnostic flags with a non-default form are refused outright.  Serial and pipelined
    kernel."""
    import vfa_amd
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import make_workload
    dev = _dev()
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=3, n_cam=2, device=dev)
    grid = wl["grid"][:, 40:72, 40:104].contiguous()
    L, W = grid.shape[1:3]
    mods = _mods(wl, dev)
    lats = [torch.cat([wl["features"][c][s] for c in range(2)]) for s in range(3)]
    zl, co = mods[0]._kernel_geometry(dev)
    kind, img_wh, sizes = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1], [tuple(l.shape[-2:]) for l in lats]
    biases = [m.collapse.bias for m in mods]
    with torch.no_grad():
        integrals = ops.integral_images(lats)
        ws3 = ops.frame_records(wl["calibs"], grid, zl, co, kind, img_wh, sizes, weights=[m.layer_major_weight().contiguous() for m in mods], terms=3)
        ok = ops.pool_collapse(integrals, biases, ws3, (L, W), terms=3)
        bad = ops.pool_collapse(integrals, biases, ws3, (L, W), terms=2)          # fp16 kernel on bf16 fragments
        assert torch.isfinite(ok).all() and torch.isnan(bad).all()
        with pytest.raises(_lib.VFAHipError):
            ops.pool_collapse(integrals, biases, ws3, (L, W), terms=3, debug=128)  # the diagnostic build is the default form's
        wp2 = ops.pipe_records(wl["calibs"], grid, zl, co, kind, img_wh, sizes, weights=[m.collapse.weight for m in mods], terms=2)
        okp = ops.pipe_collapse(integrals, biases, wp2, (L, W), 1, terms=2)
        badp = ops.pipe_collapse(integrals, biases, wp2, (L, W), 1, terms=3)       # bf16 kernel on fp16 fragments
        assert torch.isfinite(okp).all() and torch.isnan(badp).all()
        with pytest.raises(_lib.VFAHipError):
            ops.pipe_collapse(integrals, biases, wp2, (L, W), 1, terms=6, debug=128)
    torch.cuda.synchronize()
