"""Every BASELINE.json configuration at its STATED size on one MI355X: all cameras, all three scales, every z-layer, the
full grid -- checked against the CPU oracle on spot windows of cells (not against itself).   ``-m gpu``.

Per workload the whole frame runs through ``vfa_amd.aggregate_views`` (the product path).  Then, for >= 3 windows of
consecutive cells spread over the grid:
  * the voxel features of every (camera, scale) from the pooling entry point are compared BITWISE with the oracle
    (oracle/vfa_oracle.c: box parameters + integral image + box pooling on the same cells);
  * the fused BEV map of the full-size run is compared at those cells with the float64 product of the oracle's voxel
    features, bias, ReLU, scale and view sum (post-GEMM tolerance rtol 1e-4 / atol 1e-5 max|ref|).
Reference lines covered: vfa/model/vfa_op.py:61-125, vfa/model/vfanet.py:64-82.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL_REL = 1e-4, 1e-5

#  workload                     cells per window, why it is here
FULL = [
    ("multiviewc_200x200x1", 200),      # configs[1]  7 cameras, 200 x 200 x 1 (the bench workload)
    ("multiviewc_156x156x5", 156),      # configs[0/1] shipped MultiviewC, 156 x 156 x 5
    ("wildtrack_480x1440x1", 360),      # configs[2]  7 cameras 1080p -> 480 x 1440 ground-plane grid
    ("wildtrack_120x360x8", 120),       # configs[2]  shipped Wildtrack, 8 layers
    ("multiviewx_160x250x8", 125),      # configs[3]  6 cameras, 160 x 250 x 8
    ("synthetic4k_512x512x32", 64),     # configs[4]  8 cameras x 4K -> 512 x 512 x 32 (K = 8192)
]


@pytest.mark.parametrize("name,win", FULL)
def test_full_size_frame_vs_oracle_windows(oracle, name, win):
    import vfa_amd
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import make_workload
    dev = torch.device("cuda:0")
    wl = make_workload(name, channels=256, seed=4)
    n = wl["n_cam"]
    C = 256
    torch.manual_seed(2)
    mods = [vfa_amd.VFA(C, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
            for _ in range(3)]
    nl = mods[0].num_grid_layer
    with torch.no_grad():
        for m in mods:  # larger weights than the default init: the ReLU then cuts a good share of the outputs
            m.collapse.weight.mul_(3.0)
            m.collapse.bias.uniform_(-0.3, 0.1)
    lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
    calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)
    L, W = grid.shape[1:3]
    with torch.no_grad(), ops.KernelTimer() as kt:
        ortho = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
    torch.cuda.synchronize()
    assert tuple(ortho.shape) == (1, C, L, W) and torch.isfinite(ortho).all()
    # the product path of every BASELINE config is ONE fused kernel per frame (band): the serial one on single-layer grids, the
    # pipelined one (groups of views, accumulators in registers over the layers) for K = nl * 256; voxel features never in HBM
    ran = set(kt.summary())
    assert ("vfa_pool_collapse_relu_sum_f32" if nl == 1 else "vfa_pipe_collapse_relu_sum_f32") in ran, sorted(ran)
    assert not ran & {"vfa_project_gather_f32", "vfa_collapse_gemm_f32", "vfa_pool_windows_f32"}, sorted(ran)
    got_map = ortho[0].permute(1, 2, 0).reshape(L * W, C)

    n_cells = L * W
    starts = sorted({0, (n_cells // 2 // W) * W + W // 3, n_cells - win, (n_cells // 5 // W) * W + W - win})
    grid_np = wl["grid"][0].reshape(-1, 3).numpy()
    zl_h = oracle.z_layers_of(wl["grid_height"], wl["cube_size"])
    co_h = oracle.corner_offsets(wl["cube_size"])
    assert len(zl_h) == nl
    kind = _lib.CONV_KIND[wl["args"].data]
    img_wh = wl["args"].image_size[::-1]
    want = {s0: np.zeros((win, C), np.float64) for s0 in starts}
    visible_boxes = 0
    for si, m in enumerate(mods):
        zl, co = m._kernel_geometry(dev)
        w64 = m.collapse.weight.detach().cpu().double().numpy()   # reference column order c*nl + layer
        b64 = m.collapse.bias.detach().cpu().double().numpy()
        with torch.no_grad():
            integral = ops.integral_image(lats[si])
        for cam in range(n):
            feat = wl["features"][cam][si][0].numpy()
            Hf, Wf = feat.shape[1:]
            I = oracle.integral_image(feat)
            for s0 in starts:
                cells = grid_np[s0:s0 + win]
                box, area, vis = oracle.box_params(wl["calibs"][cam].numpy(), cells, zl_h, co_h, wl["args"].data,
                                                   wl["args"].image_size, Hf, Wf)
                ref = oracle.gather(I, box, area, vis)                                   # (win, C*nl), column c*nl + layer
                visible_boxes += int(vis.sum())
                with torch.no_grad():
                    vox = ops.project_gather(integral[cam:cam + 1], calibs[cam:cam + 1].reshape(1, 12).contiguous(),
                                             grid.reshape(-1, 3).contiguous(), zl, co, kind, img_wh, cell_begin=s0,
                                             cell_count=win).cpu().numpy()[0]            # layer-major
                ref_lm = ref.reshape(win, C, nl).transpose(0, 2, 1).reshape(win, nl * C)
                same = (vox.view(np.uint32) == ref_lm.view(np.uint32)) | ((vox == 0) & (ref_lm == 0))
                assert same.all(), f"{name} scale {si} camera {cam} cells {s0}..: {np.count_nonzero(~same)} voxel " \
                                   f"features differ bitwise from the oracle"
                want[s0] += np.maximum(ref.astype(np.float64) @ w64.T + b64, 0.0)
    assert visible_boxes > 0, "the windows must contain visible boxes"
    scale = max(np.abs(v).max() for v in want.values())
    assert scale > 0
    worst = 0.0
    for s0 in starts:
        got = got_map[s0:s0 + win].cpu().double().numpy()
        tol = RTOL * np.abs(want[s0]) + ATOL_REL * scale
        worst = max(worst, float((np.abs(got - want[s0]) / tol).max()))
        np.testing.assert_allclose(got, want[s0], rtol=RTOL, atol=ATOL_REL * scale,
                                   err_msg=f"{name}: fused map at cells {s0}..{s0 + win}")
    print(f"[margin] {name}: {len(starts)} windows x {win} cells, {n} cameras x 3 scales x {nl} layers, "
          f"worst |err| / tolerance = {worst:.3f}")
