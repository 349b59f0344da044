"""Gradients of the HIP path (autograd through the C ABI's backward entry points) against a float64 PyTorch
restatement differentiated by autograd.   -m gpu."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import torch_reference as ref

pytestmark = pytest.mark.gpu


def _case(data, seed, C=8):
    from vfa_amd.synthetic import ring_cameras
    from vfa_amd.utils import make_grid
    gen = torch.Generator().manual_seed(seed)
    Hf, Wf = 14, 22
    image_size = (14 * 8, 22 * 8)
    if data == "MultiviewC":
        grid = make_grid((700, 900), cube_LW=(50, 50), dataset=data)
        cube, gh = (50, 50, 40), 120
        calib = ring_cameras(3, (450., 350., 0.), 900., 350., 150., (image_size[1], image_size[0]), phase=0.4)[seed % 3]
    elif data == "Wildtrack":
        grid = make_grid((60, 80), cube_LW=(5, 5), dataset=data)
        cube, gh = (5, 5, 6), 18
        calib = ring_cameras(3, (-200., -800., 0.), 420., 160., 140., (image_size[1], image_size[0]), phase=0.2)[seed % 3]
    else:
        grid = make_grid((200, 240), cube_LW=(20, 20), dataset=data)
        cube, gh = (20, 20, 24), 72
        calib = ring_cameras(3, (3., 2.5, 0.), 9., 3., 150., (image_size[1], image_size[0]), phase=0.1)[seed % 3]
    feature = torch.randn(1, C, Hf, Wf, generator=gen)
    return dict(data=data, feature=feature, calib=calib, grid=grid, cube=cube, gh=gh, image_size=image_size, gen=gen)


@pytest.mark.parametrize("data,seed,C", [("MultiviewC", 0, 8), ("MultiviewC", 1, 8), ("Wildtrack", 2, 8), ("MultiviewX", 3, 8),
                                         ("MultiviewC", 4, 256), ("Wildtrack", 5, 256)])
def test_module_gradients_vs_float64_autograd(data, seed, C):
    """C = 8: library GEMM + run-combined atomic scatter; C = 256: the MFMA tile GEMM in the forward (`_CollapseGemm`,
    library products in its backward) and the LDS-privatised scatter."""
    import vfa_amd
    dev = torch.device("cuda:0")
    c = _case(data, seed, C)
    args = SimpleNamespace(data=data, image_size=c["image_size"])
    torch.manual_seed(seed)
    mod = vfa_amd.VFA(C, grid_height=c["gh"], cube_size=c["cube"], args=args).to(dev)
    feat = c["feature"].to(dev).requires_grad_(True)
    out = mod(feat, c["calib"].to(dev), c["grid"].to(dev)[None])
    probe = torch.randn(out.shape, generator=c["gen"])
    (out * probe.to(dev)).sum().backward()

    f64 = c["feature"].double().requires_grad_(True)
    w64 = mod.collapse.weight.detach().cpu().double().requires_grad_(True)
    b64 = mod.collapse.bias.detach().cpu().double().requires_grad_(True)
    zl = mod.z_corners[:, 0, 0, 2].cpu().double()
    co = mod.corners_offset.cpu().double().reshape(8, 3)
    want = ref.vfa_forward(f64, c["calib"].double(), c["grid"].double(), w64, b64, zl, co, data, c["image_size"])
    (want * probe.double()).sum().backward()

    def close(name, got, exp, tol):
        got, exp = got.detach().cpu().double(), exp.detach()
        scale = exp.abs().max().item()
        assert scale > 0, name
        err = (got - exp).abs().max().item()
        assert err <= tol * scale, f"{name}: max err {err:.3e} vs scale {scale:.3e}"

    close("forward", out, want, 2e-4)
    assert (want > 0).float().mean() > 0.05, "the case should have active outputs"
    close("d feature", feat.grad, f64.grad, 2e-3)
    close("d weight", mod.collapse.weight.grad, w64.grad, 2e-3)
    close("d bias", mod.collapse.bias.grad, b64.grad, 2e-3)


def test_aggregate_gradients_match_per_camera_loop():
    """Batched aggregate (3 scales x n cameras, fused epilogue) and the reference-style loop give the same grads."""
    import vfa_amd
    from vfa_amd.synthetic import make_workload
    dev = torch.device("cuda:0")
    wl = make_workload("multiviewc_156x156x5", channels=16, seed=3, n_cam=3)
    grid = wl["grid"][:, :30, :34].contiguous().to(dev)
    torch.manual_seed(1)
    mods = [vfa_amd.VFA(16, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
            for _ in range(3)]
    lats = [torch.cat([wl["features"][c][s] for c in range(3)]).to(dev).requires_grad_(True) for s in range(3)]
    calibs = wl["calibs"].to(dev)
    out = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
    probe = torch.randn_like(out)
    (out * probe).sum().backward()
    g_batched = [l.grad.clone() for l in lats] + [m.collapse.weight.grad.clone() for m in mods]
    for l in lats:
        l.grad = None
    for m in mods:
        m.zero_grad()
    acc = 0
    for cam in range(3):
        f = [mods[s](lats[s][[cam]], calibs[cam], grid) for s in range(3)]
        acc = acc + (f[0] + f[1] + f[2])
    (acc * probe).sum().backward()
    g_loop = [l.grad for l in lats] + [m.collapse.weight.grad for m in mods]
    for a, b in zip(g_batched, g_loop):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-4 * b.abs().max().item())


@pytest.mark.parametrize("workload,n_cam,cells,crop", [("multiviewc_156x156x5", 2, (0, None), None),
                                                        ("multiviewc_200x200x1", 3, (0, None), None),
                                                        ("wildtrack_120x360x8", 2, (1003, 20011), None),
                                                        ("multiviewx_160x250x8", 2, (77, 13), None),
                                                        # a grid of 7 x 13 cells out of the middle of the field: patches cut by the grid's
                                                        # right edge and its last rows, and a range that starts inside a patch
                                                        ("multiviewc_200x200x1", 2, (5, 80), (7, 13)),
                                                        ("multiviewc_156x156x5", 3, (0, None), (9, 5))])
def test_lds_scatter_matches_atomic_scatter_and_is_the_adjoint(workload, n_cam, cells, crop):
    """C = 256 takes the LDS-privatised backward (`gather_backward_cached_kernel`).  Box pooling is linear in the
    integral image, so <pool(I), G> == <I, pool^T(G)> (a size-independent property), and the kernel agrees with the
    run-combined atomic scatter."""
    from vfa_amd import _lib, ops
    from vfa_amd.synthetic import WORKLOADS, make_workload
    import vfa_amd
    if workload not in WORKLOADS:
        pytest.skip(f"{workload} not defined")
    dev = torch.device("cuda:0")
    wl = make_workload(workload, channels=256, seed=5, n_cam=n_cam)
    mod = vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
    zl, co = mod._kernel_geometry(dev)
    calibs = wl["calibs"].reshape(n_cam, 12).to(dev)
    grid4 = wl["grid"] if crop is None else wl["grid"][:, 60:60 + crop[0], 70:70 + crop[1]].contiguous()
    grid = grid4.reshape(-1, 3).to(dev)
    kind, size = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1]
    begin, count = cells
    count = grid.shape[0] - begin if count is None else count
    feat = torch.cat([wl["features"][c][1] for c in range(n_cam)]).to(dev)
    integral = ops.integral_image(feat)
    vox = ops.project_gather(integral, calibs, grid, zl, co, kind, size, cell_begin=begin, cell_count=count)
    gen = torch.Generator(device="cpu").manual_seed(11)
    gvox = torch.randn(vox.shape, generator=gen).to(dev)
    _, _, visible = ops.box_params(calibs, grid, zl, co, kind, size, feat.shape[-2:])
    live = visible[:, :, begin:begin + count].permute(0, 2, 1).bool()  # (n, cells, nl)
    live = live[..., None].expand(-1, -1, -1, 256).reshape(vox.shape)
    assert live.any()
    gvox = torch.where(live, gvox, torch.full_like(gvox, 1e30))  # masked voxels pass no gradient, whatever arrives
    grads = {}
    gw = grid4.shape[-2]  # cells per row of the ground grid: the scatter on 4 x 8 patches ("patches") instead of cells in a line
    for name, kern, w_ in (("lds", None, 0), ("patches", None, gw), ("atomics", "direct", 0)):  # per-call arguments, no library state
        grads[name] = ops.project_gather_backward(gvox, tuple(integral.shape), calibs, grid, zl, co, kind, size,
                                                  cell_begin=begin, cell_count=count, kernel=kern, grid_w=w_)
        twice = ops.project_gather_backward(gvox, tuple(integral.shape), calibs, grid, zl, co, kind, size,
                                            cell_begin=begin, cell_count=count, out=grads[name].clone(),
                                            accumulate=True, kernel=kern, grid_w=w_)
        torch.testing.assert_close(twice, 2 * grads[name], rtol=1e-4, atol=1e-5 * grads[name].abs().max().item())
    scale = grads["atomics"].abs().max().item()
    assert scale > 0 and torch.isfinite(grads["lds"]).all() and torch.isfinite(grads["patches"]).all()
    torch.testing.assert_close(grads["lds"], grads["atomics"], rtol=1e-4, atol=2e-5 * scale)
    torch.testing.assert_close(grads["patches"], grads["atomics"], rtol=1e-4, atol=2e-5 * scale)
    lhs = (vox.double() * torch.where(live, gvox, torch.zeros_like(gvox)).double()).sum().item()
    rhs = (integral.double() * grads["lds"].double()).sum().item()
    norm = (vox.double().abs() * torch.where(live, gvox, torch.zeros_like(gvox)).double().abs()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * norm, (lhs, rhs, norm)


def test_integral_backward_is_reverse_double_cumsum():
    from vfa_amd import ops
    dev = torch.device("cuda:0")
    g = torch.randn(2, 9 + 2, 37 + 2, 70, device=dev)
    want = g[:, 1:-1, 1:-1].double().flip(1).cumsum(1).flip(1).flip(2).cumsum(2).flip(2).permute(0, 3, 1, 2)
    got = ops.integral_image_backward(g.clone())
    torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-4)


def test_vfanet_trains_end_to_end():
    """A few SGD steps through backbone -> laterals -> HIP projector (forward + backward kernels) -> heads reduce a
    fixed regression loss: the autograd wiring of the whole caller works, gradients reach every parameter group."""
    import vfa_amd
    from vfa_amd.vfanet import VFANet
    from vfa_amd.synthetic import ring_cameras
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    args = SimpleNamespace(data="MultiviewC", image_size=(128, 192))
    net = VFANet(args, grid_height=64, cube_size=(50, 50, 32), angle_range=12).to(dev).train()
    images = torch.rand(2, 3, 128, 192, device=dev)
    calibs = ring_cameras(2, (400., 300., 0.), 1100., 400., 160., (192, 128)).to(dev)
    grid = vfa_amd.make_grid((600, 800), cube_LW=(50, 50), dataset="MultiviewC").to(dev)[None]
    target = torch.rand(1, 1, grid.shape[1], grid.shape[2], device=dev)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    losses = []
    for it in range(6):
        opt.zero_grad()
        with vfa_amd.ops.KernelTimer() as kt:
            out = net(images, calibs, grid)
        if it == 0:
            torch.cuda.synchronize()  # the training forward runs the fused per-frame kernel (2 layers here: the pipelined one), not the vox-through-HBM path
            ran = set(kt.summary())
            assert "vfa_pipe_collapse_relu_sum_f32" in ran and "vfa_project_gather_f32" not in ran, sorted(ran)
        loss = ((out["heatmap"] - target) ** 2).mean() + 1e-3 * out["loc_offset"].pow(2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    for name in ("base.conv1.weight", "lat8.weight", "vfa8.collapse.weight", "vfa32.collapse.bias", "map_classifier.0.weight"):
        g = dict(net.named_parameters())[name].grad
        assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0, name


@pytest.mark.parametrize("name,n_cam,crop", [("multiviewc_200x200x1", 3, (24, 40)), ("multiviewc_156x156x5", 2, (16, 24))])
def test_fused_training_forward_gives_the_gradients_of_the_unfused_path(name, n_cam, crop, monkeypatch):
    """`_FusedFrameTrain`: fused kernel in the forward, voxel features and pre-activations recomputed in the backward -- against
    the unfused autograd path (vox and lin saved by autograd) on the same inputs: same map, same gradients of the lateral maps,
    `collapse.weight` and `collapse.bias` (reference trainer.py:26, 41 trains through exactly these)."""
    import vfa_amd
    from vfa_amd import ops, vfa_op
    from vfa_amd.synthetic import make_workload
    dev = torch.device("cuda:0")
    wl = make_workload(name, channels=256, seed=2, n_cam=n_cam)
    grid = wl["grid"][:, 40:40 + crop[0], 50:50 + crop[1]].contiguous().to(dev)
    torch.manual_seed(4)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    with torch.no_grad():
        for m in mods:
            m.collapse.weight.mul_(3.0)
            m.collapse.bias.uniform_(-0.3, 0.1)
    calibs = wl["calibs"].to(dev)
    probe = torch.randn(1, 256, crop[0], crop[1], device=dev)
    # Integer-valued features: their integral images and box sums are exact in fp32, so the two fp32 paths see the same voxel
    # features bit for bit whatever the order of their sums.
    feats = [torch.round(2 * torch.cat([wl["features"][c][s] for c in range(n_cam)])) for s in range(3)]
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(vfa_op, "FUSED_TRAIN", fused)
        lats = [feats[s].to(dev).requires_grad_(True) for s in range(3)]
        for m in mods:
            m.zero_grad()
        with ops.KernelTimer() as kt:
            out = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        torch.cuda.synchronize()
        ran = set(kt.summary())
        assert (("vfa_pipe_collapse_relu_sum_f32" in ran or "vfa_pool_collapse_relu_sum_f32" in ran) and "vfa_project_gather_f32" not in ran) == fused, sorted(ran)
        (out * probe).sum().backward()
        res[fused] = (out.detach().clone(), [l.grad.clone() for l in lats], [m.collapse.weight.grad.clone() for m in mods],
                      [m.collapse.bias.grad.clone() for m in mods])
    (o1, gl1, gw1, gb1), (o0, gl0, gw0, gb0) = res[True], res[False]

    # float64 autograd of the same aggregate
    l64 = [feats[s].double().requires_grad_(True) for s in range(3)]
    w64 = [m.collapse.weight.detach().cpu().double().requires_grad_(True) for m in mods]
    b64 = [m.collapse.bias.detach().cpu().double().requires_grad_(True) for m in mods]
    zl = mods[0].z_corners[:, 0, 0, 2].cpu().double()
    co = mods[0].corners_offset.cpu().double().reshape(8, 3)
    want = 0
    for c in range(n_cam):
        for k in range(3):
            want = want + ref.vfa_forward(l64[k][[c]], wl["calibs"][c].double(), grid.cpu().double()[0], w64[k], b64[k], zl, co,
                                          wl["args"].data, wl["args"].image_size)
    (want * probe.cpu().double()).sum().backward()

    errs = []

    def close(a, b, what, tol):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        scale = b.abs().max().item()
        assert scale > 0, what
        err = (a - b).abs().max().item() / scale
        errs.append((what, err, tol))
    close(o1, o0, "map", 2e-4)
    close(o1, want, "map vs float64", 2e-4)
    # The fused path's mask is its forward's (fp16 x 2 product, recomputed bit for bit: test_recomputed_relu_mask_...), the unfused
    # path's comes from the bf16 x 2 tile GEMM: a pre-activation below either product's error (this 5-layer case holds one of 3e-8
    # among its 196 608) may fall on different sides.  On these tiny grids ONE flipped element moves a column of d weight by
    # |probe| |vox| against a maximum that is a random sum over only n_cam x cells rows: allow eight times 1 / rows.
    flip = 8.0 / (n_cam * crop[0] * crop[1])
    for k in range(3):
        close(gl1[k], gl0[k], f"d lat{k}", max(2e-3, flip / 2))     # (fp32 atomics in a different order under the two suffix sums)
        close(gw1[k], gw0[k], f"d weight{k}", max(2e-4, flip))
        close(gb1[k], gb0[k], f"d bias{k}", max(2e-4, flip))
        # float64 autograd computes the box corners in float64: a few corners land on the other side of a pixel boundary
        # (map: 1e-4 of the maximum), a few pre-activations on the other side of zero, and sums of random-sign terms move by
        # 1-5 % of their maximum -- for BOTH fp32 paths alike (measured).  A guard against gross errors (layouts, scales) only.
        close(gl1[k], l64[k].grad, f"d lat{k} vs float64", 0.1)
        close(gw1[k], w64[k].grad, f"d weight{k} vs float64", 0.1)
        close(gb1[k], b64[k].grad, f"d bias{k} vs float64", 0.1)
    assert all(e <= t for _, e, t in errs), [x for x in errs if x[1] > x[2]]


@pytest.mark.parametrize("n,cells,K", [(3, 1000, 256), (2, 333, 1280), (1, 128, 256)])
def test_relu_mask_epilogue_equals_product_then_mask(n, cells, K):
    """`vfa_collapse_gemm_relu_backward_f32` (the ReLU mask and d b as the epilogue of the recomputed product: the pre-activations
    are never written) against `vfa_collapse_gemm_f32` + `vfa_relu_mask_backward_f32`: d lin BITWISE (same product, same
    comparison), d b up to the order of its atomics."""
    from vfa_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n * 7 + K)
    vox = torch.relu(torch.randn(n, cells, K, generator=g)).to(dev)
    w = (torch.randn(256, K, generator=g) / K ** 0.5).to(dev)
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    gout = torch.randn(cells, 256, generator=g).to(dev)
    glin, gb = ops.collapse_gemm_relu_backward(vox, w, b, gout, terms=3)
    lin = ops.collapse_gemm(vox.view(n * cells, K), w, terms=3).view(n, cells, 256)
    want_glin, want_gb = ops.relu_mask_backward(gout, lin, b)
    assert torch.equal(glin, want_glin)
    assert 0.2 < (glin != 0).float().mean().item() < 0.8
    torch.testing.assert_close(gb, want_gb, rtol=1e-4, atol=1e-4 * want_gb.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K", [(20000, 256), (3331, 1280), (16, 256), (7, 512), (100003, 256)])
def test_weight_gradient_product_has_the_width_of_an_sgemm(rows, K):
    """``vfa_grad_weight_f32``: g_w = g_lin^T . vox as six bf16 MFMA products of a three-piece split (reference: the autograd of
    nn.Linear's weight, vfa_op.py:59, :123 under trainer.py:41).  Against float64: no worse than the library's fp32 product; ragged row
    counts (a last step with fewer than 16 rows), accumulation, heavy-tailed operands, and the same bits on every run."""
    from vfa_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + K)
    g_lin = torch.randn(rows, 256, generator=g)
    g_lin[g_lin.abs() < 0.6] = 0.0                     # (a ReLU mask leaves many exact zeros)
    vox = torch.randn(rows, K, generator=g) * torch.exp(2.0 * torch.randn(rows, 1, generator=g))  # (rows of very different size)
    g_lin, vox = g_lin.to(dev), vox.to(dev)
    want = g_lin.double().t() @ vox.double()
    got = ops.grad_weight(g_lin, vox)
    lib = g_lin.t() @ vox
    err = ((got.double() - want).norm() / want.norm()).item()
    err_lib = ((lib.double() - want).norm() / want.norm()).item()
    print(f"[margin] rows {rows}, K {K}: normwise error {err:.2e} (library fp32 product {err_lib:.2e})")
    assert err <= 6e-7 and err <= err_lib + 5e-8, (err, err_lib)  # (fp32 accumulation over up to 1e5 rows: 3e-7 at 2e4)
    assert torch.equal(ops.grad_weight(g_lin, vox), got)
    acc = torch.full((256, K), 0.5, device=dev)
    ops.grad_weight(g_lin, vox, out=acc, accumulate=True)
    torch.testing.assert_close(acc, got + 0.5, rtol=1e-6, atol=1e-6)
    assert torch.equal(ops.grad_weight(g_lin[:0], vox[:0]), torch.zeros(256, K, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K", [(20000, 256), (3331, 1280), (256, 256), (7, 512), (100003, 256)])
def test_input_gradient_product_has_the_width_of_an_sgemm(rows, K):
    """``vfa_grad_input_f32``: g_vox = g_lin . w as six bf16 MFMA products of a three-piece split (reference: the autograd of nn.Linear's
    input, vfa_op.py:123 under trainer.py:41).  Against float64: the class of the library's fp32 product; ragged row counts (a last
    block with fewer than 256 rows), several K tiles, and the same bits on every run."""
    from vfa_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + K + 1)
    g_lin = torch.randn(rows, 256, generator=g) * torch.exp(2.0 * torch.randn(rows, 1, generator=g))
    g_lin[g_lin.abs() < 0.6] = 0.0
    w = torch.randn(256, K, generator=g) / 16.0
    w[::9] *= 50.0                                     # (a heavy tail across the reduction index)
    g_lin, w = g_lin.to(dev), w.to(dev)
    want = g_lin.double() @ w.double()
    got = ops.grad_input(g_lin, w)
    lib = g_lin @ w
    err = ((got.double() - want).norm() / want.norm()).item()
    err_lib = ((lib.double() - want).norm() / want.norm()).item()
    print(f"[margin] rows {rows}, K {K}: normwise error {err:.2e} (library fp32 product {err_lib:.2e})")
    assert err <= 3e-7 and err <= err_lib + 1e-7, (err, err_lib)
    assert torch.equal(ops.grad_input(g_lin, w), got)
    # row by row: no row of the ragged last block is lost or written twice
    worst = ((got.double() - want).abs().amax(1) / want.abs().amax(1).clamp_min(1e-30)).max().item()
    assert worst <= 2e-6, worst


@pytest.mark.parametrize("name,n_cam,crop", [("multiviewc_200x200x1", 7, None), ("multiviewc_156x156x5", 2, (96, 156)), ("wildtrack_120x360x8", 2, (64, 360))])
def test_recomputed_relu_mask_is_the_forwards_bit_for_bit(name, n_cam, crop):
    """The training backward recomputes ``lin = vox . W^T + b`` to get the ReLU mask (the fused forward keeps no pre-activations).
    With the default arithmetic it repeats the FORWARD's product -- fp16 pieces under the frame's scales, the per-item sliver shifts,
    the accumulator started at bias 2^(ea+ew-shift), the three MFMA products in the frame kernels' order
    (``vfa_collapse_gemm_relu_backward_f16_f32`` + ``vfa_sliver_shifts_u8``) -- so the mask must be the forward's on EVERY element.  The
    forward's mask is observable on a frame of one view and one scale (out = relu(lin) 2^-k: positive exactly where lin is): every
    (camera, scale) of the frame, serial kernel (single-layer grid) and pipelined kernel (5 and 8 layers; Wildtrack has shifted items).
    The bf16 recompute of round 4 is counted beside it (reference: vfa_op.py:123-124 under trainer.py:41)."""
    import vfa_amd
    from vfa_amd import _lib, ops, vfa_op
    from vfa_amd.synthetic import make_workload
    dev = torch.device("cuda:0")
    wl = make_workload(name, channels=256, seed=4, n_cam=n_cam if name != "wildtrack_120x360x8" else 7, device=dev)
    cams = range(n_cam) if name != "wildtrack_120x360x8" else (4, 5)  # (cameras 4-5 of the Wildtrack rig see the sliver boxes)
    grid = wl["grid"] if crop is None else wl["grid"][:, :crop[0], :crop[1]].contiguous()
    L, W = grid.shape[1:3]
    torch.manual_seed(3)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    with torch.no_grad():
        for m in mods:
            m.collapse.weight.mul_(3.0)
            m.collapse.bias.uniform_(-0.3, 0.1)
    nl = mods[0].num_grid_layer
    zl, co = mods[0]._kernel_geometry(dev)
    kind, img_wh = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1]
    grid_flat = grid.reshape(-1, 3).contiguous()
    ones = torch.ones(L * W, 256, device=dev)
    total = flipped_bf16 = shifted_rows = 0
    with torch.no_grad():
        for s, m in enumerate(mods):
            w_lm = m.collapse.weight.view(256, 256, nl).permute(0, 2, 1).reshape(256, nl * 256).contiguous()
            for cam in cams:
                lat = wl["features"][cam][s]
                cal = wl["calibs"][cam:cam + 1]
                integrals = ops.integral_images([lat])
                piped = vfa_op.pipe_frame_ok([m], 1)
                frame = vfa_op.pipe_frame if piped else vfa_op.fused_frame
                out = frame([m], None, cal, grid, integrals=integrals)            # (L*W, 256) = relu(lin) of this view and scale
                mask_fwd = out > 0
                vox = ops.project_gather(integrals[0], cal.reshape(1, 12).contiguous(), grid_flat, zl, co, kind, img_wh)
                shifts = ops.sliver_shifts(cal, grid, zl, co, kind, img_wh, lat.shape[-2:], not piped)
                g, _ = ops.collapse_gemm_relu_backward(vox, w_lm, m.collapse.bias, ones, absmax=integrals.absmax[0], shift=shifts)
                g3, _ = ops.collapse_gemm_relu_backward(vox, w_lm, m.collapse.bias, ones, terms=3)
                assert torch.equal(g[0] > 0, mask_fwd), (name, s, cam, int(((g[0] > 0) != mask_fwd).sum()))
                total += mask_fwd.numel()
                flipped_bf16 += int(((g3[0] > 0) != mask_fwd).sum())
                shifted_rows += int((shifts > 0).sum())
                assert 0.02 < mask_fwd.float().mean().item() < 0.98
    print(f"[mask] {name}: {total} pre-activations, recomputed fp16 mask identical to the forward's; the bf16 recompute of round 4 "
          f"differs on {flipped_bf16}; rows with a sliver shift: {shifted_rows}")


def test_training_step_through_the_fused_node_uses_the_fp16_recompute():
    """``aggregate_views`` with gradients wanted on a single-layer frame: the backward calls the fp16 form of the tile GEMM (the
    forward's product) and the shift kernel; gradients still those of tests/test_reference_gradients.py (checked there)."""
    import vfa_amd
    from vfa_amd import ops
    from vfa_amd.synthetic import make_workload
    dev = torch.device("cuda:0")
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=2, n_cam=2, device=dev)
    grid = wl["grid"][:, 60:124, 40:136].contiguous()
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
    lats = [torch.cat([wl["features"][c][s] for c in range(2)]).requires_grad_(True) for s in range(3)]
    with ops.KernelTimer() as kt:
        out = vfa_amd.aggregate_views(*mods, *lats, wl["calibs"], grid)
        out.sum().backward()
    torch.cuda.synchronize()
    ran = set(kt.summary())
    assert {"vfa_collapse_gemm_relu_backward_f16_f32", "vfa_sliver_shifts_u8", "vfa_grad_weight_f32"} <= ran, sorted(ran)
    assert "vfa_collapse_gemm_relu_backward_f32" not in ran
    assert all(torch.isfinite(l.grad).all() for l in lats)
