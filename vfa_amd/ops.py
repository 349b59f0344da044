"""Functional wrappers, one per entry point of ``include/vfa_hip.h`` (no autograd, no fallback).

Tensors are device tensors; every call is asynchronous on the current torch HIP stream.
"""
import os
import ctypes

import torch

from . import _lib


def _f32c(t):
    return t.to(dtype=torch.float32).contiguous()


class KernelTimer:
    """Optional per-launch HIP-event timing of the entry points (bench.py's roofline leg).

    Events are recorded on the current torch stream, which is the stream every kernel is launched on.
    ``with KernelTimer() as kt: ...``; after a device synchronise ``kt.summary()`` gives, per entry point,
    the number of launches and their total milliseconds.
    """
    active = None

    def __init__(self, only=None, every=1):
        """``only``: an iterable of entry-point names; other launches are not timed (each timed launch costs two event
        records in the queue, ~3 us apiece on MI355X: 18 per frame slow a 0.9 ms frame by 6 %).  ``every``: time only every
        n-th eligible launch of an entry point (a sample spread over the whole region instead of 1 % of every step)."""
        self.records = []
        self.only = None if only is None else frozenset(only)
        self.every = max(int(every), 1)
        self.seen = {}

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        out = {}
        for name, e0, e1, tag in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, by_tag={}))
            ms = e0.elapsed_time(e1)
            d["launches"] += 1
            d["ms"] += ms
            t = d["by_tag"].setdefault(tag, dict(launches=0, ms=0.0))
            t["launches"] += 1
            t["ms"] += ms
        return out


def _launch(name, *args, tag=None):
    kt = KernelTimer.active
    # (the "rows" pre-pass call of the two-stage fused entry point is left untimed: the timed call is the persistent kernel)
    if kt is None or (kt.only is not None and name not in kt.only) or (isinstance(tag, tuple) and tag and tag[-1] == "rows"):
        _lib.call(name, *args)
        return
    if kt.every > 1:
        k = kt.seen.get(name, 0)
        kt.seen[name] = k + 1
        if k % kt.every:
            _lib.call(name, *args)
            return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(name, *args)
    e1.record()
    kt.records.append((name, e0, e1, tag))


def integral_image(features):
    """(n,C,Hf,Wf) -> (n,Hf+2,Wf+2,C) zero-bordered channels-last integral images (reference vfa_op.py:172-173)."""
    _lib.require_device(features)
    features = _f32c(features)
    n, C, Hf, Wf = features.shape
    integral = torch.empty((n, Hf + 2, Wf + 2, C), dtype=torch.float32, device=features.device)
    _launch("vfa_integral_image_f32", _lib.ptr(features), _lib.ptr(integral), n, C, Hf, Wf,
            _lib.current_stream_handle(), tag=(n, C, Hf, Wf))
    return integral


def affine_relu_integral_image(x, scale, shift):
    """Integral images of relu(x * scale[:, :, None, None] + shift[:, :, None, None]) without materialising that map: the
    GroupNorm affine + ReLU of the lateral branch fused into the row scan (reference vfanet.py:72-74 + vfa_op.py:110).
    x (n,C,Hf,Wf), scale / shift (n,C) -> (n,Hf+2,Wf+2,C)."""
    _lib.require_device(x, scale, shift)
    x, scale, shift = _f32c(x), _f32c(scale), _f32c(shift)
    n, C, Hf, Wf = x.shape
    assert tuple(scale.shape) == (n, C) and tuple(shift.shape) == (n, C)
    integral = torch.empty((n, Hf + 2, Wf + 2, C), dtype=torch.float32, device=x.device)
    _launch("vfa_affine_relu_integral_image_f32", _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(integral), n, C, Hf, Wf,
            _lib.current_stream_handle(), tag=(n, C, Hf, Wf))
    return integral


class IntegralImages(list):
    """The integral images of a frame (a plain list of tensors, one per stride) + ``absmax``: per stride the partial maxima of
    |feature| the row scan saw (uint32 bit patterns) -- what the fused frame kernels scale their fp16 operand split by
    (``pool_collapse`` / ``pipe_collapse`` pick it up from here; a plain list works too and costs one extra pass per stride)."""
    absmax = None


def feature_stats_count(n_views, C, Hf):
    return int(_lib.lib().vfa_feature_stats_count(int(n_views), int(C), int(Hf)))


def integral_absmax(integral):
    """Partial maxima of |feature| from a finished integral image (``vfa_integral_absmax_f32``): (n,Hf+2,Wf+2,C) -> int32 tensor."""
    _lib.require_device(integral)
    n, Hp, Wp, C = integral.shape
    out = torch.empty(max(feature_stats_count(n, C, Hp - 2), 1), dtype=torch.int32, device=integral.device)
    _launch("vfa_integral_absmax_f32", _lib.ptr(integral), _lib.ptr(out), n, C, Hp - 2, Wp - 2, _lib.current_stream_handle())
    return out


def _absmax_of(integrals, absmax):
    """The statistics that belong to these integral images: given explicitly, or carried by an ``IntegralImages`` list."""
    if absmax is None:
        absmax = getattr(integrals, "absmax", None)
    if absmax is None:
        return None, None
    absmax = list(absmax)
    assert len(absmax) == len(integrals)
    for i, t in zip(integrals, absmax):
        assert t is None or (t.dtype == torch.int32 and t.is_contiguous()
                             and t.numel() >= feature_stats_count(i.shape[0], i.shape[3], i.shape[1] - 2))
    return absmax, _lib.ptr_array(absmax)


def integral_images(features, scales=None, shifts=None, channels_last=False):
    """The integral images of every feature map of a frame in one launch pair (``vfa_integral_images_f32``): features = one
    (n,C,H_s,W_s) batch per stride -> one (n,H_s+2,W_s+2,C) image per stride, bit-identical to ``integral_image`` of each.
    ``scales`` / ``shifts`` (one (n,C) tensor per map): the fused GroupNorm affine + ReLU of ``affine_relu_integral_image``.
    ``channels_last``: the inputs are (n,H_s,W_s,C) (``lateral_conv``'s output; ``vfa_integral_images_hwc_f32``).
    Returns an ``IntegralImages`` list (the images + the feature statistics of the fp16 operand split)."""
    features = [_f32c(f) for f in features]
    _lib.require_device(*features)
    n = features[0].shape[0]
    C = features[0].shape[3 if channels_last else 1]
    sizes = [tuple(f.shape[1:3]) if channels_last else tuple(f.shape[2:]) for f in features]
    assert all(f.shape[0] == n and f.shape[3 if channels_last else 1] == C for f in features)
    outs = IntegralImages(torch.empty((n, h + 2, w + 2, C), dtype=torch.float32, device=f.device) for f, (h, w) in zip(features, sizes))
    counts = [max(feature_stats_count(n, C, h), 1) for h, _ in sizes]
    stats = torch.empty(sum(counts), dtype=torch.int32, device=features[0].device)  # (every entry is written: no initialisation)
    outs.absmax = list(torch.split(stats, counts))
    affine = scales is not None
    if affine:
        scales, shifts = [_f32c(t) for t in scales], [_f32c(t) for t in shifts]
        assert all(tuple(t.shape) == (n, C) for t in scales + shifts)
    hw = _lib.int_array([v for hw_ in sizes for v in hw_])
    _launch("vfa_integral_images_hwc_f32" if channels_last else "vfa_integral_images_f32", _lib.ptr_array(features),
            _lib.ptr_array(scales) if affine else None, _lib.ptr_array(shifts) if affine else None, _lib.ptr_array(outs),
            _lib.ptr_array(outs.absmax), n, C, len(features), hw, _lib.current_stream_handle(), tag=(n, C, tuple(sizes), affine))
    return outs


_lateral_ws = {}


def lateral_conv(feat, weight, bias, gamma, beta, eps=1e-5):
    """The lateral branch of one scale as far as the integral image needs it (``vfa_lateral_conv_f32``): feat (n,K,h,w) NCHW,
    weight (256,K) or (256,K,1,1), bias / gamma / beta (256) -> y (n,h,w,256) = conv1x1(feat) + bias CHANNELS-LAST, and the
    nn.GroupNorm(16, 256) affine of y as scale, shift (n,256): relu(y * scale + shift) = relu(bn(lat(feat))) (reference
    vfanet.py:72-74).  fp32 FMA chain on the matrix pipe; statistics gathered in the epilogue."""
    _lib.require_device(feat, weight, bias, gamma, beta)
    feat, weight = _f32c(feat), _f32c(weight.reshape(weight.shape[0], -1))
    bias, gamma, beta = _f32c(bias), _f32c(gamma), _f32c(beta)
    n, K, h, w = feat.shape
    assert tuple(weight.shape) == (256, K) and bias.numel() == 256 and gamma.numel() == 256 and beta.numel() == 256
    dev = feat.device
    out = torch.empty((n, h, w, 256), dtype=torch.float32, device=dev)
    scale = torch.empty((n, 256), dtype=torch.float32, device=dev)
    shift = torch.empty((n, 256), dtype=torch.float32, device=dev)
    need = _lib.lib().vfa_lateral_conv_workspace_bytes(n, h, w)
    key = (dev.index, _lib.current_stream(dev).cuda_stream, need)
    ws = _lateral_ws.get(key)  # (one per (device, stream, size): the three scales of a frame are in flight on one stream together)
    if ws is None:
        ws = _lateral_ws[key] = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
    _launch("vfa_lateral_conv_f32", _lib.ptr(feat), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(gamma), _lib.ptr(beta), float(eps),
            _lib.ptr(out), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(ws), ws.numel(), n, K, h, w, _lib.current_stream_handle(),
            tag=(n, K, h, w))
    return out, scale, shift


def lateral_convs(branches):
    """``lateral_conv`` for ALL the scales of a frame in three launches (``vfa_lateral_convs_f32``): branches = [(feat, weight, bias,
    gamma, beta, eps), ...] (at most three; the C side orders them by descending K: the deepest map first) -> [(y, scale, shift), ...], bit for bit the per-scale calls
    (reference vfanet.py:72-74 for the three scales).  The small maps' workgroups fill the tail of the large map's launch."""
    if not branches:
        return []
    if len(branches) > 3:
        return [lateral_conv(*b) for b in branches]
    dev = branches[0][0].device
    feats, weights, biases, gammas, betas, epss, outs, scales, shifts, wss, Ks, hws = [], [], [], [], [], [], [], [], [], [], [], []
    n = branches[0][0].shape[0]
    for feat, weight, bias, gamma, beta, eps in branches:
        _lib.require_device(feat, weight, bias, gamma, beta)
        feat, weight = _f32c(feat), _f32c(weight.reshape(weight.shape[0], -1))
        assert feat.shape[0] == n and tuple(weight.shape) == (256, feat.shape[1])
        _, K, h, w = feat.shape
        feats.append(feat), weights.append(weight), biases.append(_f32c(bias)), gammas.append(_f32c(gamma)), betas.append(_f32c(beta))
        epss.append(float(eps)), Ks.append(int(K)), hws.extend((int(h), int(w)))
        outs.append(torch.empty((n, h, w, 256), dtype=torch.float32, device=dev))
        scales.append(torch.empty((n, 256), dtype=torch.float32, device=dev))
        shifts.append(torch.empty((n, 256), dtype=torch.float32, device=dev))
        need = _lib.lib().vfa_lateral_conv_workspace_bytes(n, h, w)
        key = (dev.index, _lib.current_stream(dev).cuda_stream, need, len(wss))  # (one per scale: they are in flight together)
        ws = _lateral_ws.get(key)
        if ws is None:
            ws = _lateral_ws[key] = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
        wss.append(ws)
    m = len(branches)
    _launch("vfa_lateral_convs_f32", m, _lib.ptr_array(feats), _lib.ptr_array(weights), _lib.ptr_array(biases), _lib.ptr_array(gammas),
            _lib.ptr_array(betas), (ctypes.c_float * m)(*epss), _lib.ptr_array(outs), _lib.ptr_array(scales), _lib.ptr_array(shifts),
            _lib.ptr_array(wss), (ctypes.c_size_t * m)(*[w.numel() for w in wss]), n, _lib.int_array(Ks), _lib.int_array(hws),
            _lib.current_stream_handle(), tag=(n, tuple(Ks), tuple(hws)))
    return list(zip(outs, scales, shifts))


def box_params(calibs, grid_flat, z_layers, corner_off, conv_kind, image_wh, feat_hw, crange=(-1, 0.95)):
    """-> box (n,nl,cells,4), area (n,nl,cells), visible (n,nl,cells) uint8 (reference vfa_op.py:64-106)."""
    _lib.require_device(calibs, grid_flat, z_layers, corner_off)
    calibs = _f32c(calibs.reshape(-1, 12))
    grid_flat, z_layers, corner_off = _f32c(grid_flat.reshape(-1, 3)), _f32c(z_layers), _f32c(corner_off.reshape(8, 3))
    n, n_cells, nl = calibs.shape[0], grid_flat.shape[0], z_layers.numel()
    dev = calibs.device
    box = torch.empty((n, nl, n_cells, 4), dtype=torch.float32, device=dev)
    area = torch.empty((n, nl, n_cells), dtype=torch.float32, device=dev)
    visible = torch.empty((n, nl, n_cells), dtype=torch.uint8, device=dev)
    _launch("vfa_box_params_f32", _lib.ptr(calibs), _lib.ptr(grid_flat), _lib.ptr(z_layers), _lib.ptr(corner_off),
              n, n_cells, nl, int(conv_kind), float(image_wh[0]), float(image_wh[1]), int(feat_hw[0]), int(feat_hw[1]),
              float(crange[0]), float(crange[1]), _lib.ptr(box), _lib.ptr(area), _lib.ptr(visible),
              _lib.current_stream_handle())
    return box, area, visible


def gather(integral, box, area, visible, cell_begin=0, cell_count=None, layout=_lib.VOX_LAYER_MAJOR):
    """Box pooling from precomputed box parameters -> vox (n, cell_count, nl*C) (reference vfa_op.py:112-120)."""
    _lib.require_device(integral, box, area, visible)
    n, Hp, Wp, C = integral.shape
    _, nl, n_cells, _ = box.shape
    cell_count = n_cells - cell_begin if cell_count is None else cell_count
    vox = torch.empty((n, cell_count, nl * C), dtype=torch.float32, device=integral.device)
    _launch("vfa_gather_f32", _lib.ptr(integral), _lib.ptr(box), _lib.ptr(area), _lib.ptr(visible), _lib.ptr(vox),
              n, C, Hp - 2, Wp - 2, nl, n_cells, cell_begin, cell_count, layout, _lib.current_stream_handle())
    return vox


GATHER_KERNEL = os.environ.get("VFA_AMD_GATHER", "auto")  # auto | default | direct | tap_cache
_gather_choice = {}


def _pick_gather_kernel(integral, calibs, grid_flat, z_layers, corner_off, conv_kind, image_wh, crange, cell_begin,
                        cell_count):
    """Time the two pooling kernels on this problem (HIP events, one synchronise) and return the faster one's name."""
    n_cells = grid_flat.shape[0]
    count = n_cells - cell_begin if cell_count is None else cell_count
    # the whole problem when its vox fits 1 GiB (every launch of a shape then has one size), else a leading sample of cells
    sample = min(count, max(4096, (1 << 30) // max(1, integral.shape[0] * z_layers.numel() * integral.shape[3] * 4)))
    scratch = torch.empty((integral.shape[0], sample, z_layers.numel() * integral.shape[3]), dtype=torch.float32,
                          device=integral.device)
    best, best_ms = "direct", None
    for name in ("direct", "tap_cache"):
        args = (integral, calibs, grid_flat, z_layers, corner_off, conv_kind, image_wh, crange, cell_begin, sample)
        project_gather(*args, out=scratch, kernel=name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            project_gather(*args, out=scratch, kernel=name)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        if best_ms is None or ms < best_ms:
            best, best_ms = name, ms
    return best


def project_gather(integral, calibs, grid_flat, z_layers, corner_off, conv_kind, image_wh, crange=(-1, 0.95),
                   cell_begin=0, cell_count=None, layout=_lib.VOX_LAYER_MAJOR, out=None, kernel=None):
    """Fused projection + box pooling -> vox (n, cell_count, nl*C) (reference vfa_op.py:64-120).

    ``kernel``: "direct", "tap_cache" (identical results; see include/vfa_hip.h), "default" (the library's static
    rule) or None = ``GATHER_KERNEL`` (env VFA_AMD_GATHER, default "auto": both kernels are timed once per problem shape
    on first use -- which one wins depends on how many taps neighbouring boxes share and how many are masked)."""
    n, Hp, Wp, C = integral.shape
    if kernel is None:
        kernel = GATHER_KERNEL
    if kernel == "auto":
        kernel = "default"
        if C == 256 and (layout & 0xff) == _lib.VOX_LAYER_MAJOR and not torch.cuda.is_current_stream_capturing():
            cells = grid_flat.shape[0] - cell_begin if cell_count is None else cell_count
            key = (integral.device.index, n, Hp, Wp, z_layers.numel(), cells, int(conv_kind), tuple(image_wh), tuple(crange))
            kernel = _gather_choice.get(key)
            if kernel is None:
                kernel = _gather_choice[key] = _pick_gather_kernel(integral, calibs, grid_flat, z_layers, corner_off,
                                                                   conv_kind, image_wh, crange, cell_begin, cell_count)
    if kernel != "default":
        layout = layout | {"direct": _lib.VOX_KERNEL_DIRECT, "tap_cache": _lib.VOX_KERNEL_TAP_CACHE}[kernel]
    _lib.require_device(integral, calibs, grid_flat, z_layers, corner_off)
    n, Hp, Wp, C = integral.shape
    n_cells, nl = grid_flat.shape[0], z_layers.numel()
    cell_count = n_cells - cell_begin if cell_count is None else cell_count
    vox = out if out is not None else torch.empty((n, cell_count, nl * C), dtype=torch.float32,
                                                  device=integral.device)
    _launch("vfa_project_gather_f32", _lib.ptr(integral), _lib.ptr(calibs), _lib.ptr(grid_flat), _lib.ptr(z_layers),
              _lib.ptr(corner_off), _lib.ptr(vox), n, C, Hp - 2, Wp - 2, nl, n_cells, cell_begin, cell_count,
              int(conv_kind), float(image_wh[0]), float(image_wh[1]), float(crange[0]), float(crange[1]), layout,
              _lib.current_stream_handle(), tag=(n, C, Hp - 2, Wp - 2, nl, cell_count))
    return vox


def project_collapse(integral, calibs, grid_flat, z_layers, corner_off, weight_t, conv_kind, image_wh,
                     crange=(-1, 0.95), out=None):
    """Fused projection + box pooling + collapse product -> lin (n, cells, 256), no bias / ReLU (reference
    vfa_op.py:64-123).  ``weight_t`` (nl*C, C_out) = transpose of the layer-major collapse weight.  C = C_out = 256 only."""
    _lib.require_device(integral, calibs, grid_flat, z_layers, corner_off, weight_t)
    n, Hp, Wp, C = integral.shape
    n_cells, nl = grid_flat.shape[0], z_layers.numel()
    c_out = weight_t.shape[1]
    assert weight_t.shape[0] == nl * C and weight_t.is_contiguous()
    lin = out if out is not None else torch.empty((n, n_cells, c_out), dtype=torch.float32, device=integral.device)
    _launch("vfa_project_collapse_f32", _lib.ptr(integral), _lib.ptr(calibs), _lib.ptr(grid_flat), _lib.ptr(z_layers),
            _lib.ptr(corner_off), _lib.ptr(weight_t), _lib.ptr(lin), n, C, Hp - 2, Wp - 2, nl, n_cells, c_out,
            int(conv_kind), float(image_wh[0]), float(image_wh[1]), float(crange[0]), float(crange[1]),
            _lib.current_stream_handle(), tag=(n, C, Hp - 2, Wp - 2, nl, n_cells))
    return lin


def project_gather_ws(integral, calibs, grid_flat, z_layers, corner_off, conv_kind, image_wh, crange=(-1, 0.95),
                      cell_begin=0, cell_count=None, layout=_lib.VOX_LAYER_MAJOR, out=None, workspace=None):
    """Two-kernel form of ``project_gather`` (records through HBM, scalar-loaded by the pooling waves)."""
    _lib.require_device(integral, calibs, grid_flat, z_layers, corner_off)
    n, Hp, Wp, C = integral.shape
    n_cells, nl = grid_flat.shape[0], z_layers.numel()
    cell_count = n_cells - cell_begin if cell_count is None else cell_count
    vox = out if out is not None else torch.empty((n, cell_count, nl * C), dtype=torch.float32,
                                                  device=integral.device)
    need = _lib.lib().vfa_gather_workspace_bytes(n, nl, cell_count)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), dtype=torch.uint8, device=integral.device)
    _launch("vfa_project_gather_ws_f32", _lib.ptr(integral), _lib.ptr(calibs), _lib.ptr(grid_flat), _lib.ptr(z_layers),
            _lib.ptr(corner_off), _lib.ptr(vox), _lib.ptr(workspace), workspace.numel(), n, C, Hp - 2, Wp - 2, nl, n_cells,
            cell_begin, cell_count, int(conv_kind), float(image_wh[0]), float(image_wh[1]), float(crange[0]),
            float(crange[1]), layout, _lib.current_stream_handle(), tag=(n, C, Hp - 2, Wp - 2, nl, cell_count))
    return vox


def project_gather_backward(grad_vox, integral_shape, calibs, grid_flat, z_layers, corner_off, conv_kind, image_wh,
                            crange=(-1, 0.95), cell_begin=0, cell_count=None, out=None, accumulate=False, kernel=None, grid_w=0):
    """d vox (n, cell_count, nl*C) layer-major -> d integral (n, Hf+2, Wf+2, C) by scatter-add (float atomics).
    ``kernel="direct"`` selects the per-box atomic kernel instead of the LDS-privatised one (C = 256).  ``grid_w``: cells per row of
    the ground grid the cells come from (row-major; 0 = unknown) -- the LDS-privatised scatter then works on patches of 4 x 8 cells
    (``vfa_project_gather_backward_grid_f32``) instead of 32 cells in a line."""
    _lib.require_device(grad_vox, calibs, grid_flat, z_layers, corner_off, out)
    n, Hp, Wp, C = integral_shape
    n_cells, nl = grid_flat.shape[0], z_layers.numel()
    cell_count = n_cells - cell_begin if cell_count is None else cell_count
    grad_vox = _f32c(grad_vox)
    if out is None:
        out = torch.empty(tuple(integral_shape), dtype=torch.float32, device=grad_vox.device)
        accumulate = False
    grid_w = int(grid_w) if grid_w and n_cells % int(grid_w) == 0 else 0
    _launch("vfa_project_gather_backward_grid_f32", _lib.ptr(grad_vox), _lib.ptr(calibs), _lib.ptr(grid_flat),
            _lib.ptr(z_layers), _lib.ptr(corner_off), _lib.ptr(out), n, C, Hp - 2, Wp - 2, nl, n_cells, cell_begin,
            cell_count, grid_w, int(conv_kind), float(image_wh[0]), float(image_wh[1]), float(crange[0]), float(crange[1]),
            (_lib.BWD_ACCUMULATE if accumulate else 0) | (_lib.VOX_KERNEL_DIRECT if kernel == "direct" else 0),
            _lib.current_stream_handle())
    return out


def integral_image_backward(grad_integral):
    """d integral (n, Hf+2, Wf+2, C) -> d feature (n, C, Hf, Wf).  Destroys ``grad_integral`` (scans it in place)."""
    _lib.require_device(grad_integral)
    grad_integral = _f32c(grad_integral)
    n, Hp, Wp, C = grad_integral.shape
    grad_feature = torch.empty((n, C, Hp - 2, Wp - 2), dtype=torch.float32, device=grad_integral.device)
    _launch("vfa_integral_image_backward_f32", _lib.ptr(grad_integral), _lib.ptr(grad_feature), n, C, Hp - 2, Wp - 2,
            _lib.current_stream_handle())
    return grad_feature


def relu_mask_backward(grad, lin, bias):
    """-> (grad_lin (n,M,N) = grad * (lin + bias > 0), grad_bias (N) or None).  Backward of both epilogues."""
    _lib.require_device(grad, lin, bias)
    n, M, N = lin.shape
    grad = _f32c(grad)
    if N % 4 != 0 or 1024 % N != 0:  # shapes outside the kernel's fast path: plain torch on the GPU
        pre = lin if bias is None else lin + bias
        g = grad.unsqueeze(0) * (pre > 0)
        return g, (None if bias is None else g.sum(dim=(0, 1)))
    glin = torch.empty_like(lin)
    gbias = None if bias is None else torch.empty_like(bias)
    _launch("vfa_relu_mask_backward_f32", _lib.ptr(grad), _lib.ptr(lin), _lib.ptr(bias), _lib.ptr(glin), _lib.ptr(gbias),
            n, M, N, _lib.current_stream_handle())
    return glin, gbias


_grad_w_ws = {}


def grad_weight(g_lin, vox, out=None, accumulate=False):
    """out (256, K) (+)= g_lin^T . vox: the weight gradient of ``collapse`` (autograd of nn.Linear's weight, reference vfa_op.py:59, :123
    under trainer.py:41).  g_lin (rows, 256) the masked output gradient, vox (rows, K) the voxel features of the same rows, K a
    multiple of 256.  Six bf16 MFMA products of a three-piece split (sgemm class), fixed summation order (``vfa_grad_weight_f32``)."""
    _lib.require_device(g_lin, vox, out)
    g_lin, vox = _f32c(g_lin), _f32c(vox)
    rows, K = vox.shape
    assert tuple(g_lin.shape) == (rows, 256) and K % 256 == 0
    dev = vox.device
    if out is None:
        out = torch.empty((256, K), dtype=torch.float32, device=dev)
        accumulate = False
    assert tuple(out.shape) == (256, K) and out.is_contiguous() and out.dtype == torch.float32
    need = _lib.lib().vfa_grad_weight_workspace_bytes(rows, K)
    key = (dev.index, _lib.current_stream(dev).cuda_stream)
    ws = _grad_w_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _grad_w_ws[key] = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
    _launch("vfa_grad_weight_f32", _lib.ptr(g_lin), _lib.ptr(vox), _lib.ptr(out), rows, K, 1 if accumulate else 0, _lib.ptr(ws),
            ws.numel(), _lib.current_stream_handle(), tag=(rows, K))
    return out


def grad_input(g_lin, w, out=None):
    """out (rows, K) = g_lin (rows, 256) . w (256, K): the gradient of ``collapse``'s input (autograd of nn.Linear's input, reference
    vfa_op.py:123 under trainer.py:41), six bf16 MFMA products of a three-piece split (``vfa_grad_input_f32``)."""
    _lib.require_device(g_lin, w, out)
    g_lin, w = _f32c(g_lin), _f32c(w)
    rows, K = g_lin.shape[0], w.shape[1]
    assert g_lin.shape[1] == 256 and w.shape[0] == 256 and K % 256 == 0
    dev = g_lin.device
    if out is None:
        out = torch.empty((rows, K), dtype=torch.float32, device=dev)
    assert tuple(out.shape) == (rows, K) and out.is_contiguous() and out.dtype == torch.float32
    need = _lib.lib().vfa_grad_input_workspace_bytes(K)
    key = (dev.index, _lib.current_stream(dev).cuda_stream, "x")
    ws = _grad_w_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _grad_w_ws[key] = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
    _launch("vfa_grad_input_f32", _lib.ptr(g_lin), _lib.ptr(w), _lib.ptr(out), rows, K, _lib.ptr(ws), ws.numel(),
            _lib.current_stream_handle(), tag=(rows, K))
    return out


def bias_relu_accumulate(lin, bias, out=None, accumulate=False):
    """out (M,N) (+)= sum_v relu(lin[v] + bias) (reference vfa_op.py:124, vfanet.py:82)."""
    _lib.require_device(lin, bias, out)
    n, M, N = lin.shape
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=lin.device)
        accumulate = False
    _launch("vfa_bias_relu_accumulate_f32", _lib.ptr(lin), _lib.ptr(bias), _lib.ptr(out), n, M, N,
              1 if accumulate else 0, _lib.current_stream_handle())
    return out


def scale_view_sum(lin8, lin16, lin32, b8, b16, b32, out=None, accumulate=False):
    """ortho (M,N) (+)= sum_v ((relu(lin8+b8) + relu(lin16+b16)) + relu(lin32+b32)) (reference vfanet.py:79, 82)."""
    _lib.require_device(lin8, lin16, lin32, b8, b16, b32, out)
    n, M, N = lin8.shape
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=lin8.device)
        accumulate = False
    _launch("vfa_scale_view_sum_f32", _lib.ptr(lin8), _lib.ptr(lin16), _lib.ptr(lin32), _lib.ptr(b8), _lib.ptr(b16),
              _lib.ptr(b32), _lib.ptr(out), n, M, N, 1 if accumulate else 0, _lib.current_stream_handle())
    return out


def collapse_relu_sum(vox, weight, bias, out=None, accumulate=False, terms=0, reserved_cus=0):
    """out (M,N) (+)= sum_v relu(vox[v] @ weight.T + bias) in one bf16-split MFMA kernel (K = N = 256 only; reference
    vfa_op.py:121-124 + vfanet.py:82).  Raises ``VFAHipError`` (VFA_ERR_UNSUPPORTED) for other shapes."""
    _lib.require_device(vox, weight, bias, out)
    vox, weight = _f32c(vox), _f32c(weight)
    bias = None if bias is None else _f32c(bias)
    n, M, K = vox.shape
    N = weight.shape[0]
    assert weight.shape == (N, K), (tuple(weight.shape), K)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=vox.device)
        accumulate = False
    _launch("vfa_collapse_relu_sum_f32", _lib.ptr(vox), _lib.ptr(weight), _lib.ptr(bias),
            _lib.ptr(out), n, M, K, N, 1 if accumulate else 0, _lib.collapse_flags(terms, reserved_cus),
            _lib.current_stream_handle())
    return out


_gemm_ws = {}


def collapse_gemm(vox2d, weight, out=None, terms=0, reserved_cus=0):
    """lin (M,N) = vox2d (M,K) @ weight (N,K).T as a bf16-split MFMA tile GEMM (N = 256, K a multiple of 128; reference
    vfa_op.py:121-123 without bias).  Raises ``VFAHipError`` (VFA_ERR_UNSUPPORTED) for other shapes."""
    _lib.require_device(vox2d, weight, out)
    vox2d, weight = _f32c(vox2d), _f32c(weight)
    M, K = vox2d.shape
    N = weight.shape[0]
    assert weight.shape == (N, K), (tuple(weight.shape), K)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=vox2d.device)
    need = _lib.lib().vfa_collapse_gemm_workspace_bytes(K, N)
    key = (vox2d.device.index, _lib.current_stream(vox2d.device).cuda_stream)
    ws = _gemm_ws.get(key)
    if ws is None or ws.numel() < need:  # one scratch buffer per (device, stream): calls on a stream are ordered
        ws = _gemm_ws[key] = torch.empty(max(need, 1), dtype=torch.uint8, device=vox2d.device)
    _launch("vfa_collapse_gemm_f32", _lib.ptr(vox2d), _lib.ptr(weight), _lib.ptr(out), _lib.ptr(ws), ws.numel(), M, K, N,
            _lib.collapse_flags(terms, reserved_cus), _lib.current_stream_handle(), tag=(M, K, N))
    return out


def collapse_gemm_relu_backward(vox, weight, bias, grad_out, terms=0, reserved_cus=0, absmax=None, shift=None):
    """Training backward behind the fused forward: vox (n, cells, K), weight (256, K) (columns in the order of vox), bias (256) or
    None, grad_out (cells, 256) -> (grad_lin (n, cells, 256) = (vox . W^T + b > 0) ? grad_out : 0, grad_bias (256)) with the ReLU mask
    as the epilogue of the recomputed product (``vfa_collapse_gemm_relu_backward_f32``): the pre-activations never reach memory.
    ``absmax`` (the feature statistics of this scale's integral images, int32) selects the product of the FUSED FRAME KERNELS
    (``vfa_collapse_gemm_relu_backward_f16_f32``: fp16 pieces under the frame's scales); with ``shift`` = this chunk's rows of
    ``sliver_shifts`` ((n or 1, cells) uint8) the mask is then the forward's bit for bit."""
    _lib.require_device(vox, weight, bias, grad_out)
    vox, weight, grad_out = _f32c(vox), _f32c(weight), _f32c(grad_out)
    n, cells, K = vox.shape
    assert tuple(weight.shape) == (256, K) and tuple(grad_out.shape) == (cells, 256)
    bias = None if bias is None else _f32c(bias)
    dev = vox.device
    glin = torch.empty((n, cells, 256), dtype=torch.float32, device=dev)
    gbias = torch.zeros(256, dtype=torch.float32, device=dev)
    need = _lib.lib().vfa_collapse_gemm_workspace_bytes(K, 256)
    key = (dev.index, _lib.current_stream(dev).cuda_stream)
    ws = _gemm_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _gemm_ws[key] = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    if absmax is not None:
        _lib.require_device(absmax, shift)
        assert absmax.dtype == torch.int32 and absmax.is_contiguous()
        rows = tiles = None
        if shift is not None:
            assert shift.dtype == torch.uint8 and shift.shape[-1] == cells and shift.numel() in (cells, n * cells)
            rows = shift.reshape(-1, cells).expand(n, cells).contiguous().view(-1)  # one byte per row of the product (view, cell)
            pad = (-rows.numel()) % 128
            tiles = (torch.nn.functional.pad(rows, (0, pad)) if pad else rows).view(-1, 128).amax(dim=1).contiguous()
        _launch("vfa_collapse_gemm_relu_backward_f16_f32", _lib.ptr(vox), _lib.ptr(weight), _lib.ptr(bias) if bias is not None else None,
                _lib.ptr(grad_out), _lib.ptr(glin), _lib.ptr(gbias), _lib.ptr(ws), ws.numel(), n, cells, K, 256, _lib.ptr(absmax),
                absmax.numel(), _lib.ptr(rows) if rows is not None else None, _lib.ptr(tiles) if tiles is not None else None,
                _lib.collapse_flags(0, reserved_cus), _lib.current_stream_handle(), tag=(n, cells, K, "f16"))
        return glin, gbias
    _launch("vfa_collapse_gemm_relu_backward_f32", _lib.ptr(vox), _lib.ptr(weight), _lib.ptr(bias) if bias is not None else None,
            _lib.ptr(grad_out), _lib.ptr(glin), _lib.ptr(gbias), _lib.ptr(ws), ws.numel(), n, cells, K, 256,
            _lib.collapse_flags(terms, reserved_cus), _lib.current_stream_handle(), tag=(n, cells, K))
    return glin, gbias


def sliver_shifts(calibs, grid, z_layers, corner_off, conv_kind, image_wh, feat_hw, per_item, crange=(-1, 0.95)):
    """The sliver shifts of a frame per output row of the collapse product (``vfa_sliver_shifts_u8``): uint8 (n_views, L*W) for the
    serial frame kernel's items (``per_item`` True: single-layer grids, one scale) or (1, L*W) for the pipelined kernel's (tile, scale)
    over all views and layers -- what ``collapse_gemm_relu_backward(..., absmax=..., shift=...)`` needs to repeat the forward's scaling."""
    _lib.require_device(calibs, grid, z_layers, corner_off)
    grid = _f32c(grid.reshape(grid.shape[-3], grid.shape[-2], 3))
    L, W = grid.shape[:2]
    calibs = _f32c(calibs.reshape(-1, 12))
    n = calibs.shape[0]
    z_layers, corner_off = _f32c(z_layers.reshape(-1)), _f32c(corner_off.reshape(8, 3))
    out = torch.empty((n if per_item else 1, L * W), dtype=torch.uint8, device=calibs.device)
    need = _lib.lib().vfa_sliver_shifts_scratch_bytes(L, W)
    scratch = torch.empty(max(need, 16), dtype=torch.uint8, device=calibs.device)
    _launch("vfa_sliver_shifts_u8", _lib.ptr(calibs), _lib.ptr(grid), _lib.ptr(z_layers), z_layers.numel(), _lib.ptr(corner_off), n, L, W,
            int(conv_kind), float(image_wh[0]), float(image_wh[1]), float(crange[0]), float(crange[1]), int(feat_hw[0]), int(feat_hw[1]),
            1 if per_item else 0, _lib.ptr(out), _lib.ptr(scratch), scratch.numel(), _lib.current_stream_handle(), tag=(n, L, W))
    return out


def frame_records(calibs, grid, z_layers, corner_off, conv_kind, image_wh, feat_hws, weights=None, crange=(-1, 0.95),
                  workspace=None, row_slots=None, cuts=True, terms=0):
    """Geometry of one frame for the fused inference kernel (``pool_collapse``): box records of every (view, cell) for each
    feature scale + the split collapse weights -> workspace tensor (reference vfa_op.py:64-106, set-up of :112-115).

    calibs (n,3,4), grid (L,W,3) or (1,L,W,3), feat_hws = [(Hf,Wf), ...] (1..3 scales), weights = one (256,256) per scale.
    ``cuts=False``: only the boxes (``vfa_frame_boxes_f32``: what the pre-pass of ``pool_collapse`` needs); ``frame_cuts`` then
    adds the work cuts and the weight split."""
    _lib.require_device(calibs, grid, z_layers, corner_off)
    grid = _f32c(grid.reshape(grid.shape[-3], grid.shape[-2], 3))
    L, W = grid.shape[:2]
    calibs = _f32c(calibs.reshape(-1, 12))
    n = calibs.shape[0]
    z_layers, corner_off = _f32c(z_layers), _f32c(corner_off.reshape(8, 3))
    if z_layers.numel() != 1:
        raise _lib.VFAHipError("frame_records / pool_collapse cover single-layer grids (nl = 1) only")
    ns = len(feat_hws)
    need = _lib.lib().vfa_frame_workspace_bytes(n, L, W, ns)
    if row_slots is not None:  # a smaller workspace: fewer pooled-row slots for direct items (the rest takes the second launch)
        lay = frame_workspace_layout(n, L, W, ns)
        need = lay["rows"] + min(int(row_slots), lay["rows_cap"]) * 32 * 256 * 4
        workspace = None
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), dtype=torch.uint8, device=calibs.device)
    hw = _lib.int_array([v for f in feat_hws for v in f])
    wts = None
    if weights is not None:
        weights = [_f32c(w) for w in weights]
        assert len(weights) == ns and all(tuple(w.shape) == (256, 256) for w in weights)
        _lib.require_device(*weights)
        wts = _lib.ptr_array(weights)
    if not cuts:
        _launch("vfa_frame_boxes_f32", _lib.ptr(calibs), _lib.ptr(grid), _lib.ptr(z_layers), _lib.ptr(corner_off), n, L, W,
                int(conv_kind), float(image_wh[0]), float(image_wh[1]), float(crange[0]), float(crange[1]), ns, hw,
                _lib.ptr(workspace), workspace.numel(), _lib.current_stream_handle(), tag=(n, L, W, ns))
        return workspace
    _launch("vfa_frame_records_f32", _lib.ptr(calibs), _lib.ptr(grid), _lib.ptr(z_layers), _lib.ptr(corner_off), n, L, W,
            int(conv_kind), float(image_wh[0]), float(image_wh[1]), float(crange[0]), float(crange[1]), ns, hw, wts, int(terms) & 0xf,
            _lib.ptr(workspace), workspace.numel(), _lib.current_stream_handle(), tag=(n, L, W, ns))
    return workspace


def frame_cuts(workspace, n_views, grid_lw, n_scales, weights=None, terms=0):
    """Second half of ``frame_records(..., cuts=False)``: the work cuts of the persistent kernel + the split collapse weights."""
    _lib.require_device(workspace)
    wts = None
    if weights is not None:
        weights = [_f32c(w) for w in weights]
        assert len(weights) == n_scales and all(tuple(w.shape) == (256, 256) for w in weights)
        _lib.require_device(*weights)
        wts = _lib.ptr_array(weights)
    _launch("vfa_frame_cuts_f32", int(n_views), int(grid_lw[0]), int(grid_lw[1]), int(n_scales), wts, int(terms) & 0xf, _lib.ptr(workspace),
            workspace.numel(), _lib.current_stream_handle(), tag=(int(n_views), int(grid_lw[0]), int(grid_lw[1]), int(n_scales)))
    return workspace


def pool_collapse(integrals, biases, workspace, grid_lw, out=None, accumulate=False, terms=0, reserved_cus=0, debug=0, stage="all",
                  absmax=None, dump_vox=False):
    """out (L*W, 256) (+)= sum_scale sum_view relu(vox . W^T + b): pooling + collapse + ReLU + view / scale sum in one
    persistent kernel, the voxel features never touch HBM (reference vfa_op.py:112-125, vfanet.py:79, 82).

    integrals = one (n, Hf+2, Wf+2, 256) zero-bordered channels-last integral image per scale (``integral_image``);
    workspace = ``frame_records`` of the same frame.  ``stage``: "all", or the entry point in two calls -- "rows" (the pre-pass
    over the direct items: needs only the boxes of the frame; returns None) and later "main" (everything else)."""
    _lib.require_device(*integrals, workspace, out)
    stage_flag = {"all": 0, "rows": _lib.FLAG_ROWS_ONLY, "main": _lib.FLAG_SKIP_ROWS}[stage]
    ns = len(integrals)
    n = integrals[0].shape[0]
    L, W = grid_lw
    assert all(i.shape[0] == n and i.shape[3] == 256 and i.is_contiguous() and i.dtype == torch.float32 for i in integrals)
    if out is None and stage != "rows":
        out = torch.empty((L * W, 256), dtype=torch.float32, device=integrals[0].device)
        accumulate = False
    # (the kernel gets a raw pointer)
    assert out is None or (out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (L * W, 256)), \
        "out must be a contiguous fp32 (L*W, 256) tensor"
    biases = [None if b is None else _f32c(b) for b in (biases if biases is not None else [None] * ns)]
    hw = _lib.int_array([v for i in integrals for v in (i.shape[1] - 2, i.shape[2] - 2)])
    absmax, absmax_ptrs = _absmax_of(integrals, absmax)
    _launch("vfa_pool_collapse_relu_sum_f32", _lib.ptr_array(list(integrals)), absmax_ptrs, _lib.ptr_array(biases), _lib.ptr(workspace),
            workspace.numel(), _lib.ptr(out) if out is not None else None, n, L, W, ns, hw, 1 if accumulate else 0,
            _lib.collapse_flags(terms, reserved_cus) | ((int(debug) & 0xfff) << 16) | stage_flag | (_lib.FLAG_DUMP_VOX if dump_vox else 0),  # debug: diagnostic build, tools/ only
            _lib.current_stream_handle(), tag=(n, L, W, tuple((i.shape[1] - 2, i.shape[2] - 2) for i in integrals), stage))
    return out


def pool_windows(integral, workspace, grid_lw, n_scales, scale, out=None):
    """Box pooling of one scale from a ``frame_records`` workspace -> vox (n, L*W, 256) fp32, bit-identical to
    ``project_gather`` (reference vfa_op.py:112-120): tap windows through LDS, one quarter of a (view, tile) per workgroup."""
    _lib.require_device(integral, workspace, out)
    n, Hp, Wp, C = integral.shape
    assert C == 256 and integral.is_contiguous() and integral.dtype == torch.float32
    L, W = grid_lw
    vox = out if out is not None else torch.empty((n, L * W, 256), dtype=torch.float32, device=integral.device)
    _launch("vfa_pool_windows_f32", _lib.ptr(integral), _lib.ptr(workspace), workspace.numel(), _lib.ptr(vox), n, L, W,
            int(n_scales), int(scale), Hp - 2, Wp - 2, _lib.current_stream_handle(), tag=(n, C, Hp - 2, Wp - 2, 1, L * W))
    return vox


def frame_workspace_layout(n_views, L, W, n_scales):
    """Offsets inside the ``frame_records`` workspace, for tests and tools: dict with per-scale lists ``live``, ``direct``,
    ``hdrs``, ``recs``, ``wfrag``, ``overflow`` and ``diag``, ``total``, ``counter``, ``rows``, ``rows_cap``, ``tiles_l``,
    ``tiles_w``, ``max_slots``."""
    import ctypes
    off = (ctypes.c_size_t * 25)()
    tiles = (ctypes.c_int * 4)()
    _lib.call("vfa_frame_workspace_layout", int(n_views), int(L), int(W), int(n_scales), off, tiles)
    names = ("live", "direct", "hdrs", "recs", "wfrag")
    out = {nm: [int(off[5 * k + i]) for k in range(n_scales)] for i, nm in enumerate(names)}
    out["overflow"] = [int(off[17 + k]) for k in range(n_scales)]
    out.update(diag=int(off[15]), total=int(off[16]), counter=int(off[20]), rows=int(off[21]), rows_cap=int(off[22]),
               chunks=int(off[23]), ranks=int(off[24]),
               tiles_l=int(tiles[0]), tiles_w=int(tiles[1]), max_slots=int(tiles[2]), n_chunks=int(tiles[3]))
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# the frame as a producer / consumer pipeline, any number of z-layers (vfa_pipe.hip)
# ---------------------------------------------------------------------------------------------------------------------------------
def pipe_workspace_bytes(n_views, L, W, n_layers, n_scales):
    return int(_lib.lib().vfa_pipe_workspace_bytes(int(n_views), int(L), int(W), int(n_layers), int(n_scales)))


def pipe_records(calibs, grid, z_layers, corner_off, conv_kind, image_wh, feat_hws, weights=None, crange=(-1, 0.95), workspace=None,
                 cuts=True, terms=0):
    """Geometry of one frame for ``pipe_collapse``: box records and tap-window headers of every (view, cell, layer) for each
    feature scale, the work cuts and the split collapse weights -> workspace (reference vfa_op.py:64-106).

    calibs (n,3,4), grid (L,W,3) or (1,L,W,3), z_layers (nl), feat_hws = [(Hf,Wf), ...] (1..3 scales), weights = one
    (256, 256*nl) per scale in the REFERENCE column order c*nl + layer (``collapse.weight`` as it is).  ``cuts=False``: the
    boxes only (``pipe_cuts`` adds the rest).  ``terms``: the product variant ``pipe_collapse`` will be called with (6 = three bf16
    pieces per operand: smaller LDS tap windows, so the geometry has to know)."""
    _lib.require_device(calibs, grid, z_layers, corner_off)
    grid = _f32c(grid.reshape(grid.shape[-3], grid.shape[-2], 3))
    L, W = grid.shape[:2]
    calibs = _f32c(calibs.reshape(-1, 12))
    n = calibs.shape[0]
    z_layers, corner_off = _f32c(z_layers.reshape(-1)), _f32c(corner_off.reshape(8, 3))
    nl, ns = z_layers.numel(), len(feat_hws)
    need = pipe_workspace_bytes(n, L, W, nl, ns)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), dtype=torch.uint8, device=calibs.device)
        # a fresh workspace has no balance state (the geometry calls never touch it: whoever allocates clears it, like
        # `vfa_pipe_balance_f32` mode 0): garbage there could pass for bounds if its tag and signature happened to match
        if need > 0:
            bal = pipe_workspace_layout(n, L, W, nl, ns)["balance"]
            workspace[bal:bal + BALANCE_STATE_BYTES].zero_()
    hw = _lib.int_array([v for f in feat_hws for v in f])
    args = (_lib.ptr(calibs), _lib.ptr(grid), _lib.ptr(z_layers), nl, _lib.ptr(corner_off), n, L, W, int(conv_kind),
            float(image_wh[0]), float(image_wh[1]), float(crange[0]), float(crange[1]), ns, hw)
    if not cuts:
        _launch("vfa_pipe_boxes_f32", *args, int(terms) & 0xf, _lib.ptr(workspace), workspace.numel(), _lib.current_stream_handle(),
                tag=(n, L, W, nl, ns))
        return workspace
    wts = None
    if weights is not None:
        weights = [_f32c(w) for w in weights]
        assert len(weights) == ns and all(tuple(w.shape) == (256, 256 * nl) for w in weights)
        _lib.require_device(*weights)
        wts = _lib.ptr_array(weights)
    _launch("vfa_pipe_records_f32", *args, wts, int(terms) & 0xf, _lib.ptr(workspace), workspace.numel(), _lib.current_stream_handle(),
            tag=(n, L, W, nl, ns))
    return workspace


def pipe_cuts(workspace, n_views, grid_lw, n_layers, n_scales, weights=None, terms=0):
    """Second half of ``pipe_records(..., cuts=False)``: the work cuts of the persistent kernel + the split collapse weights."""
    _lib.require_device(workspace)
    wts = None
    if weights is not None:
        weights = [_f32c(w) for w in weights]
        assert len(weights) == n_scales and all(tuple(w.shape) == (256, 256 * n_layers) for w in weights)
        _lib.require_device(*weights)
        wts = _lib.ptr_array(weights)
    _launch("vfa_pipe_cuts_f32", int(n_views), int(grid_lw[0]), int(grid_lw[1]), int(n_layers), int(n_scales), wts, int(terms) & 0xf, _lib.ptr(workspace),
            workspace.numel(), _lib.current_stream_handle(), tag=(int(n_views), int(grid_lw[0]), int(grid_lw[1]), int(n_layers), int(n_scales)))
    return workspace


def pipe_collapse(integrals, biases, workspace, grid_lw, n_layers, out=None, accumulate=False, terms=0, reserved_cus=0, debug=0,
                  absmax=None, dump_vox=False):
    """out (L*W, 256) (+)= sum_scale sum_view relu(vox . W^T + b) for K = n_layers * 256: pooling, collapse, ReLU, view and scale
    sums in one persistent kernel (pooling waves and matrix waves side by side); the voxel features never touch HBM
    (reference vfa_op.py:110-125, vfanet.py:79, 82).  workspace = ``pipe_records`` of the same frame."""
    _lib.require_device(*integrals, workspace, out)
    ns = len(integrals)
    n = integrals[0].shape[0]
    L, W = grid_lw
    assert all(i.shape[0] == n and i.shape[3] == 256 and i.is_contiguous() and i.dtype == torch.float32 for i in integrals)
    if out is None:
        out = torch.empty((L * W, 256), dtype=torch.float32, device=integrals[0].device)
        accumulate = False
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (L * W, 256)
    biases = [None if b is None else _f32c(b) for b in (biases if biases is not None else [None] * ns)]
    hw = _lib.int_array([v for i in integrals for v in (i.shape[1] - 2, i.shape[2] - 2)])
    absmax, absmax_ptrs = _absmax_of(integrals, absmax)
    _launch("vfa_pipe_collapse_relu_sum_f32", _lib.ptr_array(list(integrals)), absmax_ptrs, _lib.ptr_array(biases), _lib.ptr(workspace),
            workspace.numel(), _lib.ptr(out), n, L, W, int(n_layers), ns, hw, 1 if accumulate else 0,
            _lib.collapse_flags(terms, reserved_cus) | ((int(debug) & 0xfff) << 16) | (_lib.FLAG_DUMP_VOX if dump_vox else 0),
            _lib.current_stream_handle(), tag=(n, L, W, int(n_layers), tuple((i.shape[1] - 2, i.shape[2] - 2) for i in integrals)))
    return out


def pipe_balance(workspace, n_views, grid_lw, n_layers, n_scales, reserved_cus=0, reset=False):
    """Balance state of a ``pipe_records`` workspace (``vfa_pipe_balance_f32``).  ``reset``: clear it (once, on a fresh workspace);
    otherwise, from the work cuts ``pipe_records`` has just left there: the bounds of the workgroups' shares that minimise the
    heaviest share -- used by every ``pipe_collapse`` launch of the same size on a frame with the same cuts (static cameras and
    grid); deterministic."""
    _lib.require_device(workspace)
    L, W = grid_lw
    _launch("vfa_pipe_balance_f32", int(n_views), int(L), int(W), int(n_layers), int(n_scales), int(reserved_cus), 0 if reset else 1,
            _lib.ptr(workspace), workspace.numel(), _lib.current_stream_handle())


BALANCE_STATE_BYTES = 8192  # (bounds + launch size + cost signature, and at byte 4096 the workgroups' cycle counts)


def pipe_workspace_layout(n_views, L, W, n_layers, n_scales):
    """Offsets inside the ``pipe_records`` workspace, for tests and tools."""
    import ctypes
    off = (ctypes.c_size_t * 22)()
    tiles = (ctypes.c_int * 5)()
    _lib.call("vfa_pipe_workspace_layout", int(n_views), int(L), int(W), int(n_layers), int(n_scales), off, tiles)
    names = ("live", "hdrs", "recs", "wfrag")
    out = {nm: [int(off[4 * k + i]) for k in range(n_scales)] for i, nm in enumerate(names)}
    out.update(tickets=int(off[12]), globs=int(off[17]), chunks=int(off[13]), ranks=int(off[14]), diag=int(off[15]), total=int(off[16]),
               tiles_l=int(tiles[0]), tiles_w=int(tiles[1]), max_slots=int(tiles[2]), n_chunks=int(tiles[3]),
               max_slots_3piece=int(tiles[4]), balance=int(off[18]), shifts=[int(off[19 + k]) for k in range(n_scales)])
    return out
