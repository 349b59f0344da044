"""``VFA`` -- the voxel feature projector, MI355X-native, behind the reference's Python interface.

Drop-in for the reference's ``vfa/model/vfa_op.py`` ``class VFA`` (:46-125): same constructor, same
``forward(feature, calib, grid, crange, visualize)`` signature, same ``state_dict`` keys / shapes / dtypes
(``z_corners`` int64 (nl,1,1,3), ``corners_offset`` f32 (1,1,1,1,8,3), ``collapse.weight`` (C, C*nl),
``collapse.bias`` (C)), same error for an unknown dataset name.  What differs is where the work runs:

    reference (stock torch ops, one camera at a time)          here (hand-written HIP, gfx950)
    ---------------------------------------------------------  -------------------------------------------
    cumsum(cumsum(f,-1),-2)                    vfa_op.py:173    vfa_integral_image_f32  (channels-last out)
    corners + convert + project + clamp + bbox :64-88, :104-106 } vfa_project_gather_f32 (one wave per box,
    4 x F.grid_sample + box mean + mask        :112-120         }   box parameters never leave the wave)
    nn.Linear                                  :123             rocBLAS/hipBLASLt fp32 GEMM (torch.matmul)
    relu, scale sum, view sum                  :124; vfanet.py:79,82   vfa_bias_relu_accumulate_f32 /
                                                                        vfa_scale_view_sum_f32

All cameras of one scale are processed by ONE launch of each kernel (``project_views``); the reference's
per-camera ``forward`` is the ``n_views == 1`` case.  There is no CPU path: CPU tensors raise.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib, lazy, ops
from .utils import project  # noqa: F401  (re-exported like the reference module does)

EPSILON = 1e-6
MAXIMUM_AREA_RATIO = 0.3

# bound on the transient voxel-feature buffer (n_views, cells, nl*C) fp32; larger grids are chunked over cells
VOX_BYTES_LIMIT = int(os.environ.get("VFA_AMD_VOX_BYTES", str(8 << 30)))
# fused pooling + fp32-MFMA collapse kernel (256 -> 256 channels, no gradient needed): never materialises vox, so it is
# the memory-lean path for very large grids.  On MI355X it runs at the speed of pooling kernel + library GEMM (fp32 MFMA
# is clock/power-bound either way), so it is opt-in: VFA_AMD_FUSED=1.
USE_FUSED = os.environ.get("VFA_AMD_FUSED", "0") == "1"
# `collapse` on single-layer grids (K = N = 256) when no gradient is needed: "mfma_bf16" (default) = the hand-written
# bf16-split MFMA kernel fused with bias, ReLU and the view sum (include/vfa_hip.h: vfa_collapse_relu_sum_f32);
# on multi-layer grids and in training (K = nl*C a multiple of 128, N = 256) the product alone runs as the K-looped MFMA
# tile GEMM `vfa_collapse_gemm_f32` in front of the epilogue kernels; "library" = fp32 library GEMM everywhere.
COLLAPSE_KERNEL = os.environ.get("VFA_AMD_COLLAPSE", "mfma_bf16")
# Arithmetic of the product in the fused frame kernels (VFA_FLAG_TERMS): 2 (default) = two fp16 pieces per operand with a
# power-of-two scale, three MFMA products -- the error of an fp32 sgemm, i.e. the arithmetic width of the reference's nn.Linear
# (vfa_op.py:59, :123); 3 / 4 = two bf16 pieces (16 bits, ~4e-6 normwise), 6 = three bf16 pieces (pipelined kernel only).  The unfused
# product kernels (`vfa_collapse_gemm_f32`, `vfa_collapse_relu_sum_f32`: the training backward's mask, VFA_AMD_PIPE=0 paths) have the
# bf16 forms only and read 2 / 6 as 3.
COLLAPSE_TERMS = int(os.environ.get("VFA_AMD_COLLAPSE_TERMS", "2"))
# Inference on single-layer grids with C = 256: "1" (default) = geometry once per frame (`vfa_frame_records_f32`) + ONE
# persistent kernel for pooling, collapse, bias, ReLU and the view / scale sums (`vfa_pool_collapse_relu_sum_f32`): the voxel
# features never reach HBM.  "0" = pooling kernel -> vox in HBM -> MFMA collapse kernel, per scale (the round-1 path).
FUSED_POOL = os.environ.get("VFA_AMD_FUSED_POOL", "1") == "1"
# With FUSED_POOL off: "1" (default) = the per-frame box records also feed the standalone LDS-window pooling kernel
# (`vfa_pool_windows_f32`, voxel features bit-identical to `vfa_project_gather_f32`) in front of the MFMA collapse kernel;
# "0" = the round-1 pooling kernels that project every box themselves.
WINDOW_POOL = os.environ.get("VFA_AMD_WINDOW_POOL", "1") == "1"


def _conv_kind(args):
    """Dataset name -> world-unit conversion (reference ``convert``, vfa_op.py:37-44)."""
    name = getattr(args, "data", None)
    if name not in _lib.CONV_KIND:
        # the reference falls off the end of its if/elif chain and hits an unbound local (vfa_op.py:38-44)
        raise UnboundLocalError(f"local variable 'coord' referenced before assignment (args.data={name!r} is not one "
                                f"of {sorted(_lib.CONV_KIND)})")
    return _lib.CONV_KIND[name]


class _IntegralImage(torch.autograd.Function):
    """(n,C,Hf,Wf) feature maps -> (n,Hf+2,Wf+2,C) zero-bordered channels-last integral images."""

    @staticmethod
    def forward(ctx, feature):
        return ops.integral_image(feature)

    @staticmethod
    def backward(ctx, grad_integral):
        # d/df of a double cumsum = reverse double cumsum of the incoming gradient (interior only); the kernel scans
        # its input in place, so hand it a private copy unless autograd already gave us a temporary
        return ops.integral_image_backward(grad_integral.clone())


class _BoxPool(torch.autograd.Function):
    """Integral images + camera geometry -> voxel features (n, cell_count, nl*C), column = layer*C + c."""

    @staticmethod
    def forward(ctx, integral, calibs, grid_flat, z_layers, corner_off, geom, cell_begin, cell_count):
        conv_kind, img_w, img_h, cmin, cmax = geom[:5]
        vox = ops.project_gather(integral, calibs, grid_flat, z_layers, corner_off, conv_kind, (img_w, img_h),
                                 (cmin, cmax), cell_begin, cell_count, _lib.VOX_LAYER_MAJOR)
        ctx.save_for_backward(calibs, grid_flat, z_layers, corner_off)
        ctx.meta = (geom, tuple(integral.shape), cell_begin, cell_count)
        return vox

    @staticmethod
    def backward(ctx, grad_vox):
        calibs, grid_flat, z_layers, corner_off = ctx.saved_tensors
        geom, shape, cell_begin, cell_count = ctx.meta
        conv_kind, img_w, img_h, cmin, cmax = geom[:5]
        grad_integral = ops.project_gather_backward(grad_vox, shape, calibs, grid_flat, z_layers, corner_off, conv_kind,
                                                    (img_w, img_h), (cmin, cmax), cell_begin, cell_count,
                                                    grid_w=geom[5] if len(geom) > 5 else 0)
        return grad_integral, None, None, None, None, None, None, None


class _CollapseGemm(torch.autograd.Function):
    """lin (M,N) = vox (M,K) @ weight (N,K)^T.  Forward: the hand-written bf16-split MFMA tile GEMM
    (``vfa_collapse_gemm_f32``); backward: the hand-written gradient products of vfa_grad.hip (six bf16 MFMA products of a three-piece
    split, sgemm class: ``vfa_grad_input_f32``, ``vfa_grad_weight_f32``) where their shapes fit (N = 256, K a multiple of 256), the
    fp32 library products otherwise."""

    @staticmethod
    def forward(ctx, vox2d, weight, reserved_cus=0):
        ctx.save_for_backward(vox2d, weight)
        return ops.collapse_gemm(vox2d, weight, terms=_unfused_terms(), reserved_cus=reserved_cus)

    @staticmethod
    def backward(ctx, grad):
        vox2d, weight = ctx.saved_tensors
        grad = grad.contiguous()
        hand = grad.shape[1] == 256 and weight.shape[0] == 256 and weight.shape[1] % 256 == 0 and grad.shape[0] > 0
        g_vox = g_w = None
        if ctx.needs_input_grad[0]:
            g_vox = ops.grad_input(grad, weight) if hand else torch.matmul(grad, weight)
        if ctx.needs_input_grad[1]:
            g_w = ops.grad_weight(grad, vox2d) if hand else torch.matmul(grad.t(), vox2d)
        return g_vox, g_w, None


SIDE_STREAM = os.environ.get("VFA_AMD_SIDE_STREAM", "1") == "1"
_side_streams = {}


def _side_stream(dev):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev, priority=-1)  # its small kernels go in front of the queued integral-image blocks
    return _side_streams[key]


def _fused_terms():
    """``COLLAPSE_TERMS`` as the serial fused kernel takes it (it has no three-piece form: 6 runs as the fp16 default)."""
    return 2 if COLLAPSE_TERMS in (0, 2, 6) else COLLAPSE_TERMS


def _unfused_terms(terms=None):
    """... and as the unfused bf16 product kernels take it (2 and 6 run as 3)."""
    terms = COLLAPSE_TERMS if terms is None else terms
    return 3 if terms in (0, 2, 6) else terms


def fused_frame_ok(mods, n_views, mode="fused"):
    """The per-frame-records inference paths (``mode`` "fused": one persistent kernel for everything behind the integral
    images; "window": LDS-window pooling kernel + MFMA collapse kernel per scale) cover these projector modules (one per
    feature scale) for this many cameras."""
    m0 = mods[0]
    on = FUSED_POOL if mode == "fused" else (WINDOW_POOL and not FUSED_POOL)
    return (on and COLLAPSE_KERNEL != "library" and 1 <= len(mods) <= 3 and 0 < n_views <= 32
            and all(m.channel == 256 and m.num_grid_layer == 1 and m.collapse.out_features == 256 for m in mods)
            and all(m.geometry_key == m0.geometry_key and getattr(m.args, "data", None) == getattr(m0.args, "data", None)
                    and tuple(m.args.image_size) == tuple(m0.args.image_size) for m in mods))


def fused_frame(mods, features, calibs, grid, crange=(-1, 0.95), out=None, accumulate=False, reserved_cus=0, integrals=None):
    """All scales and all cameras of one frame: ``out (L*W, 256) (+)= sum_scale sum_view relu(collapse_scale(vox))``.

    mods / features: one ``VFA`` and one (n,256,Hf,Wf) lateral batch per scale.  Integral images (one launch pair per
    scale), then ``ops.frame_records`` (geometry once per frame) and ``ops.pool_collapse`` (everything else).  Inference
    only (no autograd); needs ``fused_frame_ok``.  ``integrals``: the zero-bordered channels-last integral images when the
    caller already has them (producer fusion, ``VFANet.lateral_integrals``); ``features`` is then ignored."""
    if integrals is not None:
        features = [i.permute(0, 3, 1, 2)[:, :, 1:-1, 1:-1] for i in integrals]  # views: only their shapes are read below
    _lib.require_device(calibs, grid, *features)
    m0 = mods[0]
    conv_kind = _conv_kind(m0.args)
    img_h, img_w = (float(v) for v in m0.args.image_size)  # the path uses image_size[::-1] (vfa_op.py:75)
    length, width = grid.shape[-3], grid.shape[-2]
    dev = features[0].device
    z_layers, corner_off = m0._kernel_geometry(dev)
    with torch.no_grad():
        # geometry (camera + grid only) on a second HIP stream, beside the integral images (bandwidth-bound, feature maps only)
        cur = _lib.current_stream(dev)
        # (inside a hipGraph capture the frame stays on ONE stream: a replayed graph pays more for its cross-stream edges than the
        # overlap of geometry and integral images gives -- 0.44 against 0.20 ms per frame on a one-camera bench frame)
        side = _side_stream(dev) if SIDE_STREAM and integrals is None and not torch.cuda.is_current_stream_capturing() else cur
        weights = [m.layer_major_weight() for m in mods]
        ws = torch.empty(max(_lib.lib().vfa_frame_workspace_bytes(calibs.shape[0], length, width, len(mods)), 1),
                         dtype=torch.uint8, device=dev)
        side.wait_stream(cur)
        feat_hws = [tuple(f.shape[-2:]) for f in features]
        if side is cur:
            ops.frame_records(calibs, grid, z_layers, corner_off, conv_kind, (img_w, img_h), feat_hws, weights=weights,
                              crange=crange, workspace=ws, terms=_fused_terms())
            boxes_done = None
        else:
            # two joins: the pre-pass over the direct items needs the boxes only; the work cuts (a single-workgroup kernel that
            # is slow beside the bandwidth-bound integral images) are waited for one kernel later
            with torch.cuda.stream(side):
                ops.frame_records(calibs, grid, z_layers, corner_off, conv_kind, (img_w, img_h), feat_hws, crange=crange,
                                  workspace=ws, cuts=False)
                boxes_done = torch.cuda.Event()
                boxes_done.record(side)
                ops.frame_cuts(ws, calibs.shape[0], (length, width), len(mods), weights=weights, terms=_fused_terms())
        if integrals is None:
            integrals = ops.integral_images(features)  # all strides in one launch pair
        biases = [m.collapse.bias for m in mods]
        if boxes_done is None:
            return ops.pool_collapse(integrals, biases, ws, (length, width), out=out, accumulate=accumulate, terms=_fused_terms(),
                                     reserved_cus=reserved_cus)
        cur.wait_event(boxes_done)
        ops.pool_collapse(integrals, biases, ws, (length, width), terms=_fused_terms(), reserved_cus=reserved_cus, stage="rows")
        cur.wait_stream(side)
        return ops.pool_collapse(integrals, biases, ws, (length, width), out=out, accumulate=accumulate, terms=_fused_terms(),
                                 reserved_cus=reserved_cus, stage="main")


# Inference with C = 256 on ANY number of z-layers: "1" (default) = geometry once per frame (`vfa_pipe_records_f32`) + ONE persistent
# producer / consumer kernel (`vfa_pipe_collapse_relu_sum_f32`: pooling waves beside matrix waves, the accumulators of four views
# in registers over all layers); "0" = the older paths (FUSED_POOL on single-layer grids, vox through HBM on multi-layer ones).
PIPE = os.environ.get("VFA_AMD_PIPE", "1") == "1"
# ... on single-layer grids too ("0": the serial kernel of vfa_fused.hip keeps them, `FUSED_POOL`)
PIPE_SINGLE_LAYER = os.environ.get("VFA_AMD_PIPE_SINGLE_LAYER", "0") == "1"
# bound on the per-frame geometry workspace of the pipeline path; larger frames are processed in bands of grid rows
PIPE_WS_LIMIT = int(os.environ.get("VFA_AMD_PIPE_WS_BYTES", str(3 << 30)))


def pipe_frame_ok(mods, n_views, tensors=()):
    """The pipelined per-frame inference path covers these projector modules (one per feature scale; any layer count, the
    same for all) for this many cameras, and no gradient is wanted."""
    m0 = mods[0]
    if not (PIPE and COLLAPSE_KERNEL != "library" and 1 <= len(mods) <= 3 and 0 < n_views <= 32):
        return False
    if m0.num_grid_layer == 1 and not PIPE_SINGLE_LAYER:
        return False
    if not all(m.channel == 256 and m.collapse.out_features == 256 and m.num_grid_layer == m0.num_grid_layer
               and m.geometry_key == m0.geometry_key and getattr(m.args, "data", None) == getattr(m0.args, "data", None)
               and tuple(m.args.image_size) == tuple(m0.args.image_size) for m in mods):
        return False
    if not torch.is_grad_enabled():
        return True
    params = [p for m in mods for p in (m.collapse.weight, m.collapse.bias)] + [t for t in tensors if t is not None]
    return not any(p.requires_grad for p in params)


def pipe_frame(mods, features, calibs, grid, crange=(-1, 0.95), out=None, accumulate=False, reserved_cus=0, integrals=None,
               terms=None):
    """All scales, all cameras and all z-layers of one frame: ``out (L*W, 256) (+)= sum_scale sum_view relu(collapse_scale(vox))``
    (reference vfa_op.py:61-125 for every camera and scale, vfanet.py:79, 82).

    mods / features: one ``VFA`` and one (n,256,Hf,Wf) lateral batch per scale.  Integral images (one launch pair for all
    scales), ``ops.pipe_records`` (geometry once per frame, on a second stream beside them) and ``ops.pipe_collapse`` (everything
    else, one persistent kernel).  Inference only; needs ``pipe_frame_ok``.  Frames whose geometry workspace would exceed
    ``PIPE_WS_LIMIT`` are processed in bands of grid rows (the bands are independent: every output row belongs to one).
    ``terms``: product variant (default ``COLLAPSE_TERMS``; 6 = three bf16 pieces per operand, the arithmetic width of the reference's
    fp32 ``nn.Linear``, at twice the matrix work)."""
    terms = COLLAPSE_TERMS if terms is None else int(terms)
    if integrals is not None:
        features = [i.permute(0, 3, 1, 2)[:, :, 1:-1, 1:-1] for i in integrals]  # views: only their shapes are read below
    _lib.require_device(calibs, grid, *features)
    m0 = mods[0]
    conv_kind = _conv_kind(m0.args)
    img_h, img_w = (float(v) for v in m0.args.image_size)  # the path uses image_size[::-1] (vfa_op.py:75)
    if grid.dim() < 3:
        raise ValueError("pipe_frame needs the grid as (L, W, 3) or (1, L, W, 3): the tiles follow its rows and columns")
    grid = grid.reshape(grid.shape[-3], grid.shape[-2], 3)
    length, width = grid.shape[0], grid.shape[1]
    dev = features[0].device
    n, nl, ns = calibs.shape[0], m0.num_grid_layer, len(mods)
    z_layers, corner_off = m0._kernel_geometry(dev)
    if out is None:
        out = torch.empty((length * width, 256), dtype=torch.float32, device=dev)
        accumulate = False
    if out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape) != (length * width, 256):
        raise ValueError("pipe_frame: out must be a contiguous fp32 (L*W, 256) tensor")
    if length * width == 0:
        return out
    feat_hws = [tuple(f.shape[-2:]) for f in features]
    # bands of grid rows (multiples of the 4-row tiles) that keep the geometry workspace under the limit
    rows = length
    while rows > 4 and ops.pipe_workspace_bytes(n, rows, width, nl, ns) > PIPE_WS_LIMIT:
        rows = max(4, ((rows + 1) // 2 + 3) // 4 * 4)
    with torch.no_grad():
        cur = _lib.current_stream(dev)
        # (inside a hipGraph capture the frame stays on ONE stream: a replayed graph pays more for its cross-stream edges than the
        # overlap of geometry and integral images gives -- 0.44 against 0.20 ms per frame on a one-camera bench frame)
        side = _side_stream(dev) if SIDE_STREAM and integrals is None and not torch.cuda.is_current_stream_capturing() else cur
        weights = [m.collapse.weight for m in mods]  # reference layout: the weight-split kernel reads column c * nl + layer
        biases = [m.collapse.bias for m in mods]
        band_rows = min(rows, length)
        n_bands = (length + rows - 1) // rows
        st = _pipe_state(dev, (n, length, width, band_rows, nl, ns, terms, int(reserved_cus), conv_kind, tuple(feat_hws)),
                         ops.pipe_workspace_bytes(n, band_rows, width, nl, ns), n_bands)
        ws = st["ws"]
        balancing = PIPE_BALANCE and st["frames"] == 0 and not torch.cuda.is_current_stream_capturing()  # (a one-off: never part of a graph)
        for b, r0 in enumerate(range(0, length, rows)):
            r1 = min(length, r0 + rows)
            band = grid[r0:r1]
            if side is not cur:
                side.wait_stream(cur)  # (the previous band's -- or frame's -- kernel has read the workspace)
            with torch.cuda.stream(side):
                ops.pipe_records(calibs, band, z_layers, corner_off, conv_kind, (img_w, img_h), feat_hws, weights=weights,
                                 crange=crange, workspace=ws, terms=terms)
            if integrals is None:
                integrals = ops.integral_images(features)  # all strides in one launch pair, beside the geometry
            if side is not cur:
                cur.wait_stream(side)
            if n_bands > 1:  # one workspace for all bands: every band has its own balance state (and a shorter last band its own layout)
                bal_off = st["balance_off"] if r1 - r0 == band_rows else ops.pipe_workspace_layout(n, r1 - r0, width, nl, ns)["balance"]
                ws[bal_off:bal_off + ops.BALANCE_STATE_BYTES].copy_(st["bands"][b], non_blocking=True)
            if balancing:
                # First frame of a geometry: the shares of the kernel's workgroups that minimise the heaviest one (from this frame's
                # work cuts; deterministic).  Cameras and grid of a frame stream stand still, so every later frame finds bounds
                # that carry the signature of its own cuts; a frame with other cuts does not match and runs the uniform split.
                ops.pipe_balance(ws, n, (r1 - r0, width), nl, ns, reserved_cus=reserved_cus)
                if n_bands > 1:
                    st["bands"][b].copy_(ws[bal_off:bal_off + ops.BALANCE_STATE_BYTES], non_blocking=True)
            ops.pipe_collapse(integrals, biases, ws, (r1 - r0, width), nl, out=out[r0 * width:r1 * width], accumulate=accumulate,
                              terms=terms, reserved_cus=reserved_cus)
        st["frames"] += 1
        if side is not cur:
            ws.record_stream(side)
    return out


# "1" (default): the first frame of a geometry computes balanced work shares for the pipelined kernel's workgroups
# (`vfa_pipe_balance_f32`), kept with the persistent workspace; "0": the uniform split of the work cuts.
PIPE_BALANCE = os.environ.get("VFA_AMD_PIPE_BALANCE", "1") == "1"
# (device, stream, shapes) -> persistent workspace (+ balance state per band).  The most recent geometries are kept: a caller that
# loops over per-scale `project_sum` / `VFA.forward` calls cycles through one key per feature scale and camera set -- with two
# entries every call missed, re-allocated, re-zeroed and re-balanced (a single-workgroup kernel of milliseconds per call)
PIPE_STATES_KEPT = int(os.environ.get("VFA_AMD_PIPE_STATES", "8"))
_pipe_states = {}
_pipe_pinned = []  # ... and those a captured hipGraph replays into, when the capture named no owner (`owned_capture_states`)
_capture_owner = None


class owned_capture_states:
    """Context manager around a hipGraph capture: the workspaces the captured frames replay into are collected in ``self.states``
    instead of the process-wide ``_pipe_pinned``, so they live exactly as long as whoever keeps this object beside the graph
    (``vfa_amd.graph.GraphedAggregate`` does) -- a re-capture after a shape change frees the old ones.  Inside ONE such capture, frames
    of the same geometry share one workspace (they are ordered on the capture stream like eager frames on a stream) and find the
    balanced shares of the first."""

    def __init__(self):
        self.states = []

    def __enter__(self):
        global _capture_owner
        self._outer, _capture_owner = _capture_owner, self
        return self

    def __exit__(self, *exc):
        global _capture_owner
        _capture_owner = self._outer
        return False


def _pipe_state(dev, key, ws_bytes, n_bands):
    # one workspace per (geometry, stream): the geometry call of a frame on another stream must not overwrite records the
    # previous frame's kernel on THIS stream is still reading (`side.wait_stream(cur)` only orders against the current one)
    key = (dev.index, _lib.current_stream(dev).cuda_stream) + key
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and _capture_owner is not None:
        for st in _capture_owner.states:  # an earlier frame of this capture with the same geometry
            if st["key"] == key:
                return st
    st = _pipe_states.pop(key, None)
    if st is None and capturing:
        # A capture runs on a stream of its own: it takes over the workspace a warm-up frame of the same geometry left on another
        # stream -- with its balanced shares; a fresh one would have the one-off balance kernel (milliseconds) captured into the graph.
        # (The warm-up frame has finished: ``torch.cuda.graph`` synchronises the device before it begins the capture.)
        for k2 in list(_pipe_states):
            if k2[0] == key[0] and k2[2:] == key[2:] and _pipe_states[k2]["frames"] > 0:
                st = _pipe_states.pop(k2)
                break
    if st is None:
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        lay = ops.pipe_workspace_layout(key[2], key[5], key[4], key[6], key[7])
        st = {"ws": ws, "frames": 0, "balance_off": lay["balance"],
              "bands": [torch.zeros(ops.BALANCE_STATE_BYTES, dtype=torch.uint8, device=dev) for _ in range(n_bands)] if n_bands > 1 else None}
        ws[lay["balance"]:lay["balance"] + ops.BALANCE_STATE_BYTES].zero_()  # (= vfa_pipe_balance_f32 mode 0)
    if capturing:
        # A hipGraph that is being captured replays into this workspace for as long as it lives: the state belongs to that graph
        # from here on and NEVER goes back into `_pipe_states` -- torch hands stream handles out of a pool of 32, so an eager frame
        # (or a second capture) on a stream with the same handle would otherwise pop it and write geometry, tickets and balance state
        # into a workspace a live graph replays into.  Who keeps it alive: the owner the capture named (`owned_capture_states`:
        # freed with the graph), else `_pipe_pinned` for the life of the process (a bare ``torch.cuda.graph`` gives this module
        # nothing to tie the lifetime to: one workspace per captured frame and capture stays allocated).
        st["pinned"] = True
        st["key"] = key
        (_capture_owner.states if _capture_owner is not None else _pipe_pinned).append(st)
        return st
    _pipe_states[key] = st  # (most recent last)
    while len(_pipe_states) > PIPE_STATES_KEPT:
        _pipe_states.pop(next(iter(_pipe_states)))
    return st


# Training: "1" (default) = the forward of a frame is the fused per-frame kernel (no voxel features, no pre-activations kept);
# the backward recomputes them scale by scale from the lateral maps.  "0" = the unfused kernels with `vox` / `lin` saved by autograd.
FUSED_TRAIN = os.environ.get("VFA_AMD_FUSED_TRAIN", "1") == "1"


def _frame_kernels_cover(mods, n_views):
    """Module set / camera count the fused per-frame kernels cover (gradients or not)."""
    m0 = mods[0]
    return (COLLAPSE_KERNEL != "library" and 1 <= len(mods) <= 3 and 0 < n_views <= 32
            and all(m.channel == 256 and m.collapse.out_features == 256 and m.num_grid_layer == m0.num_grid_layer
                    and m.geometry_key == m0.geometry_key and getattr(m.args, "data", None) == getattr(m0.args, "data", None)
                    and tuple(m.args.image_size) == tuple(m0.args.image_size) for m in mods))


class _FusedFrameTrain(torch.autograd.Function):
    """The whole frame as ONE autograd node (reference: the camera loop vfanet.py:64-82 around VFA.forward vfa_op.py:61-125,
    trained through by trainer.py:26, 41).

    forward: the fused per-frame kernel (``pipe_frame`` for any layer count, ``fused_frame`` on single-layer grids): the voxel
    features never reach HBM and nothing but the INPUTS is kept for the backward.  backward: per scale, in cell chunks, the
    voxel features are pooled again from the kept integral images (bit-identical to what the forward pooled), ``lin = vox . W^T``
    is formed again with the MFMA tile GEMM (the ReLU mask), then the usual gradients: d lin = d out * (lin + b > 0),
    d W += d lin^T . vox, d b, d vox = d lin . W, scattered back through the box pooling
    (``vfa_project_gather_backward_f32``) and the two cumsums.  With the default arithmetic the recomputed product IS the forward's:
    ``vfa_collapse_gemm_relu_backward_f16_f32`` splits both operands into the same fp16 pieces under the same scales (the saved feature
    statistics give 2^ea, ``vfa_sliver_shifts_u8`` the per-item shifts), starts the accumulator at the same bias 2^(ea+ew-shift) and issues
    the three MFMA products in the same order: the ReLU mask of the backward equals the forward's bit for bit
    (tests/test_hip_backward.py).  The bf16 forms (``VFA_AMD_COLLAPSE_TERMS`` 3 / 4 / 6) recompute with the two-piece bf16 tile GEMM.
    Measured against the reference's own fp32 ``backward()`` (tests/test_reference_gradients.py): d weight / d bias within 6e-7, d feature
    within the reference's own fp32-vs-float64 noise."""

    @staticmethod
    def forward(ctx, calibs, grid, crange, meta, reserved_cus, *tensors):
        mods = meta
        ns = len(mods)
        lats, weights, biases = tensors[:ns], tensors[ns:2 * ns], tensors[2 * ns:3 * ns]
        n = calibs.shape[0]
        with torch.no_grad():
            integrals = ops.integral_images([l.detach() for l in lats])  # kept: the backward pools from them again
            piped = pipe_frame_ok(mods, n)
            frame = pipe_frame if piped else fused_frame
            out = frame(mods, None, calibs, grid, crange, reserved_cus=reserved_cus, integrals=integrals)
        # (the feature statistics too: with them the backward repeats the forward's fp16 product -- scales, shifts, order -- and its ReLU
        # mask is the forward's bit for bit)
        f16 = (COLLAPSE_TERMS if piped else _fused_terms()) in (0, 2)
        ctx.save_for_backward(calibs, grid, *integrals, *weights, *biases, *(integrals.absmax if f16 else ()))
        ctx.meta = (mods, tuple(float(c) for c in crange), piped, f16)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        mods, crange, piped, f16 = ctx.meta
        ns = len(mods)
        calibs, grid = ctx.saved_tensors[:2]
        integrals = ctx.saved_tensors[2:2 + ns]
        weights = ctx.saved_tensors[2 + ns:2 + 2 * ns]
        biases = ctx.saved_tensors[2 + 2 * ns:2 + 3 * ns]
        stats = ctx.saved_tensors[2 + 3 * ns:2 + 4 * ns] if f16 else (None,) * ns
        m0 = mods[0]
        conv_kind = _conv_kind(m0.args)
        img_h, img_w = (float(v) for v in m0.args.image_size)
        n = calibs.shape[0]
        dev = grad_out.device
        grid_flat = grid.reshape(-1, 3).to(dtype=torch.float32).contiguous()
        cal = calibs.reshape(n, 12).to(dtype=torch.float32).contiguous()
        z_layers, corner_off = m0._kernel_geometry(dev)
        n_cells, nl, C = grid_flat.shape[0], m0.num_grid_layer, 256
        grad_out = grad_out.contiguous()
        g_lats, g_ws, g_bs = [], [], []
        with torch.no_grad():
            for k, (m, integral, w, b, stat) in enumerate(zip(mods, integrals, weights, biases, stats)):
                need_lat, need_w, need_b = ctx.needs_input_grad[5 + k], ctx.needs_input_grad[5 + ns + k], ctx.needs_input_grad[5 + 2 * ns + k]
                if not (need_lat or need_w or need_b):
                    g_lats.append(None), g_ws.append(None), g_bs.append(None)
                    continue
                w_lm = w.view(C, C, nl).permute(0, 2, 1).reshape(C, nl * C).contiguous()  # column layer * C + c, like the pooled rows
                g_int = torch.zeros_like(integral) if need_lat else None
                g_w_lm = torch.zeros_like(w_lm) if need_w else None
                g_b = torch.zeros_like(b) if need_b else None
                per_cell = n * nl * C * 4 * 2  # vox and d vox
                chunk = max(1, min(n_cells, VOX_BYTES_LIMIT // max(per_cell, 1)))
                # the sliver shifts the forward's geometry pass gave this scale's items (serial kernel: per (view, tile); pipelined: per
                # tile over views and layers): the recomputed product scales its rows the same way
                shifts = None
                if stat is not None and grid.dim() >= 3:
                    shifts = ops.sliver_shifts(cal, grid, z_layers, corner_off, conv_kind, (img_w, img_h),
                                               (integral.shape[1] - 2, integral.shape[2] - 2), not piped, crange)
                for begin in range(0, n_cells, chunk):
                    count = min(chunk, n_cells - begin)
                    vox = ops.project_gather(integral, cal, grid_flat, z_layers, corner_off, conv_kind, (img_w, img_h), crange,
                                             cell_begin=begin, cell_count=count)
                    # (the product again, as the two-piece bf16 MFMA tile GEMM -- the unfused product kernels have the bf16 forms only:
                    # `_unfused_terms` -- with the ReLU mask as its epilogue: d lin and d b come out, the pre-activations are never written)
                    if count >= 32 and shifts is not None:  # the forward's own product: fp16 pieces, its scales, its shifts
                        g_lin, g_b_part = ops.collapse_gemm_relu_backward(vox, w_lm, b, grad_out[begin:begin + count], absmax=stat,
                                                                          shift=shifts[:, begin:begin + count].contiguous())
                    elif count >= 32:
                        g_lin, g_b_part = ops.collapse_gemm_relu_backward(vox, w_lm, b, grad_out[begin:begin + count], terms=_unfused_terms())
                    else:  # (fewer than 32 cells in the chunk: product, then the mask kernel)
                        lin = ops.collapse_gemm(vox.view(n * count, nl * C), w_lm, terms=_unfused_terms()).view(n, count, C)
                        g_lin, g_b_part = ops.relu_mask_backward(grad_out[begin:begin + count], lin, b)
                        del lin
                    if need_b:
                        g_b += g_b_part
                    g2 = g_lin.view(n * count, C)
                    if need_w:
                        ops.grad_weight(g2, vox.view(n * count, nl * C), out=g_w_lm, accumulate=True)
                    if need_lat:
                        g_vox = ops.grad_input(g2, w_lm).view(n, count, nl * C)
                        ops.project_gather_backward(g_vox, tuple(integral.shape), cal, grid_flat, z_layers, corner_off, conv_kind,
                                                    (img_w, img_h), crange, cell_begin=begin, cell_count=count, out=g_int,
                                                    accumulate=True, grid_w=grid.shape[-2] if grid.dim() >= 3 else 0)
                        del g_vox
                    del vox, g_lin
                g_lats.append(ops.integral_image_backward(g_int) if need_lat else None)
                g_ws.append(g_w_lm.view(C, nl, C).permute(0, 2, 1).reshape(C, C * nl) if need_w else None)
                g_bs.append(g_b)
        return (None, None, None, None, None, *g_lats, *g_ws, *g_bs)


def fused_frame_train(mods, features, calibs, grid, crange=(-1, 0.95), reserved_cus=0):
    """``sum_scale sum_view relu(collapse_scale(vox))`` as (L*W, 256) WITH autograd: fused forward, recomputing backward
    (``_FusedFrameTrain``).  Needs ``_frame_kernels_cover`` and, on single-layer grids, nothing else; see ``FUSED_TRAIN``."""
    tensors = list(features) + [m.collapse.weight for m in mods] + [m.collapse.bias for m in mods]
    return _FusedFrameTrain.apply(calibs, grid, tuple(crange), tuple(mods), reserved_cus, *tensors)


def fused_train_ok(mods, n_views, features):
    """Gradients are wanted and the fused per-frame kernels cover the forward."""
    if not (FUSED_TRAIN and torch.is_grad_enabled() and _frame_kernels_cover(mods, n_views)):
        return False
    if not (PIPE or (mods[0].num_grid_layer == 1 and FUSED_POOL)):
        return False
    tensors = [p for m in mods for p in (m.collapse.weight, m.collapse.bias)] + [f for f in features if f is not None]
    return any(t.requires_grad for t in tensors)


def window_frame(mods, features, calibs, grid, crange=(-1, 0.95), out=None, accumulate=False, reserved_cus=0):
    """Same contract as ``fused_frame``, as separate kernels per scale: geometry once per frame (``ops.frame_records``), then
    per scale the integral images, the LDS-window pooling kernel (``ops.pool_windows``: voxel features in HBM, bit-exact) and
    the MFMA collapse + bias + ReLU + view-sum kernel (``ops.collapse_relu_sum``)."""
    _lib.require_device(calibs, grid, *features)
    m0 = mods[0]
    conv_kind = _conv_kind(m0.args)
    img_h, img_w = (float(v) for v in m0.args.image_size)
    length, width = grid.shape[-3], grid.shape[-2]
    dev = features[0].device
    z_layers, corner_off = m0._kernel_geometry(dev)
    n = calibs.shape[0]
    if out is None:
        out = torch.empty((length * width, 256), dtype=torch.float32, device=dev)
        accumulate = False
    with torch.no_grad():
        ws = ops.frame_records(calibs, grid, z_layers, corner_off, conv_kind, (img_w, img_h),
                               [tuple(f.shape[-2:]) for f in features], weights=None, crange=crange)
        vox = torch.empty((n, length * width, 256), dtype=torch.float32, device=dev)
        integrals = ops.integral_images(features)
        for k, (m, integral) in enumerate(zip(mods, integrals)):
            ops.pool_windows(integral, ws, (length, width), len(mods), k, out=vox)
            ops.collapse_relu_sum(vox, m.layer_major_weight(), m.collapse.bias, out=out, accumulate=accumulate or k > 0,
                                  terms=_unfused_terms(), reserved_cus=reserved_cus)
    return out


def mfma_gemm_ok(K, N):
    """The K-looped MFMA GEMM covers this `collapse` shape (any layer count at C = 256)."""
    return COLLAPSE_KERNEL != "library" and N == 256 and K % 128 == 0


class _BiasReluSum(torch.autograd.Function):
    """out (M,N) = sum_v relu(lin[v] + bias), views in index order (vfa_op.py:124 / vfanet.py:82)."""

    @staticmethod
    def forward(ctx, lin, bias):
        out = ops.bias_relu_accumulate(lin, bias)
        ctx.save_for_backward(lin, bias)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lin, bias = ctx.saved_tensors
        return ops.relu_mask_backward(grad_out, lin, bias)


class VFA(nn.Module):
    def __init__(self, channel, grid_height=160, cube_size=(25, 25, 32), feat_scale=1, args=None):
        super().__init__()
        self.cube_height = cube_size[2]
        # (nl,1,1,3) rows (0, 0, z); dtype follows torch.arange exactly as in the reference (int64 for int sizes)
        z = torch.arange(0, grid_height, cube_size[2])
        z_corners = torch.zeros((z.numel(), 1, 1, 3), dtype=z.dtype)
        z_corners[:, 0, 0, 2] = z
        corners_offset = torch.Tensor(self.generate_cube(cube_size)).view(1, 1, 1, 1, 8, 3)
        self.register_buffer("z_corners", z_corners)
        self.register_buffer("corners_offset", corners_offset)
        self.feat_scale = feat_scale  # kept for interface parity; unused by the reference too (vfa_op.py:57, :74)
        self.args = args
        self.channel = channel
        self.num_grid_layer = z.numel()
        self.collapse = nn.Linear(channel * self.num_grid_layer, channel)
        self._geom_cache = None
        # modules built from the same cube share box geometry (the fused frame path projects each cube once for all scales)
        self.geometry_key = (tuple(float(v) for v in cube_size), float(grid_height))

    # ------------------------------------------------------------------ geometry buffers for the kernels
    def generate_cube(self, cub_size):
        """(8,3) corner offsets: x -++--++-, y --++--++, z 0000hhhh (reference vfa_op.py:127-133)."""
        l, w, h = cub_size
        sx = np.array([-1, 1, 1, -1, -1, 1, 1, -1]) * (l / 2)
        sy = np.array([-1, -1, 1, 1, -1, -1, 1, 1]) * (w / 2)
        sz = np.array([0, 0, 0, 0, 1, 1, 1, 1]) * h
        return np.stack([sx, sy, sz], axis=1)

    def _kernel_geometry(self, device):
        key = (str(device), self.z_corners._version, self.corners_offset._version, self.z_corners.data_ptr())
        if self._geom_cache is None or self._geom_cache[0] != key:
            z_layers = self.z_corners[:, 0, 0, 2].to(device=device, dtype=torch.float32).contiguous()
            corner_off = self.corners_offset.to(device=device, dtype=torch.float32).reshape(8, 3).contiguous()
            self._geom_cache = (key, z_layers, corner_off)
        return self._geom_cache[1], self._geom_cache[2]

    def layer_major_weight(self):
        """collapse.weight with columns reordered from c*nl + layer (reference) to layer*C + c (kernel output)."""
        w = self.collapse.weight
        out_c = w.shape[0]
        return w.view(out_c, self.channel, self.num_grid_layer).permute(0, 2, 1).reshape(out_c, -1)

    # ------------------------------------------------------------------ batched projector
    def project_views(self, features, calibs, grid, crange=(-1, 0.95), reserved_cus=0):
        """All cameras of one scale at once.

        features (n,C,Hf,Wf), calibs (n,3,4), grid (1,L,W,3) or (L,W,3)  ->  lin (n, L*W, C_out) =
        vox . collapse.weight^T, WITHOUT bias and ReLU (the epilogue kernels add them while summing views).
        """
        _lib.require_device(features, calibs, grid)
        conv_kind = _conv_kind(self.args)
        img_h, img_w = (float(v) for v in self.args.image_size)  # the path uses image_size[::-1] (vfa_op.py:75)
        n, C, Hf, Wf = features.shape
        if C != self.channel:
            raise ValueError(f"feature has {C} channels, VFA was built for {self.channel}")
        dev = features.device
        grid_flat = grid.reshape(-1, 3).to(dtype=torch.float32).contiguous()
        calibs = calibs.reshape(n, 12).to(dtype=torch.float32).contiguous()
        z_layers, corner_off = self._kernel_geometry(dev)
        # (+ the width of the ground grid: the backward scatter works on patches of 4 x 8 cells)
        geom = (conv_kind, img_w, img_h, float(crange[0]), float(crange[1]), int(grid.shape[-2]) if grid.dim() >= 3 else 0)
        n_cells, nl = grid_flat.shape[0], self.num_grid_layer

        if n_cells == 0 or n == 0:
            return features.new_zeros((n, n_cells, self.collapse.out_features))
        integral = _IntegralImage.apply(features)
        needs_grad = torch.is_grad_enabled() and (features.requires_grad or self.collapse.weight.requires_grad)
        if USE_FUSED and not needs_grad and C == 256 and self.collapse.out_features == 256:
            # inference: pooling feeds the fp32-MFMA collapse product through LDS, vox never reaches HBM
            w_t = self.layer_major_weight().t().contiguous()
            return ops.project_collapse(integral, calibs, grid_flat, z_layers, corner_off, w_t, conv_kind,
                                        (img_w, img_h), (geom[3], geom[4]))
        w_lm = self.layer_major_weight()
        use_mfma = mfma_gemm_ok(nl * C, self.collapse.out_features)
        per_cell = n * nl * C * 4
        chunk = max(1, min(n_cells, VOX_BYTES_LIMIT // max(per_cell, 1)))
        outs = []
        for begin in range(0, n_cells, chunk):
            count = min(chunk, n_cells - begin)
            vox = _BoxPool.apply(integral, calibs, grid_flat, z_layers, corner_off, geom, begin, count)
            if use_mfma:
                lin = _CollapseGemm.apply(vox.view(n * count, nl * C), w_lm.contiguous(), reserved_cus)
            else:
                lin = torch.matmul(vox.view(n * count, nl * C), w_lm.t())
            outs.append(lin.view(n, count, -1))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)

    def mfma_collapse_ok(self, features=None):
        """True when the inference-only bf16-split MFMA kernel (`vfa_collapse_relu_sum_f32`) covers this module."""
        if COLLAPSE_KERNEL == "library" or self.num_grid_layer * self.channel != 256 or self.collapse.out_features != 256:
            return False
        if not torch.is_grad_enabled():
            return True
        params = (self.collapse.weight, self.collapse.bias) + (() if features is None else (features,))
        return not any(p is not None and p.requires_grad for p in params)

    def project_sum(self, features, calibs, grid, crange=(-1, 0.95), out=None, accumulate=False, reserved_cus=0):
        """Inference path of one scale, all cameras: ``out (L*W, C_out) (+)= sum_v relu(collapse(vox_v))``.

        Integral images -> projection + pooling (HIP) -> `collapse` + bias + ReLU + view sum in ONE MFMA kernel
        (``ops.collapse_relu_sum``): neither ``lin`` nor a separate epilogue pass exists.  Needs ``mfma_collapse_ok``.
        """
        _lib.require_device(features, calibs, grid)
        conv_kind = _conv_kind(self.args)
        img_h, img_w = (float(v) for v in self.args.image_size)
        n, C, Hf, Wf = features.shape
        if C != self.channel:
            raise ValueError(f"feature has {C} channels, VFA was built for {self.channel}")
        grid_flat = grid.reshape(-1, 3).to(dtype=torch.float32).contiguous()
        calibs = calibs.reshape(n, 12).to(dtype=torch.float32).contiguous()
        z_layers, corner_off = self._kernel_geometry(features.device)
        n_cells, nl = grid_flat.shape[0], self.num_grid_layer
        if out is None:
            out = torch.empty((n_cells, self.collapse.out_features), dtype=torch.float32, device=features.device)
            accumulate = False
        if n_cells == 0:
            return out
        if n == 0:
            return out if accumulate else out.zero_()
        if grid.dim() >= 3 and pipe_frame_ok([self], n, (features,)):
            return pipe_frame([self], [features], calibs, grid, crange, out=out, accumulate=accumulate, reserved_cus=reserved_cus)
        if grid.dim() >= 3 and fused_frame_ok([self], n):
            return fused_frame([self], [features], calibs, grid, crange, out=out, accumulate=accumulate,
                               reserved_cus=reserved_cus)
        if grid.dim() >= 3 and fused_frame_ok([self], n, "window") and n * n_cells * C * 4 <= VOX_BYTES_LIMIT:
            return window_frame([self], [features], calibs, grid, crange, out=out, accumulate=accumulate,
                                reserved_cus=reserved_cus)
        with torch.no_grad():
            integral = ops.integral_image(features)
            weight = self.layer_major_weight()
            per_cell = n * nl * C * 4
            chunk = max(1, min(n_cells, VOX_BYTES_LIMIT // max(per_cell, 1)))
            for begin in range(0, n_cells, chunk):
                count = min(chunk, n_cells - begin)
                vox = ops.project_gather(integral, calibs, grid_flat, z_layers, corner_off, conv_kind, (img_w, img_h),
                                         (float(crange[0]), float(crange[1])), cell_begin=begin, cell_count=count)
                ops.collapse_relu_sum(vox, weight, self.collapse.bias, out=out[begin:begin + count],
                                      accumulate=accumulate, terms=_unfused_terms(), reserved_cus=reserved_cus)
        return out

    # ------------------------------------------------------------------ reference interface
    def forward(self, feature, calib, grid, crange=(-1, 0.95), visualize=False):
        """feature (1,C,Hf,Wf), calib (3,4), grid (1,L,W,3) -> (1,C,L,W), ReLU'd (reference vfa_op.py:61-125)."""
        if feature.shape[0] != 1:
            raise ValueError("VFA.forward takes one camera (batch 1) like the reference; use project_views for many")
        length, width = grid.shape[-3], grid.shape[-2]
        if visualize:
            self.visualize_cube(feature, calib, grid, crange)
        if lazy.LAZY and not torch.is_grad_enabled() and grid.dim() >= 3 and feature.is_cuda and calib.is_cuda and grid.is_cuda \
                and feature.shape[1] == self.channel and self._defer_ok():
            # Inference through a per-frame kernel: the result is DEFERRED (vfa_amd/lazy.py) -- the reference's loop over cameras and
            # scales (vfanet.py:64-82) then costs one batched frame instead of 21 launches of a persistent kernel.  (The check is the
            # cheap one on purpose: this branch is taken 21 times per frame on the host.)
            return lazy.DeferredOrtho([(self, feature, lazy.version_of(feature), calib, lazy.version_of(calib))], grid,
                                      (float(crange[0]), float(crange[1])), (1, self.collapse.out_features, length, width),
                                      feature.device)
        if grid.dim() >= 3 and fused_train_ok([self], 1, (feature,)):
            ortho = fused_frame_train([self], [feature], calib.reshape(1, 3, 4), grid, crange)
        elif self.mfma_collapse_ok(feature) or (grid.dim() >= 3 and pipe_frame_ok([self], 1, (feature,))):
            ortho = self.project_sum(feature, calib.reshape(1, 3, 4), grid, crange)
        else:
            lin = self.project_views(feature, calib.reshape(1, 3, 4), grid, crange)
            ortho = _BiasReluSum.apply(lin, self.collapse.bias)
        return ortho.view(1, length, width, self.collapse.out_features).permute(0, 3, 1, 2)

    def _defer_ok(self):
        """A per-frame inference kernel covers this module on its own (what ``_materialize`` can always fall back to), memoised on the
        switches that decide it."""
        key = (PIPE, PIPE_SINGLE_LAYER, FUSED_POOL, COLLAPSE_KERNEL, self.num_grid_layer, self.channel, self.collapse.out_features)
        cached = getattr(self, "_defer_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, bool(pipe_frame_ok([self], 1) or fused_frame_ok([self], 1)))
            self._defer_cache = cached
        return cached[1]

    def extra_repr(self):
        return f"channel={self.channel}, layers={self.num_grid_layer}, data={getattr(self.args, 'data', None)}"

    def visualize_cube(self, feature, calib, grid, crange=(-1, 0.95), viz_interval=10):
        """Slow matplotlib side path (reference vfa_op.py:90-101, 135-168): draws every ``viz_interval``-th box."""
        import matplotlib.pyplot as plt
        import matplotlib.patches as patches
        stages = box_parameters(self, calib.reshape(1, 3, 4), grid, feature.shape[-2:], crange)
        box = ((stages["box"][0].cpu() + 1) / 2).numpy()
        hf, wf = feature.shape[-2:]
        fig, ax = plt.subplots()
        ax.imshow(feature[0].detach().abs().sum(0).cpu().numpy())
        for layer in range(box.shape[0]):
            for l, t, r, b in box[layer, ::viz_interval]:
                ax.add_patch(patches.Rectangle((l * wf, t * hf), (r - l) * wf, (b - t) * hf, fill=False, linewidth=0.5))
        plt.show()
        return fig


def _materialize(terms, grid, crange):
    """The tensor behind a ``lazy.DeferredOrtho``: terms = [(module, feature (1,C,h,w), version, calib (3,4), version), ...] in call order ->
    (1, C_out, L, W) = sum over the terms of ``module.forward`` (reference vfa_op.py:61-125 per term, vfanet.py:79, 82 for the sums).
    The terms are grouped by module; when the (at most three) modules saw the same cameras in the same order -- the reference's loop --
    the whole sum is ONE batched frame (``pipe_frame`` / ``fused_frame``: what ``aggregate_views`` runs), otherwise one batched call
    per module, accumulated."""
    groups = {}
    for mod, feat, version, calib, calib_version in terms:
        if lazy.version_of(feat) != version or lazy.version_of(calib) != calib_version:
            raise RuntimeError("a feature map or calibration passed to VFA.forward was modified in place before its (deferred) "
                               "result was used; clone it, or set VFA_AMD_LAZY=0")
        groups.setdefault(id(mod), (mod, [], []))
        groups[id(mod)][1].append(feat)
        groups[id(mod)][2].append(calib)
    mods = [g[0] for g in groups.values()]
    feats = [g[1][0] if len(g[1]) == 1 else torch.cat(g[1]) for g in groups.values()]
    calib_lists = [g[2] for g in groups.values()]
    length, width = grid.shape[-3], grid.shape[-2]
    dev = feats[0].device
    c_out = mods[0].collapse.out_features
    out = torch.empty((length * width, c_out), dtype=torch.float32, device=dev)
    with torch.no_grad():
        n = len(calib_lists[0])
        same_cameras = all(len(cl) == n and all(a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape) for a, b in zip(cl, calib_lists[0]))
                           for cl in calib_lists[1:])
        calibs0 = torch.stack([c.reshape(3, 4) for c in calib_lists[0]]).to(dtype=torch.float32)
        if same_cameras and pipe_frame_ok(mods, n, feats):
            pipe_frame(mods, feats, calibs0, grid, crange, out=out)
        elif same_cameras and fused_frame_ok(mods, n):
            fused_frame(mods, feats, calibs0, grid, crange, out=out)
        else:
            for i, (m, f, cl) in enumerate(zip(mods, feats, calib_lists)):
                m.project_sum(f, torch.stack([c.reshape(3, 4) for c in cl]).to(dtype=torch.float32), grid, crange, out=out, accumulate=i > 0)
    return out.view(1, length, width, c_out).permute(0, 3, 1, 2)


def box_parameters(module, calibs, grid, feat_hw, crange=(-1, 0.95)):
    """Stage output for tests / visualisation: box (n,nl,cells,4), area, visible of every view (vfa_op.py:64-106)."""
    _lib.require_device(calibs, grid)
    conv_kind = _conv_kind(module.args)
    img_h, img_w = (float(v) for v in module.args.image_size)
    z_layers, corner_off = module._kernel_geometry(calibs.device)
    box, area, visible = ops.box_params(calibs, grid.reshape(-1, 3), z_layers, corner_off, conv_kind, (img_w, img_h),
                                        feat_hw, crange)
    return dict(box=box, area=area, visible=visible.bool())


class FrameGeometry:
    """The geometry of a STATIC camera rig, computed once instead of once per frame.

    The reference projects every cube of the grid through every camera in every ``VFA.forward`` (vfa_op.py:64-106): cameras may move.
    In a deployment they do not -- calibrations, grid and ``collapse`` weights stand still for a whole stream -- and everything the
    frame kernels read besides the feature maps (box records, tap windows, work cuts, the split weights) can be formed ONCE:

    >>> geom = vfa_amd.FrameGeometry([vfa8, vfa16, vfa32], calibs, grid, [(90, 160), (45, 80), (23, 40)])
    >>> ortho = geom.frame([lat8, lat16, lat32])            # per frame: integral images + ONE persistent kernel

    ``frame`` returns the (1, C, L, W) view ``aggregate_views`` returns, bit for bit.  This is an explicit contract of the CALLER
    (nothing here can notice a camera that moved; a changed ``collapse.weight`` / ``bias`` IS noticed -- version counters -- and
    raises): the default paths (``aggregate_views``, ``VFA.forward``, ``bench.py``'s ``value``) recompute the geometry every frame
    like the reference.  One frame at a time per object (its workspace holds the hand-off slots of the launch in flight).
    Inference only."""

    def __init__(self, mods, calibs, grid, feat_hws, crange=(-1, 0.95), reserved_cus=0):
        mods = list(mods)
        _lib.require_device(calibs, grid)
        n = calibs.shape[0]
        with torch.no_grad():  # (inference only: the parameters' requires_grad is not what decides here)
            self.pipe = pipe_frame_ok(mods, n)
        if not self.pipe and not fused_frame_ok(mods, n):
            raise ValueError("FrameGeometry needs a module set the per-frame inference kernels cover (C = 256, one layer count, <= 32 cameras)")
        m0 = mods[0]
        grid3 = grid.reshape(grid.shape[-3], grid.shape[-2], 3)
        self.length, self.width = grid3.shape[0], grid3.shape[1]
        self.mods, self.n, self.nl, self.crange, self.reserved_cus = mods, n, m0.num_grid_layer, crange, int(reserved_cus)
        self.feat_hws = [tuple(int(v) for v in hw) for hw in feat_hws]
        dev = calibs.device
        z_layers, corner_off = m0._kernel_geometry(dev)
        conv_kind = _conv_kind(m0.args)
        img_h, img_w = (float(v) for v in m0.args.image_size)
        with torch.no_grad():
            if self.pipe:
                self.terms = COLLAPSE_TERMS
                if ops.pipe_workspace_bytes(n, self.length, self.width, self.nl, len(mods)) > PIPE_WS_LIMIT:
                    raise ValueError("FrameGeometry: this frame is processed in bands of grid rows (workspace above VFA_AMD_PIPE_WS_BYTES); "
                                     "build one FrameGeometry per band of the grid")
                self.ws = ops.pipe_records(calibs, grid3, z_layers, corner_off, conv_kind, (img_w, img_h), self.feat_hws,
                                           weights=[m.collapse.weight for m in mods], crange=crange, terms=self.terms)
                if PIPE_BALANCE:
                    ops.pipe_balance(self.ws, n, (self.length, self.width), self.nl, len(mods), reserved_cus=self.reserved_cus)
            else:
                self.terms = _fused_terms()
                self.ws = ops.frame_records(calibs, grid3, z_layers, corner_off, conv_kind, (img_w, img_h), self.feat_hws,
                                            weights=[m.layer_major_weight() for m in mods], crange=crange, terms=self.terms)
        self._versions = [(m.collapse.weight._version, m.collapse.weight.data_ptr()) for m in mods]
        # the cameras and the grid are the caller's contract (a static rig), but an in-place edit is cheap to notice
        self._rig = (calibs, grid)
        self._rig_versions = (lazy.version_of(calibs), lazy.version_of(grid))
        # the records were formed on the stream current HERE: a frame on another stream waits for them
        self._built_on = _lib.current_stream(dev).cuda_stream
        self._built = torch.cuda.Event()
        self._built.record(_lib.current_stream(dev))

    def frame(self, features=None, integrals=None, out=None, accumulate=False):
        """features: one (n, 256, Hf, Wf) lateral batch per scale (or ``integrals``: their integral images) -> (1, C, L, W)."""
        if [(m.collapse.weight._version, m.collapse.weight.data_ptr()) for m in self.mods] != self._versions:
            raise RuntimeError("FrameGeometry: a collapse.weight changed since the geometry (which holds its split fragments) was built")
        if (lazy.version_of(self._rig[0]), lazy.version_of(self._rig[1])) != self._rig_versions:
            raise RuntimeError("FrameGeometry: calibs or grid were modified in place since the geometry was built (a static rig is the contract)")
        cur = _lib.current_stream(self.ws.device)
        if cur.cuda_stream != self._built_on:
            cur.wait_event(self._built)
        with torch.no_grad():
            if integrals is None:
                _lib.require_device(*features)
                if [tuple(f.shape[-2:]) for f in features] != self.feat_hws or any(f.shape[0] != self.n for f in features):
                    raise ValueError("FrameGeometry.frame: feature maps of other shapes than the geometry was built for")
                integrals = ops.integral_images(features)
            biases = [m.collapse.bias for m in self.mods]
            lw = (self.length, self.width)
            if self.pipe:
                res = ops.pipe_collapse(integrals, biases, self.ws, lw, self.nl, out=out, accumulate=accumulate, terms=self.terms,
                                        reserved_cus=self.reserved_cus)
            else:
                res = ops.pool_collapse(integrals, biases, self.ws, lw, out=out, accumulate=accumulate, terms=self.terms,
                                        reserved_cus=self.reserved_cus)
        return res.view(1, self.length, self.width, res.shape[-1]).permute(0, 3, 1, 2)
