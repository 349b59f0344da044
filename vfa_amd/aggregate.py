"""Cross-scale / cross-view aggregation of the projected BEV maps (reference ``vfa/model/vfanet.py:64-82``).

The reference loops over cameras in Python, calls ``VFA.forward`` three times per camera and adds the
results (``vfa_feat8 + vfa_feat16 + vfa_feat32`` :79, ``ortho += vfa_feats`` :82).  Here all cameras of a
scale go through one launch of each kernel and one fused epilogue forms

    ortho = sum_cam ((relu(lin8+b8) + relu(lin16+b16)) + relu(lin32+b32))        (same association order)

Multi-GPU: cameras are sharded over ranks (``camera_shard``); every rank forms the partial sum of its
cameras and one RCCL all-reduce over xGMI fuses the grid (``all_reduce_ortho``).  The reference has no
distributed code; this is the data-parallel axis the path offers.
"""
import os

import torch
import torch.distributed as dist

from . import _lib, ops, vfa_op

# The three scale chains (integral image -> projection + pooling -> collapse GEMM) are independent until the final
# sum, so they are issued on separate HIP streams: the MFMA-bound GEMM of one scale overlaps the latency-bound
# pooling kernel of another.  Measured gain on MI355X: 4 % (the GEMM fills the chip), so the default stays 1 stream,
# which also keeps per-kernel timings clean; VFA_AMD_STREAMS=3 enables the overlap.
N_STREAMS = max(1, min(3, int(os.environ.get("VFA_AMD_STREAMS", "1"))))
# Multi-GPU: the collapse kernels are persistent (one workgroup per CU holding all of its LDS), so an RCCL kernel queued
# on another stream could only start at a kernel boundary.  With world_size > 1 they leave this many CUs free, which is
# where the all-reduce of the previous frame runs while this frame is projected.  The library applies the reservation
# only to launches whose number of tile rounds it does not increase, so it costs nothing.
RESERVED_CUS = max(0, min(255, int(os.environ.get("VFA_AMD_RESERVED_CUS", "16"))))
_side_streams = {}


def _streams(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = [torch.cuda.Stream(device=device) for _ in range(2)]
    return _side_streams[key]


class _ScaleViewSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lin8, lin16, lin32, b8, b16, b32):
        ortho = ops.scale_view_sum(lin8, lin16, lin32, b8, b16, b32)
        ctx.save_for_backward(lin8, lin16, lin32, b8, b16, b32)
        return ortho

    @staticmethod
    def backward(ctx, grad):
        lin8, lin16, lin32, b8, b16, b32 = ctx.saved_tensors
        outs, bias_grads = [], []
        for lin, b in ((lin8, b8), (lin16, b16), (lin32, b32)):
            g, gb = ops.relu_mask_backward(grad, lin, b)
            outs.append(g)
            bias_grads.append(gb)
        return (*outs, *bias_grads)


def camera_shard(n_cam, rank=None, world=None):
    """Cameras owned by ``rank``: rank, rank+world, ... (one camera per GPU when world >= n_cam)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    return list(range(rank, n_cam, world))


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x.  Every rank then runs the same heads on the same y, so dL/dx = dL/dy locally."""

    @staticmethod
    def forward(ctx, x, group):
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, grad):
        return grad, None


PREHEAD_PREFIXES = ("base.", "lat8.", "lat16.", "lat32.", "bn8.", "bn16.", "bn32.", "vfa8.", "vfa16.", "vfa32.")


def all_reduce_prehead_grads(module, group=None, prefixes=PREHEAD_PREFIXES):
    """Camera-sharded TRAINING (``VFANet.forward(..., distributed=True)``): call after ``loss.backward()`` and before
    ``optimizer.step()``.

    Every rank back-propagates the same loss through the same fused map, so ``_AllReduceSum.backward`` hands each rank
    the full dL/d(map) and the heads' gradients are already identical everywhere.  The parameters in FRONT of the
    all-reduce (backbone ``base``, laterals ``lat*`` / ``bn*``, projectors ``vfa*``) were only exercised by the local
    cameras, so their gradients are partial sums: this SUMs them over ranks (not a mean -- DistributedDataParallel's
    averaging would scale them by 1/world against the heads; a rank with no camera contributes zeros).  Without it
    the replicas diverge.  Returns the number of tensors reduced."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return 0
    params = [p for n, p in module.named_parameters() if p.requires_grad and n.startswith(tuple(prefixes))]
    for p in params:
        if p.grad is None:  # e.g. the rank of an 8-GPU job that holds none of the 7 cameras
            p.grad = torch.zeros_like(p)
    if not params:
        return 0
    flat = torch.cat([p.grad.reshape(-1) for p in params])  # one bucket: a single collective over xGMI
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return len(params)


def all_reduce_ortho(ortho_nhwc, group=None):
    """Sum the partial BEV maps of all ranks in place (RCCL over xGMI; backend string "nccl" on ROCm)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if ortho_nhwc.requires_grad:
            return _AllReduceSum.apply(ortho_nhwc, group)
        dist.all_reduce(ortho_nhwc, op=dist.ReduceOp.SUM, group=group)
    return ortho_nhwc


def reduce_ortho(ortho_nhwc, dst=0, group=None):
    """Sum the partial BEV maps onto rank ``dst`` OF THE GROUP only (the BEV heads then run on one rank): half the traffic of an
    all-reduce.  Other ranks get their buffer back with unspecified contents.  Returns the tensor.  (``dst`` is a group rank:
    ``torch.distributed.reduce`` wants the global one, and a sub-group need not contain global rank ``dst``.)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(ortho_nhwc, dst=dist.get_global_rank(group, dst) if group is not None else dst, op=dist.ReduceOp.SUM, group=group)
    return ortho_nhwc


# Rows of the fused map a BEV row of the heads' outputs depends on, on either side (reference vfa/model/vfanet.py:44-52): `fuse`
# is a 3x3 convolution (1 row) followed by a 3x3 with dilation 2 (2 rows); `map_classifier` / `orient_pred` add a 3x3 with
# dilation 4 (4 rows): 1 + 2 + 4 = 7; `tytx_pred` / `thtwtl_pred` add two plain 3x3: 1 + 2 + 1 + 1 = 5.
HEAD_HALO_ROWS = 7


def row_bands(length, world):
    """BEV rows of every rank for the scattered map: ``world`` bands of equal height (the last ones may be shorter / empty)."""
    per = (length + world - 1) // world
    return [(min(length, r * per), min(length, (r + 1) * per)) for r in range(world)], per


def reduce_scatter_ortho(ortho_nhwc, length, width, halo=0, group=None):
    """Sum the partial maps and leave every rank with ONE band of BEV rows (+ ``halo`` rows of its neighbours on either side):
    the BEV heads are convolutions with a receptive field of a few rows, so each rank can run their CONVOLUTIONS on its band and
    only the small head outputs are gathered -- a reduce-scatter moves (p - 1) / p of the map once instead of the all-reduce's
    twice.  What a band-local head needs: ``HEAD_HALO_ROWS`` = 7 rows (`fuse`: 1 + 2, then the dilation-4 convolution of the heatmap
    / orientation heads: reference vfa/model/vfanet.py:44-52); with fewer the outputs within (7 - halo) rows of a band boundary are
    wrong.  NOT band-local at any halo: the GroupNorm layers of `tytx_pred` / `thtwtl_pred` and a train-mode BatchNorm in `fuse`
    take their statistics over the WHOLE map -- run those heads on the gathered `fuse` output (or all-reduce their statistics).

    ortho_nhwc (L*W, C) partial map of this rank.  Returns ``(band, (row0, row1), (top, bottom))``: band ((row1 - row0 + top +
    bottom) * W, C) holds rows [row0 - top, row1 + bottom) of the fused map; top / bottom <= halo are the halo rows that exist
    (none beyond the map's edges).  Without a process group the whole map is the band."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return ortho_nhwc, (0, length), (0, 0)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bands, per = row_bands(length, world)
    c = ortho_nhwc.shape[1]
    padded = ortho_nhwc
    if per * world != length:  # reduce_scatter_tensor wants equal shares: pad the map with zero rows
        padded = ortho_nhwc.new_zeros((per * world * width, c))
        padded[:length * width] = ortho_nhwc
    mine = ortho_nhwc.new_empty((per * width, c))
    if dist.get_backend(group) == "gloo":  # (CPU tests: gloo has no reduce-scatter; same result through an all-reduce)
        full = padded.clone()
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        mine.copy_(full[rank * per * width:(rank + 1) * per * width])
    else:
        dist.reduce_scatter_tensor(mine, padded, op=dist.ReduceOp.SUM, group=group)
    r0, r1 = bands[rank]
    mine = mine[:(r1 - r0) * width]
    if halo <= 0:
        return mine, (r0, r1), (0, 0)
    # halo exchange: every rank publishes its first and last `halo` rows (a few hundred KB), neighbours pick theirs.
    # (bands shorter than the halo would need rows from two ranks away: refuse instead of returning a wrong halo)
    assert all(b1 - b0 >= halo or b1 == b0 for b0, b1 in bands), "reduce_scatter_ortho: bands shorter than the halo"
    edge = ortho_nhwc.new_zeros((2, halo * width, c))
    h = min(halo, r1 - r0)
    if h > 0:
        edge[0, :h * width] = mine[:h * width]
        edge[1, (halo - h) * width:] = mine[(r1 - r0 - h) * width:]
    edges = [torch.empty_like(edge) for _ in range(world)]
    dist.all_gather(edges, edge, group=group)
    top = halo if (rank > 0 and r0 > 0 and r1 > r0) else 0
    nxt = next((q for q in range(rank + 1, world) if bands[q][1] > bands[q][0]), None)
    bottom = halo if (nxt is not None and r1 > r0) else 0
    prv = next((q for q in range(rank - 1, -1, -1) if bands[q][1] > bands[q][0]), None)
    parts = []
    if top:
        parts.append(edges[prv][1])
    parts.append(mine)
    if bottom:
        parts.append(edges[nxt][0])
    band = torch.cat(parts) if len(parts) > 1 else mine
    return band, (r0, r1), (top, bottom)


class PendingOrtho:
    """A fused BEV map whose all-reduce is still in flight (``aggregate_views(..., distributed="async")``).

    The collective runs on RCCL's own stream while this process goes on with the next frame; ``wait()`` makes the
    current stream wait for it and returns the (1,C,L,W) map.  Keep at most one or two pending (each holds a map).
    """

    def __init__(self, ortho_nhwc, work, shape):
        self._ortho, self._work, self._shape = ortho_nhwc, work, shape

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        length, width, c = self._shape
        return self._ortho.view(1, length, width, c).permute(0, 3, 1, 2)


def aggregate_views(vfa8, vfa16, vfa32, lat8, lat16, lat32, calibs, grid, crange=(-1, 0.95), reduce_group=None,
                    distributed=False, integrals=None, halo=None):
    """The camera loop of ``VFANet.forward`` for the cameras held by this process.

    lat* (n,C,h,w) lateral maps of the local cameras, calibs (n,3,4), grid (1,L,W,3)
    -> ortho (1,C,L,W): a permuted view of the channels-last buffer, like the reference returns.
    With ``distributed=True`` the partial sums of all ranks are all-reduced before returning; with
    ``distributed="async"`` (inference) the all-reduce is only launched and a ``PendingOrtho`` is returned, so that
    the collective of frame i overlaps the projection of frame i+1 (``"async_reduce"``: the same with a reduce onto rank 0).  ``distributed="reduce"``: the fused map lands on rank 0
    only; ``distributed="reduce_scatter"``: every rank gets its band of BEV rows plus ``halo`` rows of its neighbours (default
    ``HEAD_HALO_ROWS`` = 7: what `fuse` + the dilation-4 heads read; see ``reduce_scatter_ortho`` for what is NOT band-local) and the
    call returns ``(band (1,C,rows,W), (row0, row1), (top, bottom))``.
    ``integrals``: the three integral-image batches instead of the lateral maps (producer fusion, inference on the fused frame
    path only; ``lat*`` may then be None).
    """
    length, width = grid.shape[-3], grid.shape[-2]
    n = calibs.shape[0]
    # a per-call flag of the MFMA entry points (include/vfa_hip.h: VFA_FLAG_RESERVED_CUS), no library state
    reserved = RESERVED_CUS if (bool(distributed) and dist.is_available() and dist.is_initialized()
                                and dist.get_world_size(reduce_group) > 1) else 0
    work = ((vfa8, lat8), (vfa16, lat16), (vfa32, lat32))
    mods3 = [vfa8, vfa16, vfa32]
    if integrals is not None:
        assert n > 0 and not torch.is_grad_enabled() and (vfa_op.pipe_frame_ok(mods3, n) or vfa_op.fused_frame_ok(mods3, n)), \
            "integral-image inputs need a per-frame inference path"
        ortho = torch.empty((length * width, vfa8.collapse.out_features), dtype=torch.float32, device=grid.device)
        frame = vfa_op.pipe_frame if vfa_op.pipe_frame_ok(mods3, n) else vfa_op.fused_frame
        frame(mods3, None, calibs, grid, crange, out=ortho, reserved_cus=reserved, integrals=integrals)  # (an ops.IntegralImages keeps its feature statistics)
    elif n > 0 and vfa_op.fused_train_ok(mods3, n, (lat8, lat16, lat32)):
        # training: the fused kernel in the forward, voxel features and pre-activations recomputed scale by scale in the backward
        ortho = vfa_op.fused_frame_train(mods3, [lat8, lat16, lat32], calibs, grid, crange, reserved_cus=reserved)
    elif n > 0 and vfa_op.pipe_frame_ok(mods3, n, (lat8, lat16, lat32)):
        # inference, any number of z-layers: geometry once per frame + ONE persistent kernel (pooling waves beside matrix waves)
        ortho = torch.empty((length * width, vfa8.collapse.out_features), dtype=torch.float32, device=grid.device)
        vfa_op.pipe_frame(mods3, [lat8, lat16, lat32], calibs, grid, crange, out=ortho, reserved_cus=reserved)
    elif n > 0 and all(m.mfma_collapse_ok(lat) for m, lat in work):
        # inference on single-layer grids: per scale, pooling then ONE MFMA kernel that forms collapse + bias + ReLU and
        # sums the views into the map (sum over views per scale, then over scales: the reference's sums re-associated,
        # inside the post-GEMM tolerance)
        ortho = torch.empty((length * width, vfa8.collapse.out_features), dtype=torch.float32, device=grid.device)
        if vfa_op.fused_frame_ok([vfa8, vfa16, vfa32], n):
            # geometry once per frame + one persistent kernel for pooling, collapse, ReLU and every sum: vox stays on chip
            vfa_op.fused_frame([vfa8, vfa16, vfa32], [lat8, lat16, lat32], calibs, grid, crange, out=ortho,
                               reserved_cus=reserved)
        elif (vfa_op.fused_frame_ok([vfa8, vfa16, vfa32], n, "window")
              and n * length * width * 1024 <= vfa_op.VOX_BYTES_LIMIT):
            # geometry once per frame; per scale: LDS-window pooling kernel (vox in HBM, bit-exact) + MFMA collapse kernel
            vfa_op.window_frame([vfa8, vfa16, vfa32], [lat8, lat16, lat32], calibs, grid, crange, out=ortho,
                                reserved_cus=reserved)
        else:
            for i, (m, lat) in enumerate(work):
                m.project_sum(lat, calibs, grid, crange, out=ortho, accumulate=i > 0, reserved_cus=reserved)
    elif n > 0:
        if N_STREAMS == 1 or not grid.is_cuda:
            lins = [m.project_views(lat, calibs, grid, crange, reserved_cus=reserved) for m, lat in work]
        else:
            main = _lib.current_stream(grid.device)
            side = _streams(grid.device)
            lins = []
            for i, (m, lat) in enumerate(work):
                st = main if i % N_STREAMS == 0 else side[i % N_STREAMS - 1]
                if st is not main:
                    st.wait_stream(main)  # the inputs (and last step's consumers of recycled memory) are ready
                with torch.cuda.stream(st):
                    lins.append(m.project_views(lat, calibs, grid, crange, reserved_cus=reserved))
                if st is not main:
                    lins[-1].record_stream(main)
            for st in side:
                main.wait_stream(st)
        lin8, lin16, lin32 = lins
        ortho = _ScaleViewSum.apply(lin8, lin16, lin32, vfa8.collapse.bias, vfa16.collapse.bias, vfa32.collapse.bias)
    else:  # a rank without cameras (8 GPUs, 7 cameras) contributes zeros
        ortho = torch.zeros((length * width, vfa8.collapse.out_features), dtype=torch.float32, device=grid.device)
    c_out = vfa8.collapse.out_features
    if distributed in ("async", "async_reduce"):
        # ("async_reduce": the sum lands on rank 0 only -- the rank that runs the heads --, half the traffic of the all-reduce; the
        # other ranks' maps are unspecified after the wait)
        work = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(reduce_group) > 1:
            if distributed == "async":
                work = dist.all_reduce(ortho, op=dist.ReduceOp.SUM, group=reduce_group, async_op=True)
            else:
                dst = dist.get_global_rank(reduce_group, 0) if reduce_group is not None else 0
                work = dist.reduce(ortho, dst=dst, op=dist.ReduceOp.SUM, group=reduce_group, async_op=True)
        return PendingOrtho(ortho, work, (length, width, c_out))
    if distributed == "reduce":  # the fused map on rank 0 only (the caller runs the heads there)
        ortho = reduce_ortho(ortho, 0, reduce_group)
    elif distributed == "reduce_scatter":  # this rank's band of BEV rows + the heads' halo: (band (1,C,rows,W), rows, halo)
        band, rows, halo = reduce_scatter_ortho(ortho, length, width, halo=HEAD_HALO_ROWS if halo is None else int(halo), group=reduce_group)
        n_rows = rows[1] - rows[0] + halo[0] + halo[1]
        return band.view(1, n_rows, width, c_out).permute(0, 3, 1, 2), rows, halo
    elif distributed:
        ortho = all_reduce_ortho(ortho, reduce_group)
    return ortho.view(1, length, width, c_out).permute(0, 3, 1, 2)
