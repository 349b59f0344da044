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

from . import ops, vfa_op

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
                    distributed=False, integrals=None):
    """The camera loop of ``VFANet.forward`` for the cameras held by this process.

    lat* (n,C,h,w) lateral maps of the local cameras, calibs (n,3,4), grid (1,L,W,3)
    -> ortho (1,C,L,W): a permuted view of the channels-last buffer, like the reference returns.
    With ``distributed=True`` the partial sums of all ranks are all-reduced before returning; with
    ``distributed="async"`` (inference) the all-reduce is only launched and a ``PendingOrtho`` is returned, so that
    the collective of frame i overlaps the projection of frame i+1.
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
        frame(mods3, None, calibs, grid, crange, out=ortho, reserved_cus=reserved, integrals=list(integrals))
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
            main = torch.cuda.current_stream(grid.device)
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
    if distributed == "async":
        work = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(reduce_group) > 1:
            work = dist.all_reduce(ortho, op=dist.ReduceOp.SUM, group=reduce_group, async_op=True)
        return PendingOrtho(ortho, work, (length, width, c_out))
    if distributed:
        ortho = all_reduce_ortho(ortho, reduce_group)
    return ortho.view(1, length, width, c_out).permute(0, 3, 1, 2)
