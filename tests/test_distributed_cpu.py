"""Camera-sharded aggregation over ranks (the N>1 path), on CPU with gloo, world_size 2.

The HIP kernels cannot run here, so each rank forms the partial BEV map of ITS cameras with the CPU oracle
(test infrastructure) and the product's own sharding + all-reduce code (vfa_amd.aggregate) fuses them; the sum
must match the reference's full VFANet fixture.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, golden_path


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vfa_oracle as oracle
        from vfa_amd.aggregate import all_reduce_ortho, camera_shard
        d = np.load(golden_path(case))
        n_cam = d["calibs"].shape[0]
        mine = camera_shard(n_cam)  # rank / world from the process group
        assert mine == list(range(rank, n_cam, world))
        meta = dict(data=str(d["data"]), image_size=tuple(int(v) for v in d["image_size"]),
                    cube_size=tuple(float(v) for v in d["cube_size"]), grid_height=float(d["grid_height"]))
        C, (L, W) = d["ortho"].shape[0], d["grid"].shape[:2]
        if mine:
            part = oracle.vfanet_aggregate({s: d[f"lat{s}"][mine] for s in (8, 16, 32)}, d["calibs"][mine], d["grid"],
                                           {s: d[f"weight{s}"] for s in (8, 16, 32)},
                                           {s: d[f"bias{s}"] for s in (8, 16, 32)}, **meta)
        else:
            part = np.zeros((C, L, W), np.float32)
        # the product keeps the map channels-last (L*W, C); reduce it in that layout
        t = torch.from_numpy(np.ascontiguousarray(part.reshape(C, L * W).T))
        t = all_reduce_ortho(t)
        got = t.numpy().T.reshape(C, L, W)
        ref = d["ortho"]
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["vfanet_mc.npz", "vfanet_wt.npz"])
def test_camera_sharded_sum_matches_reference_world2(case, tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_camera_shard_partitions_every_rig():
    from vfa_amd.aggregate import camera_shard
    for n_cam in (1, 6, 7, 8, 13):
        for world in (1, 2, 4, 6, 8):
            shards = [camera_shard(n_cam, r, world) for r in range(world)]
            assert sorted(c for s in shards for c in s) == list(range(n_cam))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert camera_shard(7, 7, 8) == []  # 7 cameras on 8 GPUs: the last rank contributes zeros


def test_all_reduce_is_identity_without_a_process_group():
    from vfa_amd.aggregate import all_reduce_ortho
    t = torch.arange(6.).view(3, 2)
    assert all_reduce_ortho(t) is t


def _async_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vfa_amd.aggregate import PendingOrtho
        # what aggregate_views(distributed="async") does after forming the partial map, on CPU tensors
        maps = []
        for frame in range(3):
            part = torch.full((6, 4), float(rank + 1 + 10 * frame))
            work = dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True)
            maps.append(PendingOrtho(part, work, (2, 3, 4)))
        for frame, p in enumerate(maps):
            got = p.wait()
            assert tuple(got.shape) == (1, 4, 2, 3)
            assert torch.all(got == sum(r + 1 + 10 * frame for r in range(world)))
            assert p.wait() is not None  # idempotent
        # distributed="async_reduce": the sum lands on rank 0 only (what bench.py --gpus N uses by default)
        maps = []
        for frame in range(3):
            part = torch.full((6, 4), float(rank + 1 + 10 * frame))
            work = dist.reduce(part, dst=0, op=dist.ReduceOp.SUM, async_op=True)
            maps.append(PendingOrtho(part, work, (2, 3, 4)))
        for frame, p in enumerate(maps):
            got = p.wait()
            if rank == 0:
                assert torch.all(got == sum(r + 1 + 10 * frame for r in range(world)))
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_pending_ortho_async_all_reduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_async_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _collective_worker(rank, world, port, out_dir, length, width):
    """reduce -> rank 0 and reduce_scatter over BEV rows (+ halo) against the plain sum of the parts."""
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vfa_amd.aggregate import reduce_ortho, reduce_scatter_ortho, row_bands
        C = 8
        gen = torch.Generator().manual_seed(5)
        parts = [torch.rand(length * width, C, generator=gen) for _ in range(world)]  # (every rank builds all parts: the expected sum)
        want = torch.stack(parts).sum(0)
        got = reduce_ortho(parts[rank].clone(), dst=0)
        if rank == 0:
            torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
        for halo in (0, 4):
            band, (r0, r1), (top, bottom) = reduce_scatter_ortho(parts[rank].clone(), length, width, halo=halo)
            bands, per = row_bands(length, world)
            assert (r0, r1) == bands[rank]
            assert top == (halo if rank > 0 and r1 > r0 else 0) and bottom == (halo if r1 < length and r1 > r0 else 0)
            ref = want.view(length, width, C)[r0 - top:r1 + bottom].reshape(-1, C)
            assert band.shape == ref.shape, (band.shape, ref.shape)
            torch.testing.assert_close(band, ref, rtol=1e-6, atol=1e-6)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,length,width", [(2, 21, 6), (3, 16, 5)])
def test_reduce_and_reduce_scatter_with_halo(world, length, width, tmp_path):
    """The two alternatives to the all-reduce SURVEY section 8e names: `reduce` to the rank that runs the heads, and
    `reduce_scatter` over bands of BEV rows with a halo (4 rows here; what the heads need: test_band_local_heads_need_the_seven_row_halo)
    (reference vfa/model/vfanet.py:48, :52); ragged row counts (zero-padded shares)."""
    port = _free_port()
    mp.spawn(_collective_worker, args=(world, port, str(tmp_path), length, width), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def test_band_local_heads_need_the_seven_row_halo():
    """What `reduce_scatter` is for: each rank runs the CONVOLUTIONS of `fuse` + the dilation-4 heatmap head on its band of the
    fused map (reference vfa/model/vfanet.py:44-48: 3x3, 3x3 dilation 2, 3x3 dilation 4 -> 1 + 2 + 4 rows).  With
    ``HEAD_HALO_ROWS`` = 7 halo rows the band-local outputs equal the full-map ones on every row of the band; with the 4 rows
    the docs used to promise, rows near a band boundary differ."""
    import torch.nn as nn
    from vfa_amd.aggregate import HEAD_HALO_ROWS, row_bands
    assert HEAD_HALO_ROWS == 7
    torch.manual_seed(0)
    C, L, W, world = 16, 40, 12, 3
    fuse = nn.Sequential(nn.Conv2d(C, C, 3, padding=1), nn.BatchNorm2d(C), nn.ReLU(True),
                         nn.Conv2d(C, C, 3, padding=2, dilation=2), nn.BatchNorm2d(C), nn.ReLU(True)).eval()
    head = nn.Sequential(nn.Conv2d(C, 1, 3, padding=4, dilation=4, bias=False)).eval()
    for m in fuse:
        if isinstance(m, nn.BatchNorm2d):  # (eval mode: running statistics, band-independent)
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    ortho = torch.rand(1, C, L, W)
    with torch.no_grad():
        full = head(fuse(ortho))
        bands, _ = row_bands(L, world)
        for halo, exact in ((HEAD_HALO_ROWS, True), (4, False)):
            worst = 0.0
            for r0, r1 in bands:
                top, bottom = min(halo, r0), min(halo, L - r1)
                band = ortho[:, :, r0 - top:r1 + bottom]            # what reduce_scatter_ortho hands this rank
                local = head(fuse(band))[:, :, top:top + (r1 - r0)]  # (zero padding at the band's outer edge: only the map's own edges are real)
                worst = max(worst, (local - full[:, :, r0:r1]).abs().max().item())
            if exact:
                assert worst <= 1e-6, worst
            else:
                assert worst > 1e-3, worst  # the old 4-row halo is NOT enough


def _subgroup_reduce_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vfa_amd.aggregate import reduce_ortho
        group = dist.new_group([1, 2])  # (does not contain global rank 0)
        if rank in (1, 2):
            part = torch.full((6, 4), float(rank))
            got = reduce_ortho(part, dst=0, group=group)  # group rank 0 = global rank 1
            if rank == 1:
                assert torch.all(got == 3.0)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_reduce_onto_rank_zero_of_a_subgroup(tmp_path):
    """`reduce_ortho(dst=0, group=g)` means rank 0 OF THE GROUP, also when the group does not contain global rank 0
    (``distributed="reduce"`` of ``aggregate_views`` goes through it)."""
    port = _free_port()
    mp.spawn(_subgroup_reduce_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1", "ok2"]
