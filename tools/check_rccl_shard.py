#!/usr/bin/env python3
"""Run under torchrun with >= 2 GPUs: the camera-sharded aggregate fused by an RCCL all-reduce must equal the
single-process aggregate of all cameras (MultiviewX rig, 6 cameras: BASELINE.json configs[3], on a crop of the grid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    dist.init_process_group("nccl", device_id=dev)
    import vfa_amd
    from vfa_amd.synthetic import make_workload
    wl = make_workload("multiviewx_160x250x8", channels=256, seed=3)
    grid = wl["grid"][:, 40:104].contiguous().to(dev)
    torch.manual_seed(0)
    mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
            for _ in range(3)]
    n = wl["n_cam"]
    lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
    calibs = wl["calibs"].to(dev)
    with torch.no_grad():
        full = vfa_amd.aggregate_views(*mods, *lats, calibs, grid)
        mine = torch.tensor(vfa_amd.camera_shard(n), dtype=torch.long, device=dev)
        part = vfa_amd.aggregate_views(*mods, *(l[mine] for l in lats), calibs[mine], grid, distributed=True)
        pend = vfa_amd.aggregate_views(*mods, *(l[mine] for l in lats), calibs[mine], grid, distributed="async").wait()
        red = vfa_amd.aggregate_views(*mods, *(l[mine] for l in lats), calibs[mine], grid, distributed="reduce")
        band, (r0, r1), (top, bottom) = vfa_amd.aggregate_views(*mods, *(l[mine] for l in lats), calibs[mine], grid,
                                                                 distributed="reduce_scatter")
    torch.cuda.synchronize()
    scale = full.abs().max().item()
    for name, got in (("sync", part), ("async", pend)):
        err = (got - full).abs().max().item()
        assert err <= 2e-5 * scale, (name, err, scale)
    if rank == 0:  # reduce: the fused map on rank 0
        assert (red - full).abs().max().item() <= 2e-5 * scale
    # reduce_scatter: this rank's band of BEV rows + the halo of the heads (HEAD_HALO_ROWS = 7)
    want = full[:, :, r0 - top:r1 + bottom]
    assert band.shape == want.shape, (band.shape, want.shape)
    assert (band - want).abs().max().item() <= 2e-5 * scale
    dist.barrier()
    if rank == 0:
        print(f"rccl shard ok: world {world}, max |ortho| {scale:.3f}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
