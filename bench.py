#!/usr/bin/env python3
"""bench.py -- throughput of the multiview feature -> voxel projection + aggregation path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic frame batch already resident in HBM: for each of the
three feature scales the integral images of all local cameras, the fused projection + box-pooling kernel and the
collapse (Linear + bias + ReLU + view sum: one hand-written MFMA kernel on single-layer grids, MFMA tile GEMM + epilogue
kernel on multi-layer ones); with N > 1 ranks an RCCL all-reduce of the partial BEV map.  Unit of work ("voxel aggregated") = one (camera, scale, z-layer, BEV cell) box producing
C = 256 channels.  Scaling is weak: every rank holds `n_cam` cameras of an N-times larger
rig observing the same grid, and the grids are summed over ranks (camera-sharded data parallelism).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
FP32_MFMA_PEAK_TFLOPS = 157.3


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="multiviewc_200x200x1",
                   help="named workload of vfa_amd.synthetic.WORKLOADS (default: BASELINE.json configs[1])")
    p.add_argument("--channels", type=int, default=256)
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="weak: n_cam cameras per rank; strong: the frame's cameras are split over ranks")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 disables)")
    p.add_argument("--tune-gemm", type=int, default=0,
                   help="1: TunableOp selects the library GEMM in warm-up (only used with VFA_AMD_COLLAPSE=library)")
    return p.parse_args()


def cpu_baseline(wl, budget_s):
    """The CPU oracle (a port of the reference's arithmetic, oracle/vfa_oracle.c) timed on this box's host cores on a
    bounded sample of the same workload.  Baseline only -- never the thing shipped or measured as `value`."""
    import numpy as np
    from oracle import vfa_oracle as oracle
    oracle.build()
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    C = wl["channels"]
    nl = len(oracle.z_layers_of(wl["grid_height"], wl["cube_size"]))
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((C, C * nl)) * 0.02).astype(np.float32)
    b = np.zeros(C, np.float32)
    grid = wl["grid"][0].cpu().numpy()
    units, t0, cams = 0, time.perf_counter(), 0
    for cam in range(wl["n_cam"]):
        for s in range(3):
            f = wl["features"][cam][s][0].cpu().numpy()
            oracle.vfa_forward(f, wl["calibs"][cam].cpu().numpy(), grid, w, b, wl["args"].data, wl["args"].image_size,
                               wl["cube_size"], wl["grid_height"])
            units += nl * grid.shape[0] * grid.shape[1]
        cams += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": units / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": f"{cams} of {wl['n_cam']} cameras x 3 scales of the same frame ({units} box-units, {dt:.1f} s), "
                      f"oracle/vfa_oracle.c with OpenMP on {cores} host threads"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # functional test hook for 1-GPU boxes: all ranks on device 0 with gloo (VFA_BENCH_BACKEND=gloo); never used for numbers
    backend = os.environ.get("VFA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    import vfa_amd
    from vfa_amd import ops
    from vfa_amd.synthetic import make_workload

    torch.backends.cuda.matmul.allow_tf32 = False
    if a.tune_gemm:
        # let PyTorch's TunableOp pick the rocBLAS / hipBLASLt solution for the three collapse products during warm-up
        # (fp32 in, fp32 accumulate either way; +6 % step throughput on MI355X).  Results stay in this process only.
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(200)
        torch.cuda.tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"vfa_tunableop_{rank}.csv"))
    wl = make_workload(a.workload, channels=a.channels, seed=rank)
    n_frame = wl["n_cam"]
    cams = list(range(n_frame)) if a.scaling == "weak" else vfa_amd.camera_shard(n_frame, rank, world)
    n = len(cams)
    idx = torch.tensor(cams, dtype=torch.long)
    lats = [torch.cat([wl["features"][c][s] for c in cams]).to(dev) if n else
            torch.zeros((0, a.channels) + tuple(wl["feat_sizes"][s]), device=dev) for s in range(3)]
    calibs = wl["calibs"][idx].to(dev)
    grid = wl["grid"].to(dev)
    torch.manual_seed(0)
    mods = [vfa_amd.VFA(a.channels, grid_height=wl["grid_height"], cube_size=wl["cube_size"], feat_scale=1 / 8.,
                        args=wl["args"]).to(dev) for _ in range(3)]
    L, W = grid.shape[1:3]
    nl = mods[0].num_grid_layer
    C = a.channels
    units_rank = n * 3 * nl * L * W
    units_total = units_rank * world if a.scaling == "weak" else n_frame * 3 * nl * L * W

    # N > 1: the all-reduce of frame i is launched asynchronously and overlaps the projection of frame i+1 (one map in
    # flight); every collective completes inside the timed region.  VFA_BENCH_SYNC_REDUCE=1 reduces synchronously.
    overlap = world > 1 and os.environ.get("VFA_BENCH_SYNC_REDUCE", "0") != "1"
    pending = []

    def step():
        with torch.no_grad():
            if not overlap:
                return vfa_amd.aggregate_views(*mods, *lats, calibs, grid, distributed=world > 1)
            while pending:
                pending.pop().wait()  # frame i-1 is fused before frame i's collective is queued
            pending.append(vfa_amd.aggregate_views(*mods, *lats, calibs, grid, distributed="async"))

    def drain():
        while pending:
            pending.pop().wait()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()  # one-off set-up outside warm-up and timing: GEMM kernel selection (TunableOp) and pooling-kernel choice
    drain()
    # Warm-up steps: HIP events around EVERY entry point (the `kernels` table).  Timed steps: only around the roofline
    # kernel -- each timed launch puts two event records in the queue (~3 us apiece), and timing all nine launches
    # of a frame slowed the 0.9 ms frame by 6 %.
    with ops.KernelTimer() as kt_warm:
        for _ in range(a.warmup):
            step()
        drain()
        fence()
    with ops.KernelTimer(only=("vfa_project_gather_f32",)) as kt:
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        drain()
        fence()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    ks = kt.summary()
    # roofline of the dominant hand-written kernel: fused projection + box pooling (HBM-bound).
    # algorithmic bytes of one launch = integral images read once + voxel features written once + grid + calibs
    g = ks.get("vfa_project_gather_f32", dict(launches=0, ms=0.0, by_tag={}))
    alg_bytes = 0
    for (nv, Ct, Hf, Wf, nlt, cells), rec in g["by_tag"].items():
        alg_bytes += rec["launches"] * (nv * Ct * Hf * Wf * 4 + nv * nlt * cells * Ct * 4 + cells * 12 + nv * 48)
    # HBM traffic of that kernel from the PMC counters: collected by separate `rocprofv3 --pmc FETCH_SIZE` /
    # `--pmc WRITE_SIZE` passes over this same command (tools/pmc_to_traffic.py, summary committed under profiles/)
    traffic = None
    chosen = sorted(set(ops._gather_choice.values())) or ["default"]
    kname = {"tap_cache": "gather_cached_kernel", "direct": "gather_kernel<4, true>"}
    tpath = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
    if a.workload == "multiviewc_200x200x1" and os.path.exists(tpath) and len(chosen) == 1 and chosen[0] in kname:
        for name, rec in json.load(open(tpath))["kernels"].items():
            if kname[chosen[0]] in name:
                traffic = rec["hbm_bytes_per_dispatch"]
    roofline = None
    if g["launches"]:
        avg_ms = g["ms"] / g["launches"]
        achieved = alg_bytes / g["launches"] / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "vfa_project_gather_f32: " + "/".join(kname.get(c, c) for c in chosen) +
                    " (picked per shape on first use)", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": alg_bytes / g["launches"],
                    "launches": g["launches"]}
    ks_all = kt_warm.summary() if a.warmup > 0 else ks
    kernels = {k: {"launches": v["launches"], "avg_us": 1e3 * v["ms"] / max(v["launches"], 1)} for k, v in ks_all.items()}
    kernels_note = f"HIP events around every entry point during the {a.warmup} warm-up steps; the timed steps time only the roofline kernel"
    gemm_flops = 3 * 2.0 * n * L * W * (C * nl) * C
    hip_ms = sum(v["ms"] for v in ks_all.values()) / max(a.warmup, 1)

    ck = ks_all.get("vfa_collapse_relu_sum_f32")
    if ck and ck["launches"]:
        # hand-written kernel: every fp32 product = 3 bf16 MFMA products of an exact hi/lo split, fp32 accumulation,
        # fused with bias + ReLU + view sum (inference, K = N = 256).  TFLOP/s below counts the fp32-equivalent flops.
        collapse_info = {"flops_per_step": gemm_flops, "backend": "vfa_collapse_relu_sum_f32 (3xbf16-split MFMA, fp32 "
                         "accumulate, fused bias+ReLU+view sum)", "avg_us": 1e3 * ck["ms"] / ck["launches"],
                         "fp32_equivalent_tflops": gemm_flops / 3 / (ck["ms"] / ck["launches"] * 1e-3) / 1e12,
                         "bf16_mfma_tflops": gemm_flops / (ck["ms"] / ck["launches"] * 1e-3) / 1e12,
                         "max_rel_error_vs_fp64": "~3e-6 of max|out| (tests/test_hip_parity.py), tolerance 1e-5"}
    elif ks_all.get("vfa_collapse_gemm_f32", {}).get("launches"):
        cg = ks_all["vfa_collapse_gemm_f32"]
        collapse_info = {"flops_per_step": gemm_flops, "backend": "vfa_collapse_gemm_f32 (K-looped 3xbf16-split MFMA tile GEMM, "
                         "fp32 accumulate) + epilogue kernels", "avg_us": 1e3 * cg["ms"] / cg["launches"],
                         "fp32_equivalent_tflops": gemm_flops / 3 / (cg["ms"] / cg["launches"] * 1e-3) / 1e12,
                         "max_rel_error_vs_fp64": "~5e-6 of max|out| (tests/test_hip_parity.py), tolerance 1e-5"}
    else:
        collapse_info = {"flops_per_step": gemm_flops, "backend": "torch.matmul (rocBLAS/hipBLASLt fp32)"
                         + (", TunableOp-selected" if a.tune_gemm else ""), "peak_tflops": FP32_MFMA_PEAK_TFLOPS}

    out = None
    if rank == 0:
        out = {
            "metric": "voxels aggregated/sec (7 views->BEV grid)", "value": units_total * a.steps / dt,
            "unit": "voxels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 in / out and fp32 accumulation everywhere; pre-GEMM stages bit-exact with the reference's CPU path; the "
                          "collapse product forms each fp32 product from three bf16 MFMA products of an exact hi/lo split "
                          "(error ~3e-6 of max|out|, tolerance 1e-5; VFA_AMD_COLLAPSE=library selects the fp32 library GEMM)",
            "config": {"workload": a.workload, "cameras_per_rank": n, "cameras_total": n * world if a.scaling == "weak"
                       else n_frame, "channels": C, "feature_maps": [list(s) for s in wl["feat_sizes"]],
                       "grid": [L, W, nl], "units_per_step": units_total,
                       "parallelism": (f"camera-sharded dp{world}, RCCL all-reduce of the BEV map"
                                       + (" overlapped with the next frame" if overlap else "")) if world > 1
                       else "single GPU"},
            "bev_cells_per_s": nl * L * W * a.steps / dt,
            "roofline": roofline,
            "kernels": kernels,
            "kernels_note": kernels_note,
            "hip_kernel_ms_per_step": hip_ms,
            "collapse_gemm": collapse_info,
        }
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(wl, a.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
