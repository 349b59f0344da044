#!/usr/bin/env python3
"""bench.py -- throughput of the multiview feature -> voxel projection + aggregation path on MI355X.

    python bench.py                                   # 1 GPU, BASELINE.json configs[1]
    python bench.py --gpus N [--steps K --warmup W]   # N > 1 without torchrun: bench.py spawns the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # or under a launcher (RANK / LOCAL_RANK / WORLD_SIZE in the env)

One step = one pass of the hot path over one synthetic frame already resident in HBM: for each of the three feature
scales the integral images of the local cameras, projection + box pooling and the collapse (Linear + bias + ReLU + view
sum); with N > 1 ranks an RCCL all-reduce of the partial BEV map.  Unit of work ("voxel aggregated") = one (camera,
scale, z-layer, BEV cell) box producing C = 256 channels.

Workload and scaling.  The primary line is `multiviewc_200x200x1` (BASELINE.json configs[1]: 7 cameras 1280x720 -> the
37.5 m x 37.5 m grid) at EVERY N, so the driver's per-N values compare like with like.  N > 1 defaults to STRONG
scaling, the north-star partitioning: the 7 cameras of each frame are sharded over the ranks (one camera per GPU at
N >= 7; a rank without a camera contributes zeros), the partial maps are fused by one RCCL all-reduce over xGMI, frames
are streamed and the all-reduce of frame i overlaps the projection of frame i+1.  The same line also carries
`scaling_curve_c5`: BASELINE.json configs[4] (`synthetic4k_512x512x32`, 8 cameras x 4K, one camera per GPU at N = 8),
measured the same way with fewer steps -- the configuration on which >= 6x at 8 GPUs is attainable.
`--scaling weak` (every rank a full rig, grids summed) is kept as an option.  Side legs at N = 1 (none of them `value`):
`shipped_configs` (the reference's three shipped configs through the pipelined kernel), `training_step` (forward + backward of the
bench frame through the fused autograd node), `per_rank_proxy`, `reference_loop`, `producer_f3`, the other arithmetic forms.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense (MI355X_MICROARCH.md); the 5 PF headline includes 2:1 sparsity
PRIMARY = "multiviewc_200x200x1"
C5 = "synthetic4k_512x512x32"
SHIPPED = ("multiviewc_156x156x5", "wildtrack_120x360x8", "multiviewx_160x250x8")  # reference vfa/config.py:5-28, :60-85, :32-57
# algorithmic bytes per frame (feature maps read once + voxel features written once + the map)
SURVEY_ALG_BYTES = {"multiviewc_156x156x5": 3.03e9, "multiviewc_200x200x1": 1.28e9, "wildtrack_120x360x8": 7.85e9,
                    "wildtrack_480x1440x1": 15.95e9, "multiviewx_160x250x8": 6.26e9, "synthetic4k_512x512x32": 210.4e9}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--condition-steps", type=int, default=150,
                   help="untimed frames queued right in front of the opening barrier + synchronize of a timed region, so that a "
                        "short run does not measure the clock ramp of an idle device (0 disables)")
    p.add_argument("--workload", default=PRIMARY,
                   help="named workload of vfa_amd.synthetic.WORKLOADS (default: BASELINE.json configs[1])")
    p.add_argument("--channels", type=int, default=256)
    p.add_argument("--collective", choices=["reduce", "all_reduce"], default="reduce",
                   help="N > 1: how the partial BEV maps are fused: `reduce` onto rank 0, the rank that runs the heads (north_star: "
                        "'an RCCL reduce over xGMI to form the fused voxel grid'; half the traffic), or `all_reduce`")
    p.add_argument("--scaling", choices=["weak", "strong"], default=None,
                   help="strong (default for N > 1): the frame's cameras are split over ranks; weak: n_cam cameras per rank")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 disables)")
    p.add_argument("--c5-steps", type=int, default=3, help="timed steps of the synthetic4k_512x512x32 leg (0 disables)")
    p.add_argument("--shipped-steps", type=int, default=3,
                   help="timed blocks' steps of the `shipped_configs` leg: the reference's three shipped configs (vfa/config.py:5-85), "
                        "all through pipe_kernel (0 disables)")
    p.add_argument("--train-steps", type=int, default=5, help="timed steps of the training_step leg: forward + backward of the bench frame (0 disables)")
    p.add_argument("--rotate", type=int, default=4, help="input sets of the rotating-input leg (0 disables)")
    p.add_argument("--fp32-steps", type=int, default=20, help="timed steps of the fp32-arithmetic collapse leg (0 disables)")
    p.add_argument("--proxy-steps", type=int, default=20, help="timed steps of the per-rank proxy legs (0 disables)")
    p.add_argument("--loop-steps", type=int, default=10, help="timed frames of the reference_loop leg: the reference's 21-call camera loop (0 disables)")
    p.add_argument("--tune-gemm", type=int, default=0,
                   help="1: TunableOp selects the library GEMM in warm-up (only used with VFA_AMD_COLLAPSE=library)")
    return p.parse_args()


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher.  The parent never touches the GPU (no torch.cuda call, no
# exec of a GPU-initialised process); it starts N fresh children, relays rank 0's JSON line and returns their status.
# ------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, budget_s):
    """the torch-op CPU restatement of the reference path (oracle/torch_reference.py: the reference's
    own op sequence, bit-identical pre-GEMM to the reference's run -- tests/test_oracle_golden.py) timed on this box's
    host cores with every thread and with one thread, on a bounded sample of the same frame.  Baseline only -- never
    the thing shipped or measured as `value`."""
    import torch
    from oracle import torch_reference as tr
    from oracle import vfa_oracle as oracle
    cores = os.cpu_count() or 1
    C = wl["channels"]
    zl = torch.from_numpy(oracle.z_layers_of(wl["grid_height"], wl["cube_size"]))
    co = torch.from_numpy(oracle.corner_offsets(wl["cube_size"]))
    nl = zl.numel()
    gen = torch.Generator().manual_seed(0)
    ws = {s: (torch.rand(C, C * nl, generator=gen) - 0.5) * 0.1 for s in (8, 16, 32)}
    bs = {s: torch.zeros(C) for s in (8, 16, 32)}
    grid = wl["grid"][0].cpu()
    L, W = grid.shape[:2]
    cams_have = [c for c in range(wl["n_cam"]) if wl["features"][c] is not None]
    lats = {s: torch.cat([wl["features"][c][i] if wl["features"][c] is not None else
                          torch.zeros_like(wl["features"][cams_have[0]][i]) for c in range(wl["n_cam"])]).cpu()
            for i, s in enumerate((8, 16, 32))}
    calibs = wl["calibs"].cpu()

    def run(threads, budget):
        torch.set_num_threads(threads)
        done, t0 = 0, time.perf_counter()
        with torch.no_grad():
            for cam in cams_have:
                tr.vfanet_aggregate(lats, calibs, grid, ws, bs, zl, co, wl["args"].data, wl["args"].image_size,
                                    cameras=[cam])
                done += 1
                if time.perf_counter() - t0 > budget:
                    break
        dt = time.perf_counter() - t0
        return done * 3 * nl * L * W / dt, done, dt

    prev = torch.get_num_threads()
    by_threads = {}
    try:
        run(1, 0.0)  # warm the allocator on one camera
        # 1 thread first (cheap and always sane), then wider pools; the last, os.cpu_count(), is bounded to ONE camera:
        # on a 256-thread EPYC torch's intra-op pool is slower than one thread on these tensor sizes (oversubscription)
        plan = [1] + sorted({t for t in (8, 32) if t < cores}) + [cores]
        share = budget_s / (len(plan) + 1)
        for t in plan:
            v, n_done, secs = run(t, 0.0 if (t == cores and cores > 32) else share)
            by_threads[t] = {"value": v, "cameras": n_done, "seconds": round(secs, 2)}
    finally:
        torch.set_num_threads(prev)
    best = max(by_threads, key=lambda t: by_threads[t]["value"])
    return {"value": by_threads[best]["value"], "unit": "voxels/s", "cores": best, "kind": "port", "cpu": cpu_model(),
            "host_threads": cores, "one_thread_value": by_threads[1]["value"],
            "all_threads_value": by_threads[cores]["value"],
            "all_threads_note": ("oversubscribed: torch's intra-op pool at os.cpu_count() threads is slower than one thread on these "
                                 "tensor sizes; reported for completeness, `value` is the best setting") if cores > 32 else None,
            "by_threads": by_threads,
            "sample": f"torch-op restatement of the reference path (oracle/torch_reference.py, torch {torch.__version__} CPU, "
                      f"fp32, no_grad) on cameras x 3 scales of the same frame, at {plan} intra-op threads (cameras done and "
                      f"seconds per setting in by_threads); `value` is the best setting ({best} threads), "
                      f"`all_threads_value` is os.cpu_count() = {cores} threads"}


class Leg:
    """One workload on this rank: inputs in HBM, the three projector modules, a step function and its timing."""

    def __init__(self, name, a, rank, world, dev, scaling, rotate=1):
        import torch
        import torch.distributed as dist
        import vfa_amd
        from vfa_amd.synthetic import WORKLOADS, make_workload
        self.torch, self.dist, self.vfa_amd = torch, dist, vfa_amd
        self.name, self.world, self.rank, self.dev, self.scaling = name, world, rank, dev, scaling
        n_frame = WORKLOADS[name]["n_cam"]
        self.cams = list(range(n_frame)) if scaling == "weak" else vfa_amd.camera_shard(n_frame, rank, world)
        seeds = [rank * 100 + k for k in range(max(rotate, 1))] if scaling == "weak" else list(range(max(rotate, 1)))
        self.sets = []
        wl = None
        for seed in seeds:
            wl = make_workload(name, channels=a.channels, seed=seed, cameras=self.cams)
            self.sets.append([torch.cat([wl["features"][c][s] for c in self.cams]).to(dev) if self.cams else
                              torch.zeros((0, a.channels) + tuple(wl["feat_sizes"][s]), device=dev) for s in range(3)])
        self.wl = wl
        n = len(self.cams)
        self.calibs = wl["calibs"][torch.tensor(self.cams, dtype=torch.long)].to(dev)
        self.grid = wl["grid"].to(dev)
        torch.manual_seed(0)
        self.mods = [vfa_amd.VFA(a.channels, grid_height=wl["grid_height"], cube_size=wl["cube_size"], feat_scale=1 / 8.,
                                 args=wl["args"]).to(dev) for _ in range(3)]
        self.L, self.W = self.grid.shape[1:3]
        self.nl = self.mods[0].num_grid_layer
        self.n_frame = n_frame
        self.units_step = (n * world if scaling == "weak" else n_frame) * 3 * self.nl * self.L * self.W
        # N > 1: the all-reduce of frame i is launched asynchronously and overlaps the projection of frame i+1 (one map
        # in flight); every collective completes inside the timed region.  VFA_BENCH_SYNC_REDUCE=1 reduces synchronously.
        self.overlap = world > 1 and os.environ.get("VFA_BENCH_SYNC_REDUCE", "0") != "1"
        self.pending = []
        self.k = 0
        self.collective = True  # (False: the same steps without the collective: per-rank compute time)
        self.mode = getattr(a, "collective", "reduce")  # "reduce" -> rank 0 or "all_reduce"

    def step(self):
        torch, vfa_amd = self.torch, self.vfa_amd
        lats = self.sets[self.k % len(self.sets)]
        self.k += 1
        with torch.no_grad():
            if not self.collective:
                return vfa_amd.aggregate_views(*self.mods, *lats, self.calibs, self.grid, distributed=False)
            if not self.overlap:
                how = ("reduce" if self.mode == "reduce" else True) if self.world > 1 else False
                return vfa_amd.aggregate_views(*self.mods, *lats, self.calibs, self.grid, distributed=how)
            while self.pending:
                self.pending.pop().wait()  # frame i-1 is fused before frame i's collective is queued
            self.pending.append(vfa_amd.aggregate_views(*self.mods, *lats, self.calibs, self.grid,
                                                        distributed="async_reduce" if self.mode == "reduce" else "async"))

    def drain(self):
        while self.pending:
            self.pending.pop().wait()

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps, timer=None, lead_in=0, min_ms=0.0, max_blocks=40):
        """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks.  `lead_in`: untimed frames
        queued right in front of the opening bracket -- an MI355X that has been idle for a few milliseconds (the host-side
        garbage collection below is enough) runs its next ~50 ms of work ~10 % below its steady clock, and a short timed
        region would measure that ramp instead of the path.  `min_ms`: the bracketed block of `steps` steps is REPEATED (each
        repetition bracketed the same way) until the blocks add up to at least this much; returns the MEDIAN block (a 12 ms
        region is at the mercy of one host hiccup) and leaves all of them in `self.blocks` (seconds; GPU-side times of the same
        blocks from HIP events in `self.blocks_gpu`)."""
        torch = self.torch
        self.drain()
        self.fence()
        ctx = timer if timer is not None else _Null()
        # (no cyclic-GC pause inside the timed region: a generation-2 collection stalls the launching thread for 30-60 ms,
        # longer than the work queued ahead of it -- seen as a 2 x outlier of a 100-step leg about once in six runs)
        gc.collect()
        gc.disable()
        blocks, blocks_gpu = [], []
        try:
            for _ in range(lead_in):
                self.step()
            self.drain()
            self.fence()
            with ctx:
                while True:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0 = time.perf_counter()
                    e0.record()
                    for _ in range(steps):
                        self.step()
                    self.drain()
                    e1.record()
                    self.fence()
                    dt = time.perf_counter() - t0
                    if self.world > 1:
                        t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
                        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                        dt = t.item()
                    blocks.append(dt)
                    blocks_gpu.append(e0.elapsed_time(e1) * 1e-3)
                    if sum(blocks) * 1e3 >= min_ms or len(blocks) >= max_blocks:
                        break
        finally:
            gc.enable()
        self.blocks, self.blocks_gpu = blocks, blocks_gpu
        return sorted(blocks)[len(blocks) // 2]

    def block_stats(self, steps):
        b, g = sorted(self.blocks), sorted(self.blocks_gpu)
        return {"blocks": len(b), "steps_per_block": steps, "timed_ms_total": 1e3 * sum(b),
                "ms_per_step_min": 1e3 * b[0] / steps, "ms_per_step_median": 1e3 * b[len(b) // 2] / steps,
                "ms_per_step_max": 1e3 * b[-1] / steps, "gpu_event_ms_per_step_median": 1e3 * g[len(g) // 2] / steps,
                "note": "every block = exactly `steps` steps between barrier + synchronize brackets (host clock; MAX over ranks); the "
                        "block is repeated until >= 100 ms are timed, `value` / `ms_per_step` are the MEDIAN block; "
                        "gpu_event_*: the same blocks between two HIP events on the launch stream"}


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


# entry point -> the kernel that does the work, for the roofline label
KERNEL_NAMES = {"vfa_project_gather_f32": {"tap_cache": "gather_cached_kernel", "direct": "gather_kernel<4, true>"},
                "vfa_pool_windows_f32": {"windows": "pool_windows_kernel"},
                "vfa_pool_collapse_relu_sum_f32": {"fused": "pool_collapse_kernel"},
                "vfa_pipe_collapse_relu_sum_f32": {"pipe": "pipe_kernel"}}
ROOFLINE_ENTRY_POINTS = tuple(KERNEL_NAMES)


def committed_traffic(workload, kernel_substr):
    """HBM bytes per dispatch from the committed PMC passes of this command (profiles/rNN_pmc_traffic.json), or None."""
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        # (the PMC passes of the bench default, and -- per workload -- of `--workload <name>`)
        fname = f"{tag}_pmc_traffic.json" if workload == PRIMARY else f"{tag}_{workload}_pmc_traffic.json"
        tpath = os.path.join(REPO, "profiles", fname)
        if os.path.exists(tpath):
            for name, rec in json.load(open(tpath))["kernels"].items():
                if kernel_substr in name:
                    return rec["hbm_bytes_per_dispatch"], f"profiles/{fname}"
    return None, None


def traffic_note(src):
    return (src + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_to_traffic.py; "
            "not measured in this run)") if src else None


def live_products(leg, ops, _lib):
    """32 x 256 x 256 products the fused kernels actually issue on this leg's frame: the (view, tile, layer, scale) items with
    a live box (the kernels skip the others: a fully masked item contributes relu(bias), no product) -- counted once, outside
    every timed region, from the headers of the frame's geometry."""
    import numpy as np
    import torch
    m0 = leg.mods[0]
    n = len(leg.cams)
    if n == 0:
        return 0, 0
    zl, co = m0._kernel_geometry(leg.dev)
    hws = [tuple(f.shape[-2:]) for f in leg.sets[0]]
    L, W, nl = leg.L, leg.W, leg.nl
    rows = L
    while rows > 4 and ops.pipe_workspace_bytes(n, rows, W, nl, len(hws)) > (3 << 30):
        rows = max(4, ((rows + 1) // 2 + 3) // 4 * 4)
    live = total = 0
    with torch.no_grad():
        for r0 in range(0, L, rows):
            band = leg.grid[0, r0:min(L, r0 + rows)]
            ws = ops.pipe_records(leg.calibs, band, zl, co, _lib.CONV_KIND[leg.wl["args"].data], leg.wl["args"].image_size[::-1], hws,
                                  cuts=False)
            lay = ops.pipe_workspace_layout(n, band.shape[0], W, nl, len(hws))
            items = lay["tiles_l"] * lay["tiles_w"] * nl * n
            for s in range(len(hws)):
                flags = ws[lay["hdrs"][s]:lay["hdrs"][s] + items * 32].view(torch.int32).view(-1, 8)[:, 0]
                live += int((flags & 1).sum().item())
                total += items
            del ws
    return live, total


ARITHMETIC = {
    2: ("fp16x2", 2, 3, "two fp16 pieces per operand with a power-of-two scale, three MFMA products (v_mfma_f32_32x32x16_f16), fp32 "
                        "accumulation: 1e-7 ... 3e-7 normwise against float64, the error of an fp32 sgemm on the same operands "
                        "(tests/test_pipe_frame.py, tests/test_fused_frame.py) -- the arithmetic width of the reference's nn.Linear"),
    3: ("bf16x2", 2, 3, "two bf16 pieces per operand, three MFMA products: 16-bit operands, ~2e-6 ... 4e-6 normwise (inside the path's "
                        "tolerance, NARROWER than the reference's fp32 product)"),
    6: ("bf16x3", 3, 6, "three bf16 pieces per operand, six MFMA products: sgemm-class at twice the matrix work"),
}


def roofline_fused(g, workload, entry, live=None, terms=2):
    """Roofline of the fused pooling + collapse kernel (the bound becomes the matrix pipe).  One launch
    covers every (view, scale, layer) of the frame (or of one band of grid rows).  `achieved` = the bf16 MFMA flops the
    kernel ISSUES per launch / mean launch time: `products` bf16 products (3 of a two-piece split, 6 of a three-piece split)
    per fp32 product of the reference's sgemm, over the LIVE 32-row items only (`live`: counted from the frame's
    geometry -- fully masked (view, tile, layer, scale) items are skipped by the kernel), priced against the dense bf16 peak;
    the reference's own fp32 flops (2*M*K*N over every row, masked ones included) against the fp32 matrix peak beside it.  The HBM
    side (integral images read once, BEV map written once) is reported as `hbm_*`."""
    ref_flops = bytes_alg = 0.0
    frames = 0  # the serial kernel's entry point may be called in two stages per frame ("rows": the pre-pass, "main": the rest)
    pipe = entry == "vfa_pipe_collapse_relu_sum_f32"
    for tag, rec in g["by_tag"].items():
        if pipe:
            nv, L, W, nl, hws = tag
        else:
            (nv, L, W, hws, *stage), nl = tag, 1
            if stage and stage[0] == "rows":
                continue
        ref_flops += rec["launches"] * len(hws) * nv * L * W * nl * 2.0 * 256 * 256
        frames += rec["launches"]
        bytes_alg += rec["launches"] * (sum(nv * (h + 2) * (w + 2) * 256 * 4 for h, w in hws) + L * W * 256 * 4 + L * W * 12 + nv * 48)
    g = dict(g, launches=frames)
    avg_s = g["ms"] / g["launches"] * 1e-3
    per_launch = ref_flops / g["launches"]
    label, pieces, products, _ = ARITHMETIC[terms]
    live_frac = (live[0] / live[1]) if live and live[1] else 1.0
    issued = products * per_launch * live_frac
    achieved = issued / avg_s / 1e12
    waves = "8 matrix waves + 4 pooling waves" if terms == 6 else "8 matrix waves + 8 pooling waves"
    def _variant(tag):
        """Template arguments of the pipelined kernel this frame runs (vfa_pipe_seq.h: run_tiles_of and the launch in vfa_pipe.hip,
        restated): the four-step phase (SMALL) for small one- and two-view frames, tile by tile (RT1) for small frames of three and
        more views, else runs of 2 or 4 tiles as a template argument."""
        nv, L, W, nl, hws = tag
        tiles = ((L + 3) // 4) * ((W + 7) // 8)
        nblk = (min(256, tiles) + 7) // 8 * 8
        steps = 2 * nl * nv * len(hws) * tiles // nblk
        rt = (4 if steps >= 250 else 1) if nv <= 2 else (4 if steps >= 1600 else (2 if steps >= 1200 else 1))
        forced = os.environ.get("VFA_AMD_PIPE_RT")
        rt = int(forced) if forced in ("1", "2", "4") else rt
        if terms == 6 or terms == 4:
            return "false, false, 0", rt
        if rt == 1:
            return ("true, false, 0" if nv <= 2 else "false, true, 0"), rt
        return ("false, false, 0", rt) if terms == 3 else (f"false, false, {rt}", rt)
    variants = sorted({_variant(tag) for tag in g["by_tag"]}) if pipe else []
    kernel_id = (f"pipe_kernel<{terms}, false, {variants[0][0]}>" if pipe else f"pool_collapse_kernel<{terms}, false, false>")
    kname = (f"{kernel_id} (runs of {variants[0][1]} tile(s); persistent; {waves} per CU: box pooling from LDS tap windows beside the {label}-split MFMA "
             "collapse of the previous 64 rows x 64 channels; accumulators of four (tile, view) sub-tiles in registers across all z-layers; "
             "bias + ReLU + view / scale sum)") if pipe else \
            (f"pool_collapse_kernel<{terms}, false, false> (persistent, one launch per frame: box pooling of all views x scales from LDS "
             f"tap windows -> {label}-split MFMA collapse -> bias + ReLU + view / scale sum); the HIP events bracket the call that "
             "launches it (+ the empty launch for direct items without a row slot, ~5 us); its pre-pass pool_rows_kernel (the 4 % "
             "of items whose window exceeds LDS, ~28 us) is a separate, untimed call of the entry point")
    traffic, src = committed_traffic(workload, kernel_id)  # (the name as rocprofv3 prints it: all three template arguments)
    return {"bound": "mfma", "kernel": entry + ": " + kname, "achieved": achieved, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / BF16_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_note(src),
            "avg_launch_us": avg_s * 1e6, "mfma_flops_per_launch": issued,
            "arithmetic": label,
            "flops_note": f"{products} 16-bit MFMA products per fp32 product ({pieces}-piece {label[:4]} split; fp16 and bf16 MFMA have the same "
                          f"dense peak) x the {live_frac:.3f} of the 32-row items that have a live box (the kernel skips the rest)",
            "fp32_flops_per_launch": per_launch, "fp32_equivalent_tflops": per_launch / avg_s / 1e12,
            "fp32_mfma_peak_tflops": FP32_MFMA_PEAK_TFLOPS, "frac_of_fp32_mfma_peak": per_launch / avg_s / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "hbm_algorithmic_bytes_per_launch": bytes_alg / g["launches"],
            "hbm_achieved_gbs": bytes_alg / g["launches"] / avg_s / 1e9, "hbm_frac": bytes_alg / g["launches"] / avg_s / 1e9 / HBM_PEAK_GBS,
            "launches": g["launches"]}


def integral_kernel_of(nv, C, sizes, channels_last=False):
    """Which kernels `vfa_integral_images_f32` launches for this call (vfa_integral.hip: integral_images_launch): ONE pass where its
    tiling fits -- NCHW input, >= 64 (view, 16-channel block) units, the column accumulators of every map in LDS --, two passes else."""
    k_rows, k_ch, pitch = 32, 16, 32 * 16 + 16
    onepass = (not channels_last and C % k_ch == 0 and C % 64 == 0 and nv * (C // k_ch) >= 64 and
               all(h >= 8 and w >= 4 and k_rows * pitch * 4 + w * k_ch * 8 <= 160 * 1024 - 1024 for h, w in sizes))
    return onepass


def roofline_integral(g):
    """The integral-image entry point (HBM-bound): algorithmic bytes = every feature map read once + every
    zero-bordered integral image written once (the two-pass kernels move every byte twice: their own traffic is 2 x that, the
    ALGORITHMIC bytes stay what one pass needs), over the mean time of the call (HIP events on the launch stream)."""
    alg = 0.0
    one = two = 0
    for (nv, C, sizes, _affine), rec in g["by_tag"].items():
        alg += rec["launches"] * sum(nv * C * h * w * 4 + nv * C * (h + 2) * (w + 2) * 4 for h, w in sizes)
        if integral_kernel_of(nv, C, sizes):
            one += rec["launches"]
        else:
            two += rec["launches"]
    avg_s = g["ms"] / g["launches"] * 1e-3
    per = alg / g["launches"]
    names = []
    if one:
        names.append("integral_onepass_kernel (both cumsums of a (view, 16-channel block, column part) in one workgroup; one launch for the "
                     "three maps of the frame)")
    if two:
        names.append("rows_batched_kernel + cols_batched_kernel (two passes: fewer than 4 cameras or a shape the one-pass tiling does not "
                     "take; every byte moves twice)")
    return {"bound": "hbm", "kernel": "vfa_integral_images_f32: " + " / ".join(names), "achieved": per / avg_s / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": per / avg_s / 1e9 / HBM_PEAK_GBS, "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": per,
            "launches": g["launches"], "launches_one_pass": one, "launches_two_pass": two, "traffic": None}


def roofline_of(ks, ops, workload, live=None, terms=2):
    """Roofline of the dominant kernel of the step.  Fused path: see `roofline_fused`.  Unfused paths: the pooling kernel
    (HBM-bound by SURVEY.md 8d): algorithmic bytes of one launch = integral images read once + voxel features written once
    + grid + calibs; achieved = those bytes / mean launch time from HIP events recorded on the launch stream inside the
    timed loop."""
    for entry in ("vfa_pipe_collapse_relu_sum_f32", "vfa_pool_collapse_relu_sum_f32"):
        if ks.get(entry, {}).get("launches"):
            return roofline_fused(ks[entry], workload, entry, live, terms)
    entry = "vfa_pool_windows_f32" if ks.get("vfa_pool_windows_f32", {}).get("launches") else "vfa_project_gather_f32"
    g = ks.get(entry, dict(launches=0, ms=0.0, by_tag={}))
    if not g["launches"]:
        return None
    alg_bytes = 0
    for (nv, Ct, Hf, Wf, nlt, cells), rec in g["by_tag"].items():
        alg_bytes += rec["launches"] * (nv * Ct * Hf * Wf * 4 + nv * nlt * cells * Ct * 4 + cells * 12 + nv * 48)
    chosen = ["windows"] if entry == "vfa_pool_windows_f32" else (sorted(set(ops._gather_choice.values())) or ["default"])
    kname = KERNEL_NAMES[entry]
    traffic, src = None, None
    if len(chosen) == 1 and chosen[0] in kname:
        traffic, src = committed_traffic(workload, kname[chosen[0]])
    avg_ms = g["ms"] / g["launches"]
    achieved = alg_bytes / g["launches"] / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": entry + ": " + "/".join(kname.get(c, c) for c in chosen) +
            (" (picked per shape on first use)" if entry == "vfa_project_gather_f32" else ""), "achieved": achieved,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_note(src),
            "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": alg_bytes / g["launches"],
            "launches": g["launches"]}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to record a run of the wrong size",
              file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
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
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)

    import vfa_amd
    from vfa_amd import _lib, ops, vfa_op

    torch.backends.cuda.matmul.allow_tf32 = False
    if a.tune_gemm:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(200)
        torch.cuda.tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"vfa_tunableop_{rank}.csv"))
    scaling = a.scaling or ("strong" if world > 1 else "weak")  # at N = 1 the two coincide; "weak" per the contract

    leg = Leg(a.workload, a, rank, world, dev, scaling)
    leg.step()  # one-off set-up outside warm-up and timing: kernel selection per problem shape
    leg.drain()
    # An MI355X that has idled for a few milliseconds runs its next ~50 ms of work ~10 % below its steady clock (every
    # microbenchmark of tools/ shows it: the same launch takes 560-600 us cold and 490-520 us a second later).  A short run (the
    # driver's 20 steps are 13 ms) would measure the ramp, not the path: a fixed number of untimed frames are queued right in
    # front of the opening bracket of every timed region (`Leg.timed`), the same on every rank.
    conditioning = 0 if a.steps <= 0 else max(0, int(a.condition_steps))
    # Warm-up steps: HIP events around EVERY entry point (the `kernels` table).  Timed steps: only around the roofline
    # kernel, sampled -- each timed launch puts two event records in the queue (~3 us apiece).
    with ops.KernelTimer() as kt_warm:
        for _ in range(a.warmup):
            leg.step()
        leg.drain()
        leg.fence()
    # (every 8th launch: 25 samples spread over the 200 default steps, 3 of a 20-step run.  A timed launch costs two event
    # records in the queue and, on the host, two event creations: in a short region, where the launching thread is barely
    # ahead of the GPU, a sample cost the step ~40 us; every launch of a long one ~1 %)
    # (a sampled launch costs two event records in the queue and two event creations on the host: every 4th launch, and the
    # K-step block is repeated until >= 100 ms are timed, so also the driver's 20-step run collects >= 10 samples)
    min_ms = 100.0 if a.steps > 0 else 0.0
    # (every entry point of the step is sampled in the TIMED blocks -- every 4th launch of each: the `kernels` table and both
    # rooflines come from the region `value` is measured in, not from the warm-up)
    kt = ops.KernelTimer(every=4 if a.steps >= 8 else 1)
    dt = leg.timed(a.steps, kt, lead_in=conditioning, min_ms=min_ms)
    timing = leg.block_stats(a.steps)
    ks = kt.summary()
    live = live_products(leg, ops, _lib)
    primary_terms = 2 if vfa_op.COLLAPSE_TERMS in (0, 2) or (leg.nl == 1 and vfa_op.COLLAPSE_TERMS == 6) else vfa_op.COLLAPSE_TERMS
    roofline = roofline_of(ks, ops, a.workload, live, terms=primary_terms)
    if roofline is not None:
        roofline["sampled"] = (f"HIP events around every {kt.every}th launch of the timed region ({roofline['launches']} samples over "
                               f"{timing['blocks']} blocks of {a.steps} steps)")
    integral_roofline = roofline_integral(ks["vfa_integral_images_f32"]) if ks.get("vfa_integral_images_f32", {}).get("launches") else None
    ks_all = ks if ks else kt_warm.summary()
    def calls_per_frame(v):  # (the fused entry point is called twice per frame: "rows" pre-pass + the rest; count frames)
        rows = sum(r["launches"] for t, r in v["by_tag"].items() if isinstance(t, tuple) and t and t[-1] == "rows")
        return max(v["launches"] - rows, 1)
    kernels = {k: {"launches": calls_per_frame(v), "avg_us": 1e3 * v["ms"] / calls_per_frame(v)} for k, v in ks_all.items()}
    hip_ms = sum(v["ms"] / max(calls_per_frame(v), 1) for v in ks_all.values())  # (mean time per call, summed over the entry points of a frame)
    n, L, W, nl, C = len(leg.cams), leg.L, leg.W, leg.nl, a.channels
    gemm_flops = 3 * 2.0 * n * L * W * (C * nl) * C

    extra = {}
    # ---- same frame stream with ROTATING inputs: `--rotate` distinct lateral sets (4 x 135 MB > the 256 MB Infinity
    # Cache), so no step re-reads the laterals of the previous one from the cache
    if a.rotate > 1 and a.steps > 0:
        rot = Leg(a.workload, a, rank, world, dev, scaling, rotate=a.rotate)
        rot.mods = leg.mods
        for _ in range(max(a.rotate, 3)):
            rot.step()
        dtr = rot.timed(a.steps, lead_in=conditioning, min_ms=min_ms)
        extra["rotating_inputs"] = {"sets": a.rotate, "ms_per_step": 1e3 * dtr / a.steps,
                                    "value": rot.units_step * a.steps / dtr,
                                    "note": "the primary value re-reads the same lateral maps every step (what a frame "
                                            "stream sees when the backbone has just written them); this leg rotates "
                                            f"{a.rotate} distinct input sets through HBM"}
        del rot
    # ---- the same step with fp32 arithmetic in `collapse` (library fp32 GEMM + epilogue kernels): the default forms each
    # fp32 product from three bf16 MFMA products, which is narrower than the reference's sgemm
    if a.fp32_steps > 0 and vfa_op.COLLAPSE_KERNEL != "library":
        saved = vfa_op.COLLAPSE_KERNEL
        vfa_op.COLLAPSE_KERNEL = "library"
        try:
            for _ in range(3):
                leg.step()
            dtf = leg.timed(a.fp32_steps, lead_in=conditioning // 3, min_ms=min_ms)
        finally:
            vfa_op.COLLAPSE_KERNEL = saved
        extra["collapse_fp32_ms_per_step"] = 1e3 * dtf / a.fp32_steps
        extra["collapse_fp32_value"] = leg.units_step * a.fp32_steps / dtf
    # ---- the same frame through the other fused forms, each with its own roofline: the pipelined kernel (vfa_pipe.hip: the kernel of
    # multi-layer grids) in the default arithmetic; the NARROWER two-piece bf16 product that was the default until round 3
    # (`bf16x2_16bit`: not a creditable figure for the reference's fp32 path); three bf16 pieces / six products (pipelined kernel)
    if a.fp32_steps > 0 and a.channels == 256 and len(leg.cams) > 0 and vfa_op.COLLAPSE_KERNEL != "library":
        saved = (vfa_op.PIPE, vfa_op.PIPE_SINGLE_LAYER, vfa_op.COLLAPSE_TERMS)
        try:
            for key, single, terms in (("pipelined_kernel", True, 2), ("bf16x2_16bit", False, 3), ("bf16x3_six_products", True, 6)):
                vfa_op.PIPE, vfa_op.PIPE_SINGLE_LAYER, vfa_op.COLLAPSE_TERMS = True, single, terms
                if key == "pipelined_kernel" and roofline is not None and "vfa_pipe_" in roofline["kernel"]:
                    continue  # (the primary leg already ran this kernel)
                for _ in range(3):
                    leg.step()
                ktp = ops.KernelTimer(only=ROOFLINE_ENTRY_POINTS, every=4)
                dtp = leg.timed(a.fp32_steps, ktp, lead_in=conditioning // 3, min_ms=min_ms)
                rp = roofline_of(ktp.summary(), ops, a.workload, live, terms=terms)
                extra[key + "_ms_per_step"] = 1e3 * dtp / a.fp32_steps
                extra[key] = {"ms_per_step": 1e3 * dtp / a.fp32_steps, "value": leg.units_step * a.fp32_steps / dtp,
                              "timing": leg.block_stats(a.fp32_steps), "roofline": rp, "arithmetic": ARITHMETIC[terms][3]}
        finally:
            vfa_op.PIPE, vfa_op.PIPE_SINGLE_LAYER, vfa_op.COLLAPSE_TERMS = saved
    # ---- what ONE RANK of an N-GPU camera-sharded run computes per frame, measured on this one GPU (no collective): the cameras
    # `camera_shard(n, 0, N)` gives rank 0, for N = 2, 4, 8, on this workload and on BASELINE.json configs[4].  A PROJECTION of the
    # per-rank compute time -- the scaling curve itself needs the node (the driver's SCALE run)
    if a.proxy_steps > 0 and world == 1 and a.channels == 256 and a.workload == PRIMARY:
        proxy = {}
        for wname, steps in ((a.workload, a.proxy_steps), (C5, max(1, a.proxy_steps // 10)) if a.c5_steps > 0 else (None, 0)):
            if wname is None:
                continue
            per = {}
            full = None
            for nranks in (1, 2, 4, 8):
                lg = Leg(wname, a, 0, nranks, dev, "strong")
                lg.collective = False  # (rank 0's share of the cameras; nothing to reduce with on one GPU)
                lg.world = 1
                lg.step()
                t = lg.timed(steps, lead_in=conditioning // 3 if wname == a.workload else 1, min_ms=min_ms if wname == a.workload else 0.0)
                ms = 1e3 * t / steps
                full = ms if nranks == 1 else full
                per[str(nranks)] = {"cameras_of_rank0": len(lg.cams), "ms_per_frame": ms,
                                    "vs_full_rig_share": ms / (full * len(lg.cams) / lg.n_frame) if full and lg.cams else None}
                del lg
            proxy[wname] = per
        extra["per_rank_proxy"] = dict(proxy, note="one GPU, rank 0's cameras of an N-rank camera-sharded frame, no collective: ms_per_frame "
                                       "and its ratio to (time of the full rig on one GPU) x (share of the cameras); a projection of the "
                                       "per-rank compute time, NOT a scaling measurement")
    # ---- the drop-in AS THE REFERENCE CALLS IT (vfanet.py:64-82): per camera three `VFA.forward` calls, `f8 + f16 + f32`, `ortho +=`
    # -- what train.py / predict.py get from the one-line import swap of INTEGRATION.md section 1, without `aggregate_views`
    if a.loop_steps > 0 and world == 1 and a.channels == 256 and len(leg.cams) > 0:
        lats = leg.sets[0]
        # (the reference forms lat8 / lat16 / lat32 per camera inside its loop -- relu(bn(lat(feat))), vfanet.py:72-74 --: the
        # projector gets 21 SEPARATE tensors, resident in HBM when the path starts; they are made here, outside the timed loop)
        per_cam = [[lats[s][cam:cam + 1].clone() for s in range(3)] for cam in range(len(leg.cams))]

        def reference_loop():
            ortho = 0
            for cam in range(len(leg.cams)):
                lat8, lat16, lat32 = per_cam[cam]
                f8 = leg.mods[0](lat8, leg.calibs[cam], leg.grid)
                f16 = leg.mods[1](lat16, leg.calibs[cam], leg.grid)
                f32 = leg.mods[2](lat32, leg.calibs[cam], leg.grid)
                ortho = ortho + (f8 + f16 + f32)
            # (the reference hands `ortho` to `self.fuse`, a conv: the first torch function it meets computes a deferred result -- here
            # that use is spelled out; with VFA_AMD_LAZY=0 every call above has already computed)
            return vfa_amd.materialize(ortho)

        with torch.no_grad():
            for _ in range(3):
                got_loop = reference_loop()
            want_loop = vfa_amd.aggregate_views(*leg.mods, *lats, leg.calibs, leg.grid, distributed=False)
            leg.fence()
            scale_l = float(want_loop.abs().max())
            err_l = float((got_loop - want_loop).abs().max()) / scale_l
            reps = max(3, a.loop_steps)
            gc.collect()
            t0 = time.perf_counter()
            for _ in range(reps):
                reference_loop()
            leg.fence()
            ms_loop = 1e3 * (time.perf_counter() - t0) / reps
        extra["reference_loop"] = {"ms_per_frame": ms_loop, "value": leg.units_step / (ms_loop * 1e-3), "calls_per_frame": 3 * len(leg.cams),
                                   "vs_batched_frame": ms_loop / (1e3 * dt / a.steps), "max_abs_diff_vs_batched_over_max": err_l,
                                   "note": "the camera loop of the reference's VFANet.forward (vfanet.py:64-82) on this build's VFA modules: "
                                           "7 cameras x 3 VFA.forward calls + the Python sums -- the drop-in without the batched "
                                           "aggregate_views; same frame, same inputs; host clock around `reps` frames.  In inference VFA.forward "
                                           "DEFERS (vfa_amd/lazy.py): the calls and sums only record, the first use of `ortho` runs one batched "
                                           "frame; `eager_calls_ms_per_frame` is the same loop with VFA_AMD_LAZY=0 (21 launches of everything)"}
        from vfa_amd import lazy as _lazy
        if _lazy.LAZY:
            _lazy.LAZY = False
            try:
                with torch.no_grad():
                    reference_loop()
                    leg.fence()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        reference_loop()
                    leg.fence()
                extra["reference_loop"]["eager_calls_ms_per_frame"] = 1e3 * (time.perf_counter() - t0) / 3
            finally:
                _lazy.LAZY = True
    # ---- a STATIC rig: the geometry (box records, windows, work cuts, split weights) formed once, frames = integral images + the frame
    # kernel (vfa_amd.FrameGeometry: an explicit contract of the caller).  NOT `value`: the reference projects every frame, and so does
    # the primary leg.  For rank 0's cameras of an N-rank frame too (the fixed per-frame costs a rank cannot shrink).
    if a.loop_steps > 0 and world == 1 and a.channels == 256 and len(leg.cams) > 0:
        static = {}
        for nranks in (1, 8):
            cams = leg.cams if nranks == 1 else vfa_amd.camera_shard(leg.n_frame, 0, nranks)
            if not cams:
                continue
            idx = torch.tensor(cams, dtype=torch.long, device=dev)
            feats = [f[idx].contiguous() for f in leg.sets[0]]
            cal = leg.calibs[idx].contiguous()
            try:
                geom = vfa_amd.FrameGeometry(leg.mods, cal, leg.grid, [tuple(f.shape[-2:]) for f in feats])
            except ValueError:
                break  # (a frame processed in bands: one geometry per band, not timed here)
            with torch.no_grad():
                want_s = vfa_amd.aggregate_views(*leg.mods, *feats, cal, leg.grid, distributed=False)
                same = bool(torch.equal(geom.frame(feats), want_s))
                for _ in range(5):
                    geom.frame(feats)
                leg.fence()
                reps = max(10, 4 * a.loop_steps)
                gc.collect()
                t0 = time.perf_counter()
                for _ in range(reps):
                    geom.frame(feats)
                leg.fence()
                ms_s = 1e3 * (time.perf_counter() - t0) / reps
                t0 = time.perf_counter()
                for _ in range(reps):
                    vfa_amd.aggregate_views(*leg.mods, *feats, cal, leg.grid, distributed=False)
                leg.fence()
                ms_d = 1e3 * (time.perf_counter() - t0) / reps
            static[str(nranks)] = {"cameras": len(cams), "ms_per_frame": ms_s, "ms_per_frame_geometry_every_frame": ms_d,
                                   "bit_identical": same}
            # ... and the same frame captured into a hipGraph (static shapes, static buffers: what a deployed rank replays per frame)
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(2):
                        geom.frame(feats)
                torch.cuda.current_stream(dev).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph), torch.no_grad():
                    g_out = geom.frame(feats)
                graph.replay()
                leg.fence()
                same_g = bool(torch.equal(g_out, want_s))
                t0 = time.perf_counter()
                for _ in range(reps):
                    graph.replay()
                leg.fence()
                static[str(nranks)].update(ms_per_frame_graphed=1e3 * (time.perf_counter() - t0) / reps, graphed_bit_identical=same_g)
                del graph, g_out
            except Exception as exc:  # (a capture failure must not take the bench line down)
                static[str(nranks)]["graph_error"] = repr(exc)[:200]
            del geom
        if static:
            extra["static_rig"] = dict(static, note="vfa_amd.FrameGeometry: the rig's geometry computed ONCE (the caller's contract: cameras, grid and "
                                       "weights stand still), then per frame the integral images + the persistent kernel; keys = ranks of a "
                                       "camera-sharded frame (1: the whole rig; 8: rank 0's camera); beside it the same frames through "
                                       "aggregate_views, which recomputes the geometry every frame like the reference and like `value`")
    # ---- the producer in front of the path, NOT part of `value` (the path starts at lateral maps resident in HBM):
    # trunk outputs -> the three integral images through the hand-written lateral branch (fp32-MFMA 1x1 convolution, channels-last,
    # GroupNorm statistics in its epilogue; affine + ReLU inside the row scan) and through the library's operations
    if a.fp32_steps > 0 and a.channels == 256 and len(leg.cams) > 0 and a.workload == PRIMARY:
        import torch.nn.functional as F
        n = len(leg.cams)
        gen = torch.Generator(device="cpu").manual_seed(7)
        shapes = [(k,) + tuple(leg.sets[0][i].shape[-2:]) for i, k in enumerate((128, 256, 512))]
        feats = [torch.randn(n, k, h, w, generator=gen).to(dev) for k, h, w in shapes]
        convs = [torch.nn.Conv2d(k, 256, 1).to(dev) for k, _, _ in shapes]
        gns = [torch.nn.GroupNorm(16, 256).to(dev) for _ in shapes]

        def hand():
            parts = ops.lateral_convs([(f, c.weight, c.bias, g.weight, g.bias, g.eps) for f, c, g in zip(feats, convs, gns)])
            return ops.integral_images([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts], channels_last=True)

        def library():
            return ops.integral_images([F.relu(g(c(f))) for f, c, g in zip(feats, convs, gns)])

        def ms_of(fn, reps=20):
            with torch.no_grad():
                for _ in range(5):
                    fn()
                leg.fence()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                leg.fence()
            return 1e3 * (time.perf_counter() - t0) / reps

        extra["producer_f3"] = {"hand_written_ms_per_frame": ms_of(hand), "library_ms_per_frame": ms_of(library), "cameras": n,
                                "note": "trunk outputs (n, 128/256/512, h, w) -> three integral images: vfa_lateral_convs_f32 (the three scales in one launch) + "
                                        "vfa_integral_images_hwc_f32, against MIOpen conv + torch GroupNorm + ReLU x 3 + "
                                        "vfa_integral_images_f32; outside the timed region of `value`"}
    # ---- N > 1: what the collective costs.  The same steps without it (slowest rank's compute), and the three ways of fusing the
    # map timed alone on a map-sized tensor: all-reduce, reduce -> rank 0, reduce-scatter over BEV rows + the heads' 7-row halo
    if world > 1 and a.steps > 0:
        from vfa_amd.aggregate import all_reduce_ortho, reduce_ortho, reduce_scatter_ortho
        leg.collective = False
        dtc = leg.timed(a.steps, lead_in=conditioning // 3, min_ms=min_ms)
        leg.collective = True
        probe = torch.zeros((leg.L * leg.W, 256), dtype=torch.float32, device=dev)

        def coll_ms(fn, reps=10):
            for _ in range(3):
                fn()
            leg.fence()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            leg.fence()
            t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return 1e3 * t.item()
        extra["multi_gpu"] = {
            "ms_compute": 1e3 * dtc / a.steps,
            "ms_collective_exposed": 1e3 * (dt - dtc) / a.steps,
            "map_bytes": probe.numel() * 4,
            "collective_alone_ms": {"all_reduce": coll_ms(lambda: all_reduce_ortho(probe)),
                                    "reduce_to_rank0": coll_ms(lambda: reduce_ortho(probe, 0)),
                                    "reduce_scatter_rows_halo7": coll_ms(lambda: reduce_scatter_ortho(probe, leg.L, leg.W, halo=7))},
            "note": "ms_compute: the same steps with the collective switched off (MAX over ranks); ms_collective_exposed = ms_per_step - "
                    "ms_compute (the collective of frame i -- config.collective -- runs beside the projection of frame i + 1); collective_alone_ms: one "
                    "collective of the map at a time, nothing else on the GPUs"}
        del probe
    # ---- the reference's SHIPPED configs (vfa/config.py:5-28 MultiviewC 156 x 156 x 5, :60-85 Wildtrack 120 x 360 x 8, :32-57 MultiviewX
    # 160 x 250 x 8): multi-layer grids, all through pipe_kernel -- the kernel `value` does not run.  A few steps each, with the
    # kernel's own time (HIP events) and roofline, so that its numbers are timed by whoever runs this file (round-5 verdict, missing 3)
    if a.shipped_steps > 0 and world == 1 and a.workload == PRIMARY and a.channels == 256 and vfa_op.COLLAPSE_KERNEL != "library":
        shipped = {}
        for wname in SHIPPED:
            lg = Leg(wname, a, 0, 1, dev, "weak")
            lg.step()
            lg.drain()
            kts = ops.KernelTimer(only=ROOFLINE_ENTRY_POINTS + ("vfa_integral_images_f32",), every=1)
            dts = lg.timed(a.shipped_steps, kts, lead_in=min(conditioning, 20), min_ms=30.0 if a.steps > 0 else 0.0)
            kss = kts.summary()
            rs = roofline_of(kss, ops, wname, live_products(lg, ops, _lib), terms=primary_terms)
            alg = SURVEY_ALG_BYTES.get(wname)
            shipped[wname] = {"grid": [lg.L, lg.W, lg.nl], "cameras": len(lg.cams), "units_per_step": lg.units_step,
                              "ms_per_step": 1e3 * dts / a.shipped_steps, "value": lg.units_step * a.shipped_steps / dts, "unit": "voxels/s",
                              "timing": lg.block_stats(a.shipped_steps), "roofline": rs,
                              "frac_of_survey_hbm_roofline": (lg.units_step * a.shipped_steps / dts) / (lg.units_step / (alg / (HBM_PEAK_GBS * 1e9))) if alg else None,
                              "integral_images_avg_us": (1e3 * kss["vfa_integral_images_f32"]["ms"] / kss["vfa_integral_images_f32"]["launches"])
                              if kss.get("vfa_integral_images_f32", {}).get("launches") else None}
            del lg
        extra["shipped_configs"] = dict(shipped, note="the reference's shipped configs (vfa/config.py:5-85), one GPU, whole frame per step (integral images + "
                                        "geometry + pipe_kernel), steps bracketed like `value`; roofline = the pipelined kernel, HIP events around every launch; "
                                        "frac_of_survey_hbm_roofline = value / (units per frame / (SURVEY Appendix C algorithmic bytes / 8 TB/s))")
    # ---- the training step of the reference's trainer (trainer.py:41: loss.backward() through the aggregate): forward + backward of the
    # bench frame through the fused autograd node, a handful of steps, with the time of every entry point
    if a.train_steps > 0 and world == 1 and a.workload == PRIMARY and a.channels == 256:
        import vfa_amd
        from vfa_amd.synthetic import make_workload
        twl = make_workload(PRIMARY, channels=256, seed=0)
        tn = twl["n_cam"]
        torch.manual_seed(0)
        tmods = [vfa_amd.VFA(256, grid_height=twl["grid_height"], cube_size=twl["cube_size"], args=twl["args"]).to(dev) for _ in range(3)]
        tlats = [torch.cat([twl["features"][c][s] for c in range(tn)]).to(dev).requires_grad_(True) for s in range(3)]
        tcal, tgrid = twl["calibs"].to(dev), twl["grid"].to(dev)

        def train_step():
            out = vfa_amd.aggregate_views(*tmods, *tlats, tcal, tgrid)
            out.square().mean().backward()
            for t in tlats:
                t.grad = None
            for m in tmods:
                m.zero_grad(set_to_none=True)

        for _ in range(2):
            train_step()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.train_steps):
            train_step()
        e1.record()
        torch.cuda.synchronize()
        ktt = ops.KernelTimer(every=1)  # (a second pass for the entry points: events around every call, not part of ms_per_step)
        with ktt:
            for _ in range(a.train_steps):
                train_step()
            torch.cuda.synchronize()
        extra["training_step"] = {"workload": PRIMARY, "steps": a.train_steps, "ms_per_step": e0.elapsed_time(e1) / a.train_steps,
                                  "peak_memory_gb": torch.cuda.max_memory_allocated() / 1e9,
                                  "entry_points_ms_per_step": {k: v["ms"] / a.train_steps for k, v in ktt.summary().items()},
                                  "note": "forward (the fused frame kernel) + backward (pooling again in cell chunks, the forward's own product "
                                          "recomputed with the ReLU mask as its epilogue, the two gradient products, the scatter on 4 x 8 patches, "
                                          "the integral images' adjoint) of loss = mean(out^2) w.r.t. features, collapse weights and biases; "
                                          "HIP events around the steps; not `value`"}
        del tmods, tlats, twl
    # ---- BASELINE.json configs[4]: synthetic 8 x 4K -> 512 x 512 x 32, cameras sharded over the ranks
    if a.c5_steps > 0 and a.workload == PRIMARY and a.channels == 256:
        c5 = Leg(C5, a, rank, world, dev, "strong")
        c5.step()
        c5.drain()
        dt5 = c5.timed(a.c5_steps, lead_in=1 if conditioning else 0)
        extra["scaling_curve_c5"] = {"workload": C5, "n_gpus": world, "steps": a.c5_steps, "scaling": "strong",
                                     "cameras_per_rank": len(c5.cams), "cameras_total": c5.n_frame,
                                     "grid": [c5.L, c5.W, c5.nl], "units_per_step": c5.units_step,
                                     "ms_per_step": 1e3 * dt5 / a.c5_steps, "value": c5.units_step * a.c5_steps / dt5,
                                     "unit": "voxels/s", "ortho_bytes": c5.L * c5.W * 256 * 4}
        del c5

    ck = ks_all.get("vfa_collapse_relu_sum_f32")
    fk = ks_all.get("vfa_pool_collapse_relu_sum_f32") or ks_all.get("vfa_pipe_collapse_relu_sum_f32")
    if fk and fk["launches"]:
        us = 1e3 * fk["ms"] / fk["launches"]
        collapse_info = {"flops_per_step": gemm_flops, "backend": "fused into " + ("vfa_pool_collapse_relu_sum_f32" if "vfa_pool_collapse_relu_sum_f32" in ks_all
                                                                    else "vfa_pipe_collapse_relu_sum_f32") + " (pooled rows go "
                         "from registers to LDS fp16 hi/lo planes to the three-product fp16-split MFMA, fp32 accumulate; bias + ReLU + "
                         "view/scale sum in the epilogue; the voxel features never reach HBM)", "avg_us": us,
                         "fp32_equivalent_tflops": gemm_flops / (us * 1e-6) / 1e12,
                         "mfma_16bit_tflops": 3 * gemm_flops / (us * 1e-6) / 1e12,
                         "error_vs_fp64": "1e-7 ... 3e-7 normwise, at or below an fp32 library GEMM on the same voxel features "
                                          "(tests/test_fused_frame.py, tests/test_pipe_frame.py); the path tolerates rtol 1e-4 / atol 1e-5 max"}
    elif ck and ck["launches"]:
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
            "metric": "voxels aggregated/sec (7 views->BEV grid)", "value": leg.units_step * a.steps / dt,
            "unit": "voxels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "lead_in_frames": conditioning,  # untimed frames queued in front of the opening barrier + synchronize (clock ramp of an idle device)
            "ms_per_step": 1e3 * dt / a.steps, "timing": timing, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 in / out, fp32 accumulation everywhere.  Pre-GEMM: the reference's exact fp32 rounding sequence (the fused "
                          "kernels' pooled rows -- tap chains, box sum, correctly rounded quotient -- are the reference's voxel features bit for bit: tests/test_fused_frame.py). "
                          "The collapse product (reference: fp32 nn.Linear) = " + ARITHMETIC[primary_terms][3] + ".  Extra keys: "
                          "`bf16x2_16bit` = the narrower two-piece bf16 product (the default until round 3; NOT reference width), "
                          "`bf16x3_six_products` = three bf16 pieces, `collapse_fp32_ms_per_step` = the unfused step with the fp32 library GEMM",
            "arithmetic": ARITHMETIC[primary_terms][0],
            "config": {"workload": a.workload, "cameras_per_rank": n,
                       "cameras_total": n * world if scaling == "weak" else leg.n_frame, "channels": C,
                       "feature_maps": [list(s) for s in leg.wl["feat_sizes"]], "grid": [L, W, nl],
                       "units_per_step": leg.units_step,
                       "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                       "parallelism": (f"camera-sharded dp{world} ({scaling}), RCCL "
                                       + ("reduce of the BEV map onto rank 0" if leg.mode == "reduce" else "all-reduce of the BEV map")
                                       + (" overlapped with the next frame" if leg.overlap else "")) if world > 1
                       else "single GPU",
                       **({"collective": leg.mode} if world > 1 else {})},
            "bev_cells_per_s": nl * L * W * a.steps / dt,
            "roofline": roofline,
            "roofline_integral": integral_roofline,
            "kernels": kernels,
            "kernels_note": f"mean time per call of every entry point of the step, from HIP events around every {kt.every}th call inside the "
                            "TIMED blocks (the geometry calls run on a second stream beside the integral images)",
            "hip_kernel_ms_per_step": hip_ms,
            "collapse_gemm": collapse_info,
        }
        out.update(extra)
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(leg.wl, a.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
