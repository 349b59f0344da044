"""bench.py on the GPU box: the self-launching N > 1 path and the contract of its JSON line.   ``-m gpu``."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

FAST = ["--steps", "3", "--warmup", "1", "--c5-steps", "0", "--cpu-seconds", "0", "--rotate", "0", "--fp32-steps", "0", "--proxy-steps", "0"]


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=timeout, cwd=REPO)
    return p


def test_bench_spawns_its_own_ranks_and_shards_cameras():
    """`python bench.py --gpus 2` without a launcher must run TWO ranks (the driver calls it exactly like this).  On a
    1-GPU box both ranks share device 0 and the collective runs over gloo (functional check only, never a number)."""
    two_gpus = torch.cuda.device_count() >= 2
    p = _run(["--gpus", "2"] + FAST, env={} if two_gpus else {"VFA_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["collective"] == "reduce"
    assert line["scaling"] == "strong" and line["config"]["workload"] == "multiviewc_200x200x1"
    assert line["config"]["cameras_total"] == 7 and line["config"]["cameras_per_rank"] == 4  # rank 0: cameras 0,2,4,6
    assert line["config"]["units_per_step"] == 7 * 3 * 200 * 200
    assert line["value"] > 0 and line["ms_per_step"] > 0
    mg = line["multi_gpu"]  # what the collective costs: compute-only time of the slowest rank, and the three collectives alone
    assert mg["ms_compute"] > 0 and mg["map_bytes"] == 200 * 200 * 256 * 4
    assert set(mg["collective_alone_ms"]) == {"all_reduce", "reduce_to_rank0", "reduce_scatter_rows_halo7"}


def test_bench_refuses_a_world_size_mismatch():
    p = _run(["--gpus", "4"] + FAST, env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "refusing" in p.stderr


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_two_rank_rccl_sum_matches_single_gpu():
    """Camera-sharded aggregate over RCCL == the single-GPU aggregate of all cameras (tools/check_rccl_shard.py)."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(REPO, "tools", "check_rccl_shard.py")], capture_output=True, text=True, timeout=900,
                       cwd=REPO)
    assert p.returncode == 0 and "rccl shard ok" in p.stdout, p.stdout[-1000:] + p.stderr[-2000:]


def test_bench_single_gpu_line_has_the_contract_fields():
    p = _run(["--steps", "5", "--warmup", "2", "--c5-steps", "0", "--cpu-seconds", "2", "--rotate", "2", "--fp32-steps", "2", "--proxy-steps", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["vs_baseline"] is None and line["dtype"] == "f32"
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] in ("GB/s", "TFLOP/s")
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["one_thread_value"] > 0 and c["cpu"]
    assert line["collapse_fp32_ms_per_step"] > 0 and line["rotating_inputs"]["ms_per_step"] > 0
    t = line["timing"]  # the K-step block is repeated until >= 100 ms are timed; the line carries the median block
    assert t["blocks"] >= 1 and t["timed_ms_total"] >= 100.0 and t["ms_per_step_min"] <= line["ms_per_step"] <= t["ms_per_step_max"]
    assert r["launches"] >= 10  # HIP-event samples of the roofline kernel
    # the headline runs the reference-width product (two fp16 pieces, three products); the narrower bf16 form and the six-product
    # form of the same fused step ride along, each with its own roofline
    assert line["arithmetic"] == "fp16x2" and "pool_collapse_kernel<2" in r["kernel"] and r["arithmetic"] == "fp16x2"
    f = line["bf16x3_six_products"]
    assert f["ms_per_step"] > 0 and f["roofline"]["bound"] == "mfma" and "pipe_kernel<6" in f["roofline"]["kernel"]
    assert "pool_collapse_kernel<3" in line["bf16x2_16bit"]["roofline"]["kernel"]
    assert "pipe_kernel<2" in line["pipelined_kernel"]["roofline"]["kernel"] and line["pipelined_kernel"]["roofline"]["frac"] > 0
    ri = line["roofline_integral"]
    assert ri["bound"] == "hbm" and 0 < ri["frac"] < 1 and ri["launches"] >= 1
    assert "vfa_integral_images_f32" in line["kernels"] and "vfa_frame_cuts_f32" in line["kernels"]
    pr = line["per_rank_proxy"]["multiviewc_200x200x1"]
    assert pr["8"]["cameras_of_rank0"] == 1 and pr["2"]["cameras_of_rank0"] == 4 and pr["1"]["ms_per_frame"] > pr["8"]["ms_per_frame"] > 0
    pf = line["producer_f3"]  # the producer in front of the path (f3): hand-written lateral branch against the library's operations
    assert 0 < pf["hand_written_ms_per_frame"] < pf["library_ms_per_frame"] and pf["cameras"] == 7
    ts = line["training_step"]  # forward + backward of the bench frame through the fused autograd node (f2)
    assert ts["ms_per_step"] > 0 and ts["steps"] == 5 and "vfa_project_gather_backward_grid_f32" in ts["entry_points_ms_per_step"]
    assert "vfa_collapse_gemm_relu_backward_f16_f32" in ts["entry_points_ms_per_step"]
