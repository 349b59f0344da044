"""Step order of the pipelined frame kernel (vfa_amd/csrc/vfa_pipe_seq.h, shared host / device code) on the CPU: the C++
harness tests/native/pipe_seq_harness.cpp enumerates the steps of every workgroup for random live-view masks, layer counts and
work cuts and checks that each (tile, scale, live view, layer, quarter) is visited exactly once, in groups of at most four
views, sets alternating with the step index, with consistent first / last flags."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pipe_seq") / "harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(REPO, "tests", "native", "pipe_seq_harness.cpp")])
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_every_step_once(harness, seed):
    out = subprocess.run([harness, str(seed), "400"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
