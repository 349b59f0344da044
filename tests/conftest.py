import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(name):
    return os.path.join(GOLDEN, name)


VFA_CASES = [
    "mc_cam0_s32.npz", "mc_cam1_s16.npz", "mc_inside_s8.npz", "mc_signed_bigcube.npz", "mc_dense_s32.npz",
    "mc_nl1.npz", "wt_cam0_s8.npz", "wt_cam1_s32.npz", "wt_side_dense_s16.npz", "mx_cam0_s16.npz",
    "mx_cam1_s8.npz",
]
# *_nl1: single-layer grids at C = 256 (K = N = 256), the shape the flagship MFMA collapse kernels are built for
VFANET_CASES = ["vfanet_mc.npz", "vfanet_wt.npz", "vfanet_mc_nl1.npz", "vfanet_wt_nl1.npz"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import vfa_oracle
    vfa_oracle.build()
    return vfa_oracle
