"""The CPU oracle (oracle/vfa_oracle.c) against fixtures generated from the reference itself.

Bar: every pre-GEMM stage tensor bitwise; ortho after the GEMM within
rtol 1e-4, atol 1e-5*max|ref| (the reference's own MKL summation order is not reproducible).
"""
import numpy as np
import pytest

from conftest import VFA_CASES, VFANET_CASES, golden_path


def _bit_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.uint32)
    return a == b


def _assert_bitwise(name, got, ref, allow_zero_sign=False):
    eq = _bit_equal(got, ref)
    if allow_zero_sign:
        eq |= (np.asarray(got) == 0) & (np.asarray(ref) == 0)
    assert eq.all(), f"{name}: {np.count_nonzero(~eq)} of {eq.size} elements differ bitwise"


def _meta(d):
    return dict(data=str(d["data"]), image_size=tuple(int(v) for v in d["image_size"]),
                cube_size=tuple(float(v) for v in d["cube_size"]), grid_height=float(d["grid_height"]))


@pytest.mark.parametrize("case", VFA_CASES)
def test_oracle_stages_bitwise(oracle, case):
    d = np.load(golden_path(case))
    m = _meta(d)
    st = oracle.vfa_forward(d["feature"], d["calib"], d["grid"], d["weight"], d["bias"], stages=True, **m)
    _assert_bitwise("integral", st["integral"], d["integral"])
    _assert_bitwise("box", st["box"], d["box"])
    _assert_bitwise("area", st["area"], d["area"])
    assert np.array_equal(st["visible"], d["visible"])
    _assert_bitwise("vox", st["vox"], d["vox"])
    ref = d["ortho"]
    np.testing.assert_allclose(st["ortho"], ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("case", VFA_CASES)
def test_oracle_buffers_match_reference_state(oracle, case):
    """z_corners / corners_offset buffers of the reference module (vfa_op.py:50-55)."""
    d = np.load(golden_path(case))
    m = _meta(d)
    zl = oracle.z_layers_of(m["grid_height"], m["cube_size"])
    assert np.array_equal(zl, d["z_corners"][:, 0, 0, 2].astype(np.float32))
    assert np.all(d["z_corners"][..., :2] == 0)
    assert np.array_equal(oracle.corner_offsets(m["cube_size"]), d["corners_offset"].reshape(8, 3))


@pytest.mark.parametrize("case", VFANET_CASES)
def test_oracle_vfanet_aggregate(oracle, case):
    d = np.load(golden_path(case))
    m = _meta(d)
    lats = {s: d[f"lat{s}"] for s in (8, 16, 32)}
    ws = {s: d[f"weight{s}"] for s in (8, 16, 32)}
    bs = {s: d[f"bias{s}"] for s in (8, 16, 32)}
    out = oracle.vfanet_aggregate(lats, d["calibs"], d["grid"], ws, bs, **m)
    ref = d["ortho"]
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def test_oracle_make_grid(oracle):
    d = np.load(golden_path("make_grid.npz"))
    for key in d.files:
        if key.endswith("_args"):
            continue
        a = d[key + "_args"]
        name = key.split("_")[0]
        g = oracle.make_grid(world_size=(a[0], a[1]), cube_LW=(a[2], a[3]), grid_offset=(a[4], a[5], a[6]),
                             dataset=name)
        assert g.shape == d[key].shape, key
        assert np.array_equal(g, d[key]), key


def test_product_make_grid_matches_reference_fixture():
    """The PRODUCT's ``vfa_amd.make_grid`` (host code, runs on CPU) against the reference's own outputs
    (``vfa/utils.py:16-37``): bit-equal for the three dataset kinds, offsets and ragged sizes included."""
    import torch
    import vfa_amd
    d = np.load(golden_path("make_grid.npz"))
    n = 0
    for key in d.files:
        if key.endswith("_args"):
            continue
        a = d[key + "_args"]
        name = key.split("_")[0]
        g = vfa_amd.make_grid(world_size=(a[0], a[1]), cube_LW=[a[2], a[3]], grid_offset=(a[4], a[5], a[6]),
                              dataset=name)
        assert g.dtype == torch.float32 and tuple(g.shape) == d[key].shape, key
        assert np.array_equal(g.numpy().view(np.uint32), d[key].view(np.uint32)), key
        n += 1
    assert n == 5


@pytest.mark.parametrize("case", VFA_CASES)
def test_torch_restatement_bitwise(case):
    """oracle/torch_reference.py (the CPU baseline bench.py times, and in float64 the gradient reference) against the
    reference's own stage tensors: same torch ops in the same order, so every pre-GEMM tensor is bit-identical."""
    import torch
    from oracle import torch_reference as tr
    torch.set_num_threads(2)
    d = np.load(golden_path(case))
    m = _meta(d)
    zl = torch.from_numpy(d["z_corners"][:, 0, 0, 2].copy())  # int64, like the reference buffer
    co = torch.from_numpy(d["corners_offset"].reshape(8, 3))
    with torch.no_grad():
        st = tr.vfa_stages(torch.from_numpy(d["feature"])[None], torch.from_numpy(d["calib"]), torch.from_numpy(d["grid"]),
                           zl, co, m["data"], m["image_size"])
        out = tr.vfa_forward(torch.from_numpy(d["feature"])[None], torch.from_numpy(d["calib"]),
                             torch.from_numpy(d["grid"]), torch.from_numpy(d["weight"]), torch.from_numpy(d["bias"]), zl,
                             co, m["data"], m["image_size"])
    for key, got in (("box", st["box"][0]), ("area", st["area"][0, 0]), ("integral", st["integral"][0]), ("vox", st["vox"])):
        a, b = got.numpy(), d[key]
        assert a.shape == b.shape, key
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), f"{key}: {np.count_nonzero(~same)} elements differ"
    assert np.array_equal(st["visible"][0, 0].numpy(), d["visible"])
    np.testing.assert_allclose(out[0].numpy(), d["ortho"], rtol=1e-4, atol=1e-5 * np.abs(d["ortho"]).max())
