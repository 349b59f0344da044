"""The C-ABI library loads on a CPU-only box and exports every symbol include/vfa_hip.h declares.

No compute call is made here (there is no GPU); the parity tests proper are tests/test_hip_parity.py (-m gpu).
"""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO


@pytest.fixture(scope="module")
def built_lib():
    from vfa_amd import build
    return build.build()


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "vfa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t)\s+(vfa_\w+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    assert "vfa_project_gather_f32" in syms and "vfa_integral_image_f32" in syms
    assert len(syms) >= 7


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/vfa_hip.h but not exported"
    assert lib.vfa_abi_version() == 9
    assert not hasattr(lib, "vfa_set_option")  # since ABI v2: tuning choices are per-call flags, the library keeps no state


def test_python_binding_covers_every_declared_symbol(built_lib):
    from vfa_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    assert _lib.lib().vfa_abi_version() == _lib.ABI_VERSION


def test_code_object_targets_gfx950(built_lib):
    blob = open(built_lib, "rb").read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob


def test_product_refuses_cpu_tensors(built_lib):
    """No CPU fallback: the product path raises instead of computing on the host."""
    from types import SimpleNamespace
    import vfa_amd
    from vfa_amd._lib import VFAHipError
    m = vfa_amd.VFA(4, args=SimpleNamespace(data="MultiviewC", image_size=(720, 1280)))
    with pytest.raises(VFAHipError):
        m(torch.zeros(1, 4, 6, 8), torch.zeros(3, 4), torch.zeros(1, 3, 3, 3))


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "vfa_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "vfa_oracle" not in text, f


def test_state_dict_matches_reference_layout():
    """Keys / shapes / dtypes of the reference module's state (checkpoint row)."""
    from types import SimpleNamespace
    import vfa_amd
    m = vfa_amd.VFA(256, grid_height=160, cube_size=(25, 25, 32), feat_scale=1 / 8.,
                    args=SimpleNamespace(data="MultiviewC", image_size=(720, 1280)))
    sd = m.state_dict()
    assert list(sd) == ["z_corners", "corners_offset", "collapse.weight", "collapse.bias"]
    assert sd["z_corners"].dtype == torch.int64 and tuple(sd["z_corners"].shape) == (5, 1, 1, 3)
    assert sd["z_corners"][:, 0, 0, 2].tolist() == [0, 32, 64, 96, 128]
    assert tuple(sd["corners_offset"].shape) == (1, 1, 1, 1, 8, 3) and sd["corners_offset"].dtype == torch.float32
    assert tuple(sd["collapse.weight"].shape) == (256, 1280) and tuple(sd["collapse.bias"].shape) == (256,)


def test_unknown_dataset_raises_like_the_reference():
    from types import SimpleNamespace
    from vfa_amd.vfa_op import _conv_kind
    with pytest.raises(UnboundLocalError):
        _conv_kind(SimpleNamespace(data="KITTI", image_size=(1, 1)))


def test_vfanet_state_dict_matches_reference_checkpoint_layout():
    """Names, shapes and dtypes of every entry of the reference VFANet's state_dict (fixture generated from the
    reference by tests/golden/make_state_keys.py), so that its checkpoints load key-for-key."""
    import json
    from types import SimpleNamespace
    from conftest import golden_path
    from vfa_amd.vfanet import VFANet
    ref = json.load(open(golden_path("vfanet_state_keys.json")))
    for key, (base, mode) in {"resnet18_3D": ("resnet18", "3D"), "resnet34_2D": ("resnet34", "2D")}.items():
        m = VFANet(SimpleNamespace(data="MultiviewC", image_size=(720, 1280)), base=base, mode=mode)
        mine = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
        assert mine == ref[key]
