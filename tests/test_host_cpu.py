"""Host-side logic that runs without a GPU: the reference-named alias package, VFANet construction with the reference's
default arguments, and gradient synchronisation of camera-sharded training (gloo, world_size 2)."""
import os
import socket
import subprocess
import sys
import warnings
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import REPO


def _args():
    return SimpleNamespace(data="MultiviewC", image_size=(720, 1280))


def test_vfanet_builds_with_the_reference_default_arguments(tmp_path, monkeypatch):
    """reference train.py:90 defaults --pretrained to True and passes it straight to the constructor (:249-250)."""
    from vfa_amd.vfanet import PRETRAINED_ENV, VFANet
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "empty_home"))
    monkeypatch.delenv(PRETRAINED_ENV, raising=False)
    with pytest.raises(FileNotFoundError, match="VFA_AMD_PRETRAINED"):  # no network, no local file: loud, not a silent random start
        VFANet(_args(), grid_height=160, cube_size=(25, 25, 32), angle_range=360, mode="3D", pretrained=True)
    monkeypatch.setenv(PRETRAINED_ENV, "none")  # ... unless random initialisation is asked for by name
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        net = VFANet(_args(), grid_height=160, cube_size=(25, 25, 32), angle_range=360, mode="3D", pretrained=True)
    assert any("random" in str(x.message) for x in w)
    assert net.vfa8.collapse.weight.shape == (256, 1280)
    # a local checkpoint is loaded by key intersection like the reference's _load_pretrained (resnet.py:170-175)
    ref = {k: torch.full_like(v, 0.5) for k, v in net.base.state_dict().items() if k.startswith("layer1.0")}
    ref["fc.weight"] = torch.zeros(1000, 512)                      # not in the trunk: ignored
    ref["bn1.running_mean"] = torch.zeros(64)                      # BatchNorm statistics: no GroupNorm counterpart
    torch.save(ref, tmp_path / "resnet18-local.pth")
    monkeypatch.setenv(PRETRAINED_ENV, str(tmp_path))
    net2 = VFANet(_args(), pretrained=True)
    assert torch.all(net2.base.layer1[0].conv1.weight == 0.5) and torch.all(net2.base.layer1[0].bn1.weight == 0.5)
    assert not torch.all(net2.base.layer2[0].conv1.weight == 0.5)


def test_reference_named_alias_package_resolves_to_the_build():
    """`from vfa.model.vfanet import VFANet` (reference train.py / evaluate.py) with compat/ in front of the path."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, "compat"), REPO]))
    env.pop("VFA_REFERENCE_ROOT", None)
    code = ("import vfa.model.vfa_op as a, vfa.model.vfanet as b, vfa_amd;"
            "assert a.VFA is vfa_amd.VFA and b.VFANet.__module__ == 'vfa_amd.vfanet';"
            "assert a.EPSILON == 1e-6 and a.MAXIMUM_AREA_RATIO == 0.3; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/")
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr
    if os.path.isdir("/root/reference/vfa"):  # build container only: the REST of vfa.* is still the reference's
        env["PYTHONPATH"] += os.pathsep + "/root/reference"
        code = ("import vfa.model.loss as l, vfa.model.vfa_op as a, vfa_amd, os;"
                "assert l.__file__.startswith('/root/reference') and a.VFA is vfa_amd.VFA; print('ok')")
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/")
        assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr
        # the reference's IoU module binds its CUDA-only op to the HIP one through compat/sort_vertices.py, unedited
        code = ("import vfa.evaluation.pyeval.IoU as i, vfa_amd.eval_ops as e;"
                "assert i.__file__.startswith('/root/reference') and i.sort_v.__self__.__name__ == 'SortVertices';"
                "import sort_vertices as s; assert s.sort_vertices_forward is e.sort_vertices; print('ok')")
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/")
        assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_extension_module_alias_exports_the_reference_entry_point():
    """``import sort_vertices`` (reference cuda_op/cuda_ext.py:4) with compat/ on the path is the HIP op."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, "compat"), REPO]))
    code = "import sort_vertices as s, vfa_amd.eval_ops as e; assert s.sort_vertices_forward is e.sort_vertices; print('ok')"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/")
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


# ------------------------------------------------------------------------------------------------
# camera-sharded training: gradients of the parameters in front of the all-reduce must be SUMmed over ranks
# ------------------------------------------------------------------------------------------------
class _ToyNet(nn.Module):
    """Same wiring as VFANet.forward(distributed=True): per-camera pre-head modules (named like the real ones), a sum
    over cameras fused by `all_reduce_ortho`, replicated heads."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.base = nn.Linear(6, 8)
        self.vfa8 = nn.Linear(8, 4)
        self.fuse = nn.Linear(4, 3)

    def forward(self, cams, distributed):
        from vfa_amd.aggregate import all_reduce_ortho, camera_shard
        mine = camera_shard(cams.shape[0]) if distributed else list(range(cams.shape[0]))
        part = cams.new_zeros(5, 4)
        for c in mine:
            part = part + torch.relu(self.vfa8(torch.tanh(self.base(cams[c]))))
        if distributed:
            part = all_reduce_ortho(part if part.requires_grad else part.requires_grad_())
        return self.fuse(part).square().sum()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _grad_worker(rank, world, port, n_cam, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vfa_amd import all_reduce_prehead_grads
        cams = torch.randn(n_cam, 5, 6, generator=torch.Generator().manual_seed(1))
        single = _ToyNet()
        single(cams, distributed=False).backward()
        net = _ToyNet()
        net(cams, distributed=True).backward()
        if n_cam >= world:  # every rank holds some but not all cameras: partial before the reduction
            assert not torch.allclose(net.base.weight.grad, single.base.weight.grad)
        assert all_reduce_prehead_grads(net) == 4  # base.{weight,bias}, vfa8.{weight,bias}; the heads are not touched
        for (name, p), q in zip(net.named_parameters(), single.parameters()):
            torch.testing.assert_close(p.grad, q.grad, rtol=1e-5, atol=1e-6, msg=name)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cam", [3, 1])  # 1 camera on 2 ranks: rank 1 holds none and contributes zero gradients
def test_camera_sharded_gradients_match_single_process_world2(n_cam, tmp_path):
    mp.spawn(_grad_worker, args=(2, _free_port(), n_cam, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_prehead_grad_sync_is_a_no_op_without_a_process_group():
    from vfa_amd import all_reduce_prehead_grads
    assert all_reduce_prehead_grads(_ToyNet()) == 0
