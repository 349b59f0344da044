#!/usr/bin/env python3
"""Record the reference VFANet's state_dict layout (names, shapes, dtypes -- data, not source) as a fixture.
Runs only in the build container (imports /root/reference with the stubs of make_golden.py)."""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.modules["cv2"] = types.ModuleType("cv2")
_tv, _tvd, _tvv, _tvt = (types.ModuleType(n) for n in (
    "torchvision", "torchvision.datasets", "torchvision.datasets.vision", "torchvision.transforms"))


class _VD:
    def __init__(self, root=None, transform=None, **kw):
        pass


_tvd.VisionDataset = _tvv.VisionDataset = _VD
_tvt.ToTensor = lambda: None
_tv.datasets, _tv.transforms = _tvd, _tvt
sys.modules.update({m.__name__: m for m in (_tv, _tvd, _tvv, _tvt)})
import matplotlib  # noqa: E402

matplotlib.use("agg")
from vfa.model.vfanet import VFANet  # noqa: E402

out = {}
for base, mode in (("resnet18", "3D"), ("resnet34", "2D")):
    args = types.SimpleNamespace(data="MultiviewC", image_size=(720, 1280))
    m = VFANet(args, base=base, grid_height=160, cube_size=(25, 25, 32), angle_range=360, mode=mode, pretrained=False)
    out[f"{base}_{mode}"] = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
json.dump(out, open(os.path.join(HERE, "vfanet_state_keys.json"), "w"))
print({k: len(v) for k, v in out.items()})
