"""Reference-generated fixtures for the PRODUCER of the path: a real ``VFANet.forward`` of the reference
(/root/reference/vfa/model/vfanet.py:56-82) on tiny images, with what enters and leaves the lateral branch captured by hooks:

    feat{8,16,32}     the trunk's outputs (N, 128 / 256 / 512, h, w)  -- inputs of lat8/16/32             vfanet.py:62
    latw/latb{s}      lat{s}.weight (256, K), lat{s}.bias                                                  vfanet.py:37-39
    gnw/gnb{s}        bn{s}.weight / .bias of nn.GroupNorm(16, 256), randomised (the default 1 / 0 would not test the affine)
    lat{s}            relu(bn(lat(feat)))  (N, 256, h, w)  -- what enters vfa8/16/32                       vfanet.py:72-74
    weight/bias{s}    collapse parameters; calibs, grid; ortho = the summed map entering ``fuse``         vfanet.py:76-82, 131

Nothing of the reference is copied: it is imported from /root/reference in THIS container (the module stand-ins for cv2 /
torchvision live in make_golden.py), run, and only inputs and outputs are saved.

    python tests/golden/make_laterals.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (module stand-ins, sys.path, the reference's make_grid)
import torch  # noqa: E402

from vfa_amd.synthetic import ring_cameras  # noqa: E402


def lateral_case(fname, data, image_size, world_size, cube_size, grid_height, cube_LW, calibs, img_hw, seed):
    from vfa.model.vfanet import VFANet as RefVFANet
    torch.manual_seed(seed)
    args = types.SimpleNamespace(data=data, image_size=tuple(image_size))
    net = RefVFANet(args, grid_height=grid_height, cube_size=cube_size, mode="2D", pretrained=False).eval()
    with torch.no_grad():
        for s in (8, 16, 32):
            gn = getattr(net, f"bn{s}")
            gn.weight.uniform_(0.5, 1.5)
            gn.bias.uniform_(-0.3, 0.3)
            getattr(net, f"vfa{s}").collapse.weight.mul_(3.0)  # (so that the ReLU behind `collapse` cuts a share of the outputs)
    N = calibs.shape[0]
    images = torch.rand(N, 3, *img_hw)
    grid = mg.ref_make_grid(world_size=world_size, cube_LW=list(cube_LW), dataset=data).unsqueeze(0)
    feats, lats = {8: [], 16: [], 32: []}, {8: [], 16: [], 32: []}
    hooks, cap = [], {}
    for s in (8, 16, 32):
        hooks.append(getattr(net, f"lat{s}").register_forward_pre_hook(lambda m, i, s=s: feats[s].append(i[0].detach().clone())))
        hooks.append(getattr(net, f"vfa{s}").register_forward_pre_hook(lambda m, i, s=s: lats[s].append(i[0].detach().clone())))
    hooks.append(net.fuse.register_forward_pre_hook(lambda m, i: cap.__setitem__("ortho", i[0].detach().clone())))
    with torch.no_grad():
        net(images, calibs, grid)
    for h in hooks:
        h.remove()
    sd = {}
    for s in (8, 16, 32):
        lat, gn, m = getattr(net, f"lat{s}"), getattr(net, f"bn{s}"), getattr(net, f"vfa{s}")
        sd[f"feat{s}"] = torch.cat(feats[s], 0).numpy()
        sd[f"latw{s}"] = lat.weight.detach().reshape(256, -1).numpy()
        sd[f"latb{s}"] = lat.bias.detach().numpy()
        sd[f"gnw{s}"] = gn.weight.detach().numpy()
        sd[f"gnb{s}"] = gn.bias.detach().numpy()
        sd[f"lat{s}"] = torch.cat(lats[s], 0).numpy()
        sd[f"weight{s}"] = m.collapse.weight.detach().numpy()
        sd[f"bias{s}"] = m.collapse.bias.detach().numpy()
    np.savez_compressed(os.path.join(HERE, fname), data=data, image_size=np.array(image_size),
                        cube_size=np.array(cube_size, dtype=np.float64), grid_height=np.array(grid_height),
                        gn_eps=np.array(net.bn8.eps), calibs=calibs.numpy(), grid=grid[0].numpy(),
                        ortho=cap["ortho"].contiguous()[0].numpy(), seed=np.array(seed), **sd)
    print(f"{fname}: feat8 {sd['feat8'].shape} lat8 {sd['lat8'].shape} ortho {tuple(cap['ortho'].shape)} "
          f"max {cap['ortho'].abs().max():.4f}, positive share {(cap['ortho'] > 0).float().mean():.2f}")


def main():
    # five z-layers (K = 1280), 3 cameras: images 96 x 160 -> maps 12 x 20, 6 x 10, 3 x 5 (odd width at stride 32: the tail paths)
    lateral_case("laterals_mc.npz", "MultiviewC", (720, 1280), (3900, 3900), (25, 25, 32), 160, (390, 325),
                 ring_cameras(3, (1950.0, 1950.0, 0.0), 2800.0, 600.0, 900.0, (1280, 720)), (96, 160), 51)
    # one z-layer (K = 256), 2 cameras, maps 13 x 19 (136 = 128 + 8 pixels... a partial 128-pixel tile and a partial wave)
    lateral_case("laterals_mc_nl1.npz", "MultiviewC", (720, 1280), (3750, 3750), (18.75, 18.75, 160), 160, (150, 125),
                 ring_cameras(2, (1875.0, 1875.0, 0.0), 2700.0, 600.0, 900.0, (1280, 720)), (104, 152), 52)


if __name__ == "__main__":
    main()
