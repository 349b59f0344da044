#!/usr/bin/env python3
"""f3, the producer in front of the path: trunk outputs -> the three integral images, two ways:
  hand-written   vfa_lateral_conv_f32 (fp32 MFMA 1x1 conv, channels-last, GroupNorm statistics in the epilogue) per scale +
                 vfa_integral_images_hwc_f32 (affine + ReLU inside the row scan)                          [VFANet.lateral_integrals]
  library        MIOpen 1x1 conv + torch GroupNorm + ReLU per scale, then vfa_integral_images_f32        [VFANet.laterals]
on the bench frame's shapes (7 cameras, 720 x 1280 images: 90 x 160, 45 x 80, 23 x 40 maps with 128 / 256 / 512 channels).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vfa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
torch.manual_seed(0)
shapes = [(128, 90, 160), (256, 45, 80), (512, 23, 40)]
feats = [torch.randn(n, k, h, w, device=dev) for k, h, w in shapes]
convs = [torch.nn.Conv2d(k, 256, 1).to(dev) for k, _, _ in shapes]
gns = [torch.nn.GroupNorm(16, 256).to(dev) for _ in shapes]


def hand():
    ys, scs, shs = [], [], []
    for f, c, g in zip(feats, convs, gns):
        y, sc, sh = ops.lateral_conv(f, c.weight, c.bias, g.weight, g.bias, g.eps)
        ys.append(y), scs.append(sc), shs.append(sh)
    return ops.integral_images(ys, scs, shs, channels_last=True)


def library():
    return ops.integral_images([F.relu(g(c(f))) for f, c, g in zip(feats, convs, gns)])


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with torch.no_grad():
    a, b = hand(), library()
    for x, y in zip(a, b):
        print(f"  integral {tuple(x.shape)}: max |hand - library| / max = {((x - y).abs().max() / y.abs().max()).item():.2e}")
    print(f"{n} cameras: hand-written {timed(hand):.3f} ms, library {timed(library):.3f} ms per frame")
    with ops.KernelTimer() as kt:
        hand()
        torch.cuda.synchronize()
    for k, v in kt.summary().items():
        print(f"  {k}: {v['launches']} launches, {v['ms'] * 1e3:.1f} us")
        if len(v["by_tag"]) > 1:
            for tag, t in v["by_tag"].items():
                print(f"      {tag}: {t['ms'] * 1e3:.1f} us")
    flops = sum(2.0 * n * h * w * k * 256 for k, h, w in shapes)
    bytes_ = sum(4.0 * n * h * w * (k + 256) for k, h, w in shapes)
    print(f"  convolutions: {flops / 1e9:.1f} GFLOP (fp32 matrix pipe 155 TFLOP/s: {flops / 155e12 * 1e6:.0f} us), "
          f"{bytes_ / 1e6:.0f} MB in + out ({bytes_ / 8e12 * 1e6:.0f} us at 8 TB/s)")
