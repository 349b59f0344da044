#!/usr/bin/env python3
"""How much of the geometry pass (side stream) is exposed beside the integral images: the bench frame with the geometry (a) on the
side stream as shipped, (b) taken from a workspace computed beforehand (not a product path: numbers only), (c) on the main stream."""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vfa_amd
from vfa_amd import ops, vfa_op, _lib
from vfa_amd.synthetic import make_workload

dev = torch.device("cuda:0")
wl = make_workload("multiviewc_200x200x1", channels=256, seed=0)
mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
n = wl["n_cam"]
feats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)
L, W = grid.shape[1:3]
out = torch.empty(L * W, 256, device=dev)


def timed(fn, reps=200, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize(); gc.enable()
    return e0.elapsed_time(e1) / reps * 1e3


with torch.no_grad():
    full = timed(lambda: vfa_op.fused_frame(mods, feats, calibs, grid, out=out))
    m0 = mods[0]
    zl, co = m0._kernel_geometry(dev)
    kind = vfa_op._conv_kind(m0.args)
    img_h, img_w = (float(v) for v in m0.args.image_size)
    weights = [m.layer_major_weight() for m in mods]
    sizes = [tuple(f.shape[-2:]) for f in feats]
    ws = ops.frame_records(calibs, grid, zl, co, kind, (img_w, img_h), sizes, weights=weights)
    biases = [m.collapse.bias for m in mods]

    def cached():
        integrals = ops.integral_images(feats)
        ops.pool_collapse(integrals, biases, ws, (L, W), out=out)
    nogeo = timed(cached)

    def serial():
        ops.frame_records(calibs, grid, zl, co, kind, (img_w, img_h), sizes, weights=weights, workspace=ws)
        integrals = ops.integral_images(feats)
        ops.pool_collapse(integrals, biases, ws, (L, W), out=out)
    ser = timed(serial)

    def nosplit():
        ops.frame_records(calibs, grid, zl, co, kind, (img_w, img_h), sizes, weights=None, workspace=ws)
    geo_ns = timed(nosplit)
    geo = timed(lambda: ops.frame_records(calibs, grid, zl, co, kind, (img_w, img_h), sizes, weights=weights, workspace=ws))
    integ = timed(lambda: ops.integral_images(feats))
    ints = ops.integral_images(feats)
    coll = timed(lambda: ops.pool_collapse(ints, biases, ws, (L, W), out=out))
print(f"frame as shipped {full:.1f} us | geometry precomputed {nogeo:.1f} us | geometry on the main stream {ser:.1f} us")
print(f"alone: geometry {geo:.1f} us (without the weight split {geo_ns:.1f}), integral images {integ:.1f} us, pool_collapse {coll:.1f} us")
