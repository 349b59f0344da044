#!/usr/bin/env python3
"""Diagnostics of the fused frame kernel on the bench workload: launch time of `vfa_pool_collapse_relu_sum_f32` and of the
geometry pass, phase ablations (VFA_FLAG_DEBUG) and in-kernel cycle stamps.  Numbers only; ablated results are meaningless."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vfa_amd  # noqa: E402
from vfa_amd import _lib, ops  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "multiviewc_200x200x1"
dev = torch.device("cuda:0")
n_cam_arg = int(sys.argv[2]) if len(sys.argv) > 2 else None
wl = make_workload(name, channels=256, seed=0, **({"n_cam": n_cam_arg} if n_cam_arg else {}))
n = wl["n_cam"]
torch.manual_seed(0)
mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)
L, W = grid.shape[1:3]
zl, co = mods[0]._kernel_geometry(dev)
kind = _lib.CONV_KIND[wl["args"].data]
img_wh = wl["args"].image_size[::-1]
sizes = [tuple(l.shape[-2:]) for l in lats]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


with torch.no_grad():
    integrals = ops.integral_images(lats)  # (with the feature statistics of the fp16 split: without them every fused call adds a pass)
    weights = [m.layer_major_weight().contiguous() for m in mods]
    biases = [m.collapse.bias for m in mods]
    ws = ops.frame_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights)
    print(f"{name}: frame_records {timed(lambda: ops.frame_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights, workspace=ws)):.1f} us")
    out = torch.empty(L * W, 256, device=dev)
    vox = torch.empty(n, L * W, 256, device=dev)
    cal12, gflat = calibs.reshape(n, 12).contiguous(), grid.reshape(-1, 3).contiguous()
    for k in range(3):
        tw = timed(lambda: ops.pool_windows(integrals[k], ws, (L, W), 3, k, out=vox))
        tt = timed(lambda: ops.project_gather(integrals[k], cal12, gflat, zl, co, kind, img_wh, out=vox, kernel="tap_cache"))
        tc = timed(lambda: ops.collapse_relu_sum(vox, weights[k], biases[k], out=out))
        alg = n * 256 * sizes[k][0] * sizes[k][1] * 4 + n * L * W * 1024 + L * W * 12 + n * 48
        print(f"  scale {k}: pool_windows {tw:6.1f} us ({alg / tw / 1e3:6.0f} GB/s algorithmic, {alg / tw / 8e6:.2f} of 8 TB/s) | "
              f"round-1 tap-cache kernel {tt:6.1f} us | collapse_relu_sum {tc:6.1f} us")
    # header statistics
    host = ws.cpu().numpy()
    lay = ops.frame_workspace_layout(n, L, W, 3)
    tiles = lay["tiles_l"] * lay["tiles_w"]
    for s in range(3):
        off = lay["hdrs"][s]
        hdr = host[off:off + n * tiles * 32].view(np.uint32).reshape(n * tiles, 8)
        live = hdr[:, 0] & 1
        direct = (hdr[:, 0] >> 1) & 1
        slots = hdr[:, 1][(live == 1) & (direct == 0)]
        print(f"  scale {s}: items {n * tiles}, live {int(live.sum())}, direct {int((direct & live).sum())}, slots mean {slots.mean():.1f} max {slots.max()}")
    diag_off = lay["diag"]
    for mask, label in ((0, "full"), (128, "full + stamps"), (384, "stamps of wave 1"), (640, "stamps of wave 2"), (896, "stamps of wave 3"), (1152, "stamps of wave 4"), (1408, "stamps of wave 5"), (1664, "stamps of wave 6"), (1920, "stamps of wave 7"), (129, "no fills + stamps"), (130, "no pool + stamps"), (132, "no mfma + stamps"), (136, "one W + stamps"), (160, "no masked-item term + st"), (168, "no masked term, one W"), (144, "no records + stamps"), (152, "one W, no records + st"), (153, "one W, no rec, no fills"), (1, "no fills"), (2, "no pool"), (4, "no mfma"), (3, "no fills, no pool"),
                        (6, "no pool, no mfma"), (5, "no fills, no mfma"), (7, "skeleton only"), (64, "direct items only")):
        us = timed(lambda: ops.pool_collapse(integrals, biases, ws, (L, W), out=out, debug=mask, absmax=integrals.absmax))
        print(f"  pool_collapse [{label:>18}] {us:8.1f} us")
        if mask & 128:
            torch.cuda.synchronize()
            d = ws[diag_off:diag_off + 256 * 64].cpu().numpy().view(np.uint64).reshape(256, 8).astype(np.float64)
            d = d[d[:, 7] > 0]
            items = d[:, 7]
            names = ["tail: epilogue + seek", "wait for the window (vmcnt)", "barrier + header", "tile store + W loads + pool",
                     "wait for W + barrier", "issue fills + records + header", "MFMA phase"]
            print(f"    stamps of wave {mask >> 8}, cycles per item (mean over {len(d)} workgroups, {items.mean():.1f} items each):")
            for k, nm in enumerate(names):
                print(f"      {nm:28s} {np.mean(d[:, k] / items):9.0f}")
            print(f"      {'sum':28s} {np.mean(d[:, :7].sum(1) / items):9.0f}")
            tot = d[:, :7].sum(1)
            print(f"    per-workgroup total cycles: min {tot.min():.3g}  mean {tot.mean():.3g}  p90 {np.percentile(tot, 90):.3g}  max {tot.max():.3g}"
                  f"  (max / mean {tot.max() / tot.mean():.3f}); items per workgroup min {items.min():.0f} max {items.max():.0f}")
            if mask == 128 and len(d) == 256:
                # what the per-workgroup time depends on: items, their window slots, row items, scale changes, tiles (least squares)
                nblk, K = 256, lay["n_chunks"]
                cs = host[lay["chunks"]:lay["chunks"] + 4 * (K + 1)].view(np.int32)
                cr = host[lay["ranks"]:lay["ranks"] + 4 * (K + 1)].view(np.int32)
                hdrs = [host[lay["hdrs"][s]:lay["hdrs"][s] + n * tiles * 32].view(np.uint32).reshape(n, tiles, 8) for s in range(3)]
                ovf = [host[lay["overflow"][s]:lay["overflow"][s] + 4 * tiles].view(np.uint32) for s in range(3)]
                seq = []  # (tile, scale, view, slots, rows) in kernel order
                for t in range(tiles):
                    for s in range(3):
                        for v in range(n):
                            h = hdrs[s][v, t]
                            if (h[0] & 1) and not ((ovf[s][t] >> v) & 1):
                                seq.append((t, s, v, 0 if (h[0] & 2) else int(h[1]), 1 if (h[0] & 4) else 0))
                seq = np.array(seq)
                first = np.searchsorted(seq[:, 0], np.arange(tiles + 1))
                feats = []
                feats2 = []
                for wg in range(nblk):
                    c0, c1 = K * wg // nblk, K * (wg + 1) // nblk
                    i0, i1 = first[min(cs[c0], tiles)] + cr[c0], first[min(cs[c1], tiles)] + cr[c1]
                    part = seq[i0:i1]
                    changes = 1 + int(np.count_nonzero(np.diff(part[:, 0] * 4 + part[:, 1]))) if len(part) else 0
                    feats.append((len(part), part[:, 3].sum(), part[:, 4].sum(), changes, len(np.unique(part[:, 0])), 1.0))
                    feats2.append((np.count_nonzero(part[:, 1] == 0), np.count_nonzero(part[:, 1] == 1), np.count_nonzero(part[:, 1] == 2),
                                   part[:, 3].sum(), np.maximum(part[:, 3] - 64, 0).sum(), part[:, 4].sum(), changes, len(np.unique(part[:, 0])), 1.0))
                feats = np.array(feats, dtype=np.float64)
                # diag rows are indexed by blockIdx.x; the kernel maps it to a range index XCD-contiguously
                per = (nblk + 7) // 8
                lb = np.array([(b % 8) * per + b // 8 for b in range(nblk)])
                y = np.zeros(nblk)
                full = ws[diag_off:diag_off + 256 * 64].cpu().numpy().view(np.uint64).reshape(256, 8).astype(np.float64)
                y[lb] = full[:, :7].sum(1)
                f2 = np.array(feats2, dtype=np.float64)
                c2, *_ = np.linalg.lstsq(f2, y, rcond=None)
                r2 = y - f2 @ c2
                print("    richer model (items of scale 0 / 1 / 2, slots, slots above 64, row items, scale changes, tiles, 1):", np.array2string(np.round(c2, 1), max_line_width=1000),
                      f"residual std {r2.std():.3g}, max {r2.max():.3g} of mean {y.mean():.3g}")
                coef, *_ = np.linalg.lstsq(feats, y, rcond=None)
                res = y - feats @ coef
                xcd_of = np.arange(nblk) // per  # (range index -> XCD: xcd_contiguous)
                print("    residual by XCD (mean, max, of the mean load): " + "  ".join(
                    f"{x}: {res[xcd_of == x].mean() / y.mean():+.3f} {res[xcd_of == x].max() / y.mean():+.3f}" for x in range(8))
                    + f" | max/mean {y.max() / y.mean():.3f}")
                print(f"    cycles ~ {coef[0]:.0f} x items + {coef[1]:.1f} x slots + {coef[2]:.0f} x row items + {coef[3]:.0f} x scale changes"
                      f" + {coef[4]:.0f} x tiles + {coef[5]:.0f};  residual std {res.std():.3g} of mean {y.mean():.3g} (raw std {y.std():.3g})")
            order = np.argsort(tot)[-5:]
            print("    slowest workgroups (cycles, items, cycles/item):", [(int(tot[i]), int(items[i]), int(tot[i] / items[i])) for i in order])
