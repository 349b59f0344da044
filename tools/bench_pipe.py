#!/usr/bin/env python3
"""Diagnostics of the pipelined frame kernel (vfa_pipe.hip) on a named workload: launch time of `vfa_pipe_collapse_relu_sum_f32`,
of the geometry pass and of the serial fused kernel beside it (single-layer grids), phase ablations (VFA_FLAG_DEBUG: 1 no
window DMA, 2 no pooling, 4 no MFMA) and per-wave cycle stamps (128 | wave << 8).  Numbers only; ablated results are meaningless.

    python tools/bench_pipe.py [workload] [--stamps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import vfa_amd  # noqa: E402
from vfa_amd import _lib, ops  # noqa: E402
from vfa_amd.synthetic import make_workload  # noqa: E402

name = next((a for a in sys.argv[1:] if not a.startswith("--")), "multiviewc_200x200x1")
terms = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--terms=")), "0"))  # 0 (= 3), 3, 4, 6 products
stamps = "--stamps" in sys.argv
stamp_mask = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--stamp-mask=")), "0"))  # ablation bits under the stamps
dev = torch.device("cuda:0")
cams = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--cams=")), "0"))  # the first `cams` cameras of the rig only
wl = make_workload(name, channels=256, seed=0, **({"n_cam": cams} if cams else {}))
n = wl["n_cam"]
torch.manual_seed(0)
mods = [vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev) for _ in range(3)]
nl = mods[0].num_grid_layer
lats = [torch.cat([wl["features"][c][s] for c in range(n)]).to(dev) for s in range(3)]
calibs, grid = wl["calibs"].to(dev), wl["grid"].to(dev)
L, W = grid.shape[1:3]
rows = L
while rows > 4 and ops.pipe_workspace_bytes(n, rows, W, nl, 3) > (3 << 30):
    rows = max(4, ((rows + 1) // 2 + 3) // 4 * 4)
if rows < L:
    print(f"(frame of {L} rows measured on its first band of {rows} rows)")
    L = rows
    grid = grid[:, :L].contiguous()
zl, co = mods[0]._kernel_geometry(dev)
kind = _lib.CONV_KIND[wl["args"].data]
img_wh = wl["args"].image_size[::-1]
sizes = [tuple(l.shape[-2:]) for l in lats]


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
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
    integrals = ops.integral_images(lats)
    weights = [m.collapse.weight for m in mods]
    biases = [m.collapse.bias for m in mods]
    ws = ops.pipe_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights, terms=terms)
    t_rec = timed(lambda: ops.pipe_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights, workspace=ws, terms=terms))
    t_box = timed(lambda: ops.pipe_records(calibs, grid, zl, co, kind, img_wh, sizes, workspace=ws, cuts=False, terms=terms))
    ops.pipe_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights, workspace=ws, terms=terms)
    t_int = timed(lambda: ops.integral_images(lats))
    out = torch.empty(L * W, 256, device=dev)
    # header statistics
    lay = ops.pipe_workspace_layout(n, L, W, nl, 3)
    tiles = lay["tiles_l"] * lay["tiles_w"]
    items_live = 0
    for s in range(3):
        hdr = ws[lay["hdrs"][s]:lay["hdrs"][s] + tiles * nl * n * 32].cpu().numpy().view(np.uint32).reshape(-1, 8)
        live = (hdr[:, 0] & 1) == 1
        direct = live & (((hdr[:, 0] >> 1) & 1) == 1)
        slots = hdr[:, 1][live & ~direct]
        items_live += int(live.sum())
        print(f"  scale {s}: items {hdr.shape[0]}, live {int(live.sum())}, pooled from L2 {int(direct.sum())}, "
              f"window slots mean {slots.mean() if slots.size else 0:.1f} max {slots.max() if slots.size else 0}")
    print(f"{name}: {n} views, {L}x{W}x{nl}, {tiles} tiles | geometry {t_rec:.1f} us (boxes {t_box:.1f}) | integral images {t_int:.1f} us | "
          f"workspace {ws.numel() / 1e6:.0f} MB")
    t_full = timed(lambda: ops.pipe_collapse(integrals, biases, ws, (L, W), nl, out=out, terms=terms), reps=20, warm=10)
    flops = items_live * 3 * 2 * 32 * 256 * 256
    print(f"  pipe_collapse           {t_full:9.1f} us   {items_live} live 32x256x256 products: {flops / t_full / 1e6:7.1f} TFLOP/s bf16 issued-equivalent "
          f"({flops / t_full / 1e6 / 2500:.3f} of the dense peak), {t_full * 1e-6 * 2.4e9 * 256 / max(items_live, 1):.0f} CU-cycles @2.4GHz per product")
    if nl == 1:
        ws_old = ops.frame_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=[m.layer_major_weight().contiguous() for m in mods])
        t_old = timed(lambda: ops.pool_collapse(integrals, biases, ws_old, (L, W), out=out), reps=20, warm=10)
        print(f"  serial fused kernel     {t_old:9.1f} us")
    if terms not in (0, 2):  # (the diagnostic build exists for the default arithmetic only: the entry point refuses debug flags with other terms)
        ops.pipe_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights, workspace=ws, terms=0)
    for mask, label in ((1, "no window DMA"), (2, "no pooling"), (4, "no MFMA"), (6, "no pooling, no MFMA"), (7, "skeleton only"), (64, "loop, tables, barrier only"), (32, "diagnostic build"), (8, "no setprio"), (16, "matrix waves prio 2")):
        us = timed(lambda: ops.pipe_collapse(integrals, biases, ws, (L, W), nl, out=out, debug=mask))
        print(f"  pipe_collapse [{label:>20}] {us:9.1f} us")
    if stamps:
        names = ["step head", "wait for W", "pool / DMA issue", "multiply", "generator + wait for DMA", "tile finish", "barrier", "steps"]
        for wave in range(16):
            ops.pipe_collapse(integrals, biases, ws, (L, W), nl, out=out, debug=128 | stamp_mask | (wave << 8))
            torch.cuda.synchronize()
            d = ws[lay["diag"]:lay["diag"] + 512 * 64].cpu().numpy().view(np.uint64).reshape(512, 8).astype(np.float64)
            d = d[d[:, 7] > 0]
            steps = d[:, 7]
            per = d[:, :7].sum(0) / steps.sum()
            role = "pool" if wave >= 8 else "matrix"
            pool_names = ["step head", "window requests + pooling", "waiting for the requests to land", "", "", "tile finish", "barrier"]
            print(f"  wave {wave:2d} ({role}): cycles per step " + ", ".join(f"{(names if wave < 8 else pool_names)[k]} {per[k]:.0f}" for k in range(7) if wave < 8 or k not in (3, 4))
                  + f" | total {per.sum():.0f}, steps per workgroup {steps.mean():.0f} (max/mean {steps.max() / steps.mean():.3f}), "
                  f"cycles per workgroup max/mean {d[:, :7].sum(1).max() / d[:, :7].sum(1).mean():.3f}")

    if "--barrier" in sys.argv:
        # diagnostic 32: cycles a wave waits at the step barrier, by the position of the step in its phase (i & 7)
        for wave in range(16):
            ops.pipe_collapse(integrals, biases, ws, (L, W), nl, out=out, debug=128 | 32 | (wave << 8))
            torch.cuda.synchronize()
            d = ws[lay["diag"]:lay["diag"] + 512 * 64].cpu().numpy().view(np.uint64).reshape(512, 8).astype(np.float64)
            d = d[d.sum(1) > 0]
            print(f"  wave {wave:2d}: barrier wait per workgroup by step position 0..7 (thousand cycles): "
                  + " ".join(f"{v / 1e3:.0f}" for v in d.mean(0)) + f" | total {d.sum(1).mean() / 1e3:.0f}")

    if "--groups" in sys.argv:
        # live views per (tile, scale), groups by size, steps by kind: what the step sequence of this frame is made of
        host = ws.cpu().numpy()
        live = [host[lay["live"][s]:lay["live"][s] + 4 * tiles].view(np.uint32) for s in range(3)]
        hist = np.zeros(33, np.int64)
        by_nj = np.zeros(5, np.int64)
        for s in range(3):
            pc = np.array([bin(int(m)).count("1") for m in live[s]])
            hist += np.bincount(pc, minlength=33)[:33]
            for c in pc:
                full, rest = divmod(int(c), 4)
                by_nj[4] += full
                if rest:
                    by_nj[rest] += 1
        steps_full = by_nj[4] * 8 + by_nj[3] * 4
        steps_half = by_nj[3] * 4
        steps_small_full = by_nj[2] * 4
        steps_small_half = by_nj[1] * 4
        steps_empty = (by_nj[1] + by_nj[2]) * 4
        print("  live views per (tile, scale): " + " ".join(f"{k}:{int(v)}" for k, v in enumerate(hist) if v))
        print(f"  groups by views: 1:{by_nj[1]} 2:{by_nj[2]} 3:{by_nj[3]} 4:{by_nj[4]} | steps per layer: full {steps_full}, half (set 1 of three views) {steps_half}, "
              f"set 0 of a two-view group {steps_small_full}, of a one-view group {steps_small_half}, empty {steps_empty} "
              f"(empty share {steps_empty / max(1, steps_full + steps_half + steps_small_full + steps_small_half + steps_empty):.3f})")

    if "--fit" in sys.argv:
        # Per-workgroup cycles against what the workgroup had to do: the constants of the work-cut cost model (vfa_pipe_seq.h).
        ops.pipe_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights, workspace=ws, terms=terms)
        ops.pipe_collapse(integrals, biases, ws, (L, W), nl, out=out, debug=128 | (8 << 8))
        torch.cuda.synchronize()
        host = ws.cpu().numpy()
        d = host[lay["diag"]:lay["diag"] + 512 * 64].view(np.uint64).reshape(512, 8).astype(np.float64)
        K = lay["n_chunks"]
        start = host[lay["chunks"]:lay["chunks"] + 4 * (K + 1)].view(np.int32)
        rank = host[lay["ranks"]:lay["ranks"] + 4 * (K + 1)].view(np.int32)
        live = [host[lay["live"][s]:lay["live"][s] + 4 * tiles].view(np.uint32) for s in range(3)]
        hdrs = [host[lay["hdrs"][s]:lay["hdrs"][s] + tiles * nl * n * 32].view(np.uint32).reshape(tiles, nl, n, 8) for s in range(3)]
        nblk = int((d[:, 7] > 0).sum() + 7) // 8 * 8
        nblk = min(max(nblk, 8), 256)
        rows = []
        for blk in range(nblk):
            lb = (blk & 7) * (nblk // 8) + (blk >> 3)
            c0, c1 = K * lb // nblk, K * (lb + 1) // nblk
            tb, kb, te, ke = int(start[c0]), int(rank[c0]), int(start[c1]), int(rank[c1])
            real = half = dummy = glob = ntile = shared = groups = fills = 0
            by_scale = [0, 0, 0]
            masked = 0
            for t in range(tb, min(te + (1 if ke > 0 else 0), tiles)):
                k = 0
                mine = False
                for s in range(3):
                    views = [v for v in range(n) if (int(live[s][t]) >> v) & 1]
                    for g0 in range(0, len(views), 4):
                        grp = views[g0:g0 + 4]
                        ok = not (t == tb and k < kb) and not (t == te and k >= ke)
                        k += 1
                        if not ok:
                            continue
                        mine = True
                        groups += 1
                        for layer in range(nl):
                            for st in range(2):
                                sub = grp[2 * st:2 * st + 2]
                                lv = [v for v in sub if hdrs[s][t, layer, v, 0] & 1]
                                fills += 4 * sum(int(hdrs[s][t, layer, v, 1] + 3) // 4 for v in lv if not (hdrs[s][t, layer, v, 0] & 2))
                                if lv:
                                    by_scale[s] += 4 * len(sub)
                                    masked += 4 * (len(sub) - len(lv))
                                if lv and len(sub) == 1:
                                    half += 4  # (a set with one sub-tile: the matrix waves multiply one row block)
                                elif lv:
                                    real += 4
                                    glob += 4 * sum(1 for v in lv if hdrs[s][t, layer, v, 0] & 2)
                                else:
                                    dummy += 4
                if mine:
                    ntile += 1
                    if (t == tb and kb > 0) or (t == te and ke > 0):
                        shared += 1
            rows.append((blk, real, dummy, glob, ntile, shared, groups, d[blk, :7].sum(), half, fills, by_scale[1], by_scale[2], masked))
        A = np.array([[r[1], r[2], r[3], r[4], r[5], r[6], r[8], r[9], r[10], r[11], r[12]] for r in rows if r[7] > 0], np.float64)
        y = np.array([r[7] for r in rows if r[7] > 0])
        coef, *_ = np.linalg.lstsq(A, y, rcond=None)
        pred = A @ coef
        print("  fit (cycles): full step %.0f, empty step %.0f, + per sub-tile step pooled from L2 %.0f, per tile %.0f, per shared tile %.0f, per group %.0f, half step %.0f, per window request (4 slots) %.0f; per sub-tile step at stride 16 %+.0f, at stride 32 %+.0f, per sub-tile step without a live box in the layer %+.0f"
              % tuple(coef))
        blks = np.array([r[0] for r in rows if r[7] > 0])
        res = (y - pred) / y.mean()
        print("  residual by XCD (mean, max): " + "  ".join(f"{x}: {res[(blks & 7) == x].mean():+.3f} {res[(blks & 7) == x].max():+.3f}" for x in range(8)))
        worst = np.argsort(-res)[:6]
        print("  slowest against the model: " + ", ".join(f"block {blks[i]} (xcd {blks[i] & 7}, rank {blks[i] >> 3}) {res[i]:+.3f}" for i in worst))
        print(f"  measured max/mean {y.max() / y.mean():.3f}; residual rms {np.sqrt(np.mean((pred - y) ** 2)) / y.mean():.3f} of the mean; "
              f"if cut by this model: max/mean of the residual-corrected load {1 + (y - pred).max() / y.mean():.3f}")
