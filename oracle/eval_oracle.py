"""CPU restatement of the consumer-side kernels -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/ may import it).

``sort_vertices``: a sequential numpy / Python restatement of the reference's CUDA kernel
(``/root/reference/vfa/evaluation/pyeval/cuda_op/sort_vert_kernel.cu:15-134``), polygon by polygon, with the kernel's operand
types (float32 products, double constants).  The reference kernel is CUDA-only and its extension cannot be built in this image
(it needs nvcc and ATen's CUDA headers), and the reference ships no test vectors for it, so the restatement is PINNED AT THE
KERNEL'S CALL SITE instead: tests/golden/make_golden.py --iou runs the reference's rotated-box IoU (IoU.py:139-221, its own
torch code on CPU) with this function standing in for the extension, and every overlap area equals an independent float64
polygon clipper's (tests/golden/iou_pairs.npz, tests/test_eval_ops.py).  Bit-level agreement with the CUDA binary on inputs that
pipeline never produces (random vertex clouds) remains unpinned.

``bev_nms``: numpy restatement of ``ObjectEncoder.nms(torch.sigmoid(h))`` (``vfa/data/encoder.py:230-232``), pinned by the decode
fixtures generated from the reference (tests/golden/decode_*.npz).
"""
import numpy as np

F = np.float32
EPS = 1e-8


def _before(x1, y1, x2, y2):
    if float(abs(F(x1 - x2))) < EPS and float(abs(F(y2 - y1))) < EPS:
        return False
    if y1 > 0 and y2 < 0:
        return True
    if y1 < 0 and y2 > 0:
        return False
    n1 = F(float(F(F(x1 * x1) + F(y1 * y1))) + EPS)
    n2 = F(float(F(F(x2 * x2) + F(y2 * y2))) + EPS)
    d = F(F(F(abs(x1)) * x1) / n1) - F(F(F(abs(x2)) * x2) / n2)
    d = F(d)
    if y1 > 0 and y2 > 0:
        return float(d) > EPS
    if y1 < 0 and y2 < 0:
        return float(d) < EPS
    return False  # a y of exactly 0: the reference falls off the end of the function


def sort_vertices(vertices, mask, num_valid):
    """vertices (b,n,m,2) f32, mask (b,n,m) bool, num_valid (b,n) -> (b,n,9) int32."""
    vertices = np.asarray(vertices, dtype=np.float32)
    b, n, m, _ = vertices.shape
    out = np.zeros((b, n, 9), np.int32)
    for bi in range(b):
        for i in range(n):
            v, mk, nv = vertices[bi, i], mask[bi, i], int(num_valid[bi, i])
            pad = m - 1
            for j in range(8, m):
                if not mk[j]:
                    pad = j
                    break
            if nv < 3:
                out[bi, i] = pad
                continue
            order = [pad] * 9
            for j in range(min(nv, 8)):
                x_min, y_min, take = F(1.0), F(-EPS), 0
                for k in range(m):
                    x, y = v[k]
                    if mk[k] and _before(x, y, x_min, y_min) and (j == 0 or _before(v[order[j - 1]][0], v[order[j - 1]][1], x, y)):
                        x_min, y_min, take = x, y, k
                order[j] = take
            order[min(nv, 8)] = order[0]
            if nv == 8:
                counter = sum(1 for j in range(4) for k in range(4, 8) if order[k] == order[j])
                if counter == 4:
                    order[4] = order[0]
                    for j in range(5, 9):
                        order[j] = pad
            out[bi, i] = order
    return out


def bev_nms(heatmap):
    """(L,W) logits -> sigmoid(h) kept where it equals its 5x5 max-pool (padding 2), else 0; float32 like the reference."""
    h = np.asarray(heatmap, dtype=np.float32)
    s = (1.0 / (1.0 + np.exp(-h.astype(np.float64)))).astype(np.float32)
    L, W = s.shape
    pad = np.full((L + 4, W + 4), -np.inf, np.float32)
    pad[2:-2, 2:-2] = s
    mx = np.max(np.stack([pad[i:i + L, j:j + W] for i in range(5) for j in range(5)]), axis=0)
    return np.where(mx == s, s, np.float32(0))
