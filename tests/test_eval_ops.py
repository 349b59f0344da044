"""Consumer-side kernels: the HIP ``sort_vertices`` against the CPU restatement of the reference's CUDA
kernel, and the BEV decode (sigmoid + 5x5 NMS kernel, top-k, gathers) against fixtures generated from the reference's own
``ObjectEncoder`` (tests/golden/decode_*.npz)."""
import numpy as np
import pytest
import torch

from conftest import golden_path


def _rect_polygons(rng, b, n):
    """Candidate vertices of rectangle x rectangle intersections the way the reference builds them (IoU.py:120-137): 8 box
    corners (masked by 'corner of one box inside the other') + 16 edge-edge intersection points, centred on the mean of the
    valid ones."""
    m = 24
    verts = np.zeros((b, n, m, 2), np.float32)
    mask = np.zeros((b, n, m), bool)
    for bi in range(b):
        for i in range(n):
            def box():
                c = rng.uniform(-1, 1, 2)
                w, h, a = rng.uniform(0.5, 3.0), rng.uniform(0.5, 3.0), rng.uniform(0, np.pi)
                base = np.array([[.5, .5], [-.5, .5], [-.5, -.5], [.5, -.5]]) * [w, h]
                R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
                return base @ R.T + c
            c1, c2 = box(), (box() if i % 7 else None)
            if c2 is None:
                c2 = c1.copy()  # identical boxes: the reference's 8-vertex corner case

            def inside(p, c):
                ab, ad, ap = c[1] - c[0], c[3] - c[0], p - c[0]
                return -1e-6 < ap @ ab < ab @ ab + 1e-6 and -1e-6 < ap @ ad < ad @ ad + 1e-6
            cand, ok = list(c1) + list(c2), [inside(p, c2) for p in c1] + [inside(p, c1) for p in c2]
            for e1 in range(4):
                for e2 in range(4):
                    p, r = c1[e1], c1[(e1 + 1) % 4] - c1[e1]
                    q, s = c2[e2], c2[(e2 + 1) % 4] - c2[e2]
                    den = r[0] * s[1] - r[1] * s[0]
                    hit = False
                    pt = np.zeros(2)
                    if abs(den) > 1e-9:
                        t = ((q - p)[0] * s[1] - (q - p)[1] * s[0]) / den
                        u = ((q - p)[0] * r[1] - (q - p)[1] * r[0]) / den
                        hit = 0 < t < 1 and 0 < u < 1
                        pt = p + t * r
                    cand.append(pt)
                    ok.append(hit)
            cand, ok = np.array(cand, np.float32), np.array(ok)
            if ok[8:].all():
                ok[-1] = False
            centre = cand[ok].mean(0) if ok.any() else np.zeros(2, np.float32)
            verts[bi, i] = (cand - centre) * ok[:, None]
            mask[bi, i] = ok
    return verts, mask, mask.sum(-1).astype(np.int32)


def test_sort_vertices_oracle_orders_anticlockwise():
    """The CPU restatement itself: on convex intersection polygons the picked vertices are in anticlockwise order, the
    first index closes the loop, the padding is an invalid intersection index."""
    from oracle import eval_oracle
    rng = np.random.default_rng(0)
    v, m, nv = _rect_polygons(rng, 1, 40)
    idx = eval_oracle.sort_vertices(v, m, nv)
    seen = 0
    for i in range(40):
        k = int(nv[0, i])
        if k < 3:
            assert (idx[0, i] == idx[0, i, 0]).all() and not m[0, i, idx[0, i, 0]]
            continue
        if k == 8 and len(set(idx[0, i, :4])) == 4 and idx[0, i, 4] == idx[0, i, 0]:
            continue  # identical boxes
        sel = idx[0, i, :k]
        assert idx[0, i, k] == sel[0] and m[0, i, sel].all() and len(set(sel.tolist())) == k
        ang = np.arctan2(v[0, i, sel, 1], v[0, i, sel, 0]) % (2 * np.pi)
        assert (np.diff(ang) > 0).all(), (i, ang)
        seen += 1
    assert seen >= 20


def _iou_fixture():
    d = np.load(golden_path("iou_pairs.npz"))
    n = len(d["num_valid"])
    return d, d["vertices"].reshape(1, n, 24, 2), d["mask"].reshape(1, n, 24), d["num_valid"].reshape(1, n)


def _overlap_from(idx, raw):
    """The reference's ``calculate_area`` (IoU.py:176-192) on its un-normalised vertices: open shoelace over the 9 indices (the
    padding index points at a zeroed intersection, so its terms vanish), float32 like the reference."""
    sel = np.take_along_axis(raw, idx[..., None].astype(np.int64).repeat(2, -1), axis=1)
    tot = (sel[:, :-1, 0] * sel[:, 1:, 1] - sel[:, :-1, 1] * sel[:, 1:, 0]).sum(1, dtype=np.float32)
    return np.abs(tot) / 2


def _check_overlaps(d, idx):
    area = np.maximum(d["box1"][:, 2] * d["box1"][:, 3], d["box2"][:, 2] * d["box2"][:, 3])
    got = _overlap_from(idx, d["raw_vertices"])
    np.testing.assert_allclose(got, d["overlap"], rtol=0, atol=1e-6 * area.max())       # what the reference pipeline returned
    assert (np.abs(got - d["clipped"]) <= 1e-4 * area).all()                            # ... and the true overlap
    assert {0, 3, 4, 5, 6, 7, 8} <= set(d["num_valid"].tolist())


def test_sort_vertices_oracle_is_pinned_by_the_reference_iou_pipeline():
    """tests/golden/iou_pairs.npz: the reference's rotated-box IoU (IoU.py:139-198, its torch code run on CPU) on 256 box pairs
    with the restatement standing in for the one CUDA-only call; the overlap areas that come out equal an independent float64
    polygon clipper's (checked at generation time and again here), so the restatement orders vertices the way the reference's
    only consumer of the kernel requires -- including the 8-corner identical-box rule and the all-padding cases."""
    from oracle import eval_oracle
    d, v, m, nv = _iou_fixture()
    idx = eval_oracle.sort_vertices(v, m, nv)[0]
    assert np.array_equal(idx, d["idx"])
    _check_overlaps(d, idx)


def test_bev_nms_oracle_matches_reference_fixture():
    from oracle import eval_oracle
    for name in ("decode_mc.npz", "decode_wt.npz", "decode_mx.npz"):
        d = np.load(golden_path(name))
        got, ref = eval_oracle.bev_nms(d["heatmap"][0, 0]), d["nms"][0, 0]
        assert np.array_equal(got > 0, ref > 0), name
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-7)


@pytest.mark.gpu
def test_sort_vertices_kernel_matches_the_restatement():
    from oracle import eval_oracle
    from vfa_amd import eval_ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    v, m, nv = _rect_polygons(rng, 3, 150)
    # plus the reference's own self-test input (cuda_ext.py:21-27): random points, random masks
    g = torch.Generator().manual_seed(0)
    rv = torch.rand([2, 150, 24, 2], generator=g)
    rv = (rv - rv.mean(dim=2, keepdim=True)).numpy()
    rm = (torch.rand([2, 150, 24], generator=g) > 0.8).numpy()
    rm[..., 8:][rm[..., 8:].all(-1)] = False
    v, m = np.concatenate([v, rv]), np.concatenate([m, rm])
    nv = m.sum(-1).astype(np.int32)
    want = eval_oracle.sort_vertices(v, m, nv)
    got = eval_ops.sort_v(torch.from_numpy(v).to(dev), torch.from_numpy(m).to(dev), torch.from_numpy(nv).to(dev))
    assert got.dtype == torch.int32 and tuple(got.shape) == want.shape
    assert np.array_equal(got.cpu().numpy(), want)
    # empty batch, and a polygon count that is not a multiple of the block
    assert eval_ops.sort_v(torch.zeros(0, 5, 24, 2, device=dev), torch.zeros(0, 5, 24, dtype=torch.bool, device=dev),
                           torch.zeros(0, 5, dtype=torch.int32, device=dev)).shape == (0, 5, 9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["decode_mc.npz", "decode_wt.npz", "decode_mx.npz"])
def test_bev_decode_matches_reference_fixture(name):
    from vfa_amd import eval_ops
    dev = torch.device("cuda:0")
    d = np.load(golden_path(name))
    base = str(d["base"])
    dec = eval_ops.BEVDecoder(base, tuple(d["world_size"]), tuple(d["cube_LWH"]), dimension_mean=d["dimension_mean"], topk=100)
    heat = torch.from_numpy(d["heatmap"]).to(dev)
    nms = dec.nms(heat).cpu().numpy()
    assert np.array_equal(nms > 0, d["nms"] > 0)  # the same peaks, plateau ties and border cells included
    np.testing.assert_allclose(nms, d["nms"], rtol=0, atol=2e-7)
    pred = {"heatmap": heat, "loc_offset": torch.from_numpy(d["loc_offset"]).to(dev)}
    if base == "MultiviewC":
        pred["dim_offset"] = torch.from_numpy(d["dim_offset"]).to(dev)
        pred["rotation"] = torch.from_numpy(d["rotation_logits"]).to(dev)
    out = dec.batch_decode(pred, 0.4)
    keys = ("conf", "location", "dimension", "rotation") if base == "MultiviewC" else ("conf", "location")
    # detections with EQUAL confidence (a plateau of the heat map) come out of top-k in no particular order: sort by
    # (confidence, x, y) on both sides
    def order(conf, loc):
        return np.lexsort((np.round(loc[:, 1], 3), np.round(loc[:, 0], 3), -conf))
    order_ref = order(d["out_conf"], d["out_location"])
    order_got = order(out["conf"].cpu().numpy(), out["location"].cpu().numpy())
    for k in keys:
        got, ref = out[k].cpu().numpy()[order_got], d["out_" + k][order_ref]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5, err_msg=k)
