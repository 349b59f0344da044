"""The numerical contract of the MFMA collapse kernels, checked on the CPU: every fp32 operand is split exactly into two
bf16 values (x = hi + lo + r, |r| <= 2^-17 |x| worst case, 2^-18 typical) and a product is formed from three bf16 x bf16
products accumulated in fp32.  This restates vfa_amd/csrc/vfa_collapse*.hip:split_bf16 in numpy and bounds the error of
the 3- and 4-term schemes against float64 at the shapes of the path (K = 256 ... 2048, non-negative voxel features)."""
import numpy as np
import pytest


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return rounded.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_rne(x)
    lo = bf16_rne((x - hi).astype(np.float32))
    return hi, lo


def test_split_is_exact_to_two_bf16_mantissas():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-6, 6, 200000)).astype(np.float32)
    hi, lo = split(x)
    r = x.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64)
    assert np.all(np.abs(r) <= 2.0 ** -16 * np.abs(x))           # two 8-bit mantissas
    assert np.quantile(np.abs(r) / np.abs(x), 0.5) < 2.0 ** -18
    # both pieces are bf16 numbers, and hi carries the sign / zero of x
    for p in (hi, lo):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)
    assert np.all((hi == 0) == (x == 0))


@pytest.mark.parametrize("K,terms", [(256, 3), (256, 4), (1280, 3), (2048, 3)])
def test_three_term_product_meets_the_path_tolerance(K, terms):
    rng = np.random.default_rng(K + terms)
    M, N = 512, 256
    a = (rng.random((M, K)) * 3.0).astype(np.float32)
    a[rng.random(M) < 0.3] = 0.0
    w = ((rng.random((N, K)) - 0.5) * (2.0 / np.sqrt(K))).astype(np.float32)
    a_hi, a_lo = split(a)
    w_hi, w_lo = split(w)
    f64 = np.float64
    out = a_lo.astype(f64) @ w_hi.astype(f64).T + a_hi.astype(f64) @ w_lo.astype(f64).T + a_hi.astype(f64) @ w_hi.astype(f64).T
    if terms == 4:
        out = out + a_lo.astype(f64) @ w_lo.astype(f64).T
    ref = a.astype(f64) @ w.astype(f64).T
    scale = np.abs(ref).max()
    err = np.abs(out - ref)
    assert err.max() <= 0.7e-5 * scale                      # the kernels measure 3e-6 (K = 256) ... 5e-6 (K = 2048)
    assert np.all(err <= 1e-4 * np.abs(ref) + 1e-5 * scale)  # the path's post-GEMM tolerance, with the split error alone


def split3(x):
    p0 = bf16_rne(x)
    r1 = (x - p0).astype(np.float32)
    p1 = bf16_rne(r1)
    p2 = bf16_rne((r1 - p1).astype(np.float32))
    return p0, p1, p2


def test_three_piece_split_is_exact_to_the_fp32_mantissa():
    """vfa_pipe.hip, VFA_FLAG_TERMS 6: x = p0 + p1 + p2 + r with |r| <= 2^-24 |x| (three 8-bit mantissas cover fp32's 24 bits;
    the residuals x - p0 and x - p0 - p1 are exact in fp32)."""
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-6, 6, 200000)).astype(np.float32)
    p0, p1, p2 = split3(x)
    r = x.astype(np.float64) - p0.astype(np.float64) - p1.astype(np.float64) - p2.astype(np.float64)
    assert np.all(np.abs(r) <= 2.0 ** -24 * np.abs(x))
    assert np.mean(r == 0) > 0.9
    hi, lo = split(x)
    assert np.array_equal(p0, hi) and np.array_equal(p1, lo)  # the first two pieces ARE the two-piece split: one weight table serves both


@pytest.mark.parametrize("K", [256, 1280, 8192])
def test_six_term_product_is_sgemm_class(K):
    """The six products p0 p0, p0 p1, p1 p0, p0 p2, p2 p0, p1 p1 (everything down to 2^-16 of the largest) against float64: the
    split itself contributes <= 3e-8 normwise -- below the rounding of an fp32 accumulation over K terms (~1e-7 ... 3e-7), i.e.
    the product has the arithmetic width of the reference's fp32 nn.Linear (vfa_op.py:123).  Three products of two pieces:
    ~2e-6."""
    rng = np.random.default_rng(K)
    M, N = 256, 256
    a = (rng.standard_normal((M, K)) * np.exp(1.5 * rng.standard_normal((M, K)))).astype(np.float32)  # signed, heavy-tailed
    w = ((rng.random((N, K)) - 0.5) * (2.0 / np.sqrt(K))).astype(np.float32)
    f64 = np.float64
    a0, a1, a2 = (p.astype(f64) for p in split3(a))
    w0, w1, w2 = (p.astype(f64) for p in split3(w))
    six = a0 @ w0.T + a0 @ w1.T + a1 @ w0.T + a0 @ w2.T + a2 @ w0.T + a1 @ w1.T
    three = a0 @ w0.T + a0 @ w1.T + a1 @ w0.T
    ref = a.astype(f64) @ w.astype(f64).T
    e6 = np.linalg.norm(six - ref) / np.linalg.norm(ref)
    e3 = np.linalg.norm(three - ref) / np.linalg.norm(ref)
    assert e6 <= 3e-8, e6
    assert 3e-7 <= e3 <= 1e-5, e3


# ---------------------------------------------------------------------------------------------------------------------------
# the default arithmetic of the fused frame kernels: two fp16 pieces per operand with a power-of-two scale (vfa_split.h)
# ---------------------------------------------------------------------------------------------------------------------------
EXP_A, EXP_W, EXP_LIM = 13, 14, 40  # vfa_split.h: kExpA, kExpW, kExpLim


def split_exponent(absmax, target):
    """vfa_split.h: split_exponent -- the power of two that brings `absmax` into [2^target, 2^(target+1))."""
    bits = int(np.float32(absmax).view(np.uint32)) & 0x7FFFFFFF
    if bits == 0 or bits >= 0x7F800000:
        return 0
    return int(np.clip(target - ((bits >> 23) - 127), -EXP_LIM, EXP_LIM))


def f16_sat(x):
    """fp32 -> fp16, round to nearest even, overflow clamped to +-65504 (MODE.FP16_OVFL), NaN / Inf kept; returned as fp32."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(over="ignore"):
        h = x.astype(np.float16)
    h = np.where(np.isinf(h) & np.isfinite(x), np.copysign(np.float16(65504), h), h)
    return h.astype(np.float32)


def split16(x, e):
    s = (x * np.float32(2.0 ** e)).astype(np.float32)
    hi = f16_sat(s)
    lo = f16_sat((s - hi).astype(np.float32))
    return hi, lo


def test_fp16_split_is_exact_to_two_fp16_mantissas_over_the_range_of_the_scale():
    """x 2^e = hi + lo + r with |r| <= 2^-23 |x 2^e| wherever lo is a normal fp16 number, and |r| <= 2^-25 (half a subnormal step)
    below that: with the maximum at 2^13 (kExpA) every element down to 2^-13 of it keeps 22 bits, whatever the magnitude of the map."""
    rng = np.random.default_rng(0)
    for mag in (1.0, 1e4, 1e-6, 1e-7, 1e8):  # (|e| <= 40: maxima from 2^-26 to 2^54)
        x = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-4, 0, 100000) * mag).astype(np.float32)
        e = split_exponent(np.abs(x).max(), EXP_A)
        hi, lo = split16(x, e)
        s = x.astype(np.float64) * 2.0 ** e
        assert 2.0 ** EXP_A <= np.abs(s).max() < 2.0 ** (EXP_A + 1)
        r = s - hi.astype(np.float64) - lo.astype(np.float64)
        assert np.all(np.abs(r) <= np.maximum(2.0 ** -23 * np.abs(s), 2.0 ** -25)), mag
        big = np.abs(s) >= 1.0  # 2^-14 of the maximum and up
        assert np.all(np.abs(r[big]) <= 2.0 ** -23 * np.abs(s[big]))


@pytest.mark.parametrize("K", [256, 1280, 8192])
def test_fp16_three_product_scheme_is_sgemm_class(K):
    """hi.hi + hi.lo + lo.hi of the scaled fp16 pieces against float64, beside a plain fp32 product of the same operands
    (np.float32 matmul: what the reference's nn.Linear is, vfa_op.py:123): the split contributes <= 1e-7 normwise (the dropped
    lo.lo is 2^-22 of a product), below the rounding of an fp32 accumulation over K terms -- and an order of magnitude below the
    two-piece bf16 form.  Voxel-like A (non-negative, a third of the rows masked), nn.Linear-scale W."""
    rng = np.random.default_rng(K)
    M, N = 256, 256
    a = (rng.random((M, K)) * 3.0).astype(np.float32)
    a[rng.random(M) < 0.3] = 0.0
    w = ((rng.random((N, K)) - 0.5) * (2.0 / np.sqrt(K))).astype(np.float32)
    f64 = np.float64
    ea, ew = split_exponent(np.abs(a).max() * 4, EXP_A), split_exponent(np.abs(w).max(), EXP_W)  # (features are ~4 x the voxel features)
    a_hi, a_lo = (p.astype(f64) for p in split16(a, ea))
    w_hi, w_lo = (p.astype(f64) for p in split16(w, ew))
    out = (a_hi @ w_hi.T + a_hi @ w_lo.T + a_lo @ w_hi.T) * 2.0 ** -(ea + ew)
    ref = a.astype(f64) @ w.astype(f64).T
    e16 = np.linalg.norm(out - ref) / np.linalg.norm(ref)
    e32 = np.linalg.norm((a @ w.T).astype(f64) - ref) / np.linalg.norm(ref)
    hb, lb = split(a)
    whb, wlb = split(w)
    eb = np.linalg.norm(hb.astype(f64) @ whb.astype(f64).T + hb.astype(f64) @ wlb.astype(f64).T + lb.astype(f64) @ whb.astype(f64).T - ref) / np.linalg.norm(ref)
    assert e16 <= 1e-7, e16
    assert e16 <= e32, (e16, e32)       # the split alone stays below what fp32 accumulation costs an sgemm
    assert eb >= 10 * e16, (eb, e16)    # ... and an order of magnitude below the 16-bit two-piece bf16 split


def test_fp16_split_range_nan_and_saturation():
    """Range: the same relative error for maps of magnitude 1e4 and 1e-6 (the scale follows the maximum).  NaN and Inf survive
    both pieces; a value beyond fp16's range after scaling saturates to finite pieces (never Inf - Inf)."""
    rng = np.random.default_rng(5)
    base = (rng.random((64, 256)) * 3.0).astype(np.float32)
    w = ((rng.random((256, 256)) - 0.5) * 0.125).astype(np.float32)
    errs = []
    for mag in (1.0, 1e4, 1e-6):
        a = (base * np.float32(mag)).astype(np.float32)
        ea, ew = split_exponent(np.abs(a).max() * 4, EXP_A), split_exponent(np.abs(w).max(), EXP_W)
        a_hi, a_lo = (p.astype(np.float64) for p in split16(a, ea))
        w_hi, w_lo = (p.astype(np.float64) for p in split16(w, ew))
        out = (a_hi @ w_hi.T + a_hi @ w_lo.T + a_lo @ w_hi.T) * 2.0 ** -(ea + ew)
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        errs.append(np.linalg.norm(out - ref) / np.linalg.norm(ref))
    assert max(errs) <= 1e-7 and max(errs) <= 1.5 * min(errs), errs
    hi, lo = split16(np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3.0], np.float32), 10)
    assert np.isnan(hi[0]) and np.isnan(lo[0])
    assert np.isinf(hi[1]) and np.isinf(hi[2])                     # (an infinite feature: Inf - Inf in lo, NaN in the product -- as in the bf16 forms)
    assert hi[3] == 65504 and lo[3] == 65504 and hi[4] == -65504 and lo[4] == -65504  # saturated, finite
    assert hi[5] + lo[5] == 3.0 * 2 ** 10
    assert split_exponent(0.0, EXP_A) == 0 and split_exponent(np.nan, EXP_A) == 0 and split_exponent(np.inf, EXP_A) == 0
    assert split_exponent(1e-30, EXP_A) == EXP_LIM and split_exponent(1e30, EXP_A) == -EXP_LIM


def _np_split_fragments(w, n_layers):
    """numpy restatement of the device weight split (vfa_split.h; vfa_fused.hip: split_block, vfa_pipe.hip: pipe_split_weight_kernel):
    w (256, 256 * nl) in the reference's column order c * nl + layer -> per layer the hi / lo fp16 planes in MFMA B-fragment order,
    (8 waves, 16 k-steps, 64 lanes, 8 values), and the scale exponent ew with max|w| 2^ew in [2^14, 2^15)."""
    w = np.asarray(w, dtype=np.float32)
    amax = np.abs(w).max()
    ew = 14 - int(np.floor(np.log2(amax)))
    x = (w * np.float32(2.0 ** ew)).astype(np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    wave, s, lane, j = np.meshgrid(np.arange(8), np.arange(16), np.arange(64), np.arange(8), indexing="ij")
    n = 32 * wave + (lane & 31)
    c = 16 * s + 8 * (lane >> 5) + j
    return [(hi[n, c * n_layers + layer], lo[n, c * n_layers + layer]) for layer in range(n_layers)], ew


@pytest.mark.gpu
@pytest.mark.parametrize("n_layers", [1, 3])
def test_device_weight_fragments_are_the_fp16_split(n_layers):
    """The collapse weights as the frame kernels read them -- split in the spare blocks of the work-cuts launch (serial path) / after the
    partial maxima those blocks leave (pipelined path) -- are, bit for bit, hi = RN_f16(w 2^ew), lo = RN_f16(w 2^ew - hi) in MFMA
    B-fragment order with the exponent of vfa_split.h (reference: the fp32 nn.Linear of vfa_op.py:59, :123)."""
    import torch
    from vfa_amd import _lib, ops
    import vfa_amd
    from vfa_amd.synthetic import make_workload
    dev = torch.device("cuda:0")
    wl = make_workload("multiviewc_200x200x1", channels=256, seed=1, n_cam=2)
    grid = wl["grid"][:, 40:64, 30:70].contiguous().to(dev)
    L, W = grid.shape[1:3]
    calibs = wl["calibs"].to(dev)
    sizes = [tuple(s) for s in wl["feat_sizes"]]
    gen = torch.Generator().manual_seed(7)
    # three scales with maxima in different binades, one of them tiny
    weights = [(torch.randn(256, 256 * n_layers, generator=gen) * sc).to(dev) for sc in (0.05, 3.0, 1e-4)]
    kind, img_wh = _lib.CONV_KIND[wl["args"].data], wl["args"].image_size[::-1]
    if n_layers == 1:
        mod = vfa_amd.VFA(256, grid_height=wl["grid_height"], cube_size=wl["cube_size"], args=wl["args"]).to(dev)
        zl, co = mod._kernel_geometry(dev)
        ws = ops.frame_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights)
        lay = ops.frame_workspace_layout(2, L, W, 3)
        planes = 2
    else:
        mod = vfa_amd.VFA(256, grid_height=96, cube_size=(wl["cube_size"][0], wl["cube_size"][1], 32), args=wl["args"]).to(dev)
        assert mod.num_grid_layer == n_layers
        zl, co = mod._kernel_geometry(dev)
        ws = ops.pipe_records(calibs, grid, zl, co, kind, img_wh, sizes, weights=weights)
        lay = ops.pipe_workspace_layout(2, L, W, n_layers, 3)
        planes = 3
    torch.cuda.synchronize()
    host = ws.cpu().numpy()
    for s in range(3):
        want, _ = _np_split_fragments(weights[s].cpu().numpy(), n_layers)
        per_layer = 8 * 16 * planes * 64 * 16
        for layer in range(n_layers):
            raw = host[lay["wfrag"][s] + layer * per_layer:lay["wfrag"][s] + (layer + 1) * per_layer].view(np.float16).reshape(8, 16, planes, 64, 8)
            hi, lo = want[layer]
            assert np.array_equal(raw[:, :, 0].view(np.uint16), hi.view(np.uint16)), (s, layer, "hi")
            assert np.array_equal(raw[:, :, 1].view(np.uint16), lo.view(np.uint16)), (s, layer, "lo")
            if planes == 3:
                assert not raw[:, :, 2].view(np.uint16).any()
