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
