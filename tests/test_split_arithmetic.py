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
