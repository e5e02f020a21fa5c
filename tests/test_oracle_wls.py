"""Pins of the oracle's WLS / fast-global-smoother restatement (EXT: OpenCV is not
installed, parity unpinned): the Thomas solves must solve the published linear system
(I + lambda*A) u = f, checked against a float64 banded solver; constants are a fixed point;
ROI and border conventions follow StereoMatcher.cpp:12-29 / :154-160."""
import numpy as np
from scipy.linalg import solve_banded

import svref


def _banded(f, wt, lam):
    n = len(f)
    ab = np.zeros((3, n))
    up = -lam * wt[:-1]
    ab[0, 1:] = up
    ab[2, :-1] = up
    ab[1, :] = 1
    ab[1, :-1] += lam * wt[:-1]
    ab[1, 1:] += lam * wt[:-1]
    return solve_banded((1, 1), ab, f)


def _fgs64(g, img, lam, sig, att, iters):
    gi = g.astype(np.int64)
    h, w = img.shape
    wh, wv = np.zeros((h, w)), np.zeros((h, w))
    wh[:, :-1] = np.exp(-np.sqrt(((gi[:, 1:] - gi[:, :-1]) ** 2).sum(-1)) / sig)
    wv[:-1, :] = np.exp(-np.sqrt(((gi[1:, :] - gi[:-1, :]) ** 2).sum(-1)) / sig)
    ref = img.astype(np.float64).copy()
    for _ in range(iters):
        for i in range(h):
            ref[i] = _banded(ref[i], wh[i], lam)
        for j in range(w):
            ref[:, j] = _banded(ref[:, j], wv[:, j], lam)
        lam *= att
    return ref


def test_fgs_solves_the_wls_system():
    rng = np.random.default_rng(1)
    h, w = 37, 53
    g = (rng.integers(0, 256, (h, w, 3), dtype=np.uint8) // 32 * 8 + 100).astype(np.uint8)
    img = rng.normal(0, 100, (h, w)).astype(np.float32)
    for lam, sig, iters in ((50.0, 4.0, 1), (4500.0, 1.0, 3), (8000.0, 1.5, 3)):
        out = svref.fgs(g, img, lam, sig, 0.25, iters)
        ref = _fgs64(g, img, lam, sig, 0.25, iters)
        assert np.abs(out - ref).max() < 2e-2 * max(1.0, np.abs(ref).max() / 100), (lam, sig)


def test_fgs_constant_is_fixed_point_and_mean_preserving():
    rng = np.random.default_rng(2)
    g = rng.integers(0, 256, (20, 31, 3), dtype=np.uint8)
    const = np.full((20, 31), 37.0, np.float32)
    assert np.allclose(svref.fgs(g, const, 4500.0, 1.0), 37.0, atol=1e-3)
    # uniform guide: every weight is -1, the filter is a symmetric smoother -> the mean is kept
    flat = np.full((20, 31, 3), 90, np.uint8)
    img = rng.normal(0, 10, (20, 31)).astype(np.float32)
    out = svref.fgs(flat, img, 100.0, 1.0)
    assert abs(out.mean() - img.mean()) < 1e-3 and out.std() < 0.2 * img.std()


def test_wls_lut_and_roi():
    lut = svref.wls_lut(1.0)
    assert lut[0] == -1.0 and abs(lut[1] + np.exp(-1.0)) < 1e-7 and lut[-1] == 0.0
    assert np.all(np.diff(lut[:2000]) >= 0)  # weights decay with colour distance
    # computeROI2: x in [maxD + bs/2, W + minD - bs/2), y in [bs/2, H - bs/2)
    assert svref.wls_roi(640, 480, svref.default_params(128)) == (130, 3, 507, 474)
    assert svref.wls_roi(320, 240, svref.default_params(64, BS=5)) == (65, 2, 253, 236)


def test_wls_filter_border_and_rounding():
    rng = np.random.default_rng(3)
    W, H = 96, 40
    p = svref.default_params(32)
    roi = svref.wls_roi(W, H, p)
    d = rng.integers(-16, 31 * 16, (H, W)).astype(np.int16)
    g = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    vis = svref.disparity_vis(d, 3)
    out16, out8 = svref.wls_filter(d, g, roi, 4500.0, 1.0, vis=vis)
    x, y, w, h = roi
    mask = np.zeros((H, W), bool)
    mask[y:y + h, x:x + w] = True
    assert np.all(out16[~mask] == -16) and np.all(out8[~mask] == 0)
    # lambda = 0 is the identity inside the ROI (convertTo float and back)
    id16, id8 = svref.wls_filter(d, g, roi, 0.0, 1.0, vis=vis)
    assert np.array_equal(id16[mask], d[mask]) and np.array_equal(id8[mask], vis[mask])
    # inside: equals the float smoother rounded half to even
    f = svref.fgs(g[y:y + h, x:x + w], d[y:y + h, x:x + w].astype(np.float32), 4500.0, 1.0)
    assert np.array_equal(out16[y:y + h, x:x + w], np.rint(f).astype(np.int16))
