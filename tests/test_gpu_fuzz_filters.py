"""Seeded random geometries through the stages around the matcher (speckle filter, visualisation, remap,
WLS, both bilateral filters, depth) against the oracle, to the same bars as the fixed-size tests: integers
and the float stages that keep the oracle's operation order bit-exact, depth <= 1e-4 relative.  The
disparity maps are synthetic (piecewise planes + invalid holes + noise), so tiny and narrow images that
the matcher's geometry limits would not allow are covered too."""
import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu


def _disp_map(rng, W, H, D):
    """disparity16-like map: planes, steps, -16 holes, columns [0, D) invalid"""
    yy, xx = np.mgrid[0:H, 0:W]
    d = (16 * (2 + (D - 4) * (0.3 + 0.2 * np.sin(xx / 17.0) + 0.2 * np.cos(yy / 11.0)))).astype(np.int32)
    for _ in range(int(rng.integers(1, 6))):
        x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
        w, h = int(rng.integers(1, max(2, W // 2))), int(rng.integers(1, max(2, H // 2)))
        d[y0:y0 + h, x0:x0 + w] = 16 * int(rng.integers(1, D - 1)) + int(rng.integers(0, 16))
    d += rng.integers(-3, 4, size=d.shape)
    d = np.clip(d, 0, 16 * (D - 1))
    d[rng.random(d.shape) < 0.08] = -16
    for _ in range(int(rng.integers(0, 4))):   # small islands for the speckle filter
        x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
        d[max(y0 - 3, 0):y0 + 3, max(x0 - 3, 0):x0 + 3] = -16
        d[y0:y0 + 2, x0:x0 + 2] = 16 * int(rng.integers(1, D - 1))
    d[:, :min(D, W)] = -16
    return d.astype(np.int16)


def _geometries(seed, n, wmin=8, hmin=8):
    rng = np.random.default_rng(seed)
    for i in range(n):
        D = int(rng.choice([32, 64, 128]))
        W = D + int(rng.integers(wmin, 200))
        H = int(rng.integers(hmin, 160))
        yield i, rng, W, H, D


def test_speckle_vis_random(gpu):
    import torch
    for i, rng, W, H, D in _geometries(11, 12):
        d = _disp_map(rng, W, H, D)
        max_size, max_diff = int(rng.integers(1, 200)), int(rng.choice([0, 1, 16, 32, 512]))
        m = svhip.Matcher(W, H, svhip.default_params(D, 8))
        t = torch.from_numpy(d.copy()).cuda()
        m.filter_speckles_dev(t, max_size, max_diff)
        torch.cuda.synchronize()
        what = "case %d: %dx%d D=%d max_size=%d max_diff=%d" % (i, W, H, D, max_size, max_diff)
        assert np.array_equal(t.cpu().numpy(), svref.filter_speckles(d, -16, max_size, max_diff)), what
        assert np.array_equal(m.disparity_vis(d, 3), svref.disparity_vis(d, 3)), what
        m.close()


def test_remap_random_maps(gpu):
    for i, rng, W, H, D in _geometries(12, 8):
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        a, b = rng.uniform(0.7, 1.4, 2)
        mx = (a * xx + rng.uniform(-30, 30) + rng.normal(0, 2.0, xx.shape)).astype(np.float32)
        my = (b * yy + rng.uniform(-20, 20) + rng.normal(0, 2.0, xx.shape)).astype(np.float32)
        mx[rng.random(mx.shape) < 0.01] = np.float32(rng.choice([-1e9, 1e9, -0.5, W - 0.5, W - 1, -1]))
        m = svhip.Matcher(W, H, svhip.default_params(D, 5))
        m.upload_maps(mx, my, mx, my)
        m.compute_rgb(img, img, rectify=True)
        assert np.array_equal(m.rectified(0), svref.remap_rgb(img, mx, my)), "case %d: %dx%d" % (i, W, H)
        m.close()


def test_wls_random(gpu):
    for i, rng, W, H, D in _geometries(13, 10, wmin=12, hmin=12):
        bs = int(rng.choice([3, 5, 7, 9]))
        if W - D - 2 * (bs // 2) < 2 or H - 2 * (bs // 2) < 2:
            continue
        lam, sig = float(rng.choice([0.0, 100.0, 4500.0, 8000.0])), float(rng.choice([0.5, 1.0, 2.5]))
        d = _disp_map(rng, W, H, D)
        guide = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        ph = svhip.default_params(D, 8, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs, wls_lambda=lam, wls_sigma=sig)
        m = svhip.Matcher(W, H, ph)
        vis = m.disparity_vis(d, 3)
        roi = m.wls_roi()
        got16, got8 = m.filter_wls(d, guide, vis=vis)
        exp16, exp8 = svref.wls_filter(d, guide, roi, lam, sig, vis=vis)
        what = "case %d: %dx%d D=%d bs=%d lambda=%g sigma=%g roi=%s" % (i, W, H, D, bs, lam, sig, roi)
        assert np.array_equal(got16, exp16) and np.array_equal(got8, exp8), what
        m.close()


def test_blf_random(gpu):
    for i, rng, W, H, D in _geometries(14, 10):
        d = _disp_map(rng, W, H, D)
        sc, ss = float(rng.choice([2.0, 10.0, 25.0])), float(rng.choice([1.5, 10.0, 40.0]))
        m = svhip.Matcher(W, H, svhip.default_params(D, 8))
        what = "case %d: %dx%d D=%d sigma_color=%g sigma_space=%g" % (i, W, H, D, sc, ss)
        assert np.array_equal(m.filter_blf(d, sc, ss), svref.blf(d, sc, ss)), what
        gray = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
        rgb = np.repeat(gray[..., None], 3, axis=2)
        radius, iters, sigma = int(rng.integers(1, 9)), int(rng.integers(1, 4)), float(rng.choice([3.0, 10.0]))
        exp = svref.blf_joint(d, svref.gray_from_rgb(rgb), D, sigma, radius, iters)
        got = m.filter_blf_joint(d, D, sigma, guide_rgb=rgb, radius=radius, iters=iters)
        assert np.array_equal(got, exp), what + " joint radius=%d iters=%d sigma=%g" % (radius, iters, sigma)
        m.close()


def test_depth_random(gpu):
    for i, rng, W, H, D in _geometries(15, 8):
        d = _disp_map(rng, W, H, D)
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        mapper = np.stack([xx + rng.normal(0, 1.5, xx.shape), yy + rng.normal(0, 1.5, xx.shape)], -1).astype(np.float32)
        f, b = 0.7 * W, 0.068
        Q = np.array([[1, 0, 0, -(W / 2 - 0.5)], [0, 1, 0, -(H / 2 - 0.5)], [0, 0, 0, f], [0, 0, 1 / b, 0]], np.float64)
        a = float(rng.uniform(-0.03, 0.03))
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float64)
        m = svhip.Matcher(W, H, svhip.default_params(D, 8))
        got = m.depth(d, mapper, Q, R)
        exp = svref.depth(d, mapper, Q, R, alias=False)
        fin = np.isfinite(exp)
        assert np.array_equal(np.isfinite(got), fin), "case %d" % i
        rel = np.abs(got[fin] - exp[fin]) / np.maximum(np.abs(exp[fin]), 1e-12)
        assert rel.size == 0 or rel.max() <= 1e-4, "case %d: %dx%d rel %g" % (i, W, H, rel.max())
        m.close()
