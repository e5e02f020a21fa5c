"""CPU tests of the oracle's bilateral-grid filter (filterBLF / BLF_ORIGINAL, StereoMatcher.cpp:90-99,
fastBilateral.hpp:92-189).  The reference holds no test or fixture for it and needs OpenCV to build
(parity unpinned); the C restatement is checked against an independent numpy restatement written
from the same text, and against properties of the algorithm."""
import numpy as np
import pytest

import svref
from svhip import synth


def blf_numpy(src, sigma_color, sigma_space):
    """Second restatement, array-at-a-time (fastBilateral.hpp:113-189)."""
    H, W = src.shape
    v = src.astype(np.float64)
    vmin, vmax = v.min(), v.max()
    pad = 2
    gh = int((H - 1) / sigma_space) + 1 + 2 * pad
    gw = int((W - 1) / sigma_space) + 1 + 2 * pad
    gd = int((vmax - vmin) / sigma_color) + 1 + 2 * pad
    yy, xx = np.mgrid[0:H, 0:W]
    sy = (yy / sigma_space + 0.5).astype(np.int64) + pad
    sx = (xx / sigma_space + 0.5).astype(np.int64) + pad
    sz = ((v - vmin) / sigma_color + 0.5).astype(np.int64) + pad
    data = np.zeros((gh, gw, gd, 2))
    np.add.at(data[..., 0], (sy, sx, sz), v)
    np.add.at(data[..., 1], (sy, sx, sz), 1.0)
    buf = np.zeros_like(data)
    for dim in range(3):
        for _ in range(2):
            data, buf = buf, data
            c = buf[1:-1, 1:-1, 1:-1]
            lo = [slice(1, -1)] * 3
            hi = [slice(1, -1)] * 3
            lo[dim], hi[dim] = slice(0, -2), slice(2, None)
            data[1:-1, 1:-1, 1:-1] = (buf[tuple(lo)] + buf[tuple(hi)] + 2.0 * c) / 4.0
    val = data[..., 0] / np.where(data[..., 1] != 0, data[..., 1], 1.0)
    py, px, pz = yy / sigma_space + pad, xx / sigma_space + pad, (v - vmin) / sigma_color + pad
    y0 = np.clip(py.astype(np.int64), 0, gh - 1); y1 = np.clip(y0 + 1, 0, gh - 1)
    x0 = np.clip(px.astype(np.int64), 0, gw - 1); x1 = np.clip(x0 + 1, 0, gw - 1)
    z0 = np.clip(pz.astype(np.int64), 0, gd - 1); z1 = np.clip(z0 + 1, 0, gd - 1)
    ya, xa, za = py - y0, px - x0, pz - z0
    r = ((1.0 - ya) * (1.0 - xa) * (1.0 - za) * val[y0, x0, z0] + (1.0 - ya) * xa * (1.0 - za) * val[y0, x1, z0]
         + ya * (1.0 - xa) * (1.0 - za) * val[y1, x0, z0] + ya * xa * (1.0 - za) * val[y1, x1, z0]
         + (1.0 - ya) * (1.0 - xa) * za * val[y0, x0, z1] + (1.0 - ya) * xa * za * val[y0, x1, z1]
         + ya * (1.0 - xa) * za * val[y1, x0, z1] + ya * xa * za * val[y1, x1, z1])
    return np.clip(np.rint(r), -32768, 32767).astype(np.int16)


def _disp(seed, W, H, D):
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    return svref.sgbm(left, right, svref.default_params(D, 8))


@pytest.mark.parametrize("W,H,D,sc,ss", [(160, 96, 64, 10.0, 10.0), (131, 57, 32, 3.5, 7.25), (96, 40, 32, 30.0, 1.0), (200, 64, 64, 1.0, 30.0)])
def test_blf_two_restatements_agree(W, H, D, sc, ss):
    d = _disp(5, W, H, D)
    got = svref.blf(d, sc, ss)
    assert np.array_equal(got, blf_numpy(d, sc, ss))
    assert got.min() >= d.min() and got.max() <= d.max()   # a weighted mean of the input values


def test_blf_properties():
    c = np.full((37, 53), 16 * 21 + 3, np.int16)
    assert np.array_equal(svref.blf(c, 10.0, 10.0), c)          # a constant map is a fixed point
    d = _disp(6, 160, 96, 64)
    out = svref.blf(d, 10.0, 10.0)
    assert (out != d).mean() > 0.2                               # it does filter
    # invalid pixels (-16) lie > sigma_color away from any valid disparity >= 16: they stay near -16
    assert np.abs(out[d == -16].astype(int) + 16).max() <= 16
    # a huge range sigma turns it into a purely spatial blur of the map: strictly less contrast
    assert svref.blf(d, 1e6, 10.0).std() < d.std()
    with pytest.raises(ValueError):
        svref.blf(d, 0.0, 10.0)
    with pytest.raises(ValueError):
        svref.blf(d, 10.0, -1.0)


# ---- BLF_CUDA semantics (EXT): cv::cuda::DisparityBilateralFilter, deterministic schedule -------------

def blf_joint_python(disp, guide, ndisp, sigma, radius=7, iters=2):
    """Second restatement, scalar loops with float32 accumulation (small images only)."""
    f32 = np.float32
    H, W = disp.shape
    tc = np.array([f32(np.exp(-float(i * i) / (2 * sigma * sigma))) for i in range(256)], f32)
    ts = np.array([[np.exp(-np.sqrt(f32(y * y) + f32(x * x)) / f32(radius + 1.0)) for x in range(radius + 1)]
                   for y in range(radius + 1)], f32)
    edge = max(1, int(f32(ndisp) * f32(0.1) + 0.5))
    maxd = int(f32(ndisp) * f32(0.2) + 0.5)
    out = disp.astype(np.int64).copy()
    g = guide.astype(np.int64)
    for _ in range(iters):
        for t in (0, 1):
            cur = out.copy()
            for y in range(1, H - 1):
                for x in range(1, W - 1):
                    if (x & 1) != ((y + t) & 1):
                        continue
                    dp = [cur[y, x], cur[y - 1, x], cur[y, x - 1], cur[y + 1, x], cur[y, x + 1]]
                    if not any(abs(v - dp[0]) >= edge for v in dp[1:]):
                        continue
                    cost = [f32(0)] * 5
                    for yi in range(max(0, y - radius), min(H - 1, y + radius) + 1):
                        for xi in range(max(0, x - radius), min(W - 1, x + radius) + 1):
                            w = f32(tc[abs(g[yi, xi] - g[y, x])] * ts[abs(y - yi), abs(x - xi)])
                            for k in range(5):
                                cost[k] = f32(cost[k] + f32(f32(min(maxd, abs(cur[yi, xi] - dp[k]))) * w))
                    out[y, x] = dp[int(np.argmin(np.array(cost, f32)))]   # argmin: first minimum, as the < scan
    return out.astype(np.int16)


def test_blf_joint_two_restatements_agree():
    W, H, D = 72, 40, 32
    left, right, _ = synth.stereo_pair(12, W, H, D)
    d = svref.sgbm(left, right, svref.default_params(D, 8))
    got = svref.blf_joint(d, left, D, 10.0)
    assert np.array_equal(got, blf_joint_python(d, left, D, 10.0))
    assert 0.0 < (got != d).mean() < 0.5                     # only pixels at disparity edges move
    assert np.array_equal(got[0], d[0]) and np.array_equal(got[:, 0], d[:, 0])   # border untouched
    got1 = svref.blf_joint(d, left, D, 3.0, radius=3, iters=1)
    assert np.array_equal(got1, blf_joint_python(d, left, D, 3.0, radius=3, iters=1))
    flat = np.full((H, W), 160, np.int16)
    assert np.array_equal(svref.blf_joint(flat, left, D), flat)   # no edge, nothing to do
    with pytest.raises(ValueError):
        svref.blf_joint(d, left, D, 0.0)
