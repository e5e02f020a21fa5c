"""GPU parity of the stages around the matcher (remap, gray, vis, speckle, depth, RGB pipeline),
each called through the C ABI and compared with the CPU oracle."""
import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu


def _params(D, paths, **kw):
    ph = svhip.default_params(D, paths, **kw)
    po = svref.Params()
    for f, _ in svhip.Params._fields_:
        setattr(po, f, getattr(ph, f))
    return ph, po


def test_disparity_vis_bit_exact(gpu):
    W, H = 200, 64
    rng = np.random.default_rng(0)
    d16 = rng.integers(-16, 256 * 16, size=(H, W)).astype(np.int16)
    d16[0, :8] = [-16, 0, 8, 24, 40, 16 * 85, 16 * 86, 32767]
    m = svhip.Matcher(W, H, svhip.default_params(64, 8))
    for scale in (1, 3):
        assert np.array_equal(m.disparity_vis(d16, scale), svref.disparity_vis(d16, scale))
    m.close()


def test_remap_gray_pipeline_bit_exact(gpu):
    """compute_rgb = remap x2 + gray x2 + SGM: rectified images, and the final disparity, are
    bit-exact (the remap is the fixed-point form, so better than the 1 LSB the contract allows)."""
    W, H, D = 320, 240, 64
    left, right, _ = synth.stereo_pair(101, W, H, D)
    rng = np.random.default_rng(1)
    lrgb, rrgb = synth.rgb_from_gray(rng, left), synth.rgb_from_gray(rng, right)
    mlx, mly = synth.rectify_maps(W, H, 0)
    mrx, mry = synth.rectify_maps(W, H, 1)
    ph, po = _params(D, 8, speckle_window_size=50, speckle_range=16)
    m = svhip.Matcher(W, H, ph)
    with pytest.raises(svhip.SvError) as e:
        m.compute_rgb(lrgb, rrgb, rectify=True)
    assert e.value.code == -5  # maps not uploaded yet
    m.upload_maps(mlx, mly, mrx, mry)
    got = m.compute_rgb(lrgb, rrgb, rectify=True)
    rl, rr = svref.remap_rgb(lrgb, mlx, mly), svref.remap_rgb(rrgb, mrx, mry)
    assert np.abs(m.rectified(0).astype(int) - rl).max() <= 1 and np.abs(m.rectified(1).astype(int) - rr).max() <= 1
    assert np.array_equal(m.rectified(0), rl) and np.array_equal(m.rectified(1), rr)
    gl, gr = svref.gray_from_rgb(rl), svref.gray_from_rgb(rr)
    assert np.array_equal(got, svref.sgbm(gl, gr, po))
    # without rectification
    got2 = m.compute_rgb(lrgb, rrgb, rectify=False)
    assert np.array_equal(got2, svref.sgbm(svref.gray_from_rgb(lrgb), svref.gray_from_rgb(rrgb), po))
    m.close()


def test_remap_out_of_range_maps(gpu):
    W, H, D = 96, 40, 32
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    mx = xx * 1.3 - 20.25
    my = yy * 1.2 - 5.5
    mx[0, 0], my[0, 1] = -1e12, 1e12
    m = svhip.Matcher(W, H, svhip.default_params(D, 5))
    m.upload_maps(mx, my, mx, my)
    m.compute_rgb(img, img, rectify=True)
    assert np.array_equal(m.rectified(0), svref.remap_rgb(img, mx, my))
    m.close()


@pytest.mark.parametrize("max_size,max_diff", [(50, 256), (10, 16), (400, 32)])
def test_speckle_filter_bit_exact(gpu, max_size, max_diff):
    import torch
    W, H, D = 240, 120, 64
    left, right, _ = synth.stereo_pair(33, W, H, D)
    ph, po = _params(D, 8, uniqueness_ratio=25)
    m = svhip.Matcher(W, H, ph)
    d = m.compute_gray(left, right)
    t = torch.from_numpy(d.copy()).cuda()
    m.filter_speckles_dev(t, max_size, max_diff)
    torch.cuda.synchronize()
    exp = svref.filter_speckles(d, -16, max_size, max_diff)
    assert (exp != d).any()
    assert np.array_equal(t.cpu().numpy(), exp)
    m.close()


def test_depth_within_tolerance(gpu):
    """depthFromDisparity: <= 1e-4 relative against the oracle's non-aliased form."""
    W, H, D = 320, 240, 64
    left, right, _ = synth.stereo_pair(303, W, H, D)
    ph, po = _params(D, 8)
    m = svhip.Matcher(W, H, ph)
    d = m.compute_gray(left, right)
    mx, my = synth.rectify_maps(W, H, 0)
    mapper = np.stack([mx, my], -1)
    f, b = 0.7 * W, 0.068
    Q = np.array([[1, 0, 0, -(W / 2 - 0.5)], [0, 1, 0, -(H / 2 - 0.5)], [0, 0, 0, f], [0, 0, 1 / b, 0]], np.float64)
    a = 0.01
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float64)
    got = m.depth(d, mapper, Q, R)
    exp = svref.depth(d, mapper, Q, R, alias=False)
    fin = np.isfinite(exp)
    assert np.array_equal(np.isfinite(got), fin)
    rel = np.abs(got[fin] - exp[fin]) / np.maximum(np.abs(exp[fin]), 1e-12)
    assert rel.max() <= 1e-4, rel.max()
    assert (got[fin] != 0).mean() > 0.5
    m.close()
