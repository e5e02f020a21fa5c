"""GPU parity of the colour-cost variant (SV_FLAG_COST_RGB: cv::StereoSGBM on the 8UC3 pair, matchSGBM,
StereoMatcher.cpp:392-417) against the CPU oracle: cost volume, aggregated volume and final disparity
bit-exact, with and without rectification."""
import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu


def _params(D, bs, **kw):
    ph = svhip.default_params(D, 8, block_size=bs, P1=8 * 3 * bs * bs, P2=32 * 3 * bs * bs, **kw)
    ph.flags |= svhip.FLAG_COST_RGB
    po = svref.Params()
    for f, _ in svhip.Params._fields_:
        setattr(po, f, getattr(ph, f))
    return ph, po


def _rgb_pair(seed, W, H, D):
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    rng = np.random.default_rng(seed + 1)
    return synth.rgb_from_gray(rng, left), synth.rgb_from_gray(rng, right)


@pytest.mark.parametrize("W,H,D,bs", [(96, 40, 32, 7), (131, 57, 64, 5), (200, 50, 96, 3), (320, 240, 64, 7), (640, 480, 128, 7)])
def test_rgb_cost_pipeline_bit_exact(gpu, W, H, D, bs):
    lrgb, rrgb = _rgb_pair(31, W, H, D)
    ph, po = _params(D, bs)
    ref, st = svref.sgbm(lrgb, rrgb, po, stages=("C", "S"))
    m = svhip.Matcher(W, H, ph)
    assert np.array_equal(m.compute_rgb(lrgb, rrgb, rectify=False), ref), "production path"
    for mode in (4, 5, 2):
        m.set_debug(mode)
        assert np.array_equal(m.compute_rgb(lrgb, rrgb, rectify=False), ref), "debug mode %d" % mode
    m.set_debug(True)
    got = m.compute_rgb(lrgb, rrgb, rectify=False)
    assert np.array_equal(m.debug_volume(0), st["C"]), "colour cost volume differs"
    assert np.array_equal(m.debug_volume(1), st["S"]), "aggregated volume differs"
    assert np.array_equal(got, ref)
    m.close()


def test_rgb_cost_with_rectification_and_flag_rules(gpu):
    W, H, D = 320, 240, 64
    lrgb, rrgb = _rgb_pair(32, W, H, D)
    mlx, mly = synth.rectify_maps(W, H, 0)
    mrx, mry = synth.rectify_maps(W, H, 1)
    ph, po = _params(D, 7, speckle_window_size=50, speckle_range=16)
    m = svhip.Matcher(W, H, ph)
    m.upload_maps(mlx, mly, mrx, mry)
    got = m.compute_rgb(lrgb, rrgb, rectify=True)
    rl, rr = svref.remap_rgb(lrgb, mlx, mly), svref.remap_rgb(rrgb, mrx, mry)
    assert np.array_equal(got, svref.sgbm(rl, rr, po))
    with pytest.raises(svhip.SvError):
        m.compute_gray(lrgb[..., 0], rrgb[..., 0])       # the gray entry points reject the flag
    # switching the flag off on the same context: the gray pipeline again (buffers re-sized)
    pg = svhip.default_params(D, 8)
    m.set_params(pg)
    pog = svref.Params()
    for f, _ in svhip.Params._fields_:
        setattr(pog, f, getattr(pg, f))
    assert np.array_equal(m.compute_rgb(lrgb, rrgb, rectify=False),
                          svref.sgbm(svref.gray_from_rgb(lrgb), svref.gray_from_rgb(rrgb), pog))
    with pytest.raises(svhip.SvError):
        bad = svhip.default_params(D, 8, block_size=9, P1=8 * 3 * 81, P2=32 * 3 * 81)
        bad.flags |= svhip.FLAG_COST_RGB
        svhip.Matcher(W, H, bad)                          # colour cost would wrap int16
    m.close()
