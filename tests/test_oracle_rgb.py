"""CPU tests of the oracle's colour-cost variant (matchSGBM, StereoMatcher.cpp:392-417: cv::StereoSGBM on
the 8UC3 pair, P1 = 8*3*SAD^2, P2 = 32*3*SAD^2).  EXT item, parity unpinned (OpenCV absent): checked
against the independent second restatement applied plane by plane, and against the gray matcher."""
import numpy as np
import pytest

import naive_sgm
import svref
from svhip import synth


def _rgb_pair(seed, W, H, D):
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    rng = np.random.default_rng(seed + 1)
    return synth.rgb_from_gray(rng, left), synth.rgb_from_gray(rng, right), left, right


@pytest.mark.parametrize("W,H,D,bs", [(96, 40, 32, 7), (131, 57, 64, 5), (90, 33, 32, 3)])
def test_rgb_cost_is_the_sum_over_planes(W, H, D, bs):
    lrgb, rrgb, _, _ = _rgb_pair(21, W, H, D)
    p = svref.default_params(D, 8, BS=bs)
    p.P1, p.P2 = 8 * 3 * bs * bs, 32 * 3 * bs * bs
    p.flags |= svref.FLAG_COST_RGB
    d, st = svref.sgbm(lrgb, rrgb, p, stages=("C", "S"))
    exp = np.full((H, W - D, D), p.P2, np.int64)
    for ch in range(3):
        c, _ = naive_sgm.cost_volume(lrgb[..., ch], rrgb[..., ch], D, bs, 0, 63, flags=p.flags & 0x3)
        exp += c
    assert np.array_equal(st["C"], exp.astype(np.int16))
    # everything after the cost volume is the gray matcher's: the naive aggregation of this C gives S
    S, _ = naive_sgm.aggregate(st["C"], p.P1, p.P2, 0xFF)
    assert np.array_equal(st["S"], S)
    assert (d[:, D:] >= 0).mean() > 0.5


def test_rgb_entry_rules_and_relation_to_gray():
    W, H, D = 96, 40, 32
    lrgb, rrgb, left, right = _rgb_pair(22, W, H, D)
    p = svref.default_params(D, 8)
    with pytest.raises(ValueError):
        svref.sgbm(lrgb, rrgb, p)                       # colour images need the flag
    p.flags |= svref.FLAG_COST_RGB
    with pytest.raises(ValueError):
        svref.sgbm(left, right, p)                      # and the flag needs colour images
    # three identical planes: C - P2 is three times the gray C - P2, so with P1, P2 scaled by 3 every
    # path cost scales by 3 and winner / sub-pixel / LR check are those of the gray matcher
    g3l, g3r = np.repeat(left[..., None], 3, -1), np.repeat(right[..., None], 3, -1)
    # (two paths only, so that no S reaches the 32767 saturation, which does not scale)
    p.P1, p.P2, p.uniqueness_ratio, p.path_mask = 3 * 8 * 49, 3 * 32 * 49, 0, 0x05
    pg = svref.default_params(D, 8, uniq=0)
    pg.path_mask = 0x05
    d3, st3 = svref.sgbm(g3l, g3r, p, stages=("S",))
    dg, stg = svref.sgbm(left, right, pg, stages=("S",))
    unsat = st3["S"] < 32767
    assert unsat.mean() > 0.9 and np.array_equal(st3["S"][unsat], 3 * stg["S"][unsat].astype(np.int32))
    assert np.array_equal(d3, dg)
    # block 9 with three channels would wrap the int16 cost: rejected
    p9 = svref.default_params(D, 8, BS=9)
    p9.flags |= svref.FLAG_COST_RGB
    with pytest.raises(ValueError):
        svref.sgbm(lrgb, rrgb, p9)
