"""GPU parity with minDisparity != 0 (EXT item: cv::StereoSGBM as matchSGBM parameterises it, StereoMatcher.cpp:399;
RPC setMinDisp, DispModule.cpp:1355-1420): computable columns [max(minD + D, 0), W + min(minD, 0)), right pixel
x - d - minD, disparity16 = (d + minD) * 16 + sub-pixel, invalid = (minD - 1) * 16.  Every stage through the C ABI
against the oracle (whose min_disparity arithmetic a second restatement confirms, tests/test_oracle.py)."""
import os

import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu


def _params(D, paths, minD, **kw):
    ph = svhip.default_params(D, paths, **kw)
    ph.min_disparity = minD
    po = svref.Params()
    for f, _ in svhip.Params._fields_:
        setattr(po, f, getattr(ph, f))
    return ph, po


@pytest.mark.parametrize("W,H,D,minD,paths,mode", [
    (200, 70, 64, 8, 8, 1),      # debug: per-path kernels, S materialised
    (200, 70, 64, 8, 8, 3),      # fused sweeps
    (200, 70, 64, 8, 8, 5),      # all paths in one launch + sum/WTA
    (200, 70, 64, 8, 8, 2),      # WTA fused into the last path
    (230, 64, 32, -12, 8, 3),    # negative: valid disparities below zero
    (230, 64, 32, -12, 5, 1),
    (300, 50, 96, 37, 5, 0),     # production choice, odd offset
    (320, 80, 128, -140, 8, 3),  # whole range negative (maxD < 0): columns [0, W + minD)
])
def test_min_disparity_bit_exact(gpu, W, H, D, minD, paths, mode):
    left, right, _ = synth.stereo_pair(900 + minD, W, H, D)
    ph, po = _params(D, paths, minD, speckle_window_size=30, speckle_range=2)
    ref, st = svref.sgbm(left, right, po, stages=("C", "S", "mins", "raw", "disp_wta", "disp_lr"))
    col0, colN, W1, inv = svref.geometry(W, po)
    m = svhip.Matcher(W, H, ph)
    m.set_debug(mode)
    got = m.compute_gray(left, right)
    assert np.array_equal(m.debug_volume(0), st["C"]), "C differs"
    if mode == 1:
        assert np.array_equal(m.debug_volume(1), st["S"]), "S differs"
        assert np.array_equal(m.debug_plane(3), st["disp_lr"]), "LR-checked map differs"
    assert np.array_equal(m.debug_plane(2)[:, col0:colN], st["mins"][:, col0:colN]), "minS differs"
    assert np.array_equal(m.debug_plane(1), st["raw"]), "raw index differs"
    assert np.array_equal(m.debug_plane(0), st["disp_wta"]), "WTA disparity differs"
    assert np.array_equal(got, ref), "final disparity16 differs in %d pixels" % int((got != ref).sum())
    assert (got[:, :col0] == inv).all() and (got[:, colN:] == inv).all()
    assert (got[:, col0:colN] != inv).mean() > 0.2
    m.close()


def test_min_disparity_batch_sweeps_and_param_change(gpu):
    """A batch through the fused sweeps with minD = 16, then sv_set_params to minD = 0 on the same context (the
    volumes change width: W1 = W - D - minD -> W - D)."""
    import torch
    W, H, D, nb = 240, 120, 64, 12
    pairs = [synth.stereo_pair(950 + i, W, H, D) for i in range(4)]
    lt = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(nb)])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    ph, po = _params(D, 8, 16)
    os.environ["SV_SWEEP_NCU"] = "6"
    try:
        m = svhip.Matcher(W, H, ph)
    finally:
        del os.environ["SV_SWEEP_NCU"]
    m.set_batch(nb)
    m.set_debug(3)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    for minD in (16, 0, -5):
        ph, po = _params(D, 8, minD)
        m.set_params(ph)
        m.compute_gray_batch_dev(nb, lt, rt, out)
        m.synchronize()
        assert m.last_aggregation() == (2, 2)
        got = out.cpu().numpy()
        for i in range(4):
            ref = svref.sgbm(pairs[i][0], pairs[i][1], po)
            for k in range(i, nb, 4):
                assert np.array_equal(got[k], ref), (minD, k)
    m.close()


def test_min_disparity_colour_cost_and_filters(gpu):
    """matchSGBM proper: colour cost, P1/P2 x3, minDisparity 10, speckle filter with its (minD - 1) * 16 marker, vis."""
    W, H, D, minD = 200, 80, 64, 10
    left, right, _ = synth.stereo_pair(77, W, H, D)
    rng = np.random.default_rng(11)
    lrgb, rrgb = synth.rgb_from_gray(rng, left), synth.rgb_from_gray(rng, right)
    ph, po = _params(D, 8, minD, speckle_window_size=50, speckle_range=16)
    for p in (ph, po):
        p.P1, p.P2 = 3 * p.P1, 3 * p.P2
        p.flags |= svhip.FLAG_COST_RGB
    m = svhip.Matcher(W, H, ph)
    got, vis = m.compute_rgb_vis(lrgb, rrgb, rectify=False)
    ref = svref.sgbm(lrgb, rrgb, po)
    assert np.array_equal(got, ref)
    assert np.array_equal(vis, svref.disparity_vis(ref, 3))
    assert (got == (minD - 1) * 16).any() and (got > minD * 16).any()
    m.close()


def test_min_disparity_out_of_range_rejected(gpu):
    with pytest.raises(svhip.SvError):
        svhip.Matcher(320, 240, svhip.default_params(64, 8, min_disparity=2000))
    with pytest.raises(svhip.SvError):
        svhip.Matcher(100, 50, svhip.default_params(64, 8, min_disparity=40))   # no computable column left
