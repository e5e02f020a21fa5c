"""GPU parity of rectification-map generation (cv::initUndistortRectifyMap, stereoCamera.cpp:569-570)
and of the inverse mapper (cv::undistortPoints on the pixel grid, stereoCamera.cpp:2462-2463),
SURVEY 8(f) rank 1, through the C ABI.  Floating point: the contract is remap <= 1 LSB and depth
<= 1e-4 relative; the maps themselves are compared at <= 2e-4 px (float32 spacing at 640 is 6e-5)."""
import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu
MAP_TOL = 2e-4


@pytest.mark.parametrize("W,H", [(320, 240), (640, 480), (1920, 1080), (97, 31)])
def test_maps_and_mapper_match_oracle(gpu, W, H):
    m = svhip.Matcher(W, H, svhip.default_params(32, 8))
    with pytest.raises(svhip.SvError) as e:
        m.maps(0)
    assert e.value.code == -5
    for which in (0, 1):
        K, dist, R, P = synth.camera(W, H, which)
        m.init_rectify_map(which, K, dist, R, P)
        m.update_mapper(which, K, dist, R, P)
        mx, my = m.maps(which)
        ex, ey = svref.init_rectify_map(K, dist, R, P, W, H)
        assert np.abs(mx - ex).max() <= MAP_TOL and np.abs(my - ey).max() <= MAP_TOL
        assert np.array_equal(mx, ex) and np.array_equal(my, ey)  # same FP64 order: in fact bit-exact
        mp = m.mapper(which)
        emp = svref.undistort_grid(K, dist, R, P, W, H)
        assert np.abs(mp - emp).max() <= MAP_TOL
        assert (mp == emp).mean() > 0.999
    m.close()


def test_bad_camera_arguments(gpu):
    m = svhip.Matcher(96, 40, svhip.default_params(32, 8))
    K, dist, R, P = synth.camera(96, 40, 0)
    with pytest.raises(svhip.SvError):
        m.init_rectify_map(0, K, np.zeros(12), R, P)      # thin-prism model not supported
    with pytest.raises(svhip.SvError):
        m.init_rectify_map(0, K, dist, R, np.zeros((3, 4)))  # singular P*R
    with pytest.raises(svhip.SvError):
        m.init_rectify_map(2, K, dist, R, P)
    m.close()


def test_full_pipeline_with_generated_maps(gpu):
    """C3: camera matrices in -> maps on the device -> remap -> gray -> SGM -> WLS -> depth with the
    device-resident MapperL; every stage against the oracle fed with the oracle's own maps."""
    W, H, D = 320, 240, 64
    left, right, _ = synth.stereo_pair(303, W, H, D)
    rng = np.random.default_rng(3)
    lrgb, rrgb = synth.rgb_from_gray(rng, left), synth.rgb_from_gray(rng, right)
    ph = svhip.default_params(D, 8, speckle_window_size=50, speckle_range=16)
    po = svref.Params()
    for f, _t in svhip.Params._fields_:
        setattr(po, f, getattr(ph, f))
    m = svhip.Matcher(W, H, ph)
    cams = [synth.camera(W, H, i) for i in (0, 1)]
    lrgb, rrgb = synth.raw_pair(lrgb, rrgb, cams)   # what the two cameras see of the row-aligned scene
    for i, cam in enumerate(cams):
        m.init_rectify_map(i, *cam)
    m.update_mapper(0, *cams[0])
    got = m.compute_rgb(lrgb, rrgb, rectify=True)
    maps = [svref.init_rectify_map(*cam, W, H) for cam in cams]
    rl, rr = svref.remap_rgb(lrgb, *maps[0]), svref.remap_rgb(rrgb, *maps[1])
    assert np.abs(m.rectified(0).astype(int) - rl).max() <= 1 and np.abs(m.rectified(1).astype(int) - rr).max() <= 1
    assert np.array_equal(m.rectified(0), rl) and np.array_equal(m.rectified(1), rr)
    exp = svref.sgbm(svref.gray_from_rgb(rl), svref.gray_from_rgb(rr), po)
    assert np.array_equal(got, exp)
    w16 = m.filter_wls(got)  # guide = rectified left image on the device
    assert np.array_equal(w16, svref.wls_filter(exp, rl, svref.wls_roi(W, H, po), 4500.0, 1.0))
    f, b = 0.7 * W, 0.068
    Q = np.array([[1, 0, 0, -(W / 2 - 0.5)], [0, 1, 0, -(H / 2 - 0.5)], [0, 0, 0, f], [0, 0, 1 / b, 0]], np.float64)
    dep = m.depth(w16, None, Q, cams[0][2])
    assert (got[:, D:] >= 0).mean() > 0.9, "rectification should give a pair the matcher resolves"
    mapper = svref.undistort_grid(*cams[0], W, H)
    edep = svref.depth(w16, mapper, Q, cams[0][2], alias=False)
    # every pixel <= 1e-4 relative, except where the oracle's mapper lies within 2e-4 px (the GPU mapper's tolerance) of a
    # rounding boundary x.5 and cvRound may pick the neighbouring rectified pixel: named here, under 0.1 % of the image
    frac = mapper - np.floor(mapper)
    exempt = (np.abs(frac - 0.5) < 2e-4).any(-1)
    assert exempt.mean() < 1e-3
    fin = np.isfinite(edep)
    assert np.array_equal(np.isfinite(dep)[~exempt], fin[~exempt])
    ok = fin & ~exempt
    rel = np.abs(dep[ok] - edep[ok]) / np.maximum(np.abs(edep[ok]), 1e-12)
    assert rel.max() <= 1e-4, (rel.max(), int((rel > 1e-4).sum()))
    assert (dep[fin] != 0).mean() > 0.5
    m.close()


def test_batched_full_pipeline_matches_single_frame_calls(gpu):
    """sv_compute_rgb_batch_dev + sv_disparity_vis_batch_dev + sv_filter_wls_dev + sv_depth_batch_dev on three
    device-resident frames of one camera pair == the single-frame host entry points frame by frame
    (which the test above checks against the oracle), for the gray and for the colour cost."""
    import torch
    W, H, D, nb = 320, 240, 64, 3
    rng = np.random.default_rng(4)
    pairs = [synth.stereo_pair(310 + i, W, H, D) for i in range(nb)]
    lrgb = np.stack([synth.rgb_from_gray(rng, p[0]) for p in pairs])
    rrgb = np.stack([synth.rgb_from_gray(rng, p[1]) for p in pairs])
    cams = [synth.camera(W, H, i) for i in (0, 1)]
    f, b = 0.7 * W, 0.068
    Q = np.array([[1, 0, 0, -(W / 2 - 0.5)], [0, 1, 0, -(H / 2 - 0.5)], [0, 0, 0, f], [0, 0, 1 / b, 0]], np.float64)
    for colour in (False, True):
        ph = svhip.default_params(D, 8)
        if colour:
            ph.P1, ph.P2 = 3 * ph.P1, 3 * ph.P2
            ph.flags |= svhip.FLAG_COST_RGB
        m = svhip.Matcher(W, H, ph)
        for i, cam in enumerate(cams):
            m.init_rectify_map(i, *cam)
        m.update_mapper(0, *cams[0])
        m.set_batch(nb)
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        lt, rt = torch.from_numpy(lrgb).cuda(), torch.from_numpy(rrgb).cuda()
        rect = torch.empty((nb, H, W, 3), dtype=torch.uint8, device="cuda")
        d16 = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
        vis = torch.empty((nb, H, W), dtype=torch.uint8, device="cuda")
        w16, w8 = torch.empty_like(d16), torch.empty_like(vis)
        dep = torch.empty((nb, H, W), dtype=torch.float32, device="cuda")
        m.compute_rgb_batch_dev(nb, lt, rt, d16, rectify=True, rect_left_t=rect)
        m.disparity_vis_batch_dev(nb, d16, vis, 3)
        m.filter_wls_dev(nb, d16, rect, w16, vis, w8)
        m.depth_batch_dev(nb, w16, Q, cams[0][2], dep)
        torch.cuda.synchronize()
        got = [t.cpu().numpy() for t in (rect, d16, vis, w16, w8, dep)]
        for i in range(nb):
            e16 = m.compute_rgb(lrgb[i], rrgb[i], rectify=True)
            assert np.array_equal(got[0][i], m.rectified(0)), (colour, i)
            assert np.array_equal(got[1][i], e16), (colour, i)
            e8 = m.disparity_vis(e16, 3)
            assert np.array_equal(got[2][i], e8)
            ew16, ew8 = m.filter_wls(e16, vis=e8)
            assert np.array_equal(got[3][i], ew16) and np.array_equal(got[4][i], ew8)
            assert np.array_equal(got[5][i], m.depth(ew16, None, Q, cams[0][2]), equal_nan=True)
        m.close()
