"""GPU parity of the WLS post-filter (filterWLS, StereoMatcher.cpp:127-198) through the C ABI.
The kernels keep the oracle's float operation order, so the rounded outputs are compared bit for
bit (the contract only asks for <= 1 LSB of disparity16)."""
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


@pytest.mark.parametrize("W,H,D,bs,lam,sig", [
    (320, 240, 64, 7, 4500.0, 1.0),    # module defaults (DispModule.cpp:391-393) at C1 size
    (203, 77, 32, 5, 8000.0, 1.5),     # ragged ROI (not a multiple of the 64-row / 64-column tiles)
    (640, 480, 128, 7, 4500.0, 1.0),   # C2/C3 geometry
    (150, 66, 32, 3, 0.0, 2.0),        # lambda 0: identity inside the ROI
])
def test_wls_matches_oracle(gpu, W, H, D, bs, lam, sig):
    left, right, _ = synth.stereo_pair(900 + W, W, H, D)
    rng = np.random.default_rng(W)
    guide = synth.rgb_from_gray(rng, left)
    ph, po = _params(D, 8, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs, wls_lambda=lam, wls_sigma=sig)
    m = svhip.Matcher(W, H, ph)
    d = m.compute_gray(left, right)
    vis = m.disparity_vis(d, 3)
    roi = m.wls_roi()
    assert roi == svref.wls_roi(W, H, po)
    got16, got8 = m.filter_wls(d, guide, vis=vis)
    exp16, exp8 = svref.wls_filter(d, guide, roi, lam, sig, vis=vis)
    assert np.abs(got16.astype(int) - exp16).max() <= 1, "beyond the 1 LSB contract"
    assert np.array_equal(got16, exp16) and np.array_equal(got8, exp8)
    # 16-bit map alone (no visualisation) gives the same result
    assert np.array_equal(m.filter_wls(d, guide), exp16)
    if lam > 0:
        x, y, w, h = roi
        assert (got16[y:y + h, x:x + w] != d[y:y + h, x:x + w]).mean() > 0.05  # it did filter
    m.close()


def test_wls_batched_and_rectified_guide(gpu):
    """Device entry point on a batch; host entry point with the guide kept by compute_rgb."""
    import torch
    W, H, D, nb = 200, 80, 32, 3
    ph, po = _params(D, 8)
    rng = np.random.default_rng(5)
    pairs = [synth.stereo_pair(40 + i, W, H, D) for i in range(nb)]
    guides = np.stack([synth.rgb_from_gray(rng, p[0]) for p in pairs])
    m = svhip.Matcher(W, H, ph)
    disps = np.stack([m.compute_gray(p[0], p[1]) for p in pairs])
    dt, gt = torch.from_numpy(disps).cuda(), torch.from_numpy(guides).cuda()
    out = torch.empty_like(dt)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.filter_wls_dev(nb, dt, gt, out)
    torch.cuda.synchronize()
    roi = m.wls_roi()
    for i in range(nb):
        assert np.array_equal(out[i].cpu().numpy(), svref.wls_filter(disps[i], guides[i], roi, 4500.0, 1.0)), i
    # full RGB pipeline, then WLS guided by the rectified left image held on the device
    lrgb, rrgb = guides[0], synth.rgb_from_gray(rng, pairs[0][1])
    d = m.compute_rgb(lrgb, rrgb, rectify=False)
    assert np.array_equal(m.filter_wls(d), svref.wls_filter(d, lrgb, roi, 4500.0, 1.0))
    m.close()


def test_wls_large_batch_uses_the_tile_row_pass(gpu):
    """A launch with more than four waves per CU switches the row pass from the lane-per-row kernel to the LDS-tile
    kernel (k_fgs_hpass_tiles): 300 frames of a ragged ROI (not a multiple of the 64x64 tiles), 16-bit map + vis,
    every frame against the oracle's result for its pair."""
    import torch
    W, H, D, nb, nd = 173, 101, 32, 300, 6
    ph, po = _params(D, 8)
    rng = np.random.default_rng(9)
    pairs = [synth.stereo_pair(60 + i, W, H, D) for i in range(nd)]
    guides = [synth.rgb_from_gray(rng, p[0]) for p in pairs]
    m = svhip.Matcher(W, H, ph)
    disps = [m.compute_gray(p[0], p[1]) for p in pairs]
    viss = [m.disparity_vis(d, 3) for d in disps]
    dt = torch.from_numpy(np.stack([disps[i % nd] for i in range(nb)])).cuda()
    vt = torch.from_numpy(np.stack([viss[i % nd] for i in range(nb)])).cuda()
    gt = torch.from_numpy(np.stack([guides[i % nd] for i in range(nb)])).cuda()
    o16, o8 = torch.empty_like(dt), torch.empty_like(vt)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.filter_wls_dev(nb, dt, gt, o16, vt, o8)
    torch.cuda.synchronize()
    roi = m.wls_roi()
    assert ((roi[3] + 63) // 64) * nb * 2 > 4 * 256   # really the tile kernel
    got16, got8 = o16.cpu().numpy(), o8.cpu().numpy()
    for i in range(nd):
        e16, e8 = svref.wls_filter(disps[i], guides[i], roi, 4500.0, 1.0, vis=viss[i])
        for k in range(i, nb, nd):
            assert np.array_equal(got16[k], e16) and np.array_equal(got8[k], e8), k
    m.close()
