"""GPU parity of filterBLF / BLF_ORIGINAL (sv_filter_blf_*, StereoMatcher.cpp:90-99,
fastBilateral.hpp:92-189) against the CPU oracle: FP64 in the reference's order of operations, so
the 16-bit result is compared bit-exact (tolerance 0)."""
import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu


def _disp(m, seed, W, H, D):
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    return m.compute_gray(left, right)


@pytest.mark.parametrize("W,H,D,sc,ss", [(160, 96, 64, 10.0, 10.0), (131, 57, 32, 3.5, 7.25), (96, 40, 32, 30.0, 1.0),
                                         (200, 64, 64, 1.0, 30.0), (640, 480, 128, 10.0, 10.0)])
def test_blf_bit_exact(gpu, W, H, D, sc, ss):
    m = svhip.Matcher(W, H, svhip.default_params(D, 8))
    d = _disp(m, 7, W, H, D)
    exp = svref.blf(d, sc, ss)
    got, vis = m.filter_blf(d, sc, ss, vis=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(vis, svref.disparity_vis(exp, 3))
    assert (got != d).mean() > 0.05            # it did filter
    assert np.array_equal(m.filter_blf(d, sc, ss), exp)   # without the 8-bit output, and repeatable
    m.close()


def test_blf_device_entry_in_place_and_errors(gpu):
    import torch
    W, H, D = 200, 90, 64
    m = svhip.Matcher(W, H, svhip.default_params(D, 8))
    d = _disp(m, 8, W, H, D)
    exp = svref.blf(d, 10.0, 10.0)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    t = torch.from_numpy(d.copy()).cuda()
    v = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    m.filter_blf_dev(t, t, 10.0, 10.0, v)      # output over the input
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), exp)
    assert np.array_equal(v.cpu().numpy(), svref.disparity_vis(exp, 3))
    c = np.full((H, W), 16 * 21 + 3, np.int16)
    assert np.array_equal(m.filter_blf(c), c)  # constant map: fixed point; value range 0 -> grid depth 5
    with pytest.raises(svhip.SvError):
        m.filter_blf(d, 0.0, 10.0)
    with pytest.raises(svhip.SvError) as e:
        m.filter_blf(np.arange(H * W, dtype=np.int16).reshape(H, W) * 5, 0.001, 1.0)   # grid far too large
    assert e.value.code == -2   # SV_ERR_UNSUPPORTED
    m.close()


@pytest.mark.parametrize("W,H,D,sigma,radius,iters", [(160, 96, 64, 10.0, 7, 2), (131, 57, 32, 3.0, 3, 1), (640, 480, 128, 10.0, 7, 2)])
def test_blf_joint_bit_exact(gpu, W, H, D, sigma, radius, iters):
    """BLF_CUDA semantics (sv_filter_blf_joint_*) against the oracle's deterministic restatement: same
    float operations in the same order, so the comparison is exact (tolerance 0)."""
    left, right, _ = synth.stereo_pair(9, W, H, D)
    rng = np.random.default_rng(5)
    lrgb = synth.rgb_from_gray(rng, left)
    m = svhip.Matcher(W, H, svhip.default_params(D, 8))
    d = m.compute_gray(left, right)
    gray = svref.gray_from_rgb(lrgb)
    exp = svref.blf_joint(d, gray, D, sigma, radius, iters)
    got, vis = m.filter_blf_joint(d, D, sigma, guide_rgb=lrgb, radius=radius, iters=iters, vis=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(vis, svref.disparity_vis(exp, 3))
    assert 0.0 < (got != d).mean() < 0.5
    m.close()


def test_blf_joint_device_entry_and_default_guide(gpu):
    import torch
    W, H, D = 200, 90, 64
    left, right, _ = synth.stereo_pair(10, W, H, D)
    rng = np.random.default_rng(6)
    lrgb, rrgb = synth.rgb_from_gray(rng, left), synth.rgb_from_gray(rng, right)
    m = svhip.Matcher(W, H, svhip.default_params(D, 8))
    with pytest.raises(svhip.SvError):
        m.filter_blf_joint(np.zeros((H, W), np.int16), D)          # no guide yet
    d = m.compute_rgb(lrgb, rrgb, rectify=False)
    gray = svref.gray_from_rgb(lrgb)
    exp = svref.blf_joint(d, gray, D, 10.0)
    assert np.array_equal(m.filter_blf_joint(d, D), exp)           # guide = the image kept by compute_rgb
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    t = torch.from_numpy(d.copy()).cuda()
    m.filter_blf_joint_dev(t, torch.from_numpy(gray).cuda(), t, D)  # in place
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), exp)
    with pytest.raises(svhip.SvError):
        m.filter_blf_joint(d, D, sigma_range=0.0)
    m.close()
