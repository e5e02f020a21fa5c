"""CPU tests of the oracle's world image (SFM::fillWorld3D, SFM.cpp:1284-1347, with the point
arithmetic of DispModule::get3DPoints, DispModule.cpp:1103-1160).  No fixture exists in the reference
(parity unpinned): checked against an array-at-a-time numpy restatement and against the depth image."""
import numpy as np

import svref
from svhip import synth


def _scene(W=160, H=96, D=64, seed=9):
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    d = svref.sgbm(left, right, svref.default_params(D, 8))
    mx, my = synth.rectify_maps(W, H, 0)
    mapper = np.stack([mx, my], -1).astype(np.float32)
    f, b = 0.7 * W, 0.068
    Q = np.array([[1, 0, 0, -(W / 2 - 0.5)], [0, 1, 0, -(H / 2 - 0.5)], [0, 0, 0, f], [0, 0, 1 / b, 0]], np.float64)
    R = synth.rodrigues(np.array([0.01, -0.02, 0.005]))
    Hroot = np.eye(4)
    Hroot[:3, :3] = synth.rodrigues(np.array([0.3, -0.1, 0.2]))
    Hroot[:3, 3] = [0.05, -0.03, 0.4]
    return d, mapper, Q, R, Hroot


def world_numpy(d, mapper, Q, R, Hroot):
    H, W = d.shape
    us, vs = mapper[..., 0], mapper[..., 1]
    u_, v_ = np.rint(us).astype(np.int64), np.rint(vs).astype(np.int64)
    inside = (u_ >= 0) & (u_ < W) & (v_ >= 0) & (v_ < H)
    disp = d[np.clip(v_, 0, H - 1), np.clip(u_, 0, W - 1)].astype(np.float64) / 16.0
    w = disp * Q[3, 2] + Q[3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        x = ((us.astype(np.float64) + 1) * Q[0, 0] + Q[0, 3]) / w
        y = ((vs.astype(np.float64) + 1) * Q[1, 1] + Q[1, 3]) / w
        z = Q[2, 3] / w
    keep = inside & ~((z > 10.0) | (z < 0.0))
    Hr = np.eye(4)
    Hr[:3, :3] = R.T
    Hr = Hroot @ Hr
    P = np.stack([x, y, z, np.ones_like(x)], -1) @ Hr.T
    cart = np.where(keep[..., None], P[..., :3], 0.0).astype(np.float32)
    cyl = np.stack([np.hypot(P[..., 0], P[..., 1]), np.arctan2(P[..., 1], P[..., 0]), P[..., 2]], -1)
    return cart, np.where(keep[..., None], cyl, 0.0).astype(np.float32), keep


def test_world_image_two_restatements_agree():
    d, mapper, Q, R, Hroot = _scene()
    cart, cyl = svref.world_image(d, mapper, Q, R, Hroot)
    ecart, ecyl, keep = world_numpy(d, mapper, Q, R, Hroot)
    assert 0.3 < keep.mean() < 1.0
    assert np.allclose(cart, ecart, rtol=1e-6, atol=1e-7) and np.allclose(cyl, ecyl, rtol=1e-6, atol=1e-7)
    assert not cart[~keep].any() and not cyl[~keep].any()


def test_world_image_against_depth_image():
    """Identity root pose and rectification rotation: cart z is depthFromDisparity's value
    (DispModule.cpp:320-366: R02 = R12 = 0, R22 = 1 -> Q23 / (d*Q32 + Q33)) wherever 0 <= z <= 10."""
    d, mapper, Q, _, _ = _scene()
    cart, cyl = svref.world_image(d, mapper, Q, np.eye(3), np.eye(4))
    depth = svref.depth(d, mapper, Q, np.eye(3), alias=False)
    keep = cart[..., 2] != 0
    assert keep.mean() > 0.3
    assert np.allclose(cart[..., 2][keep], depth[keep], rtol=1e-5)
    assert np.allclose(cyl[..., 0], np.hypot(cart[..., 0], cart[..., 1]), rtol=1e-6, atol=1e-7)
    # mapper entries outside the image leave zeros
    mp = mapper.copy()
    mp[:5] = -7.0
    c2, y2 = svref.world_image(d, mp, Q, np.eye(3), np.eye(4))
    assert not c2[:5].any() and not y2[:5].any()
