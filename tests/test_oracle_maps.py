"""Pins of the oracle's map generation (EXT cv::initUndistortRectifyMap / cv::undistortPoints, parity
unpinned: OpenCV absent): identity camera = identity grid, pure-radial closed form, and the two
functions (forward distortion vs iterative inverse, written separately) are mutual inverses."""
import numpy as np
from scipy.ndimage import map_coordinates

import svref
from svhip import synth


def test_identity_camera_gives_identity_grid():
    W, H = 64, 48
    K = np.array([[50.0, 0, 31.5], [0, 50.0, 23.5], [0, 0, 1]])
    P = np.hstack([K, np.zeros((3, 1))])
    mx, my = svref.init_rectify_map(K, np.zeros(4), np.eye(3), P, W, H)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    assert np.abs(mx - xx).max() < 1e-4 and np.abs(my - yy).max() < 1e-4
    mp = svref.undistort_grid(K, np.zeros(4), np.eye(3), P, W, H)
    assert np.abs(mp[..., 0] - xx).max() < 1e-4 and np.abs(mp[..., 1] - yy).max() < 1e-4


def test_pure_radial_closed_form():
    W, H, f = 80, 60, 70.0
    K = np.array([[f, 0, 39.5], [0, f, 29.5], [0, 0, 1]])
    P = np.hstack([K, np.zeros((3, 1))])
    k1, k2 = -0.2, 0.05
    mx, my = svref.init_rectify_map(K, [k1, k2, 0, 0], np.eye(3), P, W, H)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    x, y = (xx - 39.5) / f, (yy - 29.5) / f
    r2 = x * x + y * y
    s = 1 + k1 * r2 + k2 * r2 * r2
    assert np.abs(mx - (x * s * f + 39.5)).max() < 1e-4 and np.abs(my - (y * s * f + 29.5)).max() < 1e-4


def test_forward_map_and_inverse_mapper_are_inverses():
    W, H = 320, 240
    for which in (0, 1):
        K, dist, R, P = synth.camera(W, H, which)
        mx, my = svref.init_rectify_map(K, dist, R, P, W, H)
        mp = svref.undistort_grid(K, dist, R, P, W, H)
        yy, xx = np.mgrid[0:H, 0:W]
        ok = (mx > 1) & (mx < W - 2) & (my > 1) & (my < H - 2)
        bx = map_coordinates(mp[..., 0], [my, mx], order=1)
        by = map_coordinates(mp[..., 1], [my, mx], order=1)
        assert ok.mean() > 0.9
        assert np.abs(bx - xx)[ok].max() < 2e-3 and np.abs(by - yy)[ok].max() < 2e-3
