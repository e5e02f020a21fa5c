"""synth.py — synthetic rectified stereo pairs (SURVEY.md 8(d) "Synthetic inputs").

Left image: three octaves of box-blurred uniform noise rescaled to [16, 240];
ground-truth disparity: fronto-parallel rectangles over a slanted plane; right
image = left warped by -d with occlusions filled from the farther layer, plus
Gaussian noise sigma = 2.  Seeds: C1 101, C2 202, C3 303, C4 404, C5 500+stream.
"""
import numpy as np


def _box_blur(img, r):
    if r <= 0:
        return img
    k = 2 * r + 1
    p = np.pad(img, ((r, r), (r, r)), mode="reflect")
    c = np.cumsum(np.cumsum(p, axis=0, dtype=np.float64), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    s = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
    return s / (k * k)


def texture(rng, H, W):
    acc = np.zeros((H, W))
    for r, wgt in ((1, 1.0), (4, 1.5), (16, 2.0)):
        n = rng.integers(0, 256, size=(H, W)).astype(np.float64)
        b = _box_blur(n, r)
        b = (b - b.mean()) / (b.std() + 1e-9)
        acc += wgt * b
    acc = (acc - acc.min()) / (acc.max() - acc.min() + 1e-9)
    return 16.0 + acc * 224.0


def disparity_field(rng, H, W, D):
    yy, xx = np.mgrid[0:H, 0:W]
    a = rng.uniform(-0.02, 0.02) * D / max(W, 1) * 4
    b = rng.uniform(0.0, 0.25) * D / max(H, 1)
    c = rng.uniform(0.1, 0.3) * D
    d = np.clip(a * xx + b * yy + c, 2, D - 4)
    n_rect = int(rng.integers(6, 13))
    for _ in range(n_rect):
        h = int(rng.integers(max(H // 10, 2), max(H // 3, 3)))
        w = int(rng.integers(max(W // 10, 2), max(W // 3, 3)))
        y0 = int(rng.integers(0, max(H - h, 1)))
        x0 = int(rng.integers(0, max(W - w, 1)))
        dv = int(rng.integers(4, max(D - 8, 5)))
        d[y0:y0 + h, x0:x0 + w] = np.maximum(d[y0:y0 + h, x0:x0 + w], dv)
    return d


def stereo_pair(seed, W, H, D, noise=2.0):
    """Returns (left_u8, right_u8, gt_disparity_float) with shape [H][W]."""
    rng = np.random.default_rng(seed)
    left = texture(rng, H, W)
    d = disparity_field(rng, H, W, D)
    right = np.zeros((H, W))
    zbuf = np.full((H, W), -1.0)
    xs = np.arange(W)
    for y in range(H):
        xr = np.rint(xs - d[y]).astype(np.int64)
        ok = (xr >= 0) & (xr < W)
        order = np.argsort(d[y][ok], kind="stable")  # nearer (larger d) written last
        src = xs[ok][order]
        dst = xr[ok][order]
        right[y, dst] = left[y, src]
        zbuf[y, dst] = d[y, src]
        holes = zbuf[y] < 0
        if holes.any():
            # fill occlusion holes from the farther (left-hand) neighbour
            idx = np.where(~holes, np.arange(W), -1)
            idx = np.maximum.accumulate(idx)
            fill = idx.copy()
            fill[fill < 0] = np.argmax(~holes) if (~holes).any() else 0
            right[y, holes] = right[y, fill[holes]]
    left_n = left + rng.normal(0, noise, size=(H, W))
    right_n = right + rng.normal(0, noise, size=(H, W))
    return (np.clip(np.rint(left_n), 0, 255).astype(np.uint8), np.clip(np.rint(right_n), 0, 255).astype(np.uint8),
            d.astype(np.float32))


def rgb_from_gray(rng, gray):
    """Gray replicated x3 plus +-3 per-channel uniform noise (C1/C3 inputs)."""
    n = rng.integers(-3, 4, size=gray.shape + (3,))
    return np.clip(gray[..., None].astype(np.int64) + n, 0, 255).astype(np.uint8)


def rectify_maps(W, H, which=0):
    """Smooth synthetic undistort/rectify maps (CV_32FC1 pair) of the shape
    rectifyImages() would upload (stereoCamera.cpp:569-570): a mild radial
    distortion plus a small rotation, different for the two cameras."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    f = 0.7 * W
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    k1, k2 = (-0.05, 0.01) if which == 0 else (-0.045, 0.012)
    ang = 0.004 if which == 0 else -0.003
    xn, yn = (xx - cx) / f, (yy - cy) / f
    ca, sa = np.cos(ang), np.sin(ang)
    xr, yr = ca * xn - sa * yn, sa * xn + ca * yn
    r2 = xr * xr + yr * yr
    s = 1 + k1 * r2 + k2 * r2 * r2
    mx = xr * s * f + cx + (0.3 if which else -0.2)
    my = yr * s * f + cy
    return mx.astype(np.float32), my.astype(np.float32)


def rodrigues(v):
    """Rotation matrix of the rotation vector v (what cv::Rodrigues returns)."""
    v = np.asarray(v, np.float64)
    th = np.linalg.norm(v)
    if th == 0:
        return np.eye(3)
    k = v / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def camera(W, H, which=0):
    """Synthetic calibrated camera of SURVEY 8(d) (config C3): K with f = 0.7 W, radial + tangential
    distortion, a small rectifying rotation and the rectified projection P (right camera: baseline
    term in P[0][3], as cv::stereoRectify returns it)."""
    f = 0.7 * W
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    if which == 0:
        K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], np.float64)
        dist = np.array([-0.05, 0.01, 1e-3, 1e-3], np.float64)
        R = rodrigues([0.005, -0.01, 0.0025])
        P = np.array([[f, 0, cx, 0], [0, f, cy, 0], [0, 0, 1, 0]], np.float64)
    else:
        K = np.array([[f * 1.01, 0, cx + 2.0], [0, f * 1.008, cy - 1.5], [0, 0, 1]], np.float64)
        dist = np.array([-0.045, 0.012, -8e-4, 1.2e-3, 1e-3], np.float64)
        R = rodrigues([-0.005, 0.01, -0.0025])
        P = np.array([[f, 0, cx, -0.068 * f], [0, f, cy, 0], [0, 0, 1, 0]], np.float64)
    return K, dist, R, P


def rectified_coords(K, dist, R, P, W, H, iters=30):
    """For every pixel of a camera's RAW image, where it lies in the rectified image: the pixel is normalised with K, the
    distortion model (k1 k2 p1 p2 k3 k4 k5 k6) is inverted by fixed-point iteration, then R and the 3x3 part of P are
    applied - what cv::undistortPoints does for the mapper (stereoCamera.cpp:2462).  Test-scene construction only
    (numpy, float64): the product's mapper is k_undistort_grid, checked against the oracle's."""
    k = np.zeros(8)
    d = np.asarray(dist, np.float64).ravel()
    k[:d.size] = d
    k1, k2, p1, p2, k3, k4, k5, k6 = k
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    x0, y0 = (xx - K[0, 2]) / K[0, 0], (yy - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        ic = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * ic, (y0 - dy) * ic
    X = R[0, 0] * x + R[0, 1] * y + R[0, 2]
    Y = R[1, 0] * x + R[1, 1] * y + R[1, 2]
    Z = R[2, 0] * x + R[2, 1] * y + R[2, 2]
    return P[0, 0] * X / Z + P[0, 2], P[1, 1] * Y / Z + P[1, 2]


def sample_bilinear(img, mx, my):
    """img[H][W][C] u8 sampled at float coordinates (outside the image: 0), rounded to u8."""
    H, W = img.shape[:2]
    x0, y0 = np.floor(mx).astype(np.int64), np.floor(my).astype(np.int64)
    fx, fy = (mx - x0)[..., None], (my - y0)[..., None]
    src = img.astype(np.float64)

    def tap(yi, xi):
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        v = src[np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
        return v * ok[..., None]
    out = (tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx) * (1 - fy) + (tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def raw_pair(left_rgb, right_rgb, cams):
    """What the two cameras of `cams` (camera(W, H, 0 / 1)) would have seen for the row-aligned pair to come out of
    rectification: every raw pixel takes the colour of the point of the ideal rectified image it maps to (config C3,
    SURVEY 8(d)).  Rectifying the result (initUndistortRectifyMap + remap) gives the pair back up to two bilinear
    interpolations, so the matcher finds the scene's disparities in it - unlike the row-aligned pair fed through two
    different distortions, which it mostly cannot match."""
    H, W = left_rgb.shape[:2]
    out = []
    for img, (K, dist, R, P) in zip((left_rgb, right_rgb), cams):
        mx, my = rectified_coords(K, dist, R, P, W, H)
        out.append(sample_bilinear(img, mx, my))
    return out[0], out[1]
