"""svref.py — ctypes access to oracle/libsv_ref.so (the CPU oracle).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  Parity unpinned
(see oracle/sv_ref.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsv_ref.so")

INVALID = -16
PATHS_5, PATHS_8 = 0x1F, 0xFF
FLAG_C_COL0_P2, FLAG_C_TAIL_P2, FLAG_LR_OPENCV, FLAG_MEDIAN_REPLICATE = 1, 2, 4, 8
FLAG_NO_LRCHECK, FLAG_NO_MEDIAN = 0x10, 0x20
FLAG_COST_RGB = 0x40   # cost summed over R, G, B (cv::StereoSGBM on 8UC3 images: matchSGBM)
FLAGS_REPO = 0
FLAGS_HH = FLAG_C_COL0_P2 | FLAG_C_TAIL_P2 | FLAG_LR_OPENCV | FLAG_MEDIAN_REPLICATE


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "min_disparity", "num_disparities", "block_size", "P1", "P2", "pre_filter_cap",
        "uniqueness_ratio", "disp12_max_diff", "speckle_window_size", "speckle_range")] + [
        ("path_mask", C.c_uint32), ("flags", C.c_uint32), ("wls_lambda", C.c_float), ("wls_sigma", C.c_float)]


class Stages(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("grad_l", "grad_r", "raw_l", "raw_r", "pix", "hsum", "C")] + [
        ("L", C.c_void_p * 8)] + [(n, C.c_void_p) for n in (
            "S", "disp_wta", "raw", "mins", "disp2", "disp_lr", "disp_med")]


def default_params(D, paths=8, BS=7, uniq=15, disp12=0, cap=63, speckle_window=0, speckle_range=0, flags=None,
                   path_mask=None):
    p = Params()
    p.min_disparity, p.num_disparities, p.block_size = 0, D, BS
    p.P1, p.P2, p.pre_filter_cap = 8 * BS * BS, 32 * BS * BS, cap
    p.uniqueness_ratio, p.disp12_max_diff = uniq, disp12
    p.speckle_window_size, p.speckle_range = speckle_window, speckle_range
    p.path_mask = path_mask if path_mask is not None else (PATHS_8 if paths == 8 else PATHS_5)
    p.flags = flags if flags is not None else (FLAGS_HH if paths == 8 else FLAGS_REPO)
    p.wls_lambda, p.wls_sigma = 4500.0, 1.0
    return p


def build(force=False):
    src = os.path.join(_HERE, "sv_ref.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        if not os.path.exists(src):
            raise RuntimeError("oracle source missing")
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsv_ref.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.svref_version.restype = C.c_char_p
        L.svref_sgbm.restype = C.c_int
        L.svref_sgbm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                 C.POINTER(Params), C.c_void_p, C.c_size_t, C.POINTER(Stages)]
        L.svref_sgbm_rgb.restype = C.c_int
        L.svref_sgbm_rgb.argtypes = L.svref_sgbm.argtypes
        L.svref_check_params.argtypes = [C.POINTER(Params), C.c_int, C.c_int]
        L.svref_cost_volume.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_void_p,
                                        C.POINTER(Stages)]
        L.svref_aggregate_path.restype = None
        L.svref_aggregate_path.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
        L.svref_wta.restype = None
        L.svref_wta.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 3
        L.svref_median3.restype = None
        L.svref_median3.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.svref_filter_speckles.restype = None
        L.svref_filter_speckles.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.svref_disparity_vis.restype = None
        L.svref_disparity_vis.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.svref_gray_from_rgb.restype = None
        L.svref_gray_from_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.svref_remap_rgb.restype = None
        L.svref_remap_rgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.svref_depth.restype = None
        L.svref_depth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_void_p]
        L.svref_wls_lut.restype = None
        L.svref_wls_lut.argtypes = [C.c_float, C.c_void_p]
        L.svref_wls_roi.restype = None
        L.svref_wls_roi.argtypes = [C.c_int, C.c_int, C.POINTER(Params), C.POINTER(C.c_int * 4)]
        L.svref_fgs.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_int]
        L.svref_wls_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int * 4),
                                       C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.svref_world_image.restype = None
        L.svref_world_image.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.svref_blf_joint.restype = C.c_int
        L.svref_blf_joint.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.svref_blf.restype = C.c_int
        L.svref_blf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.svref_init_rectify_map.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.svref_undistort_grid.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_STAGE_SPECS = {
    "grad_l": ("px", np.uint8), "grad_r": ("px", np.uint8), "raw_l": ("px", np.uint8), "raw_r": ("px", np.uint8),
    "pix": ("vol", np.int16), "hsum": ("vol", np.int16), "C": ("vol", np.int16), "S": ("vol", np.int16),
    "disp_wta": ("px", np.int16), "raw": ("px", np.int16), "mins": ("px", np.int16), "disp2": ("px", np.int16),
    "disp_lr": ("px", np.int16), "disp_med": ("px", np.int16)}


def geometry(W, params):
    """(col0, colN, W1, invalid): computable columns [col0, colN), their count, the invalid marker (min_disparity, EXT)."""
    D, minD = params.num_disparities, params.min_disparity
    col0, colN = max(minD + D, 0), W + min(minD, 0)
    return col0, colN, colN - col0, (minD - 1) * 16


def sgbm(left, right, params, stages=(), path_L=False):
    """Run the oracle.  Returns disp16 [H][W] int16 (and a dict of requested stages)."""
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    rgb = left.ndim == 3          # [H][W][3]: matchSGBM semantics, needs FLAG_COST_RGB in params.flags
    H, W = left.shape[:2]
    D = params.num_disparities
    W1 = geometry(W, params)[2]
    out = np.empty((H, W), np.int16)
    st = Stages()
    keep = {}
    for name in stages:
        kind, dt = _STAGE_SPECS[name]
        arr = np.zeros((H, W) if kind == "px" else (H, W1, D), dt)
        keep[name] = arr
        setattr(st, name, _ptr(arr).value)
    if path_L:
        Ls = {}
        for k in range(8):
            if params.path_mask & (1 << k):
                Ls[k] = np.zeros((H, W1, D), np.int16)
                st.L[k] = _ptr(Ls[k]).value
        keep["L"] = Ls
    fn = lib().svref_sgbm_rgb if rgb else lib().svref_sgbm
    cn = 3 if rgb else 1
    rc = fn(_ptr(left), cn * W, _ptr(right), cn * W, W, H, C.byref(params), _ptr(out), 2 * W,
            C.byref(st) if (stages or path_L) else None)
    if rc != 0:
        raise ValueError("svref_sgbm failed: %d" % rc)
    return (out, keep) if (stages or path_L) else out


def cost_volume(left, right, params):
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    H, W = left.shape
    D = params.num_disparities
    Cv = np.empty((H, geometry(W, params)[2], D), np.int16)
    rc = lib().svref_cost_volume(_ptr(left), _ptr(right), W, H, C.byref(params), _ptr(Cv), None)
    if rc:
        raise ValueError("svref_cost_volume failed: %d" % rc)
    return Cv


def aggregate_path(Cv, P1, P2, direction, S=None):
    H, W1, D = Cv.shape
    Cv = np.ascontiguousarray(Cv, np.int16)
    if S is None:
        S = np.zeros_like(Cv)
    L = np.empty_like(Cv)
    lib().svref_aggregate_path(_ptr(Cv), W1, H, D, P1, P2, direction, _ptr(S), _ptr(L))
    return S, L


def wta(S, W, uniq):
    H, W1, D = S.shape
    S = np.ascontiguousarray(S, np.int16)
    disp, raw, mins = (np.empty((H, W), np.int16) for _ in range(3))
    lib().svref_wta(_ptr(S), W, H, D, uniq, _ptr(disp), _ptr(raw), _ptr(mins))
    return disp, raw, mins


def median3(img, replicate=False):
    img = np.ascontiguousarray(img, np.int16)
    out = np.empty_like(img)
    lib().svref_median3(_ptr(img), _ptr(out), img.shape[1], img.shape[0], int(replicate))
    return out


def filter_speckles(img, new_val, max_size, max_diff):
    out = np.ascontiguousarray(img, np.int16).copy()
    lib().svref_filter_speckles(_ptr(out), out.shape[1], out.shape[0], new_val, max_size, max_diff)
    return out


def disparity_vis(disp16, scale=3):
    d = np.ascontiguousarray(disp16, np.int16)
    out = np.empty(d.shape, np.uint8)
    lib().svref_disparity_vis(_ptr(d), _ptr(out), d.size, scale)
    return out


def gray_from_rgb(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    out = np.empty(rgb.shape[:2], np.uint8)
    lib().svref_gray_from_rgb(_ptr(rgb), _ptr(out), out.size)
    return out


def remap_rgb(src, mapx, mapy):
    src = np.ascontiguousarray(src, np.uint8)
    mapx = np.ascontiguousarray(mapx, np.float32)
    mapy = np.ascontiguousarray(mapy, np.float32)
    H, W = src.shape[:2]
    out = np.empty_like(src)
    lib().svref_remap_rgb(_ptr(src), W, H, _ptr(mapx), _ptr(mapy), _ptr(out))
    return out


def depth(disp16, mapper, Q, R, alias=False):
    d = np.ascontiguousarray(disp16, np.int16)
    H, W = d.shape
    mapper = np.ascontiguousarray(mapper, np.float32)
    Q = np.ascontiguousarray(Q, np.float64)
    R = np.ascontiguousarray(R, np.float64)
    out = np.empty((H, W), np.float32)
    lib().svref_depth(_ptr(d), W, H, _ptr(mapper), _ptr(Q), _ptr(R), int(alias), _ptr(out))
    return out


def wls_roi(W, H, params):
    """computeROI2 (StereoMatcher.cpp:12-29): (x, y, w, h)."""
    r = (C.c_int * 4)()
    lib().svref_wls_roi(W, H, C.byref(params), C.byref(r))
    return tuple(r)


def wls_lut(sigma):
    out = np.empty(3 * 256 * 256, np.float32)
    lib().svref_wls_lut(sigma, _ptr(out))
    return out


def fgs(guide_rgb, img, lam, sigma, att=0.25, iters=3):
    """fastGlobalSmootherFilter on a float32 image with an 8UC3 guide of the same size."""
    g = np.ascontiguousarray(guide_rgb, np.uint8)
    out = np.ascontiguousarray(img, np.float32).copy()
    h, w = out.shape
    assert g.shape == (h, w, 3)
    rc = lib().svref_fgs(_ptr(g), 3 * w, _ptr(out), w, h, lam, sigma, att, iters)
    if rc:
        raise ValueError("svref_fgs failed: %d" % rc)
    return out


def wls_filter(disp16, guide_rgb, roi, lam, sigma, vis=None):
    """filterWLS (StereoMatcher.cpp:154-160): returns disp16_wls (and vis_wls when vis is given)."""
    d = np.ascontiguousarray(disp16, np.int16)
    g = np.ascontiguousarray(guide_rgb, np.uint8)
    H, W = d.shape
    assert g.shape == (H, W, 3)
    out16 = np.empty((H, W), np.int16)
    v = np.ascontiguousarray(vis, np.uint8) if vis is not None else None
    out8 = np.empty((H, W), np.uint8) if vis is not None else None
    r = (C.c_int * 4)(*roi)
    rc = lib().svref_wls_filter(_ptr(d), _ptr(v) if v is not None else None, _ptr(g), W, H, C.byref(r), lam, sigma,
                                _ptr(out16), _ptr(out8) if out8 is not None else None)
    if rc:
        raise ValueError("svref_wls_filter failed: %d" % rc)
    return (out16, out8) if vis is not None else out16


def _cam(K, dist, R, P):
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
    d = np.zeros(8, np.float64)
    dd = np.asarray(dist, np.float64).ravel()
    d[:len(dd)] = dd
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    P = np.ascontiguousarray(P, np.float64).reshape(3, 4)
    return K, d, R, P


def init_rectify_map(K, dist, R, P, W, H):
    """cv::initUndistortRectifyMap(..., CV_32FC1) -> (mapx, mapy)."""
    K, d, R, P = _cam(K, dist, R, P)
    mx, my = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
    rc = lib().svref_init_rectify_map(_ptr(K), _ptr(d), _ptr(R), _ptr(P), W, H, _ptr(mx), _ptr(my))
    if rc:
        raise ValueError("svref_init_rectify_map failed: %d" % rc)
    return mx, my


def undistort_grid(K, dist, R, P, W, H):
    """cv::undistortPoints on the pixel grid (updateMappings) -> mapper [H][W][2]."""
    K, d, R, P = _cam(K, dist, R, P)
    out = np.empty((H, W, 2), np.float32)
    rc = lib().svref_undistort_grid(_ptr(K), _ptr(d), _ptr(R), _ptr(P), W, H, _ptr(out))
    if rc:
        raise ValueError("svref_undistort_grid failed: %d" % rc)
    return out


def blf(disp16, sigma_color, sigma_space):
    """filterBLF / BLF_ORIGINAL: cv_extend::bilateralFilter on a 16-bit disparity map."""
    src = np.ascontiguousarray(disp16, dtype=np.int16)
    dst = np.empty_like(src)
    if lib().svref_blf(_ptr(src), src.shape[1], src.shape[0], float(sigma_color), float(sigma_space), _ptr(dst)) != 0:
        raise ValueError("svref_blf: bad arguments")
    return dst


def world_image(disp16, mapper, Q, RLrect, HL_root):
    """SFM::fillWorld3D: (cartesian, cylindrical) [H][W][3] float images of the original left view."""
    d = np.ascontiguousarray(disp16, dtype=np.int16)
    H, W = d.shape
    mp = np.ascontiguousarray(mapper, dtype=np.float32)
    Qm, Rm, Hm = (np.ascontiguousarray(a, dtype=np.float64) for a in (Q, RLrect, HL_root))
    cart = np.empty((H, W, 3), np.float32)
    cyl = np.empty((H, W, 3), np.float32)
    lib().svref_world_image(_ptr(d), W, H, _ptr(mp), _ptr(Qm), _ptr(Rm), _ptr(Hm), _ptr(cart), _ptr(cyl))
    return cart, cyl


def blf_joint(disp16, guide_gray, ndisp, sigma_range=10.0, radius=7, iters=2):
    """filterBLF / BLF_CUDA: cv::cuda::DisparityBilateralFilter(ndisp, 7, 2) with a gray guide (EXT restatement)."""
    src = np.ascontiguousarray(disp16, dtype=np.int16)
    g = np.ascontiguousarray(guide_gray, dtype=np.uint8)
    assert g.shape == src.shape
    dst = np.empty_like(src)
    if lib().svref_blf_joint(_ptr(src), _ptr(g), src.shape[1], src.shape[0], ndisp, radius, iters, float(sigma_range), _ptr(dst)) != 0:
        raise ValueError("svref_blf_joint: bad arguments")
    return dst
