"""GPU test of the wire formats either side of the path: the module hands the results to YARP images -
ImageOf<PixelMono> for the 8-bit view, ImageOf<PixelFloat> for depth (DispModule.cpp:757-760,794-798), ImageOf<PixelRgbFloat>
for the world images (SFM.cpp:1284-1347) - whose rows are padded to a multiple of 8 bytes.  At a width that is no multiple
of 8 (323) every *_host entry point that fills such an image is called through the C ABI with the padded row strides and
guard bytes around and between the rows: the payload must equal the oracle's, every padding and guard byte must stay as it
was.  (No YARP here: the buffers are laid out by hand the way yarp::sig::Image lays them out.)"""
import ctypes as C

import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu

GUARD = 0xA5


def _pad8(nbytes):
    return (nbytes + 7) // 8 * 8


class Padded:
    """H rows of `row_bytes` payload at a stride padded to 8 bytes, 64 guard bytes before and after."""

    def __init__(self, H, row_bytes, dtype, src=None):
        self.H, self.row_bytes, self.stride, self.dtype = H, row_bytes, _pad8(row_bytes), np.dtype(dtype)
        assert self.stride > row_bytes, "pick a width whose rows need padding"
        self.buf = np.full(64 + H * self.stride + 64, GUARD, np.uint8)
        if src is not None:
            self.rows()[:, :row_bytes] = np.ascontiguousarray(src).view(np.uint8).reshape(H, row_bytes)

    def rows(self):
        return self.buf[64:64 + self.H * self.stride].reshape(self.H, self.stride)

    def ptr(self):
        return C.c_void_p(self.buf.ctypes.data + 64)

    def payload(self, shape):
        return np.ascontiguousarray(self.rows()[:, :self.row_bytes]).view(self.dtype).reshape(shape)

    def untouched(self):
        return bool((self.buf[:64] == GUARD).all() and (self.buf[-64:] == GUARD).all() and (self.rows()[:, self.row_bytes:] == GUARD).all())


def test_yarp_padded_rows_through_the_c_abi(gpu):
    W, H, D = 323, 120, 64
    L = svhip.load()
    left, right, _ = synth.stereo_pair(4242, W, H, D)
    rng = np.random.default_rng(42)
    lrgb, rrgb = synth.rgb_from_gray(rng, left), synth.rgb_from_gray(rng, right)
    ph = svhip.default_params(D, 8, speckle_window_size=50, speckle_range=16)
    po = svref.Params()
    for f, _t in svhip.Params._fields_:
        setattr(po, f, getattr(ph, f))
    m = svhip.Matcher(W, H, ph)
    h = m._h

    def ok(rc):
        assert rc == 0, (rc, L.sv_last_error(h))

    # compute(): padded RGB rows in (ImageOf<PixelRgb>: 969 -> 976 bytes), disparity16 and the PixelMono view out
    pl, pr = Padded(H, 3 * W, np.uint8, lrgb), Padded(H, 3 * W, np.uint8, rrgb)
    d16, vis = Padded(H, 2 * W, np.int16), Padded(H, W, np.uint8)
    L.sv_compute_rgb_vis_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.c_int]
    ok(L.sv_compute_rgb_vis_host(h, pl.ptr(), pl.stride, pr.ptr(), pr.stride, 0, d16.ptr(), d16.stride, vis.ptr(), vis.stride, 3))
    e16 = svref.sgbm(svref.gray_from_rgb(lrgb), svref.gray_from_rgb(rrgb), po)
    assert np.array_equal(d16.payload((H, W)), e16) and np.array_equal(vis.payload((H, W)), svref.disparity_vis(e16, 3))
    assert d16.untouched() and vis.untouched() and pl.untouched() and pr.untouched()
    assert (e16[:, D:] >= 0).mean() > 0.5

    # filterBLF (BLF_ORIGINAL) and filterWLS: 16-bit map + view in, both out, all padded
    b16, b8 = Padded(H, 2 * W, np.int16), Padded(H, W, np.uint8)
    L.sv_filter_blf_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    ok(L.sv_filter_blf_host(h, d16.ptr(), d16.stride, 10.0, 10.0, b16.ptr(), b16.stride, b8.ptr(), b8.stride))
    eb16 = svref.blf(e16, 10.0, 10.0)
    assert np.array_equal(b16.payload((H, W)), eb16) and np.array_equal(b8.payload((H, W)), svref.disparity_vis(eb16, 3))
    assert b16.untouched() and b8.untouched() and d16.untouched()

    w16, w8 = Padded(H, 2 * W, np.int16), Padded(H, W, np.uint8)
    L.sv_filter_wls_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_size_t]
    ok(L.sv_filter_wls_host(h, b16.ptr(), b16.stride, b8.ptr(), b8.stride, pl.ptr(), pl.stride, w16.ptr(), w16.stride, w8.ptr(), w8.stride))
    ew16, ew8 = svref.wls_filter(eb16, lrgb, svref.wls_roi(W, H, po), 4500.0, 1.0, vis=svref.disparity_vis(eb16, 3))
    assert np.array_equal(w16.payload((H, W)), ew16) and np.array_equal(w8.payload((H, W)), ew8)
    assert w16.untouched() and w8.untouched() and b16.untouched() and b8.untouched() and pl.untouched()

    # depthFromDisparity into an ImageOf<PixelFloat> (1292 -> 1296 bytes per row), host mapper
    K, dist, R, P = synth.camera(W, H, 0)
    mapper = svref.undistort_grid(K, dist, R, P, W, H)
    fx, base = 0.7 * W, 0.068
    Q = np.array([[1, 0, 0, -(W / 2 - 0.5)], [0, 1, 0, -(H / 2 - 0.5)], [0, 0, 0, fx], [0, 0, 1 / base, 0]], np.float64)
    dep = Padded(H, 4 * W, np.float32)
    Rm = np.ascontiguousarray(R, np.float64)
    mp = np.ascontiguousarray(mapper, np.float32)
    L.sv_depth_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    ok(L.sv_depth_host(h, w16.ptr(), w16.stride, mp.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p), Rm.ctypes.data_as(C.c_void_p),
                       dep.ptr(), dep.stride))
    edep = svref.depth(ew16, mp, Q, Rm, alias=False)
    got = dep.payload((H, W))
    fin = np.isfinite(edep)
    assert np.array_equal(np.isfinite(got), fin)
    assert (np.abs(got[fin] - edep[fin]) <= 1e-4 * np.maximum(np.abs(edep[fin]), 1e-12)).all()   # same mapper on both sides: every pixel
    assert dep.untouched() and w16.untouched()

    # fillWorld3D into two ImageOf<PixelRgbFloat> (3876 -> 3880 bytes per row)
    from test_oracle_world import _scene
    _, _, Qw, Rw, Hroot = _scene(W, H, D, seed=12)
    cart, cyl = Padded(H, 12 * W, np.float32), Padded(H, 12 * W, np.float32)
    L.sv_world_image_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_size_t]
    Qw, Rw, Hroot = (np.ascontiguousarray(a, np.float64) for a in (Qw, Rw, Hroot))
    ok(L.sv_world_image_host(h, w16.ptr(), w16.stride, mp.ctypes.data_as(C.c_void_p), Qw.ctypes.data_as(C.c_void_p),
                             Rw.ctypes.data_as(C.c_void_p), Hroot.ctypes.data_as(C.c_void_p), cart.ptr(), cart.stride, cyl.ptr(), cyl.stride))
    ecart, ecyl = svref.world_image(ew16, mp, Qw, Rw, Hroot)
    assert np.array_equal(cart.payload((H, W, 3)), ecart) and np.allclose(cyl.payload((H, W, 3)), ecyl, rtol=2e-7, atol=0)
    assert cart.untouched() and cyl.untouched() and w16.untouched()
    m.close()
