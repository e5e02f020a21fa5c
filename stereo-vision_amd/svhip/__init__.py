"""svhip — Python plumbing over the libsv_hip C ABI (include/sv_hip.h).

The product is libsv_hip.so (hand-written gfx950 kernels behind a C ABI) plus the
C++ host shim in ../host; this module only binds the ABI with ctypes so that the
parity tests and bench.py can drive it, and so that PyTorch can own device
memory and streams.  There is NO CPU fallback: if the shared library is missing
or no gfx950 device is present, every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVHIP_LIB") or os.path.join(os.path.dirname(_HERE), "libsv_hip.so")  # SVHIP_LIB: diagnostic builds

INVALID = -16
PATH_R, PATH_DR, PATH_D, PATH_DL, PATH_L, PATH_UL, PATH_U, PATH_UR = (1 << i for i in range(8))
PATHS_5, PATHS_8 = 0x1F, 0xFF
FLAG_C_COL0_P2, FLAG_C_TAIL_P2, FLAG_LR_OPENCV, FLAG_MEDIAN_REPLICATE = 1, 2, 4, 8
FLAG_NO_LRCHECK, FLAG_NO_MEDIAN = 0x10, 0x20
FLAG_COST_RGB = 0x40   # cost summed over R, G, B (cv::StereoSGBM on 8UC3 images: matchSGBM)
FLAGS_REPO = 0
FLAGS_HH = FLAG_C_COL0_P2 | FLAG_C_TAIL_P2 | FLAG_LR_OPENCV | FLAG_MEDIAN_REPLICATE
N_TIMINGS = 6
TIMING_NAMES = ("prefilter", "cost", "aggregation", "wta", "tail", "total")

EXPORTS = (
    "sv_default_params", "sv_create", "sv_destroy", "sv_set_params", "sv_get_params", "sv_set_stream",
    "sv_compute_gray_host", "sv_submit_gray_host", "sv_wait_host", "sv_submit_gray_batch_host", "sv_wait_batch_host", "sv_host_alloc", "sv_host_free", "sv_compute_gray_dev", "sv_set_batch", "sv_compute_gray_batch_dev", "sv_gray_from_rgb_dev", "sv_upload_maps_host",
    "sv_remap_rgb_dev", "sv_compute_rgb_host", "sv_get_rectified_host", "sv_disparity_vis_dev",
    "sv_disparity_vis_host", "sv_filter_speckles_dev", "sv_depth_host", "sv_depth_dev",
    "sv_wls_roi", "sv_filter_wls_dev", "sv_filter_wls_host", "sv_filter_blf_dev", "sv_filter_blf_host", "sv_filter_blf_joint_dev", "sv_filter_blf_joint_host", "sv_compute_rgb_batch_dev", "sv_disparity_vis_batch_dev", "sv_depth_batch_dev", "sv_world_image_dev", "sv_world_image_host",
    "sv_synchronize", "sv_last_aggregation", "sv_init_rectify_map", "sv_update_mapper", "sv_get_maps_host", "sv_get_mapper_host",
    "sv_set_debug", "sv_debug_read_volume_host", "sv_debug_read_plane_host", "sv_enable_timing", "sv_get_timings",
    "sv_algorithmic_bytes", "sv_last_error", "sv_version", "sv_get_aggregation_timings", "sv_filter_speckles_static_dev", "sv_last_sweep_waves", "sv_debug_sweep_order", "sv_debug_selftest_min3", "sv_compute_rgb_vis_host",
    "sv_pool_create", "sv_pool_destroy", "sv_pool_size", "sv_pool_plan", "sv_pool_compute_gray_host", "sv_pool_last_error")
STATIC_MAX_PIXELS = 1920 * 1088   # SV_STATIC_MAX_PIXELS


class Params(C.Structure):
    """Mirror of struct sv_params (include/sv_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "min_disparity", "num_disparities", "block_size", "P1", "P2", "pre_filter_cap",
        "uniqueness_ratio", "disp12_max_diff", "speckle_window_size", "speckle_range")] + [
        ("path_mask", C.c_uint32), ("flags", C.c_uint32), ("wls_lambda", C.c_float), ("wls_sigma", C.c_float)]

    def copy(self):
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q


class SvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsv_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """dlopen libsv_hip.so and declare the ABI.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsv_hip.so not built (%s): run `make -C stereo-vision_amd/csrc` or "
                           "__graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64; importing torch first makes
    # libsv_hip's dependency resolve to that already-loaded runtime (same soname) so that device
    # pointers, streams and devices are shared.  Loading the system runtime first and torch's
    # second leaves the later one without devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    L.sv_version.restype = C.c_char_p
    L.sv_last_error.restype = C.c_char_p
    L.sv_last_error.argtypes = [vp]
    L.sv_default_params.argtypes = [C.POINTER(Params), i32, i32]
    L.sv_create.argtypes = [C.POINTER(vp), i32, i32, i32, C.POINTER(Params)]
    L.sv_destroy.argtypes = [vp]
    L.sv_set_params.argtypes = [vp, C.POINTER(Params)]
    L.sv_get_params.argtypes = [vp, C.POINTER(Params)]
    L.sv_set_stream.argtypes = [vp, vp]
    L.sv_compute_gray_host.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    L.sv_submit_gray_host.argtypes = [vp, vp, sz, vp, sz, C.POINTER(C.c_int)]
    L.sv_wait_host.argtypes = [vp, i32, vp, sz]
    L.sv_submit_gray_batch_host.argtypes = [vp, i32, C.POINTER(vp), sz, C.POINTER(vp), sz, C.POINTER(vp), sz, C.POINTER(C.c_int)]
    L.sv_wait_batch_host.argtypes = [vp, i32]
    L.sv_host_alloc.argtypes = [C.POINTER(vp), sz]
    L.sv_host_free.argtypes = [vp]
    L.sv_compute_gray_dev.argtypes = [vp, vp, vp, vp]
    L.sv_set_batch.argtypes = [vp, i32]
    L.sv_compute_gray_batch_dev.argtypes = [vp, i32, vp, vp, vp]
    L.sv_gray_from_rgb_dev.argtypes = [vp, vp, vp]
    L.sv_upload_maps_host.argtypes = [vp, vp, vp, vp, vp]
    L.sv_remap_rgb_dev.argtypes = [vp, i32, vp, vp]
    L.sv_compute_rgb_host.argtypes = [vp, vp, sz, vp, sz, i32, vp, sz]
    L.sv_compute_rgb_vis_host.argtypes = [vp, vp, sz, vp, sz, i32, vp, sz, vp, sz, i32]
    L.sv_get_rectified_host.argtypes = [vp, i32, vp, sz]
    L.sv_disparity_vis_dev.argtypes = [vp, vp, vp, i32]
    L.sv_disparity_vis_host.argtypes = [vp, vp, vp, i32]
    L.sv_filter_speckles_dev.argtypes = [vp, vp, i32, i32]
    L.sv_depth_host.argtypes = [vp, vp, sz, vp, vp, vp, vp, sz]
    L.sv_depth_dev.argtypes = [vp, vp, vp, vp, vp, vp]
    L.sv_wls_roi.argtypes = [vp, C.POINTER(C.c_int * 4)]
    L.sv_filter_wls_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.sv_filter_wls_host.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz]
    L.sv_world_image_dev.argtypes = [vp] * 8
    L.sv_world_image_host.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, sz, vp, sz]
    L.sv_compute_rgb_batch_dev.argtypes = [vp, i32, vp, vp, i32, vp, vp]
    L.sv_disparity_vis_batch_dev.argtypes = [vp, i32, vp, vp, i32]
    L.sv_depth_batch_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.sv_filter_blf_joint_dev.argtypes = [vp, vp, vp, i32, i32, i32, C.c_double, vp, vp]
    L.sv_filter_blf_joint_host.argtypes = [vp, vp, sz, vp, sz, i32, i32, i32, C.c_double, vp, sz, vp, sz]
    L.sv_filter_blf_dev.argtypes = [vp, vp, C.c_double, C.c_double, vp, vp]
    L.sv_filter_blf_host.argtypes = [vp, vp, sz, C.c_double, C.c_double, vp, sz, vp, sz]
    L.sv_init_rectify_map.argtypes = [vp, i32, vp, vp, i32, vp, vp]
    L.sv_update_mapper.argtypes = [vp, i32, vp, vp, i32, vp, vp]
    L.sv_get_maps_host.argtypes = [vp, i32, vp, vp]
    L.sv_get_mapper_host.argtypes = [vp, i32, vp]
    L.sv_synchronize.argtypes = [vp]
    L.sv_last_aggregation.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sv_set_debug.argtypes = [vp, i32]
    L.sv_debug_read_volume_host.argtypes = [vp, i32, vp, sz]
    L.sv_debug_read_plane_host.argtypes = [vp, i32, vp]
    L.sv_enable_timing.argtypes = [vp, i32]
    L.sv_get_timings.argtypes = [vp, C.POINTER(C.c_float), i32]
    L.sv_get_aggregation_timings.argtypes = [vp, C.POINTER(C.c_float), i32, C.POINTER(C.c_int)]
    L.sv_filter_speckles_static_dev.argtypes = [vp, vp, i32, i32]
    L.sv_last_sweep_waves.argtypes = [vp]
    L.sv_debug_selftest_min3.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.sv_debug_sweep_order.argtypes = [vp, i32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.sv_pool_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), i32, i32, i32, C.POINTER(Params), i32]
    L.sv_pool_destroy.argtypes = [vp]
    L.sv_pool_size.argtypes = [vp]
    L.sv_pool_plan.argtypes = [i32, i32, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sv_pool_compute_gray_host.argtypes = [vp, i32, C.POINTER(vp), sz, C.POINTER(vp), sz, C.POINTER(vp), sz]
    L.sv_pool_last_error.restype = C.c_char_p
    L.sv_pool_last_error.argtypes = [vp]
    L.sv_algorithmic_bytes.restype = C.c_int64
    L.sv_algorithmic_bytes.argtypes = [vp, i32]
    _lib = L
    return L


def default_params(num_disparities, paths=8, **over):
    p = Params()
    rc = load().sv_default_params(C.byref(p), num_disparities, paths)
    if rc:
        raise SvError(rc, "sv_default_params")
    for k, v in over.items():
        setattr(p, k, v)
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Matcher:
    """One sv_ctx.  Mirrors the matcher part of StereoMatcherNew for one geometry."""

    def __init__(self, width, height, params, device=0):
        self._L = load()
        self._h = C.c_void_p()
        self.W, self.H = width, height
        rc = self._L.sv_create(C.byref(self._h), device, width, height, C.byref(params))
        if rc:
            msg = self._L.sv_last_error(self._h).decode() if self._h else "sv_create failed"
            if self._h:
                self._L.sv_destroy(self._h)
                self._h = C.c_void_p()
            raise SvError(rc, msg)
        self.params = params.copy()

    def _chk(self, rc):
        if rc:
            raise SvError(rc, self._L.sv_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.sv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params):
        self._chk(self._L.sv_set_params(self._h, C.byref(params)))
        self.params = params.copy()

    def set_stream(self, stream_handle):
        self._chk(self._L.sv_set_stream(self._h, C.c_void_p(stream_handle)))

    def enable_timing(self, on=True):
        self._chk(self._L.sv_enable_timing(self._h, int(on)))

    def timings(self):
        ms = (C.c_float * N_TIMINGS)()
        self._chk(self._L.sv_get_timings(self._h, ms, N_TIMINGS))
        return dict(zip(TIMING_NAMES, list(ms)))

    def aggregation_timings(self):
        """ms of each kernel launch of the aggregation stage, in launch order (mean over the timed calls)."""
        ms = (C.c_float * 8)()
        n = C.c_int()
        self._chk(self._L.sv_get_aggregation_timings(self._h, ms, 8, C.byref(n)))
        return list(ms)[:n.value]

    def algorithmic_bytes(self, which=4):
        return int(self._L.sv_algorithmic_bytes(self._h, which))

    # ---- host-buffer entry points (numpy) -------------------------------------------------
    def compute_gray(self, left, right):
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        assert left.shape == (self.H, self.W) and right.shape == (self.H, self.W)
        out = np.empty((self.H, self.W), np.int16)
        self._chk(self._L.sv_compute_gray_host(self._h, _ptr(left), self.W, _ptr(right), self.W, _ptr(out), 2 * self.W))
        return out

    def submit_gray(self, left, right):
        """Asynchronous compute_gray: returns a ticket for wait()."""
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        assert left.shape == (self.H, self.W) and right.shape == (self.H, self.W)
        t = C.c_int()
        self._chk(self._L.sv_submit_gray_host(self._h, _ptr(left), self.W, _ptr(right), self.W, C.byref(t)))
        return t.value

    def submit_gray_batch(self, lefts, rights, outs, left_stride=None, right_stride=None, out_stride=None):
        """Asynchronous batch through host buffers (sv_submit_gray_batch_host): lefts / rights / outs are
        sequences of per-frame addresses (ints) or numpy arrays; every buffer must stay alive and untouched
        until wait_batch(ticket).  Strides in bytes (default: dense rows)."""
        n = len(lefts)
        addr = lambda a: a if isinstance(a, int) else a.ctypes.data
        pl = (C.c_void_p * n)(*[addr(a) for a in lefts])
        pr = (C.c_void_p * n)(*[addr(a) for a in rights])
        po = (C.c_void_p * n)(*[addr(a) for a in outs])
        t = C.c_int()
        self._chk(self._L.sv_submit_gray_batch_host(self._h, n, pl, left_stride or self.W, pr, right_stride or self.W,
                                                    po, out_stride or 2 * self.W, C.byref(t)))
        return t.value

    def wait_batch(self, ticket):
        self._chk(self._L.sv_wait_batch_host(self._h, ticket))

    def wait(self, ticket):
        out = np.empty((self.H, self.W), np.int16)
        self._chk(self._L.sv_wait_host(self._h, ticket, _ptr(out), 2 * self.W))
        return out

    def upload_maps(self, mlx, mly, mrx, mry):
        maps = [np.ascontiguousarray(m, np.float32) for m in (mlx, mly, mrx, mry)]
        for m in maps:
            assert m.shape == (self.H, self.W)
        self._chk(self._L.sv_upload_maps_host(self._h, *[_ptr(m) for m in maps]))

    def compute_rgb(self, left_rgb, right_rgb, rectify=True):
        left_rgb = np.ascontiguousarray(left_rgb, np.uint8)
        right_rgb = np.ascontiguousarray(right_rgb, np.uint8)
        assert left_rgb.shape == (self.H, self.W, 3) and right_rgb.shape == (self.H, self.W, 3)
        out = np.empty((self.H, self.W), np.int16)
        self._chk(self._L.sv_compute_rgb_host(self._h, _ptr(left_rgb), 3 * self.W, _ptr(right_rgb), 3 * self.W,
                                              int(rectify), _ptr(out), 2 * self.W))
        return out

    def compute_rgb_vis(self, left_rgb, right_rgb, rectify=True, scale=3):
        """compute_rgb + getDisparityVis(.,.,scale) in one call -> (disp16, vis)."""
        left_rgb = np.ascontiguousarray(left_rgb, np.uint8)
        right_rgb = np.ascontiguousarray(right_rgb, np.uint8)
        assert left_rgb.shape == (self.H, self.W, 3) and right_rgb.shape == (self.H, self.W, 3)
        out = np.empty((self.H, self.W), np.int16)
        vis = np.empty((self.H, self.W), np.uint8)
        self._chk(self._L.sv_compute_rgb_vis_host(self._h, _ptr(left_rgb), 3 * self.W, _ptr(right_rgb), 3 * self.W,
                                                  int(rectify), _ptr(out), 2 * self.W, _ptr(vis), self.W, scale))
        return out, vis

    def rectified(self, which):
        out = np.empty((self.H, self.W, 3), np.uint8)
        self._chk(self._L.sv_get_rectified_host(self._h, which, _ptr(out), 3 * self.W))
        return out

    def disparity_vis(self, disp16, scale=3):
        d = np.ascontiguousarray(disp16, np.int16)
        out = np.empty((self.H, self.W), np.uint8)
        self._chk(self._L.sv_disparity_vis_host(self._h, _ptr(d), _ptr(out), scale))
        return out

    def depth(self, disp16, mapper, Q, R):
        """mapper None = the device-resident MapperL made by update_mapper(0, ...)."""
        d = np.ascontiguousarray(disp16, np.int16)
        Q = np.ascontiguousarray(Q, np.float64)
        R = np.ascontiguousarray(R, np.float64)
        assert Q.shape == (4, 4) and R.shape == (3, 3)
        if mapper is not None:
            mapper = np.ascontiguousarray(mapper, np.float32)
            assert mapper.shape == (self.H, self.W, 2)
        out = np.empty((self.H, self.W), np.float32)
        self._chk(self._L.sv_depth_host(self._h, _ptr(d), 2 * self.W, _ptr(mapper) if mapper is not None else None,
                                        _ptr(Q), _ptr(R), _ptr(out), 4 * self.W))
        return out

    @staticmethod
    def _cam(K, dist, R, P):
        K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
        d = np.ascontiguousarray(np.asarray(dist, np.float64).ravel())
        R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
        P = np.ascontiguousarray(P, np.float64).reshape(3, 4)
        return K, d, R, P

    def init_rectify_map(self, which, K, dist, R, P):
        K, d, R, P = self._cam(K, dist, R, P)
        self._chk(self._L.sv_init_rectify_map(self._h, which, _ptr(K), _ptr(d), len(d), _ptr(R), _ptr(P)))

    def update_mapper(self, which, K, dist, R, P):
        K, d, R, P = self._cam(K, dist, R, P)
        self._chk(self._L.sv_update_mapper(self._h, which, _ptr(K), _ptr(d), len(d), _ptr(R), _ptr(P)))

    def maps(self, which):
        mx, my = np.empty((self.H, self.W), np.float32), np.empty((self.H, self.W), np.float32)
        self._chk(self._L.sv_get_maps_host(self._h, which, _ptr(mx), _ptr(my)))
        return mx, my

    def mapper(self, which):
        out = np.empty((self.H, self.W, 2), np.float32)
        self._chk(self._L.sv_get_mapper_host(self._h, which, _ptr(out)))
        return out

    def wls_roi(self):
        r = (C.c_int * 4)()
        self._chk(self._L.sv_wls_roi(self._h, C.byref(r)))
        return tuple(r)

    def filter_wls(self, disp16, guide_rgb=None, vis=None):
        """filterWLS on host arrays; guide_rgb None = rectified left image of the last compute_rgb."""
        d = np.ascontiguousarray(disp16, np.int16)
        out16 = np.empty((self.H, self.W), np.int16)
        g = np.ascontiguousarray(guide_rgb, np.uint8) if guide_rgb is not None else None
        v = np.ascontiguousarray(vis, np.uint8) if vis is not None else None
        out8 = np.empty((self.H, self.W), np.uint8) if vis is not None else None
        self._chk(self._L.sv_filter_wls_host(
            self._h, _ptr(d), 2 * self.W, _ptr(v) if v is not None else None, self.W,
            _ptr(g) if g is not None else None, 3 * self.W, _ptr(out16), 2 * self.W,
            _ptr(out8) if out8 is not None else None, self.W))
        return (out16, out8) if vis is not None else out16

    def filter_wls_dev(self, n, disp_t, guide_t, out_t, vis_t=None, vis_out_t=None):
        self._chk(self._L.sv_filter_wls_dev(
            self._h, n, C.c_void_p(disp_t.data_ptr()), C.c_void_p(vis_t.data_ptr()) if vis_t is not None else None,
            C.c_void_p(guide_t.data_ptr()), C.c_void_p(out_t.data_ptr()),
            C.c_void_p(vis_out_t.data_ptr()) if vis_out_t is not None else None))

    def world_image(self, disp16, Q, RLrect, HL_root, mapper=None):
        """SFM::fillWorld3D on host arrays -> (cartesian, cylindrical) [H][W][3] float; mapper None = device MapperL."""
        d = np.ascontiguousarray(disp16, np.int16)
        mp = np.ascontiguousarray(mapper, np.float32) if mapper is not None else None
        Qm, Rm, Hm = (np.ascontiguousarray(a, np.float64) for a in (Q, RLrect, HL_root))
        cart = np.empty((self.H, self.W, 3), np.float32)
        cyl = np.empty((self.H, self.W, 3), np.float32)
        self._chk(self._L.sv_world_image_host(self._h, _ptr(d), 2 * self.W, _ptr(mp) if mp is not None else None, _ptr(Qm), _ptr(Rm),
                                              _ptr(Hm), _ptr(cart), 12 * self.W, _ptr(cyl), 12 * self.W))
        return cart, cyl

    def compute_rgb_batch_dev(self, n, left_t, right_t, out_t, rectify=True, rect_left_t=None):
        self._chk(self._L.sv_compute_rgb_batch_dev(
            self._h, n, C.c_void_p(left_t.data_ptr()), C.c_void_p(right_t.data_ptr()), int(rectify),
            C.c_void_p(rect_left_t.data_ptr()) if rect_left_t is not None else None, C.c_void_p(out_t.data_ptr())))

    def disparity_vis_batch_dev(self, n, disp_t, vis_t, scale=3):
        self._chk(self._L.sv_disparity_vis_batch_dev(self._h, n, C.c_void_p(disp_t.data_ptr()), C.c_void_p(vis_t.data_ptr()), scale))

    def depth_batch_dev(self, n, disp_t, Q, R, depth_t, mapper_t=None):
        Qm, Rm = np.ascontiguousarray(Q, np.float64), np.ascontiguousarray(R, np.float64)
        self._chk(self._L.sv_depth_batch_dev(self._h, n, C.c_void_p(disp_t.data_ptr()),
                                             C.c_void_p(mapper_t.data_ptr()) if mapper_t is not None else None,
                                             _ptr(Qm), _ptr(Rm), C.c_void_p(depth_t.data_ptr())))

    def filter_blf(self, disp16, sigma_color=10.0, sigma_space=10.0, vis=False):
        """filterBLF / BLF_ORIGINAL on a host array; vis=True also returns getDisparityVis(out, 3)."""
        d = np.ascontiguousarray(disp16, np.int16)
        out16 = np.empty((self.H, self.W), np.int16)
        out8 = np.empty((self.H, self.W), np.uint8) if vis else None
        self._chk(self._L.sv_filter_blf_host(self._h, _ptr(d), 2 * self.W, sigma_color, sigma_space, _ptr(out16), 2 * self.W,
                                             _ptr(out8) if vis else None, self.W))
        return (out16, out8) if vis else out16

    def filter_blf_joint(self, disp16, ndisp, sigma_range=10.0, guide_rgb=None, radius=7, iters=2, vis=False):
        """filterBLF / BLF_CUDA semantics on host arrays; guide_rgb None = rectified left image of the last compute_rgb."""
        d = np.ascontiguousarray(disp16, np.int16)
        g = np.ascontiguousarray(guide_rgb, np.uint8) if guide_rgb is not None else None
        out16 = np.empty((self.H, self.W), np.int16)
        out8 = np.empty((self.H, self.W), np.uint8) if vis else None
        self._chk(self._L.sv_filter_blf_joint_host(self._h, _ptr(d), 2 * self.W, _ptr(g) if g is not None else None, 3 * self.W,
                                                   ndisp, radius, iters, sigma_range, _ptr(out16), 2 * self.W,
                                                   _ptr(out8) if vis else None, self.W))
        return (out16, out8) if vis else out16

    def filter_blf_joint_dev(self, disp_t, gray_t, out_t, ndisp, sigma_range=10.0, radius=7, iters=2, vis_out_t=None):
        self._chk(self._L.sv_filter_blf_joint_dev(self._h, C.c_void_p(disp_t.data_ptr()), C.c_void_p(gray_t.data_ptr()), ndisp, radius,
                                                  iters, sigma_range, C.c_void_p(out_t.data_ptr()),
                                                  C.c_void_p(vis_out_t.data_ptr()) if vis_out_t is not None else None))

    def filter_blf_dev(self, disp_t, out_t, sigma_color=10.0, sigma_space=10.0, vis_out_t=None):
        self._chk(self._L.sv_filter_blf_dev(self._h, C.c_void_p(disp_t.data_ptr()), sigma_color, sigma_space,
                                            C.c_void_p(out_t.data_ptr()),
                                            C.c_void_p(vis_out_t.data_ptr()) if vis_out_t is not None else None))

    def last_aggregation(self):
        """(aggregation launches, fused-sweep launches among them) of the last compute call."""
        a, b = C.c_int(), C.c_int()
        self._chk(self._L.sv_last_aggregation(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_sweep_waves(self):
        return int(self._L.sv_last_sweep_waves(self._h))

    def sweep_order(self, enable=True):
        """sv_debug_sweep_order: (launches checked, workgroups that started > 20 us before their predecessor on the same
        XCD, largest such lead in 10 ns ticks, largest lead over workgroup blockIdx - 1) since the previous call; enable
        switches the recording for the launches that follow."""
        n, v, w, wp = C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong()
        self._chk(self._L.sv_debug_sweep_order(self._h, 1 if enable else 0, C.byref(n), C.byref(v), C.byref(w), C.byref(wp)))
        return n.value, v.value, w.value, wp.value

    def synchronize(self):
        self._chk(self._L.sv_synchronize(self._h))

    def selftest_min3(self):
        """sv_debug_selftest_min3: mismatches of the sweeps' packed binary16 three-input minimum against the integer minimum."""
        n = C.c_ulonglong(1)
        self._chk(self._L.sv_debug_selftest_min3(self._h, C.byref(n)))
        return int(n.value)

    def set_debug(self, on=True):
        self._chk(self._L.sv_set_debug(self._h, int(on)))

    def debug_volume(self, which):
        D, minD = self.params.num_disparities, self.params.min_disparity
        out = np.empty((self.H, (self.W + min(minD, 0)) - max(minD + D, 0), D), np.int16)
        self._chk(self._L.sv_debug_read_volume_host(self._h, which, _ptr(out), out.size))
        return out

    def debug_plane(self, which):
        out = np.empty((self.H, self.W), np.int16)
        self._chk(self._L.sv_debug_read_plane_host(self._h, which, _ptr(out)))
        return out

    # ---- device-pointer entry points (torch tensors own the memory) -----------------------
    def compute_gray_dev(self, left_t, right_t, out_t):
        """left_t/right_t: contiguous uint8 CUDA tensors [H][W]; out_t: int16 CUDA tensor [H][W]."""
        self._chk(self._L.sv_compute_gray_dev(self._h, C.c_void_p(left_t.data_ptr()), C.c_void_p(right_t.data_ptr()),
                                              C.c_void_p(out_t.data_ptr())))

    def set_batch(self, n):
        self._chk(self._L.sv_set_batch(self._h, n))

    def compute_gray_batch_dev(self, n, left_t, right_t, out_t):
        """left_t/right_t: uint8 CUDA tensors [n][H][W]; out_t: int16 CUDA tensor [n][H][W]."""
        self._chk(self._L.sv_compute_gray_batch_dev(self._h, n, C.c_void_p(left_t.data_ptr()),
                                                    C.c_void_p(right_t.data_ptr()), C.c_void_p(out_t.data_ptr())))

    def filter_speckles_dev(self, disp_t, max_size, max_diff):
        self._chk(self._L.sv_filter_speckles_dev(self._h, C.c_void_p(disp_t.data_ptr()), max_size, max_diff))

    def filter_speckles_static_dev(self, disp_t, max_size, max_diff):
        """The allocation-free variant (static device scratch, frames up to STATIC_MAX_PIXELS pixels)."""
        self._chk(self._L.sv_filter_speckles_static_dev(self._h, C.c_void_p(disp_t.data_ptr()), max_size, max_diff))

    def remap_rgb_dev(self, which, src_t, dst_t):
        self._chk(self._L.sv_remap_rgb_dev(self._h, which, C.c_void_p(src_t.data_ptr()), C.c_void_p(dst_t.data_ptr())))

    def gray_from_rgb_dev(self, rgb_t, gray_t):
        self._chk(self._L.sv_gray_from_rgb_dev(self._h, C.c_void_p(rgb_t.data_ptr()), C.c_void_p(gray_t.data_ptr())))


def pool_plan(n_frames, n_contexts, batch, frame):
    """(context index, chunk index) of `frame` under sv_pool's dealing rule (pure function, needs no GPU)."""
    a, b = C.c_int(), C.c_int()
    rc = load().sv_pool_plan(n_frames, n_contexts, batch, frame, C.byref(a), C.byref(b))
    if rc:
        raise SvError(rc, "sv_pool_plan")
    return a.value, b.value


class Pool:
    """sv_pool: one context + stream + host thread per entry of `devices`."""

    def __init__(self, devices, width, height, params, batch):
        self._L = load()
        self._h = C.c_void_p()
        self.W, self.H = width, height
        devs = (C.c_int * len(devices))(*devices)
        rc = self._L.sv_pool_create(C.byref(self._h), devs, len(devices), width, height, C.byref(params), batch)
        if rc:
            msg = self._L.sv_pool_last_error(self._h).decode() if self._h else "sv_pool_create failed"
            if self._h:
                self._L.sv_pool_destroy(self._h)
                self._h = C.c_void_p()
            raise SvError(rc, msg)

    def compute_gray(self, lefts, rights, outs):
        """lefts / rights / outs: sequences of numpy arrays (or addresses), one per pair; blocks until all are done."""
        n = len(lefts)
        addr = lambda a: a if isinstance(a, int) else a.ctypes.data
        pl = (C.c_void_p * n)(*[addr(a) for a in lefts])
        pr = (C.c_void_p * n)(*[addr(a) for a in rights])
        po = (C.c_void_p * n)(*[addr(a) for a in outs])
        rc = self._L.sv_pool_compute_gray_host(self._h, n, pl, self.W, pr, self.W, po, 2 * self.W)
        if rc:
            raise SvError(rc, self._L.sv_pool_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.sv_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedArray:
    """numpy view of page-locked host memory from sv_host_alloc (freed with free() or when collected)."""

    def __init__(self, shape, dtype):
        self._L = load()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = C.c_void_p()
        rc = self._L.sv_host_alloc(C.byref(self._p), self.nbytes)
        if rc:
            raise SvError(rc, "sv_host_alloc(%d bytes)" % self.nbytes)
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(self.nbytes,)).view(dtype).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            self._L.sv_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
