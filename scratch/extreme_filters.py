"""Ad-hoc: post-filter stages on extreme shapes vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import svhip, svref
from test_gpu_fuzz_filters import _disp_map
for W, H, D in [(20000, 16, 32), (72, 30000, 32), (4099, 777, 64)]:
    rng = np.random.default_rng(W)
    d = _disp_map(rng, W, H, D)
    m = svhip.Matcher(W, H, svhip.default_params(D, 8))
    def chk(name, f):
        try:
            print("%dx%d %-10s %s" % (W, H, name, "ok" if f() else "MISMATCH"), flush=True)
        except Exception as e:
            print("%dx%d %-10s ERROR %s" % (W, H, name, str(e)[:150]), flush=True)
    t = torch.from_numpy(d.copy()).cuda()
    def speck():
        m.filter_speckles_dev(t, 50, 32); torch.cuda.synchronize()
        return np.array_equal(t.cpu().numpy(), svref.filter_speckles(d, -16, 50, 32))
    chk("speckle", speck)
    chk("vis", lambda: np.array_equal(m.disparity_vis(d, 3), svref.disparity_vis(d, 3)))
    guide = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    def wls():
        vis = m.disparity_vis(d, 3); roi = m.wls_roi()
        g16, g8 = m.filter_wls(d, guide, vis=vis)
        e16, e8 = svref.wls_filter(d, guide, roi, 4500.0, 1.0, vis=vis)
        return np.array_equal(g16, e16) and np.array_equal(g8, e8)
    chk("wls", wls)
    chk("blf", lambda: np.array_equal(m.filter_blf(d, 10.0, 10.0), svref.blf(d, 10.0, 10.0)))
    gray = rng.integers(0, 256, size=(H, W), dtype=np.uint8); rgb = np.repeat(gray[..., None], 3, axis=2)
    chk("blf_joint", lambda: np.array_equal(m.filter_blf_joint(d, D, 10.0, guide_rgb=rgb), svref.blf_joint(d, svref.gray_from_rgb(rgb), D, 10.0, 7, 2)))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    mx = (xx * 1.01 - 3.3).astype(np.float32); my = (yy * 0.99 + 1.2).astype(np.float32)
    def remap():
        m.upload_maps(mx, my, mx, my); m.compute_rgb(guide, guide, rectify=True)
        return np.array_equal(m.rectified(0), svref.remap_rgb(guide, mx, my))
    chk("remap", remap)
    def depth():
        mapper = np.stack([xx, yy], -1).astype(np.float32)
        Q = np.array([[1, 0, 0, -W / 2], [0, 1, 0, -H / 2], [0, 0, 0, 0.7 * W], [0, 0, 1 / 0.068, 0]], np.float64)
        got = m.depth(d, mapper, Q, np.eye(3)); exp = svref.depth(d, mapper, Q, np.eye(3), alias=False)
        fin = np.isfinite(exp)
        return np.array_equal(np.isfinite(got), fin) and (np.abs(got[fin] - exp[fin]) <= 1e-4 * np.abs(exp[fin]) + 1e-12).all()
    chk("depth", depth)
    m.close()
