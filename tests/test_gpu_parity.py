"""GPU parity: libsv_hip (through its C ABI) vs the CPU oracle, bit-exact on integers."""
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


CASES = [
    # W, H, D, paths, flags, block, seed
    (96, 40, 32, 5, svhip.FLAGS_REPO, 7, 1),
    (96, 40, 32, 8, svhip.FLAGS_HH, 7, 2),
    (131, 57, 64, 8, svhip.FLAGS_HH, 5, 3),     # ragged: W1 = 67, H = 57 (not multiples of 4)
    (200, 50, 96, 5, svhip.FLAGS_REPO, 9, 4),   # D = 96: 3 packed registers per lane
    (320, 240, 64, 8, svhip.FLAGS_HH, 7, 101),  # C1 geometry
    (300, 33, 160, 8, svhip.FLAGS_REPO, 3, 5),  # D = 160
    (330, 40, 256, 8, svhip.FLAGS_HH, 7, 6),    # D = 256
]


@pytest.mark.parametrize("W,H,D,paths,flags,bs,seed", CASES)
def test_stages_bit_exact(gpu, W, H, D, paths, flags, bs, seed):
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    ph, po = _params(D, paths, flags=flags, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs)
    ref, st = svref.sgbm(left, right, po, stages=("C", "S", "disp_wta", "raw", "mins", "disp_lr"))
    m = svhip.Matcher(W, H, ph)
    assert np.array_equal(m.compute_gray(left, right), ref), "final disparity16 differs (production path, one frame)"
    m.set_debug(4)                       # plain per-path launches + WTA kernel
    assert np.array_equal(m.compute_gray(left, right), ref), "final disparity16 differs (per-path launches)"
    m.set_debug(5)                       # concurrent paths into separate volumes + sum/WTA kernel
    assert np.array_equal(m.compute_gray(left, right), ref), "final disparity16 differs (concurrent paths)"
    assert np.array_equal(m.debug_plane(2), st["mins"]), "minS differs (concurrent paths)"
    m.set_debug(2)                       # mid-size batches: WTA fused into the last path, S never stored
    fused = m.compute_gray(left, right)
    assert np.array_equal(fused, ref), "final disparity16 differs (fused WTA)"
    assert np.array_equal(m.debug_plane(2), st["mins"]), "minS differs (fused WTA)"
    assert np.array_equal(m.debug_plane(1), st["raw"]), "raw disparity differs (fused WTA)"
    assert np.array_equal(m.debug_plane(0), st["disp_wta"]), "sub-pixel disparity differs (fused WTA)"
    m.set_debug(True)                    # debug path: S materialised, WTA as its own kernel
    got = m.compute_gray(left, right)
    assert np.array_equal(m.debug_volume(0), st["C"]), "cost volume differs"
    assert np.array_equal(m.debug_volume(1), st["S"]), "aggregated volume differs"
    assert np.array_equal(m.debug_plane(2), st["mins"]), "minS differs"
    assert np.array_equal(m.debug_plane(1), st["raw"]), "raw disparity differs"
    assert np.array_equal(m.debug_plane(0), st["disp_wta"]), "sub-pixel disparity differs"
    assert np.array_equal(m.debug_plane(3), st["disp_lr"]), "LR-checked disparity differs"
    assert np.array_equal(got, ref), "final disparity16 differs"
    m.close()


@pytest.mark.parametrize("mask", [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x55, 0xAA])
def test_single_paths(gpu, mask):
    W, H, D = 150, 61, 64
    left, right, _ = synth.stereo_pair(11, W, H, D)
    ph, po = _params(D, 8, path_mask=mask, flags=0)
    ref, st = svref.sgbm(left, right, po, stages=("S",))
    m = svhip.Matcher(W, H, ph)
    assert np.array_equal(m.compute_gray(left, right), ref)
    m.set_debug(2)
    assert np.array_equal(m.compute_gray(left, right), ref)
    m.set_debug(5)
    assert np.array_equal(m.compute_gray(left, right), ref)
    m.set_debug(True)
    got = m.compute_gray(left, right)
    assert np.array_equal(m.debug_volume(1), st["S"])
    assert np.array_equal(got, ref)
    m.close()


def test_batched_launch_matches_oracle(gpu):
    """Frame-batched entry point: every frame of the batch equals the oracle's frame."""
    import torch
    W, H, D, nb = 160, 64, 64, 5
    ph, po = _params(D, 8, speckle_window_size=30, speckle_range=2)
    pairs = [synth.stereo_pair(700 + i, W, H, D) for i in range(nb)]
    lt = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    rt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    m = svhip.Matcher(W, H, ph)
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in range(nb):
        assert np.array_equal(got[i], svref.sgbm(pairs[i][0], pairs[i][1], po)), "frame %d differs" % i
    # a smaller batch on the same context
    m.compute_gray_batch_dev(2, lt[3:], rt[3:], out)
    torch.cuda.synchronize()
    assert np.array_equal(out[:2].cpu().numpy(), got[3:5])
    with pytest.raises(svhip.SvError):
        m.compute_gray_batch_dev(nb + 1, lt, rt, out)
    m.close()


@pytest.mark.parametrize("W,H,D,seed", [(640, 480, 128, 202), (1920, 1080, 256, 404)])
def test_full_size_configs_bit_exact(gpu, W, H, D, seed):
    """BASELINE.json configs C2 and C4 at full size: the oracle needs 0.4 s / ~6 s per frame, so the
    comparison is direct; plus the shift property (right = left shifted by k -> disparity 16k)."""
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    ph, po = _params(D, 8)
    m = svhip.Matcher(W, H, ph)
    got = m.compute_gray(left, right)
    assert np.array_equal(got, svref.sgbm(left, right, po))
    m.set_debug(2)
    assert np.array_equal(m.compute_gray(left, right), got), "fused-WTA path differs at full size"
    m.set_debug(0)
    k = D // 3
    shifted = np.roll(left, -k, axis=1)
    d = m.compute_gray(left, shifted)
    inner = d[16:-16, D + 16:W - k - 16]
    assert (inner == 16 * k).mean() > 0.99
    m.close()


def test_async_submit_wait(gpu):
    """sv_submit_gray_host / sv_wait_host: several pairs in flight, results identical to the
    synchronous entry point, ring overflow and stale tickets rejected."""
    W, H, D = 200, 90, 64
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(60 + i, W, H, D) for i in range(11)]
    m = svhip.Matcher(W, H, ph)
    ref = [m.compute_gray(l, r) for l, r, _ in pairs]
    tickets = [m.submit_gray(pairs[i][0], pairs[i][1]) for i in range(8)]
    with pytest.raises(svhip.SvError):
        m.submit_gray(pairs[8][0], pairs[8][1])           # ninth pair: all slots busy
    for i in range(3):
        assert np.array_equal(m.wait(tickets[i]), ref[i])
    tickets += [m.submit_gray(pairs[i][0], pairs[i][1]) for i in range(8, 11)]
    for i in range(3, 11):
        assert np.array_equal(m.wait(tickets[i]), ref[i]), i
    with pytest.raises(svhip.SvError):
        m.wait(tickets[2])                                 # already collected
    m.close()


def test_random_geometries_all_variants(gpu):
    """Seeded random image sizes, disparity ranges, block sizes, path sets and flags - small, ragged, narrow
    and short ones included - through every aggregation variant (production choice, per-path launches,
    all paths in one launch, fused sweeps of both workgroup shapes) against the oracle, bit-exact."""
    rng = np.random.default_rng(20240611)
    n = 0
    while n < 28:
        D = int(rng.choice([32, 64, 96, 128, 160, 224, 256]))
        bs = int(rng.choice([1, 3, 5, 7, 9, 11]))
        paths = int(rng.choice([5, 8]))
        flags = int(rng.choice([svhip.FLAGS_REPO, svhip.FLAGS_HH]))
        W = D + bs + int(rng.integers(0, 140))
        H = bs + int(rng.integers(0, 150))
        cap = int(rng.choice([15, 31, 63]))
        uniq = int(rng.choice([0, 5, 15, 40]))
        P1, P2 = 8 * bs * bs, 32 * bs * bs
        if (2 * cap + 63) * bs * bs + P2 > 32767:
            continue
        n += 1
        left, right, _ = synth.stereo_pair(1000 + n, W, H, D)
        ph, po = _params(D, paths, flags=flags, block_size=bs, P1=P1, P2=P2, pre_filter_cap=cap, uniqueness_ratio=uniq)
        ref = svref.sgbm(left, right, po)
        what = "W=%d H=%d D=%d bs=%d paths=%d flags=%#x cap=%d uniq=%d" % (W, H, D, bs, paths, flags, cap, uniq)
        m = svhip.Matcher(W, H, ph)
        for mode in (0, 4, 5, 3):
            m.set_debug(mode)
            assert np.array_equal(m.compute_gray(left, right), ref), "%s, debug mode %d" % (what, mode)
        m.close()


@pytest.mark.parametrize("name", ["synthetic_64x48_d32", "icub_320x240_d64", "view_463x370_d96"])
def test_golden_fixtures_on_gpu(gpu, name):
    """The committed fixtures (tests/golden/: the reference's own sample images, sgbm_cuda/lefti.png etc.,
    with the parameters of sgbm_cuda/main.cpp:47-54) read as data, not through the oracle: final
    disparity16 and the CRC of every stage, on every aggregation variant."""
    import os
    import zlib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    D, paths = int(g["D"]), int(g["paths"])
    ph = svhip.default_params(D, paths, flags=int(g["flags"]), uniqueness_ratio=int(g["uniq"]), disp12_max_diff=int(g["disp12"]))
    left, right = np.ascontiguousarray(g["left"]), np.ascontiguousarray(g["right"])
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
    m = svhip.Matcher(left.shape[1], left.shape[0], ph)
    for mode in (0, 2, 3, 4, 5):
        m.set_debug(mode)
        assert np.array_equal(m.compute_gray(left, right), g["disp16"]), "debug mode %d" % mode
    m.set_debug(True)
    assert np.array_equal(m.compute_gray(left, right), g["disp16"])
    assert crc(m.debug_volume(0)) == int(g["crc_C"])
    assert crc(m.debug_volume(1)) == int(g["crc_S"])
    assert crc(m.debug_plane(0)) == int(g["crc_disp_wta"])
    assert crc(m.debug_plane(3)) == int(g["crc_disp_lr"])
    m.close()


@pytest.mark.parametrize("uniq", [100, 99, 50, 1])
def test_uniqueness_ratio_extremes(gpu, uniq):
    """Uniqueness ratios at the ends of the accepted range (100: every pixel with minS > 0 is rejected),
    every aggregation variant."""
    W, H, D = 150, 61, 64
    left, right, _ = synth.stereo_pair(12, W, H, D)
    ph, po = _params(D, 8, uniqueness_ratio=uniq)
    ref = svref.sgbm(left, right, po)
    m = svhip.Matcher(W, H, ph)
    for mode in (0, 2, 3, 4, 5):
        m.set_debug(mode)
        assert np.array_equal(m.compute_gray(left, right), ref), "debug mode %d" % mode
    m.close()


def test_async_batch_host(gpu):
    """sv_submit_gray_batch_host / sv_wait_batch_host: two batches in flight, pageable and pinned buffers,
    scattered frames, one dense array of frames and padded rows; every frame equal to the synchronous entry point."""
    W, H, D, nb = 200, 90, 64, 5
    ph, _ = _params(D, 8)
    m = svhip.Matcher(W, H, ph)
    m.set_batch(nb)
    pairs = [synth.stereo_pair(60 + i, W, H, D) for i in range(2 * nb)]
    refs = [m.compute_gray(p[0], p[1]) for p in pairs]
    # batch A: scattered pageable frames; batch B: one pinned array of frames (dense: merged copies)
    outs_a = [np.full((H, W), 777, np.int16) for _ in range(nb)]
    pin_l, pin_r = svhip.PinnedArray((nb, H, W), np.uint8), svhip.PinnedArray((nb, H, W), np.uint8)
    pin_o = svhip.PinnedArray((nb, H, W), np.int16)
    for i in range(nb):
        pin_l.array[i], pin_r.array[i] = pairs[nb + i][:2]
    pin_o.array[:] = 777
    ta = m.submit_gray_batch([p[0] for p in pairs[:nb]], [p[1] for p in pairs[:nb]], outs_a)
    tb = m.submit_gray_batch([pin_l.array[i] for i in range(nb)], [pin_r.array[i] for i in range(nb)], [pin_o.array[i] for i in range(nb)])
    with pytest.raises(svhip.SvError):
        m.submit_gray_batch([pairs[0][0]], [pairs[0][1]], [outs_a[0]])   # both slots in flight
    m.wait_batch(ta)
    for i in range(nb):
        assert np.array_equal(outs_a[i], refs[i]), "batch A frame %d" % i
    m.wait_batch(tb)
    for i in range(nb):
        assert np.array_equal(pin_o.array[i], refs[nb + i]), "batch B frame %d" % i
    with pytest.raises(svhip.SvError):
        m.wait_batch(tb)                                                    # already collected
    # padded rows (a YARP image: rows padded to 8 bytes; here stride W + 24) and a smaller batch
    ls = W + 24
    padl, padr = np.zeros((3, H, ls), np.uint8), np.zeros((3, H, ls), np.uint8)
    pado = np.full((3, H, ls), 777, np.int16)
    for i in range(3):
        padl[i, :, :W], padr[i, :, :W] = pairs[i][:2]
    t = m.submit_gray_batch([padl[i] for i in range(3)], [padr[i] for i in range(3)], [pado[i] for i in range(3)],
                            left_stride=ls, right_stride=ls, out_stride=2 * ls)
    m.wait_batch(t)
    for i in range(3):
        assert np.array_equal(pado[i, :, :W], refs[i]) and (pado[i, :, W:] == 777).all()
    with pytest.raises(svhip.SvError):
        m.submit_gray_batch([p[0] for p in pairs[:nb + 1]], [p[1] for p in pairs[:nb + 1]], outs_a + [outs_a[0]])  # > sv_set_batch
    for a in (pin_l, pin_r, pin_o):
        a.free()
    m.close()


@pytest.mark.parametrize("W,H,D,bs,cap,paths", [
    (200, 60, 64, 13, 31, 8),     # blocks beyond 11 with the caps that keep every cost inside int16
    (180, 70, 32, 15, 15, 5),
    (330, 48, 256, 15, 15, 8),    # D = 256: the 16-row ring of a 15 x 15 block needs the 24-column tile to fit the LDS
    (160, 50, 64, 7, 100, 8),     # the GUI's upper preFilterCap (cvgui.cpp:102)
    (160, 50, 64, 7, 127, 5),     # the largest cap whose clip table fits a byte
    (160, 50, 64, 5, 0, 8),       # the GUI's lower end: every gradient clipped to 0
])
def test_gui_parameter_ranges_bit_exact(gpu, W, H, D, bs, cap, paths):
    """SADWindowSize / preFilterCap beyond the module defaults (cvgui.cpp:100,102): cost volume, S and
    disparity against the oracle, single frame (every aggregation variant) and a small batch."""
    import torch
    left, right, _ = synth.stereo_pair(4000 + bs + cap, W, H, D)
    ph, po = _params(D, paths, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs, pre_filter_cap=cap)
    ref, st = svref.sgbm(left, right, po, stages=("C", "S"))
    m = svhip.Matcher(W, H, ph)
    for mode in (0, 4, 5, 3):
        m.set_debug(mode)
        assert np.array_equal(m.compute_gray(left, right), ref), "debug mode %d" % mode
    m.set_debug(True)
    assert np.array_equal(m.compute_gray(left, right), ref)
    assert np.array_equal(m.debug_volume(0), st["C"]) and np.array_equal(m.debug_volume(1), st["S"])
    m.set_debug(0)
    nb = 9   # the batched cost kernel (frame -> XCD order, 64-column tiles where they fit)
    m.set_batch(nb)
    lt, rt = torch.from_numpy(np.stack([left] * nb)).cuda(), torch.from_numpy(np.stack([right] * nb)).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()
    assert all(np.array_equal(o, ref) for o in out.cpu().numpy())
    m.close()
    with pytest.raises(svhip.SvError):
        svhip.Matcher(W, H, svhip.default_params(D, paths, block_size=17, P1=8, P2=32, pre_filter_cap=1))   # beyond 15: refused whatever the cap
