"""GPU parity of the fused four-path sweep kernels (sv_sweep.hip.h) through the C ABI: forced on
(sv_set_debug mode 3) they must reproduce the oracle, and the per-path kernels, bit for bit —
single frames, ragged band heights, 5-path masks (first sweep fused, the rest per path), batches
that stream several frames through one workgroup pipeline."""
import os

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


@pytest.mark.parametrize("W,H,D,paths,flags,seed", [
    (96, 40, 32, 8, svhip.FLAGS_HH, 2),       # one band, N = 1
    (131, 57, 64, 8, svhip.FLAGS_HH, 3),      # one ragged band
    (200, 75, 96, 8, svhip.FLAGS_REPO, 4),    # two bands (60 + 15 rows), N = 3
    (320, 240, 64, 8, svhip.FLAGS_HH, 101),   # C1: four bands
    (260, 130, 128, 5, svhip.FLAGS_REPO, 7),  # 5 paths: sweep 1 fused, path L per-path kernel; three bands
    (640, 480, 128, 8, svhip.FLAGS_HH, 202),  # C2: eight bands
    (300, 65, 160, 8, svhip.FLAGS_REPO, 5),   # D > 128: 8-wave workgroups (28 rows per band), N = 5, three bands
    (330, 40, 256, 8, svhip.FLAGS_HH, 6),     # D = 256, two bands (28 + 12 rows)
    (420, 90, 224, 5, svhip.FLAGS_HH, 8),     # N = 7, 5 paths, four bands
])
def test_fused_sweeps_bit_exact(gpu, W, H, D, paths, flags, seed):
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    ph, po = _params(D, paths, flags=flags)
    ref, st = svref.sgbm(left, right, po, stages=("mins", "raw", "disp_wta"))
    m = svhip.Matcher(W, H, ph)
    m.set_debug(3)
    got = m.compute_gray(left, right)
    assert np.array_equal(m.debug_plane(2)[:, D:], st["mins"][:, D:]), "minS differs"
    assert np.array_equal(m.debug_plane(1)[:, D:], st["raw"][:, D:]), "raw disparity differs"
    assert np.array_equal(got, ref), "final disparity16 differs"
    m.close()


def test_fused_sweeps_stream_many_frames(gpu):
    """More frames than workgroup pipelines: every group streams several frames back to back."""
    import torch
    W, H, D, nb = 160, 100, 64, 11
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(800 + i, W, H, D) for i in range(nb)]
    lt = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    rt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    os.environ["SV_SWEEP_NCU"] = "6"   # 2 bands -> 3 groups -> 4, 4 and 3 frames per group
    try:
        m = svhip.Matcher(W, H, ph)
    finally:
        del os.environ["SV_SWEEP_NCU"]
    m.set_batch(nb)
    m.set_debug(3)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()
    got = out.cpu().numpy()
    for i in range(nb):
        assert np.array_equal(got[i], svref.sgbm(pairs[i][0], pairs[i][1], po)), "frame %d differs" % i
    # a second call on the same context (progress counters are reset per launch), smaller batch
    m.compute_gray_batch_dev(5, lt[2:], rt[2:], out)
    m.synchronize()
    assert np.array_equal(out[:5].cpu().numpy(), got[2:7])
    m.close()


def test_fused_sweeps_d256_batch(gpu):
    """D = 256 through the production heuristics: a batch large enough for the sweeps, several frames
    per workgroup pipeline, against the per-path kernels (which the parity tests pin to the oracle)."""
    import torch
    W, H, D, nb = 400, 56, 256, 9
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(900 + i, W, H, D) for i in range(3)]
    lt = torch.from_numpy(np.stack([pairs[i % 3][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % 3][1] for i in range(nb)])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    os.environ["SV_SWEEP_NCU"] = "6"   # 2 bands of 28 rows -> 3 groups of 3 frames
    try:
        m = svhip.Matcher(W, H, ph)
    finally:
        del os.environ["SV_SWEEP_NCU"]
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()
    assert m.last_aggregation() == (2, 2), "the fused sweeps should have run"
    got = out.cpu().numpy()
    for i in range(3):
        ref = svref.sgbm(pairs[i][0], pairs[i][1], po)
        for k in range(i, nb, 3):
            assert np.array_equal(got[k], ref), "frame %d differs" % k
    m.close()
