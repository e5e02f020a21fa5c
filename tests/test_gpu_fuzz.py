"""Seeded random batches through the frame-batched entry point (sv_compute_gray_batch_dev): random
geometry, batch size, workgroup-pipeline count (SV_SWEEP_NCU: few groups, so every group streams
several frames and the last one holds fewer), both sweep workgroup shapes, speckle / median / LR
flags - every frame against the oracle, bit-exact.  Plus the bench configuration (C2) with a batch
that the production heuristics send through the fused sweeps with ragged groups."""
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


def _matcher(W, H, ph, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return svhip.Matcher(W, H, ph)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_random_batches_bit_exact(gpu):
    import torch
    rng = np.random.default_rng(777)
    for case in range(14):
        D = int(rng.choice([32, 64, 96, 128, 160, 256]))
        bs = int(rng.choice([3, 5, 7, 9]))
        W = D + bs + int(rng.integers(1, 120))
        H = bs + int(rng.integers(1, 140))
        nb = int(rng.integers(2, 10))
        paths = int(rng.choice([5, 8]))
        flags = int(rng.choice([svhip.FLAGS_REPO, svhip.FLAGS_HH]))
        flags |= int(rng.choice([0, svhip.FLAG_NO_MEDIAN])) | int(rng.choice([0, 0, svhip.FLAG_NO_LRCHECK]))
        speckle = int(rng.choice([0, 20]))
        env = {"SV_SWEEP_NCU": str(int(rng.integers(2, 13)))}
        cw = int(rng.choice([0, 7, 15]))
        if cw and not (cw == 15 and D > 128):
            env["SV_SWEEP_CW"] = str(cw)
        mode = int(rng.choice([0, 3]))
        what = "case %d: W=%d H=%d D=%d bs=%d nb=%d paths=%d flags=%#x speckle=%d env=%s mode=%d" % (
            case, W, H, D, bs, nb, paths, flags, speckle, env, mode)
        ph, po = _params(D, paths, flags=flags, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs,
                         speckle_window_size=speckle, speckle_range=2 if speckle else 0)
        pairs = [synth.stereo_pair(3000 + 16 * case + i, W, H, D) for i in range(nb)]
        lt = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
        rt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
        out = torch.full((nb, H, W), 12345, dtype=torch.int16, device="cuda")
        m = _matcher(W, H, ph, env)
        m.set_batch(nb)
        m.set_debug(mode)
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        m.compute_gray_batch_dev(nb, lt, rt, out)
        m.synchronize()
        got = out.cpu().numpy()
        for i in range(nb):
            assert np.array_equal(got[i], svref.sgbm(pairs[i][0], pairs[i][1], po)), "%s, frame %d" % (what, i)
        m.close()


def test_c2_batch_through_production_sweeps(gpu):
    """640x480 D=128 8 paths, 70 pairs in one launch (32 workgroup pipelines: groups of 3 and 2 frames),
    the kernels bench.py times; 5 distinct pairs, every frame against the oracle."""
    import torch
    W, H, D, nb, nd = 640, 480, 128, 70, 5
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(nd)]
    refs = [svref.sgbm(p[0], p[1], po) for p in pairs]
    lt = torch.from_numpy(np.stack([pairs[i % nd][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % nd][1] for i in range(nb)])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    m = svhip.Matcher(W, H, ph)
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()
    assert m.last_aggregation() == (2, 2), "the fused sweeps should have run"
    got = out.cpu().numpy()
    for i in range(nb):
        assert np.array_equal(got[i], refs[i % nd]), "frame %d differs" % i
    m.close()


@pytest.mark.parametrize("W,H,D", [(200, 8000, 32), (40000, 8, 32), (40, 65535, 32), (2049, 1537, 256)])
def test_extreme_geometries(gpu, W, H, D):
    """Very tall (more row bands than CUs), widest and tallest accepted, and a large ragged D=256 frame:
    production choice, per-path launches and forced sweeps against the oracle."""
    left, right, _ = synth.stereo_pair(7, W, H, D)
    ph, po = _params(D, 8)
    ref = svref.sgbm(left, right, po)
    m = svhip.Matcher(W, H, ph)
    for mode in (0, 4, 3):
        m.set_debug(mode)
        assert np.array_equal(m.compute_gray(left, right), ref), "debug mode %d" % mode
    m.close()


def test_too_wide_is_rejected(gpu):
    with pytest.raises(svhip.SvError) as e:
        svhip.Matcher(40001, 8, svhip.default_params(32, 8))
    assert e.value.code == -1   # SV_ERR_INVALID_ARG


def test_batch_above_1024_small_frames(gpu):
    """sv_set_batch accepts up to 2048 pairs: 1500 small pairs in one launch (fused sweeps, many frames per workgroup
    pipeline), 6 distinct pairs looped, every frame against the oracle's result for its pair."""
    import torch
    W, H, D, nb = 112, 40, 32, 1500
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(8100 + i, W, H, D) for i in range(6)]
    refs = [svref.sgbm(p[0], p[1], po) for p in pairs]
    lt = torch.from_numpy(np.stack([pairs[i % 6][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % 6][1] for i in range(nb)])).cuda()
    out = torch.full((nb, H, W), 12345, dtype=torch.int16, device="cuda")
    m = svhip.Matcher(W, H, ph)
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()
    assert m.last_aggregation() == (2, 2)   # two launches, both fused sweeps
    got = out.cpu().numpy()
    for i in range(nb):
        assert np.array_equal(got[i], refs[i % 6]), "frame %d" % i
    with pytest.raises(svhip.SvError):
        m.set_batch(2049)
    m.close()
