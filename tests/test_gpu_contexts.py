"""Context lifetime rules of the boundary: several contexts in one process used from different
threads at the same time, parameter pushes that change the geometry of the volumes (updateCUDAParams,
StereoMatcher.cpp:271-289, with numberOfDisparities honoured), batch resizing, reuse after an error."""
import threading

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


def test_two_contexts_from_two_threads(gpu):
    import torch
    specs = [(200, 90, 64, 8), (163, 71, 32, 5)]
    work, errors = [], []
    for W, H, D, paths in specs:
        ph, po = _params(D, paths)
        pairs = [synth.stereo_pair(300 + i, W, H, D) for i in range(3)]
        refs = [svref.sgbm(p[0], p[1], po) for p in pairs]
        m = svhip.Matcher(W, H, ph)
        s = torch.cuda.Stream()
        m.set_stream(s.cuda_stream)
        work.append((m, pairs, refs, s))

    def run(k):
        m, pairs, refs, _ = work[k]
        try:
            for it in range(12):
                i = it % len(pairs)
                if not np.array_equal(m.compute_gray(pairs[i][0], pairs[i][1]), refs[i]):
                    errors.append("context %d iteration %d differs" % (k, it))
        except Exception as e:  # noqa: BLE001
            errors.append("context %d: %s" % (k, e))

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(work))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for m, _, _, _ in work:
        m.close()
    assert not errors, errors


def test_parameter_pushes_and_batch_resizing(gpu):
    import torch
    W, H = 230, 80
    left, right, _ = synth.stereo_pair(21, W, H, 96)
    ph, po = _params(64, 8)
    m = svhip.Matcher(W, H, ph)
    assert np.array_equal(m.compute_gray(left, right), svref.sgbm(left, right, po))
    # the GUI's sliders: every field, the number of disparities included (volumes are reallocated)
    for D, bs, uniq, cap, paths in [(96, 5, 10, 31, 8), (32, 9, 0, 63, 5), (128, 3, 25, 15, 8), (64, 7, 15, 63, 8)]:
        ph, po = _params(D, paths, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs, uniqueness_ratio=uniq, pre_filter_cap=cap)
        m.set_params(ph)
        assert np.array_equal(m.compute_gray(left, right), svref.sgbm(left, right, po)), (D, bs, uniq, cap, paths)
    # a rejected push leaves the context as it was
    bad = svhip.default_params(64, 8, block_size=8)
    with pytest.raises(svhip.SvError):
        m.set_params(bad)
    assert np.array_equal(m.compute_gray(left, right), svref.sgbm(left, right, po))
    # batch resizing up and down
    lt, rt = torch.from_numpy(np.stack([left] * 6)).cuda(), torch.from_numpy(np.stack([right] * 6)).cuda()
    out = torch.empty((6, H, W), dtype=torch.int16, device="cuda")
    ref = svref.sgbm(left, right, po)
    for nb in (6, 2, 4):
        m.set_batch(nb)
        m.compute_gray_batch_dev(nb, lt, rt, out)
        m.synchronize()
        assert all(np.array_equal(out[i].cpu().numpy(), ref) for i in range(nb)), nb
    m.close()


def test_failed_set_batch_leaves_a_usable_context(gpu):
    """a sv_set_batch that runs out of memory must not leave NULL volumes behind a batch
    size that passes the checks: the context keeps its previous batch, computes correctly with it, and
    a call with more frames is rejected with an error code (never a launch on a missing volume)."""
    import torch
    W, H, D = 640, 480, 128
    hogs = []
    try:
        # take HBM until a 1024-pair context (140 GB) cannot fit any more
        for _ in range(3):
            h = svhip.Matcher(W, H, svhip.default_params(D, 8))
            try:
                h.set_batch(900)
            except svhip.SvError:
                h.close()
                break
            hogs.append(h)
        m = svhip.Matcher(W, H, svhip.default_params(D, 8))
        m.set_batch(4)
        with pytest.raises(svhip.SvError) as e:
            m.set_batch(1024)
        assert e.value.code == -6, e.value
        l, r, _ = synth.stereo_pair(202, W, H, D)
        ref = svhip.Matcher(W, H, svhip.default_params(D, 8)).compute_gray(l, r)
        lt = torch.from_numpy(np.stack([l] * 8)).cuda()
        rt = torch.from_numpy(np.stack([r] * 8)).cuda()
        out = torch.empty((8, H, W), dtype=torch.int16, device="cuda")
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        m.compute_gray_batch_dev(4, lt, rt, out)     # right after the failed call: the previous batch still works
        m.synchronize()
        assert np.array_equal(out[:4].cpu().numpy(), np.stack([ref] * 4))
        with pytest.raises(svhip.SvError):
            m.compute_gray_batch_dev(8, lt, rt, out)  # more than the batch it kept: an error, not a fault
        m.close()
    finally:
        for h in hogs:
            h.close()
