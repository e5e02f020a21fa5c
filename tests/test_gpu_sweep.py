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


def _oracle_many(pairs, po, workers=8):
    """svref.sgbm on many pairs, frame-parallel (ctypes releases the GIL inside the C oracle)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(lambda p: svref.sgbm(p[0], p[1], po), pairs))


def _matcher_with_env(W, H, ph, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return svhip.Matcher(W, H, ph)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("cw", [15, 7])
def test_sweep_grid_exceeds_residency(gpu, cw):
    """band b waits for band b-1 (the next lower blockIdx) through HBM.  With SV_SWEEP_NCU = 1024
    the launch holds four times as many workgroups as the chip has slots for them, so most bands start only
    when earlier workgroups have left: still bit-exact against the per-path kernels and the oracle, no timeout."""
    import torch
    W, H, D, nb, nd = 160, 100, 64, 512, 16
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(1300 + i, W, H, D) for i in range(nd)]
    lt = torch.from_numpy(np.stack([pairs[i % nd][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % nd][1] for i in range(nb)])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    m = _matcher_with_env(W, H, ph, SV_SWEEP_NCU=1024, SV_SWEEP_CW=cw)
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.sweep_order(True)   # every workgroup stamps its start: the order the hand-off relies on is checked, not assumed
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()   # reports a hand-off timeout
    assert m.last_aggregation() == (2, 2), "the fused sweeps should have run"
    launches, early, worst, worst_prev = m.sweep_order(False)
    # the rule: workgroup i -> XCD i mod 8, each XCD in blockIdx order (against blockIdx - 1, on another XCD, a workgroup of
    # this oversubscribed launch may well start 100+ us early: worst_prev, for the record)
    print("sweep dispatch order, %d waves: worst lead over blockIdx-8 %d ticks, over blockIdx-1 %d ticks" % (cw, worst, worst_prev))
    assert launches == 2 and early == 0, "workgroups started before their XCD's previous workgroup: %d of them, by up to %d ticks" % (early, worst)
    got = out.cpu().numpy()
    refs = _oracle_many(pairs, po)
    for i in range(nb):
        assert np.array_equal(got[i], refs[i % nd]), "frame %d differs" % i
    m.close()


def test_sweep_bench_shape_every_frame(gpu):
    """The shape bench.py streams (640x480 D=128, 16-wave workgroups, 32 frames per group) on two groups with 64
    distinct pairs, EVERY output frame against the oracle."""
    import torch
    W, H, D, nb = 640, 480, 128, 64
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(nb)]
    lt = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    rt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    m = _matcher_with_env(W, H, ph, SV_SWEEP_NCU=16, SV_SWEEP_CW=15)   # 8 bands -> 2 groups of 32 frames
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()
    assert m.last_aggregation() == (2, 2)
    got = out.cpu().numpy()
    refs = _oracle_many(pairs, po, workers=16)
    bad = [i for i in range(nb) if not np.array_equal(got[i], refs[i])]
    assert not bad, "frames %s differ from the oracle" % bad
    m.close()


def test_sweep_frames_per_group_clamp(gpu):
    """A group's block of C must stay below 2^34 bytes (32-bit dword offsets in the kernel): at 1920x1080 D=256
    that is 17 frames.  20 frames on ONE workgroup pipeline trip the clamp (two groups of 17 and 3 frames, more
    workgroups than SV_SWEEP_NCU asked for): every frame against the per-path kernels, one against the oracle."""
    import torch
    W, H, D, nb = 1920, 1080, 256, 20
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(404 + i, W, H, D) for i in range(2)]
    lt = torch.from_numpy(np.stack([pairs[i % 2][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % 2][1] for i in range(nb)])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    m = _matcher_with_env(W, H, ph, SV_SWEEP_NCU=20)   # 39 bands of 28 rows, two workgroups per CU -> one pipeline
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.set_debug(3)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()
    assert m.last_aggregation() == (2, 2)
    got = out.cpu().numpy()
    m.set_debug(4)   # plain per-path launches
    out2 = torch.empty((2, H, W), dtype=torch.int16, device="cuda")
    m.compute_gray_batch_dev(2, lt, rt, out2)
    m.synchronize()
    exp = out2.cpu().numpy()
    assert np.array_equal(exp[0], svref.sgbm(pairs[0][0], pairs[0][1], po))
    for i in range(nb):
        assert np.array_equal(got[i], exp[i % 2]), "frame %d differs" % i
    m.close()


def test_two_contexts_sweep_concurrently(gpu):
    """Two contexts on two streams, both in the fused-sweep path at the same time (each launch alone would fill
    the chip): their band hand-offs and progress counters are per context, results bit-exact."""
    import torch
    W, H, D, nb, nd = 320, 240, 64, 64, 8
    ph, po = _params(D, 8)
    sets = []
    for k in range(2):
        pairs = [synth.stereo_pair(1500 + 10 * k + i, W, H, D) for i in range(nd)]
        st = torch.cuda.Stream()
        m = svhip.Matcher(W, H, ph)
        m.set_batch(nb)
        m.set_debug(3)
        m.set_stream(st.cuda_stream)
        lt = torch.from_numpy(np.stack([pairs[i % nd][0] for i in range(nb)])).cuda()
        rt = torch.from_numpy(np.stack([pairs[i % nd][1] for i in range(nb)])).cuda()
        out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
        sets.append((m, st, pairs, lt, rt, out))
    torch.cuda.synchronize()
    for _ in range(3):   # several rounds so that the launches of the two streams really overlap
        for m, st, pairs, lt, rt, out in sets:
            m.compute_gray_batch_dev(nb, lt, rt, out)
    for m, st, pairs, lt, rt, out in sets:
        m.synchronize()
        assert m.last_aggregation() == (2, 2)
        got = out.cpu().numpy()
        refs = _oracle_many(pairs, po)
        for i in range(nb):
            assert np.array_equal(got[i], refs[i % nd]), "frame %d differs" % i
        m.close()


def test_sweep_timeout_falls_back_to_per_path_kernels(gpu):
    """a sweep that times out on its neighbour band (injected here: SV_SWEEP_FAULT makes the next sweep launch
    leave the error word a timed-out band would) must not cost the caller the batch.  Device entry point: sv_synchronize
    reports it once and the context stays on the per-path kernels; the repeated call is bit-exact.  Batch-through-host entry
    point: sv_wait_batch_host reruns the batch from the slot's device buffers itself and returns the right frames."""
    import torch
    W, H, D, nb, nd = 320, 240, 64, 40, 4
    ph, po = _params(D, 8)
    pairs = [synth.stereo_pair(1700 + i, W, H, D) for i in range(nd)]
    refs = _oracle_many(pairs, po)
    lefts = np.stack([pairs[i % nd][0] for i in range(nb)])
    rights = np.stack([pairs[i % nd][1] for i in range(nb)])
    # (a) device-resident batch
    m = _matcher_with_env(W, H, ph, SV_SWEEP_FAULT=1)
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    lt, rt = torch.from_numpy(lefts).cuda(), torch.from_numpy(rights).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    m.compute_gray_batch_dev(nb, lt, rt, out)
    assert m.last_aggregation()[1] == 2, "the fused sweeps should have run"
    with pytest.raises(svhip.SvError) as e:
        m.synchronize()
    assert "per-path" in str(e.value)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    m.synchronize()
    assert m.last_aggregation()[1] == 0, "after a sweep timeout the context must stay off the sweeps"
    got = out.cpu().numpy()
    for i in range(nb):
        assert np.array_equal(got[i], refs[i % nd]), "frame %d differs" % i
    m.close()
    # (b) batch through host buffers: the wait retries by itself
    m = _matcher_with_env(W, H, ph, SV_SWEEP_FAULT=1)
    m.set_batch(nb)
    outs = np.full((nb, H, W), -32768, np.int16)
    t = m.submit_gray_batch([lefts[i] for i in range(nb)], [rights[i] for i in range(nb)], [outs[i] for i in range(nb)])
    m.wait_batch(t)
    assert m.last_aggregation()[1] == 0
    for i in range(nb):
        assert np.array_equal(outs[i], refs[i % nd]), "frame %d differs" % i
    m.close()


@pytest.mark.parametrize("P2,sweeps", [(29900, 2), (30100, 0)])
def test_sweep_value_range_gate(gpu, P2, sweeps):
    """The sweeps order their u16 values with the packed binary16 three-input minimum, exact below 0x7C00: with 3x3 blocks,
    cap 63 and P1 = 100 the largest cost + P1 is 31 701 at P2 = 29 900 (the sweeps run, every L within 2000 of the limit) and
    31 901 at P2 = 30 100 (sv_check_params still accepts it; the per-path kernels take it)."""
    W, H, D = 170, 66, 64
    rng = np.random.default_rng(5 + P2)
    left = rng.integers(0, 256, (H, W), dtype=np.uint8)
    right = rng.integers(0, 256, (H, W), dtype=np.uint8)
    ph, po = _params(D, 8, block_size=3, pre_filter_cap=63, P1=100, P2=P2)
    ref, st = svref.sgbm(left, right, po, stages=("C", "mins"))
    assert int(st["C"].max()) + 100 > 0x7800, "the costs should sit at the top of the range"
    m = svhip.Matcher(W, H, ph)
    m.set_debug(3)
    got = m.compute_gray(left, right)
    assert m.last_aggregation()[1] == sweeps
    assert np.array_equal(m.debug_plane(2)[:, D:], st["mins"][:, D:]), "minS differs"
    assert np.array_equal(got, ref)
    m.close()


def test_packed_f16_minimum_is_the_integer_minimum(gpu):
    """The library's own build of pk_min3 against the integer minimum on every pair of [0, 0x7C00)^2 (both halves, hashed third
    operand): the claim the sweeps' arithmetic rests on, under the flags libsv_hip was compiled with."""
    m = svhip.Matcher(96, 40, svhip.default_params(32, 8))
    assert m.selftest_min3() == 0
    m.close()
