"""D > 128 in large batches: the cost volume is computed in two disparity parts of at most 128 by the D <= 128 kernel
(k_cost2, `doff`).  Every frame of a batch against the oracle, for every D in (128, 256] and both tile widths, with
minDisparity, and against the one-part kernel of the same library (SV_COST_NOPARTS)."""
import os
import numpy as np
import pytest
import svhip
import svref
from svhip import synth
from test_gpu_parity import _params

pytestmark = pytest.mark.gpu


def _batch(W, H, D, nb, seed, **over):
    import torch
    ph, po = _params(D, 8, **over)
    pairs = [synth.stereo_pair(seed + i, W, H, D) for i in range(nb)]
    lt = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    rt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device="cuda")
    m = svhip.Matcher(W, H, ph)
    m.set_batch(nb)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.compute_gray_batch_dev(nb, lt, rt, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    m.close()
    return pairs, po, got


@pytest.mark.parametrize("D,bs,minD", [(160, 7, 0), (192, 5, 0), (224, 3, 0), (256, 7, 0), (256, 1, 0), (192, 7, -5), (256, 5, 9), (256, 9, 0)])
def test_batches_above_128_disparities(gpu, D, bs, minD):
    W, H, nb = D + 75, 37, 9
    pairs, po, got = _batch(W, H, D, nb, 4200 + D + bs, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs, min_disparity=minD)
    for i in range(nb):
        assert np.array_equal(got[i], svref.sgbm(pairs[i][0], pairs[i][1], po)), "frame %d of D=%d bs=%d minD=%d" % (i, D, bs, minD)


def test_parts_equal_the_one_part_kernel_on_a_wide_frame(gpu):
    """Several 58-column tiles and a ragged last one, 1080p width: both forms of the cost kernel end in the same disparities."""
    W, H, D, nb = 1920, 24, 256, 8
    _, _, got = _batch(W, H, D, nb, 77)
    os.environ["SV_COST_NOPARTS"] = "1"
    try:
        _, _, ref = _batch(W, H, D, nb, 77)
    finally:
        del os.environ["SV_COST_NOPARTS"]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("D", [32, 64, 96, 128])
@pytest.mark.parametrize("bs", [1, 3, 5, 7, 9, 11])
def test_batches_every_block_size_up_to_128_disparities(gpu, D, bs):
    """The large-batch form of the gray cost kernel (64-column tiles, LDS stage, hand-counted memory queue; 32-column tiles where the
    ring of a 64-column tile would pass the LDS) at every block size: nine frames, each against the oracle.  The window radius is a
    compile-time case of the kernel's row loop, so every radius is a different piece of code (and register allocation)."""
    W, H, nb = D + 83, 29 + bs, 9
    pairs, po, got = _batch(W, H, D, nb, 5100 + D + bs, block_size=bs, P1=8 * bs * bs, P2=32 * bs * bs, pre_filter_cap=31)
    for i in range(nb):
        assert np.array_equal(got[i], svref.sgbm(pairs[i][0], pairs[i][1], po)), "frame %d of D=%d bs=%d" % (i, D, bs)
