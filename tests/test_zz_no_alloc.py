"""GPU test of the allocation-free entry point sv_filter_speckles_static_dev (include/sv_hip.h): its
union-find scratch is a file-scope __device__ array of the library whose per-device address is looked up
with hipGetSymbolAddress in sv_create.  It must give, bit for bit, what sv_filter_speckles_dev gives (and
what the oracle's restatement of cv::filterSpeckles gives), on two calls in a row that share the GPU's
array, the second on a smaller frame; frames beyond SV_STATIC_MAX_PIXELS are rejected.

Collected last on purpose (file name): the round's end runs pytest -x."""
import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu


def _disparity(W, H, D, seed):
    left, right, _ = synth.stereo_pair(seed, W, H, D)
    m = svhip.Matcher(W, H, svhip.default_params(D, 8, uniqueness_ratio=25))
    d = m.compute_gray(left, right)
    return m, d


def test_static_scratch_speckle_matches_allocating_entry_point(gpu):
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    for (W, H, D, seed, max_size, max_diff) in [(400, 300, 64, 61, 80, 32), (240, 120, 64, 33, 50, 16)]:  # the second frame is smaller
        m, d = _disparity(W, H, D, seed)
        m.set_stream(stream)
        a = torch.from_numpy(d.copy()).cuda()
        b = torch.from_numpy(d.copy()).cuda()
        # the property the entry point exists for: the call allocates nothing (the context has no speckle scratch
        # of its own yet: sv_filter_speckles_dev below is what would allocate it)
        m.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        m.filter_speckles_static_dev(a, max_size, max_diff)   # static scratch: no allocation
        m.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "sv_filter_speckles_static_dev changed the free device memory"
        m.filter_speckles_dev(b, max_size, max_diff)          # scratch allocated by the context
        m.synchronize()
        got, exp = a.cpu().numpy(), b.cpu().numpy()
        assert (exp != d).any(), "the filter should have removed something"
        assert np.array_equal(got, exp), "%dx%d: static-scratch result differs in %d pixels" % (W, H, int((got != exp).sum()))
        assert np.array_equal(got, svref.filter_speckles(d, -16, max_size, max_diff))
        m.close()


def test_static_scratch_rejects_large_frames(gpu):
    import torch
    W, H, D = 2048, 1100, 32   # 2 252 800 pixels > SV_STATIC_MAX_PIXELS
    assert W * H > svhip.STATIC_MAX_PIXELS
    m = svhip.Matcher(W, H, svhip.default_params(D, 5))
    t = torch.zeros((H, W), dtype=torch.int16, device="cuda")
    with pytest.raises(svhip.SvError) as e:
        m.filter_speckles_static_dev(t, 50, 16)
    assert e.value.code == -2 and "static scratch" in str(e.value)
    m.filter_speckles_dev(t, 50, 16)   # the allocating entry point still takes it
    m.synchronize()
    m.close()
