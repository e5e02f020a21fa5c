"""GPU parity of the world image (sv_world_image_*, SFM.cpp:1284-1347) against the CPU oracle.
FP64 on both sides in the same order of operations; sqrt / atan2 come from different maths
libraries, so the stated tolerance is 1 float ulp (rtol 2e-7) rather than 0."""
import numpy as np
import pytest

import svref
import svhip
from svhip import synth

pytestmark = pytest.mark.gpu

from test_oracle_world import _scene


@pytest.mark.parametrize("W,H,D", [(160, 96, 64), (640, 480, 128)])
def test_world_image_matches_oracle(gpu, W, H, D):
    d, mapper, Q, R, Hroot = _scene(W, H, D, seed=10)
    m = svhip.Matcher(W, H, svhip.default_params(D, 8))
    cart, cyl = m.world_image(d, Q, R, Hroot, mapper=mapper)
    ecart, ecyl = svref.world_image(d, mapper, Q, R, Hroot)
    assert np.array_equal(cart == 0, ecart == 0) and np.array_equal(cyl == 0, ecyl == 0)
    assert np.array_equal(cart, ecart)                       # no library call on this side: exact
    assert np.allclose(cyl, ecyl, rtol=2e-7, atol=0)
    assert (ecart[..., 2] != 0).mean() > 0.3
    m.close()


def test_world_image_device_mapper_and_errors(gpu):
    W, H, D = 320, 240, 64
    d, _, Q, R, Hroot = _scene(W, H, D, seed=11)
    K, dist, Rr, P = synth.camera(W, H, 0)
    m = svhip.Matcher(W, H, svhip.default_params(D, 8))
    with pytest.raises(svhip.SvError) as e:
        m.world_image(d, Q, R, Hroot)                         # no mapper yet
    assert e.value.code == -5
    m.update_mapper(0, K, dist, Rr, P)
    cart, cyl = m.world_image(d, Q, R, Hroot)                 # MapperL made on the device
    ecart, ecyl = svref.world_image(d, m.mapper(0), Q, R, Hroot)
    assert np.array_equal(cart, ecart) and np.allclose(cyl, ecyl, rtol=2e-7, atol=0)
    m.close()
