"""Ad-hoc: a batch that does not fit fails cleanly with SV_ERR_NOMEM and the context works with a smaller one."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W, H, D = 640, 480, 128
a = svhip.Matcher(W, H, svhip.default_params(D, 8)); a.set_batch(1024)
a2 = svhip.Matcher(W, H, svhip.default_params(D, 8)); a2.set_batch(900)
b = svhip.Matcher(W, H, svhip.default_params(D, 8))
try:
    b.set_batch(1024); print("second 1024 fitted?!")
except svhip.SvError as e:
    print("expected failure:", e.code, str(e)[:120])
b.set_batch(256)
l, r, _ = synth.stereo_pair(202, W, H, D)
ref = b.compute_gray(l, r)
lt = torch.from_numpy(np.stack([l] * 256)).cuda(); rt = torch.from_numpy(np.stack([r] * 256)).cuda()
out = torch.empty((256, H, W), dtype=torch.int16, device='cuda')
b.compute_gray_batch_dev(256, lt, rt, out); b.synchronize()
print("batch after failure ok:", bool((out.cpu().numpy() == ref[None]).all()))
a.close(); a2.close(); b.close()
