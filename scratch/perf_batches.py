"""Throughput vs frames per launch (C2, device-resident), production heuristics."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W, H, D = 640, 480, 128
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(4)]
m = svhip.Matcher(W, H, svhip.default_params(D, 8))
m.set_stream(torch.cuda.current_stream().cuda_stream)
batches = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256, 512, 1024, 1536]
print('lib', os.environ.get('SVHIP_LIB', 'default'), 'SV_FAN_CHUNK', os.environ.get('SV_FAN_CHUNK'))
for nb in batches:
    m.set_batch(nb)
    lt = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).cuda()
    rt = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(nb)])).cuda()
    out = torch.empty((nb, H, W), dtype=torch.int16, device='cuda')
    for i in range(3): m.compute_gray_batch_dev(nb, lt, rt, out)
    torch.cuda.synchronize()
    reps = max(3, min(200, 2048 // nb))
    t = time.perf_counter()
    for i in range(reps): m.compute_gray_batch_dev(nb, lt, rt, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("batch %4d: %8.3f ms per launch  %8.1f frames/s  aggregation launches %s" % (nb, dt * 1e3, nb / dt, m.last_aggregation()), flush=True)
m.close()
