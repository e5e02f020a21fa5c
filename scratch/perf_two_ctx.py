"""Experiment: one context with 1024 pairs per launch vs two contexts with 512 each on two streams (do the
fill / drain bubbles of one stream's sweeps get filled by the other stream's kernels?)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W, H, D = 640, 480, 128
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(8)]
def run(nctx, nb, reps=8):
    ctxs = []
    for k in range(nctx):
        m = svhip.Matcher(W, H, svhip.default_params(D, 8)); m.set_batch(nb)
        s = torch.cuda.Stream(); m.set_stream(s.cuda_stream)
        lt = torch.from_numpy(np.stack([pairs[i % 8][0] for i in range(nb)])).cuda()
        rt = torch.from_numpy(np.stack([pairs[i % 8][1] for i in range(nb)])).cuda()
        out = torch.empty((nb, H, W), dtype=torch.int16, device='cuda')
        ctxs.append((m, lt, rt, out))
    for _ in range(2):
        for m, lt, rt, out in ctxs: m.compute_gray_batch_dev(nb, lt, rt, out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for m, lt, rt, out in ctxs: m.compute_gray_batch_dev(nb, lt, rt, out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("%d context(s) x %4d pairs: %8.1f frames/s" % (nctx, nb, reps * nctx * nb / dt), flush=True)
    for m, *_ in ctxs: m.close()
    del ctxs; torch.cuda.empty_cache()
run(1, 1024); run(2, 512); run(1, 512); run(2, 256); run(4, 256)
