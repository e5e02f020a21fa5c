"""Single-frame latency (device-resident): wall time per call over many back-to-back calls vs. per-call
with a sync after each, at C2 and C1."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
for (W, H, D) in ((640, 480, 128), (320, 240, 64)):
    l, r, _ = synth.stereo_pair(202, W, H, D)
    m = svhip.Matcher(W, H, svhip.default_params(D, 8))
    lt, rt = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    out = torch.empty((H, W), dtype=torch.int16, device='cuda')
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(20): m.compute_gray_dev(lt, rt, out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(200): m.compute_gray_dev(lt, rt, out)
    torch.cuda.synchronize()
    a = (time.perf_counter() - t) / 200
    t = time.perf_counter()
    for i in range(200):
        m.compute_gray_dev(lt, rt, out); torch.cuda.synchronize()
    b = (time.perf_counter() - t) / 200
    print("%dx%d D=%d: back-to-back %.3f ms/frame, call+sync %.3f ms" % (W, H, D, a * 1e3, b * 1e3), flush=True)
    m.close()
