"""Stage times of the C2 batch (no output checks; for diagnostic builds): perf_stages.py [batch] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stereo-vision_amd"))
import numpy as np, torch
import svhip
from svhip import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W, H, D = 640, 480, 128
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(16)]
m = svhip.Matcher(W, H, svhip.default_params(D, 8))
m.set_batch(B)
lt = torch.from_numpy(np.stack([pairs[i % 16][0] for i in range(B)])).cuda()
rt = torch.from_numpy(np.stack([pairs[i % 16][1] for i in range(B)])).cuda()
out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
m.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(2):
    m.compute_gray_batch_dev(B, lt, rt, out)
m.enable_timing(True)
for _ in range(steps):
    m.compute_gray_batch_dev(B, lt, rt, out)
m.synchronize()
t = m.timings()
print(os.environ.get("SVHIP_LIB", "default"), "B=%d" % B, {k: round(v, 3) for k, v in t.items()}, "agg", [round(x, 3) for x in m.aggregation_timings()],
      "fps %.0f" % (B / t["total"] * 1e3))
