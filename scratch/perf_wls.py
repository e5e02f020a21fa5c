import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
nb=int(sys.argv[1]) if len(sys.argv)>1 else 1
l,r,_=synth.stereo_pair(202,W,H,D)
rng=np.random.default_rng(0)
lrgb=synth.rgb_from_gray(rng,l)
p=svhip.default_params(D,8,speckle_window_size=50,speckle_range=16)
m=svhip.Matcher(W,H,p); m.set_batch(nb)
lt=torch.from_numpy(np.stack([l]*nb)).cuda(); rt=torch.from_numpy(np.stack([r]*nb)).cuda(); out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
gt=torch.from_numpy(np.stack([lrgb]*nb)).cuda(); o2=torch.empty_like(out)
vis=torch.zeros((nb,H,W),dtype=torch.uint8,device='cuda'); v2=torch.empty_like(vis)
m.set_stream(torch.cuda.current_stream().cuda_stream)
for i in range(3):
    m.compute_gray_batch_dev(nb,lt,rt,out)
    m.filter_wls_dev(nb,out,gt,o2,vis,v2)
torch.cuda.synchronize()
reps = 10
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps):
    m.filter_wls_dev(nb, out, gt, o2, vis, v2)
e1.record(); torch.cuda.synchronize()
print("WLS (16-bit map + vis image) over %d frames: %.3f ms per call, %.2f us per frame%s" % (nb, e0.elapsed_time(e1) / reps, e0.elapsed_time(e1) / reps / nb * 1e3,
      "  [64-column tiles]" if os.environ.get("SV_FGS_TC64") else ""))
