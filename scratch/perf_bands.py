import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,D=640,128
for H in (60,120,240):
    l,r,_=synth.stereo_pair(202,W,H,D)
    p=svhip.default_params(D,8)
    for nb in (1,64):
        lt=torch.from_numpy(np.stack([l]*nb)).cuda(); rt=torch.from_numpy(np.stack([r]*nb)).cuda(); out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
        m=svhip.Matcher(W,H,p); m.set_batch(nb); m.set_debug(3)
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(3): m.compute_gray_batch_dev(nb,lt,rt,out)
        m.synchronize(); m.enable_timing(True)
        for _ in range(5): m.compute_gray_batch_dev(nb,lt,rt,out)
        m.synchronize(); t=m.timings(); m.close()
        print("H=%3d batch %3d: agg %.3f ms"%(H,nb,t['aggregation']), flush=True)
