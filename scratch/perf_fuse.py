import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
l,r,_=synth.stereo_pair(202,W,H,D)
p=svhip.default_params(D,8)
for nb in (1,2,3,4,6,8,16):
    lt=torch.from_numpy(np.stack([l]*nb)).cuda(); rt=torch.from_numpy(np.stack([r]*nb)).cuda(); out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
    res=[]
    for mode in (1,2):   # 1: unfused (debug), 2: fused
        m=svhip.Matcher(W,H,p); m.set_batch(nb); m.set_debug(mode)
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(5): m.compute_gray_batch_dev(nb,lt,rt,out)
        torch.cuda.synchronize(); t0=time.perf_counter()
        n=40
        for _ in range(n): m.compute_gray_batch_dev(nb,lt,rt,out)
        torch.cuda.synchronize(); res.append((time.perf_counter()-t0)/n*1e3)
        m.close()
    print("batch %2d: unfused %.3f ms  fused %.3f ms  (per frame %.3f / %.3f)"%(nb,res[0],res[1],res[0]/nb,res[1]/nb), flush=True)
