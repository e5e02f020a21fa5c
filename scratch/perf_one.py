import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
nb=int(sys.argv[1]); mode=int(sys.argv[2]); reps=int(sys.argv[3]) if len(sys.argv)>3 else 2
l,r,_=synth.stereo_pair(202,W,H,D)
p=svhip.default_params(D,8)
m=svhip.Matcher(W,H,p); m.set_batch(nb); m.set_debug(mode)
lt=torch.from_numpy(np.stack([l]*nb)).cuda(); rt=torch.from_numpy(np.stack([r]*nb)).cuda(); out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
m.set_stream(torch.cuda.current_stream().cuda_stream)
for i in range(reps): m.compute_gray_batch_dev(nb,lt,rt,out)
m.synchronize()
