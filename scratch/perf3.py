import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
nb=int(sys.argv[1]) if len(sys.argv)>1 else 16
pairs=[synth.stereo_pair(202+i,W,H,D) for i in range(4)]
p=svhip.default_params(D,8)
m=svhip.Matcher(W,H,p); m.set_batch(nb)
lt=torch.from_numpy(np.stack([pairs[i%4][0] for i in range(nb)])).cuda()
rt=torch.from_numpy(np.stack([pairs[i%4][1] for i in range(nb)])).cuda()
out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
m.set_stream(torch.cuda.current_stream().cuda_stream)
for i in range(3): m.compute_gray_batch_dev(nb,lt,rt,out)
torch.cuda.synchronize()
