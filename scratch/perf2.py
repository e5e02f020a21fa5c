import sys, time
sys.path.insert(0,'stereo-vision_amd'); sys.path.insert(0,'oracle')
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
pairs=[synth.stereo_pair(202+i,W,H,D) for i in range(8)]
p=svhip.default_params(D,8)
for nb in (1,2,4,8,16,32):
    m=svhip.Matcher(W,H,p); m.set_batch(nb)
    lt=torch.from_numpy(np.stack([pairs[i%8][0] for i in range(nb)])).cuda()
    rt=torch.from_numpy(np.stack([pairs[i%8][1] for i in range(nb)])).cuda()
    out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    m.enable_timing(True)
    for i in range(3): m.compute_gray_batch_dev(nb,lt,rt,out)
    torch.cuda.synchronize()
    acc={}
    R=10
    for i in range(R):
        m.compute_gray_batch_dev(nb,lt,rt,out); torch.cuda.synchronize()
        for k,v in m.timings().items(): acc[k]=acc.get(k,0)+v/R
    m.enable_timing(False)
    torch.cuda.synchronize(); t=time.time()
    n=max(4, 200//nb)
    for i in range(n): m.compute_gray_batch_dev(nb,lt,rt,out)
    torch.cuda.synchronize(); dt=time.time()-t
    V=(W-D)*H*D*2
    print("batch %2d: %.1f fps; per-batch ms %s; agg launch %.1f us -> %.2f TB/s algorithmic"%(nb, n*nb/dt, {k:round(v,3) for k,v in acc.items()}, acc['aggregation']/8*1e3, 3*V*nb/(acc['aggregation']/8*1e-3)/1e12))
    m.close()
