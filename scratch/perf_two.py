import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
l,r,_=synth.stereo_pair(202,W,H,D)
p=svhip.default_params(D,8)
for (ns,nb) in ((1,256),(2,128),(2,256),(3,128),(4,64),(1,512)):
    ms=[]; ins=[]; streams=[torch.cuda.Stream() for _ in range(ns)]
    for s in streams:
        m=svhip.Matcher(W,H,p); m.set_batch(nb); m.set_stream(s.cuda_stream); ms.append(m)
        lt=torch.from_numpy(np.stack([l]*nb)).cuda(); rt=torch.from_numpy(np.stack([r]*nb)).cuda(); out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
        ins.append((lt,rt,out))
    torch.cuda.synchronize()
    def step():
        for m,(lt,rt,out) in zip(ms,ins): m.compute_gray_batch_dev(nb,lt,rt,out)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0=time.perf_counter(); n=6
    for _ in range(n): step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
    print("%d streams x batch %3d: %.2f ms per round, %.0f fps"%(ns,nb,dt*1e3,ns*nb/dt), flush=True)
    for m in ms: m.close()
    del ins; torch.cuda.empty_cache()
