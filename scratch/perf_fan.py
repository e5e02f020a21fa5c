import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
for (W,H,D,batches) in ((640,480,128,(1,2,4,8,12,16)),(320,240,64,(1,4,16)),(1920,1080,256,(1,2))):
    l,r,_=synth.stereo_pair(202,W,H,D)
    p=svhip.default_params(D,8)
    for nb in batches:
        lt=torch.from_numpy(np.stack([l]*nb)).cuda(); rt=torch.from_numpy(np.stack([r]*nb)).cuda(); out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
        res=[]
        for mode in (4,2,5,0):
            m=svhip.Matcher(W,H,p); m.set_batch(nb); m.set_debug(mode)
            m.set_stream(torch.cuda.current_stream().cuda_stream)
            for _ in range(5): m.compute_gray_batch_dev(nb,lt,rt,out)
            m.synchronize(); t0=time.perf_counter(); n=30
            for _ in range(n): m.compute_gray_batch_dev(nb,lt,rt,out)
            m.synchronize(); res.append((time.perf_counter()-t0)/n*1e3); m.close()
        print("%dx%d D=%d batch %2d: per-path %.3f ms  fusedWTA %.3f  fan-out %.3f  auto %.3f  (auto %.0f fps)"%(W,H,D,nb,*res,nb/res[3]*1e3), flush=True)
