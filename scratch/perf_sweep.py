import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
l,r,_=synth.stereo_pair(202,W,H,D)
p=svhip.default_params(D,8)
batches=[int(a) for a in sys.argv[1:]] or [1,8,16,32,64,128,256]
for nb in batches:
    lt=torch.from_numpy(np.stack([l]*nb)).cuda(); rt=torch.from_numpy(np.stack([r]*nb)).cuda(); out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
    res=[]; stages=[]
    for mode in (4,3):   # 4: per-path kernels, 3: fused sweeps
        m=svhip.Matcher(W,H,p); m.set_batch(nb); m.set_debug(mode)
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(3): m.compute_gray_batch_dev(nb,lt,rt,out)
        m.synchronize(); m.enable_timing(True); t0=time.perf_counter()
        n=max(3, 64//nb)
        for _ in range(n): m.compute_gray_batch_dev(nb,lt,rt,out)
        m.synchronize(); res.append((time.perf_counter()-t0)/n*1e3)
        stages.append(m.timings())
        m.close()
    print("batch %3d: per-path %.3f ms (%.0f fps)  sweeps %.3f ms (%.0f fps)  agg ms %.3f -> %.3f  cost %.3f"%(nb,res[0],nb/res[0]*1e3,res[1],nb/res[1]*1e3,stages[0]['aggregation'],stages[1]['aggregation'],stages[1]['cost']), flush=True)
