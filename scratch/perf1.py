import sys, time
sys.path.insert(0,'stereo-vision_amd'); sys.path.insert(0,'oracle')
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
t=time.time(); pairs=[synth.stereo_pair(202+i,W,H,D) for i in range(4)]; print("gen %.2fs/pair"%((time.time()-t)/4))
p=svhip.default_params(D,8)
m=svhip.Matcher(W,H,p)
lt=[torch.from_numpy(a[0]).cuda() for a in pairs]; rt=[torch.from_numpy(a[1]).cuda() for a in pairs]
out=torch.empty((H,W),dtype=torch.int16,device='cuda')
m.set_stream(torch.cuda.current_stream().cuda_stream)
m.enable_timing(True)
for i in range(5): m.compute_gray_dev(lt[i%4],rt[i%4],out)
torch.cuda.synchronize()
acc={}
for i in range(20):
    m.compute_gray_dev(lt[i%4],rt[i%4],out); torch.cuda.synchronize()
    for k,v in m.timings().items(): acc[k]=acc.get(k,0)+v/20
print("single stream per-stage ms:", {k:round(v,4) for k,v in acc.items()})
m.enable_timing(False)
torch.cuda.synchronize(); t=time.time()
n=200
for i in range(n): m.compute_gray_dev(lt[i%4],rt[i%4],out)
torch.cuda.synchronize(); dt=time.time()-t
print("single stream: %.1f fps (%.3f ms/frame)"%(n/dt, dt/n*1e3))
# multi-stream
for ns in (2,4,8,16):
    ms=[svhip.Matcher(W,H,p) for _ in range(ns)]
    ss=[torch.cuda.Stream() for _ in range(ns)]
    outs=[torch.empty((H,W),dtype=torch.int16,device='cuda') for _ in range(ns)]
    for mm,s in zip(ms,ss): mm.set_stream(s.cuda_stream)
    for i in range(ns*3): ms[i%ns].compute_gray_dev(lt[i%4],rt[i%4],outs[i%ns])
    torch.cuda.synchronize(); t=time.time()
    n=100*ns
    for i in range(n): ms[i%ns].compute_gray_dev(lt[i%4],rt[i%4],outs[i%ns])
    torch.cuda.synchronize(); dt=time.time()-t
    print("%d streams: %.1f fps (%.3f ms/frame)"%(ns, n/dt, dt/n*1e3))
    for mm in ms: mm.close()
