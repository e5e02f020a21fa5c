import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
l,r,_=synth.stereo_pair(202,W,H,D)
rng=np.random.default_rng(0)
lrgb,rrgb=synth.rgb_from_gray(rng,l),synth.rgb_from_gray(rng,r)
def t(f,n=20):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for sp in (0,50):
    p=svhip.default_params(D,8,speckle_window_size=sp,speckle_range=16)
    m=svhip.Matcher(W,H,p)
    print("speckle",sp,"compute_gray host %.3f ms"%t(lambda:m.compute_gray(l,r)), "compute_rgb host (no rectify) %.3f ms"%t(lambda:m.compute_rgb(lrgb,rrgb,rectify=False)))
    d=m.compute_gray(l,r)
    print("  vis host %.3f ms"%t(lambda:m.disparity_vis(d,3)), "wls host %.3f ms"%t(lambda:m.filter_wls(d,lrgb,vis=m.disparity_vis(d,3))))
    mx,my=synth.rectify_maps(W,H,0); mapper=np.stack([mx,my],-1)
    Q=np.eye(4); R=np.eye(3)
    print("  depth host %.3f ms"%t(lambda:m.depth(d,mapper,Q,R)))
    lt=torch.from_numpy(l).cuda(); rt=torch.from_numpy(r).cuda(); out=torch.empty((H,W),dtype=torch.int16,device='cuda')
    print("  compute_gray dev %.3f ms"%t(lambda:m.compute_gray_dev(lt,rt,out)))
    m.close()

p=svhip.default_params(D,8)
m=svhip.Matcher(W,H,p)
import collections
def pipe(n=200, depth=6):
    q=collections.deque(); done=0
    t0=time.perf_counter()
    for i in range(n):
        if len(q)==depth: m.wait(q.popleft()); done+=1
        q.append(m.submit_gray(l,r))
    while q: m.wait(q.popleft())
    return n/(time.perf_counter()-t0)
pipe(20)
for d in (1,2,4,7): print("async host pipeline depth %d: %.0f fps"%(d,pipe(300,d)))
print("synchronous host: %.0f fps"%(1e3/t(lambda:m.compute_gray(l,r))))
m.close()
