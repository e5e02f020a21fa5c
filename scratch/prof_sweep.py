import sys, os, ctypes as C
os.environ["SVHIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd', 'libsv_hip_prof.so')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np, torch
import svhip
from svhip import synth
W,H,D=640,480,128
nb=int(sys.argv[1]); cw=int(os.environ.get("SV_SWEEP_CW","7"))
l,r,_=synth.stereo_pair(202,W,H,D)
p=svhip.default_params(D,8)
m=svhip.Matcher(W,H,p); m.set_batch(nb); m.set_debug(3)
lt=torch.from_numpy(np.stack([l]*nb)).cuda(); rt=torch.from_numpy(np.stack([r]*nb)).cuda(); out=torch.empty((nb,H,W),dtype=torch.int16,device='cuda')
m.set_stream(torch.cuda.current_stream().cuda_stream)
for i in range(2): m.compute_gray_batch_dev(nb,lt,rt,out)
m.synchronize()
L=svhip.load(); L.sv_debug_sweep_prof.argtypes=[C.c_void_p,C.c_void_p,C.c_int]
n=L.sv_debug_sweep_prof(m._h,None,0)
buf=np.zeros(n,np.uint64)
L.sv_debug_sweep_prof(m._h,buf.ctypes.data_as(C.c_void_p),n)
a=buf.reshape(2,-1,cw+1,4).astype(np.float64)
for sw in (0,1):
    comp=a[sw,:,:cw,:].reshape(-1,4); serv=a[sw,:,cw,:]
    tot=comp.sum(1).mean()
    print("sweep %d: compute waves: mean cycles total %.0f | lds-read-wait %.1f%% compute %.1f%% write+issue %.1f%% barrier %.1f%%"%(sw+1,tot,*(100*comp.mean(0)/tot)))
    print("         service wave : work %.1f%% barrier %.1f%% (total %.0f)"%(100*serv[:,0].mean()/(serv[:,0].mean()+serv[:,3].mean()),100*serv[:,3].mean()/(serv[:,0].mean()+serv[:,3].mean()),serv[:,0].mean()+serv[:,3].mean()))
    # per band
    nbands=(H+4*cw-1)//(4*cw)
    wg=a[sw].reshape(-1,nbands,cw+1,4)
    print("         per-band compute barrier share:", np.round(100*wg[:,:,:cw,3].mean((0,2))/wg[:,:,:cw,:].sum(3).mean((0,2)),1))
    print("         per-band service work cycles :", np.round(wg[:,:,cw,0].mean(0)/1e3,0))
    print("         per-band compute-wave kcycles [read, compute, write, barrier]:")
    for b in range(nbands): print("            band %2d:"%b, np.round(wg[:,b,:cw,:].mean((0,1))/1e3,0), " service [work, barrier]:", np.round(wg[:,b,cw,[0,3]].mean(0)/1e3,0))
