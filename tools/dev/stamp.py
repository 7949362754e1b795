import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from slam3d_amd import particlefilter as pfm, _lib
import _synth
n=16000000
pf=pfm.DeviceParticleFilterLoc(n, seed=1)
pf.set_force_branch(1)
bx,truth,rng=_synth.synthetic_scene(40)
pf.depositVio(1.0,float(truth[0,0]),float(truth[0,1]),float(truth[0,2]),0.0)
pf.depositRange(float(bx[0,0]),float(bx[0,1]),float(bx[0,2]),float(rng[0,0]),0.1)
for k in range(10):
    pf.depositVio(1.0+k,float(truth[k,0]),float(truth[k,1]),float(truth[k,2]),0.0)
    pf.depositRange(float(bx[k%8,0]),float(bx[k%8,1]),float(bx[k%8,2]),float(rng[k,0]),0.1)
pf.sync()
nt=(n+1023)//1024
buf=np.zeros(nt*8,dtype=np.int64)
L=_lib.lib()
L.pfx_loc_debug_read.argtypes=[C.c_void_p,C.c_void_p,C.c_int64]
L.pfx_loc_debug_read(pf._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
b=buf.reshape(nt,8)
print('phase medians (cycles @100MHz? s_memtime ticks):')
for q in range(7): print(q, np.median(b[:,q]), np.percentile(b[:,q],10), np.percentile(b[:,q],90))
t0=b[:,7]; end=t0+b[:,6]
print('kernel span ticks', end.max()-t0.min(), 'mean WG lifetime', b[:,6].mean(), 'concurrency', b[:,6].sum()/(end.max()-t0.min()))
