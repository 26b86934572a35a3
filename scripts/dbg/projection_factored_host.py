"""The projection's cov2d in the textbook order against the 2 x 3 factored order (csrc/mgs_math.h, MGS_PROJ_FACTORED) on the CPU: the
header compiled for the host twice (tests/host_harness), both against the NumPy fp64 oracle on the clustered scene -- relative error of
the conic, and |d sigma| at pixels inside the footprint of the needle-like Gaussians (what the blend sees of the inputs' noise).
    python scripts/dbg/projection_factored_host.py"""
import ctypes, math, os, subprocess, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robosimgs_amd import synthetic_scene_heavy_tailed, camera_ring
from oracle import gs_oracle_np as O
HERE=os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests')
def build(flag, name):
    so=f'/tmp/libhh_{name}.so'
    subprocess.run(["g++","-O2","-std=c++17","-shared","-fPIC",flag,os.path.join(HERE,"host_harness","harness.cpp"),"-o",so],check=True)
    return ctypes.CDLL(so)
_p=lambda a:a.ctypes.data_as(ctypes.c_void_p); _f=lambda a:np.ascontiguousarray(a,dtype=np.float32)
g = synthetic_scene_heavy_tailed(300000, sh_degree=0, seed=0)
w,h=1920,1080
cam = camera_ring(1, w, h, thetas=[0.3])[0]
n=len(g)
def run(hh):
    radii=np.zeros(n,np.int32); m2d,dep,con,comp=(np.zeros((n,2),np.float32),np.zeros(n,np.float32),np.zeros((n,3),np.float32),np.zeros(n,np.float32))
    vm,K=_f(cam.viewmat()),_f(cam.K)
    hh.hh_project(n,_p(_f(g.means)),_p(_f(g.quats)),_p(_f(g.scales)),_p(vm),_p(K),w,h,ctypes.c_float(0.3),ctypes.c_float(0.01),ctypes.c_float(1e10),ctypes.c_float(0.0),_p(radii),_p(m2d),_p(dep),_p(con),_p(comp))
    return radii,m2d,dep,con,comp
import inspect
ref = O.project(g.means.astype(np.float64), g.quats.astype(np.float64), g.scales.astype(np.float64), _f(cam.viewmat()).astype(np.float64), _f(cam.K).astype(np.float64), w, h) if 'project' in dir(O) else None
print(type(ref), [k for k in (ref.keys() if isinstance(ref,dict) else [])][:10] if ref is not None else None)
rc = ref['conics'] if isinstance(ref,dict) else ref[3]
rr = ref['radii'] if isinstance(ref,dict) else ref[0]
for flag,name in (("-DMGS_PROJ_FACTORED=1","fac"),("-DMGS_PROJ_FACTORED=0","std")):
    radii,m2d,dep,con,comp = run(build(flag,name))
    vis=(radii>0)&(np.asarray(rr).reshape(n,-1)[:,0]>0)
    err=np.abs(con.astype(np.float64)-rc).max(1)/np.abs(rc).max(1).clip(1e-30)
    e=err[vis]
    print(name,'visible',vis.sum(),'rel conic err: median %.2e p99 %.2e p99.9 %.2e max %.2e'%(np.median(e),np.quantile(e,.99),np.quantile(e,.999),e.max()), 'radius mismatches', int((radii[vis]!=np.asarray(rr).reshape(n,-1)[vis,0]).sum()))

# sigma at sample offsets: the inputs' noise alone (sigma evaluated in fp64 from each projection's outputs)
refm = ref['means2d']
a_,b_,c_ = None,None,None
Q = rc; tr = Q[:,0]+Q[:,2]; dd = Q[:,0]*Q[:,2]-Q[:,1]**2
l_small = tr/2-np.sqrt(np.maximum(tr*tr/4-dd,0)); l_big = tr/2+np.sqrt(np.maximum(tr*tr/4-dd,0))   # eigenvalues of the CONIC
kap = l_big/np.maximum(l_small,1e-300)
# long axis direction = eigenvector of the conic's small eigenvalue
ang = 0.5*np.arctan2(2*Q[:,1], Q[:,0]-Q[:,2]); 
u_big = np.stack([np.cos(ang), np.sin(ang)],1)         # eigenvector of the larger conic eigenvalue (thin direction)
u_small = np.stack([-np.sin(ang), np.cos(ang)],1)      # long direction
rng = np.random.default_rng(0)
for flag,name in (("-DMGS_PROJ_FACTORED=1","fac"),("-DMGS_PROJ_FACTORED=0","std")):
    radii,m2d,dep,con,comp = run(build(flag,name))
    vis=(radii>0)&(kap>1e3)&(np.asarray(rr).reshape(n,-1)[:,0]>0)
    idx=np.nonzero(vis)[0]
    errs=[]
    for rep in range(8):
        t_long = rng.uniform(-2.5,2.5,len(idx))/np.sqrt(l_small[idx])     # along the needle, up to 2.5 sigma
        t_thin = rng.uniform(-2.5,2.5,len(idx))/np.sqrt(l_big[idx])
        p = refm[idx] + u_small[idx]*t_long[:,None] + u_big[idx]*t_thin[:,None]
        p = np.floor(p)+0.5                                               # a pixel centre
        def sig(m,q):
            d = p - m
            return 0.5*(q[:,0]*d[:,0]**2 + q[:,2]*d[:,1]**2) + q[:,1]*d[:,0]*d[:,1]
        s_ref = sig(refm[idx], Q[idx]); s_new = sig(m2d[idx].astype(np.float64), con[idx].astype(np.float64))
        s_cononly = sig(refm[idx], con[idx].astype(np.float64))
        keep = s_ref < 8
        errs.append(np.stack([np.abs(s_new-s_ref)[keep], np.abs(s_cononly-s_ref)[keep]],1))
    e=np.concatenate(errs)
    print(name, 'needles', len(idx), '|d sigma| (mean2d + conic noise): median %.2e p99 %.2e max %.2e;  conic noise alone: median %.2e p99 %.2e max %.2e' % (np.median(e[:,0]),np.quantile(e[:,0],.99),e[:,0].max(),np.median(e[:,1]),np.quantile(e[:,1],.99),e[:,1].max()))
