import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd')
import numpy as np, ctypes as C
from oisatgmi import _hip, synthetic as syn, dense
from oracle import oi_oracle as orc
ny,nx,m,L=36,72,300,800.0
p = syn.point_obs_case(ny,nx,m,1000+m)
cell = dense.regular_grid_cell(p.lat,p.lon,p.obs_lat,p.obs_lon)
y=np.where(p.obs_y<0,0,p.obs_y)
ref = orc.dense_oi(p.lat,p.lon,p.Xa,p.Sa,p.obs_lat,p.obs_lon,cell,y,p.obs_var,L)
sb=np.sqrt(p.Sa.ravel()); po=orc.unit_vectors(p.obs_lat,p.obs_lon)
S=orc.gaussian_corr(po,po,L)*sb[cell][:,None]*sb[cell][None,:]; S[np.diag_indices(m)]+=p.obs_var
print('cond', np.linalg.cond(S))
plan=dense.DenseAnalysis(p.lat,p.lon,max_obs=m,dtype=np.float64)
plan.load_background(p.Xa,p.Sa); plan.load_obs(p.obs_lat,p.obs_lon,cell,y,p.obs_var)
for refine in (0,1,2,3):
    res=plan.run(L,refine=refine,check_pd=True,want_resid=True)
    z=plan.download_z()
    d=ref['d']
    print(refine,'our resid',res,'oracle-resid of our z', np.linalg.norm(d-S@z)/np.linalg.norm(d), 'z err', np.abs(z-ref['z']).max()/np.abs(ref['z']).max())
# residual kernel vs numpy for a random z
ctx=plan.ctx
zz=np.random.default_rng(0).normal(size=m)
zb=ctx.upload(zz); rb=ctx.alloc(m*8)
ctx.check(ctx.lib.oisat_cov_residual(ctx.h, plan.oxyz.ptr, plan.osig.ptr, plan.ovar.ptr, m, dense.decay_constant(L), plan.d.ptr, zb.ptr, rb.ptr))
r=ctx.download(rb.ptr,(m,),np.float64)
print('residual kernel vs numpy:', np.abs(r-(ref['d']-S@zz)).max()/np.abs(S@zz).max())
