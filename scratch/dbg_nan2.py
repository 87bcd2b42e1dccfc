import sys, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd')
import numpy as np
from oisatgmi import synthetic as syn
from oisatgmi.interpolator import interpolator, _GranuleRegridder
ctm = syn.regional_ctm_grid(-30.0, 50.0, -25.0, 45.0, 2.0, 2.5)
s = syn.swath_granule(32, nscan=60, npix=20)
with contextlib.redirect_stdout(io.StringIO()):
    r0 = interpolator(4, 0.25, s, ctm, 0.75)
print('clean ->', None if r0 is None else np.isfinite(r0.vcd).sum())
s.latitude_center[5:9, :] = np.nan
s.longitude_center[5:9, :] = np.nan
rg = _GranuleRegridder(s, 0.25, ctm, 0.75, 4)
idx = rg.ctx.download(rg.idx_fine.ptr, rg.fine_shape, np.int32)
print('idx_fine found', (idx >= 0).sum(), 'fine shape', rg.fine_shape)
X, Y, Z, need = rg.regrid([s.vcd])
print('finite out', np.isfinite(Z[0]).sum(), need)
pidx = rg.ctx.download(rg.plan.idx.ptr, rg.plan.out_shape, np.int32)
print('plan idx found', (pidx >= 0).sum())
