import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
g=np.load('tests/golden/oi_72x144.npz')
Xa, Y, Sa, So = g["Xa"].copy(), g["Y"].copy(), g["Sa"].copy(), g["So"].copy()
lat, lon = syn.global_grid(72, 144)
ok = np.isfinite(Y) & np.isfinite(So) & np.isfinite(Xa) & np.isfinite(Sa)
xb, inc, info = dense.OI_dense(Xa, Y.copy(), Sa, So, lat, lon, L_km=1e-3, refine=1, dtype=np.float64)
want = g["off_Xb"].reshape(72, 144)
rel = np.abs(xb-want)/np.abs(want)
bad = np.argwhere(ok & (rel>1e-7))
print(info['residuals'])
for i,j in bad[:10]:
    print(i,j,'xb',xb[i,j],'want',want[i,j],'Xa',Xa[i,j],'Y',Y[i,j],'Sa',Sa[i,j],'So',So[i,j],'inc',inc[i,j], 'wantinc', want[i,j]-Xa[i,j])
