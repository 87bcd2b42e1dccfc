import sys, time, io, contextlib
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd')
import numpy as np
from oisatgmi import synthetic as syn, dense
from oisatgmi.optimal_interpolation import OI
c = syn.diag_case(720, 1440, 100000, 3001)
for dt in (np.float64, np.float32):
    Xa, Y, Sa, So = (a.astype(dt) for a in (c.Xa, c.Y, c.Sa, c.So))
    with contextlib.redirect_stdout(io.StringIO()):
        OI(Xa, Y.copy(), Sa, So, True)
        t0 = time.perf_counter()
        for _ in range(5): OI(Xa, Y.copy(), Sa, So, True)
        dt_s = (time.perf_counter() - t0) / 5
    print(f"OI() numpy in/out 720x1440 {np.dtype(dt).name}: {dt_s*1e3:.2f} ms  {Xa.size/dt_s/1e6:.0f} Mcell/s")
p = syn.point_obs_case(360, 720, 10000, 4000)
t0 = time.perf_counter()
for _ in range(3):
    dense.OI_dense(p.Xa, None, p.Sa, None, p.lat, p.lon, 500.0, refine=1, dtype=np.float32, obs=dict(lat=p.obs_lat, lon=p.obs_lon, y=p.obs_y, var=p.obs_var))
print(f"OI_dense() numpy in/out config 2: {(time.perf_counter()-t0)/3*1e3:.1f} ms")
