"""Where one type-1 (Delaunay linear) granule spends its time: qhull on the host, the triangulation's move to the device
(barycentric transforms are computed there: oisat_tri_transform), point location, the 73-field regrid.
usage (GPU box): python tools/type1_profile.py"""
import contextlib
import io
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np                                             # noqa: E402
from scipy.spatial import Delaunay                             # noqa: E402
from oisatgmi import synthetic as syn, _hip                    # noqa: E402
from oisatgmi import interpolator as itp                       # noqa: E402

g = syn.swath_granule(7007, nscan=1644, npix=60, lat0=-70.0, lat1=70.0, lon_c=20.0, width_deg=24.0)
rng = np.random.default_rng(5)
g.scattering_weights = rng.uniform(0.1, 2.0, size=(35,) + g.vcd.shape).astype(np.float32)
g.pressure_mid = rng.uniform(50, 1000, size=(35,) + g.vcd.shape).astype(np.float32)
ctm = syn.regional_ctm_grid(-89.875, 89.875, -179.875, 179.875, 0.25, 0.25)
ctx = _hip.context()
with contextlib.redirect_stdout(io.StringIO()):
    itp.interpolator(1, 0.25, g, ctm, 0.75)
pts = np.column_stack((np.ravel(g.longitude_center), np.ravel(g.latitude_center))).astype(np.float64)
t0 = time.perf_counter(); tri = Delaunay(pts); t1 = time.perf_counter()
tri.vertex_to_simplex, tri.neighbors
t2 = time.perf_counter(); ti = itp.TriIndex(tri); ctx.sync(); t3 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    ta = time.perf_counter(); itp.interpolator(1, 0.25, g, ctm, 0.75); ctx.sync(); tb = time.perf_counter()
ref = Delaunay(pts)
tc = time.perf_counter(); ref.transform; td = time.perf_counter()
print(f"qhull {t1 - t0:.3f} s; vertex_to_simplex + neighbors {t2 - t1:.3f} s; to the device incl. transforms {t3 - t2:.3f} s "
      f"({ti.ns} simplices); the whole interpolator() call {tb - ta:.3f} s; scipy's Delaunay.transform on this host {td - tc:.3f} s")
