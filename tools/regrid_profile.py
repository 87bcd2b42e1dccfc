import sys, os, io, contextlib, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oi-sat-gmi_amd"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oisatgmi import synthetic as syn, _hip
from oisatgmi.interpolator import interpolator
g = syn.swath_granule(7007, nscan=1644, npix=60, lat0=-70.0, lat1=70.0, lon_c=20.0, width_deg=24.0)
nz = 35
rng = np.random.default_rng(5)
g.scattering_weights = rng.uniform(0.1, 2.0, size=(nz,) + g.vcd.shape).astype(np.float32)
g.pressure_mid = rng.uniform(50, 1000, size=(nz,) + g.vcd.shape).astype(np.float32)
ctm = syn.regional_ctm_grid(-89.875, 89.875, -179.875, 179.875, 0.25, 0.25)
for it in (4, 3):
    with contextlib.redirect_stdout(io.StringIO()):
        interpolator(it, 0.25, g, ctm, 0.75); interpolator(it, 0.25, g, ctm, 0.75)
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter(); pr.enable(); r = interpolator(it, 0.25, g, ctm, 0.75); pr.disable(); dt = time.perf_counter() - t0
    print("type", it, "seconds", round(dt, 4))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print("\n".join(s.getvalue().splitlines()[6:26]))
