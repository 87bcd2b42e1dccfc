import sys, os, time, io, contextlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, synthetic as syn
from oisatgmi.interpolator import interpolator, _plan_cache
g = np.load('/root/repo/tests/golden/interpolator_rbf.npz')
s = syn.swath_granule(5005)
ctx = _hip.context()
for tag, (dlat, dlon) in {"fine": (0.25, 0.25), "coarse": (2.0, 2.5)}.items():
    ctm = syn.regional_ctm_grid(-30.0, 50.0, -25.0, 45.0, dlat, dlon)
    with contextlib.redirect_stdout(io.StringIO()):
        r = interpolator(3, 0.25, s, ctm, 0.75)
    for f in ("vcd", "amf", "uncertainty"):
        w = g[f"{tag}_t3_{f}"]; a = np.asarray(getattr(r, f))
        print(tag, f, "max abs err / scale = %.3e" % (np.nanmax(np.abs(a - w)) / np.nanmax(np.abs(w))))
# big granule
gr = syn.swath_granule(7007, nscan=1644, npix=60, lat0=-70.0, lat1=70.0, lon_c=20.0, width_deg=24.0)
rng = np.random.default_rng(5)
gr.scattering_weights = rng.uniform(0.1, 2.0, size=(35,) + gr.vcd.shape).astype(np.float32)
gr.pressure_mid = rng.uniform(50, 1000, size=(35,) + gr.vcd.shape).astype(np.float32)
ctm = syn.regional_ctm_grid(-89.875, 89.875, -179.875, 179.875, 0.25, 0.25)
for it in (4, 1, 3):
    _plan_cache.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        interpolator(it, 0.25, gr, ctm, 0.75); ctx.sync()
        t0 = time.perf_counter(); interpolator(it, 0.25, gr, ctm, 0.75); ctx.sync()
    print("type", it, "%.3f s per 73-field granule" % (time.perf_counter() - t0))
ctx.prof_reset(); ctx.prof_enable(True)
with contextlib.redirect_stdout(io.StringIO()):
    interpolator(3, 0.25, gr, ctm, 0.75)
for k, v in sorted(ctx.prof_collect().items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
    print("  %-16s %8.3f ms  x%d" % (k, v["total_ms"], v["launches"]))
