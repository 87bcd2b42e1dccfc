import os, sys, time, io, contextlib
sys.path[:0] = ["/root/repo", "/root/repo/oi-sat-gmi_amd"]
import numpy as np
from oisatgmi import synthetic as syn, _hip
from oisatgmi import interpolator as itp
g = syn.swath_granule(7007, nscan=1644, npix=60, lat0=-70.0, lat1=70.0, lon_c=20.0, width_deg=24.0)
ctm = syn.regional_ctm_grid(-89.875, 89.875, -179.875, 179.875, 0.25, 0.25)
ctx = _hip.context()
for rep in range(2):
    t0 = time.perf_counter(); tri = itp._triangulate(g.longitude_center, g.latitude_center); t1 = time.perf_counter()
    ti = itp.TriIndex(tri); ctx.sync(); t2 = time.perf_counter()
    rg = itp._GranuleRegridder(g, 0.25, ctm, 0.75, 1, tri); ctx.sync(); t3 = time.perf_counter()
    print("qhull %.3f  TriIndex upload %.3f  regridder init (incl. TriIndex, nn query, locate) %.3f  ambiguous %d degenerate %s" % (t1 - t0, t2 - t1, t3 - t2, rg.tri.ambiguous, rg.tri.has_degenerate))
    t0 = time.perf_counter(); nn = itp.NNIndex(g.longitude_center, g.latitude_center); idx, _ = nn.query_device(rg.lons_grid, rg.lats_grid, rg.cell, resolve_ties=False); ctx.sync(); print("  nn query %.3f" % (time.perf_counter() - t0))
    t0 = time.perf_counter(); f = rg.tri.locate(rg.tgt, rg.Tfine, rg.idx_fine, rg.lons_grid, rg.lats_grid); ctx.sync(); print("  locate %.3f" % (time.perf_counter() - t0))
    fields = [g.vcd] * 73
    t0 = time.perf_counter(); rg.regrid(fields); ctx.sync(); print("  regrid 73 fields %.3f" % (time.perf_counter() - t0))
print("cpus", len(os.sched_getaffinity(0)))
