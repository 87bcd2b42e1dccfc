import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
ctx = _hip.context()
p = syn.point_obs_case(360, 720, m, 4000, swaths=False)
cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=ctx)
plan.load_background(p.Xa, p.Sa)
plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
lib, h = ctx.lib, ctx.h
L = {}
for mode in ("0", "1"):
    os.environ["OISAT_DAG"] = mode
    ctx.check(lib.oisat_cov_build(h, plan.oxyz.ptr, plan.osig.ptr, plan.ovar.ptr, plan.m, dense.decay_constant(500.0), plan.S.ptr, plan.mp))
    ctx.check(lib.oisat_potrf(h, plan.S.ptr, plan.m, plan.mp, None))
    ctx.sync()
    L[mode] = plan.download_S().astype(np.float64)
    print(mode, ctx.solve_status(clear=True))
nb = plan.mp // 128
np.set_printoptions(linewidth=250, precision=1)
D = np.zeros((nb, nb))
for i in range(nb):
    for j in range(i + 1):
        a = L["0"][i*128:(i+1)*128, j*128:(j+1)*128]; b = L["1"][i*128:(i+1)*128, j*128:(j+1)*128]
        if i == j: a = np.tril(a); b = np.tril(b)
        D[i, j] = np.nanmax(np.abs(a - b)) if np.isfinite(b).all() else np.inf
print(D[:min(nb, 12), :min(nb, 12)])
a = np.tril(L["0"][:128, :128]); b = np.tril(L["1"][:128, :128])
d = np.abs(a - b).reshape(8, 16, 8, 16).max(axis=(1, 3))
print("tile (0,0) by 16x16 sub-tile:"); print(d)
i, j = np.unravel_index(np.argmax(np.abs(a - b)), a.shape)
print("worst", i, j, a[i, j], b[i, j])
print("row 0..3 col 0..3 rec:\n", a[:4, :4], "\ndag:\n", b[:4, :4])
