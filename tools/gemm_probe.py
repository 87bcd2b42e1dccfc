"""profiling aid: one lower-triangular C -= A A^T update (the shape of the Cholesky's trailing updates) through oisat_gemm_nt.
usage: python tools/gemm_probe.py ROWS_BLOCKS K [reps]     e.g. 80 512 -> 3240 tiles of 128x128 at K = 512 (gemm_nt_kernel)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oisatgmi import _hip
nb, K = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = _hip.context()
M = nb * 128
rng = np.random.default_rng(0)
A = ctx.upload(rng.standard_normal((M, K)).astype(np.float32))
Cb = ctx.upload(np.zeros((M, M), dtype=np.float32))
def run():
    ctx.check(ctx.lib.oisat_gemm_nt(ctx.h, Cb.ptr, M, A.ptr, K, A.ptr, K, M, M, K, 0, 1))
run(); ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    run()
ctx.sync()
el = (time.perf_counter() - t0) / reps
tiles = nb * (nb + 1) // 2
print("tiles %d K %d: %.1f us, %.1f TFLOP/s" % (tiles, K, 1e6 * el, 2.0 * 128 * 128 * K * tiles / el / 1e12))
