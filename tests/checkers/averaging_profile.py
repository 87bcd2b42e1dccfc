"""averaging() end to end (host records in, host fields out) on a month of 30 granules at 720x1440, against the oracle on this host.
usage (GPU box): python tests/checkers/averaging_profile.py"""
import contextlib, io, os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oisatgmi import synthetic as syn, config as cfg
from oisatgmi.averaging import averaging
from oracle import oi_oracle as orc


class R:
    pass


r = R()
r.sat_data = syn.granule_stack(720, 1440, 30, 11)
with contextlib.redirect_stdout(io.StringIO()):
    averaging("2019-06-01", "2019-07-01", r)
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable(); out = averaging("2019-06-01", "2019-07-01", r); pr.disable(); t1 = time.perf_counter()
    t2 = time.perf_counter(); ref = orc.averaging("2019-06-01", "2019-07-01", r, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt); t3 = time.perf_counter()
print(f"averaging(): HIP path {t1 - t0:.3f} s, oracle (vectorised NumPy) {t3 - t2:.3f} s on this host")
for a, b in zip(out[:5], ref[:5]):
    np.testing.assert_allclose(a, b, rtol=1e-13, equal_nan=True)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8); print("\n".join(s.getvalue().splitlines()[6:18]))
