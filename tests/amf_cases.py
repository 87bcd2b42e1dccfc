"""Seeded AMF-recalculation cases; identical to the builders in tests/golden/make_golden.py."""
import numpy as np
from oisatgmi import synthetic as syn


def amf_cases():
    def case_a():
        ctm = syn.ctm_days(10, 14, 12, 2, 9101, averaged=False)
        return ctm, syn.amf_granules(ctm, 9, 3, 9102, with_sw=True, with_trop=True)

    def case_b():
        ctm = syn.ctm_days(8, 9, 20, 1, 9201, averaged=True, dtype=np.float64)
        return ctm, syn.amf_granules(ctm, 35, 2, 9202, with_sw=True, with_trop=False)

    def case_c():
        ctm = syn.ctm_days(10, 14, 12, 2, 9301, averaged=False)
        return ctm, syn.amf_granules(ctm, 9, 2, 9302, with_sw=False, with_trop=True)

    def case_d():
        ctm = syn.ctm_days(49, 65, 6, 1, 9401, averaged=True, lat0=-12.0, lat1=12.0, lon0=-16.0, lon1=16.0)
        coarse = syn.ctm_days(9, 11, 6, 1, 9402, averaged=True, lat0=-10.0, lat1=10.0, lon0=-12.5, lon1=12.5)
        sat = syn.amf_granules(coarse, 7, 2, 9403, with_sw=True, with_trop=True)
        for s in sat:
            if s is not None:
                s.ctm_upscaled_needed = True
        return ctm, sat
    return {"a": case_a, "b": case_b, "c": case_c, "d": case_d}


def akconv_cases():
    """name -> (sensor, ctm_data, sat_data) for the averaging-kernel convolution (ak_conv_mopitt.py / ak_conv_gosat.py)"""
    def upscaled(ctm_seed, coarse_seed, sat_seed, nz, nzs, sensor):
        ctm = syn.ctm_monthly(49, 65, nz, 1, ctm_seed, ctmtype="FREE", lat0=-12.0, lat1=12.0, lon0=-16.0, lon1=16.0)
        coarse = syn.ctm_monthly(9, 11, nz, 1, coarse_seed, ctmtype="FREE", lat0=-10.0, lat1=10.0, lon0=-12.5, lon1=12.5)
        sat = syn.opt_granules(coarse, nzs, 2, sat_seed, sensor=sensor)
        for s in sat:
            if s is not None:
                s.ctm_upscaled_needed = True
        return ctm, sat

    def m_eccoh():
        ctm = syn.ctm_monthly(10, 14, 15, 2, 9501, ctmtype="ECCOH")
        return "MOPITT", ctm, syn.opt_granules(ctm, 9, 3, 9502, sensor="MOPITT")

    def first_of_month(sat):
        # the reference indexes the RECORD list with the index of the closest TIME SLOT (ak_conv_mopitt.py:47-49,:68):
        # with 8 slots in one record that only stays in range when slot 0 is the closest, i.e. on the 1st at 00:00
        import datetime
        for s in sat:
            if s is not None:
                s.time = datetime.datetime(2019, 5, 1, 11, 15)
        return sat

    def m_gmi64():
        ctm = syn.ctm_monthly(8, 9, 12, 1, 9511, ctmtype="GMI", dtype=np.float64)
        return "MOPITT", ctm, first_of_month(syn.opt_granules(ctm, 9, 2, 9512, sensor="MOPITT"))

    def m_up():
        return ("MOPITT",) + upscaled(9521, 9522, 9523, 6, 9, "MOPITT")

    def g_eccoh():
        ctm = syn.ctm_monthly(10, 14, 15, 2, 9601, ctmtype="ECCOH")
        return "GOSAT", ctm, syn.opt_granules(ctm, 20, 3, 9602, sensor="GOSAT")

    def g_gmi64():
        ctm = syn.ctm_monthly(8, 9, 12, 1, 9611, ctmtype="GMI", dtype=np.float64)
        return "GOSAT", ctm, first_of_month(syn.opt_granules(ctm, 20, 2, 9612, sensor="GOSAT"))

    def g_up():
        return ("GOSAT",) + upscaled(9621, 9622, 9623, 6, 20, "GOSAT")
    return {"m_eccoh": m_eccoh, "m_gmi64": m_gmi64, "m_up": m_up, "g_eccoh": g_eccoh, "g_gmi64": g_gmi64, "g_up": g_up}


def pwv_cases():
    """name -> (ctm_data, sat_data) for the SSMIS precipitable-water operator (pwv_cal.py)"""
    def eccoh():
        ctm = syn.ctm_monthly(10, 14, 15, 2, 9701, ctmtype="ECCOH")
        return ctm, syn.ssmis_granules(ctm, 3, 9702)

    def gmi64():
        import datetime
        ctm = syn.ctm_monthly(8, 9, 12, 1, 9711, ctmtype="GMI", dtype=np.float64)
        sat = syn.ssmis_granules(ctm, 2, 9712)
        for s in sat:
            if s is not None:
                s.time = datetime.datetime(2019, 5, 1, 6, 0)      # slot 0 must be the closest: see akconv_cases
        return ctm, sat

    def up():
        ctm = syn.ctm_monthly(49, 65, 6, 1, 9721, ctmtype="FREE", lat0=-12.0, lat1=12.0, lon0=-16.0, lon1=16.0)
        coarse = syn.ctm_monthly(9, 11, 6, 1, 9722, ctmtype="FREE", lat0=-10.0, lat1=10.0, lon0=-12.5, lon1=12.5)
        sat = syn.ssmis_granules(coarse, 2, 9723)
        for s in sat:
            if s is not None:
                s.ctm_upscaled_needed = True
        return ctm, sat
    return {"eccoh": eccoh, "gmi64": gmi64, "up": up}


def check_pwv_against_golden(g, tag, sat, rtol):
    k = 0
    for r in sat:
        if r is None:
            continue
        np.testing.assert_allclose(np.asarray(r.ctm_vcd, dtype=np.float64), g[f"{tag}_{k}_ctm_vcd"], rtol=rtol, atol=0,
                                   equal_nan=True, err_msg=f"{tag} granule {k}")
        k += 1
    assert k == int(g[f"{tag}_n"])


def check_akconv_against_golden(g, tag, sat, rtol):
    k = 0
    for r in sat:
        if r is None:
            continue
        for f in ("ctm_vcd", "ctm_xcol"):
            np.testing.assert_allclose(np.asarray(getattr(r, f), dtype=np.float64), g[f"{tag}_{k}_{f}"], rtol=rtol, atol=0,
                                       equal_nan=True, err_msg=f"{tag} granule {k} {f}")
        assert float(r.ctm_time_at_sat) == float(g[f"{tag}_{k}_time"])
        k += 1
    assert k == int(g[f"{tag}_n"])


def check_against_golden(g, tag, sat, rtol):
    k = 0
    for r in sat:
        if r is None:
            continue
        for f in ("vcd", "ctm_vcd", "new_amf", "old_amf"):
            if np.size(g[f"{tag}_{k}_{f}"]) == 1:        # np.empty((1)) placeholders (amf_recal.py:169-170): uninitialised
                assert np.size(getattr(r, f)) == 1
                continue
            np.testing.assert_allclose(np.asarray(getattr(r, f), dtype=np.float64), g[f"{tag}_{k}_{f}"], rtol=rtol, atol=0,
                                       equal_nan=True, err_msg=f"{tag} granule {k} {f}")
        assert float(r.ctm_time_at_sat) == float(g[f"{tag}_{k}_time"])
        k += 1
    assert k == int(g[f"{tag}_n"])
