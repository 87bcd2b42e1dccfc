"""AMF recalculation on the MI355X -- the stage immediately upstream of ``averaging()``.

Drop-in for ``oisatgmi/amf_recal.py`` of the reference (``amf_recal(ctm_data, sat_data)``; SURVEY.md
section 8(f) row 3).  The time matching and record bookkeeping stay on the host; the model partial
columns (:51-56), the per-pixel vertical interpolation + AMF (:93-119, a Python double loop with one
scipy ``interp1d`` per pixel in the reference) and the no-scattering-weight column sum (:160-171) run in
``csrc/amf.hip``.  The optional model upscaling (:58-83) reuses the regridding plan of ``_upscaler``.
"""
from __future__ import annotations

import numpy as np

from . import _hip
from .interpolator import _upscale_plan, _regrid_dtype


def _flatten_time(t):
    return (t.year * 10000 + t.month * 100 + t.day + t.hour / 24.0 + t.minute / 60.0 / 24.0 + t.second / 3600.0 / 24.0)


def _hour_only_time(t):
    return (t.hour / 24.0 + t.minute / 60.0 / 24.0 + t.second / 3600.0 / 24.0)


def _partial_column(ctx, deltap, profile):
    """deltap*profile/g/Mair*N_A*1e-4*1e-15*100*1e-9 in the arrays' own dtype (amf_recal.py:51-56)."""
    dt = _hip.compute_dtype(deltap, profile)
    if np.result_type(deltap, profile) == np.float32:
        dt = np.dtype(np.float32)
    n = int(np.size(deltap))
    buf = ctx.alloc(3 * n * dt.itemsize)
    ctx.upload_into(buf.at(0), np.ravel(deltap), dtype=dt)
    ctx.upload_into(buf.at(n * dt.itemsize), np.ravel(profile), dtype=dt)
    ctx.check(ctx.lib.oisat_partial_column(ctx.h, _hip.dtype_code(dt), buf.at(0), buf.at(n * dt.itemsize), n,
                                           buf.at(2 * n * dt.itemsize)))
    return ctx.download(buf.at(2 * n * dt.itemsize), np.shape(deltap), dt)


def _upscale_cube(ctx, ctm_lon, ctm_lat, cubes, sat_coord, gridsize_ctm, threshold_sat):
    """every level of every cube through ONE regridding plan (the reference calls _upscaler 2*nz times, :76-82)"""
    plan = _upscale_plan(ctm_lon, ctm_lat, sat_coord, gridsize_ctm, threshold_sat)
    if not plan.needed:
        raise ValueError("the satellite grid is not coarser than the model grid: nothing to upscale (amf_recal.py:154)")
    dt = _regrid_dtype()
    stack = np.concatenate([np.asarray(c, dtype=dt) for c in cubes], axis=0)
    nf = stack.shape[0]
    zb = ctx.upload(stack, dtype=dt)
    out = plan.run(zb, nf, dt, False)
    res = ctx.download(out.ptr, (nf,) + tuple(plan.out_shape), dt)
    sizes = np.cumsum([0] + [np.shape(c)[0] for c in cubes])
    return [res[sizes[i]:sizes[i + 1]] for i in range(len(cubes))]


def amf_recal(ctm_data: list, sat_data: list):
    print('AMF Recal begins...')
    ctx = _hip.context()
    time_ctm, time_ctm_hour_only, time_ctm_datetype = [], [], []
    for rec in ctm_data:
        time_ctm.extend([_flatten_time(t) for t in rec.time])
        time_ctm_hour_only.extend([_hour_only_time(t) for t in rec.time])
        time_ctm_datetype.append(rec.time)
    time_ctm = np.array(time_ctm)
    time_ctm_hour_only = np.array(time_ctm_hour_only)
    for L2 in sat_data:
        if L2 is None:
            continue
        t_sat, t_sat_h = _flatten_time(L2.time), _hour_only_time(L2.time)
        if not ctm_data[0].averaged:                                    # amf_recal.py:26-37
            closest = int(np.argmin(np.abs(t_sat - time_ctm)))
            day, hour = int(np.floor(closest / 8.0)), int(closest % 8)
        else:
            closest = int(np.argmin(np.abs(t_sat_h - time_ctm_hour_only)))
            day, hour = 0, closest
        print(f"The closest GMI file used for the L2 at {L2.time} is at {time_ctm_datetype[day][hour]}")
        if ctm_data[0].ctmtype == "FREE":                               # :39-49
            pmid = ctm_data[day].pressure_mid.squeeze()
            prof = ctm_data[day].gas_profile.squeeze()
            delp = ctm_data[day].delta_p.squeeze()
        else:
            pmid = ctm_data[day].pressure_mid[hour].squeeze()
            prof = ctm_data[day].gas_profile[hour].squeeze()
            delp = ctm_data[day].delta_p[hour].squeeze()
        partial = _partial_column(ctx, delp, prof)
        if L2.ctm_upscaled_needed == True:                              # noqa: E712   :154-158
            print("Upscaling of the model is needed.")
            sat_coord = {"Longitude": L2.longitude_center, "Latitude": L2.latitude_center}
            dlon_s = np.abs(sat_coord["Longitude"][0, 0] - sat_coord["Longitude"][0, 1])
            dlat_s = np.abs(sat_coord["Latitude"][0, 0] - sat_coord["Latitude"][1, 0])
            thr_sat = np.sqrt(dlon_s ** 2 + dlat_s ** 2)
            clon, clat = ctm_data[0].longitude, ctm_data[0].latitude
            gs_ctm = np.sqrt(np.abs(clon[0, 0] - clon[0, 1]) ** 2 + np.abs(clat[0, 0] - clat[1, 0]) ** 2)
            pmid, partial = _upscale_cube(ctx, clon, clat, [pmid, partial], sat_coord, gs_ctm, thr_sat)
        nzc = int(np.shape(pmid)[0])
        shape = np.shape(L2.vcd)
        n = int(np.size(L2.vcd))
        has_trop = np.size(L2.tropopause) != 1
        vcd_b = ctx.upload(np.ravel(L2.vcd), dtype=np.float64)
        trop_b = ctx.upload(np.ravel(L2.tropopause), dtype=np.float64) if has_trop else None
        if np.size(L2.scattering_weights) == 1:                         # :160-171
            print('No scattering weights found, recalculation is not possible..just grabbing VCDs')
            dt = np.dtype(np.float32) if np.result_type(pmid, partial) == np.float32 else np.dtype(np.float64)
            cube = ctx.alloc((2 * nzc + 1) * n * dt.itemsize)
            ctx.upload_into(cube.at(0), np.ravel(pmid), dtype=dt)
            ctx.upload_into(cube.at(nzc * n * dt.itemsize), np.ravel(partial), dtype=dt)
            out_ptr = cube.at(2 * nzc * n * dt.itemsize)
            ctx.check(ctx.lib.oisat_column_sum(ctx.h, _hip.dtype_code(dt), cube.at(0), cube.at(nzc * n * dt.itemsize), nzc,
                                               trop_b.ptr if has_trop else None, vcd_b.ptr, n, out_ptr))
            L2.ctm_vcd = ctx.download(out_ptr, shape, dt)
            L2.ctm_time_at_sat = time_ctm[closest]
            L2.old_amf = np.empty((1))
            L2.new_amf = np.empty((1))
            continue
        nzs = int(np.shape(L2.pressure_mid)[0])
        cdt = np.dtype(np.float32) if np.result_type(pmid, partial) == np.float32 else np.dtype(np.float64)
        cube = ctx.alloc((2 * nzs + 2 * nzc + 4) * n * 8 + 256)      # + slack for the 16-byte alignment of each block
        off = 0

        def put(a, dt=np.float64):
            nonlocal off
            off = -(-off // 16) * 16
            ptr = cube.at(off)
            off += ctx.upload_into(ptr, np.ravel(a), dtype=dt)
            return ptr
        p_sat, p_sw = put(L2.pressure_mid), put(L2.scattering_weights)
        p_cp, p_pc = put(pmid, cdt), put(partial, cdt)
        p_amf = put(L2.amf)
        off = -(-off // 16) * 16
        p_new, p_vcd, p_cvcd = cube.at(off), cube.at(off + n * 8), cube.at(off + 2 * n * 8)
        ctx.check(ctx.lib.oisat_amf_recal(ctx.h, p_sat, p_sw, nzs, _hip.dtype_code(cdt), p_cp, p_pc, nzc,
                                          trop_b.ptr if has_trop else None, vcd_b.ptr, p_amf, n, p_new, p_vcd, p_cvcd))
        res = ctx.download(p_new, (3,) + tuple(shape), np.float64)
        L2.old_amf = getattr(L2, 'amf', None)                            # :175
        L2.new_amf = res[0]
        L2.vcd = res[1]
        L2.ctm_vcd = res[2]
        L2.ctm_time_at_sat = time_ctm[closest]
    return sat_data
