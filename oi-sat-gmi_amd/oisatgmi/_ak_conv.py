"""Averaging-kernel convolution on the MI355X -- what ``oisatgmi.conv_ak`` runs for optimal-estimation products.

Shared body of the drop-ins for ``oisatgmi/ak_conv_mopitt.py`` and ``oisatgmi/ak_conv_gosat.py`` of the
reference (the two files are identical up to the per-pixel formula).  Time matching and record bookkeeping stay
on the host; the model columns (:60-77), the optional model upscaling (:79-116, one regridding plan instead of
4*nz ``_upscaler`` calls) and the per-pixel log-pressure interpolation + averaging kernels (:118-138, a Python
double loop with one scipy ``interp1d`` per pixel in the reference) run on the device (``csrc/amf.hip``).
"""
from __future__ import annotations

import numpy as np

from . import _hip
from .amf_recal import _flatten_time, _partial_column, _upscale_cube


def _time_mean(ctx, cube):
    """``np.nanmean(cube, axis=0)`` of a (nt, nz, ny, nx) model cube in its own dtype (ak_conv_mopitt.py:70-75)."""
    cube = np.asarray(cube)
    dt = np.dtype(np.float32) if cube.dtype == np.float32 else np.dtype(np.float64)
    k = int(cube.shape[0])
    n = int(cube[0].size)
    buf = ctx.upload(cube, dtype=dt)
    out = ctx.alloc(n * dt.itemsize)
    ctx.check(ctx.lib.oisat_nanmean_stack(ctx.h, _hip.dtype_code(dt), buf.ptr, k, n, 0, out.ptr))
    return ctx.download(out.ptr, cube.shape[1:], dt).squeeze()


def _air_column(ctx, deltap):
    """deltap/g/Mair*N_A*1e-4*1e-15*100 in the array's own dtype (ak_conv_mopitt.py:66)."""
    dt = np.dtype(np.float32) if np.asarray(deltap).dtype == np.float32 else np.dtype(np.float64)
    n = int(np.size(deltap))
    buf = ctx.alloc(2 * n * dt.itemsize)
    ctx.upload_into(buf.at(0), np.ravel(deltap), dtype=dt)
    ctx.check(ctx.lib.oisat_partial_column(ctx.h, _hip.dtype_code(dt), buf.at(0), None, n, buf.at(n * dt.itemsize)))
    return ctx.download(buf.at(n * dt.itemsize), np.shape(deltap), dt)


def ak_conv(ctm_data: list, sat_data: list, sensor: str):
    print('Averaging Kernel Conv begins...')
    ctx = _hip.context()
    time_ctm = np.array([_flatten_time(t) for rec in ctm_data for t in rec.time])
    time_ctm_datetype = [rec.time for rec in ctm_data]
    for L2 in sat_data:
        if L2 is None:
            continue
        t_sat = L2.time.year * 10000 + L2.time.month * 100 + L2.time.day            # day resolution only (:42-45)
        closest = int(np.argmin(np.abs(t_sat - time_ctm))) if not ctm_data[0].averaged else 0
        # the reference uses the time-slot index as the record index (:47-49,:61): same IndexError when it is out of range
        print("The closest GMI file used for the L2 at " + str(L2.time) + " is at " + str(time_ctm_datetype[closest]))
        rec = ctm_data[closest]
        kind = ctm_data[0].ctmtype
        if kind in ("ECCOH", "FREE"):
            pmid, prof, delp = rec.pressure_mid.squeeze(), rec.gas_profile.squeeze(), rec.delta_p.squeeze()
        elif kind == "GMI":
            pmid, prof, delp = _time_mean(ctx, rec.pressure_mid), _time_mean(ctx, rec.gas_profile), _time_mean(ctx, rec.delta_p)
        else:                               # the reference leaves the names unbound for any other model (:60-77)
            raise NameError(f"name 'ctm_mid_pressure' is not defined (ctmtype {kind!r} is not handled by the AK convolution)")
        air = _air_column(ctx, delp)
        if L2.ctm_upscaled_needed == True:                                          # noqa: E712   :79
            sat_coord = {"Longitude": L2.longitude_center, "Latitude": L2.latitude_center}
            dlon_s = np.abs(sat_coord["Longitude"][0, 0] - sat_coord["Longitude"][0, 1])
            dlat_s = np.abs(sat_coord["Latitude"][0, 0] - sat_coord["Latitude"][1, 0])
            thr_sat = np.sqrt(dlon_s ** 2 + dlat_s ** 2)
            clon, clat = ctm_data[0].longitude, ctm_data[0].latitude
            gs_ctm = np.sqrt(np.abs(clon[0, 0] - clon[0, 1]) ** 2 + np.abs(clat[0, 0] - clat[1, 0]) ** 2)
            pmid, prof, air = _upscale_cube(ctx, clon, clat, [pmid, prof, air], sat_coord, gs_ctm, thr_sat)
        nzc = int(np.shape(pmid)[0])
        nzs = int(np.shape(L2.pressure_mid)[0])
        shape = np.shape(L2.vcd)
        n = int(np.size(L2.vcd))
        cdt = np.dtype(np.float32) if np.result_type(pmid, prof, air) == np.float32 else np.dtype(np.float64)
        blocks = [(pmid, cdt), (prof, cdt), (air, cdt), (L2.pressure_mid, np.float64), (L2.averaging_kernels, np.float64),
                  (L2.apriori_profile, np.float64)]
        if sensor == "MOPITT":
            blocks += [(L2.aprior_column, np.float64), (L2.apriori_surface, np.float64), (L2.vcd, np.float64)]
        else:
            blocks += [(L2.pressure_weight, np.float64), (L2.x_col, np.float64)]
        total = sum(int(np.size(a)) * 8 + 16 for a, _ in blocks) + 2 * n * 8 + 64
        cube = ctx.alloc(total)
        off = 0
        ptrs = []
        for a, dt in blocks:
            off = -(-off // 16) * 16
            ptrs.append(cube.at(off))
            off += ctx.upload_into(ptrs[-1], np.ravel(a), dtype=dt)
        off = -(-off // 16) * 16
        p_out = cube.at(off)
        code = _hip.dtype_code(cdt)
        if sensor == "MOPITT":
            if int(np.shape(L2.averaging_kernels)[0]) != nzs + 1:
                raise ValueError("MOPITT averaging kernels must hold one surface row plus one row per profile level")
            ctx.check(ctx.lib.oisat_ak_conv_mopitt(ctx.h, code, ptrs[0], ptrs[1], ptrs[2], nzc, ptrs[3], ptrs[4], ptrs[5], nzs,
                                                   ptrs[6], ptrs[7], ptrs[8], n, p_out, cube.at(off + n * 8)))
            res = ctx.download(p_out, (2,) + tuple(shape), np.float64)
            L2.ctm_vcd, L2.ctm_xcol = res[0], res[1]
        else:
            ctx.check(ctx.lib.oisat_ak_conv_gosat(ctx.h, code, ptrs[0], ptrs[1], nzc, ptrs[3], ptrs[4], ptrs[5], ptrs[6], nzs,
                                                  ptrs[7], n, p_out))
            L2.ctm_vcd = np.zeros_like(L2.vcd) * np.nan            # NaN for GOSAT: only XCH4 is used (:138)
            L2.ctm_xcol = ctx.download(p_out, shape, np.float64)
        L2.ctm_time_at_sat = time_ctm[closest]
        cube.free()
    return sat_data
