"""Model precipitable water on the MI355X -- what ``oisatgmi.cal_pwv`` runs for SSMIS.

Drop-in for ``oisatgmi/pwv_cal.py`` of the reference (``pwv_calculator(ctm_data, sat_data)``): the third of the
three operators ``run/job.py:65-72`` chooses between before the monthly averaging (``recal_amf`` / ``conv_ak`` /
``cal_pwv``).  Time matching on the host; water partial columns (:60-70), the optional model upscaling (:72-94, one
regridding plan for all levels) and the column sum with the observation mask (:96-98) on the device.
"""
from __future__ import annotations

import numpy as np

from . import _hip
from ._ak_conv import _time_mean
from .amf_recal import _flatten_time, _upscale_cube


def pwv_calculator(ctm_data: list, sat_data: list):
    print('PWV begins...')
    ctx = _hip.context()
    time_ctm = np.array([_flatten_time(t) for rec in ctm_data for t in rec.time])
    time_ctm_datetype = [rec.time for rec in ctm_data]
    for L2 in sat_data:
        if L2 is None:
            continue
        t_sat = L2.time.year * 10000 + L2.time.month * 100 + L2.time.day            # :41-42
        closest = int(np.argmin(np.abs(t_sat - time_ctm))) if not ctm_data[0].averaged else 0
        print("The closest CTM file used for the L2 at " + str(L2.time) + " is at " + str(time_ctm_datetype[closest]))
        rec = ctm_data[closest]                 # (the reference indexes the records with the time-slot index, :60)
        kind = ctm_data[0].ctmtype
        if kind in ("ECCOH", "FREE"):
            delp, prof = rec.delta_p.squeeze(), rec.gas_profile.squeeze()
        elif kind == "GMI":
            prof, delp = _time_mean(ctx, rec.gas_profile), _time_mean(ctx, rec.delta_p)
        else:
            raise NameError(f"name 'ctm_deltap' is not defined (ctmtype {kind!r} is not handled by the PWV calculator)")
        dt = np.dtype(np.float32) if np.result_type(delp, prof) == np.float32 else np.dtype(np.float64)
        nz = int(np.shape(delp)[0])
        ncube = int(np.size(delp))
        buf = ctx.alloc(3 * ncube * dt.itemsize)
        ctx.upload_into(buf.at(0), np.ravel(delp), dtype=dt)
        ctx.upload_into(buf.at(ncube * dt.itemsize), np.ravel(prof), dtype=dt)
        part_ptr = buf.at(2 * ncube * dt.itemsize)
        ctx.check(ctx.lib.oisat_water_column(ctx.h, _hip.dtype_code(dt), buf.at(0), buf.at(ncube * dt.itemsize), ncube, part_ptr))
        shape = np.shape(L2.vcd)
        n = int(np.size(L2.vcd))
        vcd_b = ctx.upload(np.ravel(L2.vcd), dtype=np.float64)
        if L2.ctm_upscaled_needed == True:                                          # noqa: E712   :72
            partial = ctx.download(part_ptr, np.shape(delp), dt)
            sat_coord = {"Longitude": L2.longitude_center, "Latitude": L2.latitude_center}
            dlon_s = np.abs(sat_coord["Longitude"][0, 0] - sat_coord["Longitude"][0, 1])
            dlat_s = np.abs(sat_coord["Latitude"][0, 0] - sat_coord["Latitude"][1, 0])
            thr_sat = np.sqrt(dlon_s ** 2 + dlat_s ** 2)
            clon, clat = ctm_data[0].longitude, ctm_data[0].latitude
            gs_ctm = np.sqrt(np.abs(clon[0, 0] - clon[0, 1]) ** 2 + np.abs(clat[0, 0] - clat[1, 0]) ** 2)
            (partial,) = _upscale_cube(ctx, clon, clat, [partial], sat_coord, gs_ctm, thr_sat)
            dt = np.dtype(partial.dtype)
            pb = ctx.upload(partial, dtype=dt)
            part_ptr = pb.ptr
        out = ctx.alloc(n * dt.itemsize)
        ctx.check(ctx.lib.oisat_pwv_sum(ctx.h, _hip.dtype_code(dt), part_ptr, nz, vcd_b.ptr, n, out.ptr))
        L2.ctm_vcd = ctx.download(out.ptr, shape, dt)
    return sat_data
