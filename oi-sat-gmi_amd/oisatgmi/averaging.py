"""Monthly averaging and error propagation on the MI355X.

Drop-in for ``oisatgmi/averaging.py`` of the reference: ``averaging(startdate, enddate,
reader_obj)`` -> ``(sat_vcd, sat_err, ctm_vcd, aux1, aux2, avg_datetime)`` and
``error_averager(error_X)``.  Granule selection, ``None`` handling and the record-type dispatch
stay on the host (a few hundred Python objects); every granule field is uploaded straight into
one contiguous (k, ny, nx) device stack -- no host-side ``np.array(list)`` copy -- and reduced by
``oisat_nanmean_stack`` / ``oisat_error_average`` (csrc/averaging.hip).
"""
from __future__ import annotations

import datetime

import numpy as np

from . import _hip
from .config import satellite_amf, satellite_opt


def _daterange(start_date, end_date):
    for n in range(int((end_date - start_date).days)):
        yield start_date + datetime.timedelta(n)


def _reduce_stack(ctx, fields, dt, kind, flag):
    """fields: list of k equally-shaped arrays -> one reduced field (NumPy array)."""
    k = len(fields)
    shape = np.shape(fields[0])
    n = int(np.prod(shape))
    item = dt.itemsize
    row = n                                               # granule-major, contiguous (odd n: scalar path)
    stack = ctx.alloc((k * row + row) * item)
    for g, f in enumerate(fields):
        a = np.asarray(f)
        if a.shape != shape:
            raise ValueError(f"granule {g} has shape {a.shape}, expected {shape}")
        ctx.upload_into(stack.at(g * row * item), a, dtype=dt)
    out_ptr = stack.at(k * row * item)
    code = _hip.dtype_code(dt)
    if kind == "mean":
        ctx.check(ctx.lib.oisat_nanmean_stack(ctx.h, code, stack.ptr, k, n, 1 if flag else 0, out_ptr))
    else:
        ctx.check(ctx.lib.oisat_error_average(ctx.h, code, stack.ptr, k, n, 1 if flag else 0, out_ptr))
    out = ctx.download(out_ptr, shape, dt)
    stack.free()
    return out


def error_averager(error_X: np.ndarray):
    """sqrt(sum of valid variances)/count per cell over axis 0 (averaging.py:11-24).
    ``error_X`` holds variances (already squared), shape (k, ny, nx)."""
    error_X = np.asarray(error_X)
    ctx = _hip.context()
    dt = _hip.compute_dtype(error_X)
    return _reduce_stack(ctx, list(error_X), dt, "err", False)


def averaging(startdate: str, enddate: str, reader_obj):
    """Drop-in for ``averaging`` (averaging.py:26-120): the per-granule satellite and model columns of ``reader_obj`` between
    the two dates (``'YYYY-mm-dd'`` strings, the end exclusive) reduced to monthly means on the model grid: ``nanmean`` of the
    columns and auxiliaries, ``error_averager`` of the errors.  The stacks are reduced on the device (csrc/averaging.hip)."""
    ctx = _hip.context()
    start_date = datetime.date(int(startdate[0:4]), int(startdate[5:7]), int(startdate[8:10]))
    end_date = datetime.date(int(enddate[0:4]), int(enddate[5:7]), int(enddate[8:10]))
    months = np.array([d.month for d in _daterange(start_date, end_date)])
    years = np.array([d.year for d in _daterange(start_date, end_date)])
    m0, m1 = int(np.min(months)), int(np.max(months))
    y0, y1 = int(np.min(years)), int(np.max(years))

    first = next(g for g in reader_obj.sat_data if g is not None)
    ny, nx = np.shape(first.latitude_center)[0], np.shape(first.latitude_center)[1]
    nm, nyr = m1 - m0 + 1, y1 - y0 + 1
    sat_averaged_vcd = np.zeros((ny, nx, nm, nyr))        # zeros, not NaN (averaging.py:53-58)
    sat_averaged_error = np.full((ny, nx, nm, nyr), np.nan)
    ctm_averaged_vcd = np.full((ny, nx, nm, nyr), np.nan)
    sat_aux1 = np.full((ny, nx, nm, nyr), np.nan)
    sat_aux2 = np.full((ny, nx, nm, nyr), np.nan)

    time_chosen = []
    for year in range(y0, y1 + 1):
        chosen = {"vcd": [], "err": [], "ctm": [], "a1": [], "a2": []}
        month = m0
        for month in range(m0, m1 + 1):
            # the reference rebuilds its lists per month and reduces AFTER the month loop
            # (averaging.py:66-108): only the last month of a multi-month window is averaged
            chosen = {"vcd": [], "err": [], "ctm": [], "a1": [], "a2": []}
            time_chosen = []
            for g in reader_obj.sat_data:
                if g is None:
                    continue
                if g.time.year == year and g.time.month == month:
                    time_chosen.append(g.time)
                    chosen["vcd"].append(g.vcd)
                    chosen["err"].append(g.uncertainty)
                    chosen["ctm"].append(g.ctm_vcd)
                    if isinstance(g, satellite_amf):
                        chosen["a1"].append(g.new_amf)
                        chosen["a2"].append(g.old_amf)
                    elif isinstance(g, satellite_opt):
                        chosen["a1"].append(g.x_col)
                        chosen["a2"].append(g.ctm_xcol)
                    else:
                        chosen["a1"].append(np.nan * g.vcd)
                        chosen["a2"].append(np.nan * g.vcd)
        mi, yi = month - m0, year - y0
        if len(chosen["vcd"]) != 0 and np.size(chosen["vcd"][0]) != 0:
            # every granule counts, as np.array(list) would promote: one float64 granule makes the stack float64
            dt = _hip.compute_dtype(*chosen["vcd"], *chosen["err"], *chosen["ctm"])
            sat_averaged_vcd[:, :, mi, yi] = _reduce_stack(ctx, chosen["vcd"], dt, "mean", True)
            sat_averaged_error[:, :, mi, yi] = _reduce_stack(ctx, chosen["err"], dt, "err", True)
            ctm_averaged_vcd[:, :, mi, yi] = _reduce_stack(ctx, chosen["ctm"], dt, "mean", False)
        if len(chosen["a1"]) != 0 and np.size(chosen["a1"][0]) != 0:
            dt = _hip.compute_dtype(*chosen["a1"], *chosen["a2"])
            sat_aux1[:, :, mi, yi] = _reduce_stack(ctx, chosen["a1"], dt, "mean", False)
            sat_aux2[:, :, mi, yi] = _reduce_stack(ctx, chosen["a2"], dt, "mean", False)

    sat_averaged_vcd = sat_averaged_vcd.squeeze()
    sat_averaged_error = sat_averaged_error.squeeze()
    ctm_averaged_vcd = ctm_averaged_vcd.squeeze()
    sat_aux1 = sat_aux1.squeeze()
    sat_aux2 = sat_aux2.squeeze()
    timestamps = [t.timestamp() for t in time_chosen]
    avg_datetime = datetime.datetime.fromtimestamp(sum(timestamps) / len(timestamps))
    print(avg_datetime)
    return sat_averaged_vcd, sat_averaged_error, ctm_averaged_vcd, sat_aux1, sat_aux2, avg_datetime
