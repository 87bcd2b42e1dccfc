"""The ``oisatgmi`` facade -- hot-path methods on the MI355X.

Drop-in for the methods of ``oisatgmi/driver.py`` that lie on the optimal-interpolation path:
``average`` (:53-63), ``bias_correct`` (:65-106) and ``oi`` (:108-114), the two vertical operators
feeding them -- ``recal_amf`` (:35-38), ``cal_pwv`` (:42-44) and ``conv_ak`` (:46-51) -- and the output stage ``write_to_nc``
(:156-227; same variables, the scaling-factor rule evaluated on the device), with the same attribute
names set on ``self``; ``savedaily`` (:135-155) is plain file writing and is kept as is.  The remaining methods
of the reference's class (``read_data``, ``reporting``) are file formats and plotting outside this path: they raise ``NotImplementedError`` here -- see INTEGRATION.md
for binding the HIP path into the reference's own class instead.
"""
from __future__ import annotations

import numpy as np

import os

from . import _hip
from .averaging import averaging
from .optimal_interpolation import OI

#: (sensor, gas) -> (offset, slope) of the validated linear bias corrections, driver.py:68-100
BIAS_CORRECTIONS = {
    ("TROPOMI", "NO2"): (0.32, 0.66),
    ("TROPOMI", "HCHO"): (0.90, 0.59),
    ("OMI", "NO2"): (0.32, 0.63),
    ("OMI", "HCHO"): (0.821, 0.79),
}

#: DU <-> 1e15 molec/cm2 factor applied to the model O3 column, driver.py:62-63
O3_DIVISOR = 2.69e16 * 1e-15


class oisatgmi(object):

    def __init__(self) -> None:
        pass

    # ---- hot path ---------------------------------------------------------------------------
    def average(self, startdate: str, enddate: str, gasname=None):
        """Monthly means of what the reader holds between the two ``'YYYY-mm-dd'`` dates (driver.py:53-70); ``gasname='O3'``
        additionally converts the model column to Dobson units."""
        (self.sat_averaged_vcd, self.sat_averaged_error, self.ctm_averaged_vcd, self.aux1, self.aux2,
         self.avg_time) = averaging(startdate, enddate, self.reader_obj)
        if gasname == 'O3':
            self.ctm_averaged_vcd = self.ctm_averaged_vcd / O3_DIVISOR

    def bias_correct(self, sat_type, gasname):
        # apply bias correction based on several validation studies
        corr = BIAS_CORRECTIONS.get((sat_type, gasname))
        if corr is None:
            print("NOT applying the bias correction for satellite VCDs")
            return
        print("applying the bias correction for " + str(sat_type) + " " + str(gasname))
        offset, slope = corr
        self.sat_averaged_vcd = (self.sat_averaged_vcd - offset) / slope

    # ---- analysis mode (extension; the signature of oi() is the reference's) -----------------------------
    #   mode        attribute `oi_mode`        | env OISAT_OI_MODE        diag (default) | dense | tiled
    #   L           attribute `corr_length_km` | env OISAT_CORR_LENGTH_KM km, default 300           (dense, tiled)
    #   tile size   attribute `tile_deg`       | env OISAT_TILE_DEG       degrees, default 30       (tiled; halo = 3 L)
    #   unobserved  attribute `oi_unobserved`  | env OISAT_UNOBSERVED     nan (default) | xa        (dense, tiled)
    #   grid        attributes `grid_lat`, `grid_lon` (ny, nx), else the first granule's latitude_center/longitude_center
    #   knee        attribute `oi_reg_index`: force the index into the 99-scaling sweep (all modes)
    #   error/AK    attribute `oi_want_error`  | env OISAT_OI_ERROR       1 (default) | 0: error_OI / ak_OI left NaN -- the
    #               posterior error costs n m^2 flop, 1e16 for a global 720x1440 / 1e5-observation analysis (dense, tiled)
    def _oi_setting(self, attr, env, default, cast=str):
        v = getattr(self, attr, None)
        if v is None:
            v = os.environ.get(env)
        return cast(default if v in (None, "") else v)

    def _oi_grid(self):
        lat, lon = getattr(self, "grid_lat", None), getattr(self, "grid_lon", None)
        if lat is None or lon is None:
            first = next(g for g in self.reader_obj.sat_data if g is not None)
            lat, lon = first.latitude_center, first.longitude_center
        return np.asarray(lat, dtype=np.float64), np.asarray(lon, dtype=np.float64)

    def oi(self, sensor: str, error_ctm=50.0):
        """driver.py:108-114.  Default (``diag``): the reference's element-wise analysis.  ``dense`` / ``tiled``: the
        Gaussian-B generalisation x_a = x_b + B H^T (H B H^T + R)^-1 (y - H x_b) with B = s D^1/2 C D^1/2 -- global, or
        localised to tiles with a 3 L halo; D = diag(Sa) and R = diag(So) are the reference's variances, s is the
        regularisation factor picked by the reference's own knee sweep over the element-wise curve
        (optimal_interpolation.py:15-41), and L -> 0 reproduces the ``diag`` attributes at observed cells.  The same
        four attributes are filled.  Unobserved cells: ``nan`` (default) keeps the reference's convention -- NaN in all
        four fields, optimal_interpolation.py:49-50 with Y = NaN; ``xa`` keeps what the dense analysis really says there:
        background + spread increment, posterior error, averaging kernel 0."""
        if sensor != 'GOSAT':
            xa, y = self.ctm_averaged_vcd, self.sat_averaged_vcd
        else:
            xa, y = self.aux2, self.aux1
        Sa, So = (xa * error_ctm / 100.0) ** 2, self.sat_averaged_error ** 2
        mode = self._oi_setting("oi_mode", "OISAT_OI_MODE", "diag").lower()
        reg_index = getattr(self, "oi_reg_index", None)
        if mode == "diag":
            self.ctm_averaged_vcd_corrected, self.ak_OI, self.increment_OI, self.error_OI = OI(
                xa, y, Sa, So, regularization_on=True, reg_index=reg_index)
            return
        if mode not in ("dense", "tiled"):
            raise ValueError(f"OISAT_OI_MODE / oi_mode must be diag, dense or tiled, not {mode!r}")
        L = self._oi_setting("corr_length_km", "OISAT_CORR_LENGTH_KM", 300.0, float)
        unobserved = self._oi_setting("oi_unobserved", "OISAT_UNOBSERVED", "nan").lower()
        if unobserved not in ("nan", "xa"):
            raise ValueError(f"OISAT_UNOBSERVED / oi_unobserved must be nan or xa, not {unobserved!r}")
        want_error = self._oi_setting("oi_want_error", "OISAT_OI_ERROR", "1").lower() not in ("0", "false", "no", "off")
        lat, lon = self._oi_grid()
        xa, y = np.asarray(xa), np.asarray(y)
        if lat.shape != xa.shape[:2] or lon.shape != xa.shape[:2]:
            raise ValueError(f"oi(): the grid is {lat.shape} but the averaged fields are {xa.shape}; set grid_lat / grid_lon "
                             f"(ny, nx) to the model grid the fields live on")
        print('Optimal interpolation begins...')
        y[y < 0] = 0.0                                           # in place, optimal_interpolation.py:14
        tile = self._oi_setting("tile_deg", "OISAT_TILE_DEG", 30.0, float)
        # averaging() hands back (ny, nx) for the one-month windows run/job.py:77-82 passes and (ny, nx, months[, years])
        # for longer ones (averaging.py:53-58,:110-114: squeezed 4-D buffers).  The element-wise OI does not care; a
        # spatial analysis is one analysis per (month, year) slice.
        outs = [np.full(xa.shape, np.nan) for _ in range(4)]
        infos = []
        for tail in np.ndindex(*xa.shape[2:]):
            sl = (slice(None), slice(None)) + tail
            res, info = self._oi_spatial(mode, xa[sl], y[sl], np.asarray(Sa)[sl], np.asarray(So)[sl], lat, lon, L, tile,
                                         unobserved, reg_index, want_error)
            for o, r in zip(outs, res):
                o[sl] = r
            infos.append(info)
        self.ctm_averaged_vcd_corrected, self.ak_OI, self.increment_OI, self.error_OI = outs
        self.oi_info = dict(infos[0]) if len(infos) == 1 else {**infos[0], "slices": infos}

    @staticmethod
    def _oi_spatial(mode, xa, y, Sa, So, lat, lon, L, tile, unobserved, reg_index, want_error):
        """One (ny, nx) Gaussian-B analysis -> ((Xb, AK, increment, error), info) in the reference's attribute order."""
        from . import dense
        from . import optimal_interpolation as oi_mod
        index, scale, curve, found = oi_mod.regularization_pick(Sa, So, reg_index)
        print("The regularization factor is " + str(scale))
        if mode == "dense":
            xb, inc, info = dense.OI_dense(xa, y, Sa, So, lat, lon, L, scale=scale, want_error=want_error)
        else:
            xb, inc, info = dense.OI_tiled(xa, y, Sa, So, lat, lon, L, tile_deg=tile, scale=scale, want_error=want_error)
        xb, inc = np.array(xb, dtype=np.float64), np.array(inc, dtype=np.float64)
        if want_error:
            err, ak = np.array(info["err"], dtype=np.float64), np.array(info["ak"], dtype=np.float64)
        else:        # posterior error and averaging kernel cost n m^2 flop (1e16 at 720x1440 / 1e5 obs): skipped on request
            err, ak = np.full(xb.shape, np.nan), np.full(xb.shape, np.nan)
        # cells the reference's OI gives numbers for: y, So, xa, Sa all non-NaN.  So = +inf there means K = 0 (no weight):
        # the dense analysis simply does not use such an observation, and AK = 0 as in the reference; Sa*reg = 0 makes the
        # reference's AK = 1 - Sb/(Sa*reg) a 0/0 (optimal_interpolation.py:31): mirrored.
        observed = np.isfinite(y) & ~np.isnan(So) & np.isfinite(xa) & np.isfinite(Sa)
        if want_error:
            ak[observed & np.isinf(So)] = 0.0
            ak[observed & (Sa * scale == 0)] = np.nan
        if unobserved == "nan":
            for a in (xb, inc, err, ak):
                a[~observed] = np.nan
        else:
            if want_error:
                ak[~observed & np.isfinite(xa)] = 0.0
            for a in (inc, err, ak):
                a[~np.isfinite(xa)] = np.nan
        return (xb, ak, inc, err), {"mode": mode, "corr_length_km": L, "scale": scale, "reg_index": index, "knee_found": found,
                                    "nobs": info["nobs"], "unobserved": unobserved, "want_error": want_error}

    def scaling_factor(self):
        """posterior / prior model column with NaN, inf and 0 mapped to 1.0 -- the field the downstream
        emission tools consume (write_to_nc, driver.py:204-206); evaluated on the device."""
        post = np.ascontiguousarray(self.ctm_averaged_vcd_corrected)
        prior = np.ascontiguousarray(self.ctm_averaged_vcd)
        ctx = _hip.context()
        dt = _hip.compute_dtype(post, prior)
        n = int(post.size)
        buf = ctx.alloc(3 * n * dt.itemsize)
        ctx.upload_into(buf.at(0), post, dtype=dt)
        ctx.upload_into(buf.at(n * dt.itemsize), prior, dtype=dt)
        ctx.check(ctx.lib.oisat_scaling_factor(ctx.h, _hip.dtype_code(dt), buf.at(0), buf.at(n * dt.itemsize), n,
                                               buf.at(2 * n * dt.itemsize)))
        return ctx.download(buf.at(2 * n * dt.itemsize), post.shape, dt)

    def output_fields(self):
        """The variables of the reference's output file, in its order and as float32 (driver.py:179-225)."""
        first = next(g for g in self.reader_obj.sat_data if g is not None)
        f32 = lambda a: np.asarray(a, dtype=np.float32)       # noqa: E731
        return {
            "sat_averaged_vcd": f32(self.sat_averaged_vcd), "ctm_averaged_vcd_prior": f32(self.ctm_averaged_vcd),
            "ctm_averaged_vcd_posterior": f32(self.ctm_averaged_vcd_corrected), "sat_averaged_error": f32(self.sat_averaged_error),
            "ak_OI": f32(self.ak_OI), "error_OI": f32(self.error_OI), "scaling_factor": f32(self.scaling_factor()),
            "lon": f32(first.longitude_center), "lat": f32(first.latitude_center), "aux1": f32(self.aux1), "aux2": f32(self.aux2),
        }

    def write_to_nc(self, output_file, output_folder='diag'):
        '''
        Write the final results to a netcdf (same variable names, types and dimensions as the reference,
        driver.py:156-227).  Uses netCDF4 when it is installed; otherwise SciPy's NetCDF-3 writer.
        ``output_file``: file name without the extension; ``output_folder`` is created when missing.
        '''
        if not os.path.exists(output_folder):
            os.makedirs(output_folder)
        fields = self.output_fields()
        path = output_folder + '/' + output_file + '.nc'
        time_string = self.avg_time.strftime("%Y-%m-%d %H:%M:%S")
        nx_, ny_ = np.shape(self.sat_averaged_vcd)[0], np.shape(self.sat_averaged_vcd)[1]
        try:
            from netCDF4 import Dataset
        except ImportError:
            Dataset = None
        if Dataset is not None:
            nc = Dataset(path, 'w')
            nc.createDimension('x', nx_)
            nc.createDimension('y', ny_)
            nc.createDimension('t', None)
            tv = nc.createVariable('time', 'S1', ('t'))
            tv[:] = np.array(list(time_string), 'S1')
            for name, arr in fields.items():
                v = nc.createVariable(name, 'f4', ('x', 'y'))
                v[:, :] = arr
            nc.close()
        else:
            from scipy.io import netcdf_file
            nc = netcdf_file(path, 'w')
            nc.createDimension('x', nx_)
            nc.createDimension('y', ny_)
            nc.createDimension('t', len(time_string))       # NetCDF-3 via SciPy: fixed length instead of unlimited
            tv = nc.createVariable('time', 'c', ('t',))
            tv[:] = np.array(list(time_string), 'S1')
            for name, arr in fields.items():
                v = nc.createVariable(name, 'f', ('x', 'y'))
                v[:, :] = arr
            nc.close()
        return path

    # ---- outside the hot path ---------------------------------------------------------------
    def _out_of_scope(self, name):
        raise NotImplementedError(
            f"oisatgmi.{name} (file I/O / sensor operators / reporting) is outside the MI355X hot path; "
            f"use the reference implementation for it and bind this package for average/bias_correct/oi "
            f"(INTEGRATION.md)")

    def read_data(self, *a, **k):
        self._out_of_scope("read_data")

    def recal_amf(self):
        from .amf_recal import amf_recal
        self.reader_obj.sat_data = amf_recal(self.reader_obj.ctm_data, self.reader_obj.sat_data)

    def cal_pwv(self):
        """driver.py:42-44 of the reference."""
        from .pwv_cal import pwv_calculator
        self.reader_obj.sat_data = pwv_calculator(self.reader_obj.ctm_data, self.reader_obj.sat_data)

    def conv_ak(self, sensor: str):
        """driver.py:46-51 of the reference."""
        if sensor == 'MOPITT':
            from .ak_conv_mopitt import ak_conv_mopitt
            self.reader_obj.sat_data = ak_conv_mopitt(self.reader_obj.ctm_data, self.reader_obj.sat_data)
        if sensor == 'GOSAT':
            from .ak_conv_gosat import ak_conv_gosat
            self.reader_obj.sat_data = ak_conv_gosat(self.reader_obj.ctm_data, self.reader_obj.sat_data)

    def reporting(self, *a, **k):
        self._out_of_scope("reporting")

    def savedaily(self, folder, gasname, date):
        """Per-granule .mat dumps (driver.py:135-155 of the reference): host-side file writing only, same file names
        and variable names; kept so that run/job.py's ``save_daily`` switch works with this facade."""
        from scipy.io import savemat
        if not os.path.exists(folder):
            os.makedirs(folder)
        first_valid_idx = next(i for i, sat_data in enumerate(self.reader_obj.sat_data) if sat_data is not None)
        latitude = self.reader_obj.ctm_data[first_valid_idx].latitude
        longitude = self.reader_obj.ctm_data[first_valid_idx].longitude
        for counter, sat in enumerate(self.reader_obj.sat_data):
            if sat is None:
                continue
            time_sat = 10000.0 * sat.time.year + 100.0 * sat.time.month + sat.time.day + sat.time.hour / 24.0
            savemat(folder + "/" + "sat_data_" + gasname + "_" + str(time_sat) + str(counter) + ".mat",
                    {"vcd_sat": sat.vcd, "vcd_ctm": sat.ctm_vcd, "vcd_err": sat.uncertainty, "time_sat": time_sat,
                     "lat": latitude, "lon": longitude})

