"""The ``oisatgmi`` facade -- hot-path methods on the MI355X.

Drop-in for the three methods of ``oisatgmi/driver.py`` that lie on the optimal-interpolation
path: ``average`` (:53-63), ``bias_correct`` (:65-106) and ``oi`` (:108-114), with the same
attribute names set on ``self``.  The I/O methods of the reference's class (``read_data``,
``recal_amf``, ``cal_pwv``, ``conv_ak``, ``reporting``, ``savedaily``, ``write_to_nc``) are file
formats, plotting and sensor-specific operators outside this path:
they raise ``NotImplementedError`` here -- see INTEGRATION.md for binding the HIP path into the
reference's own class instead.
"""
from __future__ import annotations

import numpy as np

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
        '''
            average the data
            Input:
                startdate [str]: starting date in YYYY-mm-dd format string
                enddate [str]: ending date in YYYY-mm-dd format string
        '''
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

    def oi(self, sensor: str, error_ctm=50.0):
        if sensor != 'GOSAT':
            xa, y = self.ctm_averaged_vcd, self.sat_averaged_vcd
        else:
            xa, y = self.aux2, self.aux1
        self.ctm_averaged_vcd_corrected, self.ak_OI, self.increment_OI, self.error_OI = OI(
            xa, y, (xa * error_ctm / 100.0) ** 2, self.sat_averaged_error ** 2, regularization_on=True)

    # ---- outside the hot path ---------------------------------------------------------------
    def _out_of_scope(self, name):
        raise NotImplementedError(
            f"oisatgmi.{name} (file I/O / sensor operators / reporting) is outside the MI355X hot path; "
            f"use the reference implementation for it and bind this package for average/bias_correct/oi "
            f"(INTEGRATION.md)")

    def read_data(self, *a, **k):
        self._out_of_scope("read_data")

    def recal_amf(self, *a, **k):
        self._out_of_scope("recal_amf")

    def cal_pwv(self, *a, **k):
        self._out_of_scope("cal_pwv")

    def conv_ak(self, *a, **k):
        self._out_of_scope("conv_ak")

    def reporting(self, *a, **k):
        self._out_of_scope("reporting")

    def savedaily(self, *a, **k):
        self._out_of_scope("savedaily")

    def write_to_nc(self, *a, **k):
        self._out_of_scope("write_to_nc")
