"""Record types that flow through the hot path.

Drop-in for the reference's data model (config.py:6-73 in the reference): four plain record
classes whose **field names and positional order are the contract** -- the reference builds them
positionally everywhere (e.g. interpolator.py:285-290, reader.py:894-895), and ``averaging()``
dispatches on ``isinstance(x, satellite_amf)`` / ``satellite_opt`` (averaging.py:82-90).

The records are generated from the field tables below so the order is stated once, in one place,
and can be checked by the tests against the golden field lists.
"""
from __future__ import annotations

import datetime
from dataclasses import make_dataclass

import numpy as np

_ND = np.ndarray
_DT = datetime.datetime

#: (name, type) in positional order -- 17 fields
SATELLITE_AMF_FIELDS = (
    ("vcd", _ND), ("amf", _ND), ("time", _DT), ("tropopause", _ND),
    ("latitude_center", _ND), ("longitude_center", _ND),
    ("latitude_corner", _ND), ("longitude_corner", _ND),
    ("uncertainty", _ND), ("quality_flag", _ND),
    ("pressure_mid", _ND), ("scattering_weights", _ND),
    ("ctm_upscaled_needed", bool), ("ctm_vcd", _ND), ("ctm_time_at_sat", _DT),
    ("old_amf", _ND), ("new_amf", _ND),
)

#: 23 fields
SATELLITE_OPT_FIELDS = (
    ("vcd", _ND), ("time", _DT), ("profile", _ND), ("tropopause", _ND),
    ("latitude_center", _ND), ("longitude_center", _ND),
    ("latitude_corner", _ND), ("longitude_corner", _ND),
    ("uncertainty", _ND), ("quality_flag", _ND),
    ("pressure_mid", _ND), ("averaging_kernels", _ND),
    ("ctm_upscaled_needed", bool), ("ctm_vcd", _ND), ("ctm_xcol", _ND), ("ctm_time_at_sat", _DT),
    ("aprior_column", _ND), ("apriori_profile", _ND), ("surface_pressure", _ND),
    ("apriori_surface", _ND), ("x_col", _ND), ("pressure_weight", _ND), ("sensor", str),
)

#: 8 fields
SATELLITE_SSMIS_FIELDS = (
    ("vcd", _ND), ("uncertainty", _ND), ("time", _DT),
    ("latitude_center", _ND), ("longitude_center", _ND),
    ("ctm_upscaled_needed", bool), ("ctm_vcd", _ND), ("sensor", str),
)

#: 9 fields ("tempeature_mid" is the reference's own spelling and is part of the interface)
CTM_MODEL_FIELDS = (
    ("latitude", _ND), ("longitude", _ND), ("time", list), ("gas_profile", _ND),
    ("pressure_mid", _ND), ("tempeature_mid", _ND), ("delta_p", _ND),
    ("ctmtype", str), ("averaged", bool),
)

satellite_amf = make_dataclass("satellite_amf", SATELLITE_AMF_FIELDS)
satellite_opt = make_dataclass("satellite_opt", SATELLITE_OPT_FIELDS)
satellite_ssmis = make_dataclass("satellite_ssmis", SATELLITE_SSMIS_FIELDS)
ctm_model = make_dataclass("ctm_model", CTM_MODEL_FIELDS)

for _cls in (satellite_amf, satellite_opt, satellite_ssmis, ctm_model):
    _cls.__module__ = __name__          # picklable across joblib workers like the reference's

__all__ = ["satellite_amf", "satellite_opt", "satellite_ssmis", "ctm_model"]
