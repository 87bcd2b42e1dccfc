"""MI355X-native drop-in for the optimal-interpolation hot path of ahsouri/OI-SAT-GMI.

    from oisatgmi import oisatgmi                                   # the reference's import
    from oisatgmi.optimal_interpolation import OI
    from oisatgmi.averaging import averaging, error_averager
    from oisatgmi.interpolator import interpolator, _upscaler
    from oisatgmi.config import satellite_amf, satellite_opt, satellite_ssmis, ctm_model

Importing the package never touches the GPU; the first compute call loads ``liboisat_hip.so``
(C-ABI in ``include/oisat.h``) and raises ``OisatUnavailable`` if it cannot -- there is no CPU
fallback.
"""
from .driver import oisatgmi  # noqa: F401
from .optimal_interpolation import OI  # noqa: F401

__all__ = ["oisatgmi", "OI"]
__version__ = "0.1.0"
