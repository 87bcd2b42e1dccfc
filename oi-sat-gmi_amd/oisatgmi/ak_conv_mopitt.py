"""Drop-in for ``oisatgmi/ak_conv_mopitt.py`` of the reference: ``ak_conv_mopitt(ctm_data, sat_data)``."""
from ._ak_conv import ak_conv


def ak_conv_mopitt(ctm_data: list, sat_data: list):
    """MOPITT CO: log10-space averaging kernels applied to the model profile (ak_conv_mopitt.py:8-149);
    sets ``ctm_vcd``, ``ctm_xcol`` (ppmv) and ``ctm_time_at_sat`` on every granule."""
    return ak_conv(ctm_data, sat_data, "MOPITT")
