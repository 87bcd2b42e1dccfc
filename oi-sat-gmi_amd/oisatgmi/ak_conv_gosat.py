"""Drop-in for ``oisatgmi/ak_conv_gosat.py`` of the reference: ``ak_conv_gosat(ctm_data, sat_data)``."""
from ._ak_conv import ak_conv


def ak_conv_gosat(ctm_data: list, sat_data: list):
    """GOSAT XCH4: pressure-weighted averaging kernels applied to the model profile (ak_conv_gosat.py:8-146);
    sets ``ctm_xcol`` (ppbv), an all-NaN ``ctm_vcd`` and ``ctm_time_at_sat`` on every granule."""
    return ak_conv(ctm_data, sat_data, "GOSAT")
