import sys, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd'); sys.path.insert(0, 'tests')
import numpy as np
from amf_cases import amf_cases
from oisatgmi.amf_recal import amf_recal
for tag in ("a", "b", "c", "d", "c"):
    ctm, sat = amf_cases()[tag]()
    try:
        amf_recal(ctm, sat)
        print(tag, "ok")
    except Exception:
        print(tag, "FAILED")
        traceback.print_exc()
        break
