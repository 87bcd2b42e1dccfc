import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd')
import numpy as np
from oisatgmi import synthetic as syn
from oisatgmi.interpolator import NNIndex
s = syn.swath_granule(32, nscan=60, npix=20)
lon, lat = s.longitude_center.copy(), s.latitude_center.copy()
tg = np.stack(np.meshgrid(np.arange(-25, 45, 0.25), np.arange(-30, 50, 0.25)), axis=-1)
d, i = NNIndex(lon, lat).query(tg, max_dist=0.5)
print('clean: found', (i >= 0).sum())
lat[5:9, :] = np.nan; lon[5:9, :] = np.nan
d, i = NNIndex(lon, lat).query(tg, max_dist=0.5)
print('nan rows: found', (i >= 0).sum())
lat2, lon2 = s.latitude_center.copy(), s.longitude_center.copy()
lat2[5, 3] = np.nan
d, i = NNIndex(lon2, lat2).query(tg, max_dist=0.5)
print('one nan (lat only): found', (i >= 0).sum())
