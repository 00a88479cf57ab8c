import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from janusx_amd import janusx as jxrs, bed
from janusx_amd._lib import lib
print("devices", lib().jxg_device_count())
packed, g = bed.synth_panel_numpy(200, 300, seed=1)
try:
    k = jxrs.grm_packed_f32(packed, 200, np.zeros(300, bool), np.full(300, 0.3, np.float32))
    print("grm ok", k.shape, float(k[0, 0]))
except Exception as e:
    print("ERR", e)
