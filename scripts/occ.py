import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
h = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "janusx_amd", "libjxgpu.so"))
out = (ctypes.c_int * 10)()
h.jxg_debug_occupancy(out)
print("occ128", out[0], "occ256", out[1], "regs", out[2], out[3], "lds", out[4], out[5], "ldsPerCU", out[6], "ldsPerBlock", out[7], "regsPerCU", out[8], "regsPerBlock", out[9])
