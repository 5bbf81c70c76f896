"""temporary: phase trace of dw_bn_bwd (block 7, all 8 waves)"""
import ctypes as C, sys, os, runpy
import numpy as np
sys.argv = ["kbench.py", "bf16", "dw_bn_bwd"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kbench.py"), run_name="__main__")
from isa_amd import lib as L
lib = L.lib()
buf = (C.c_longlong * 8192)()
lib.isa_debug_trace.argtypes = [C.c_void_p, C.c_int]
print("rc", lib.isa_debug_trace(buf, 8192))
a = np.array(buf[:], dtype=np.int64)
t0 = a[0]
names = ["entry", "prologue done", "after sync", "first tile staged", "tiles done", "fold done", "after sync", "exit"]
for w in range(8):
    print("wave", w, " ".join("%s=%d" % (names[i], a[64 * w + i] - t0) for i in range(8) if a[64 * w + i] != 0))
