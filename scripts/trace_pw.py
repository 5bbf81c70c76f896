import ctypes as C, sys, os, runpy
import numpy as np
sys.argv = ["kbench.py", "bf16", "pw_bn_bwd"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kbench.py"), run_name="__main__")
from isa_amd import lib as L
lib = L.lib()
buf = (C.c_longlong * 1024)()
lib.isa_debug_trace2.argtypes = [C.c_void_p, C.c_int]
print("rc", lib.isa_debug_trace2(buf, 1024))
a = np.array(buf[:], dtype=np.int64)
t0 = a[0]
names = ["entry", "cst+sync", "wb", "fetch0", "stash(last)", "loop done", "sync", "acc fold", "slab", "exit"]
for w in range(4):
    print("wave", w, " ".join("%s=%d" % (names[i], a[16 * w + i] - t0) for i in range(10) if a[16 * w + i] != 0))
