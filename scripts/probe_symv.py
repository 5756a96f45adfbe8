"""trd_colB roofline probe alone. Usage: probe_symv.py n"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sclens_amd._lib import Context
n = int(sys.argv[1]); ctx = Context(0)
for rep in range(3):
    l, ms, b = C.c_int64(0), C.c_double(0), C.c_double(0)
    ctx.check(ctx.lib.sclens_hip_symv_probe(ctx.h, n, C.byref(l), C.byref(ms), C.byref(b)))
print(f"n={n} avg_us={ms.value*1e3/l.value:.2f} GB/s={b.value/ms.value/1e6:.1f}")
