import ctypes as C, sys
sys.path.insert(0, ".")
from pygradflow_amd import _lib
lib = _lib.load()
variant, N, KB, reps = (int(v) for v in sys.argv[1:5])
ms, fl = C.c_double(0), C.c_double(0)
assert lib.pgf_bench_update(N, KB, variant, reps, 0, C.byref(ms), C.byref(fl)) == 0
print(f"variant={variant} N={N} KB={KB}: {ms.value*1e3:8.1f} us  {fl.value/ms.value/1e9:6.1f} TFLOP/s")
