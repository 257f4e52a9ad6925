"""Trailing-update kernel in isolation: TFLOP/s vs region size and K-depth."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from pygradflow_amd import _lib

lib = _lib.load()
variants = [int(v) for v in sys.argv[1:]] or [0]
for variant in variants:
    for N, KB in [(4864, 64), (1280, 64), (4864, 256), (2560, 256), (1280, 256)]:
        ms, fl = C.c_double(0), C.c_double(0)
        rc = lib.pgf_bench_update(N, KB, variant, 5, 0, C.byref(ms), C.byref(fl))
        assert rc == 0, rc
        print(f"variant={variant} N={N} KB={KB}: {ms.value*1e3:8.1f} us  {fl.value/ms.value/1e9:6.1f} TFLOP/s", flush=True)
