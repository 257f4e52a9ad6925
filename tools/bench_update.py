"""Trailing-update kernel in isolation: TFLOP/s vs region size and K-depth.
Variants (pgf_bench_update): 0 = 64x64 tiles / 256 threads, 3 / 11 = 128x128 / 1024 threads
with K-chunks of 16 / 32; 21.. = experiment toggles of update_tile (EXP, wrong results)."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from pygradflow_amd import _lib

lib = _lib.load()
variants = [int(v) for v in sys.argv[1:]] or [0]
for variant in variants:
    for N, KB in [(9600, 256), (4864, 256), (2560, 256), (4864, 512)]:
        ms, fl = C.c_double(0), C.c_double(0)
        rc = lib.pgf_bench_update(N, KB, variant, 5, 0, C.byref(ms), C.byref(fl))
        assert rc == 0, rc
        print(f"variant={variant} N={N} KB={KB}: {ms.value*1e3:8.1f} us  {fl.value/ms.value/1e9:6.1f} TFLOP/s", flush=True)
