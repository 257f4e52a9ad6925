"""Build recipe for libpgf_hip.so (hipcc, gfx950 only; cross-compiles without a GPU)."""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpgf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# (source, extra flags).  The elementwise file must not contract a*b+c into fma:
# active-set masks are compared bit-for-bit with numpy expressions.
SOURCES = [
    ("pgf_kernels.hip", ["-ffp-contract=off"]),
    ("pgf_sparse.hip", ["-ffp-contract=off"]),
    ("pgf_ldlt.hip", ["-DPGF_RECIP_ONE_STEP"] if os.environ.get("PGF_BUILD_RECIP1") else []),
    ("pgf_factor2.hip", []),
    ("pgf_lu.hip", []),
    ("pgf_api.hip", ["-ffp-contract=off"]),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "pgf_hip.h"))
    objs = []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or not _newer(op, [sp] + headers):
            cmd = [HIPCC, *COMMON, *extra, "-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(op)
    if force or not _newer(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
