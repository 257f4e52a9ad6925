#!/usr/bin/env python3
"""Batched Newton steps of B instances of (n, m) on one GPU: ms per batched step and
instance-steps/s for several B (what one rank of an 8-GPU run of BASELINE config 4 sees is
B = 32)."""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygradflow_amd import problems  # noqa: E402
from pygradflow_amd.batched import BatchedDeviceNewton  # noqa: E402

n, m = 1024, 256
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]:
    bn = BatchedDeviceNewton(lambda i: problems.dense_qp(n, m, seed=i, boxed_frac=0.1, box=0.05), B,
                             "Full", 1.0, 1.0)
    for _ in range(2):
        bn.step()
    K = 10
    gc.collect()
    gc.disable()  # a collection of the problem generators' garbage stalls a step by tens of ms
    t0 = time.perf_counter()
    for _ in range(K):
        bn.step()
    el = time.perf_counter() - t0
    gc.enable()
    print(f"B={B:4d}: {1e3 * el / K:7.3f} ms per batched step  {B * K / el:9.0f} instance-steps/s", flush=True)
    bn.close()
