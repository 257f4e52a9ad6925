"""Host-side timing of the banded path's calls (diagnostic)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from pygradflow_amd import problems
from pygradflow_amd.newton import DeviceNewton

m = 50000
prob = problems.sparse_ocp(m, seed=0)
n = prob.num_vars
dn = DeviceNewton(prob, "Full", np.zeros(n), np.zeros(m), 1.0, 1.0)
acc = {}
def t(name, f):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    acc.setdefault(name, []).append(time.perf_counter() - t0); return r
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for i in range(NS):
    if i % 2 == 0 and i > 0:
        t("advance", dn.advance_outer)
    t("step_async", dn.step_async)
    t("sync", dn.sync)
    t("resnorm", dn.residual_norm)
for k, v in acc.items():
    print(k, "median us", 1e6 * float(np.median(v)), "max", 1e6 * max(v))

v = np.array(acc["step_async"]) * 1e6
for k in range(0, len(v), 10):
    print(k, " ".join(f"{x:7.0f}" for x in v[k:k + 10]))
print("residual", dn.residual_norm())

# bench-like loops: no per-call device sync, wall time over K steps
norm_slot = torch.zeros(1, dtype=torch.float64, device="cuda")
for K in (20, 50, 200, 20, 200):
    dn.set_outer(np.zeros(n), np.zeros(m), 1.0, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(K):
        if i % 2 == 0 and i > 0:
            dn.advance_outer()
        dn.step()
        dn.residual_norm(norm_slot.data_ptr())
        marks.append(time.perf_counter())
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d = np.diff(np.array([t0] + marks)) * 1e6
    print(K, "steps:", 1e3 * el / K, "ms/step; per-step us (every 10th):", " ".join(f"{x:.0f}" for x in d[::10]))
