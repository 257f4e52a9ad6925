"""Batched mode: independent NLP instances sharded over the GPUs of one node.

The reference has no distributed code; its only parallelism is a process pool over
independent instances (``pygradflow/runners/runner.py:107-153``).  Here instance ``i`` of
``B`` lives on rank ``i // ceil(B / world)`` (contiguous blocks), every rank advances its
own instances with the device-resident Newton step, and per batched step there is exactly
ONE collective: an all-gather of the ranks' residual norms ``||F(z_i)||_2`` (fp64, 8 bytes
per instance) so that every rank sees all ``B`` norms for a global accept / stop decision
(SURVEY.md 8e).  RCCL over xGMI when the process group is ``nccl``; the same code runs on
``gloo`` for the CPU tests.
"""

from __future__ import annotations

import numpy as np


def shard_range(num_instances: int, world: int, rank: int):
    """Contiguous block of instance indices owned by ``rank`` (last ranks may own fewer)."""
    if world < 1 or not (0 <= rank < world) or num_instances < 0:
        raise ValueError("bad shard arguments")
    per = -(-num_instances // world) if num_instances else 0
    lo = min(rank * per, num_instances)
    hi = min(lo + per, num_instances)
    return lo, hi


def per_rank_capacity(num_instances: int, world: int) -> int:
    return -(-num_instances // world) if num_instances else 0


def gather_residual_norms(local_norms, num_instances: int, group=None):
    """All-gather the ranks' residual norms into one vector of length ``num_instances``.

    ``local_norms``: 1-d float64 torch tensor (device tensor for nccl, CPU tensor for gloo)
    holding this rank's norms in shard order.  Returns a tensor of all norms on every rank.
    Single process (no initialised group): returns ``local_norms`` unchanged.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local_norms[:num_instances]
    world = dist.get_world_size(group)
    cap = per_rank_capacity(num_instances, world)
    send = torch.zeros(cap, dtype=torch.float64, device=local_norms.device)
    send[: local_norms.numel()] = local_norms
    out = torch.empty(cap * world, dtype=torch.float64, device=local_norms.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if out.is_cuda:  # the solver's queue should not run beside the collective's
        torch.cuda.current_stream(out.device).synchronize()
    # drop the padding of short shards
    keep = []
    for r in range(world):
        lo, hi = shard_range(num_instances, world, r)
        keep.append(out[r * cap : r * cap + (hi - lo)])
    return torch.cat(keep) if keep else out[:0]


class BatchedDeviceNewton:
    """This rank's shard of a batch of linear-quadratic instances, all resident in HBM.

    ``make_problem(i)`` builds instance ``i`` (all of one shape).  The shard's instances are
    grouped into ONE device batch (``pgf_batch_*``): every kernel of the Newton step runs
    over all local instances at once, with no host round trip inside a step.  ``step()``
    advances every local instance by one Newton step and returns the gathered residual
    norms of the whole batch (one all-gather, nothing else crosses ranks).
    ``sequential=True`` drives the instances one after another through their own handles
    instead (the same kernels without the batch dimension; kept as the cross-check).
    """

    def __init__(self, make_problem, num_instances, newton_type, dt, rho, device=0, rank=0,
                 world=1, group=None, tau=None, sequential=False):
        import ctypes as C
        import math

        import torch

        from . import _lib
        from .newton import _POLICY_BITS, DeviceNewton
        from .params import enum_name

        self.num_instances = num_instances
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = shard_range(num_instances, world, rank)
        self.count = self.hi - self.lo
        self.kind = enum_name(newton_type) if not isinstance(newton_type, str) else newton_type
        self.policy = _POLICY_BITS[self.kind]
        self.tau = math.nan if tau is None else float(tau)
        self.dt, self.rho = float(dt), float(rho)
        self.sequential = bool(sequential)
        self.solvers = []
        for i in range(self.lo, self.hi):
            prob = make_problem(i)
            x0, y0 = np.zeros(prob.num_vars), np.zeros(prob.num_cons)
            self.solvers.append(DeviceNewton(prob, newton_type, x0, y0, dt, rho, tau=tau,
                                             device=device))
        self._torch = torch
        self._lib, self._C = _lib, C
        self.device = torch.device("cuda", device)
        self.norms = torch.zeros(max(1, self.count), dtype=torch.float64, device=self.device)
        self._b = None
        self._frozen = None
        self._outer_points = [s.point() for s in self.solvers] if self.sequential else []
        if self.count and not self.sequential:
            shapes = {(s.n, s.m, s.sparse) for s in self.solvers}
            if len(shapes) != 1 or next(iter(shapes))[2]:
                raise ValueError("a device batch needs dense instances of one (n, m)")
            self.n, self.m = self.solvers[0].n, self.solvers[0].m
            lib = _lib.load()
            arr = (C.c_void_p * self.count)(*[s._hd.h for s in self.solvers])
            b = C.c_void_p()
            rc = lib.pgf_batch_create(arr, self.count, C.byref(b))
            _lib.check(rc, self.solvers[0]._hd.h, "pgf_batch_create")
            self._b = b
            self._begin_outer()

    def _begin_outer(self):
        lib, b = self._lib.load(), self._b
        self._lib.check(lib.pgf_batch_advance_outer(b, self.dt, self.rho), batch=b,
                        what="pgf_batch_advance_outer")
        if self.kind == "Simplified":
            self._lib.check(lib.pgf_batch_update_active_set(b, self.tau), batch=b,
                            what="pgf_batch_update_active_set")

    def advance_outer(self, dt=None, rho=None):
        self.dt = self.dt if dt is None else float(dt)
        self.rho = self.rho if rho is None else float(rho)
        if self._b is None:
            self._frozen = None
            for k, s in enumerate(self.solvers):
                self._outer_points[k] = s.point()
                s.advance_outer(self.dt, self.rho)
        else:
            self._begin_outer()

    def advance_outer_each(self, dt, rho, accepted=None):
        """New outer step with per-instance ``dt[i]``, ``rho[i]``; ``accepted[i]`` False sends
        instance i back to its outer point (a rejected step) before it retries."""
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        rho = np.ascontiguousarray(np.broadcast_to(rho, dt.shape), dtype=np.float64)
        if dt.shape != (self.count,):
            raise ValueError("one dt per local instance")
        if self._b is None:
            for k, s in enumerate(self.solvers):
                if accepted is not None and not accepted[k]:
                    s.set_point(*self._outer_points[k])
                self._outer_points[k] = s.point()
                s.advance_outer(float(dt[k]), float(rho[k]))
            return
        acc = None
        if accepted is not None:
            acc = np.ascontiguousarray(accepted, dtype=np.bool_)
        lib, b = self._lib.load(), self._b
        self._lib.check(lib.pgf_batch_advance_outer_each(
            b, self._lib.dptr(dt), self._lib.dptr(rho), self._lib.u8ptr(acc)), batch=b,
            what="pgf_batch_advance_outer_each")
        if self.kind == "Simplified":
            self._lib.check(lib.pgf_batch_update_active_set(b, self.tau), batch=b,
                            what="pgf_batch_update_active_set")

    def set_frozen(self, frozen=None):
        """Instances with ``frozen[i]`` True sit out the following Newton steps (until the next
        ``advance_outer*``).  Device batch only."""
        if self._b is None:
            self._frozen = None if frozen is None else np.array(frozen, dtype=bool)
            return
        fz = None if frozen is None else np.ascontiguousarray(frozen, dtype=np.bool_)
        self._lib.check(self._lib.load().pgf_batch_set_frozen(self._b, self._lib.u8ptr(fz)),
                        batch=self._b, what="pgf_batch_set_frozen")

    def residual_norms_local(self):
        """Unscaled residual norms of the local instances at their current points (host)."""
        if self._b is None:
            return np.array([s.residual_norm() for s in self.solvers])
        out = np.empty(self.count)
        self._lib.check(self._lib.load().pgf_batch_residual_norms(self._b, self._lib.dptr(out),
                                                                   None), batch=self._b,
                        what="pgf_batch_residual_norms")
        return out

    def step_local(self):
        """One Newton step of every local instance; returns (status, n_neg, diff) arrays.
        The norms of the new points are left in ``self.norms`` (device)."""
        C = self._C
        cnt = self.count
        status = np.zeros(cnt, dtype=np.int32)
        n_neg = np.zeros(cnt, dtype=np.int32)
        diff = np.zeros(cnt)
        if cnt == 0:
            return status, n_neg, diff
        if self._b is None:
            from .errors import StepSolverError

            base = self.norms.data_ptr()
            frozen = getattr(self, "_frozen", None)
            for k, s in enumerate(self.solvers):
                if frozen is not None and frozen[k]:
                    continue
                try:
                    diff[k], n_neg[k] = s.step()
                except StepSolverError:
                    status[k] = self._lib.PGF_SINGULAR
                    continue
                s.residual_norm(base + 8 * k)
            return status, n_neg, diff
        lib, b = self._lib.load(), self._b
        self._lib.check(lib.pgf_batch_step_async(b, self.policy, self.tau), batch=b,
                        what="pgf_batch_step_async")
        ip = C.POINTER(C.c_int)
        self._lib.check(lib.pgf_batch_sync(b, status.ctypes.data_as(ip), n_neg.ctypes.data_as(ip),
                                           self._lib.dptr(diff)), batch=b, what="pgf_batch_sync")
        self._lib.check(lib.pgf_batch_residual_norms(b, None, self.norms.data_ptr()), batch=b,
                        what="pgf_batch_residual_norms")
        return status, n_neg, diff

    def step(self):
        self.step_local()
        return gather_residual_norms(self.norms[: self.count], self.num_instances, self.group)

    def points(self):
        """(x[count][n], y[count][m]) of the local instances (host copies)."""
        if self._b is None:
            pts = [s.point() for s in self.solvers]
            return np.array([p[0] for p in pts]), np.array([p[1] for p in pts])
        x = np.empty((self.count, self.n))
        y = np.empty((self.count, self.m))
        self._lib.check(self._lib.load().pgf_batch_get_points(self._b, self._lib.dptr(x),
                                                               self._lib.dptr(y)), batch=self._b)
        return x, y

    def masks(self):
        if self._b is None:
            return np.array([s.mask() for s in self.solvers])
        mk = np.empty((self.count, self.n), dtype=np.bool_)
        self._lib.check(self._lib.load().pgf_batch_get_masks(self._b, self._lib.u8ptr(mk)),
                        batch=self._b)
        return mk

    def measures(self, active_tol=1e-8):
        """[count][4] array: stat_res, cons_violation, bound_violation, ||y||_inf per local
        instance (``Iterate.stat_res`` etc., evaluated on device)."""
        if self._b is None:
            rows = [s.measures(active_tol) for s in self.solvers]
            return np.array([[r["stat_res"], r["cons_violation"], r["bound_violation"], r["y_inf"]]
                             for r in rows]).reshape(self.count, 4)
        out = np.empty((self.count, 4))
        self._lib.check(self._lib.load().pgf_batch_measures(self._b, float(active_tol),
                                                            self._lib.dptr(out)), batch=self._b)
        return out

    # -- device-resident step controller (pgf_batch_ctl_*) ------------------------------
    def ctl_init(self, params, rho=None, max_iterations=64, lamb_init=None):
        """Arm the device-resident ``DistanceRatioController`` of every local instance."""
        if self._b is None:
            raise NotImplementedError("device batch only")
        vals = np.array([params.newton_tol, params.lamb_red, params.lamb_min, params.lamb_inc,
                         params.theta_max, params.K_P, params.K_I, params.theta_ref], dtype=np.float64)
        rho = self.rho if rho is None else float(rho)
        lamb = float(params.lamb_init if lamb_init is None else lamb_init)
        self._ctl_cap = int(max_iterations)
        self._lib.check(self._lib.load().pgf_batch_ctl_init(self._b, lamb, rho, self._lib.dptr(vals),
                                                            self._ctl_cap), batch=self._b,
                        what="pgf_batch_ctl_init")

    def ctl_iterate(self, iterations):
        """Enqueue ``iterations`` outer iterations (two batched Newton steps + the controller's
        decisions each) without a host synchronisation in between."""
        self._lib.check(self._lib.load().pgf_batch_ctl_iterate(self._b, self.policy, self.tau,
                                                               int(iterations)), batch=self._b,
                        what="pgf_batch_ctl_iterate")

    def ctl_read(self):
        """Wait; returns (lamb_next[count], accepted[count], log[iterations][count][3]) with
        log rows (lambda used, lambda next, accepted) per outer iteration so far."""
        lamb = np.empty(self.count)
        acc = np.empty(self.count, dtype=np.bool_)
        log = np.zeros((self._ctl_cap, self.count, 3))
        self._lib.check(self._lib.load().pgf_batch_ctl_read(
            self._b, self._lib.dptr(lamb), self._lib.u8ptr(acc), self._lib.dptr(log), self._ctl_cap),
            batch=self._b, what="pgf_batch_ctl_read")
        return lamb, acc, log

    def repaired(self):
        """Instances whose step the accuracy guard repaired inside a batched step so far
        (``pgf_batch_refinement_stats``): sampled residual failed, the instance's handle refined
        or fell back to the pivoted LU."""
        C = self._C
        k = C.c_int(0)
        if self._b is not None:
            self._lib.check(self._lib.load().pgf_batch_refinement_stats(self._b, C.byref(k)), batch=self._b)
        return k.value

    def profile(self, on=True):
        if self._b is not None:
            self._lib.check(self._lib.load().pgf_batch_profile_enable(self._b, int(on)),
                            batch=self._b)

    def profile_read(self):
        """Device time / launches / algorithmic flops of the trailing-update launches."""
        C = self._C
        ms, cnt, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        if self._b is not None:
            self._lib.check(self._lib.load().pgf_batch_profile_read(
                self._b, C.byref(ms), C.byref(cnt), C.byref(fl)), batch=self._b)
        return dict(update_ms=ms.value, update_launches=cnt.value, update_flops=fl.value)

    def close(self):
        if self._b is not None:
            self._lib.load().pgf_batch_destroy(self._b)
            self._b = None
        for s in self.solvers:
            s.close()
        self.solvers = []
