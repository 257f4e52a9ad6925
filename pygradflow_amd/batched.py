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

    ``make_problem(i)`` builds instance ``i``; ``step()`` advances every local instance by
    one Newton step and returns the gathered residual norms of the whole batch.
    """

    def __init__(self, make_problem, num_instances, newton_type, dt, rho, device=0, rank=0,
                 world=1, group=None):
        import torch

        from .newton import DeviceNewton

        self.num_instances = num_instances
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = shard_range(num_instances, world, rank)
        self.solvers = []
        for i in range(self.lo, self.hi):
            prob = make_problem(i)
            x0, y0 = np.zeros(prob.num_vars), np.zeros(prob.num_cons)
            self.solvers.append(DeviceNewton(prob, newton_type, x0, y0, dt, rho, device=device))
        self._torch = torch
        self.device = torch.device("cuda", device)
        self.norms = torch.zeros(max(1, self.hi - self.lo), dtype=torch.float64, device=self.device)

    def advance_outer(self):
        for s in self.solvers:
            s.advance_outer()

    def step(self):
        base = self.norms.data_ptr()
        for k, s in enumerate(self.solvers):
            s.step()
            s.residual_norm(base + 8 * k)
        return gather_residual_norms(self.norms[: self.hi - self.lo], self.num_instances, self.group)

    def close(self):
        for s in self.solvers:
            s.close()
        self.solvers = []
