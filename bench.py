#!/usr/bin/env python3
"""Newton steps/sec on the dense KKT hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

One "step" = one Full Newton step (NewtonType.Full, StepSolverType.Symmetric,
SURVEY.md 8d): evaluate c, g at the device point -> active-set mask -> gather-assemble
K -> LDL^T factor -> solve -> update + clip.  Problem data (Q, A, q, b) and the iterate
are resident in HBM before the timed region starts.  Every second step begins a new
outer (implicit Euler) step on device, as the reference's default DistanceRatio
controller does (2 Newton steps per outer iteration), so each step does real work.

N > 1: one process per GPU (launched by torch.distributed.run); each rank owns an
independent instance of the same size (seed = rank): weak scaling over instances, and
per step ONE RCCL all-gather of the ranks' residual norms ||F(z)||_2.

Prints one JSON line (rank 0) with `roofline` (dominant launch = k_chain_update, the
diagonal chain beside the FP64-MFMA trailing update, timed with HIP events on the solver's
stream around the PRODUCTION launches in an instrumented pass over the same steps; the
per-kernel view of the unfused schedule rides along under its own key) and `cpu_baseline`
(the numpy/scipy restatement of the reference path, oracle/, timed on the host cores; kind
"port").  At --gpus 1 the line also carries BASELINE configs[3] on one GPU (`batch256`: all
256 instances; `shard32`: the 32 instances one rank of an 8-GPU run holds) and `plugin_step`:
the drop-in path (HipStepSolver under newton_method, host callbacks, PCIe uploads).
"""

import argparse
import json
import gc
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# FP64 matrix (MFMA) dense peak of MI355X.  MI355X_MICROARCH.md lists no FP64 row; the
# vendor figure is 78.6 TFLOP/s = 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz.
PEAK_FP64_MFMA_TFLOPS = 78.6

WORKLOADS = {
    "dense_qp_n4096_m1024": dict(n=4096, m=1024, batch=1),   # BASELINE configs[1]
    "dense_qp_n1024_m256": dict(n=1024, m=256, batch=1),     # element of configs[3]
    "dense_qp_n512_m128": dict(n=512, m=128, batch=1),       # quick check
    # BASELINE configs[3]: 256 instances sharded over the ranks (32 per GPU at 8 GPUs);
    # a "step" is one Newton step of EVERY instance + one all-gather of 256 norms
    "batch256_n1024_m256": dict(n=1024, m=256, batch=256),
    # BASELINE configs[2]: sparse optimal-control NLP, banded path (CSR + banded LDL^T)
    "sparse_ocp_n100000_m50000": dict(n=100000, m=50000, batch=1, sparse=True),
    # BASELINE configs[4] as given: tridiagonal box QP, 50 % of the bounds active at the start
    "box_qp_n16384": dict(n=16384, m=0, batch=1, sparse=True, box=True),
    # variant 5b: the same box QP with a DENSE Q (2 GiB): the H[I, I] gather is
    # HBM-bound and the reduced system (|I| ~ 8192) goes through the dense factorisation; the
    # CPU restatement would spend minutes in one sparse LU of a dense 8192^2 matrix, so this
    # workload reports no cpu_baseline
    "box_qp_dense_n16384": dict(n=16384, m=0, batch=1, box_dense=True, no_cpu=True),
}

# HBM peak of MI355X (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.3 TB/s achievable)
PEAK_HBM_GBS = 8000.0


def cpu_baseline(problem, max_seconds):
    """One Full Newton step of the CPU restatement (same bmat + splu calls as the
    reference) from x0 = y0 = 0; returns (steps/s, record)."""
    from oracle import newton_oracle as O

    n, m = problem.num_vars, problem.num_cons
    t0 = time.perf_counter()
    orc = O.NewtonOracle(problem, "Full", np.zeros(n), np.zeros(m), 1.0, 1.0)
    recs = []
    x, y = np.zeros(n), np.zeros(m)
    steps = 0
    while True:
        x, y, _ = orc.step(x, y)
        recs.append(dict(orc.solver.record))
        steps += 1
        el = time.perf_counter() - t0
        if el > max_seconds or steps >= 20:
            break
    return steps / el, recs, steps, el


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc
    passes (profiles/pmc_traffic.json, produced by tools/pmc_summary.py with the gfx950
    FETCH_SIZE x2 correction of MI355X_MICROARCH.md); None if no such measurement."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
        return rec.get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def bench_batched(args, wl, rank, local_rank, world, dist, torch, emit=True, steps=None, warmup=None):
    """BASELINE configs[3]: B independent instances, contiguous shards per rank; every rank
    advances its shard as ONE device batch (pgf_batch_*), one all-gather of norms per step."""
    from pygradflow_amd import problems
    from pygradflow_amd.batched import BatchedDeviceNewton

    n, m, B = wl["n"], wl["m"], wl["batch"]
    nsteps = args.steps if steps is None else steps
    nwarm = args.warmup if warmup is None else warmup
    bd = BatchedDeviceNewton(lambda i: problems.dense_qp(n, m, seed=i), B, "Full", 1.0, 1.0,
                             device=local_rank, rank=rank, world=world)
    dev = torch.device("cuda", local_rank)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def one_step(i):
        if i % 2 == 0 and i > 0:
            bd.advance_outer()
        return bd.step()

    # parity of instance 0's first step against the CPU restatement (rank 0, bounded)
    parity = cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and emit:
        rate, recs, csteps, cel = cpu_baseline(problems.dense_qp(n, m, seed=0), args.cpu_seconds)
        bd.step_local()
        xg, yg = bd.points()
        mk = bd.masks()
        rel = lambda a, b: float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if b.size else 0.0
        # every local instance's mask and the iterates of 8 of them against the CPU restatement
        from oracle import newton_oracle as O
        sample = sorted(set(np.linspace(0, bd.hi - bd.lo - 1, 8).astype(int).tolist()))
        ham, worst = 0, 0.0
        for i in range(bd.hi - bd.lo):
            pi = problems.dense_qp(n, m, seed=bd.lo + i)
            if i == 0:
                ri = recs[0]
            elif i in sample:
                orc = O.NewtonOracle(pi, "Full", np.zeros(n), np.zeros(m), 1.0, 1.0)
                orc.step(np.zeros(n), np.zeros(m))
                ri = orc.solver.record
            else:
                # unbounded variables: the reference's mask is empty whatever the point
                ri = dict(mask=np.zeros(n, dtype=bool)) if not pi.var_bounded else None
            if ri is None:
                continue
            ham += int(np.count_nonzero(mk[i] != ri["mask"]))
            if "xn" in ri:
                worst = max(worst, rel(xg[i], ri["xn"]), rel(yg[i], ri["yn"]))
        parity = dict(instances_masks_checked=bd.hi - bd.lo, instances_iterates_checked=len(sample),
                      mask_hamming=ham, iterate_rel_err=worst)
        cpu = dict(value=rate, unit="instance Newton steps/s", cores=1, kind="port",
                   sample=f"{csteps} Full Newton step(s) of ONE n={n} m={m} instance from x0=y0=0 "
                          f"(scipy bmat + SuperLU splu as the reference calls them; the "
                          f"reference runs instances in a process pool, so scale by the cores "
                          f"used), {cel:.1f} s; host has {os.cpu_count()} cores")
    for i in range(nwarm):
        one_step(i)
    fence()
    # (the instance generators leave garbage behind: a cyclic collection inside the timed loop
    # stalls ONE batched step by tens of milliseconds)
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(nsteps):
        norms = one_step(i)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # EVERY rank walks the same steps (each one ends in the all-gather); only rank 0 records
    roof = None
    if rank == 0:
        bd.profile(True)
    for i in range(nsteps):
        one_step(i)
    if rank == 0:
        pr = bd.profile_read()
        bd.profile(False)
        if pr["update_launches"] > 0 and pr["update_ms"] > 0:
            achieved = pr["update_flops"] / (pr["update_ms"] * 1e-3) / 1e12
            roof = dict(bound="mfma", kernel="kb_ldlt_update", achieved=achieved,
                        peak=PEAK_FP64_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=achieved / PEAK_FP64_MFMA_TFLOPS, traffic=pmc_traffic("kb_ldlt_update"),
                        launches_per_step=pr["update_launches"] / nsteps,
                        avg_launch_us=1e3 * pr["update_ms"] / pr["update_launches"],
                        flops_per_step=pr["update_flops"] / nsteps,
                        step_flops=(bd.hi - bd.lo) * ((n + m) ** 3 / 3.0 + 2.0 * (n + m) ** 2))
            # step level: 1.80e11 flop per batched step over all ranks
            sf = B * ((n + m) ** 3 / 3.0 + 2.0 * (n + m) ** 2 + 2.0 * n * n + 4.0 * n * m)
            roof["step_achieved"] = sf * (nsteps / elapsed) / 1e12
            roof["step_frac"] = roof["step_achieved"] / (PEAK_FP64_MFMA_TFLOPS * world)
    record = None
    if rank == 0:
        assert norms.numel() == B
        record = {
            "metric": f"Newton steps/sec on dense KKT n={n} m={m}, batch of {B} instances",
            "value": nsteps * B / elapsed, "unit": "instance Newton steps/s",
            "n_gpus": world, "steps": nsteps, "warmup": nwarm,
            "ms_per_step": 1e3 * elapsed / nsteps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batch{B}_n{n}_m{m}", "n": n, "m": m, "instances": B,
                       "instances_per_gpu": bd.hi - bd.lo, "newton_type": "Full",
                       "path": "device batch (pgf_batch_*, instance = XCD-pinned workgroup range)",
                       "collective": f"all_gather({B} residual norms)" if world > 1 else "none"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
        }
    bd.close()
    if not emit:
        return record
    if rank == 0:
        print(json.dumps(record), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_plugin(problem, x0, y0, device, steps=6):
    """The drop-in path at config 2: the reference's own hook, Params(step_solver=HipStepSolver)
    (params.py:234, step/solver/__init__.py:18-19), driven by newton_method / FullNewtonMethod
    (newton.py:63-89): g, c, the derivatives and the active set come from HOST callbacks every
    step, H and J cross PCIe whenever update_derivs sees new matrices.  ms per Full Newton step:
      constant_derivs   linear-quadratic problem, H / J uploaded once and kept resident
      dense_upload      the same matrices treated as fresh every step (168 MB over PCIe per step)
      csr_upload        a 1 %-dense sparse H and J: 12 bytes per stored entry, densified on the device
    """
    import scipy.sparse as sps

    import pygradflow_amd as pgf
    from pygradflow_amd import problems

    n, m = problem.num_vars, problem.num_cons

    def run(prob, label):
        params = pgf.Params(newton_type="Full", step_solver=pgf.HipStepSolver)
        orig = pgf.Iterate(prob, params, x0, y0)
        method = pgf.newton_method(prob, params, orig, 1.0, 1.0)
        it = orig
        res = method.step(it)  # warm-up: first upload, allocations
        it = res.iterate
        t0 = time.perf_counter()
        for _ in range(steps):
            res = method.step(it)
            it = res.iterate
        ms = 1e3 * (time.perf_counter() - t0) / steps
        try:
            method.step_solver.close()
        except Exception:
            pass
        return dict(ms_per_step=ms, steps=steps, what=label)

    out = {}
    try:
        out["constant_derivs"] = run(problem, "H, J resident in HBM after the first step; per step: host "
                                              "g = Qx+q+A'(rho c+y), c = Ax-b, mask and vectors over PCIe")
        fresh = problems.LinearQuadraticProblem(problem.Q, problem.q, problem.A, problem.b,
                                                problem.var_lb, problem.var_ub)
        fresh.pgf_constant_derivs = False
        out["dense_upload"] = run(fresh, f"dense H ({8e-6 * n * n:.0f} MB) and J ({8e-6 * n * m:.0f} MB) "
                                         "uploaded every step (pgf_set_derivs_dense)")
        rng = np.random.default_rng(7)
        Qs = sps.random(n, n, density=0.005, random_state=rng, format="csr")
        Qs = (Qs + Qs.T + 4.0 * sps.identity(n)).tocsr()
        As = sps.random(m, n, density=0.01, random_state=rng, format="csr")
        sp = problems.LinearQuadraticProblem(Qs, problem.q, As, problem.b, problem.var_lb, problem.var_ub)
        sp.pgf_constant_derivs = False
        out["csr_upload"] = run(sp, f"sparse H ({Qs.nnz} entries, {100.0 * Qs.nnz / (n * n):.1f} %) and J "
                                    f"({As.nnz}) uploaded as CSR every step (pgf_set_derivs_csr), "
                                    "densified on the device")
    except Exception as e:  # the headline record stays valid without this extra
        out["error"] = f"{type(e).__name__}: {str(e)[:200]}"
    out["note"] = ("HipStepSolver under newton_method: the reference's plug-in boundary; host "
                   "evaluation and PCIe are inside these times, unlike `value`")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dense_qp_n4096_m1024", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the nested records of the default workload at --gpus 1 "
                         "(batch256, shard32, plugin_step)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world

    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PGF_BENCH_ONE_GPU=1 (rehearsal of the multi-rank control flow on a one-GPU box): all
        # ranks share cuda:0 and talk over gloo; never used for reported numbers
        if os.environ.get("PGF_BENCH_ONE_GPU"):
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    from pygradflow_amd import problems
    from pygradflow_amd.newton import DeviceNewton

    wl = WORKLOADS[args.workload]
    n, m = wl["n"], wl["m"]
    if wl["batch"] > 1:
        return bench_batched(args, wl, rank, local_rank, world, dist, torch)
    is_sparse = wl.get("sparse", False)
    if wl.get("box_dense"):
        problem = problems.box_qp(n, seed=rank, dense=True)
    elif wl.get("box"):
        problem = problems.box_qp(n, seed=rank)
        problem.pgf_force_band = True
    elif is_sparse:
        problem = problems.sparse_ocp(m, seed=rank)
    else:
        problem = problems.dense_qp(n, m, seed=rank)
    x0, y0 = np.zeros(n), np.zeros(m)
    dn = DeviceNewton(problem, "Full", x0, y0, 1.0, 1.0, device=local_rank)

    dev = torch.device("cuda", local_rank)
    norms_local = torch.zeros(1, dtype=torch.float64, device=dev)
    norms_all = torch.zeros(world, dtype=torch.float64, device=dev)

    def one_step(i):
        if i % 2 == 0 and i > 0:
            dn.advance_outer()
        # the step is enqueued, the residual norm of the new point is enqueued behind it, and
        # ONE host synchronisation (inside residual_norm) covers both; sync() then only reads
        # the factorisation's status words
        dn.step_async()
        dn.residual_norm(norms_local.data_ptr())
        dn.sync()
        if dist is not None:
            dist.all_gather_into_tensor(norms_all, norms_local)  # the one collective / step
            # keep one queue active at a time for the solver (look-ahead)
            torch.cuda.current_stream(dev).synchronize()

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # parity of the first step against the CPU restatement (rank 0 only, bounded)
    parity = None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not wl.get("no_cpu"):
        rate, recs, csteps, cel = cpu_baseline(problem, args.cpu_seconds)
        dn.step()
        xg, yg = dn.point()
        r0 = recs[0]
        def _rel(a, b):
            return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if b.size else 0.0

        parity = dict(
            mask_hamming=int(np.count_nonzero(dn.mask() != r0["mask"])),
            iterate_rel_err=max(_rel(xg, r0["xn"]), _rel(yg, r0["yn"])),
        )
        cpu = dict(value=rate, unit="Newton steps/s", cores=1, kind="port",
                   sample=f"{csteps} Full Newton step(s) of {args.workload} from x0=y0=0 "
                          f"(scipy bmat + SuperLU splu as the reference calls them; "
                          f"SuperLU is single-threaded), {cel:.1f} s; host has {os.cpu_count()} cores")
        dn.set_outer(x0, y0, 1.0, 1.0)
        # a FAIR dense CPU number beside the reference-shaped one, so that
        # the ratio is not only an artefact of the reference storing a dense matrix sparsely:
        # LAPACK LU (dgetrf / dgetrs through scipy, all cores) of the same assembled KKT
        # matrix -- factor + solve only, no assembly, no residual.  Dense workloads only.
        if not is_sparse and cpu is not None:
            try:
                import scipy.linalg as sla

                Kd = r0["K"].toarray() if hasattr(r0["K"], "toarray") else np.asarray(r0["K"])
                rhs_d = np.asarray(r0["rhs"], dtype=np.float64)
                # pinned thread count: with every hardware thread of a 256-core host the LU of
                # a 5120^2 matrix is oversubscribed (53 GFLOP/s in round 1); 16 threads = the
                # CPU share of one GPU on the bench box
                nthr = min(16, os.cpu_count() or 1)
                import contextlib

                try:
                    from threadpoolctl import threadpool_limits
                    limiter = threadpool_limits(limits=nthr, user_api="blas")
                except Exception:
                    limiter, nthr = contextlib.nullcontext(), os.cpu_count()
                with limiter:
                    sla.lu_solve(sla.lu_factor(Kd), rhs_d)  # warm up the BLAS threads
                    reps, tl = 0, time.perf_counter()
                    while reps < 5 and time.perf_counter() - tl < 10.0:
                        sla.lu_solve(sla.lu_factor(Kd), rhs_d)
                        reps += 1
                    dense_s = (time.perf_counter() - tl) / reps
                cpu["dense_lapack"] = dict(
                    value=1.0 / dense_s, unit="factor+solve/s", cores=nthr,
                    gflops=(2.0 / 3.0 * Kd.shape[0] ** 3) / dense_s / 1e9,
                    sample=f"{reps} x scipy.linalg.lu_factor + lu_solve of the {Kd.shape[0]}^2 KKT "
                           f"matrix of step 1 ({nthr} BLAS threads), {1e3 * dense_s:.1f} ms each")
                del Kd
            except Exception as e:  # the port number above stays valid without this extra
                cpu["dense_lapack"] = dict(value=None, error=str(e)[:200])

    for i in range(args.warmup):
        one_step(i)
    dn.set_outer(x0, y0, 1.0, 1.0)
    fence()
    trace = [] if os.environ.get("PGF_BENCH_TRACE") else None
    # the CPU-baseline leg leaves garbage behind (scipy factors, per-step records); a cyclic
    # collection in the middle of the timed loop was seen to stall ONE step by 60 ms, which
    # matters for the sub-millisecond banded workloads
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
        if trace is not None:
            trace.append(time.perf_counter())
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if trace is not None and rank == 0:  # diagnostic: wall time of every step (us)
        d = np.diff(np.array([t0] + trace)) * 1e6
        print("step us:", " ".join(f"{v:.0f}" for v in d), file=sys.stderr, flush=True)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # instrumented pass: HIP events around every trailing-update launch (same steps)
    # EVERY rank walks the same steps (each one ends in the all-gather); only rank 0 records
    roof = None
    dn.set_outer(x0, y0, 1.0, 1.0)
    if rank == 0:
        dn.profile(True)
    for i in range(args.steps):
        one_step(i)
    pr = pp = None
    if rank == 0:
        pr = dn.profile_read()
        dn.profile(False)
    if not is_sparse:
        # the PRODUCTION launches: a second instrumented pass, one span per launch -- walked by
        # every rank as well (a step ends in the all-gather: rank 0 alone would leave the others
        # behind in a different collective)
        dn.set_outer(x0, y0, 1.0, 1.0)
        if rank == 0:
            dn.profile(2)
        for i in range(args.steps):
            one_step(i)
        if rank == 0:
            pp = dn.profile_read()
            dn.profile(False)
    if rank == 0:
        if is_sparse and pr["update_launches"] > 0 and pr["update_ms"] > 0:
            # banded path: the dominant piece is the block-cyclic-reduction solve (one span =
            # extract + log2(N/8) invert/reduce levels + the back-substitution levels); the
            # "flops" slot of the profile carries its algorithmic bytes
            gbs = pr["update_flops"] / (pr["update_ms"] * 1e-3) / 1e9
            # algorithmic bytes of one STEP, from the workload itself: H and J as
            # CSR read once (12 bytes per stored entry), the band of the permuted KKT matrix
            # ((bw + 1) doubles per row) written once and read twice, ~12 vector passes
            # (43 MB for config 3, 3 MB for config 5 at their BASELINE sizes)
            Hs, Js = problem.hess_sparse(), problem.jac_sparse()
            bw = int(getattr(getattr(dn._hd, "plan", None), "bw", 1))
            Nk = n + m
            step_bytes = (12.0 * (Hs.nnz + (Js.nnz if m else 0)) + 3.0 * 8.0 * Nk * (bw + 1)
                          + 12.0 * 8.0 * Nk)
            step_gbs = step_bytes * (args.steps / elapsed) / 1e9
            roof = dict(bound="hbm", kernel="bcr_solve (k_bcr_extract/invert/reduce/back)",
                        achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS,
                        step_bytes=step_bytes, step_achieved=step_gbs,
                        step_frac=step_gbs / PEAK_HBM_GBS,
                        traffic=None, spans_per_step=pr["update_launches"] / args.steps,
                        avg_span_us=1e3 * pr["update_ms"] / pr["update_launches"],
                        bytes_per_span=pr["update_flops"] / pr["update_launches"],
                        note="~2 log2(N/8) dependent launches of a few microseconds each: "
                             "launch/latency-bound, not bandwidth-bound, at this size")
        elif pr["update_launches"] > 0 and pr["update_ms"] > 0:
            # per-kernel view: the UNFUSED schedule of an instrumented pass (diagonal chain and
            # trailing update as separate launches, HIP events around each)
            achieved_u = pr["update_flops"] / (pr["update_ms"] * 1e-3) / 1e12
            unfused = dict(
                kernel="k_update_jobs (= update role of k_chain_update as its own launch)",
                achieved=achieved_u, frac=achieved_u / PEAK_FP64_MFMA_TFLOPS,
                launches_per_step=pr["update_launches"] / args.steps,
                avg_launch_us=1e3 * pr["update_ms"] / pr["update_launches"],
                factor_ms_per_step=pr["factor_ms"] / args.steps,
                kernel_ms_per_step={"k_diag_chain": pr["chain_ms"] / args.steps,
                                    "k_update_jobs": pr["update_ms"] / args.steps,
                                    "k_trsm_block": pr["trsm_ms"] / args.steps,
                                    "k_update_diag": pr["udiag_ms"] / args.steps},
                note=("NOT the timed schedule: the factorisation's kernels as separate launches "
                      "with HIP events around them; in the timed region the diagonal chain and "
                      "the trailing update share one launch (k_chain_update: same tile code, "
                      "same job table), T(k) and the next diagonal block's update another "
                      "(k_trsm_ud)"))
            fused_ok = pp is not None and pp["fused_launches"] > 0 and pp["fused_ms"] > 0
            achieved = (pp["fused_flops"] / (pp["fused_ms"] * 1e-3) / 1e12) if fused_ok else achieved_u
            traffic = (pmc_traffic("k_chain_update")
                       if args.workload == "dense_qp_n4096_m1024" and fused_ok else None)
            alg_bytes = (pp["fused_bytes"] / pp["fused_launches"]) if fused_ok else None
            roof = dict(
                bound="mfma",
                kernel=("k_chain_update (production launch: the diagonal chain D(k+1) of one "
                        "workgroup beside the trailing-update tiles of the lazy plan)"
                        if fused_ok else unfused["kernel"]),
                achieved=achieved,
                peak=PEAK_FP64_MFMA_TFLOPS, unit="TFLOP/s", frac=achieved / PEAK_FP64_MFMA_TFLOPS,
                traffic=traffic,
                algorithmic_bytes_per_launch=alg_bytes,
                traffic_over_algorithmic=(traffic / alg_bytes) if (traffic and alg_bytes) else None,
                launches_per_step=(pp["fused_launches"] / args.steps) if fused_ok else None,
                avg_launch_us=(1e3 * pp["fused_ms"] / pp["fused_launches"]) if fused_ok else None,
                flops_per_step=(pp["fused_flops"] / args.steps) if fused_ok else None,
                trsm_ud_ms_per_step=pp["trsmud_ms"] / args.steps,
                first_chain_ms_per_step=pp["chain_ms"] / args.steps,
                factor_ms_per_step=pp["factor_ms"] / args.steps,
                unfused_instrumented_schedule=unfused,
                note=("achieved = algorithmic flops of the launches' update jobs / launch "
                      "durations (HIP events around every production launch in an instrumented "
                      "pass over the same steps); a launch lasts as long as the slower of its "
                      "two roles -- the one-workgroup diagonal chain or the update tiles"),
            )
            # the STEP against the FP64-MFMA roof -- algorithmic flops of one
            # Full Newton step (factor N^3/3, solves 2 N^2, residual 2 n^2 + 4 n m) x steps/s
            # reduced size: inactive variables + constraints (= n + m without active bounds)
            N_ = n - int(np.count_nonzero(dn.mask())) + m
            step_flops = N_ ** 3 / 3.0 + 2.0 * N_ ** 2 + 2.0 * n * n + 4.0 * n * m
            roof["reduced_size"] = N_
            roof["step_flops"] = step_flops
            roof["step_achieved"] = step_flops * (args.steps / elapsed) / 1e12
            roof["step_frac"] = roof["step_achieved"] / PEAK_FP64_MFMA_TFLOPS
            # where the step's time goes (instrumented pass, ms per step)
            parts = {"k_diag_chain": pr["chain_ms"], "k_update_jobs": pr["update_ms"],
                     "k_trsm_block": pr["trsm_ms"], "k_update_diag": pr["udiag_ms"]}
            dom = max(parts, key=parts.get)
            roof["time_dominant_kernel"] = dict(
                kernel=dom, ms_per_step=parts[dom] / args.steps, schedule="unfused instrumented pass",
                launches_per_step=(pr["chain_launches"] / args.steps if dom == "k_diag_chain" else None),
                note=("the factorisation's serial pivot chain: ONE workgroup per 256-column "
                      "block; in the production schedule it runs beside trailing-update tiles "
                      "inside one launch" if dom == "k_diag_chain" else None))

    # SURVEY.md 8d "reported separately": the back-solve step of the Simplified policy
    # (2nd+ Newton step of an outer iteration: residual, reduced rhs, forward + backward
    # triangular solves, update -- no assembly, no factorisation).  Outside the timed region.
    backsolve = None
    if rank == 0 and world == 1:
        dn.close()
        ds = DeviceNewton(problem, "Simplified", x0, y0, 1.0, 1.0, device=local_rank)
        ds.step()  # factorises
        for _ in range(3):
            ds.step()
        torch.cuda.synchronize(dev)
        tb = time.perf_counter()
        nbs = max(10, args.steps)
        for _ in range(nbs):
            ds.step()
        torch.cuda.synchronize(dev)
        bms = 1e3 * (time.perf_counter() - tb) / nbs
        backsolve = dict(ms_per_step=bms, steps_per_s=1e3 / bms, steps=nbs)
        dn = ds

    # BASELINE configs[3] (what north_star names for multi-GPU scaling): at N > 1 the same
    # launch also times the 256-instance batch, 256 / N instances per rank, one all-gather of
    # 256 residual norms per batched step -- a nested record, strong scaling over N
    batch_rec = shard_rec = plugin_rec = None
    if args.workload == "dense_qp_n4096_m1024" and not (world == 1 and args.no_extras):
        dn.close()
        dn = None
        nb_steps = min(args.steps, 8 if world > 1 else 20)
        batch_rec = bench_batched(args, WORKLOADS["batch256_n1024_m256"], rank, local_rank, world,
                                  dist, torch, emit=False, steps=nb_steps, warmup=2)
        if world == 1:
            # what ONE rank of the 8-GPU run of configs[3] holds: 32 instances on this GPU
            shard_rec = bench_batched(args, dict(n=1024, m=256, batch=32), rank, local_rank, world,
                                      dist, torch, emit=False, steps=min(args.steps, 40), warmup=3)
            if shard_rec is not None and batch_rec is not None:
                shard_rec["projected_8gpu_scaling_vs_1gpu_batch256"] = dict(
                    value=8.0 * shard_rec["value"] / batch_rec["value"],
                    note="PROJECTED: 8 x this rate / the 256-instance rate of one GPU; no "
                         "multi-GPU run behind it (the all-gather of 256 norms is not in it)")
            plugin_rec = bench_plugin(problem, x0, y0, local_rank)
    if rank == 0:
        total_steps = args.steps * world
        out = {
            "metric": ("Newton steps/sec on dense KKT n=4096 m=1024; iterate match <=1e-10"
                       if args.workload == "dense_qp_n4096_m1024" else
                       f"Newton steps/sec, {args.workload}; iterate match <=1e-10"),
            "value": total_steps / elapsed,
            "unit": "Newton steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "n": n, "m": m, "newton_type": "Full",
                       "step_solver": "Symmetric", "instances_per_gpu": 1,
                       "path": "banded (CSR + banded LDL^T)" if is_sparse else "dense LDL^T",
                       "collective": "all_gather(residual norms)" if world > 1 else "none"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
            "backsolve_step": backsolve,
            "batch256": batch_rec,
            "shard32": shard_rec,
            "plugin_step": plugin_rec,
        }
        print(json.dumps(out), flush=True)
    if dn is not None:
        dn.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
