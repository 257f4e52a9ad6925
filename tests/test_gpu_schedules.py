"""Every schedule switch of the dense factorisation gives the same factors (GPU).

The switches are read once per process, so each variant runs tools/check_factor.py --no-time in
a child process: HipLinearSolver against numpy on quasi-definite matrices of awkward sizes
(1 ... 2561), solution, inertia and |L D L' - K| checked there."""

import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {
    "default": {},
    "no_helpers": {"PGF_CHAIN_HELP": "0"},
    "chain_8_wavefronts": {"PGF_CHAIN_WAVES": "8"},
    "unfused_launches": {"PGF_FUSED": "0"},
    "separate_update_diag": {"PGF_FUSED_UD": "0"},
    "eager_update_plan": {"PGF_LAZY_BUDGET": "0"},
    "tight_update_budget": {"PGF_LAZY_BUDGET": "40", "PGF_LAZY_CAP": "1"},
    "legacy_schedule": {"PGF_FACTOR": "1"},
    "solves_per_super_block": {"PGF_TRSV_CHAIN": "0"},
    "register_resident_chain": {"PGF_CHAIN": "3"},  # csrc/pgf_chain3.h (experimental, not the default)
}


BATCH_VARIANTS = {
    "fused_lookahead": {},                          # default for small batches
    "chain_per_block": {"PGF_BATCH_FUSED_MAX": "0"},  # what larger batches run
    "split_panel_steps": {"PGF_BATCH_CHAIN": "0"},    # the round-1 schedule
    "solves_per_super_block": {"PGF_TRSV_CHAIN": "0"},  # batched solves without the chained kernels
    "condensed_order": {"PGF_CONDENSED": "2"},        # constraint block first (what n=1024, m=256 runs by default)
    "natural_order": {"PGF_CONDENSED": "0"},
    "condensed_chain_per_block": {"PGF_CONDENSED": "2", "PGF_BATCH_FUSED_MAX": "0"},  # large batches
}


BAND_VARIANTS = {
    "paired_levels": {},                              # default: two levels per launch where launch-bound
    "one_level_per_launch": {"PGF_BCR_PAIRS": "0"},
    "separate_invert_reduce": {"PGF_BCR_FUSED": "0"},
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(BAND_VARIANTS))
def test_banded_schedule_variant_matches_oracle(gpu_available, name):
    """tests/check_band.py: banded OCP and tridiagonal box QP (churning mask) against the CPU
    oracle under each cyclic-reduction schedule."""
    if not gpu_available:
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    env.update(BAND_VARIANTS[name])
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "check_band.py")], env=env,
                         cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "band ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(BATCH_VARIANTS))
def test_batched_schedule_variant_matches_one_by_one(gpu_available, name):
    """tools/check_batch.py: a device batch of 11 instances with different reduced sizes against
    the same instances driven one by one, under each batched factorisation schedule."""
    if not gpu_available:
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    env.update(BATCH_VARIANTS[name])
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_batch.py")], env=env,
                         cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "batch ok" in out.stdout


CONDENSED_VARIANTS = {
    "condensed_forced": {"PGF_CONDENSED": "2"},     # constraint block eliminated first at every size
    "natural_order": {"PGF_CONDENSED": "0"},        # the reduced KKT matrix as it stands
    "condensed_unfused": {"PGF_CONDENSED": "2", "PGF_FUSED": "0"},
    "condensed_tight_budget": {"PGF_CONDENSED": "2", "PGF_LAZY_BUDGET": "40", "PGF_LAZY_CAP": "1"},
    "condensed_no_helpers": {"PGF_CONDENSED": "2", "PGF_CHAIN_HELP": "0"},
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CONDENSED_VARIANTS))
def test_condensed_factorisation_matches_oracle(gpu_available, name):
    """tools/check_condensed.py: Full / Simplified / ActiveSet steps on boxed dense QPs whose
    constraint block spans zero to three 256-column blocks (ragged), against the CPU oracle, with
    the pivot order the variant asks for (checked through pgf_debug_factor_kind), no refinement and
    inertia m."""
    if not gpu_available:
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    env.update(CONDENSED_VARIANTS[name])
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_condensed.py")], env=env,
                         cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "condensed ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed,condensed", [(0, "2"), (1, "2"), (2, "0"), (3, "1")])
def test_random_shapes_against_oracle(gpu_available, seed, condensed):
    """tools/check_random.py: 24 boxed dense QPs of random shape (n around the 256-column block
    boundaries, m from 0 to 600, boxed share, box width, dt from 0.1 to 1000, rho, policy) through
    DeviceNewton against the CPU oracle: masks bit for bit, iterates to 1e-10, inertia m, never the
    LU -- with the condensed pivot order wherever its bounds allow (PGF_CONDENSED=2), never (0) and
    by the default rule (1)."""
    if not gpu_available:
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    env["PGF_CONDENSED"] = condensed
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_random.py"), str(seed), "24"],
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "random ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed,condensed", [(0, "2"), (1, "2"), (0, "0")])
def test_random_batches_against_one_by_one(gpu_available, seed, condensed):
    """tools/check_batch_random.py: device batches of random shape and size with per-instance dt and
    rho, rejected instances and all three policies against the same instances driven one by one."""
    if not gpu_available:
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    env["PGF_CONDENSED"] = condensed
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_batch_random.py"), str(seed)],
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "batch random ok" in out.stdout


@pytest.mark.gpu
def test_golden_replays_in_the_condensed_order(gpu_available):
    """The reference's recorded steps once more with the constraint block eliminated first wherever
    the growth bound allows (PGF_CONDENSED=2; the default mode only picks it from m = 64 and a saved
    column block, which none of the small golden cases has): every golden replay, the free-running
    policies, the device-resident driver, the hard-regime and ill-conditioned fixtures (where the
    forward-error bound of condensed_growth_ok must send the cond >= 1e7 systems back to the natural
    order), the inertia / singular-matrix error paths and the device batches of
    tests/test_gpu_parity.py, in a child process (the switch is read once per process)."""
    if not gpu_available:
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    env["PGF_CONDENSED"] = "2"
    # (not test_unstable_pivot_*: its tiny pivot is one of the NATURAL order; eliminated after the
    # constraint block the same matrix needs no repair, which is not what that test is about)
    sel = ("replays_golden or free_running or device_resident or hard_regime or illconditioned or inertia "
           "or singular_kkt or device_batch_matches or device_batch_against")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_parity.py"),
                          "-x", "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"],
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and " failed" not in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_schedule_variant_factorises_correctly(gpu_available, name):
    if not gpu_available:
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    env.update(VARIANTS[name])
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_factor.py"), "--no-time"],
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "linear solver ok" in out.stdout
