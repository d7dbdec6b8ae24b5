"""Host-only invariants of the work plan (bundle_adjustment_solver_amd/csrc/
ba_plan.cpp replaces the reference's FinalizeParameters / SetProblemSize /
connectivity build, core/full_bundle_adjustment_solver.cpp:182-206,243-308,
668-700): landmark order is a permutation, pairs have exactly one writer
(SURVEY Q1), Schur super-runs / chunks / triple words / lane tables are
consistent, every triple is filed exactly once.  tests/cpp/plan_check.cpp is
compiled with g++ against the planner sources (no GPU, no HIP)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bundle_adjustment_solver_amd", "csrc")


@pytest.fixture(scope="module")
def plan_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("plan") / "plan_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", CSRC,
                    os.path.join(ROOT, "tests", "cpp", "plan_check.cpp"),
                    os.path.join(CSRC, "ba_plan.cpp"),
                    os.path.join(CSRC, "ba_dense_sched.cpp"), "-o", exe, "-pthread"], check=True)
    return exe


@pytest.mark.parametrize("args,env", [
    (["120", "30000", "5", "2"], {}),                       # stereo windows, C3/C4 structure: covisibility groups
    (["40", "3000", "9", "1"], {}),                         # mono, wide windows (C2 structure): 64-wide groups
    (["30", "40000", "4", "1"], {}),                        # groups larger than one workgroup piece
    (["120", "30000", "5", "2"], {"BA_NO_GROUPS": "1"}),    # the same through the super-runs only
    (["40", "3000", "9", "1"], {"BA_NO_GROUPS": "1"}),
    (["60", "8000", "5", "2"], {"BA_NO_INTERLEAVE": "1", "BA_NO_GROUPS": "1"}),  # plain locality order
    (["30", "400", "5", "2"], {"BA_SUP_CAP": "7", "BA_NO_GROUPS": "1"}),         # tiny runs
    (["200", "6000", "3", "3"], {"BA_SUP_CAP": "1000", "BA_NO_GROUPS": "1"}),    # runs ended by the slot / chunk limits
    (["200", "6000", "3", "3"], {}),                        # groups beside super-runs (groups below 24 stay in runs)
    (["60", "3000", "20", "1"], {}),                        # windows of 20 poses: 128-wide covisibility groups (11..20 poses)
    (["60", "3000", "20", "1"], {"BA_NO_SPLIT": "1"}),      # (the ungrouped rest on the global triple list)
    (["60", "3000", "20", "1"], {"BA_NO_GROUPS": "1"}),     # windows of 20 poses: landmarks split into pose-group classes
    (["60", "3000", "20", "1"], {"BA_NO_GROUPS": "1", "BA_NO_SPLIT": "1"}),  # the same on the global triple list
    (["60", "3000", "16", "2"], {}),                        # 16 stereo views: 32 pattern slots, 16 poses
    (["60", "3000", "20", "1", "15"], {}),                  # 20-pose windows with dropout: masked wide groups
    (["90", "2000", "37", "2"], {"BA_NO_GROUPS": "1"}),     # four pose groups per landmark, ten classes
    (["120", "30000", "5", "2"], {"BA_LIN_STEPS": "2"}),    # many k_lin_grp pieces per group
    (["120", "30000", "5", "2"], {"BA_NO_LINGRP": "1"}),    # groups for the Schur kernel only
    (["120", "30000", "5", "2", "15"], {}),                 # 15 % of the observations dropped: superset (masked) groups
    (["40", "3000", "9", "1", "25"], {}),                   # the same, mono with wide windows
    (["120", "30000", "5", "2", "15"], {"BA_NO_SUPERSET": "1"}),  # ... with exact groups only
])
def test_plan_invariants(plan_check, args, env):
    r = subprocess.run([plan_check] + args, env=dict(os.environ, **env),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout


@pytest.mark.parametrize("args", [["120", "30000", "5", "2"], ["90", "2000", "37", "2"],
                                  ["300", "100000", "5", "2"]])
def test_threaded_plan_equals_the_serial_plan(plan_check, args):
    """The planner's per-landmark passes run on host threads (csrc/ba_plan.cpp
    parallel_for; replaces reference :182-206, :243-308, :668-700).  Every array the
    kernels consume must be identical for 1 and 7 threads (checksum printed by
    tests/cpp/plan_check.cpp), and the invariants hold for both."""
    sums = []
    for nt in ("1", "7"):
        r = subprocess.run([plan_check] + args, env=dict(os.environ, BA_PLAN_THREADS=nt),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout
        sums.append([l for l in r.stdout.splitlines() if l.startswith("plan checksum")])
    assert sums[0] and sums[0] == sums[1], sums


def test_dense_level_schedule_executes_to_the_dense_solution(tmp_path):
    """csrc/ba_dense_sched.cpp (replaces the dense LDLT call, reference
    core/full_bundle_adjustment_solver.cpp:905): the level schedule executed with
    1x1 tiles on random SPD matrices of band / dense / random / disconnected tile
    patterns reproduces a plain dense Cholesky factor and solution."""
    exe = str(tmp_path / "dense_sched_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", CSRC,
                    os.path.join(ROOT, "tests", "cpp", "dense_sched_check.cpp"),
                    os.path.join(CSRC, "ba_dense_sched.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "DENSE SCHEDULE CHECK OK" in r.stdout, r.stdout[-3000:]
